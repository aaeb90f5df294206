#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats plus separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
tag=${1:-r01}; shift || true
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/bench_trace.json" 2> "$out/trace.err" || { echo "trace pass failed"; tail -5 "$out/trace.err"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- python3 bench.py $args > "$out/bench_$name.json" 2> "$out/pmc_$name.err" || { echo "pmc pass $pass failed"; tail -3 "$out/pmc_$name.err"; }
done
python3 tools/summarize_profile.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
# keep the merged-back payload small
find "$out" -name "*.csv" -size +8M -delete
