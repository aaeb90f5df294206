#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass over a short bench.py run.  Usage: tools/pmc_one.sh <tag> <counters...>
set -u
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench.json" 2> "$out/err.txt"
echo "exit $?"; wc -c "$out/bench.json"; grep -v "^[EWI]2026" "$out/err.txt" | tail -3
