#!/bin/bash
# Runs on the GPU box: one --pmc pass over a short bench.py run with a device-library variant.
# Usage: tools/pmc_variant.sh <variant> <counters...>
set -u
v=$1; shift
out=gpurun_out/pmc_$v
mkdir -p "$out"
export TMPDIR=/tmp
export MIRT_DEVICE_LIB=libmirt_$v.so
[ "$v" = default ] && export MIRT_DEVICE_LIB=libmirt.so
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench.json" 2> "$out/err.txt"
python3 - "$out" <<'PY'
import csv,glob,collections,sys
tot=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/pmc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_extend<false>' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
print(sys.argv[1], {k:"%.4g"%v for k,v in tot.items()}, "wave_cycles/busy_cycles %.2f" % (tot.get('SQ_WAVE_CYCLES',0)/max(tot.get('SQ_BUSY_CYCLES',1),1)))
PY
