#!/bin/bash
# A/B on one box: tools/ab_bench.sh <libA.so> <libB.so> ... -> extend ms per launch and Mrays/s for each, twice.
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    MIRT_DEVICE_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 3 ${AB_ARGS:-} > gpurun_out/ab_$lib.json 2>gpurun_out/ab_$lib.err || exit 1
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab_$lib.json')); n=d['roofline']['launches']; k=d['kernel_ms']; print('$lib', round(d['value'],1), 'Mrays/s  per launch: extend', round(k['extend']/n,3), 'shade', round(k['shade']/n,3), 'sample', round(k['sample']/n,3), 'ms')"
  done
done
