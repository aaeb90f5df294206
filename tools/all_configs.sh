#!/bin/bash
# Runs on the GPU box: one short bench line per configuration (value, ms per step, kernel times).
mkdir -p gpurun_out
for c in 1 2 3 4 5; do
  timeout -k 10 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cfg$c.json 2> gpurun_out/cfg$c.err || { echo "config $c failed"; tail -3 gpurun_out/cfg$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/cfg$c.json')); k=d['kernel_ms']; print('config $c', d['config']['workload'][:70], '|', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {a: round(b,1) for a,b in k.items()})"
done
