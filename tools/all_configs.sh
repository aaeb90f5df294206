#!/bin/bash
# Runs on the GPU box: one short bench line per configuration (value, ms per step, kernel times).
mkdir -p gpurun_out
for c in 1 2 3 4 5; do
  timeout -k 10 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cfg$c.json 2> gpurun_out/cfg$c.err || { echo "config $c failed"; tail -3 gpurun_out/cfg$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/cfg$c.json')); k=d['kernel_ms']; print('config $c', d['config']['workload'][:70], '|', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {a: round(b,1) for a,b in k.items()})"
done
# the headline configuration on the second dragon-class geometry
timeout -k 10 400 python bench.py --config 3 --geometry suzanne --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cfg3_suzanne.json 2> gpurun_out/cfg3_suzanne.err || { echo "config 3 (suzanne) failed"; tail -3 gpurun_out/cfg3_suzanne.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/cfg3_suzanne.json')); k=d['kernel_ms']; r=d['roofline']['per_ray']; print('config 3 suzanne x4^5 |', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {a: round(b,1) for a,b in k.items()}, {a: round(b,2) for a,b in r.items()}, 'frac', round(d['roofline']['frac'],3))"
