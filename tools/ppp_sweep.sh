#!/bin/bash
# Runs on the GPU box: the headline workload with 16 / 32 / 48 / 64 paths in flight per pixel.
mkdir -p gpurun_out
for p in 16 32 48 64; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 3 --paths-per-pixel $p > gpurun_out/ppp_$p.json 2> gpurun_out/ppp_$p.err || { echo "ppp $p failed"; tail -3 gpurun_out/ppp_$p.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ppp_$p.json')); n=d['roofline']['launches']; k=d['kernel_ms']; print('$p paths per pixel:', round(d['value'],1), 'Mrays/s, extend', round(k['extend']/n,2), 'ms x', n, 'launches, shade', round(k['shade']/n,2))"
done
