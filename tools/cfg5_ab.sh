#!/bin/bash
# Runs on the GPU box: config 5 with the previous library, the new one, and the new one with one kernel per material.
mkdir -p gpurun_out
run() {  # tag, lib, extra args
  MIRT_DEVICE_LIB=$2 timeout -k 10 300 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary $3 > gpurun_out/c5_$1.json 2> gpurun_out/c5_$1.err || { echo "$1 failed"; tail -5 gpurun_out/c5_$1.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/c5_$1.json')); k=d['kernel_ms']; print('$1', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],1), 'ms/step', {a: round(b,1) for a,b in k.items()})"
}
run old libmirt_head0.so "" && run new libmirt.so "" && run new_stage libmirt.so "--stage-kernels" && run old2 libmirt_head0.so "" && run new2 libmirt.so ""
