#!/bin/bash
# Runs on the GPU box: the per-frame call at 512 x 512, one path per pixel, with several device-library variants, twice.
mkdir -p gpurun_out/step_ab
for round in 1 2; do
  for lib in "$@"; do
    MIRT_DEVICE_LIB=$lib timeout -k 10 200 python bench.py --mode step --steps 2 --warmup 1 --width 512 --height 512 --step-paths-per-pixel 1 --no-cpu-baseline > gpurun_out/step_ab/${lib%.so}_512.json 2> gpurun_out/step_ab/${lib%.so}_512.err || { echo "$lib failed"; tail -3 gpurun_out/step_ab/${lib%.so}_512.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/step_ab/${lib%.so}_512.json')); print('$lib', round(d['value'],1), 'Mrays/s', round(d['frames_per_s'],1), 'fps', {k: round(v,3) for k,v in d['kernel_ms_per_frame_by_kernel'].items()})"
  done
done
