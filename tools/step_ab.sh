#!/bin/bash
# Runs on the GPU box: mirt_step at the reference's operating point with several device-library variants.
mkdir -p gpurun_out/step_ab
for lib in "$@"; do
  for spec in "512 512 1" "512 512 4" "1920 1080 1"; do
    set -- $spec
    MIRT_DEVICE_LIB=$lib timeout -k 10 200 python bench.py --mode step --steps 2 --warmup 1 --width $1 --height $2 --step-paths-per-pixel $3 --no-cpu-baseline > gpurun_out/step_ab/${lib%.so}_$1_$3.json 2> gpurun_out/step_ab/${lib%.so}_$1_$3.err || { echo "$lib $spec failed"; tail -3 gpurun_out/step_ab/${lib%.so}_$1_$3.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/step_ab/${lib%.so}_$1_$3.json')); print('$lib', '$spec', round(d['value'],1), 'Mrays/s', round(d['frames_per_s'],1), 'fps', {k: round(v,3) for k,v in d['kernel_ms_per_frame_by_kernel'].items()})"
  done
done
