#!/bin/bash
# Runs on the GPU box: mirt_step at the reference's operating point and around it.
mkdir -p gpurun_out/step
for spec in "512 512 1" "512 512 4" "512 512 16" "512 512 64" "1920 1080 1" "1920 1080 8"; do
  set -- $spec
  timeout -k 10 200 python bench.py --mode step --steps 2 --warmup 1 --width $1 --height $2 --step-paths-per-pixel $3 > gpurun_out/step/s_$1_$3.json 2> gpurun_out/step/s_$1_$3.err || { echo "step $spec failed"; tail -3 gpurun_out/step/s_$1_$3.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/step/s_$1_$3.json')); print('$spec', round(d['value'],1), 'Mrays/s', round(d['frames_per_s'],1), 'fps', 'wall/frame', round(d['wall_ms_per_frame'],3), 'kernel/frame', round(d['kernel_ms_per_frame'],3), 'sync/frame', round(d['wall_ms_per_frame_synchronised_every_frame'],3), {k: round(v,3) for k,v in d['kernel_ms_per_frame_by_kernel'].items()})"
done
