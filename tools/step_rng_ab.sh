#!/bin/bash
# Runs on the GPU box: mirt_step with per-sample and per-slot random streams at the two sizes the
# per-slot sample stage is held to (512^2 with one slot per pixel, 1080p with eight).
mkdir -p gpurun_out/step
for spec in "512 512 1" "1920 1080 8"; do
  for rng in per-sample per-slot; do
    set -- $spec
    f=gpurun_out/step/rng_$1_$3_$rng
    timeout -k 10 200 python bench.py --mode step --rng $rng --steps 3 --warmup 1 --width $1 --height $2 --step-paths-per-pixel $3 > $f.json 2> $f.err || { echo "step $spec $rng failed"; tail -3 $f.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$f.json')); print('$spec', '$rng', round(d['value'],1), 'Mrays/s', 'wall/frame', round(d['wall_ms_per_frame'],4), 'kernel/frame', round(d['kernel_ms_per_frame'],4), {k: round(v,4) for k,v in d['kernel_ms_per_frame_by_kernel'].items()})"
  done
done
