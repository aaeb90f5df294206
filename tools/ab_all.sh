#!/bin/bash
# Runs on the GPU box: tools/ab_bench.sh over the headline, the second geometry, config 5 and config 2, then the
# per-frame call at 512 x 512, for the given device-library variants.  Usage: tools/ab_all.sh <libA.so> <libB.so> ...
set -e
echo "== headline"; bash tools/ab_bench.sh "$@"
echo "== suzanne"; AB_ARGS="--geometry suzanne" bash tools/ab_bench.sh "$@"
echo "== config 5"; AB_ARGS="--config 5 --steps 1" bash tools/ab_bench.sh "$@"
echo "== config 2"; AB_ARGS="--config 2 --steps 2" bash tools/ab_bench.sh "$@"
echo "== per-frame 512x512"; bash tools/step512_ab.sh "$@"
