#!/bin/bash
# Runs on the GPU box: tools/ab_bench.sh over the headline, the second geometry and config 2 for the given variants.
set -e
echo "== headline"; bash tools/ab_bench.sh "$@"
echo "== suzanne"; AB_ARGS="--geometry suzanne" bash tools/ab_bench.sh "$@"
echo "== config 2"; AB_ARGS="--config 2 --steps 2" bash tools/ab_bench.sh "$@"
