#!/bin/bash
# Runs on the GPU box: the GPU test suite with the full report kept under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 ${1:-400} python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_gpu.log
exit $rc
