#!/bin/bash
# Runs on the GPU box: bench.py once per device-library variant.  Usage: tools/variant_sweep.sh <name>...
mkdir -p gpurun_out/variants
for v in "$@"; do
  lib=libmirt.so; [ "$v" != "default" ] && lib=libmirt_$v.so
  MIRT_DEVICE_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/variants/$v.json 2> gpurun_out/variants/$v.err || { echo "$v failed"; tail -3 gpurun_out/variants/$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/variants/$v.json')); print('$v', round(d['value'],1), 'Mrays/s  extend ms/launch', round(d['roofline']['launch_ms'],3))"
done
