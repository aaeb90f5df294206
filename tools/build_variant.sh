#!/bin/bash
# Builds the device library with extra -D definitions into raytracer_amd/lib/libmirt_<name>.so
# (selected at run time with MIRT_DEVICE_LIB=libmirt_<name>.so).  Usage: tools/build_variant.sh <name> -DA=1 -DB=2 ...
set -eu
name=$1; shift
out=build/variants/$name
mkdir -p "$out"
objs=""
for src in raytracer_amd/csrc/device/*.hip; do
  o=$out/$(basename "${src%.hip}").o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-rdc -Wall -Iinclude "$@" -c "$src" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/libmirt_$name.so $objs
echo "built raytracer_amd/lib/libmirt_$name.so"
