#!/bin/bash
# Build a variant of libmirt.so with extra compiler flags: tools/build_variant.sh <name> <flags...>
# -> raytracer_amd/lib/libmirt_<name>.so (select it with MIRT_DEVICE_LIB=libmirt_<name>.so)
set -e
name=$1; shift
mkdir -p build/variant_$name
for f in raytracer_amd/csrc/device/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Iinclude "$@" -c $f -o build/variant_$name/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/libmirt_$name.so build/variant_$name/*.o
echo built raytracer_amd/lib/libmirt_$name.so
