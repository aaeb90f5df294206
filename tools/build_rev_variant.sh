#!/bin/bash
# Builds the device library as it was at a git revision into raytracer_amd/lib/libmirt_<name>.so, for same-box A/B
# runs against the working tree (MIRT_DEVICE_LIB=libmirt_<name>.so).  Usage: tools/build_rev_variant.sh <name> <rev> [-D...]
set -eu
name=$1; rev=$2; shift 2
src=build/rev_src/$name
rm -rf "$src"; mkdir -p "$src" build/variants/$name
git archive "$rev" raytracer_amd/csrc/device include | tar -x -C "$src"
objs=""
for f in "$src"/raytracer_amd/csrc/device/*.hip; do
  o=build/variants/$name/$(basename "${f%.hip}").o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-rdc -Wall -I"$src"/include "$@" -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raytracer_amd/lib/libmirt_$name.so $objs
echo "built raytracer_amd/lib/libmirt_$name.so from $rev"
