#!/bin/bash
# Print VGPR / SGPR / spill / occupancy per kernel of one .hip file.
f=${1:-raytracer_amd/csrc/device/kernels.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Iinclude -c "$f" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/ {printf "\n%s", $0; next} {printf " | %s", $0} END {print ""}' | sed -E 's/Function Name: _ZN4mirt12_GLOBAL__N_1[0-9]+//; s/ENS_6D.*E \|/ |/'
