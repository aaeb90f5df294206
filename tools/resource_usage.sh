#!/bin/bash
# Print SGPR / VGPR / scratch / occupancy per kernel of one .hip file (extra args go to hipcc).
f=${1:-raytracer_amd/csrc/device/kernels.hip}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Iinclude "$@" -c "$f" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c '
import re, sys
name = None
vals = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        if name: print(name, vals)
        name = re.sub(r"^_ZN4mirt12_GLOBAL__N_1\d+", "", m.group(1))[:28]; vals = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m: vals[m.group(1).split()[0]] = int(m.group(2))
if name: print(name, vals)
'
