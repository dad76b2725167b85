#!/bin/bash
# usage: tools/kres.sh file.hip [filter]  -- per-kernel register / spill / LDS summary of one translation unit (gfx950)
cd "$(dirname "$0")/../lighthand_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|TotalSGPRs" | sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, l; n=$3; l=""; next}{l=l" | "$0}END{print n,l}' | c++filt | grep -E "${2:-.}"
