#!/bin/bash
# VGPRs / spills / scratch / occupancy of the kernels whose mangled name matches $1 (default: all), from the compiler's
# own resource-usage remarks (no GPU needed).   usage: scripts/kernel_resources.sh [regex]
cd "$(dirname "$0")/.." || exit 1
pat="${1:-.}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Rpass-analysis=kernel-resource-usage \
    -o /tmp/kernel_resources.o tdoa-geolocation_amd/csrc/tdoa_mi355x.hip 2>&1 |
  awk -v pat="$pat" '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name); show = (name ~ pat) }
    show && /VGPRs:/ && !/Spill/ { v=$0; sub(/.*VGPRs: /,"",v); sub(/ .*/,"",v) }
    show && /ScratchSize/ { s=$0; sub(/.*: /,"",s); sub(/ .*/,"",s) }
    show && /VGPRs Spill/ { sp=$0; sub(/.*Spill: /,"",sp); sub(/ .*/,"",sp) }
    show && /Occupancy/ { o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o) }
    show && /LDS Size/ { printf "%-110s vgpr %3s spill %3s scratch %4s occ %s\n", substr(name,1,110), v, sp, s, o }'
