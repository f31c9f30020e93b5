#!/bin/bash
# Register / spill / occupancy summary of every kernel of one HIP source (compiler's view):
#   scripts/kernel_resources.sh qed_splatter_amd/csrc/composite.hip [-DFLAG ...]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-Rpass.*/,"",n); name=n}
       /TotalSGPRs:/ {sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ \[.*/,"",sg)}
       / VGPRs:/ && !/Spill/ {vg=$0; sub(/.*VGPRs: /,"",vg); sub(/ \[.*/,"",vg)}
       /ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)}
       /VGPR Spill/ {sp=$0; sub(/.*Spill: /,"",sp); sub(/ \[.*/,"",sp)}
       /Occupancy/ {oc=$0; sub(/.*: /,"",oc); sub(/ \[.*/,"",oc)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-110s sgpr %3s vgpr %3s spill %3s scratch %5s occ %2s lds %6s\n", substr(name,1,110), sg, vg, sp, sc, oc, l}'
