#!/bin/bash
# Register / LDS / scratch use of every kernel in hb_kernels.hip as the compiler reports it (no GPU needed):
#   tools/kernel_resources.sh [extra hipcc flags]
# (occupancy: waves per SIMD = min(8, floor(512 / ceil8(VGPRs + AGPRs))); LDS is dynamic for the step kernels: see DESIGN.md)
cd "$(dirname "$0")/.."
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -fno-slp-vectorize -fno-vectorize "$@" -Rpass-analysis=kernel-resource-usage -c -x hip humanoid_mujoco_amd/csrc/hb_kernels.hip -o /tmp/hb_kernels_res.o 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: [^ ]* *//' | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '/Function Name/ {printf "\n%s", $0; next} {printf "  %s", $0}'
echo
