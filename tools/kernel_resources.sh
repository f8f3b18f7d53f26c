#!/bin/bash
# Register / LDS / scratch use of every kernel of the kernel translation units as the compiler reports it (no GPU needed):
#   tools/kernel_resources.sh [unit ...]      (default: hb_step hb_step_duo hb_narrow hb_env; EXTRA=... adds hipcc flags)
# (occupancy: waves per SIMD = min(8, floor(512 / ceil8(VGPRs + AGPRs))); LDS is dynamic for the step kernels: see DESIGN.md)
cd "$(dirname "$0")/.."
UNITS=${@:-hb_step hb_step_duo hb_narrow hb_env}
for u in $UNITS; do
  [ -f humanoid_mujoco_amd/csrc/$u.hip ] || continue
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=on -fno-slp-vectorize -fno-vectorize $EXTRA -Rpass-analysis=kernel-resource-usage -c -x hip humanoid_mujoco_amd/csrc/$u.hip -o /tmp/${u}_res.o 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: [^ ]* *//' | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '/Function Name/ {printf "\n%s", $0; next} {printf "  %s", $0}'
done
echo
