#!/bin/bash
# profiles/tools/swq_regs.sh [extra flags]: VGPR need of every distinct quad body of k_sw_quads (one compile per body, launch bounds relaxed)
cd /root/repo
for s in 16 17 18 19 20 21 22 23 24 25 26 27 28 29; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGEOSRAD_PART=4 -DSWQ_OCC_CLR=1 -DSWQ_OCC_CLD=1 -DSWQ_ONLY_BAND=$s "$@" -c geosradiation_gridcomp_amd/csrc/sw_quads.hip -o /tmp/swq_only_$s.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "k_sw_quads" | grep -E " VGPRs:" | awk -v s=$s '{printf "band %s vgpr %s\n", s, $4}' | paste -sd' ' ) &
done
wait
