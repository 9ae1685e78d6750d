#!/bin/bash
# profiles/tools/swr_regs.sh [extra flags]: VGPR need of every band body of k_sw_reform (one compile per band, launch bounds relaxed)
cd /root/repo
for s in 16 17 18 19 20 21 22 23 24 25 26 27 28 29; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DGEOSRAD_PART=4 -DSWR_OCC_CLR=1 -DSWR_OCC_CLD=1 -DSWR_ONLY_BAND=$s "$@" -c geosradiation_gridcomp_amd/csrc/sw_reform.hip -o /tmp/swr_only_$s.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "k_sw_reform" | grep -E " VGPRs:| VGPRs Spill:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste -sd' ' | sed "s/^/band $s: /" ) &
done
wait
