#!/bin/bash
# profiles/tools/mkall.sh NAME "<extra hipcc flags>": an A/B build of libgeosrad.so whose fp32 objects (geosrad.hip, lw_cols.hip, sw_reform.hip, part 4) and the
# extern "C" layer are compiled with extra flags; the fp64 objects are the current build's
set -e
cd /root/repo
N=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=4 -fno-slp-vectorize "$@" -c geosradiation_gridcomp_amd/csrc/geosrad.hip -o build/obj/a4_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=0 "$@" -c geosradiation_gridcomp_amd/csrc/geosrad.hip -o build/obj/a0_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=4 -fno-slp-vectorize "$@" -c geosradiation_gridcomp_amd/csrc/lw_cols.hip -o build/obj/ac4_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=4 -fno-slp-vectorize "$@" -c geosradiation_gridcomp_amd/csrc/sw_reform.hip -o build/obj/aq4_$N.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/a4_$N.o build/obj/a0_$N.o build/obj/ac4_$N.o build/obj/aq4_$N.o build/obj/geosrad_part8.o build/obj/lw_cols_part8.o build/obj/sw_reform_part8.o -o variants/lib_$N.so
echo built $N
