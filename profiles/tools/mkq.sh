#!/bin/bash
# profiles/tools/mkq.sh NAME "<extra hipcc flags>" : an A/B build of libgeosrad.so whose fp32 quad-mapped SW object (sw_reform.hip part 4) is
# compiled with extra flags; every other object is the current build's
set -e
cd /root/repo
N=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGEOSRAD_PART=4 -fno-slp-vectorize "$@" -c geosradiation_gridcomp_amd/csrc/sw_reform.hip -o build/obj/q4_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/q4_$N.o build/obj/geosrad_part4.o build/obj/geosrad_part8.o build/obj/geosrad_part0.o build/obj/lw_cols_part4.o build/obj/lw_cols_part8.o build/obj/lw_split_part4.o build/obj/lw_split_part8.o build/obj/sw_reform_part8.o -o variants/lib_$N.so
echo built $N
