#!/bin/bash
# profiles/tools/mkv.sh NAME "<extra hipcc flags>" : an A/B build of libgeosrad.so whose fp32 object of geosrad.hip (part 4: every fp32 kernel but
# k_sw_reform and the lw_cols ones) is compiled with extra flags; every other object is the current build's
set -e
cd /root/repo
N=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGEOSRAD_PART=4 -fno-slp-vectorize "$@" -c geosradiation_gridcomp_amd/csrc/geosrad.hip -o build/obj/g4_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/g4_$N.o build/obj/sw_reform_part4.o build/obj/geosrad_part8.o build/obj/geosrad_part0.o build/obj/lw_cols_part4.o build/obj/lw_cols_part8.o build/obj/lw_split_part4.o build/obj/lw_split_part8.o build/obj/sw_reform_part8.o -o variants/lib_$N.so
echo built $N
