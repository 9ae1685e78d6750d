#!/bin/bash
# profiles/tools/mkall.sh NAME "<extra hipcc flags>" : an A/B build of libgeosrad.so with the flags on geosrad.hip (fp32 and fp64 kernels) and
# sw_reform.hip (fp32 and fp64); the other objects are the current build's
set -e
cd /root/repo
N=$1; shift
mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize"
S=geosradiation_gridcomp_amd/csrc
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=4 "$@" -c $S/geosrad.hip -o build/obj/a_g4_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=8 "$@" -c $S/geosrad.hip -o build/obj/a_g8_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=4 "$@" -c $S/sw_reform.hip -o build/obj/a_q4_$N.o &
/opt/rocm/bin/hipcc $F -DGEOSRAD_PART=8 "$@" -c $S/sw_reform.hip -o build/obj/a_q8_$N.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/a_g4_$N.o build/obj/a_g8_$N.o build/obj/a_q4_$N.o build/obj/a_q8_$N.o build/obj/geosrad_part0.o build/obj/lw_cols_part4.o build/obj/lw_cols_part8.o build/obj/lw_split_part4.o build/obj/lw_split_part8.o -o variants/lib_$N.so
rm -f build/obj/a_*_$N.o
echo built $N
