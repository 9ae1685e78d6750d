#!/bin/bash
# profiles/tools/mk4.sh NAME "<extra hipcc flags>" : an A/B build of libgeosrad.so whose fp32 kernel object (geosrad.hip part 4) is compiled with extra flags
set -e
cd /root/repo
N=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGEOSRAD_PART=4 "$@" -c geosradiation_gridcomp_amd/csrc/geosrad.hip -o build/obj/v4_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/obj/v4_$N.o build/obj/geosrad_part8.o build/obj/geosrad_part0.o build/obj/lw_cols_part4.o build/obj/lw_cols_part8.o build/obj/lw_split_part4.o build/obj/lw_split_part8.o -o build/variants/lib_$N.so
echo built $N
