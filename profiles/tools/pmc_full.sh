#!/bin/bash
# profiles/tools/pmc_full.sh TAG LIBPATH bench-args... : rocprofv3 --pmc passes (issue / wait counters, cache counters, FETCH_SIZE, WRITE_SIZE; one
# pass each, program directly behind `--`) of a short one-stream bench run; per-kernel counter means into gpurun_out/pmc_TAG.txt
TAG=$1; LIB=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH GRBM_GUI_ACTIVE"
P3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export GEOSRAD_LIB=$LIB
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P -d gpurun_out/pmc_${TAG}_$i -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-overlap --steps 2 --warmup 1 "$@" > gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${TAG}_$i.log; }
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 gpurun_out/pmc_${TAG}_4 gpurun_out/pmc_${TAG}_5 > gpurun_out/pmc_${TAG}.txt
grep -A26 "k_lw_bands\|k_sw_bands\|k_sw_reform" gpurun_out/pmc_${TAG}.txt | head -150
