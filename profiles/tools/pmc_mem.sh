#!/bin/bash
# profiles/tools/pmc_mem.sh TAG bench-args... : memory-pipeline counters of the step's kernels (one rocprofv3 --pmc pass per group,
# program directly behind `--`), per-kernel means into gpurun_out/pmcmem_TAG.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
P1="TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P2="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
P3="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
P5="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P -d gpurun_out/pmcmem_${TAG}_$i -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-f64 --no-overlap --steps 2 --warmup 1 "$@" > gpurun_out/pmcmem_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcmem_${TAG}_$i.log; }
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmcmem_${TAG}_1 gpurun_out/pmcmem_${TAG}_2 gpurun_out/pmcmem_${TAG}_3 gpurun_out/pmcmem_${TAG}_4 gpurun_out/pmcmem_${TAG}_5 > gpurun_out/pmcmem_${TAG}.txt
grep -A26 "k_lw_bands\|k_sw_reform" gpurun_out/pmcmem_${TAG}.txt | head -150
