#!/bin/bash
# profiles/tools/pmc_traffic.sh TAG bench-args... : FETCH_SIZE and WRITE_SIZE (separate rocprofv3 --pmc passes) of a bench configuration
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --pmc $C -d gpurun_out/pmc_${TAG}_$C -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --steps 2 --warmup 1 "$@" > gpurun_out/pmc_${TAG}_$C.log 2>&1 || echo "pass $C failed"
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/pmc_${TAG}.txt
grep -A2 "geosrad" gpurun_out/pmc_${TAG}.txt | head -60
