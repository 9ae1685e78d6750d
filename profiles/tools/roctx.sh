#!/bin/bash
# profiles/tools/roctx.sh TAG: the reference's timer names on the GPU timeline - roctx ranges (GEOSRAD_ROCTX=1) of a short one-stream bench run,
# rocprofv3 --marker-trace --kernel-trace --stats; the marker summary goes to gpurun_out/roctx_TAG/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
TAG=$1; shift
export GEOSRAD_ROCTX=1
timeout -k 10 200 rocprofv3 --marker-trace --kernel-trace --stats -d gpurun_out/roctx_$TAG -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-overlap --steps 3 --warmup 1 "$@" > gpurun_out/roctx_$TAG.log 2>&1 || { echo roctx run failed; tail -5 gpurun_out/roctx_$TAG.log; exit 1; }
ls gpurun_out/roctx_$TAG/; head -20 gpurun_out/roctx_$TAG/*marker*stats*.csv 2>/dev/null || head -20 gpurun_out/roctx_$TAG/x_domain_stats.csv
