#!/bin/bash
# profiles/tools/kstat.sh TAG LIB [bench args]: per-instantiation kernel times (rocprofv3 --kernel-trace --stats) of a variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
TAG=$1; LIB=$2; shift; shift
export GEOSRAD_LIB=$LIB
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$TAG -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-overlap --steps 5 --warmup 2 "$@" > gpurun_out/ks_$TAG.log 2>&1 || { echo kstat $TAG failed; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$TAG/**/x_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bands" in r["Name"]: print("$TAG", r["Name"][:70], "calls", r["Calls"], "avg_ms %.3f" % (float(r["AverageNs"])/1e6))
PY
