#!/bin/bash
# profiles/tools/kq.sh TAG LIB [bench args]: per-instantiation times of the SW band kernels (rocprofv3 --kernel-trace --stats) of a variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
TAG=$1; LIB=$2; shift; shift
[ "$LIB" != "-" ] && export GEOSRAD_LIB=$LIB
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$TAG -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-overlap --steps 5 --warmup 2 "$@" > gpurun_out/ks_$TAG.log 2>&1 || { echo kstat $TAG failed; tail -5 gpurun_out/ks_$TAG.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$TAG/**/x_kernel_stats.csv", recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_sw_bands" in n or "k_sw_reform" in n or "k_sw_reduce" in n or "k_lw_bands" in n or "k_swr" in n or "k_mcica" in n:
        short=n.split("geosrad::")[1].split("(")[0]
        out.append("%s=%.3f" % (short, float(r["AverageNs"])/1e6))
print("$TAG", " ".join(sorted(out)))
PY
