#!/bin/bash
# profiles/tools/pmc_occ.sh TAG LIB [bench args]: achieved wavefronts per SIMD of the step's kernels = 4 * SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
TAG=$1; LIB=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export GEOSRAD_LIB=$LIB
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_ANY -d gpurun_out/occ_$TAG -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-f64 --no-overlap --steps 2 --warmup 1 "$@" > gpurun_out/occ_$TAG.log 2>&1 || { echo "failed"; tail -3 gpurun_out/occ_$TAG.log; exit 1; }
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/occ_$TAG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    if "k_lw_bands" in k or "k_sw_reform" in k or "k_mcica" in k:
        m={n:sum(v)/len(v) for n,v in c.items()}
        print("$TAG", k.split("geosrad::")[-1], "waves/SIMD %.2f" % (4*m["SQ_WAVE_CYCLES"]/(m["GRBM_GUI_ACTIVE"]/8*1024)), "waiting %.2f" % (m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]), "waves %d" % m["SQ_WAVES"])
PY
