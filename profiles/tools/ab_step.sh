#!/bin/bash
# profiles/tools/ab_step.sh TAG LIB: the default bench line's step time (two streams) of a variant library, three runs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=$1; LIB=$2
[ "$LIB" != "-" ] && export GEOSRAD_LIB=$LIB
for i in 1 2 3; do python3 bench.py --no-pmc --no-cpu --no-f64 --no-parity 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$TAG', round(d['ms_per_step'],3))"; done
