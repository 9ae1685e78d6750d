#!/bin/bash
# profiles/tools/ab_env.sh TAG [bench args]: step time of the default bench line, three runs, with whatever environment the caller set
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=$1; shift
for i in 1 2 3; do python3 bench.py --no-pmc --no-cpu --no-f64 --no-parity "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$TAG', round(d['ms_per_step'],3))"; done
