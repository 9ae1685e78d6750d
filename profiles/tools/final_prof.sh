#!/bin/bash
# profiles/tools/final_prof.sh: the round's profile set of the shipped build (run on the GPU box through gpurun; outputs under gpurun_out/final)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/final; mkdir -p $O
b() { n=$1; shift; python3 bench.py "$@" > $O/$n.json 2> $O/$n.err; echo "$n: $(python3 -c "import json;d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]);print('%.2f ms/step %.3g %s' % (d['ms_per_step'], d['value'], d['unit']))" 2>&1)"; }
b bench_lwsw --steps 20
b bench_lwsw_one_stream --no-overlap --no-cpu
b bench_lwsw_host_api --no-cpu --host-api --steps 5
b bench_lwsw_half_lit --lit 0.5
b bench_cfg1_lw_clear --scheme lw --cloudy 0 --no-aerosol --ncol 100000 --no-cpu
b bench_lwsw_f64 --real 8 --no-cpu --steps 5
b bench_gridcomp --scheme gridcomp --no-cpu
b bench_chou --scheme chou --no-cpu --steps 3
b bench_mcica --scheme mcica --no-cpu
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_one -o x --output-format csv -- python3 bench.py --no-cpu --no-overlap --steps 5 --warmup 2 > $O/stats_one.log 2>&1 || echo stats_one failed
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_two -o x --output-format csv -- python3 bench.py --no-cpu --steps 5 --warmup 2 > $O/stats_two.log 2>&1 || echo stats_two failed
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C -d $O/pmc_$C -o x --output-format csv -- python3 bench.py --no-cpu --no-overlap --steps 2 --warmup 1 > $O/pmc_$C.log 2>&1 || echo pmc $C failed
done
python3 profiles/tools/pmc_sum.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.txt
echo done
