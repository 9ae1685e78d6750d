#!/bin/bash
# profiles/tools/final_prof.sh: the round's profile set of the shipped build (run on the GPU box through gpurun; outputs under gpurun_out/final).
# bench lines of every scheme, rocprofv3 kernel stats (one stream / two streams), and PMC passes - each counter group in a run of its own,
# the program directly behind `--`: issue / wait counters, instruction counts, FETCH_SIZE, WRITE_SIZE - for the default workload and the Chou pair.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/final; rm -rf $O; mkdir -p $O
b() { n=$1; shift; python3 bench.py "$@" > $O/$n.json 2> $O/$n.err; echo "$n: $(python3 -c "import json;d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]);print('%.2f ms/step %.3g %s' % (d['ms_per_step'], d['value'], d['unit']))" 2>&1)"; }
b bench_lwsw --steps 20
b bench_lwsw_one_stream --no-pmc --no-overlap --no-cpu --no-parity --no-configs
b bench_lwsw_host_api --no-pmc --no-cpu --no-parity --host-api --steps 5
GEOSRAD_LW_PATH=split python3 bench.py --no-pmc --no-cpu --no-parity --no-f64 --no-configs --no-overlap > $O/bench_lwsw_one_stream_lw_split.json 2> $O/bench_lwsw_one_stream_lw_split.err || echo split failed
b bench_lwsw_half_lit --no-pmc --lit 0.5
b bench_cfg1_lw_clear --no-pmc --scheme lw --cloudy 0 --no-aerosol --ncol 100000 --no-cpu --no-f64
b bench_lwsw_f64 --no-pmc --real 8 --no-cpu --steps 5
b bench_gridcomp --no-pmc --scheme gridcomp --no-cpu
b bench_chou --scheme chou --no-cpu --steps 3
b bench_mcica --no-pmc --scheme mcica --no-cpu
python3 bench.py --ranks-per-gpu 1,2,4,6 --steps 5 > $O/bench_ranks_per_gpu.json 2> $O/bench_ranks_per_gpu.err || echo ranks failed
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_one -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-configs --no-overlap --steps 5 --warmup 2 > $O/stats_one.log 2>&1 || echo stats_one failed
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_two -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-configs --steps 5 --warmup 2 > $O/stats_two.log 2>&1 || echo stats_two failed
# the legs of the default line's `configs` object, one rocprofv3 run per leg (kernels of one name run in several legs)
for L in cfg0_irrad_1000_clear cfg1_lw_clear_100k cfg2_sw_noaer_100k cfg2_sorad_100k cfg2_irrad_100k cfg2_mcica_200 cfg4_c720_share_137l_rrtmg_standin; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_$L -o x --output-format csv -- python3 bench.py --configs-only --configs-legs $L --steps 5 --warmup 3 > $O/stats_$L.log 2>&1 || echo stats_$L failed
done
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE"
pmc() { tag=$1; shift; i=0
  for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
    timeout -k 10 250 rocprofv3 --pmc $P -d $O/pmc_${tag}_$i -o x --output-format csv -- python3 bench.py --no-pmc --no-cpu --no-parity --no-configs --no-overlap --steps 2 --warmup 1 "$@" > $O/pmc_${tag}_$i.log 2>&1 || echo "pmc $tag pass $i failed"
  done
  python3 profiles/tools/pmc_sum.py $O/pmc_${tag}_1 $O/pmc_${tag}_2 $O/pmc_${tag}_3 $O/pmc_${tag}_4 > $O/pmc_${tag}_counters.txt; }
python3 profiles/tools/pcie.py > $O/pcie.txt 2>&1
pmc lwsw
pmc chou --scheme chou
echo done
