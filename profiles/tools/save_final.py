# profiles/tools/save_final.py ROUND: copy the profile set of profiles/tools/final_prof.sh (gpurun_out/final) into profiles/rNN_final_* and build
# profiles/rNN_counters.json - per dominant kernel of a workload: HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
# MI355X_MICROARCH.md), vector-instruction counts and the issue / wait fractions - which bench.py quotes as roofline.traffic / roofline_compute
import json, re, shutil, sys
RN = sys.argv[1] if len(sys.argv) > 1 else "r04"
S = "gpurun_out/final"
for f in ("bench_lwsw", "bench_lwsw_one_stream", "bench_lwsw_host_api", "bench_lwsw_half_lit", "bench_cfg1_lw_clear", "bench_lwsw_f64",
          "bench_gridcomp", "bench_chou", "bench_mcica", "bench_ranks_per_gpu", "bench_lwsw_one_stream_lw_split"):
    try: shutil.copy(f"{S}/{f}.json", f"profiles/{RN}_final_{f}.json")
    except OSError as e: print("missing", f, e)
shutil.copy(f"{S}/stats_one/x_kernel_stats.csv", f"profiles/{RN}_final_lwsw_one_stream_kernel_stats.csv")
shutil.copy(f"{S}/stats_two/x_kernel_stats.csv", f"profiles/{RN}_final_lwsw_two_streams_kernel_stats.csv")
import csv, os
rows = []            # per leg: the kernels above 1 % of the leg's GPU time (the whole files stay on the GPU box's scratch)
for leg in ("cfg0_irrad_1000_clear", "cfg1_lw_clear_100k", "cfg2_sw_noaer_100k", "cfg2_sorad_100k", "cfg2_irrad_100k", "cfg2_mcica_200",
            "cfg4_c720_share_137l_rrtmg_standin"):
    try:
        for r in csv.DictReader(open(f"{S}/stats_{leg}/x_kernel_stats.csv")):
            if float(r["Percentage"]) >= 1.0 and "at::native" not in r["Name"]:
                rows.append({"leg": leg, **r})
    except OSError as e:
        print("missing", leg, e)
if rows:
    with open(f"profiles/{RN}_configs_kernel_stats.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
try:
    shutil.copy(f"{S}/pcie.txt", f"profiles/{RN}_pcie.txt")
except OSError as e:
    print("missing", e)
NSIMD = 1024                       # 256 CUs x 4


def counters(path):
    k = {}; cur = None
    for ln in open(path):
        if not ln.startswith(" "): cur = ln.strip()
        else:
            m = re.match(r"\s+(\w+)\s+(\d+)", ln)
            if m: k.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return k


out = {"_comment": "per step, the instantiations of a kernel summed; rocprofv3 --pmc, one counter group per run (profiles/tools/final_prof.sh), one "
       "stream; traffic_bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes; gfx950 correction of MI355X_MICROARCH.md, HBM section); valu_insts = "
       "SQ_INSTS_VALU (wave instructions); lane_ops_per_column = valu_insts x 64 / columns; valu_util = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE "
       "/ 8 x 1024 SIMDs): share of the kernel's SIMD-cycles in which a vector instruction is in the pipe (the counter charges at least one "
       "quad-cycle per instruction, so it is an upper bound of the issue-slot share); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
for tag, key, ncol, names in (("lwsw", "lwsw_97200_72_0.6_aer_f32", 97200, ("k_sw_reform", "k_sw_bands", "k_lw_bands", "k_mcica", "k_lw_reduce", "k_swr_reduce")),
                              ("chou", "chou_100000_72_0.6_aer_f32", 100000, ("k_chou_bands", "k_sorad_pass"))):
    try:
        k = counters(f"{S}/pmc_{tag}_counters.txt")
    except OSError as e:
        print("missing counters", tag, e); continue
    shutil.copy(f"{S}/pmc_{tag}_counters.txt", f"profiles/{RN}_final_{tag}_pmc_counters.txt")
    out[key] = {}
    for name in names:
        sel = [v for n, v in k.items() if name in n]
        if not sel: continue
        g = lambda c: sum(v.get(c, 0) for v in sel)
        f, w = g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
        simd_cycles = g("GRBM_GUI_ACTIVE") / 8.0 * NSIMD
        e = {"traffic_bytes": 2 * f + w, "fetch_GB_raw": round(f / 1e9, 3), "write_GB": round(w / 1e9, 3), "valu_insts": g("SQ_INSTS_VALU"),
             "salu_insts": g("SQ_INSTS_SALU"), "vmem_rd_insts": g("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": g("SQ_INSTS_VMEM_WR"),
             "lane_ops_per_column": round(g("SQ_INSTS_VALU") * 64.0 / ncol), "valu_util": round(g("SQ_ACTIVE_INST_VALU") * 4.0 / simd_cycles, 3) if simd_cycles else None,
             "wait_frac": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAVE_CYCLES") else None,
             "issue_stall_frac": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAVE_CYCLES") else None}
        out[key][name] = e
        print(key, name, "%.2f GB" % ((2 * f + w) / 1e9), "valu_util", e["valu_util"], "wait", e["wait_frac"], "lane ops/col", e["lane_ops_per_column"])
json.dump(out, open(f"profiles/{RN}_counters.json", "w"), indent=1)
for f in ("bench_lwsw", "bench_lwsw_one_stream"):
    d = json.loads(open(f"{S}/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"], d.get("kernels_ms_per_step"))
