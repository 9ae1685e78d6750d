# profiles/tools/save_final.py: copy the profile set of profiles/tools/final_prof.sh (gpurun_out/final) into profiles/r02_final_* and rebuild profiles/r02_traffic.json
import json, re, shutil, os
S = "gpurun_out/final"
for f in ("bench_lwsw", "bench_lwsw_one_stream", "bench_lwsw_host_api", "bench_lwsw_half_lit", "bench_cfg1_lw_clear", "bench_lwsw_f64",
          "bench_gridcomp", "bench_chou", "bench_mcica"):
    shutil.copy(f"{S}/{f}.json", f"profiles/r02_final_{f}.json")
shutil.copy(f"{S}/stats_one/x_kernel_stats.csv", "profiles/r02_final_lwsw_one_stream_kernel_stats.csv")
shutil.copy(f"{S}/stats_two/x_kernel_stats.csv", "profiles/r02_final_lwsw_two_streams_kernel_stats.csv")
shutil.copy(f"{S}/pmc_traffic.txt", "profiles/r02_final_lwsw_pmc_traffic_counters.txt")
k = {}; cur = None
for ln in open(f"{S}/pmc_traffic.txt"):
    if not ln.startswith(" "): cur = ln.strip()
    else:
        m = re.match(r"\s+(\w+)\s+(\d+)", ln)
        if m: k.setdefault(cur, {})[m.group(1)] = int(m.group(2))
out = {"_comment": "HBM bytes per step of the kernels of the SHIPPED build (both instantiations of a kernel summed), rocprofv3 --pmc FETCH_SIZE / "
       "WRITE_SIZE in separate passes, gfx950 correction FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section); workload: bench.py default "
       "(97 200 columns x 72 layers, 60 % cloudy, aerosols, fp32), one stream; source profiles/r02_final_lwsw_pmc_traffic_counters.txt",
       "lwsw_97200_72_0.6_aer_f32": {}}
for name in ("k_sw_bands", "k_lw_bands", "k_mcica", "k_lw_reduce", "k_sw_reduce"):
    f = sum(v.get("FETCH_SIZE", 0) for n, v in k.items() if name in n) * 1024
    w = sum(v.get("WRITE_SIZE", 0) for n, v in k.items() if name in n) * 1024
    out["lwsw_97200_72_0.6_aer_f32"][name] = {"traffic_bytes": 2 * f + w, "fetch_GB_raw": round(f / 1e9, 3), "write_GB": round(w / 1e9, 3)}
    print(name, "%.2f GB" % ((2 * f + w) / 1e9))
try:
    prev = json.load(open("profiles/r02_traffic.json"))      # keep the other workloads' entries (e.g. the Chou pair)
    for kk, vv in prev.items():
        if kk not in out: out[kk] = vv
except (OSError, ValueError):
    pass
json.dump(out, open("profiles/r02_traffic.json", "w"), indent=1)
for f in ("bench_lwsw", "bench_lwsw_one_stream"):
    d = json.loads(open(f"{S}/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"], d.get("kernels_ms_per_step"))
