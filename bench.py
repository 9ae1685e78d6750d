#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic columns resident in HBM.  BASELINE.json's
metric is "columns/sec (LW+SW, 72 layers)"; it is quoted on configs[3] (C360 tile = 777 600 columns over 8
GPUs, full LW+SW with McICA clouds + aerosols).  The default workload is that configuration's per-GPU share:
97 200 columns/GPU, 72 layers, RRTMG_LW (140 g-points) + RRTMG_SW (112 g-points, every column lit), 60 % of
the columns cloudy (McICA, ih = 1), aerosols on -- so the driver's N = 8 weak-scaling run IS configs[3].
`--scheme lw --cloudy 0 --no-aerosol --ncol 100000` reproduces configs[1] (RRTMG_LW clear-sky); `--scheme chou` runs the
Chou-Suarez pair irrad + sorad (configs[0] / configs[2] schemes), `--scheme irrad` / `--scheme sorad` one of them.
Columns shard embarrassingly: every rank owns its own batch (weak scaling), there is no data-path
collective; the only collectives are the timing barrier and the max-over-ranks of the time.

One JSON line is printed by rank 0 (contract in the task statement), with
  roofline     : the kernel with the largest share of the step, timed live with HIP events on the launch stream
  cpu_baseline : RRTMG_LW = the reference's own Fortran (oracle/_ref); RRTMG_SW = the plain-C oracle (the
                 reference's rrtmg_sw driver needs ESMF/MAPL and cannot be built here) -- timed on the host
                 cores on a bounded sample of the same workload.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# algorithmic (compulsory) bytes per column at the solver API, fp32, SURVEY.md 8(d):
#   RRTMG_LW clear-sky, no aerosol array: every input read once + every output written once
LW_IN2D = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr",
           "cldf", "ciwp", "clwp", "rei", "rel"]


def algorithmic_bytes_lw(nlay, real_bytes, aerosol):
    n_in = 18 * nlay + 2 * (nlay + 1) + 2 + 16 + (16 * nlay if aerosol else 0)      # 2 612 with tauaer @72
    n_out = 6 * (nlay + 1) + 4
    return (n_in + n_out) * real_bytes


def algorithmic_bytes_sw(nlay, real_bytes, aerosol):
    """17 452 B/column with the 3 aerosol arrays, 5 356 without, fp32 @72 layers"""
    n_in = 13 * nlay + (nlay + 1) + 6 + (3 * 14 * nlay if aerosol else 0)
    n_out = 4 * (nlay + 1) + 6 + 14 + 8 + 4
    return (n_in + n_out) * real_bytes


SW_IN = ["coszen", "asdir", "asdif", "aldir", "aldif"]
SW_AER = ["tauaer_sw", "ssaaer_sw", "asmaer_sw"]
SW_OUT1 = ["nirr", "nirf", "parr", "parf", "uvrr", "uvrf", "cotdtp", "cotdhp", "cotdmp", "cotdlp", "cotntp", "cotnhp", "cotnmp", "cotnlp"]


def shard_start(rank, ncol_per_gpu):
    """first global column of a rank's batch: contiguous blocks, rank-major (no overlap, no exchange)"""
    return rank * ncol_per_gpu


def launch_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks ourselves, as CHILD processes of a
    parent that has not touched the GPU (one rank per device), and hand their exit code on.  The ranks only ever exchange a barrier
    and the MAX of one double, so the rendezvous is the one torch.distributed.run sets up on 127.0.0.1."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd)


def control_path_only(a, rank, world):
    """everything of an N-rank run except the GPU: rank -> shard, rendezvous, barrier, MAX over ranks, one JSON line from rank 0.
    (tests/test_dist.py drives `python bench.py --gpus 2 --control-path-only` through the launcher on the CPU.)"""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                      # the "steps": the slowest rank sets the time
    own = time.perf_counter() - t0                     # this rank's own time, before it waits for the others
    if world > 1:
        dist.barrier()
    mine = time.perf_counter() - t0
    per_rank_ms = all_over_ranks(own * 1e3, world)
    elapsed = max_over_ranks(mine, world, torch.device("cpu"))
    starts = [shard_start(r, a.ncol) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "control path only (no GPU work)", "value": None, "unit": "columns/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3, "scaling": "weak", "per_rank_ms": per_rank_ms,
                          "config": {"columns_per_gpu": a.ncol, "shard_starts": starts}}))
    if world > 1:
        dist.destroy_process_group()


def max_over_ranks(seconds, world, device):
    """the only cross-rank reduction of the benchmark: MAX of the elapsed time"""
    if world <= 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_over_ranks(value, world):
    """every rank's figure on every rank (a straggler GPU shows in the N > 1 line's per_rank_ms)"""
    if world <= 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64)
    got = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(got, t)
    return [float(g.item()) for g in got]


def _cpu_worker(args):
    lw_kind, scheme, start, ncol, nlay, cloudy, aerosol = args
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    inp = synth.make_columns(ncol, nlay, start=start, cloudy_frac=cloudy, aerosol=aerosol)
    ih = 1 if cloudy > 0 else 0
    t_lw = t_sw = 0.0
    if "lw" in scheme:
        if lw_kind == "reference":
            from oracle import reflib
            reflib.lib("r4"); reflib.set_inhomogeneity(ih, "r4")
            t = time.perf_counter()
            reflib.rrtmg_lw(inp, "r4", psize=4)        # GEOS default RRTMGLW_PARTITION_SIZE=4 (IRR:3185)
            t_lw = time.perf_counter() - t
        else:
            clib.lib(); clib.set_inhomogeneity(ih, "f32")
            t = time.perf_counter()
            clib.rrtmg_lw(inp, "f32")
            t_lw = time.perf_counter() - t
    t_ref_stages = t_port_stages = 0.0
    if "sw" in scheme:
        clib.lib(); clib.set_inhomogeneity(ih, "f32")
        t = time.perf_counter()
        clib.rrtmg_sw(inp, prec="f32", iaer=10 if aerosol else 0, normFlx=1)
        t_sw = time.perf_counter() - t
        # bracket the port by reference code: the stages of rrtmg_sw that DO build from the reference's sources (setcoef_sw + taumol_sw,
        # the McICA generator, cldprmc_sw) timed as the reference's own Fortran and as the port, on a quarter of the sample
        from oracle import reflib
        if reflib.available("r4"):
            nq = max(64, ncol // 4)
            sub = {k: (np.ascontiguousarray(v[..., :nq]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v) for k, v in inp.items()}
            reflib.lib("r4"); reflib.set_inhomogeneity(ih, "r4")
            svar = [np.float32(1.0)] * 3; sb = np.ones((3, 29), dtype=np.float32)
            for which in ("reference", "port"):
                t = time.perf_counter()
                if which == "reference":
                    reflib.sw_setcoef_taumol(sub, isolvar=0, svar=svar, svar_bnd=sb, kind="r4")
                    cl, ci, cw = reflib.mcica(sub["zm"], sub["alat"], int(inp["dyofyr"]), sub["play"], sub["cldf"], sub["ciwp"], sub["clwp"], 112,
                                              seed_order=(4, 3, 2, 1), kind="r4")
                    reflib.sw_cldprmc(cl, ci, cw, sub["rei"], sub["rel"], iceflag=3, kind="r4")
                    t_ref_stages = (time.perf_counter() - t) * (ncol / nq)
                else:
                    clib.sw_setcoef_taumol(sub, isolvar=0, svar=svar, svar_bnd=sb, prec="f32")
                    cl, ci, cw = clib.mcica(sub["zm"], sub["alat"], int(inp["dyofyr"]), sub["play"], sub["cldf"], sub["ciwp"], sub["clwp"], 112,
                                            seed_order=(4, 3, 2, 1), prec="f32")
                    clib.sw_cldprmc(cl, ci, cw, sub["rei"], sub["rel"], iceflag=3, prec="f32")
                    t_port_stages = (time.perf_counter() - t) * (ncol / nq)
    if scheme in ("chou", "irrad"):
        ch = synth.chou_lw_inputs(inp, aerosol=aerosol)
        clib.lib()
        t = time.perf_counter()
        clib.irrad(ch, "f32")
        t_lw = time.perf_counter() - t
    if scheme in ("chou", "sorad"):
        cs = synth.chou_sw_inputs(inp, aerosol=aerosol)
        clib.lib()
        t = time.perf_counter()
        clib.sorad(cs, "f32")
        t_sw = time.perf_counter() - t
    return t_lw, t_sw, t_ref_stages, t_port_stages


def cpu_baseline(nlay, scheme, cloudy, aerosol, per_core=4096):
    """Time the CPU baseline on all host cores (one process per core: the reference keeps module state)."""
    import multiprocessing as mp
    from oracle import reflib
    lw_kind = "reference" if reflib.available("r4") else "port"
    kind = lw_kind if scheme == "lw" else "port"
    if scheme in ("chou", "irrad", "sorad"):
        per_core = 1024
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    ctx = mp.get_context("fork")
    jobs = [(lw_kind, scheme, 10_000_000 + i * per_core, per_core, nlay, cloudy, aerosol) for i in range(cores)]
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        per = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(p[0] + p[1] for p in per)
    lw_s = sum(p[0] for p in per) / len(per); sw_s = sum(p[1] for p in per) / len(per)
    ref_st = sum(p[2] for p in per) / len(per); port_st = sum(p[3] for p in per) / len(per)
    legs = []
    if "lw" in scheme:
        legs.append(f"rrtmg_lw = {'reference Fortran (oracle/_ref), psize=4' if lw_kind == 'reference' else 'plain-C oracle'} "
                    f"{per_core / lw_s:.0f} col/s/core")
    if "sw" in scheme:
        legs.append(f"rrtmg_sw = plain-C oracle (reference driver needs ESMF/MAPL: unbuildable here) {per_core / sw_s:.0f} col/s/core")
        if ref_st > 0:
            legs.append(f"of it the stages the reference's own Fortran provides (setcoef_sw + taumol_sw + generate_stochastic_clouds + cldprmc_sw): "
                        f"reference {ref_st:.2f} s vs port {port_st:.2f} s per {per_core} columns = {100 * port_st / sw_s:.0f} % of the port's "
                        f"rrtmg_sw time; with those stages at the reference's speed the SW leg would take {sw_s - port_st + ref_st:.2f} s instead of {sw_s:.2f} s")
    if scheme in ("chou", "irrad"):
        legs.append(f"irrad = plain-C oracle (irrad.F90 needs MAPL: unbuildable here) {per_core / lw_s:.0f} col/s/core")
    if scheme in ("chou", "sorad"):
        legs.append(f"sorad = plain-C oracle (sorad.F90 needs MAPL: unbuildable here) {per_core / sw_s:.0f} col/s/core")
    return {"value": cores * per_core / busy, "unit": "columns/s", "cores": cores, "kind": kind,
            "sample": f"{cores} processes x {per_core} columns of the bench workload ({nlay} layers, cloudy fraction {cloudy}, "
                      f"aerosol {aerosol}); " + "; ".join(legs) + f"; slowest process {busy:.2f} s (pool wall {wall:.2f} s incl. input generation)",
            "single_core_columns_per_s": per_core / (lw_s + sw_s),
            "sw_reference_stages": None if ref_st <= 0 else {"reference_s": ref_st, "port_s": port_st, "port_rrtmg_sw_s": sw_s,
                                                              "fraction_of_port_sw": port_st / sw_s, "columns": per_core}}


def _hb_cpu_worker(args):
    """heartbeat leg of the CPU baseline: the plain-C restatement of Update_Flx + UPDATE_EXPORT + heating rates"""
    seed, n, lm = args
    from geosradiation_gridcomp_amd import gridcomp as G
    from oracle import clib
    clib.lib()
    rng = np.random.default_rng(seed)
    f32 = np.float32
    st = {k: rng.uniform(-400, 400, (lm + 1, n)).astype(f32) for k in G.LWU_IN if k not in ("TSINST", "TS_INT", "SFCEM_INT", "FCLD") + tuple(G.LWU_IN_NA)}
    st["TSINST"] = rng.uniform(270, 300, n).astype(f32); st["TS_INT"] = st["TSINST"] - 1; st["SFCEM_INT"] = rng.uniform(300, 450, n).astype(f32)
    st["FCLD"] = np.zeros((lm, n), f32)
    sw = {k: rng.uniform(0, 1, (lm + 1, n)).astype(f32) for k in G.SWU_IN[1:9]}
    sw["SLR"] = rng.uniform(0, 1300, n).astype(f32)
    sw["FSWBANDN"] = rng.uniform(0, 1, (14, n)).astype(f32); sw["FSWBANDNAN"] = sw["FSWBANDN"]
    rt = {k: rng.uniform(-300, 300, (lm + 1, n)).astype(f32) for k in G.RT_IN[1:8]}
    rt["PLE"] = np.cumsum(rng.uniform(100, 3000, (lm + 1, n)), axis=0).astype(f32)
    rt["DSFDTS"] = st["SFCEM_INT"]; rt["SFCEM"] = st["SFCEM_INT"]; rt["TRD"] = st["TSINST"]
    t = time.perf_counter()
    for _ in range(HB_REPS):
        clib.lw_update_flx(st, lm, True, 30, 47, 1e15, "f32", want=HB_LW)
        clib.sw_update_export(sw, lm, 14, "f32", want=HB_SW)
        clib.rad_tendencies(rt, lm, 9.80665, 1004.683, "f32", want=HB_RT)
    return time.perf_counter() - t


HB_REPS = 16
# exports the heartbeat benchmark requests (what a typical GEOS history + the parent's couplings ask for every model step)
HB_LW = ["FLX", "FLC", "FLXU", "FLCU", "FLXD", "FLCD", "OLR", "OLC", "OLCC5", "DSFDTS", "SFCEM", "LWS", "LCS", "LCSC5", "FLNS", "FLNSC",
         "DSFDTS0", "SFCEM0", "TSREFF", "CLDTT"]
HB_SW = ["FSW", "FSC", "FSWNA", "FSCNA", "FSWU", "FSCU", "FSWD", "FSCD", "FSWBAND", "RSR", "RSC", "RSRS", "RSCS", "OSR", "OSRCLR"]
HB_RT = ["DTDT", "RADLW", "RADSW", "RADLWC", "RADSWC", "BLW", "ALW", "RADSRF"]


def heartbeat_cpu_baseline(lm, per=8192):
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    with mp.get_context("fork").Pool(cores) as pool:
        busy = max(pool.map(_hb_cpu_worker, [(i, per, lm) for i in range(cores)]))
    return {"value": cores * per * HB_REPS / busy, "unit": "columns/s", "cores": cores, "kind": "port",
            "sample": f"{cores} processes x {HB_REPS} passes over {per} columns x {lm} layers of the same three updates, plain-C restatement "
                      f"(the GridComps need ESMF/MAPL: unbuildable here); slowest process {busy:.2f} s"}


def bench_gridcomp(a, rank, world, dev, local_rank, aerosol, cpu):
    """--scheme gridcomp : the RRTMG branches of LW_Driver + SORADCORE on GEOS-native fields (prep / flip + solver + post), i.e. the
                           default workload entered one level higher
       --scheme heartbeat: Update_Flx + UPDATE_EXPORT (flux part) + the parent's heating rates: pure streaming,
                           the one genuinely HBM-bound piece next to the solvers; roofline = algorithmic bytes / time"""
    import torch
    import torch.distributed as dist
    from geosradiation_gridcomp_amd import gridcomp as G
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context
    ncol, lm = a.ncol, a.nlay
    tdt = torch.float32 if a.real == 4 else torch.float64
    ctx = Context(a.real, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    to = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev, dtype=tdt)
    zeros = lambda *s: torch.zeros(*s, device=dev, dtype=tdt)
    hb_bytes = 0
    if a.scheme == "gridcomp":
        inp = synth.make_columns(ncol, lm, start=shard_start(rank, ncol), cloudy_frac=a.cloudy, aerosol=aerosol)
        fl, fs = synth.geos_lw_fields(inp), synth.geos_sw_fields(inp)
        if not aerosol:
            for k in ("TAUA", "SSAA"):
                fl[k] = None
            for k in ("TAUA", "SSAA", "ASYA"):
                fs[k] = None
        tl = {k: to(v) for k, v in fl.items() if isinstance(v, np.ndarray)}
        ts = {k: to(v) for k, v in fs.items() if isinstance(v, np.ndarray)}
        aer0 = {k: ts[k].clone() for k in ("TAUA", "SSAA", "ASYA") if k in ts}
        for k in G.LWD_OUT:
            tl[k] = zeros(lm + 1, ncol) if k in G.LWD_OUT_3D else (zeros(ncol, 16) if k in ("OLRB", "DOLRB") else zeros(ncol))
        for k in G.SWD_OUT:
            if k.endswith("NA") and not a.na_pass:
                continue
            ts[k] = (zeros(lm + 1, ncol) if k in ("FSW", "FSC", "FSWU", "FSCU", "FSWNA", "FSCNA", "FSWUNA", "FSCUNA")
                     else (zeros(14, ncol) if k.startswith("FSWBAND") else zeros(ncol)))
        pl = {k: v.data_ptr() for k, v in tl.items()}; ps = {k: v.data_ptr() for k, v in ts.items()}
        cl, cs = G.lwd_consts(), G.swd_consts()
        ctx.set_inhomogeneity(1 if a.cloudy > 0 else 0)
        doy = int(inp["dyofyr"])

        side = None if a.no_overlap else torch.cuda.Stream()        # the two drivers are independent: two HIP streams
        sw_stream = stream if side is None else side.cuda_stream

        rats = G.RAT_GAS[:a.rats]
        if rats:
            for k in G.LWD_RAT_OUT:
                tl[k] = zeros(len(rats), ncol) if k == "SFCEM_RAT" else zeros(len(rats), lm + 1, ncol)
            pl = {k: v.data_ptr() for k, v in tl.items()}

        def step():
            if rats:          # RATS_DIAGNOSTICS: the reference re-runs rrtmg_lw per gas (IRR:3405-3468); here the gases ride on the call
                ctx.lw_driver_rrtmg_rats_dev(stream, ncol, lm, 16 if aerosol else 0, pl, cl, 3, 1, doy, fl["LCLDLM"], fl["LCLDMH"], rats)
            else:
                ctx.lw_driver_rrtmg_dev(stream, ncol, lm, 16 if aerosol else 0, pl, cl, 3, 1, doy, fl["LCLDLM"], fl["LCLDMH"])
            with torch.cuda.stream(side) if (side is not None and sw_stream != stream) else contextlib.nullcontext():
                for k in aer0:                  # the SW driver normalises the aerosol triplet in place: restore the inputs
                    ts[k].copy_(aer0[k])
            ctx.sw_driver_rrtmg_dev(sw_stream, ncol, lm, 14 if aerosol else 0, ps, cs, 3, 1, 1361.0, 1.0, 0, doy, aerosol, fs["LCLDLM"], fs["LCLDMH"], 1)
    else:
        g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
        rnd = lambda lo, hi, *s: (torch.rand(*s, device=dev, generator=g, dtype=torch.float32) * (hi - lo) + lo).to(tdt)
        lw = {k: rnd(-400, 400, lm + 1, ncol) for k in G.LWU_IN if k not in ("TSINST", "TS_INT", "SFCEM_INT", "FCLD") + tuple(G.LWU_IN_NA)}
        lw["TS_INT"] = rnd(270, 300, ncol); lw["TSINST"] = lw["TS_INT"] + rnd(-3, 3, ncol); lw["SFCEM_INT"] = rnd(300, 450, ncol)
        lw["FCLD"] = rnd(0, 1, lm, ncol) * (rnd(0, 1, lm, ncol) < 0.05)
        sw = {k: rnd(0, 1, lm + 1, ncol) for k in G.SWU_IN[1:9]}
        sw["SLR"] = rnd(0, 1300, ncol); sw["FSWBANDN"] = rnd(0, 1, 14, ncol); sw["FSWBANDNAN"] = rnd(0, 1, 14, ncol)
        rt = {"PLE": torch.cumsum(rnd(100, 3000, lm + 1, ncol), 0), "DSFDTS": rnd(4, 6, ncol), "SFCEM": rnd(300, 450, ncol), "TRD": rnd(270, 300, ncol)}
        nin = sum(v.numel() for v in lw.values()) + sum(v.numel() for k, v in sw.items() if k != "FSWBANDNAN" and k not in ("FSWUNAN", "FSCUNAN"))
        for k in HB_LW:
            lw[k] = zeros(lm + 1, ncol) if k in G.LWU_OUT_3D else zeros(ncol)
        for k in HB_SW:
            sw[k] = zeros(lm + 1, ncol) if k in G.SWU_OUT_3D else (zeros(14, ncol) if k in G.SWU_OUT_BAND else zeros(ncol))
        # the parent reads the children's exports
        rt.update(FLW=lw["FLX"], FSW=sw["FSW"], FLWCLR=lw["FLC"], FSWCLR=sw["FSC"])
        for k in HB_RT:
            rt[k] = zeros(lm, ncol) if k in G.RT_OUT_3D else zeros(ncol)
        nin += rt["PLE"].numel() + 4 * (lm + 1) * ncol + 3 * ncol
        nout = sum(lw[k].numel() for k in HB_LW) + sum(sw[k].numel() for k in HB_SW) + sum(rt[k].numel() for k in HB_RT)
        hb_bytes = (nin + nout) * a.real            # every field a requested export needs read once + every export written once
        p1 = {k: v.data_ptr() for k, v in lw.items()}; p2 = {k: v.data_ptr() for k, v in sw.items()}; p3 = {k: v.data_ptr() for k, v in rt.items()}

        def step():
            ctx.lw_update_flx_dev(stream, ncol, lm, True, 30, 47, 1e15, p1)
            ctx.sw_update_export_dev(stream, ncol, lm, 14, p2)
            ctx.rad_tendencies_dev(stream, ncol, lm, 9.80665, 1004.683, p3)

    for _ in range(a.warmup):
        step()
    ctx.check(stream)
    if a.scheme == "gridcomp":
        ctx.profile(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # the kernels run on torch's current stream
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, torch.device("cpu"))
    ctx.check(stream)
    dev_ms = ev0.elapsed_time(ev1) / a.steps
    if a.scheme == "gridcomp" and not a.no_overlap:
        dev_ms = elapsed / a.steps * 1e3          # two streams: the events of one do not bracket the other
    if rank != 0:
        return
    value = world * ncol * a.steps / elapsed
    if a.scheme == "heartbeat":
        achieved = hb_bytes / (dev_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_lw_update_flx + k_sw_update_export + k_rad_tendencies", "achieved": achieved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None, "algorithmic_bytes_per_column": hb_bytes / ncol,
                "avg_launch_ms": dev_ms, "launches": a.steps, "columns_per_launch": ncol,
                "note": "three streaming launches per step timed together with HIP events on the launch stream; algorithmic bytes = every "
                        "internal field a requested export needs read once + every export written once"}
        wl = (f"heartbeat update of {ncol} columns/GPU x {lm} layers: Update_Flx (RRTMG flavour, {len(HB_LW)} exports) + UPDATE_EXPORT flux part "
              f"({len(HB_SW)} exports) + parent heating rates ({len(HB_RT)} exports)")
        kms = {"heartbeat (3 launches)": dev_ms}
    else:
        prof = ctx.profile_read()
        # the dominant kernel's own launch duration: from two further steps on ONE stream (as in the default bench)
        prof1, nst = prof, a.steps
        if side is not None and sw_stream != stream:
            torch.cuda.synchronize()
            sw_stream = stream
            ctx.profile(True)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            prof1, nst = ctx.profile_read(), 2
        cand = {k: v for k, v in prof1.items() if k in LW_BAND_KERNELS + SW_BAND_KERNELS and v[1] > 0}
        kname = max(cand, key=lambda k: cand[k][0])
        ms, n = prof1[kname]
        abytes = (algorithmic_bytes_lw if kname.startswith("k_lw") else algorithmic_bytes_sw)(lm, a.real, aerosol)
        per_launch_s = (ms / max(n, 1)) * 1e-3
        achieved = abytes * ncol / (n / nst) / per_launch_s / 1e9
        roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                "algorithmic_bytes_per_column": abytes, "avg_launch_ms": ms / max(n, 1), "launches": n,
                "note": "same dominant kernel as the default bench; the driver adds prep / flip / post streaming kernels around the solvers"}
        wl = (f"RRTMG branches of LW_Driver + SORADCORE on GEOS-native fields (model ordering, SI units): {ncol} columns/GPU, {lm} layers, "
              f"McICA clouds on {100 * a.cloudy:.0f} % of the columns, aerosols {'on' if aerosol else 'off'}"
              + (f", RATS diagnostics for {', '.join(G.RAT_GAS[:a.rats])}" if a.rats else ""))
        kms = {k: v[0] / a.steps for k, v in prof.items() if v[1] > 0}
        kms["whole step (device)"] = dev_ms
    print(json.dumps({
        "metric": "columns/sec (LW+SW, 72 layers)" if a.scheme == "gridcomp" else "columns/sec", "value": value, "unit": "columns/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if a.real == 4 else "f64", "data": "synthetic",
        "config": {"workload": wl, "columns_per_gpu": ncol, "layers": lm, "sharding": "independent column batches per GPU, no collective"},
        "roofline": roof, "kernels_ms_per_step": kms, "cpu_baseline": cpu}))


def committed_counters(ckey, kname):
    """PMC figures of the dominant kernel for the exact configuration they were collected on (separate rocprofv3 --pmc passes, FETCH_SIZE
    x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes; profiles/tools/save_final.py builds the file from the counter dumps of the round's
    final build).  The fallback of live_counters(): no profiler on the box, a failed pass, several ranks, or bench.py itself being profiled."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r04_counters.json", "r03_counters.json", "r02_traffic.json"):
        try:
            with open(os.path.join(here, name)) as fh:
                e = json.load(fh)[ckey]
            t = e.get(kname) or (e.get("k_sw_bands") if kname == "k_sw_reform" else None)     # rounds 2-3 filed the SW band sweeps under their first name
            if t:
                return t, "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def under_profiler():
    """True when this process was started by rocprofv3 (its tool library is preloaded and has initialised the GPU already: no child
    process may be started from here)"""
    return any(k.startswith(("ROCPROF", "ROCTRACER", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


PMC_GROUPS = {"k_sw_reform": ("k_sw_reform",), "k_sw_bands": ("k_sw_bands",), "k_lw_bands": ("k_lw_bands",), "k_mcica": ("k_mcica",),
              "k_mcica_sa": ("k_mcica_sa",), "k_chou_bands": ("k_chou_bands",), "k_sorad_pass": ("k_sorad_pass",)}


def live_counters(a, argv, cache=None, configs=False, legs=None, mem_only=False):
    """The dominant kernels' PMC figures of THIS build on THIS box, collected the way MI355X_MICROARCH.md prescribes: rocprofv3 --pmc around a
    short one-stream run of the same workload, one counter group per run (FETCH_SIZE | WRITE_SIZE | the issue counters), the program directly
    behind `--`.  Runs as child processes BEFORE this process touches the GPU.  Returns {kernel group: {...}} per step, or None (no profiler,
    a failed pass): bench.py then quotes the figures committed under profiles/ and says so in traffic_source."""
    import csv, glob, shutil, subprocess, tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    steps, warm = 2, 1
    keep = [x for x in argv if x in ("--no-aerosol",)]
    child = [sys.executable, os.path.join(here, "bench.py"), "--no-pmc", "--no-cpu", "--no-parity", "--no-f64", "--no-configs", "--no-overlap",
             "--steps", str(steps), "--warmup", str(warm), "--scheme", a.scheme, "--ncol", str(a.ncol), "--nlay", str(a.nlay), "--cloudy", str(a.cloudy),
             "--real", str(a.real), "--lit", str(a.lit), "--coherent", str(a.coherent)] + keep
    if configs:             # legs of the `configs` object, 3 steps each (configs_gpu); a child run holds legs whose kernels do not share names
        child = [sys.executable, os.path.join(here, "bench.py"), "--configs-only", "--steps", str(steps), "--warmup", str(warm)]
        if legs:
            child += ["--configs-legs", ",".join(legs)]
    if cache:
        child += ["--inputs-cache", cache]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    sums = {}
    tmp = tempfile.mkdtemp(prefix="geosrad_pmc_", dir="/tmp")
    try:
        groups = (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"])
        for i, grp in enumerate(groups[:2] if mem_only else groups):
            d = os.path.join(tmp, "p%d" % i)
            # a session of its own, so that a pass that overruns can be ended together with the program it started
            pr = subprocess.Popen([prof, "--pmc"] + grp + ["-d", d, "-o", "x", "--output-format", "csv", "--"] + child, cwd="/tmp", env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=240 if configs else 120)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        kn = row["Kernel_Name"]
                        for g, pats in PMC_GROUPS.items():
                            if any(("geosrad::" + p_ + "<") in kn or ("geosrad::" + p_ + "(") in kn for p_ in pats):
                                e = sums.setdefault(g, {})
                                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nstep = steps + warm
    out = {}
    for g, c in sums.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        e = {"traffic_bytes": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / nstep}     # KB; gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md)
        if c.get("GRBM_GUI_ACTIVE"):
            e["valu_insts"] = c["SQ_INSTS_VALU"] / nstep
            e["valu_util"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 3)
            e["wait_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None
            e["resident_waves_per_simd"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 2)
        out[g] = e
    return out or None


def roofline_note(kname, two_streams):
    """what a `launch` of the dominant kernel is and why its fraction of the HBM peak on the compulsory bytes is what it is"""
    if kname in LW_BAND_KERNELS + SW_BAND_KERNELS:
        note = ("a `launch` here is the kernel's cloud-free and cloudy instantiation launched back to back over the batch (one HIP-event "
                "span; in the rocprofv3 kernel stats: the sum of the two instantiations' average durations); the fused k-distribution + "
                "vertical-sweep kernel parks per-cell state between its two sweeps, so its HBM traffic is a multiple of the compulsory "
                "bytes and the fraction on the compulsory bytes is small by construction (~170 FLOP per compulsory byte); `traffic` = PMC "
                "bytes of this kernel per step / its launches per step (profiles/, see roofline.traffic_source)")
    elif kname == "k_chou_bands":
        note = ("irrad's level-pair integration (irrad.F90:960-1294), one wavefront per (column, band): O(np^2) transmittance products per "
                "column from LDS-resident layer terms - latency / LDS bound, the compulsory bytes are a small part of its work")
    else:
        note = ("sorad's spectral passes (sorad.F90:470-1423), lane = column: deledd + CLDFLX adding with the per-level arrays of a pass in "
                "HBM scratch planes - bound by that scratch traffic, a multiple of the compulsory bytes")
    if two_streams:
        note += "; the timed steps run LW and SW on two streams, avg_launch_ms is this kernel's own duration from two further steps on one stream"
    return note


PARITY_COLS = 256
SW_BAND_KERNELS = ("k_sw_reform", "k_sw_bands")        # the default RRTMG_SW band sweeps | GEOSRAD_SW_PATH=bands (first mapping)
LW_BAND_KERNELS = ("k_lw_bands", "k_lw_cols", "k_lw_cells+k_lw_sweep")      # RRTMG_LW band sweeps: default | GEOSRAD_LW_PATH=cols | =split
PARITY_LW = ("uflx", "dflx", "uflxc", "dflxc")
PARITY_SW = ("swuflx", "swdflx", "swuflxc", "swdflxc")


def lwsw_f64_leg(inp, dev, local_rank, aerosol, do_lw, do_sw, cloudy, steps=5, warmup=3):
    """The default workload once more with real_kind 8 - the instantiation that meets BASELINE.json's 1e-6 W m-2 - on two HIP streams
    like the headline; returns (ms per step, the first PARITY_COLS columns of its fluxes on the host)."""
    import torch
    from geosradiation_gridcomp_amd.api import Context
    nlay, ncol = inp["play"].shape
    tdt = torch.float64
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat"] + LW_IN2D + (["tauaer"] if aerosol else [])
    if do_sw:
        names += SW_IN + (SW_AER if aerosol else [])
    d = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).to(dev, dtype=tdt) for k in names}
    for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs", "swuflx", "swdflx", "swuflxc", "swdflxc"):
        d[k] = torch.zeros((nlay + 1, ncol), device=dev, dtype=tdt)
    for k in SW_OUT1:
        d[k] = torch.zeros(ncol, device=dev, dtype=tdt)
    d["fswband"] = torch.zeros((14, ncol), device=dev, dtype=tdt)
    d["clearCounts"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
    d["clearCounts_sw"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
    ptr = {k: v.data_ptr() for k, v in d.items()}
    ctx = Context(8, device=local_rank)
    ctx.set_inhomogeneity(1 if cloudy > 0 else 0)
    stream = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream() if (do_lw and do_sw) else None
    sw_stream = side.cuda_stream if side is not None else stream
    doy, lm, mh = int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])

    def step():
        if do_lw:
            ctx.rrtmg_lw_dev(stream, ncol, nlay, True, ptr, 3, 1, doy, lm, mh)
        if do_sw:
            ctx.rrtmg_sw_dev(sw_stream, ncol, nlay, 1361.0, 1.0, 0, ptr, 3, 1, doy, 10 if aerosol else 0, lm, mh, normFlx=1)

    for _ in range(warmup):
        step()
    ctx.check(stream)
    if side is not None:
        ctx.check(sw_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ctx.check(stream)
    ns = min(PARITY_COLS, ncol)
    sample = {k: d[k][..., :ns].cpu().numpy() for k in PARITY_LW + PARITY_SW + ("clearCounts", "clearCounts_sw")}
    ctx.close()
    del d
    torch.cuda.empty_cache()
    return ms, sample


def lwsw_parity(inp_s, got4, got8, masks4, aerosol, do_lw, do_sw, cloudy):
    """In-run parity of the timed build: the first PARITY_COLS columns of the batch the timed steps computed (a column's arithmetic does
    not depend on its batch), against oracle/liboracle.so - the CHECKER, single-threaded, after the timed regions.  LW in W m-2;
    SW as the timed call returns it (normFlx = 1: fractions of the column's TOA flux) and, through the oracle's un-normalised TOA flux,
    in W m-2.  Columns whose clearCounts differ from the oracle's (fp32 only: a McICA overlap decision hanging on the last bit of an
    exp()) are counted and left out of the SW / total-sky figures, as in tests/test_gpu_sw.py."""
    from oracle import clib
    ih = 1 if cloudy > 0 else 0
    out = {}
    for rk, prec, got in ((4, "f32", got4), (8, "f64", got8)):
        if got is None:
            continue
        clib.lib(); clib.set_inhomogeneity(ih, prec)
        res = {"columns": int(inp_s["play"].shape[1]), "oracle": "oracle/liboracle.so (" + prec + ")"}
        if do_lw:
            o = clib.rrtmg_lw(inp_s, prec)
            same = (got["clearCounts"] == o["clearCounts"]).all(axis=0)
            e_clr = max(float(np.abs(got[k].astype(np.float64) - o[k].astype(np.float64)).max()) for k in ("uflxc", "dflxc"))
            e_tot = max(float(np.abs(got[k].astype(np.float64) - o[k].astype(np.float64))[:, same].max()) for k in ("uflx", "dflx"))
            res["lw_max_abs_Wm2"] = max(e_clr, e_tot)
            res["lw_columns_with_other_clearCounts"] = int((~same).sum())
        if do_sw:
            q = clib.rrtmg_sw(inp_s, prec=prec, iaer=10 if aerosol else 0, normFlx=1)
            q0 = clib.rrtmg_sw(inp_s, prec=prec, iaer=10 if aerosol else 0, normFlx=0)
            toa = q0["swdflx"][-1].astype(np.float64)
            same = (got["clearCounts_sw"] == q["clearCounts"]).all(axis=0)
            rel = np.zeros(res["columns"])
            for k in PARITY_SW:
                dk = np.abs(got[k].astype(np.float64) - q[k].astype(np.float64)).max(axis=0)
                rel = np.maximum(rel, dk if k.endswith("c") else np.where(same, dk, 0.0))
            res["sw_max_rel_toa"] = float(rel.max())
            res["sw_max_abs_Wm2"] = float((rel * toa).max())
            res["sw_columns_with_other_clearCounts"] = int((~same).sum())
        if rk == 4 and masks4 is not None:
            flips = cells = 0
            for nsub, so, m in masks4:
                rl, _, _ = clib.mcica(inp_s["zm"], inp_s["alat"], int(inp_s["dyofyr"]), inp_s["play"], inp_s["cldf"], inp_s["ciwp"],
                                      inp_s["clwp"], nsub, seed_order=so, prec=prec)
                flips += int((m != rl.astype(np.int32)).sum())
                cells += int((inp_s["cldf"] > 0).sum()) * nsub
            res["mcica_mask_flip_rate"] = flips / max(cells, 1)
            res["mcica_cells_with_a_decision"] = cells
        clib.set_inhomogeneity(0, prec)
        # the bounds the GPU tests assert (tests/test_gpu_fullsize.py::test_c360_share_lw_sw, tests/test_gpu_sw.py): fp64 = BASELINE.json's
        # 1e-6 W m-2; fp32 LW 4e-3 W m-2 on cloudy columns with aerosols, fp32 SW 5e-3 of the column's TOA flux for every column
        tol = {"lw_max_abs_Wm2": 4e-3 if rk == 4 else 1e-6, ("sw_max_rel_toa" if rk == 4 else "sw_max_abs_Wm2"): 5e-3 if rk == 4 else 1e-6}
        res["tolerance"] = tol
        res["within_tolerance"] = all(res[k] <= v for k, v in tol.items() if k in res)
        out["f32" if rk == 4 else "f64"] = res
    return out


LW_FILE_ORDER = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr",
                 "cfc12vmr", "cfc22vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel", "tauaer", "zm", "alat"]
SW_FILE_ORDER = ["coszen", "play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel",
                 "zm", "alat", "tauaer_sw", "ssaaer_sw", "asmaer_sw", "asdir", "asdif", "aldir", "aldif"]


def bench_ranks_per_gpu(a):
    """--ranks-per-gpu K1,K2,...: what the MPI ranks of a GEOS job sharing one GPU get (GEOS_SolarGridComp.F90:3701-3709 balances the
    work of many ranks per node; each is its own process with its own context).  For every K the batch is cut into K shards and K
    CHILD processes - started before anything in this process touches the GPU - each run the Fortran callers of the drop-in
    (fortran/lw_driver.F90, sw_driver.F90: the reference's module names and argument lists, host arrays in and out) on one shard,
    concurrently, all on ONE device (GEOSRAD_DEVICE, default 0, in the children's environment - whatever the node holds; the node-local
    rank -> device choice of an MPI job is tests/test_host.py::test_device_choice_follows_the_local_mpi_rank's).  Reported: the caller-side time of one rrtmg_lw / rrtmg_sw call of the slowest rank, the aggregate rate."""
    import re
    import subprocess
    import tempfile
    from geosradiation_gridcomp_amd import synth
    fdir = os.path.join(ROOT, "geosradiation_gridcomp_amd", "fortran", "bin")
    exe_lw, exe_sw = os.path.join(fdir, "lw_driver_r4"), os.path.join(fdir, "sw_driver_r4")
    if not (os.path.exists(exe_lw) and os.path.exists(exe_sw)):
        sys.exit("bench.py --ranks-per-gpu: the Fortran drivers are not built (python -c 'import __graft_entry__ as g; g.build()')")
    ncol, nlay, aerosol = a.ncol, a.nlay, not a.no_aerosol
    inp = synth.make_columns(ncol, nlay, start=0, cloudy_frac=a.cloudy, aerosol=True)
    if not aerosol:
        for k in ("tauaer", "tauaer_sw"):
            inp[k] = np.zeros_like(inp[k])
    ih = 1 if a.cloudy > 0 else 0
    env0 = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    env0["GEOSRAD_DEVICE"] = os.environ.get("GEOSRAD_DEVICE", "0")        # "K processes sharing ONE GPU": every child on the same device
    reps = max(3, a.steps)
    res = {}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as tmp:
        ks = [int(x) for x in a.ranks_per_gpu.split(",")]
        if max(ks) > 6:
            sys.exit("bench.py --ranks-per-gpu: at most 6 processes may share the GPU of a box of this pool")
        for K in ks:
            per = (ncol + K - 1) // K
            files = []
            for r in range(K):
                sl = slice(r * per, min(ncol, (r + 1) * per))
                n = sl.stop - sl.start
                cut = lambda v: np.ascontiguousarray(v[..., sl], dtype=np.float32)
                fl, fs = os.path.join(tmp, f"lw_{K}_{r}.bin"), os.path.join(tmp, f"sw_{K}_{r}.bin")
                with open(fl, "wb") as f:
                    np.array([n, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])], dtype=np.int32).tofile(f)
                    for k in LW_FILE_ORDER:
                        cut(inp[k]).tofile(f)
                with open(fs, "wb") as f:
                    np.array([n, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]), 10 if aerosol else 0, 1, 0], dtype=np.int32).tofile(f)
                    np.array([1361.0], dtype=np.float32).tofile(f)
                    for k in SW_FILE_ORDER:
                        cut(inp[k]).tofile(f)
                files.append((fl, fs, n))
            out = {}
            for which, exe, ix in (("rrtmg_lw", exe_lw, 0), ("rrtmg_sw", exe_sw, 1)):
                t0 = time.perf_counter()
                procs = [subprocess.Popen([exe, files[r][ix], os.path.join(tmp, f"out_{r}.bin"), str(reps)], stdout=subprocess.PIPE, text=True,
                                          env=dict(env0, OMPI_COMM_WORLD_LOCAL_RANK=str(r), OMPI_COMM_WORLD_LOCAL_SIZE=str(K))) for r in range(K)]
                ms = []
                for pr in procs:
                    txt = pr.communicate()[0]
                    if pr.returncode != 0:
                        sys.exit(f"bench.py --ranks-per-gpu: {which} rank failed (rc {pr.returncode}): {txt[-300:]}")
                    ms.append(float(re.search(r"ms per call\s+([0-9.]+)", txt).group(1)))
                out[which] = {"ms_per_call_slowest_rank": max(ms), "ms_per_call_mean": sum(ms) / len(ms), "wall_s_incl_file_io_and_init": time.perf_counter() - t0}
            tot = out["rrtmg_lw"]["ms_per_call_slowest_rank"] + out["rrtmg_sw"]["ms_per_call_slowest_rank"]
            out["columns_per_rank"] = per
            out["aggregate_columns_per_s"] = ncol / (tot * 1e-3)
            res[str(K)] = out
            for fl, fs, _ in files:
                os.remove(fl); os.remove(fs)
    best = max(res.values(), key=lambda v: v["aggregate_columns_per_s"])
    print(json.dumps({
        "metric": "columns/sec (LW+SW, 72 layers)", "value": best["aggregate_columns_per_s"], "unit": "columns/s", "n_gpus": 1, "steps": reps,
        "warmup": 1, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Fortran drop-in (rrtmg_lw + rrtmg_sw, host arrays, PCIe included), {ncol} columns x {nlay} layers cut into K shards, "
                               f"K processes sharing one GPU, McICA clouds on {100 * a.cloudy:.0f} % of the columns, aerosols {'on' if aerosol else 'off'}",
                   "columns": ncol, "layers": nlay},
        "ranks_per_gpu": res}))


def _mcica_cpu_worker(args):
    start, ncol, nlay, nsub, cloudy = args
    from geosradiation_gridcomp_amd import synth
    from oracle import reflib, clib
    batch, busy = 2048, 0.0              # outputs are 12 B x nsub x nlay per column: batches keep a process below 0.4 GB
    for b0 in range(0, ncol, batch):
        inp = synth.make_columns(min(batch, ncol - b0), nlay, start=start + b0, cloudy_frac=cloudy, aerosol=False)
        a = (inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"], inp["ciwp"], inp["clwp"], nsub)
        if reflib.available("r4"):
            reflib.lib("r4"); reflib.set_inhomogeneity(1, "r4")
            t = time.perf_counter(); reflib.mcica(*a, kind="r4")
        else:
            clib.lib(); clib.set_inhomogeneity(1, "f32")
            t = time.perf_counter(); clib.mcica(*a, prec="f32")
        busy += time.perf_counter() - t
    return busy


def mcica_cpu_baseline(nlay, nsub, cloudy, per_core=14336):
    import multiprocessing as mp
    from oracle import reflib
    kind = "reference" if reflib.available("r4") else "port"
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    with mp.get_context("fork").Pool(cores) as pool:
        per = pool.map(_mcica_cpu_worker, [(20_000_000 + i * per_core, per_core, nlay, nsub, cloudy) for i in range(cores)])
    return {"value": cores * per_core / max(per), "unit": "columns/s", "cores": cores, "kind": kind,
            "sample": f"{cores} processes x {per_core} columns, generate_stochastic_clouds(nsubcol={nsub}) "
                      f"({'reference Fortran, oracle/_ref' if kind == 'reference' else 'plain-C oracle'}), {nlay} layers, cloudy fraction "
                      f"{cloudy}, ih=1; slowest process {max(per):.2f} s",
            "single_core_columns_per_s": per_core / (sum(per) / len(per))}


def bench_mcica(a, rank, world, dev, local_rank, cpu):
    """--scheme mcica: the stand-alone generator API (cloud_subcol_gen.F90:132, generate_stochastic_clouds) with BASELINE configs[2]'s
    200 sub-columns: every cell of cldy_stoch (Fortran logical, 4 B), ciwp_stoch, clwp_stoch materialised in HBM.  The one unit of
    work of SURVEY 8(d) whose compulsory bytes (5 layer fields in, 12 B per (layer, sub-column) cell out) are of the order of
    its arithmetic: the HBM roofline is the relevant bound."""
    import torch
    import torch.distributed as dist
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context
    ncol, nlay, nsub = a.ncol, a.nlay, a.nsubcol
    tdt = torch.float32 if a.real == 4 else torch.float64
    inp = synth.make_columns(ncol, nlay, start=shard_start(rank, ncol), cloudy_frac=a.cloudy, aerosol=False)
    d = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).to(dev, dtype=tdt) for k in ("zm", "alat", "play", "cldf", "ciwp", "clwp")}
    d["cldy_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=torch.int32, device=dev)
    d["ciwp_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=tdt, device=dev)
    d["clwp_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=tdt, device=dev)
    ptr = {k: v.data_ptr() for k, v in d.items()}
    ctx = Context(a.real, device=local_rank)
    ctx.set_inhomogeneity(1)
    stream = torch.cuda.current_stream().cuda_stream
    doy = int(inp["dyofyr"])

    def step():
        ctx.generate_stochastic_clouds_dev(stream, ncol, nsub, nlay, ptr, doy, 1e-20)

    for _ in range(a.warmup):
        step()
    ctx.check(stream)
    ctx.profile(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, torch.device("cpu"))
    ctx.check(stream)
    prof = ctx.profile_read()
    if rank != 0:
        return
    ms, n = prof["k_mcica"]
    abytes = (5 * nlay + 1) * a.real + nsub * nlay * (4 + 2 * a.real)      # zm, play, cldf, ciwp, clwp, alat in; mask + 2 water paths out
    per_launch_s = ms / max(n, 1) * 1e-3
    achieved = abytes * ncol / per_launch_s / 1e9
    frac_cloudy_cells = float((d["cldy_stoch"][: min(ncol, 2000)] != 0).float().mean())
    print(json.dumps({
        "metric": "columns/sec", "value": world * ncol * a.steps / elapsed, "unit": "columns/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 (KISS) + " + ("f32" if a.real == 4 else "f64"), "data": "synthetic",
        "config": {"workload": f"McICA generator stand-alone (generate_stochastic_clouds, BASELINE configs[2]): {ncol} columns/GPU, {nlay} layers, "
                               f"{nsub} sub-columns, clouds on {100 * a.cloudy:.0f} % of the columns, beta-PDF condensate inhomogeneity (ih=1); "
                               f"{100 * frac_cloudy_cells:.1f} % of the cells cloudy",
                   "columns_per_gpu": ncol, "layers": nlay, "sub_columns": nsub, "sharding": "independent column batches per GPU, no collective"},
        "roofline": {"bound": "hbm", "kernel": "k_mcica", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": 18.6e9 if (ncol, nlay, nsub, a.cloudy, a.real) == (100_000, 72, 200, 0.6, 4) else None,      # profiles/r01_v11_mcica.md
                     "algorithmic_bytes_per_column": abytes, "avg_launch_ms": ms / max(n, 1), "launches": n,
                     "columns_per_launch": ncol,
                     "note": "algorithmic bytes = the five layer fields + latitude read once, every (layer, sub-column) cell of the three "
                             "outputs written once (the Fortran logical mask is 4 B); launch duration from HIP events on the launch stream"},
        "kernels_ms_per_step": {k: v[0] / a.steps for k, v in prof.items() if v[1] > 0}, "cpu_baseline": cpu}))


# ======================================================================================================================================
# BASELINE.json configs[1] and configs[2] in the default line ("configs"): 100 000 columns x 72 layers each, one launch stream, fp32
#   cfg1_lw_clear_100k : RRTMG_LW 140 g-points, clear-sky, no aerosol array                            (configs[1])
#   cfg2_sw_noaer_100k : RRTMG_SW 112 g-points, McICA clouds on 60 % of the columns (ih = 1), no aerosol (configs[2])
#   cfg2_sorad_100k    : Chou-Suarez sorad on the same columns (8 bands, aerosols)                     (configs[2])
#   cfg2_irrad_100k    : Chou-Suarez irrad on the same columns (10 bands, trace gases, aerosols)       (configs[0]'s scheme at configs[2]'s size)
#   cfg2_mcica_200     : the McICA generator stand-alone, 200 sub-columns                              (configs[2])
#   cfg0_irrad_1000_clear : Chou-Suarez irrad on 1 000 clear-sky columns                                  (configs[0]: the reference's CPU-runnable case)
#   cfg4_c720_share_137l_rrtmg_standin : RRTMG_LW + RRTMG_SW on configs[4]'s per-GPU share, 388 800 columns x 137 layers (C720 tile / 8), McICA
#                        clouds + aerosols.  configs[4] names the RRTMGP k-distribution path, whose source and coefficient files are not in
#                        the reference repository: this leg is the RRTMG kernels at that size - the HBM-pressure / deep-atmosphere
#                        stress, NOT an RRTMGP number.  The batch is 16 copies of 24 300 generated columns (columns are independent).
# Each entry: value / ms_per_step from CFG_WARMUP + CFG_STEPS steps timed on the host around a device synchronisation; roofline of the leg's
# dominant kernel (HIP-event launch duration from the same steps, algorithmic bytes of SURVEY 8(d), PMC traffic from a rocprofv3 child run of
# `bench.py --configs-only`), a cpu_baseline from one pool over the host cores, in-run parity of CFG_PARITY columns against the oracle.
# ======================================================================================================================================
CFG_NCOL, CFG_NLAY, CFG_NSUB = 100_000, 72, 200
CFG_WARMUP, CFG_STEPS = 3, 5
CFG_PARITY = 64
CFG_CPU_COLS = 1024
CFG_START = 30_000_000              # first global column of the configs' batch (disjoint from the headline batch and the CPU samples)
CFG_NAMES = ("cfg0_irrad_1000_clear", "cfg1_lw_clear_100k", "cfg2_sw_noaer_100k", "cfg2_sorad_100k", "cfg2_irrad_100k", "cfg2_mcica_200",
             "cfg4_c720_share_137l_rrtmg_standin")
CFG0_NCOL = 1000
CFG4_BASE, CFG4_TILES, CFG4_NLAY, CFG4_START = 24_300, 16, 137, 50_000_000          # 16 x 24 300 = 388 800 = 6 x 720^2 / 8
CFG4_STEPS, CFG4_WARMUP = 3, 2
IRRAD_IN = ("ple", "ta", "wa", "oa", "tb", "n2o", "ch4", "cfc11", "cfc12", "cfc22", "cwc", "fcld", "reff", "fs", "tg", "eg", "tv", "ev", "rv",
            "taua", "ssaa", "asya")
IRRAD_OUT = ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts")
SORAD_IN = ("cosz", "pl", "ta", "wa", "oa", "cwc", "fcld", "reff", "taua", "ssaa", "asya", "rsuvbm", "rsuvdf", "rsirbm", "rsirdf")
SW_LAY = ["play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel", "zm", "alat"]


def cfg_algorithmic_bytes(name, nlay=CFG_NLAY, nsub=CFG_NSUB, real_bytes=4):
    """compulsory bytes per column at the solver API: every input once + every output once"""
    if name == "cfg0_irrad_1000_clear":
        return 19476 * real_bytes // 4
    if name == "cfg4_c720_share_137l_rrtmg_standin":          # the dominant kernel's solver: RRTMG_SW with aerosols at 137 layers
        return algorithmic_bytes_sw(CFG4_NLAY, real_bytes, True)
    if name == "cfg1_lw_clear_100k":
        return algorithmic_bytes_lw(nlay, real_bytes, False)               # 7 608 @72 layers
    if name == "cfg2_sw_noaer_100k":
        return algorithmic_bytes_sw(nlay, real_bytes, False)               # 5 356
    if name == "cfg2_sorad_100k":
        return 11904 * real_bytes // 4                                     # 2 670 in + 306 out reals
    if name == "cfg2_irrad_100k":
        return 19476 * real_bytes // 4                                     # 3 491 in + 1 378 out reals
    return (5 * nlay + 1) * real_bytes + nsub * nlay * (4 + 2 * real_bytes)    # generator: 174 244


def cfg_inputs(ncol, start, want_chou=True):
    """the two column batches of the configs: cloud-free without aerosols (configs[1]) and 60 % cloudy with aerosols (configs[2]; the
    Chou-Suarez inputs are the same columns in that scheme's units and ordering)"""
    from geosradiation_gridcomp_amd import synth
    clr = synth.make_columns(ncol, CFG_NLAY, start=start, cloudy_frac=0.0, aerosol=False)
    cld = synth.make_columns(ncol, CFG_NLAY, start=start, cloudy_frac=0.6, aerosol=True)
    ch = synth.chou_lw_inputs(cld, aerosol=True) if want_chou else None
    cs = synth.chou_sw_inputs(cld, aerosol=True) if want_chou else None
    return clr, cld, ch, cs


def _cache_put(cdir, key, d):
    os.makedirs(os.path.join(cdir, key), exist_ok=True)
    scal = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1:
            np.save(os.path.join(cdir, key, k + ".npy"), v)
        else:
            scal[k] = (float(v) if isinstance(v, (float, np.floating)) else int(v))
    with open(os.path.join(cdir, key, "scalars.json"), "w") as fh:
        json.dump(scal, fh)


def _cache_get(cdir, key):
    d = os.path.join(cdir or "", key)
    if not cdir or not os.path.exists(os.path.join(d, "scalars.json")):
        return None
    with open(os.path.join(d, "scalars.json")) as fh:
        out = dict(json.load(fh))
    for f in os.listdir(d):
        if f.endswith(".npy"):
            out[f[:-4]] = np.load(os.path.join(d, f), mmap_mode="r")
    return out


def _cfg_cpu_worker(args):
    start, n = args
    from geosradiation_gridcomp_amd import synth
    from oracle import clib, reflib
    clr, cld, ch, cs = cfg_inputs(n, start)
    t = {}
    ref = reflib.available("r4")
    clib.lib()
    if ref:
        reflib.lib("r4"); reflib.set_inhomogeneity(0, "r4")
        t0 = time.perf_counter(); reflib.rrtmg_lw(clr, "r4", psize=4); t["cfg1_lw_clear_100k"] = time.perf_counter() - t0
    else:
        clib.set_inhomogeneity(0, "f32")
        t0 = time.perf_counter(); clib.rrtmg_lw(clr, "f32"); t["cfg1_lw_clear_100k"] = time.perf_counter() - t0
    clib.set_inhomogeneity(1, "f32")
    t0 = time.perf_counter(); clib.rrtmg_sw(cld, prec="f32", iaer=0, normFlx=1); t["cfg2_sw_noaer_100k"] = time.perf_counter() - t0
    t0 = time.perf_counter(); clib.sorad(cs, "f32"); t["cfg2_sorad_100k"] = time.perf_counter() - t0
    t0 = time.perf_counter(); clib.irrad(ch, "f32"); t["cfg2_irrad_100k"] = time.perf_counter() - t0
    a = (cld["zm"], cld["alat"], int(cld["dyofyr"]), cld["play"], cld["cldf"], cld["ciwp"], cld["clwp"], CFG_NSUB)
    if ref:
        reflib.set_inhomogeneity(1, "r4")
        t0 = time.perf_counter(); reflib.mcica(*a, kind="r4"); t["cfg2_mcica_200"] = time.perf_counter() - t0
    else:
        t0 = time.perf_counter(); clib.mcica(*a, prec="f32"); t["cfg2_mcica_200"] = time.perf_counter() - t0
    # configs[0]: irrad on clear-sky columns (the clear batch in Chou units)
    ch0 = synth.chou_lw_inputs(clr, aerosol=False)
    t0 = time.perf_counter(); clib.irrad(ch0, "f32"); t["cfg0_irrad_1000_clear"] = time.perf_counter() - t0
    # configs[4] stand-in: RRTMG LW + SW at 137 layers on half as many columns
    n4 = max(64, n // 2)
    deep = synth.make_columns(n4, CFG4_NLAY, start=start, cloudy_frac=0.6, aerosol=True)
    t0 = time.perf_counter()
    if ref:
        reflib.set_inhomogeneity(1, "r4"); reflib.rrtmg_lw(deep, "r4", psize=4)
    else:
        clib.set_inhomogeneity(1, "f32"); clib.rrtmg_lw(deep, "f32")
    clib.set_inhomogeneity(1, "f32")
    clib.rrtmg_sw(deep, prec="f32", iaer=10, normFlx=1)
    t["cfg4_c720_share_137l_rrtmg_standin"] = (time.perf_counter() - t0) * (n / n4)          # per n columns, like the other legs
    return t


def configs_cpu_baseline(per_core=CFG_CPU_COLS):
    """one pool over the host cores, every process runs every leg on its own `per_core` columns"""
    import multiprocessing as mp
    from oracle import reflib
    ref = reflib.available("r4")
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    with mp.get_context("fork").Pool(cores) as pool:
        per = pool.map(_cfg_cpu_worker, [(40_000_000 + i * per_core, per_core) for i in range(cores)])
    what = {"cfg1_lw_clear_100k": ("reference" if ref else "port", "rrtmg_lw = " + ("reference Fortran (oracle/_ref), psize=4" if ref else "plain-C oracle")),
            "cfg2_sw_noaer_100k": ("port", "rrtmg_sw = plain-C oracle (the reference driver needs ESMF/MAPL: unbuildable here)"),
            "cfg2_sorad_100k": ("port", "sorad = plain-C oracle (sorad.F90 needs MAPL: unbuildable here)"),
            "cfg2_irrad_100k": ("port", "irrad = plain-C oracle (irrad.F90 needs MAPL: unbuildable here)"),
            "cfg2_mcica_200": ("reference" if ref else "port", f"generate_stochastic_clouds(nsubcol={CFG_NSUB}) = "
                               + ("reference Fortran (oracle/_ref)" if ref else "plain-C oracle")),
            "cfg0_irrad_1000_clear": ("port", "irrad = plain-C oracle (irrad.F90 needs MAPL: unbuildable here), clear-sky columns"),
            "cfg4_c720_share_137l_rrtmg_standin": ("port", f"{CFG4_NLAY} layers, half the columns per process: rrtmg_lw = "
                                                   + ("reference Fortran" if ref else "plain-C oracle") + ", rrtmg_sw = plain-C oracle")}
    out = {}
    for name in CFG_NAMES:
        busy = max(p[name] for p in per); mean = sum(p[name] for p in per) / len(per)
        out[name] = {"value": cores * per_core / busy, "unit": "columns/s", "cores": cores, "kind": what[name][0],
                     "sample": f"{cores} processes x {per_core} columns of the config's workload; {what[name][1]}; slowest process {busy:.2f} s",
                     "single_core_columns_per_s": per_core / mean}
    return out


def configs_gpu(dev, local_rank, start, ncol, warmup, steps, cache=None, want_parity=True, only=None):
    """the configuration legs on the device, one after the other, one stream; returns {name: {...}} with the leg's timings, the
    profile of its kernels and (want_parity) the first CFG_PARITY columns of its outputs"""
    import torch
    from geosradiation_gridcomp_amd.api import Context
    nlay = CFG_NLAY
    tdt = torch.float32
    got = _cache_get(cache, "cfg")
    if got is not None:
        split = lambda pre: {k[len(pre):]: v for k, v in got.items() if k.startswith(pre)}
        clr, cld, ch, cs = split("clr_"), split("cld_"), split("ch_"), split("cs_")
    else:
        clr, cld, ch, cs = cfg_inputs(ncol, start)
    stream = torch.cuda.current_stream().cuda_stream
    to = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev, dtype=tdt)
    zeros = lambda *sh: torch.zeros(*sh, device=dev, dtype=tdt)
    ns = min(CFG_PARITY, ncol)
    ctx = Context(4, device=local_rank)
    res = {}

    def run(name, step, kernel, sample):
        for _ in range(warmup):
            step()
        ctx.check(stream)
        ctx.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        ctx.check(stream)
        prof = ctx.profile_read()
        ctx.profile(False)
        if isinstance(kernel, tuple):                     # the band sweeps have two mappings (GEOSRAD_SW_PATH): the one that ran
            kernel = next((k for k in kernel if prof.get(k, (0, 0))[1] > 0), kernel[0])
        slot = "k_mcica" if kernel == "k_mcica_sa" else kernel            # the generator's two kernels share a profile slot
        ms, n = prof.get(slot, (0.0, 0))
        res[name] = {"ms_per_step": dt_ / steps * 1e3, "value": ncol / (dt_ / steps), "kernel": kernel, "kernel_ms_per_step": ms / steps,
                     "kernel_launches_per_step": n / steps, "kernels_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1] > 0},
                     "sample": sample() if want_parity else None}

    doy, lm, mh = int(cld["dyofyr"]), int(cld["cloudLM"]), int(cld["cloudMH"])
    # ---- configs[1]: RRTMG_LW clear-sky -------------------------------------------------------------------------------------------
    if only is None or "cfg1_lw_clear_100k" in only:
        d = {k: to(clr[k]) for k in ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat"] + LW_IN2D}
        for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs"):
            d[k] = zeros(nlay + 1, ncol)
        d["clearCounts"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        ctx.set_inhomogeneity(0)
        run("cfg1_lw_clear_100k", lambda: ctx.rrtmg_lw_dev(stream, ncol, nlay, True, ptr, 3, 1, int(clr["dyofyr"]), int(clr["cloudLM"]), int(clr["cloudMH"])),
            "k_lw_bands", lambda: {k: d[k][..., :ns].cpu().numpy() for k in PARITY_LW})
        del d, ptr
    # ---- configs[2]: RRTMG_SW with McICA clouds, no aerosol --------------------------------------------------------------------------
    if only is None or "cfg2_sw_noaer_100k" in only:
        d = {k: to(cld[k]) for k in SW_LAY + SW_IN}
        for k in PARITY_SW:
            d[k] = zeros(nlay + 1, ncol)
        for k in SW_OUT1:
            d[k] = zeros(ncol)
        d["fswband"] = zeros(14, ncol)
        d["clearCounts_sw"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        ctx.set_inhomogeneity(1)
        run("cfg2_sw_noaer_100k", lambda: ctx.rrtmg_sw_dev(stream, ncol, nlay, 1361.0, 1.0, 0, ptr, 3, 1, doy, 0, lm, mh, normFlx=1),
            SW_BAND_KERNELS, lambda: {k: d[k][..., :ns].cpu().numpy() for k in PARITY_SW + ("clearCounts_sw",)})
        del d, ptr
    # ---- configs[2]: sorad ---------------------------------------------------------------------------------------------------------------
    if only is None or "cfg2_sorad_100k" in only:
        d = {k: to(cs[k]) for k in SORAD_IN}
        for k in ("flx", "flc", "flxu", "flcu"):
            d[k] = zeros(nlay + 1, ncol)
        for k in ("fdiruv", "fdifuv", "fdirpar", "fdifpar", "fdirir", "fdifir"):
            d[k] = zeros(ncol)
        d["flx_sfc_band"] = zeros(8, ncol)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        run("cfg2_sorad_100k", lambda: ctx.sorad_dev(stream, ncol, nlay, 8, ptr, cs["co2"], cs["ict"], cs["icb"], cs["hk_uv"], cs["hk_ir"]),
            "k_sorad_pass", lambda: {k: d[k][..., :ns].cpu().numpy() for k in ("flx", "flc", "flxu", "flcu")})
        del d, ptr
    # ---- irrad (taua / ssaa / asya are in-out, rescaled in place: the inputs are restored every step, inside the timed region) ------------
    if only is None or "cfg2_irrad_100k" in only:
        d = {k: to(ch[k]) for k in IRRAD_IN}
        for k in IRRAD_OUT:
            d[k] = zeros(nlay + 1, ncol)
        d["sfcem"] = zeros(ncol); d["taudiag"] = zeros(10, nlay, ncol)
        aer0 = {k: d[k].clone() for k in ("taua", "ssaa", "asya")}
        ptr = {k: v.data_ptr() for k, v in d.items()}

        def irrad_step():
            for k in aer0:
                d[k].copy_(aer0[k])
            ctx.irrad_dev(stream, ncol, nlay, ptr, ch["co2"], True, ch["ict"], ch["icb"], ch["ns"], ch["na"], ch["nb"])
        run("cfg2_irrad_100k", irrad_step, "k_chou_bands", lambda: {k: d[k][..., :ns].cpu().numpy() for k in ("flxu", "flxd", "flcu", "flcd")})
        del d, ptr, aer0
    # ---- configs[2]: the generator stand-alone, 200 sub-columns ------------------------------------------------------------------------
    if only is None or "cfg2_mcica_200" in only:
        d = {k: to(cld[k]) for k in ("zm", "alat", "play", "cldf", "ciwp", "clwp")}
        d["cldy_stoch"] = torch.zeros((ncol, CFG_NSUB, nlay), dtype=torch.int32, device=dev)
        d["ciwp_stoch"] = zeros(ncol, CFG_NSUB, nlay); d["clwp_stoch"] = zeros(ncol, CFG_NSUB, nlay)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        ctx.set_inhomogeneity(1)
        run("cfg2_mcica_200", lambda: ctx.generate_stochastic_clouds_dev(stream, ncol, CFG_NSUB, nlay, ptr, doy, 1e-20),
            "k_mcica_sa", lambda: {k: d[k][:ns].cpu().numpy() for k in ("cldy_stoch", "ciwp_stoch", "clwp_stoch")})
        del d, ptr
    # ---- configs[0]: irrad, 1 000 clear-sky columns (its own size; value = columns of THIS leg / time) ----------------------------------
    if only is None or "cfg0_irrad_1000_clear" in only:
        from geosradiation_gridcomp_amd import synth
        n0 = min(CFG0_NCOL, ncol)
        ch0 = synth.chou_lw_inputs({k: (np.ascontiguousarray(v[..., :n0]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v)
                                    for k, v in clr.items()}, aerosol=False)
        d = {k: to(ch0[k]) for k in IRRAD_IN}
        for k in IRRAD_OUT:
            d[k] = zeros(nlay + 1, n0)
        d["sfcem"] = zeros(n0); d["taudiag"] = zeros(10, nlay, n0)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        ncol_save, ncol = ncol, n0          # run() reports columns / time of the leg's own batch
        run("cfg0_irrad_1000_clear", lambda: ctx.irrad_dev(stream, n0, nlay, ptr, ch0["co2"], True, ch0["ict"], ch0["icb"], ch0["ns"], ch0["na"], ch0["nb"]),
            "k_chou_bands", lambda: {k: d[k][..., :min(ns, n0)].cpu().numpy() for k in ("flxu", "flxd", "flcu", "flcd")})
        res["cfg0_irrad_1000_clear"]["columns"] = n0
        ncol = ncol_save
        del d, ptr
    # ---- configs[4] stand-in: RRTMG LW + SW, 388 800 columns x 137 layers (16 copies of 24 300 generated columns), one stream ------------
    if (only is None or "cfg4_c720_share_137l_rrtmg_standin" in only) and ncol == CFG_NCOL:
        from geosradiation_gridcomp_amd import synth
        deep = _cache_get(cache, "cfg4")
        if deep is None:
            deep = synth.make_columns(CFG4_BASE, CFG4_NLAY, start=CFG4_START, cloudy_frac=0.6, aerosol=True)
        n4, l4 = CFG4_BASE * CFG4_TILES, CFG4_NLAY
        tile = lambda v: to(v).repeat(*([1] * (np.ndim(v) - 1)), CFG4_TILES)
        names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat", "tauaer"] + LW_IN2D + SW_IN + SW_AER
        d = {k: tile(deep[k]) for k in names}
        for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs") + PARITY_SW:
            d[k] = zeros(l4 + 1, n4)
        for k in SW_OUT1:
            d[k] = zeros(n4)
        d["fswband"] = zeros(14, n4)
        d["clearCounts"] = torch.zeros((4, n4), device=dev, dtype=torch.int32)
        d["clearCounts_sw"] = torch.zeros((4, n4), device=dev, dtype=torch.int32)
        ptr = {k: v.data_ptr() for k, v in d.items()}
        ctx.set_inhomogeneity(1)
        doy4, lm4, mh4 = int(deep["dyofyr"]), int(deep["cloudLM"]), int(deep["cloudMH"])

        def deep_step():
            ctx.rrtmg_lw_dev(stream, n4, l4, True, ptr, 3, 1, doy4, lm4, mh4)
            ctx.rrtmg_sw_dev(stream, n4, l4, 1361.0, 1.0, 0, ptr, 3, 1, doy4, 10, lm4, mh4, normFlx=1)
        ncol_save, nlay_save, ncol = ncol, nlay, n4
        w_save, s_save = warmup, steps
        warmup, steps = min(warmup, CFG4_WARMUP), min(steps, CFG4_STEPS)
        run("cfg4_c720_share_137l_rrtmg_standin", deep_step, SW_BAND_KERNELS,
            lambda: {k: d[k][..., :ns].cpu().numpy() for k in PARITY_LW + PARITY_SW + ("clearCounts", "clearCounts_sw")})
        res["cfg4_c720_share_137l_rrtmg_standin"].update(columns=n4, layers=l4, steps=steps, warmup=warmup)
        ncol, nlay, warmup, steps = ncol_save, nlay_save, w_save, s_save
        del d, ptr
    ctx.close()
    torch.cuda.empty_cache()
    return res


def configs_parity(res, start):
    """CFG_PARITY columns of every leg's output against oracle/liboracle.so (the checker; single-threaded, after the timed regions)"""
    from oracle import clib
    from geosradiation_gridcomp_amd import synth
    ns = CFG_PARITY
    clr, cld, ch, cs = cfg_inputs(ns, start)
    clib.lib()
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    out = {}
    g = res.get("cfg1_lw_clear_100k", {}).get("sample")
    if g:
        clib.set_inhomogeneity(0, "f32")
        o = clib.rrtmg_lw(clr, "f32")
        out["cfg1_lw_clear_100k"] = {"columns": ns, "max_abs_Wm2": max(float(np.abs(f64(g[k]) - f64(o[k])).max()) for k in PARITY_LW),
                                     "tolerance_Wm2": 2e-3, "oracle": "oracle/liboracle.so (f32), pinned bit for bit to the reference's rrtmg_lw"}
    g = res.get("cfg2_sw_noaer_100k", {}).get("sample")
    if g:
        clib.set_inhomogeneity(1, "f32")
        q = clib.rrtmg_sw(cld, prec="f32", iaer=0, normFlx=1)
        same = (g["clearCounts_sw"] == q["clearCounts"]).all(axis=0)
        rel = np.zeros(ns)
        for k in PARITY_SW:
            dk = np.abs(f64(g[k]) - f64(q[k])).max(axis=0)
            rel = np.maximum(rel, dk if k.endswith("c") else np.where(same, dk, 0.0))
        out["cfg2_sw_noaer_100k"] = {"columns": ns, "max_rel_toa": float(rel.max()), "tolerance_rel_toa": 5e-4,
                                     "columns_with_other_clearCounts": int((~same).sum()),
                                     "oracle": "oracle/liboracle.so (f32): setcoef / taumol / McICA / cldprmc pinned to the reference, two-stream + driver parity unpinned"}
        clib.set_inhomogeneity(0, "f32")
    g = res.get("cfg2_sorad_100k", {}).get("sample")
    if g:
        o = clib.sorad(cs, "f32")
        out["cfg2_sorad_100k"] = {"columns": ns, "max_abs_fraction_of_insolation": max(float(np.abs(f64(g[k]) - f64(o[k])).max()) for k in g),
                                  "tolerance": 2e-5, "oracle": "oracle/liboracle.so (f32), parity unpinned (sorad.F90 needs MAPL)"}
    g = res.get("cfg2_irrad_100k", {}).get("sample")
    if g:
        o = clib.irrad(ch, "f32")
        out["cfg2_irrad_100k"] = {"columns": ns, "max_abs_Wm2": max(float(np.abs(f64(g[k]) - f64(o[k])).max()) for k in g),
                                  "tolerance_Wm2": 2e-2, "oracle": "oracle/liboracle.so (f32), parity unpinned (irrad.F90 needs MAPL)"}
    g = res.get("cfg0_irrad_1000_clear", {}).get("sample")
    if g:
        o = clib.irrad(synth.chou_lw_inputs(clr, aerosol=False), "f32")
        out["cfg0_irrad_1000_clear"] = {"columns": ns, "max_abs_Wm2": max(float(np.abs(f64(g[k]) - f64(o[k])).max()) for k in g),
                                        "tolerance_Wm2": 2e-2, "oracle": "oracle/liboracle.so (f32), parity unpinned (irrad.F90 needs MAPL)"}
    g = res.get("cfg4_c720_share_137l_rrtmg_standin", {}).get("sample")
    if g:
        deep = synth.make_columns(ns, CFG4_NLAY, start=CFG4_START, cloudy_frac=0.6, aerosol=True)
        clib.set_inhomogeneity(1, "f32")
        o = clib.rrtmg_lw(deep, "f32")
        q = clib.rrtmg_sw(deep, prec="f32", iaer=10, normFlx=1)
        clib.set_inhomogeneity(0, "f32")
        same_l = (g["clearCounts"] == o["clearCounts"]).all(axis=0); same_s = (g["clearCounts_sw"] == q["clearCounts"]).all(axis=0)
        e_lw = max(float(np.abs(f64(g[k]) - f64(o[k]))[:, (same_l if not k.endswith("c") else np.ones(ns, bool))].max()) for k in PARITY_LW)
        rel = np.zeros(ns)
        for k in PARITY_SW:
            dk = np.abs(f64(g[k]) - f64(q[k])).max(axis=0)
            rel = np.maximum(rel, dk if k.endswith("c") else np.where(same_s, dk, 0.0))
        out["cfg4_c720_share_137l_rrtmg_standin"] = {"columns": ns, "lw_max_abs_Wm2": e_lw, "sw_max_rel_toa": float(rel.max()),
                                                     "tolerance": {"lw_max_abs_Wm2": 4e-3, "sw_max_rel_toa": 5e-3},
                                                     "columns_with_other_clearCounts": int((~same_l).sum() + (~same_s).sum()),
                                                     "oracle": "oracle/liboracle.so (f32) at 137 layers: LW pinned to the reference, SW two-stream parity unpinned"}
    g = res.get("cfg2_mcica_200", {}).get("sample")
    if g:
        clib.set_inhomogeneity(1, "f32")
        m, ci, cw = clib.mcica(cld["zm"], cld["alat"], int(cld["dyofyr"]), cld["play"], cld["cldf"], cld["ciwp"], cld["clwp"], CFG_NSUB, prec="f32")
        clib.set_inhomogeneity(0, "f32")
        flips = int(((g["cldy_stoch"] != 0) != (m != 0)).sum())
        agree = (g["cldy_stoch"] != 0) == (m != 0)
        wp = max(float(np.abs(f64(g["ciwp_stoch"]) - f64(ci))[agree].max()), float(np.abs(f64(g["clwp_stoch"]) - f64(cw))[agree].max()))
        out["cfg2_mcica_200"] = {"columns": ns, "mask_flips": flips, "cells_with_a_decision": int((cld["cldf"] > 0).sum()) * CFG_NSUB,
                                 "max_abs_water_path_g_m2": wp, "oracle": "oracle/liboracle.so (f32), pinned bit for bit to the reference's generator"}
    return out


def configs_only(a, local_rank):
    """`bench.py --configs-only [--configs-legs ...]`: the legs alone (what the rocprofv3 child runs of the default line execute; also the command behind
    profiles/r04_configs_kernel_stats.csv)"""
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    only = [x for x in a.configs_legs.split(",") if x] or None
    res = configs_gpu(dev, local_rank, CFG_START, CFG_NCOL, a.warmup, a.steps, a.inputs_cache or None, want_parity=False, only=only)
    print(json.dumps({"configs": configs_assemble(res, None, None, None, CFG_NCOL, a.steps, a.warmup)}))


def configs_assemble(res, parity, cpu, live, ncol_default, steps_default, warmup_default, live2=None):
    """the `configs` object of the bench line"""
    out = {}
    for name in CFG_NAMES:
        r = res.get(name)
        if r is None:
            continue
        kern = r["kernel"]
        ncol, steps, warmup = r.get("columns", ncol_default), r.get("steps", steps_default), r.get("warmup", warmup_default)
        abytes = cfg_algorithmic_bytes(name)
        lps = max(r["kernel_launches_per_step"], 1e-9)
        launch_s = r["kernel_ms_per_step"] / lps * 1e-3
        achieved = abytes * (ncol / lps) / launch_s / 1e9 if launch_s > 0 else 0.0
        t = ((live2 if name in (CFG_NAMES[0], CFG_NAMES[6]) else live) or {}).get(kern)
        e = {"value": r["value"], "unit": "columns/s", "ms_per_step": r["ms_per_step"], "steps": steps, "warmup": warmup, "columns": ncol,
             "layers": r.get("layers", CFG_NLAY), "dtype": "u32 (KISS) + f32" if name == "cfg2_mcica_200" else "f32",
             "roofline": {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                          "traffic": None if t is None else t["traffic_bytes"] / lps,
                          "traffic_source": None if t is None else "measured in this run: rocprofv3 --pmc child runs of `bench.py --configs-only --configs-legs ...` (3 steps per leg, same columns)",
                          "algorithmic_bytes_per_column": abytes, "avg_launch_ms": r["kernel_ms_per_step"] / lps,
                          "launches_per_step": r["kernel_launches_per_step"], "columns_per_launch": ncol / lps},
             "kernels_ms_per_step": r["kernels_ms_per_step"], "cpu_baseline": (cpu or {}).get(name), "parity": (parity or {}).get(name)}
        if t is not None and "valu_util" in t:
            e["roofline_compute"] = {"kernel": kern, "valu_util": t["valu_util"], "wait_frac": t.get("wait_frac"),
                                     "resident_waves_per_simd": t.get("resident_waves_per_simd"),
                                     "lane_ops_per_column": round(t["valu_insts"] * 64.0 / ncol) if "valu_insts" in t else None}
        out[name] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=97_200, help="columns per GPU (default: C360 tile / 8)")
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--real", type=int, default=4, choices=[4, 8], help="arithmetic type: 4 = the reference's default real")
    ap.add_argument("--scheme", default="lwsw", choices=["lwsw", "lw", "sw", "chou", "irrad", "sorad", "gridcomp", "heartbeat", "mcica"])
    ap.add_argument("--nsubcol", type=int, default=200, help="mcica: sub-columns per column (BASELINE configs[2]: 200)")
    ap.add_argument("--cloudy", type=float, default=0.6, help="fraction of cloudy columns (0 = clear-sky)")
    ap.add_argument("--no-aerosol", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="lwsw: RRTMG_LW and RRTMG_SW on ONE stream (default: two HIP streams - the two solvers are independent)")
    ap.add_argument("--na-pass", action="store_true", help="gridcomp: also request the no-aerosol flavour of the SW fluxes (FSWNA ...)")
    ap.add_argument("--lit", type=float, default=1.0, help="lwsw / sw: fraction of the columns RRTMG_SW runs on (packed daytime columns; default: all)")
    ap.add_argument("--rats", type=int, default=0, help="gridcomp: RATS diagnostics for the first N gases of gridcomp.RAT_GAS (0-8)")
    ap.add_argument("--coherent", type=int, default=1, help="repeat every K-th profile K times (gather-divergence sensitivity)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-f64", action="store_true", help="lwsw / lw / sw at --real 4: skip the real_kind 8 leg (3 + 5 steps after the timed region)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect the dominant kernels' PMC counters in child rocprofv3 runs before the timed run (roofline.traffic, roofline_compute "
                         "then quote the figures committed under profiles/)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity figures (256 columns of the timed batch against the oracle) and the f64 leg")
    ap.add_argument("--host-api", action="store_true",
                    help="lwsw / lw / sw: also time the drop-in host-pointer entry points (geosrad_rrtmg_lw / _sw: pinned staging, chunk-pipelined "
                         "H2D / kernels / D2H) on the same columns and report the PCIe-inclusive rate next to the device-resident one")
    ap.add_argument("--ranks-per-gpu", default="",
                    help="K1,K2,...: for each K, K child processes (Fortran drop-in callers, host arrays) share the GPU on 1/K of the batch each; "
                         "aggregate columns/s per K (what the MPI ranks of a node get)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default line (lwsw, N = 1): skip the `configs` object - BASELINE configs[1] / [2] (RRTMG_LW clear-sky; RRTMG_SW + sorad + "
                         "irrad + the 200-sub-column generator; 100 000 columns each) timed after the headline's legs")
    ap.add_argument("--configs-only", action="store_true", help="run only the configuration legs (the PMC child of the default line)")
    ap.add_argument("--configs-legs", default="", help="with --configs-only: comma list of legs (default: all)")
    ap.add_argument("--inputs-cache", default="", help="internal: directory with the parent run's generated inputs (np.save files)")
    ap.add_argument("--control-path-only", action="store_true",
                    help="no GPU work: launcher, rank -> shard, barrier, MAX over ranks and rank 0's JSON line only (CPU rehearsal / tests)")
    a = ap.parse_args()
    import warnings
    warnings.filterwarnings("ignore", message="The given NumPy array is not writable")      # inputs mapped read-only from the cache
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:                   # the driver may start us as a plain `python bench.py --gpus N`: become the launcher
            sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s: one rank per GPU" % (a.gpus, os.environ["WORLD_SIZE"]))
    aerosol = not a.no_aerosol
    do_lw, do_sw = "lw" in a.scheme, "sw" in a.scheme
    do_irrad, do_sorad = a.scheme in ("chou", "irrad"), a.scheme in ("chou", "sorad")
    if (do_irrad or do_sorad) and a.ncol == 97_200:
        a.ncol = 100_000             # BASELINE configs[2]: 100 000 columns

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if a.control_path_only:
        return control_path_only(a, rank, world)
    if a.ranks_per_gpu:
        return bench_ranks_per_gpu(a)

    if a.configs_only:
        return configs_only(a, local_rank)
    # the `configs` object rides on the default line only (one GPU, fp32 headline, every column lit)
    want_cfg = (rank == 0 and world == 1 and a.gpus == 1 and a.scheme == "lwsw" and a.real == 4 and a.lit == 1.0 and a.coherent == 1
                and not a.no_configs and not a.host_api and not os.environ.get("GEOSRAD_BENCH_SORTED_CLOUDS"))
    want_pmc = (rank == 0 and a.gpus == 1 and world == 1 and not a.no_pmc and a.scheme in ("lwsw", "lw", "sw", "chou", "irrad", "sorad")
                and not under_profiler())
    live = live_cfg = live_cfg2 = cache = None
    inp_main = None
    if want_pmc and not a.inputs_cache:
        # the generated inputs are written once (np.save, memory-backed directory) and mapped by the rocprofv3 child runs and by this process,
        # instead of being generated again by each of them; removed at exit
        import atexit, shutil, tempfile
        from geosradiation_gridcomp_amd import synth as _synth
        base = None
        for cand_dir in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):       # ~3 GB (headline batch) + ~7 GB (the configs' batches)
            try:
                if os.path.isdir(cand_dir) and os.access(cand_dir, os.W_OK) and shutil.disk_usage(cand_dir).free > (16 << 30):
                    base = cand_dir
                    break
            except OSError:
                pass
        try:
            if base is None:
                raise OSError("no directory with 16 GB free for the inputs cache")
            cache = tempfile.mkdtemp(prefix="geosrad_bench_", dir=base)
            atexit.register(shutil.rmtree, cache, True)
            if a.scheme in ("lwsw", "lw", "sw") and a.coherent == 1 and not os.environ.get("GEOSRAD_BENCH_SORTED_CLOUDS"):
                inp_main = _synth.make_columns(a.ncol, a.nlay, start=shard_start(rank, a.ncol), cloudy_frac=a.cloudy, aerosol=not a.no_aerosol)
                _cache_put(cache, "main", inp_main)
            if want_cfg:
                clr, cld, ch, cs = cfg_inputs(CFG_NCOL, CFG_START)
                _cache_put(cache, "cfg", {**{"clr_" + k: v for k, v in clr.items()}, **{"cld_" + k: v for k, v in cld.items()},
                                          **{"ch_" + k: v for k, v in ch.items()}, **{"cs_" + k: v for k, v in cs.items()}})
                del clr, cld, ch, cs
                _cache_put(cache, "cfg4", _synth.make_columns(CFG4_BASE, CFG4_NLAY, start=CFG4_START, cloudy_frac=0.6, aerosol=True))
        except OSError as e:          # no room: every process generates its own inputs (slower, same numbers)
            print("bench.py: inputs cache not used (%s)" % e, file=sys.stderr)
            if cache:
                shutil.rmtree(cache, True)
            cache = None
    elif a.inputs_cache:
        cache = a.inputs_cache
    if want_pmc:
        live = live_counters(a, sys.argv[1:], cache)          # child processes, before this one touches the GPU
        if want_cfg:      # two sets of child runs: legs that share kernel names (irrad at two sizes; RRTMG at two sizes) must not share a run
            live_cfg = live_counters(a, sys.argv[1:], cache, configs=True, legs=CFG_NAMES[1:6])
            live_cfg2 = live_counters(a, sys.argv[1:], cache, configs=True, legs=(CFG_NAMES[0], CFG_NAMES[6]), mem_only=True)
    cpu = cpu_cfg = None
    if a.lit < 1.0:
        a.no_cpu = True           # the CPU leg times LW and SW on the same columns: not the --lit workload
    if want_cfg and not a.no_cpu:
        cpu_cfg = configs_cpu_baseline()
    if rank == 0 and a.gpus == 1 and not a.no_cpu:                 # before any GPU initialisation in this process (fork pool)
        if a.scheme == "heartbeat":
            cpu = heartbeat_cpu_baseline(a.nlay)
        elif a.scheme == "mcica":
            cpu = mcica_cpu_baseline(a.nlay, a.nsubcol, a.cloudy)
        elif a.scheme == "gridcomp":      # the solvers dominate: same baseline as the default line
            cpu = cpu_baseline(a.nlay, "lwsw", a.cloudy, aerosol)
        else:
            cpu = cpu_baseline(a.nlay, a.scheme, a.cloudy, aerosol)

    import torch
    import torch.distributed as dist
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context

    # GEOSRAD_BENCH_REHEARSAL=1: rehearse the N > 1 control path on fewer GPUs than ranks (gloo instead of RCCL, which refuses two
    # ranks on one device; ranks wrap around the visible devices).  Never set by the driver.
    rehearsal = os.environ.get("GEOSRAD_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks but %d visible GPUs (one rank per GPU)" % (world, torch.cuda.device_count()))
    if world > 1:
        # the path has no data exchange: the ranks meet only in the timing barrier and the MAX of the elapsed time, a
        # host-side double - gloo carries that; no RCCL communicator is built for it
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if a.scheme == "mcica":
        if a.ncol == 97_200:
            a.ncol = 100_000          # BASELINE configs[2]
        bench_mcica(a, rank, world, dev, local_rank, cpu)
        if world > 1:
            dist.destroy_process_group()
        return
    if a.scheme in ("gridcomp", "heartbeat"):
        bench_gridcomp(a, rank, world, dev, local_rank, aerosol, cpu)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- inputs resident in HBM -------------------------------------------------------------------------------
    ncol, nlay = a.ncol, a.nlay
    inp = inp_main if inp_main is not None else _cache_get(cache, "main")
    if inp is None or inp["play"].shape != (nlay, ncol):
        inp = synth.make_columns(ncol, nlay, start=shard_start(rank, ncol), cloudy_frac=a.cloudy, aerosol=aerosol)
    if os.environ.get("GEOSRAD_BENCH_SORTED_CLOUDS"):
        # experiment (profiles/): the batch's cloud-free columns first - the library's clear | cloudy partition is then the identity and every
        # array in the caller's column order is read fully coalesced
        order = np.argsort((inp["cldf"] > 0).any(axis=0), kind="stable")
        inp = {k: (np.ascontiguousarray(v[..., order]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v) for k, v in inp.items()}
    if a.coherent > 1:
        # sensitivity knob, not a headline configuration: runs of `coherent` identical profiles (neighbouring lanes then gather the
        # same k-distribution rows, as spatially smooth model fields largely do; the default columns are independent draws)
        src = (np.arange(ncol) // a.coherent) * a.coherent
        inp = {k: (np.ascontiguousarray(v[..., src]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v)
               for k, v in inp.items()}
    tdt = torch.float32 if a.real == 4 else torch.float64
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat"] + LW_IN2D + (["tauaer"] if aerosol else [])
    if do_sw:
        names += SW_IN + (SW_AER if aerosol else [])
    d = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).to(dev, dtype=tdt) for k in names}
    for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs", "swuflx", "swdflx", "swuflxc", "swdflxc"):
        d[k] = torch.zeros((nlay + 1, ncol), device=dev, dtype=tdt)
    for k in SW_OUT1:
        d[k] = torch.zeros(ncol, device=dev, dtype=tdt)
    d["fswband"] = torch.zeros((14, ncol), device=dev, dtype=tdt)
    d["clearCounts"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
    d["clearCounts_sw"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
    ch = cs = None
    if do_irrad:
        ch = synth.chou_lw_inputs(inp, aerosol=aerosol)
        for k in ("ple", "ta", "wa", "oa", "tb", "n2o", "ch4", "cfc11", "cfc12", "cfc22", "cwc", "fcld", "reff", "fs", "tg", "eg", "tv", "ev", "rv",
                  "taua", "ssaa", "asya"):
            d["ch_" + k] = torch.from_numpy(np.ascontiguousarray(ch[k])).to(dev, dtype=tdt)
        for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts"):
            d["ch_" + k] = torch.zeros((nlay + 1, ncol), device=dev, dtype=tdt)
        d["ch_sfcem"] = torch.zeros(ncol, device=dev, dtype=tdt)
        d["ch_taudiag"] = torch.zeros((10, nlay, ncol), device=dev, dtype=tdt)
        aer0 = {k: d["ch_" + k].clone() for k in ("taua", "ssaa", "asya")}
    if do_sorad:
        cs = synth.chou_sw_inputs(inp, aerosol=aerosol)
        for k in ("cosz", "pl", "ta", "wa", "oa", "cwc", "fcld", "reff", "taua", "ssaa", "asya", "rsuvbm", "rsuvdf", "rsirbm", "rsirdf"):
            d["so_" + k] = torch.from_numpy(np.ascontiguousarray(cs[k])).to(dev, dtype=tdt)
        for k in ("flx", "flc", "flxu", "flcu"):
            d["so_" + k] = torch.zeros((nlay + 1, ncol), device=dev, dtype=tdt)
        for k in ("fdiruv", "fdifuv", "fdirpar", "fdifpar", "fdirir", "fdifir"):
            d["so_" + k] = torch.zeros(ncol, device=dev, dtype=tdt)
        d["so_flx_sfc_band"] = torch.zeros((8, ncol), device=dev, dtype=tdt)
    ptr = {k: v.data_ptr() for k, v in d.items()}
    # --lit f < 1: day / night.  A pseudo-random fraction f of the columns is lit; like SORADCORE after `daytime = ZTH > 0.` + PackIt
    # (SOL:3686, :7753-7773; SURVEY 8(d) cfg 4 has about half of a tile lit) RRTMG_SW runs on the packed daytime columns only, and every
    # step builds the lit index, packs the SW inputs, and unpacks the SW outputs into the full-size fields (geosrad_lit_*_dev)
    ncol_sw = ncol
    ptr_sw = ptr
    lit = None
    if do_sw and a.lit < 1.0:
        h = (np.arange(shard_start(rank, ncol), shard_start(rank, ncol) + ncol, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
        day = (h.astype(np.float64) / float(1 << 24)) < a.lit
        zth = np.where(day, inp["coszen"], -0.5).astype(np.float32 if a.real == 4 else np.float64)
        ncol_sw = int(day.sum())
        sw_in = ["play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel", "zm", "alat"] \
            + SW_IN + (SW_AER if aerosol else [])
        sw_out = ["swuflx", "swdflx", "swuflxc", "swdflxc", "fswband"] + SW_OUT1
        dsw = {k: torch.zeros(d[k].shape[:-1] + (ncol_sw,), device=dev, dtype=tdt) for k in sw_in + sw_out}
        dsw["clearCounts_sw"] = torch.zeros((4, ncol_sw), device=dev, dtype=torch.int32)
        ptr_sw = {k: v.data_ptr() for k, v in dsw.items()}
        lit = {"zth": torch.from_numpy(zth).to(dev), "idx": torch.zeros(ncol, device=dev, dtype=torch.int32),
               "pos": torch.zeros(ncol, device=dev, dtype=torch.int32), "n": torch.zeros(1, device=dev, dtype=torch.int32),
               "in": [(k, int(np.prod(d[k].shape[:-1])) if d[k].dim() > 1 else 1) for k in sw_in],
               "out": [(k, int(np.prod(d[k].shape[:-1])) if d[k].dim() > 1 else 1) for k in sw_out]}
    ptr_ch = {k[3:]: v for k, v in ptr.items() if k.startswith("ch_")}
    ptr_so = {k[3:]: v for k, v in ptr.items() if k.startswith("so_")}

    ctx = Context(a.real, device=local_rank)
    ctx.set_inhomogeneity(1 if a.cloudy > 0 else 0)          # GEOS default RAD_CONDENSATE_INHOMOGENEITY=1
    stream = torch.cuda.current_stream().cuda_stream
    doy, lm, mh = int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])

    # RRTMG_LW and RRTMG_SW are independent (two sibling GridComps in GEOS): enqueued on two HIP streams their kernels share the
    # GPU - the latency-bound LW band kernel and the HBM-bound SW one complement each other, and no launch has an idle tail
    # the latency-bound solver (RRTMG_LW: ~1.5-1.9 wavefronts per SIMD resident, two thirds of their cycles waiting) goes on a
    # high-priority stream: its blocks are placed first and RRTMG_SW's fill what is left (19.06 -> 18.90 ms; GEOSRAD_BENCH_PRIO=none | sw | lw)
    prio = os.environ.get("GEOSRAD_BENCH_PRIO", "lw")
    side = torch.cuda.Stream(priority=-1 if prio == "sw" else 0) if (not a.no_overlap and do_lw and do_sw) else None
    sw_stream = side.cuda_stream if side is not None else stream
    lw_side = torch.cuda.Stream(priority=-1) if (side is not None and prio == "lw") else None
    lw_stream = lw_side.cuda_stream if lw_side is not None else stream

    def step():
        if do_lw:
            ctx.rrtmg_lw_dev(lw_stream, ncol, nlay, True, ptr, 3, 1, doy, lm, mh)
        if do_sw:      # GEOS call: isolvar 0 scaled to scon, normalised fluxes (SOL:6230-6300)
            if lit is not None:
                ctx.lit_index_dev(sw_stream, ncol, lit["zth"].data_ptr(), lit["idx"].data_ptr(), lit["pos"].data_ptr(), lit["n"].data_ptr(),
                                  want_count=False)
                for k, nlev in lit["in"]:
                    ctx.lit_pack_dev(sw_stream, ncol_sw, ncol, nlev, lit["idx"].data_ptr(), lit["n"].data_ptr(), ptr[k], ptr_sw[k])
            ctx.rrtmg_sw_dev(sw_stream, ncol_sw, nlay, 1361.0, 1.0, 0, ptr_sw, 3, 1, doy, 10 if aerosol else 0, lm, mh, normFlx=1)
            if lit is not None:
                for k, nlev in lit["out"]:
                    ctx.lit_unpack_dev(sw_stream, ncol_sw, ncol, nlev, lit["pos"].data_ptr(), ptr_sw[k], ptr[k], default=0.0)
        if do_irrad:
            for k in aer0:                    # taua / ssaa / asya are in-out (rescaled in place): restore the inputs
                d["ch_" + k].copy_(aer0[k])
            ctx.irrad_dev(stream, ncol, nlay, ptr_ch, ch["co2"], True, ch["ict"], ch["icb"], ch["ns"], ch["na"], ch["nb"])
        if do_sorad:
            ctx.sorad_dev(stream, ncol, nlay, 8, ptr_so, cs["co2"], cs["ict"], cs["icb"], cs["hk_uv"], cs["hk_ir"])

    for _ in range(a.warmup):
        step()
    ctx.check(stream)                                           # input checks of the warm-up (also synchronises)
    if side is not None:
        ctx.check(sw_stream)
    if lw_side is not None:
        ctx.check(lw_stream)
    ctx.profile(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    own = None
    if world > 1:                       # this rank's own steps done on its own GPU, before it waits for the others in the barrier
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = all_over_ranks((own if own is not None else elapsed) / a.steps * 1e3, world)
    elapsed = max_over_ranks(elapsed, world, torch.device("cpu"))
    ctx.check(stream)
    prof = ctx.profile_read()
    # the dominant kernel's OWN launch duration: with the two solvers on two streams the HIP events around a kernel also cover the time
    # it shares the GPU with the other solver's kernels, so the roofline figure comes from two further steps enqueued on one stream
    prof1 = prof
    if side is not None:
        sw_stream = lw_stream = stream
        ctx.profile(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        prof1 = ctx.profile_read()

    host_api = None
    if a.host_api and rank == 0 and (do_lw or do_sw):
        # the reference interface: host arrays in, host arrays out (what the Fortran drop-in of INTEGRATION.md 1-3 calls); PCIe included
        hsteps = max(2, min(a.steps, 5))
        ht = []
        hlw = hsw = None
        for it in range(hsteps + 1):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if do_lw:       # output arrays are the caller's and persist between calls, as a Fortran caller's do
                hlw = ctx.rrtmg_lw_columns(inp if aerosol else {k: v for k, v in inp.items() if k != "tauaer"}, out=hlw)
            if do_sw:
                hsw = ctx.rrtmg_sw_columns(inp, iaer=10 if aerosol else 0, normFlx=1, out=hsw)
            ht.append(time.perf_counter() - t1)
        hbest = sorted(ht[1:])[len(ht[1:]) // 2]           # median of the steps after the first (which allocates the staging slots)
        hbytes = (algorithmic_bytes_lw(nlay, a.real, aerosol) if do_lw else 0) + (algorithmic_bytes_sw(nlay, a.real, aerosol) if do_sw else 0)
        host_api = {"value": ncol / hbest, "unit": "columns/s", "ms_per_step": hbest * 1e3, "steps": hsteps,
                    "pcie_bytes_per_column": hbytes, "pcie_GB_per_s": hbytes * ncol / hbest / 1e9,
                    "note": "geosrad_rrtmg_lw + geosrad_rrtmg_sw on host arrays, one after the other (numpy -> ctypes, pageable caller memory): "
                            "copy threads -> pinned staging -> H2D | kernels | D2H on three streams, chunks of 16 384 columns"}

    # ---- the tolerance-qualifying precision and the in-run parity figures (N = 1, RRTMG schemes, every column lit) ---------------------
    f64_leg = parity = None
    want_parity = (rank == 0 and world == 1 and (do_lw or do_sw) and lit is None and a.coherent == 1 and not a.no_parity
                   and not os.environ.get("GEOSRAD_BENCH_SORTED_CLOUDS"))         # the parity sample is taken in the generator's column order
    if want_parity:
        ns = min(PARITY_COLS, ncol)
        keys = (PARITY_LW if do_lw else ()) + (PARITY_SW if do_sw else ()) + (("clearCounts",) if do_lw else ()) + (("clearCounts_sw",) if do_sw else ())
        got4 = got8 = None
        inp_s = synth.make_columns(ns, nlay, start=shard_start(rank, ncol), cloudy_frac=a.cloudy, aerosol=aerosol)
        got = {k: d[k][..., :ns].cpu().numpy() for k in keys}
        masks4 = None
        if a.real == 4:
            got4 = got
            if a.cloudy > 0:       # the generator's decisions themselves (stand-alone API, LW and SW seedings) for the flip rate
                masks4 = [(nsub, so, ctx.generate_stochastic_clouds(ns, nsub, nlay, inp_s["zm"], inp_s["alat"], doy, inp_s["play"], inp_s["cldf"],
                                                                    inp_s["ciwp"], inp_s["clwp"], 1e-20, seed_order=so)[0])
                          for nsub, so in ((140, (1, 2, 3, 4)), (112, (4, 3, 2, 1))) if (do_lw if nsub == 140 else do_sw)]
        else:
            got8 = got
        if a.real == 4 and not a.no_f64:
            ctx.close()                     # the fp64 workspace of 97 200 columns is ~100 GB: release the fp32 one first
            del d
            torch.cuda.empty_cache()
            ms8, got8 = lwsw_f64_leg(inp, dev, local_rank, aerosol, do_lw, do_sw, a.cloudy)
            f64_leg = {"value": ncol / (ms8 * 1e-3), "unit": "columns/s", "ms_per_step": ms8, "steps": 5, "warmup": 3, "dtype": "f64",
                       "note": "same workload and columns, real_kind 8 (the instantiation that meets BASELINE.json's 1e-6 W m-2), two HIP "
                               "streams, timed after the f32 region"}
        parity = lwsw_parity(inp_s, got4, got8, masks4, aerosol, do_lw, do_sw, a.cloudy)
        if f64_leg is not None and "f64" in parity:
            f64_leg["max_abs_err_vs_oracle_Wm2"] = max(parity["f64"].get("lw_max_abs_Wm2", 0.0), parity["f64"].get("sw_max_abs_Wm2", 0.0))

    cfg_out = None
    if want_cfg:
        if f64_leg is None:               # the headline's context and arrays are still alive: release the device memory first
            ctx.close()
            d = ptr = None
            torch.cuda.empty_cache()
        cres = configs_gpu(dev, local_rank, CFG_START, CFG_NCOL, CFG_WARMUP, CFG_STEPS, cache, want_parity=not a.no_parity)
        cpar = configs_parity(cres, CFG_START) if not a.no_parity else None
        cfg_out = configs_assemble(cres, cpar, cpu_cfg, live_cfg, CFG_NCOL, CFG_STEPS, CFG_WARMUP, live2=live_cfg2)

    if rank == 0:
        total_cols = world * ncol * a.steps
        value = total_cols / elapsed
        # dominant kernel = largest total time; its algorithmic bytes are those of the solver it belongs to
        # (compulsory bytes at the solver API, every input read once + every output written once)
        cand = {k: v for k, v in prof1.items() if k in LW_BAND_KERNELS + ("k_chou_bands", "k_sorad_pass", "k_sorad_col") + SW_BAND_KERNELS and v[1] > 0}
        kname = max(cand, key=lambda k: cand[k][0])
        ms, n = prof1[kname]
        launches_per_step = n / (2 if prof1 is not prof else a.steps)
        if kname == "k_chou_bands":
            abytes = 19476 * a.real // 4                # Chou irrad 3 491 in + 1 378 out reals @72 layers
        elif kname in ("k_sorad_pass", "k_sorad_col"):
            abytes = 11904 * a.real // 4                # Chou sorad 2 670 in + 306 out reals
        else:
            abytes = (algorithmic_bytes_lw if kname.startswith("k_lw") else algorithmic_bytes_sw)(nlay, a.real, aerosol)
        per_launch_s = (ms / max(n, 1)) * 1e-3
        ncol_k = ncol_sw if kname in SW_BAND_KERNELS else ncol          # columns the dominant kernel's launches cover
        achieved = abytes * (ncol_k / launches_per_step) / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        # HBM bytes per launch of the dominant kernel: measured by this run's own rocprofv3 --pmc child runs (live_counters), else from the
        # passes committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc runs) - those only for the exact configuration they
        # were collected on
        traffic = traffic_source = compute = None
        default_paths = not (os.environ.get("GEOSRAD_LW_PATH") or os.environ.get("GEOSRAD_SORAD_PATH") or os.environ.get("GEOSRAD_LIB"))
        ckey = None
        if a.scheme == "lwsw" and default_paths and (ncol, ncol_sw, nlay, a.cloudy, aerosol, a.real) == (97_200, 97_200, 72, 0.6, True, 4):
            ckey = "lwsw_97200_72_0.6_aer_f32"
        elif a.scheme in ("chou", "irrad", "sorad") and default_paths and (ncol, nlay, a.cloudy, aerosol, a.real) == (100_000, 72, 0.6, True, 4):
            ckey = "chou_100000_72_0.6_aer_f32"
        t = None
        if live is not None and kname in live:
            t, traffic_source = dict(live[kname]), "measured in this run: rocprofv3 --pmc child runs of 3 one-stream steps of the same workload"
            t["lane_ops_per_column"] = round(t["valu_insts"] * 64.0 / ncol_k) if "valu_insts" in t else None
        elif ckey is not None:
            t, traffic_source = committed_counters(ckey, kname)
        if t is not None:
            traffic = t["traffic_bytes"] / launches_per_step
            if "valu_util" in t:
                # the fused path is ~170 FLOP per compulsory byte, so the vector-ALU side is the second roofline
                compute = {"kernel": kname, "valu_util": t["valu_util"], "lane_ops_per_column": t.get("lane_ops_per_column"),
                           "valu_insts_per_step": t.get("valu_insts"), "wait_frac": t.get("wait_frac"), "source": traffic_source,
                           "resident_waves_per_simd": t.get("resident_waves_per_simd"),
                           "note": "valu_util = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): share of the kernel's SIMD-cycles "
                                   "with a vector instruction in the pipe; lane_ops_per_column = SQ_INSTS_VALU x 64 lanes / columns; wait_frac = "
                                   "SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
        schemes = {"lwsw": "RRTMG_LW (140 g-points) + RRTMG_SW (112 g-points)", "lw": "RRTMG_LW (140 g-points)",
                   "sw": "RRTMG_SW (112 g-points)", "chou": "Chou-Suarez irrad (10 bands) + sorad (8 bands, 35 spectral passes)",
                   "irrad": "Chou-Suarez irrad (10 bands, trace gases on)", "sorad": "Chou-Suarez sorad (8 bands, 35 spectral passes)"}[a.scheme]
        if a.scheme == "lw" and a.cloudy == 0 and not aerosol:
            wl = "BASELINE configs[1]: %d columns/GPU, %d layers, RRTMG_LW 140 g-points clear-sky" % (ncol, nlay)
        else:
            wl = ("%sper-GPU share: %d columns/GPU, %d layers, %s, McICA clouds on %.0f %% of the columns (ih=1), aerosols %s, "
                  "%s" % ("BASELINE configs[3] (C360 tile / 8 GPUs) " if ncol == 97_200 and a.scheme == "lwsw" else "",
                                         ncol, nlay, schemes, 100 * a.cloudy, "on" if aerosol else "off",
                                         "every column lit" if ncol_sw == ncol else "RRTMG_SW on the %d lit columns (lit index + PackIt / UnPackIt on the device every step)" % ncol_sw))
        out = {
            "metric": "columns/sec (LW+SW, 72 layers)" if a.scheme == "lwsw" else "columns/sec", "value": value, "unit": "columns/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.real == 4 else "f64", "data": "synthetic",
            "config": {"workload": wl, "schemes": schemes, "columns_per_gpu": ncol, "layers": nlay,
                       "cloudy_fraction": a.cloudy, "aerosol": aerosol, "hip_streams": 2 if side is not None else 1,
                       "sharding": "independent column batches per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_column": abytes, "avg_launch_ms": ms / max(n, 1), "launches": n,
                         "columns_per_launch": ncol_k / launches_per_step,
                         "note": roofline_note(kname, side is not None)},
            "kernels_ms_per_step": {k: v[0] / a.steps for k, v in prof.items() if v[1] > 0},
            "cpu_baseline": cpu,
        }
        if world > 1:
            out["per_rank_ms"] = per_rank_ms             # each rank's own ms per step, taken before the closing barrier (value uses the MAX over ranks of the time between the barriers)
        if compute is not None:
            out["roofline_compute"] = compute
        if f64_leg is not None:
            out["f64"] = f64_leg
        if parity is not None:
            bad = [k for k, v in parity.items() if not v.get("within_tolerance", True)]
            if bad:
                print("bench.py: in-run parity OUTSIDE the tolerance for " + ", ".join(bad) + ": " + json.dumps({k: parity[k] for k in bad}), file=sys.stderr)
            out["parity_ok"] = not bad
            out["parity"] = parity.get("f32" if a.real == 4 else "f64")
            if a.real == 4 and "f64" in parity:
                out["f64"]["parity"] = parity["f64"]
        if host_api is not None:
            out["host_api"] = host_api
        if cfg_out is not None:
            out["configs"] = cfg_out
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
