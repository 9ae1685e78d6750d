#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic columns resident in HBM.  Workload at
N = 1 is BASELINE.json configs[1]: 100 000 clear-sky columns, 72 layers, RRTMG_LW (140 g-points).
Columns shard embarrassingly: every rank owns its own 100 000-column batch (weak scaling), there is no
data-path collective; the only collectives are the timing barrier and the max-over-ranks of the time.

One JSON line is printed by rank 0 (contract in the task statement), with
  roofline     : dominant kernel (k_lw_bands) timed live with HIP events on the launch stream
  cpu_baseline : the reference's own Fortran (oracle/_ref, kind "reference") -- or the plain-C oracle
                 (kind "port") when _ref is absent -- timed on the host cores on a bounded sample.
"""
import argparse
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


def shard_start(rank, ncol_per_gpu):
    """first global column of a rank's batch: contiguous blocks, rank-major (no overlap, no exchange)"""
    return rank * ncol_per_gpu


def max_over_ranks(seconds, world, device):
    """the only cross-rank reduction of the benchmark: MAX of the elapsed time"""
    if world <= 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _cpu_worker(args):
    kind, start, ncol, nlay = args
    from geosradiation_gridcomp_amd import synth
    inp = synth.make_columns(ncol, nlay, start=start)
    if kind == "reference":
        from oracle import reflib
        reflib.lib("r4")
        t = time.perf_counter()
        reflib.rrtmg_lw(inp, "r4", psize=4)        # GEOS default RRTMGLW_PARTITION_SIZE=4 (IRR:3185)
        return time.perf_counter() - t
    from oracle import clib
    clib.lib()
    t = time.perf_counter()
    clib.rrtmg_lw(inp, "f32")
    return time.perf_counter() - t


def cpu_baseline(nlay, per_core=8192):
    """Time the CPU reference on all host cores (one process per core: the reference keeps module state)."""
    import multiprocessing as mp
    from oracle import reflib
    kind = "reference" if reflib.available("r4") else "port"
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    ctx = mp.get_context("fork")
    jobs = [(kind, 10_000_000 + i * per_core, per_core, nlay) for i in range(cores)]
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        per = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(per)
    return {"value": cores * per_core / busy, "unit": "columns/s", "cores": cores, "kind": kind,
            "sample": f"{cores} processes x {per_core} clear-sky 72-layer columns of the bench workload, rrtmg_lw psize=4, "
                      f"slowest process {busy:.2f} s (pool wall {wall:.2f} s incl. input generation)",
            "single_core_columns_per_s": per_core / (sum(per) / len(per))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=100_000, help="columns per GPU")
    ap.add_argument("--nlay", type=int, default=72)
    ap.add_argument("--real", type=int, default=4, choices=[4, 8], help="arithmetic type: 4 = the reference's default real")
    ap.add_argument("--cloudy", type=float, default=0.0, help="fraction of cloudy columns (0 = configs[1] clear-sky)")
    ap.add_argument("--aerosol", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    cpu = None
    if rank == 0 and a.gpus == 1 and not a.no_cpu:
        cpu = cpu_baseline(a.nlay)              # before any GPU initialisation in this process (fork pool)

    import torch
    import torch.distributed as dist
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- inputs resident in HBM -------------------------------------------------------------------------------
    ncol, nlay = a.ncol, a.nlay
    inp = synth.make_columns(ncol, nlay, start=shard_start(rank, ncol), cloudy_frac=a.cloudy, aerosol=a.aerosol)
    tdt = torch.float32 if a.real == 4 else torch.float64
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat"] + LW_IN2D + (["tauaer"] if a.aerosol else [])
    d = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).to(dev, dtype=tdt) for k in names}
    for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs"):
        d[k] = torch.zeros((nlay + 1, ncol), device=dev, dtype=tdt)
    d["clearCounts"] = torch.zeros((4, ncol), device=dev, dtype=torch.int32)
    ptr = {k: v.data_ptr() for k, v in d.items()}

    ctx = Context(a.real, device=local_rank)
    ctx.set_inhomogeneity(1 if a.cloudy > 0 else 0)          # GEOS default RAD_CONDENSATE_INHOMOGENEITY=1
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.rrtmg_lw_dev(stream, ncol, nlay, True, ptr, 3, 1, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]))

    for _ in range(a.warmup):
        step()
    ctx.check(stream)                                           # input checks of the warm-up (also synchronises)
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
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, dev)
    ctx.check(stream)
    prof = ctx.profile_read()

    if rank == 0:
        total_cols = world * ncol * a.steps
        value = total_cols / elapsed
        ms, n = prof["k_lw_bands"]
        abytes = algorithmic_bytes_lw(nlay, a.real, a.aerosol)
        per_launch_s = (ms / max(n, 1)) * 1e-3
        achieved = abytes * ncol / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        out = {
            "metric": "columns/sec", "value": value, "unit": "columns/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.real == 4 else "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: %d columns/GPU, %d layers, RRTMG_LW 140 g-points clear-sky" % (ncol, nlay))
                       if a.cloudy == 0 and not a.aerosol else
                       ("%d columns/GPU, %d layers, RRTMG_LW + McICA (ih=1), cloudy fraction %.2f, aerosol %s" % (ncol, nlay, a.cloudy, a.aerosol)),
                       "schemes": "RRTMG_LW (SW / Chou paths: later rounds)", "columns_per_gpu": ncol, "layers": nlay,
                       "sharding": "independent column batches per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_lw_bands", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "algorithmic_bytes_per_column": abytes, "avg_launch_ms": ms / max(n, 1), "launches": n,
                         "note": "fused k-distribution + two-stream sweep is FP32/latency bound (~170 FLOP per algorithmic byte, "
                                 "SURVEY 8(d)); HBM fraction of the compulsory bytes is expected to be small"},
            "kernels_ms_per_step": {k: v[0] / a.steps for k, v in prof.items()},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
