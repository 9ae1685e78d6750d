"""CPU, world_size 2, gloo: the N > 1 path of bench.py shards columns with no data-path collective -- each rank owns
columns [rank*n, (rank+1)*n) -- and only a barrier + MAX all-reduce of the elapsed time cross ranks."""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geosradiation_gridcomp_amd import synth
    import bench
    shard = synth.make_columns(n, 72, start=bench.shard_start(rank, n), cloudy_frac=0.5)
    # a per-shard checksum is all that ever needs to leave the rank (outputs stay with their columns)
    chk = torch.tensor([float(np.float64(shard["tlay"]).sum()), float(np.float64(shard["cldf"]).sum())], dtype=torch.float64)
    allchk = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allchk, chk)
    t = bench.max_over_ranks(0.010 * (rank + 1), world, torch.device("cpu"))
    dist.barrier()
    q.put((rank, [c.tolist() for c in allchk], t))
    dist.destroy_process_group()


def test_two_ranks_shard_columns_without_exchange():
    world, n = 2, 48
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from geosradiation_gridcomp_amd import synth
    full = synth.make_columns(world * n, 72, cloudy_frac=0.5)
    for rank, allchk, t in res:
        assert abs(t - 0.020) < 1e-12                      # MAX over ranks
        for r in range(world):
            sl = slice(r * n, (r + 1) * n)
            assert allchk[r][0] == float(np.float64(full["tlay"][:, sl]).sum())
            assert allchk[r][1] == float(np.float64(full["cldf"][:, sl]).sum())


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run environment (as the driver may start it) must become two ranks by
    itself; --control-path-only keeps the run on the CPU: rendezvous, barrier, MAX over ranks, one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--control-path-only", "--ncol", "1000"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["shard_starts"] == [0, 1000]
    assert out["ms_per_step"] >= 20.0                    # the slower rank (2 x 10 ms) sets the time
    # every rank's own time, taken before it waits for the others: rank 0 slept 10 ms, rank 1 20 ms
    assert len(out["per_rank_ms"]) == 2 and 10.0 <= out["per_rank_ms"][0] < out["per_rank_ms"][1] <= out["ms_per_step"]
    # a rank count that contradicts --gpus is refused, not silently run on one GPU
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--control-path-only"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "one rank per GPU" in bad.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The N > 1 path of bench.py on a GPU, as the driver's multi-GPU run takes it (launcher -> two ranks -> barrier, MAX over ranks, one
    JSON line), rehearsed with both ranks on the box's one device (GEOSRAD_BENCH_REHEARSAL=1: gloo carries the barrier and the MAX - the
    path has no data exchange).  Started as a child process; 2 x 20 000 columns."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GEOSRAD_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ncol", "20000",
                        "--no-cpu", "--no-parity"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["columns_per_gpu"] == 20000
    assert out["value"] == pytest.approx(2 * 20000 / (out["ms_per_step"] * 1e-3), rel=1e-9)
    assert len(out["per_rank_ms"]) == 2 and max(out["per_rank_ms"]) <= out["ms_per_step"] * (1 + 1e-9) and min(out["per_rank_ms"]) > 0
    roof = out["roofline"]
    assert roof["kernel"] in ("k_sw_reform", "k_lw_bands") and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes_per_column"] * roof["columns_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert "configs" not in out and "cpu_baseline" in out


@pytest.mark.gpu
def test_bench_configs_legs_alone():
    """`bench.py --configs-only --configs-legs ...` (what the default line's rocprofv3 child runs execute): two cheap legs, one JSON line with
    a roofline per leg whose figures are consistent with each other."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--configs-only", "--configs-legs", "cfg0_irrad_1000_clear,cfg1_lw_clear_100k",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["configs"]
    assert set(out) == {"cfg0_irrad_1000_clear", "cfg1_lw_clear_100k"}
    for name, e in out.items():
        roof = e["roofline"]
        assert e["value"] == pytest.approx(e["columns"] / (e["ms_per_step"] * 1e-3), rel=1e-9) and e["steps"] == 2
        assert roof["kernel"] == {"cfg0_irrad_1000_clear": "k_chou_bands", "cfg1_lw_clear_100k": "k_lw_bands"}[name]
        assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes_per_column"] * roof["columns_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
        assert 0 < roof["frac"] < 1 and roof["avg_launch_ms"] <= e["ms_per_step"]
    assert out["cfg0_irrad_1000_clear"]["columns"] == 1000 and out["cfg1_lw_clear_100k"]["columns"] == 100_000
