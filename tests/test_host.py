"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re
import numpy as np
import pytest
from tests.conftest import ROOT
from geosradiation_gridcomp_amd import synth, tableblob, _lib


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "geosrad.h")).read()
    declared = set(re.findall(r"\b(geosrad_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    _lib.build()
    L = ctypes.CDLL(_lib.SO)
    for name in declared:
        assert hasattr(L, name), name


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _lib.build()
    from geosradiation_gridcomp_amd.api import Context, GeosradError
    with pytest.raises(GeosradError):
        Context(4)


def test_device_choice_follows_the_local_mpi_rank():
    """geosrad_pick_device: GEOSRAD_DEVICE wins; else the launcher's node-local rank modulo the device count (96 ranks on an 8-GPU node
    share the GPUs 12 to one); nothing set -> device 0.  Pure function of the environment: runs without a GPU, in a child process each
    (getenv is read by the C library)."""
    import subprocess
    import sys
    _lib.build()
    code = "import ctypes,sys; L=ctypes.CDLL(sys.argv[1]); print(L.geosrad_pick_device(int(sys.argv[2])))"
    base = {k: v for k, v in os.environ.items() if k not in ("GEOSRAD_DEVICE", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID",
                                                             "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "PMI_LOCAL_RANK")}

    def pick(ndev, **env):
        return int(subprocess.check_output([sys.executable, "-c", code, _lib.SO, str(ndev)], env=dict(base, **env)).split()[-1])
    assert pick(8) == 0
    assert pick(8, OMPI_COMM_WORLD_LOCAL_RANK="5") == 5
    assert pick(8, OMPI_COMM_WORLD_LOCAL_RANK="95") == 95 % 8
    assert pick(8, SLURM_LOCALID="11") == 3
    assert pick(4, MV2_COMM_WORLD_LOCAL_RANK="6") == 2
    assert pick(8, GEOSRAD_DEVICE="2", OMPI_COMM_WORLD_LOCAL_RANK="5") == 2          # the explicit choice wins
    assert pick(8, SLURM_LOCALID="x7") == 0 and pick(0, SLURM_LOCALID="3") == 0      # unparsable / no device: 0
    # an explicit id is never wrapped: out of range or unparsable -> -1, which geosrad_create turns into GEOSRAD_ENODEV
    assert pick(4, GEOSRAD_DEVICE="5") == -1 and pick(4, GEOSRAD_DEVICE="3x") == -1 and pick(4, GEOSRAD_DEVICE="-2") == -1
    assert pick(4, GEOSRAD_DEVICE="3", SLURM_LOCALID="9") == 3


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "geosradiation_gridcomp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".F90")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and "libref_" not in src, f


def test_blob_roundtrip(tmp_path):
    a = {"x": np.arange(12, dtype=np.float64).reshape(3, 4), "n": np.int32(7), "v": np.arange(5, dtype=np.int32)}
    p = tmp_path / "t.grtb"
    tableblob.write_blob(str(p), 8, a)
    rb, b = tableblob.read_blob(str(p))
    assert rb == 8
    np.testing.assert_array_equal(b["x"], a["x"]); assert int(b["n"]) == 7; np.testing.assert_array_equal(b["v"], a["v"])


def test_shipped_tables_are_consistent():
    rb4, t4 = tableblob.read_blob(os.path.join(_lib.DATA, "rrtmg_lw_r4.grtb"))
    rb8, t8 = tableblob.read_blob(os.path.join(_lib.DATA, "rrtmg_lw_r8.grtb"))
    assert (rb4, rb8) == (4, 8) and set(t4) == set(t8)
    assert t4["b03_absa"].shape == (585, 16) and t4["exp_tbl"].shape == (10001,)
    assert list(t4["ngs"]) == [10, 22, 38, 52, 68, 76, 88, 96, 108, 114, 122, 130, 134, 136, 138, 140]
    for k in ("b03_absa", "totplnk"):
        assert np.allclose(t4[k], t8[k], rtol=3e-7, atol=0)
    assert np.allclose(t4["exp_tbl"], t8["exp_tbl"], rtol=0, atol=2e-7)
    assert np.allclose(t4["tfn_tbl"], t8["tfn_tbl"], rtol=0, atol=5e-5)   # fp32 cancellation in 1/tau - e/(1-e)
    for nm in ("beta", "gamma"):
        _, x4 = tableblob.read_blob(os.path.join(_lib.DATA, f"xcw_{nm}_r4.grtb"))
        _, x8 = tableblob.read_blob(os.path.join(_lib.DATA, f"xcw_{nm}_r8.grtb"))
        assert x4["xcw"].shape == (1000, 140) and np.allclose(x4["xcw"], x8["xcw"], rtol=1e-7)


def test_synth_is_shardable_and_valid():
    a = synth.make_columns(64, 72, cloudy_frac=0.6, aerosol=True)
    b = synth.make_columns(16, 72, start=32, cloudy_frac=0.6, aerosol=True)
    for k, v in b.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == 16:
            np.testing.assert_array_equal(v, a[k][..., 32:48], err_msg=k)
    assert np.all(np.diff(a["plev"], axis=0) < 0) and a["plev"].min() > 0
    assert a["tlay"].min() > 160 and a["tlay"].max() < 340
    for k in ("h2ovmr", "o3vmr", "cldf", "ciwp", "clwp", "tauaer"):
        assert a[k].min() >= 0
    assert 2.5 <= a["rel"].min() and a["rel"].max() <= 60 and 5 <= a["rei"].min() and a["rei"].max() <= 140
    assert a["cldf"].max() == 1.0 and 0.3 < (a["cldf"].max(axis=0) > 0).mean() < 0.9


@pytest.mark.parametrize("kind", ["r4", "r8"])
@pytest.mark.parametrize("ih", [0, 1, 2])
def test_fortran_host_functions_match_reference(tmp_path, kind, ih):
    """The McICA host functions the unchanged GridComps import from the shim modules (zcw_lookup, correlation_length_cloud_fraction,
    correlation_length_condensate; GEOS_IrradGridComp.F90:1472-1474) against the reference's own Fortran, bit for bit.  No device."""
    import subprocess
    from oracle import reflib
    fdir = os.path.join(ROOT, "geosradiation_gridcomp_amd", "fortran")
    exe = os.path.join(fdir, "bin", f"hostfn_driver_{kind}")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(exe):
        pytest.skip("no Fortran compiler")
    if not reflib.available(kind):
        pytest.skip("oracle/_ref not built")
    _lib.build()
    subprocess.check_call(["make", "-s", "-C", fdir, f"bin/hostfn_driver_{kind}"], stderr=subprocess.DEVNULL)
    dt = np.float32 if kind == "r4" else np.float64
    rng = np.random.default_rng(5 + ih)
    n = 4096
    cdf = rng.uniform(0, 1, n).astype(dt); cdf[:4] = [0, 1, 0.5, 1e-7]
    sigma = rng.uniform(0.0, 4.0, n).astype(dt)         # beyond both table edges (40 sigma - 3 in [1, 139])
    alat = rng.uniform(-np.pi / 2, np.pi / 2, n).astype(dt)
    for doy in (17, 181, 182, 300):
        fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
        with open(fin, "wb") as f:
            np.array([ih, doy, n], dtype=np.int32).tofile(f)
            cdf.tofile(f); sigma.tofile(f); alat.tofile(f)
        env = dict(os.environ, GEOSRAD_DATA=_lib.DATA)
        subprocess.check_call([exe, str(fin), str(fout)], env=env)
        z, adl, rdl = np.fromfile(fout, dtype=dt).reshape(3, n)
        reflib.set_inhomogeneity(0, kind); reflib.set_inhomogeneity(ih, kind)
        zr = reflib.zcw_lookup(cdf, sigma, kind)
        ar, rr = reflib.corr_lengths(alat, doy, kind)
        reflib.set_inhomogeneity(0, kind)
        np.testing.assert_array_equal(z, zr)
        np.testing.assert_array_equal(adl, ar)
        np.testing.assert_array_equal(rdl, rr)
        assert (z == 1).all() if ih == 0 else z.std() > 0.1


def test_bench_does_not_spawn_profiler_children_under_a_profiler(monkeypatch):
    """bench.py collects the dominant kernels' PMC counters in child rocprofv3 runs BEFORE it touches the GPU; when it is itself started
    by rocprofv3 (whose tool library has initialised the GPU already) it must not start any child."""
    import bench
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCTRACER", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert bench.under_profiler()
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
    assert set(bench.PMC_GROUPS) >= {"k_sw_reform", "k_sw_bands", "k_lw_bands", "k_chou_bands", "k_sorad_pass", "k_mcica_sa"}


def test_bench_counter_passes_are_parsed_as_the_guide_prescribes(tmp_path, monkeypatch):
    """bench.live_counters: one rocprofv3 --pmc pass per counter group, per-kernel sums over the dispatches of 3 steps, HBM bytes =
    (FETCH_SIZE x 2 + WRITE_SIZE) KB (the gfx950 correction of MI355X_MICROARCH.md).  A stand-in `rocprofv3` that writes the csv a real
    pass would leave checks the arithmetic and the kernel grouping without a GPU."""
    import argparse
    import stat
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
cs = a[a.index("--pmc") + 1:a.index("-d")]
d = a[a.index("-d") + 1]
assert a[a.index("--") + 1].endswith("python") or "python" in a[a.index("--") + 1]
assert "--no-pmc" in a and "--no-overlap" in a
os.makedirs(os.path.join(d, "h", "1"), exist_ok=True)
val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "SQ_INSTS_VALU": 3.0e6, "SQ_ACTIVE_INST_VALU": 2.0e6, "GRBM_GUI_ACTIVE": 8.0e4, "SQ_WAVE_CYCLES": 5.0e6,
       "SQ_WAIT_ANY": 2.5e6}
kern = ["void geosrad::k_sw_reform<float, true>(geosrad::SwArgs<float>)", "void geosrad::k_sw_reform<float, false>(geosrad::SwArgs<float>)",
        "void geosrad::k_lw_bands<float, true, false>(geosrad::LwArgs<float>)", "geosrad::k_partition(int, unsigned char const*, int*, int*)"]
with open(os.path.join(d, "h", "1", "x_counter_collection.csv"), "w") as f:
    f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
    n = 0
    for step in range(3):
        for k in kern:
            for c in cs:
                n += 1
                f.write('%d,"%s",%s,%r\\n' % (n, k, c, val[c]))
''')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    a = argparse.Namespace(scheme="lwsw", ncol=97200, nlay=72, cloudy=0.6, real=4, lit=1.0, coherent=1)
    r = bench.live_counters(a, [])
    assert set(r) == {"k_sw_reform", "k_lw_bands"}                      # k_partition belongs to no group; the two k_sw_reform instantiations to one
    sw, lw = r["k_sw_reform"], r["k_lw_bands"]
    assert sw["traffic_bytes"] == pytest.approx(2 * (2 * 1000.0 + 500.0) * 1024.0) and lw["traffic_bytes"] == pytest.approx((2 * 1000.0 + 500.0) * 1024.0)
    assert sw["valu_insts"] == pytest.approx(2 * 3.0e6)
    assert lw["valu_util"] == pytest.approx(round(2.0e6 * 4 / (8.0e4 / 8 * 1024), 3)) and lw["wait_frac"] == 0.5
    assert lw["resident_waves_per_simd"] == pytest.approx(round(4 * 5.0e6 / (8.0e4 / 8 * 1024), 2))
    # the child of the `configs` object runs the five legs alone
    fake_cfg = fake.read_text().replace('assert "--no-pmc" in a and "--no-overlap" in a', 'assert "--configs-only" in a and "--inputs-cache" in a')
    fake.write_text(fake_cfg)
    assert set(bench.live_counters(a, [], cache="/nonexistent", configs=True)) == {"k_sw_reform", "k_lw_bands"}
    # a failing pass -> None (bench.py then quotes the committed figures and says so)
    fake.write_text("#!/bin/sh\\nexit 3\\n")
    assert bench.live_counters(a, []) is None


def test_bench_inputs_cache_and_config_bytes(tmp_path):
    """bench.py hands its generated inputs to its rocprofv3 child runs through np.save files (mapped read-only): arrays and scalars survive
    the round trip; the algorithmic bytes of the `configs` legs are SURVEY 8(d)'s figures."""
    import bench
    from geosradiation_gridcomp_amd import synth
    inp = synth.make_columns(5, 8, start=3, cloudy_frac=1.0, aerosol=True)
    bench._cache_put(str(tmp_path), "main", inp)
    got = bench._cache_get(str(tmp_path), "main")
    assert set(got) == set(inp)
    for k, v in inp.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1:
            np.testing.assert_array_equal(np.asarray(got[k]), v)
        else:
            assert got[k] == int(v)
    assert bench._cache_get(str(tmp_path), "absent") is None and bench._cache_get(None, "main") is None
    assert bench.cfg_algorithmic_bytes("cfg1_lw_clear_100k") == 7608 and bench.cfg_algorithmic_bytes("cfg2_sw_noaer_100k") == 5356
    assert bench.cfg_algorithmic_bytes("cfg2_sorad_100k") == 11904 and bench.cfg_algorithmic_bytes("cfg2_irrad_100k") == 19476
    assert bench.cfg_algorithmic_bytes("cfg2_mcica_200") == 174244
    assert bench.cfg_algorithmic_bytes("cfg4_c720_share_137l_rrtmg_standin") == bench.algorithmic_bytes_sw(137, 4, True)
    assert len(bench.CFG_NAMES) == 7 and bench.cfg_algorithmic_bytes("cfg0_irrad_1000_clear") == 19476
