"""GPU parity tests of RRTMG_SW (pytest -m gpu): the HIP path, called through the C ABI, against
  (a) the committed golden vectors of the reference's own setcoef_sw/taumol_sw (tests/golden/sw_stages_*.npz),
  (b) the plain-C oracle on seeded inputs -- its setcoef/taumol/cldprmc/McICA stages are pinned bit-exactly to the
      reference; its two-stream/adding/driver part is "parity unpinned" (the reference files need ESMF/MAPL, absent here),
  (c) size-independent properties at BASELINE's full size (100 000 columns).
Tolerances: real_kind 8: <= 1e-6 W m-2 (the north-star bar; measured <= 1e-8).  real_kind 4: relative to the column's
TOA incoming flux -- median 1e-6, 99.9 % of columns <= 2e-4, worst of 4000 columns 6e-4 (measured): the PIFM two-stream
has a removable singularity at mu0 = 1/k (SW/rrtmg_sw_spcvmc.F90:1300-1345, `zdenr`), near which fp32 loses all digits; the
r4 and r8 instantiations of the oracle itself differ by up to 3.5e-3 of the TOA flux on such columns.  Hence: every
column <= 5e-3, and at most 1 % of the columns above 5e-4."""
import numpy as np
import pytest
from tests.conftest import sub_columns
from tests.test_oracle_sw import load_sw_golden, SW_GOLDEN

pytestmark = pytest.mark.gpu

TOL_FLUX = {4: 5e-4, 8: 1e-6}      # real_kind 4: relative to the TOA incoming flux of the column


def flux_tol(rk, toa):
    """per-column absolute tolerance; `toa` = TOA downward flux of the reference (1 when normalised)"""
    toa = np.asarray(toa, dtype=np.float64)
    return np.full(toa.shape, TOL_FLUX[8]) if rk == 8 else np.maximum(TOL_FLUX[4] * toa, 1e-6)
SWFLUX = ("swuflx", "swdflx", "swuflxc", "swdflxc")
SFC = ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf", "fswband")
COT = ("cotdtp", "cotdhp", "cotdmp", "cotdlp", "cotntp", "cotnhp", "cotnmp", "cotnlp")


def _kind(rk):
    return "r4" if rk == 4 else "r8"


@pytest.mark.parametrize("name", SW_GOLDEN)
@pytest.mark.parametrize("rk", [8, 4])
def test_sw_taumol_matches_reference_golden(gpu_ctx, name, rk):
    ctx = gpu_ctx[rk]
    inp, g, _ = load_sw_golden(name)
    kind = _kind(rk)
    for isol in (0, -1, 2, 3):
        taug, taur, ssi = ctx.rrtmg_sw_taumol(inp, scon=float(g["scon"]), isolvar=isol, bndscl=g["bndscl"] if isol == 3 else None,
                                              indsolvar=g["indsolvar"] if isol == 2 else None)
        want_g, want_r = g[f"{kind}_taug"], g[f"{kind}_taur"]
        want_s = g[f"{kind}_ssi_isolvar{isol}".replace("-", "m")]
        if rk == 8:
            np.testing.assert_allclose(taug, want_g, rtol=1e-12, atol=1e-300)
            np.testing.assert_allclose(taur, want_r, rtol=1e-13)
            np.testing.assert_allclose(ssi, want_s, rtol=1e-13)
        else:
            # fp32: typical error 3e-7; a handful of cells (p/T weights extrapolating past the table edge, terms of
            # opposite sign) reach 3e-4 relative because the kernel sums the interpolation in a different order
            m = want_g != 0
            rel = np.abs(taug - want_g)[m] / np.abs(want_g[m])
            assert rel.max() <= 2e-3 and np.quantile(rel, 0.999) <= 1e-5, (rel.max(), np.quantile(rel, 0.999))
            assert (taug[~m] == 0).all()
            np.testing.assert_allclose(taur, want_r, rtol=2e-6)
            np.testing.assert_allclose(ssi, want_s, rtol=2e-6)


def test_sw_isolvar1_solar_source_matches_reference_golden(gpu_ctx):
    """isolvar = 1 (rrtmg_sw_rad.F90:906-930,994-1008,1060-1079): the spectral solar source per g-point the library forms from
    (scon, solcycfrac, indsolvar) against the REFERENCE's taumol_sw driven with the scalars the reference's NRLSSI2 routines give
    (tests/golden/nrlssi2_isolvar1.npz, make_golden.main_nrlssi2), and the host scalars themselves for every committed case"""
    import ast, os
    from tests.conftest import GOLDEN
    from geosradiation_gridcomp_amd import synth
    g = np.load(os.path.join(GOLDEN, "nrlssi2_isolvar1.npz"))
    inp = synth.make_columns(**ast.literal_eval(str(g["kw_json"])))
    for rk in (8, 4):
        kind = _kind(rk)
        _, _, ssi = gpu_ctx[rk].rrtmg_sw_taumol(inp, scon=float(g["case_scon"][0]), isolvar=1, indsolvar=g["case_indsolvar"][0],
                                                solcycfrac=float(g["case_solcycfrac"][0]))
        np.testing.assert_allclose(ssi, g[f"{kind}_ssi_case0"], rtol=1e-13 if rk == 8 else 2e-6)
        # ssi is linear in (svar_f, svar_s, svar_i): every other committed case through its ratio to an isolvar 0 source is not
        # available per g-point without the reference, so those cases are held through the oracle (pinned bit for bit to the
        # reference's routines in tests/test_oracle_sw.py) in test_sw_fluxes_match_oracle below


CASES = [dict(), dict(iaer=10), dict(normFlx=1, do_drfband=True), dict(isolvar=-1), dict(isolvar=2, indsolvar=(0.158, 80.0)),
         dict(isolvar=3, bndscl=np.linspace(0.9, 1.1, 14)), dict(iceflg=1), dict(iceflg=2), dict(iceflg=4),
         dict(scon=0.0, isolvar=2, indsolvar=(0.155, 60.0)), dict(adjes=1.0334, iaer=10, do_drfband=True),
         dict(isolvar=1, solcycfrac=0.30, indsolvar=(1.15, 0.9)), dict(isolvar=1, solcycfrac=0.0189, scon=0.0),
         dict(isolvar=1, solcycfrac=0.85, indsolvar=(1.0, 1.7), iaer=10)]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("rk", [8, 4])
def test_sw_fluxes_match_oracle(gpu_ctx, rk, case):
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[rk]
    kind = _kind(rk)
    kw = CASES[case]
    nlay = 137 if case == 5 else 72
    inp = synth.make_columns(96, nlay, start=2000 + 100 * case, aerosol=True, cloudy_frac=0.6)
    ih = (1, 0, 2)[case % 3]
    ctx.set_inhomogeneity(ih); clib.set_inhomogeneity(ih, kind)
    try:
        g = ctx.rrtmg_sw_columns(inp, **kw)
        o = clib.rrtmg_sw(inp, prec=kind, **kw)
    finally:
        ctx.set_inhomogeneity(0); clib.set_inhomogeneity(0, kind)
    assert o["rc"] == 0
    if rk == 8:
        np.testing.assert_array_equal(g["clearCounts"], o["clearCounts"])
        same = np.ones(inp["play"].shape[1], dtype=bool)
    else:   # fp32 exp() of the overlap correlations may flip a sub-column decision at the 1e-7 level: skip such columns
        same = (g["clearCounts"] == o["clearCounts"]).all(axis=0)
        assert same.mean() >= 0.95
    tol = flux_tol(rk, o["swdflx"][nlay])
    for k in SWFLUX + SFC + (("drband", "dfband") if kw.get("do_drfband") else ()):
        err = (np.abs(g[k].astype(np.float64) - o[k].astype(np.float64)) / tol)[..., same]
        worst = err.reshape(-1, err.shape[-1]).max(axis=0)            # per column
        if rk == 8:
            assert worst.max() <= 1.0, (k, worst.max())
        else:
            assert worst.max() <= 10.0 and (worst > 1.0).mean() <= 0.01, (k, worst.max(), (worst > 1.0).mean())
    cot = np.stack([g[k] for k in COT]).astype(np.float64)
    ref = o["cot"].astype(np.float64)
    assert (np.abs(cot - ref)[:, same] <= (1e-11 if rk == 8 else 2e-5) * np.maximum(np.abs(ref[:, same]), 1.0)).all()


def test_sw_chunking_and_determinism(gpu_ctx):
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    inp = synth.make_columns(300, 72, start=99, cloudy_frac=0.5, aerosol=True)
    ctx.set_inhomogeneity(1)
    a = ctx.rrtmg_sw_columns(inp, iaer=10, do_drfband=True)
    a2 = ctx.rrtmg_sw_columns(inp, iaer=10, do_drfband=True)
    ctx.set_chunk(128)            # 3 ragged batches: 128 + 128 + 44
    b = ctx.rrtmg_sw_columns(inp, iaer=10, do_drfband=True)
    ctx.set_chunk(131072)
    ctx.set_inhomogeneity(0)
    for k in SWFLUX + SFC + COT + ("drband", "dfband", "clearCounts"):
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)       # run-to-run bitwise (no float atomics)
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)        # batching is invisible


def test_sw_reference_error_stops_become_errors(gpu_ctx):
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import GeosradInputError
    ctx = gpu_ctx[4]
    inp = synth.make_columns(8, 72, start=5, cloudy_frac=1.0)
    bad = dict(inp); bad["tlay"] = inp["tlay"].copy(); bad["tlay"][3, 2] = -1.0
    with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
        ctx.rrtmg_sw_columns(bad)
    bad = dict(inp); bad["asdif"] = -inp["asdif"] - 0.1
    with pytest.raises(GeosradInputError, match="surface albedo"):
        ctx.rrtmg_sw_columns(bad)
    bad = dict(inp); bad["cloudLM"] = 5; bad["cloudMH"] = 5
    with pytest.raises(GeosradInputError, match="invalid pressure super-layers"):
        ctx.rrtmg_sw_columns(bad)
    with pytest.raises(GeosradInputError, match="invalid iceflag"):
        ctx.rrtmg_sw_columns(inp, iceflg=7)
    with pytest.raises(GeosradInputError, match="invalid liqflag"):
        ctx.rrtmg_sw_columns(inp, liqflg=0)
    with pytest.raises(GeosradInputError, match="isolvar == 1 requires solcycfrac"):
        ctx.rrtmg_sw_columns(inp, isolvar=1)
    with pytest.raises(GeosradInputError, match=r"solcycfr must be in \[0,1\]"):
        ctx.rrtmg_sw_columns(inp, isolvar=1, solcycfrac=1.25, indsolvar=(1.2, 0.8))
    with pytest.raises(GeosradInputError, match="invalid isolvar"):
        ctx.rrtmg_sw_columns(inp, isolvar=4)
    with pytest.raises(GeosradInputError, match="scon"):
        ctx.rrtmg_sw_columns(inp, scon=-1.0)
    o = ctx.rrtmg_sw_columns(inp)       # and the context still works afterwards
    assert np.isfinite(o["swuflx"]).all()


def test_host_entry_points_error_slots(gpu_ctx):
    """Error flags of the host-pointer entry points (chunk pipeline): a chunk that trips an input assertion is not scattered into the
    caller's arrays; a flag left behind by an UNCHECKED `_dev` call of the other solver is neither reported by nor lost to a host call;
    the context works afterwards."""
    import torch
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import GeosradInputError
    ctx = gpu_ctx[4]
    n, nlay = 40_000, 72                                    # three chunks of the host pipeline (16 384 columns each)
    inp = synth.make_columns(n, nlay, start=77_000)
    good = ctx.rrtmg_lw_columns(inp)
    bad = dict(inp); bad["tlay"] = inp["tlay"].copy(); bad["tlay"][5, 39_000] = -3.0          # in the last chunk
    out = {k: np.full_like(v, -777.0) if v.dtype.kind == "f" else v.copy() for k, v in good.items()}
    with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
        ctx.rrtmg_lw_columns(bad, out=out)
    assert (out["uflx"][:, 32_768:] == -777.0).all()        # the offending chunk was not delivered
    np.testing.assert_array_equal(out["uflx"][:, :16_384], good["uflx"][:, :16_384])      # the first chunk was, before the error was seen
    # an unchecked RRTMG_SW `_dev` error must not surface in (or be cleared by) an RRTMG_LW host call
    small = synth.make_columns(64, nlay, start=5)
    tdt = torch.float32
    names = ["play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel", "zm", "alat", "coszen",
             "asdir", "asdif", "aldir", "aldif"]
    d = {k: torch.from_numpy(np.ascontiguousarray(small[k])).to("cuda", dtype=tdt) for k in names}
    d["tlay"][3, 2] = -1.0
    for k in SWFLUX:
        d[k] = torch.zeros((nlay + 1, 64), device="cuda", dtype=tdt)
    for k in SFC + COT:
        d[k] = torch.zeros(64, device="cuda", dtype=tdt)
    d["fswband"] = torch.zeros((14, 64), device="cuda", dtype=tdt)
    d["clearCounts_sw"] = torch.zeros((4, 64), device="cuda", dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    ctx.rrtmg_sw_dev(st, 64, nlay, 1361.0, 1.0, 0, {k: v.data_ptr() for k, v in d.items()}, 3, 1, int(small["dyofyr"]), 0,
                     int(small["cloudLM"]), int(small["cloudMH"]))
    torch.cuda.synchronize()
    again = ctx.rrtmg_lw_columns(inp)                       # LW host call: clean, although the SW slot is up
    np.testing.assert_array_equal(again["uflx"], good["uflx"])
    with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
        ctx.check(st)                                       # the `_dev` caller's own check still finds it
    ctx.check(st)                                           # ... once


@pytest.mark.parametrize("ncol", [1, 257])
def test_sw_ragged_sizes(gpu_ctx, ncol):
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[8]
    inp = synth.make_columns(ncol, 72, start=10 * ncol, cloudy_frac=0.5)
    g = ctx.rrtmg_sw_columns(inp)
    o = clib.rrtmg_sw(sub_columns(inp, min(ncol, 16)), prec="r8")
    for k in SWFLUX:
        assert np.abs(g[k][..., :16] - o[k]).max() <= TOL_FLUX[8], k


def test_sw_full_size_properties(gpu_ctx):
    """BASELINE config 2 size (100 000 columns, 72 layers), 30 % of them cloudy: properties that need no oracle."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[4]
    n = 100_000
    inp = synth.make_columns(n, 72, cloudy_frac=0.3)
    ctx.set_inhomogeneity(1)
    o = ctx.rrtmg_sw_columns(inp, do_drfband=True)
    for k in SWFLUX + SFC:
        assert np.isfinite(o[k]).all(), k
    mu0 = np.maximum(inp["coszen"].astype(np.float64), 1e-10)
    np.testing.assert_allclose(o["swdflx"][72], 1361.0 * mu0, rtol=2e-5)
    clear = ~(inp["cldf"] > 0).any(axis=0)
    np.testing.assert_array_equal(o["swuflx"][:, clear], o["swuflxc"][:, clear])      # no cloud: clear == total, bitwise
    assert (o["clearCounts"][:, clear] == 112).all() and (o["clearCounts"][0, ~clear] < 112).all()
    for up, dn in (("swuflx", "swdflx"), ("swuflxc", "swdflxc")):
        net = o[dn].astype(np.float64) - o[up]
        viol = (np.diff(net, axis=0) < -1e-3 * o[dn][72]).any(axis=0)      # the atmosphere only absorbs ...
        assert viol.mean() <= 1e-3, viol.mean()                              # ... except near the fp32 singularity (header)
    sfc_dn = o["swdflx"][0].astype(np.float64)
    part = sum(o[k].astype(np.float64) for k in ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf"))
    np.testing.assert_allclose(part, sfc_dn, rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(o["fswband"].astype(np.float64).sum(axis=0), sfc_dn - o["swuflx"][0], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose((o["drband"].astype(np.float64) + o["dfband"]).sum(axis=0), sfc_dn, rtol=2e-5, atol=1e-3)
    # in double precision the property is exact: net flux never increases downwards (4096 of the columns)
    sub = sub_columns(inp, 4096)
    gpu_ctx[8].set_inhomogeneity(1)
    o8 = gpu_ctx[8].rrtmg_sw_columns(sub)
    gpu_ctx[8].set_inhomogeneity(0)
    for up, dn in (("swuflx", "swdflx"), ("swuflxc", "swdflxc")):
        assert (np.diff(o8[dn] - o8[up], axis=0) >= -1e-9 * o8[dn][72]).all()
    # column independence: any shard equals the same columns computed alone, bitwise
    sl = slice(54_321, 54_321 + 257)
    shard = synth.make_columns(257, 72, start=54_321, cloudy_frac=0.3)
    p = ctx.rrtmg_sw_columns(shard, do_drfband=True)
    for k in SWFLUX + SFC + ("clearCounts",):
        np.testing.assert_array_equal(p[k], o[k][..., sl], err_msg=k)
    # spot parity against the oracle on 32 of the 100 000 columns
    clib.set_inhomogeneity(1, "r4")
    r = clib.rrtmg_sw(sub_columns(shard, 32), prec="r4")
    clib.set_inhomogeneity(0, "r4"); ctx.set_inhomogeneity(0)
    same = (p["clearCounts"][:, :32] == r["clearCounts"]).all(axis=0)
    for k in ("swuflx", "swdflx"):
        assert (np.abs(p[k][:, :32].astype(np.float64) - r[k]) <= 10 * flux_tol(4, r["swdflx"][72]))[:, same].all()


@pytest.mark.parametrize("rk", [8, 4])
def test_fresh_context_with_sw_tables_only(rk):
    """A context that never saw rrtmg_lw_ini, set_inhomogeneity or initialize_cloud_subcol_gen (fortran/sw_driver.F90 with ih = 0
    is that case) must still run McICA on the default decorrelation parameters and a null xcw table: cloudy columns against the
    oracle, and the stand-alone generator on a context with no tables at all against the reference-pinned one."""
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context
    from oracle import clib
    kind = _kind(rk)
    inp = synth.make_columns(64, 72, start=7100, aerosol=True, cloudy_frac=1.0)
    clib.set_inhomogeneity(0, kind)
    ctx = Context(rk, tables=False)
    try:
        cl, ci, cw = ctx.generate_stochastic_clouds(64, 112, 72, inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"],
                                                    inp["ciwp"], inp["clwp"], 1e-20, seed_order=(4, 3, 2, 1))
        rl, ri, rw = clib.mcica(inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"], inp["ciwp"], inp["clwp"], 112,
                                seed_order=(4, 3, 2, 1), prec=kind)
        flips = int((cl != rl.astype(np.int32)).sum())
        assert flips == 0 if rk == 8 else flips <= 1e-5 * cl.size, flips
        assert cl.sum() > 0.01 * cl.size            # really cloudy: an all-zero alpha (random overlap) or a wild xcw would show here
        ctx.rrtmg_sw_ini()
        g = ctx.rrtmg_sw_columns(inp, iaer=10)
    finally:
        ctx.close()
    o = clib.rrtmg_sw(inp, prec=kind, iaer=10)
    same = (g["clearCounts"] == o["clearCounts"]).all(axis=0)
    assert same.all() if rk == 8 else same.mean() >= 0.95
    tol = flux_tol(rk, o["swdflx"][72])
    for k in SWFLUX:
        err = (np.abs(g[k].astype(np.float64) - o[k].astype(np.float64)) / tol)[..., same]
        assert err.max() <= (1.0 if rk == 8 else 10.0), (k, err.max())


@pytest.mark.parametrize("name", SW_GOLDEN)
@pytest.mark.parametrize("rk", [8, 4])
def test_sw_cldprmc_stage_matches_reference_golden(gpu_ctx, name, rk):
    """The cloud optics the GPU solver's own McICA + cldprmc_sw produce (debug tap geosrad_rrtmg_sw_cldprmc) against the
    REFERENCE's cldprmc_sw outputs on the reference's own sub-columns (tests/golden/sw_stages_*.npz), every ice parameterisation."""
    ctx = gpu_ctx[rk]
    inp, g, ih = load_sw_golden(name)
    kind = _kind(rk)
    sub = sub_columns(inp, 2)
    ctx.set_inhomogeneity(ih)
    try:
        for ice in (1, 2, 3, 4):
            got = ctx.rrtmg_sw_cldprmc(sub, iceflg=ice)
            want = [g[f"{kind}_ice{ice}_{nm}"] for nm in ("taucmc", "ssacmc", "asmcmc")]
            assert (want[0] > 0).sum() > 20
            if rk == 8:
                for a, b, nm in zip(got, want, ("taucmc", "ssacmc", "asmcmc")):
                    np.testing.assert_allclose(a, b, rtol=1e-12, atol=0, err_msg=f"iceflag {ice} {nm}")
            else:
                # fp32: a sub-column decision may fall the other way where exp() of an overlap correlation differs by an ulp
                same = (got[0] > 0) == (want[0] > 0)
                assert (~same).sum() <= 2e-4 * same.size, (~same).sum()
                for a, b, nm in zip(got, want, ("taucmc", "ssacmc", "asmcmc")):
                    rel = np.abs(a[same] - b[same]) / np.maximum(np.abs(b[same]), 1e-30)
                    assert (rel > 2e-5).mean() <= 1e-3, (ice, nm, rel.max())     # condensate-overlap flips change a few values
    finally:
        ctx.set_inhomogeneity(0)


def test_sw_fp32_is_as_accurate_as_the_reference_precision(gpu_ctx):
    """What the fp32 instantiation (hardware 1-ulp rcp / sqrt / exp in the two-stream) costs in accuracy, measured against the r8
    oracle as the truth and set next to the error of the r4 oracle (= the reference's arithmetic in the reference's production
    precision) on the same columns: the GPU's fp32 fluxes are as close to the truth as default-real CPU fluxes are."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    n, nlay = 1024, 72
    inp = synth.make_columns(n, nlay, start=700_000, aerosol=True, cloudy_frac=0.6)
    ctx = gpu_ctx[4]
    ctx.set_inhomogeneity(1)
    try:
        g4 = ctx.rrtmg_sw_columns(inp, iaer=10)
    finally:
        ctx.set_inhomogeneity(0)
    o = {}
    for kind in ("r4", "r8"):
        clib.set_inhomogeneity(1, kind)
        try:
            o[kind] = clib.rrtmg_sw(inp, prec=kind, iaer=10)
        finally:
            clib.set_inhomogeneity(0, kind)
    # the r4 and r8 builds seed McICA from different pressure bits, so their sub-columns differ by construction: the comparison is made on
    # the clear-sky fluxes of every column and on the total-sky fluxes of the cloud-free columns
    free = ~(inp["cldf"] > 0).any(axis=0)
    assert free.sum() >= 300
    toa = o["r8"]["swdflx"][nlay].astype(np.float64)

    def err(x):
        e = np.zeros(n)
        for k in ("swuflxc", "swdflxc"):
            e = np.maximum(e, np.abs(x[k].astype(np.float64) - o["r8"][k].astype(np.float64)).max(axis=0))
        for k in ("swuflx", "swdflx"):
            e = np.maximum(e, np.where(free, np.abs(x[k].astype(np.float64) - o["r8"][k].astype(np.float64)).max(axis=0), 0.0))
        return e / toa

    eg, eo = err(g4), err(o["r4"])
    q = lambda e: (np.median(e), np.percentile(e, 99), e.max())
    print("fp32 error vs r8 oracle, fraction of the TOA flux (median, 99 %%, max): GPU %.2e %.2e %.2e | r4 oracle %.2e %.2e %.2e" % (q(eg) + q(eo)))
    assert np.median(eg) <= 1.25 * np.median(eo) + 1e-7
    assert np.percentile(eg, 99) <= 1.25 * np.percentile(eo, 99) + 1e-6
    assert eg.max() <= max(1.5 * eo.max(), 1e-4)      # measured: GPU 2.1e-5 / 6.1e-5 / 4.3e-4, r4 oracle 2.2e-5 / 6.1e-5 / 3.4e-4


@pytest.mark.parametrize("rk", [8, 4])
def test_sw_gpu_layer_split_invariance(gpu_ctx, rk):
    """The GPU's own two-stream + adding pair checked against itself, no oracle involved (tests/conftest.py split_layers): 72 layers
    against the same atmosphere in 144 half layers, clear sky with aerosols."""
    from geosradiation_gridcomp_amd import synth
    from tests.conftest import split_layers
    inp = synth.make_columns(64, 72, start=910, cloudy_frac=0.0, aerosol=True)
    a = gpu_ctx[rk].rrtmg_sw_columns(inp, iaer=10); b = gpu_ctx[rk].rrtmg_sw_columns(split_layers(inp), iaer=10)
    toa = a["swdflx"][72].astype(np.float64)
    for k in ("swuflx", "swdflx"):
        d = (np.abs(b[k][0::2].astype(np.float64) - a[k].astype(np.float64)) / toa).max()
        assert d <= (5e-5 if rk == 8 else 3e-4), (k, d)
