"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
  (a) the committed golden vectors (outputs of the reference's own Fortran, tests/golden/),
  (b) the pinned plain-C oracle on fresh seeded inputs,
  (c) size-independent properties at BASELINE's full size (100 000 columns).
Tolerances: real_kind 8 vs the reference built with -fdefault-real-8: <= 1e-6 W m-2 (the north-star bar);
real_kind 4 vs the default-real reference: <= 2e-3 W m-2 (fp32 fluxes of O(400) W m-2 carry ~3e-5 per
rounding; the reference's own r4 and r8 builds differ by up to 5e-4 on these columns)."""
import numpy as np
import pytest
from tests.conftest import load_golden, GOLDEN_CASES, FLUX, sub_columns

pytestmark = pytest.mark.gpu

TOL_FLUX = {4: 2e-3, 8: 1e-6}
TOL_DFDT = {4: 2e-5, 8: 1e-8}


def _kind(rk):
    return "r4" if rk == 4 else "r8"


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("rk", [8, 4])
def test_fluxes_match_reference_golden(gpu_ctx, name, rk):
    ctx = gpu_ctx[rk]
    inp, g, ih = load_golden(name)
    kind = _kind(rk)
    ctx.set_inhomogeneity(ih)
    bo = np.ones(16, dtype=np.int32) if f"{kind}_olrb" in g else None
    o = ctx.rrtmg_lw_columns(inp, band_output=bo)
    ctx.set_inhomogeneity(0)
    for k in FLUX:
        tol = TOL_DFDT[rk] if "dTs" in k else TOL_FLUX[rk]
        err = np.abs(o[k].astype(np.float64) - g[f"{kind}_{k}"].astype(np.float64)).max()
        assert err <= tol, (k, err)
    if rk == 8:
        np.testing.assert_array_equal(o["clearCounts"], g[f"{kind}_clearCounts"])
    else:   # fp32 exp() of the overlap correlations may flip a sub-column at the 1e-7 level
        assert np.abs(o["clearCounts"] - g[f"{kind}_clearCounts"]).max() <= 1
    if bo is not None:
        assert np.abs(o["olrb"] - g[f"{kind}_olrb"]).max() <= TOL_FLUX[rk]
        assert np.abs(o["dolrb_dTs"] - g[f"{kind}_dolrb_dTs"]).max() <= TOL_DFDT[rk]


@pytest.mark.parametrize("ice", [0, 1, 2, 4])
@pytest.mark.parametrize("rk", [8, 4])
def test_cldprmc_iceflags_match_reference_golden(gpu_ctx, rk, ice):
    """LW cldprmc's ice parameterisations other than the GEOS default (rrtmg_lw_cldprmc.F90:138-226,270-316) against the reference's
    own fluxes (tests/golden/lw_iceflags_72.npz)."""
    ctx = gpu_ctx[rk]
    inp, g, ih = load_golden("lw_iceflags_72")
    kind = _kind(rk)
    ctx.set_inhomogeneity(ih)
    o = ctx.rrtmg_lw_columns(inp, iceflg=ice)
    ctx.set_inhomogeneity(0)
    for k in FLUX:
        tol = TOL_DFDT[rk] if "dTs" in k else TOL_FLUX[rk]
        err = np.abs(o[k].astype(np.float64) - g[f"{kind}_ice{ice}_{k}"].astype(np.float64)).max()
        assert err <= tol, (k, err)
    assert np.abs(o["clearCounts"] - g[f"{kind}_ice{ice}_clearCounts"]).max() <= (0 if rk == 8 else 1)


@pytest.mark.parametrize("name", ["lw_aer_72", "lw_cloudy_ih2_137"])
@pytest.mark.parametrize("rk", [8, 4])
def test_taumol_intermediates_match_reference(gpu_ctx, name, rk):
    ctx = gpu_ctx[rk]
    inp, g, _ = load_golden(name)
    kind = _kind(rk)
    taug, pfr = ctx.rrtmg_lw_taumol(sub_columns(inp, 2))
    rt = 1e-11 if rk == 8 else 5e-4     # fp32: a few cells suffer cancellation in the cubic edge weights
    np.testing.assert_allclose(taug, g[f"{kind}_taug2"], rtol=rt, atol=1e-30 if rk == 8 else 1e-12)
    np.testing.assert_allclose(pfr, g[f"{kind}_pfracs2"], rtol=rt, atol=0)


@pytest.mark.parametrize("name", ["lw_cloudy_ih1_72", "lw_cloudy_ih0_72", "lw_cloudy_ih2_137"])
@pytest.mark.parametrize("rk", [8, 4])
def test_mcica_generator_matches_reference(gpu_ctx, name, rk):
    ctx = gpu_ctx[rk]
    inp, g, ih = load_golden(name)
    kind = _kind(rk)
    s4 = sub_columns(inp, 4)
    nlay = s4["play"].shape[0]
    ctx.set_inhomogeneity(ih)
    for tag, nsub, so in (("lw", 140, (1, 2, 3, 4)), ("sw", 112, (4, 3, 2, 1))):
        cl, ci, cw = ctx.generate_stochastic_clouds(4, nsub, nlay, s4["zm"], s4["alat"], int(inp["dyofyr"]), s4["play"], s4["cldf"],
                                                    s4["ciwp"], s4["clwp"], 1e-20, seed_order=so)
        ref = g[f"{kind}_mc_{tag}_cldy"].astype(np.int32)
        nflip = int((cl != ref).sum())
        assert nflip == 0 if rk == 8 else nflip <= 2, nflip          # integer KISS stream is bit-exact
        same = cl == ref
        for got, want in ((ci, g[f"{kind}_mc_{tag}_ciwp"]), (cw, g[f"{kind}_mc_{tag}_clwp"])):
            rel = np.abs(got[same] - want[same]) / np.maximum(np.abs(want[same]), 1e-300)
            if rk == 8:
                assert rel.max() <= 1e-13, rel.max()
            else:
                # fp32: exp() of the condensate overlap correlation differs by an ulp between ocml and libm, so a
                # handful of `cdf2 < rcorr` decisions (cloud_subcol_gen.F90:424) fall the other way
                assert (rel > 1e-6).mean() <= 1e-3, (rel > 1e-6).mean()
        cnt = ctx.clearCounts_threeBand(4, nsub, nlay, int(inp["cloudLM"]), int(inp["cloudMH"]), cl)
        from oracle import clib
        np.testing.assert_array_equal(cnt, clib.clearcounts(cl, int(inp["cloudLM"]), int(inp["cloudMH"])))
    ctx.set_inhomogeneity(0)


@pytest.mark.parametrize("rk", [8, 4])
def test_against_oracle_on_fresh_columns(gpu_ctx, rk):
    """seeded inputs the golden files do not contain, checked against the pinned C oracle."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[rk]
    kind = _kind(rk)
    inp = synth.make_columns(200, 72, start=31337, aerosol=True, cloudy_frac=0.6)
    for ih in (1, 0):
        ctx.set_inhomogeneity(ih); clib.set_inhomogeneity(ih, kind)
        o = ctx.rrtmg_lw_columns(inp, dudTs=(ih == 1))
        r = clib.rrtmg_lw(inp, kind, dudTs=(ih == 1))
        assert r["rc"] == 0
        for k in FLUX:
            if "dTs" in k and ih == 0:
                continue
            tol = TOL_DFDT[rk] if "dTs" in k else TOL_FLUX[rk]
            assert np.abs(o[k].astype(np.float64) - r[k].astype(np.float64)).max() <= tol, k
        if rk == 8:
            np.testing.assert_array_equal(o["clearCounts"], r["clearCounts"])
    ctx.set_inhomogeneity(0); clib.set_inhomogeneity(0, kind)


def test_generator_stops_at_the_waves_highest_cloud(gpu_ctx):
    """k_mcica ends its walk above the highest layer in which a column of the wave has cloud fraction (the reference consumes the streams
    up to the top; nothing there can become cloudy).  Cloud tops from the lowest to the very top layer inside one 64-column wave, a wave
    whose only cloud is the top layer, one with cloud in the lowest layer only: clearCounts and fluxes of RRTMG_LW and RRTMG_SW against
    the oracle in fp64 (which walks every layer)."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    n, nlay = 192, 72
    inp = synth.make_columns(n, nlay, start=777, aerosol=True, cloudy_frac=1.0)
    cldf, clwp, ciwp = inp["cldf"].copy(), inp["clwp"].copy(), inp["ciwp"].copy()
    cldf[:, 64:] = 0; clwp[:, 64:] = 0; ciwp[:, 64:] = 0
    for c in range(0, 64, 3):                         # wave 0: the synthetic decks + a thin cloud anywhere up to the top layer
        l = (c * 71) // 63
        cldf[l, c] = 0.4; ciwp[l, c] = 6.0
    cldf[nlay - 1, 64 + 17] = 0.7; ciwp[nlay - 1, 64 + 17] = 3.0          # wave 1: one column, top layer only
    cldf[0, 128 + 5] = 0.5; clwp[0, 128 + 5] = 40.0                       # wave 2: one column, lowest layer only
    inp.update(cldf=cldf, clwp=clwp, ciwp=ciwp)
    ctx = gpu_ctx[8]
    ctx.set_inhomogeneity(1); clib.set_inhomogeneity(1, "r8")
    try:
        o = ctx.rrtmg_lw_columns(inp); r = clib.rrtmg_lw(inp, "r8")
        q = ctx.rrtmg_sw_columns(inp, iaer=10, normFlx=1); qr = clib.rrtmg_sw(inp, prec="r8", iaer=10, normFlx=1)
    finally:
        ctx.set_inhomogeneity(0); clib.set_inhomogeneity(0, "r8")
    assert r["rc"] == 0 and qr["rc"] == 0
    np.testing.assert_array_equal(o["clearCounts"], r["clearCounts"])
    np.testing.assert_array_equal(q["clearCounts"], qr["clearCounts"])
    assert (o["clearCounts"][0, 64 + 17] < 140) and (o["clearCounts"][0, 128 + 5] < 140)
    for k in FLUX:
        if "dTs" not in k:
            assert np.abs(o[k] - r[k]).max() <= TOL_FLUX[8], k
    for k in ("swuflx", "swdflx", "swuflxc", "swdflxc"):
        assert np.abs(q[k] - qr[k]).max() <= 1e-9, k


def test_wide_cloud_free_blocks_equal_the_narrow_ones(gpu_ctx):
    """A mostly cloud-free batch (>= 32 768 cloud-free columns, >= 4/5 of the batch) runs the cloud-free instantiation with 768-thread blocks,
    4 g-points per evaluation (lw_kernels.hpp k_lw_bands<..., 768>); a small batch of the same columns runs the 256-thread, 8-g-point one.
    The same columns must come out bitwise the same: ragged sizes (not a multiple of 768 or 256), a mixed block at the clear | cloudy
    boundary, 91 layers."""
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    n, nlay = 36_111, 91
    inp = synth.make_columns(n, nlay, start=123_456, cloudy_frac=0.03, aerosol=True)
    ncld = int((inp["cldf"] > 0).any(axis=0).sum())
    assert n - ncld >= 32_768 and 5 * (n - ncld) >= 4 * n and ncld > 500
    ctx.set_inhomogeneity(1)
    try:
        big = ctx.rrtmg_lw_columns(inp, dudTs=True)
        for lo in (0, 20_000, n - 777):
            sl = slice(lo, lo + 777)
            sub = {k: (np.ascontiguousarray(v[..., sl]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == n else v) for k, v in inp.items()}
            small = ctx.rrtmg_lw_columns(sub, dudTs=True)
            for k in FLUX + ("clearCounts",):
                np.testing.assert_array_equal(small[k], big[k][..., sl], err_msg=k)
    finally:
        ctx.set_inhomogeneity(0)
    assert np.isfinite(big["uflx"]).all() and (big["uflx"][nlay] > 50).all()


def test_chunking_is_invisible(gpu_ctx):
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    inp = synth.make_columns(300, 72, start=99, cloudy_frac=0.5, aerosol=True)
    ctx.set_inhomogeneity(1)
    a = ctx.rrtmg_lw_columns(inp)
    ctx.set_chunk(128)            # 3 ragged batches: 128 + 128 + 44
    b = ctx.rrtmg_lw_columns(inp)
    ctx.set_chunk(131072)
    ctx.set_inhomogeneity(0)
    for k in FLUX + ("clearCounts",):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_reference_error_stops_become_errors(gpu_ctx):
    from geosradiation_gridcomp_amd.api import GeosradInputError
    ctx = gpu_ctx[4]
    inp, _, _ = load_golden("lw_clear_72")
    bad = dict(inp); bad["tlay"] = inp["tlay"].copy(); bad["tlay"][3, 2] = -1.0
    with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
        ctx.rrtmg_lw_columns(bad)
    bad = dict(inp); bad["play"] = inp["play"].copy(); bad["play"][40, 1] = 500.0
    with pytest.raises(GeosradInputError, match="RRTMG LW pressure misordering"):
        ctx.rrtmg_lw_columns(bad)
    bad = dict(inp); bad["cloudLM"] = 5; bad["cloudMH"] = 5
    with pytest.raises(GeosradInputError, match="invalid pressure super-layers"):
        ctx.rrtmg_lw_columns(bad)
    with pytest.raises(GeosradInputError, match="invalid iceflag"):
        ctx.rrtmg_lw_columns(inp, iceflg=7)
    cl, _, _ = load_golden("lw_cloudy_ih0_72")
    bad = dict(cl); bad["rel"] = cl["rel"].copy(); bad["rel"][:] = 90.0     # > 60 um: liqflag 1 extrapolation forbidden
    with pytest.raises(GeosradInputError, match="high-radius extrapolation forbidden"):
        ctx.rrtmg_lw_columns(bad)
    # and the context still works afterwards
    o = ctx.rrtmg_lw_columns(inp)
    assert np.isfinite(o["uflx"]).all()


def test_full_size_properties(gpu_ctx):
    """BASELINE config 2 size (100 000 clear-sky columns, 72 layers): properties that need no oracle."""
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    n = 100_000
    inp = synth.make_columns(n, 72)
    o = ctx.rrtmg_lw_columns(inp)
    for k in FLUX:
        assert np.isfinite(o[k]).all()
    np.testing.assert_array_equal(o["uflx"], o["uflxc"])          # no cloud: clear == total, bitwise
    np.testing.assert_array_equal(o["dflx"], o["dflxc"])
    assert (o["dflx"][-1] == 0).all()                              # no downward LW at TOA
    assert (o["clearCounts"] == 140).all()
    assert (np.diff(o["dflx"], axis=0) <= 1e-3).all()              # downward flux grows towards the surface
    sigma_t4 = 5.670374e-8 * inp["tsfc"].astype(np.float64) ** 4
    assert np.abs(o["uflx"][0] / sigma_t4 - 1.0).max() < 0.03       # surface emission ~ eps sigma T^4 (+ reflection)
    assert (o["uflx"][-1] > 120).all() and (o["uflx"][-1] < 400).all()
    assert (o["duflx_dTs"][0] > 3).all() and (o["duflx_dTs"] >= 0).all()
    # column independence: any shard equals the same columns computed alone, bitwise
    sl = slice(54_321, 54_321 + 257)
    shard = synth.make_columns(257, 72, start=54_321)
    p = ctx.rrtmg_lw_columns(shard)
    for k in FLUX:
        np.testing.assert_array_equal(p[k], o[k][:, sl], err_msg=k)
    # spot parity against the oracle on 32 of the 100 000 columns
    from oracle import clib
    r = clib.rrtmg_lw(sub_columns(shard, 32), "r4")
    for k in ("uflx", "dflx"):
        assert np.abs(p[k][:, :32] - r[k]).max() <= TOL_FLUX[4]


def test_device_pointer_entry_points_match_host_entry_points(gpu_ctx):
    """the `_dev` variants (HBM-resident arrays + caller's stream) give the same bits as the host-pointer ones"""
    import torch
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    ncol, nlay, nsub = 96, 72, 200
    inp = synth.make_columns(ncol, nlay, start=4321, cloudy_frac=0.7, aerosol=True)
    dev = torch.device("cuda", 0)
    ctx.set_inhomogeneity(1)
    # stand-alone McICA generator with BASELINE's 200 sub-columns
    cl, ci, cw = ctx.generate_stochastic_clouds(ncol, nsub, nlay, inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"],
                                                inp["ciwp"], inp["clwp"], 1e-20)
    d = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in ("zm", "alat", "play", "cldf", "ciwp", "clwp")}
    d["cldy_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=torch.int32, device=dev)
    d["ciwp_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=torch.float32, device=dev)
    d["clwp_stoch"] = torch.zeros((ncol, nsub, nlay), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ctx.generate_stochastic_clouds_dev(st, ncol, nsub, nlay, {k: v.data_ptr() for k, v in d.items()}, int(inp["dyofyr"]), 1e-20)
    ctx.check(st)
    np.testing.assert_array_equal(d["cldy_stoch"].cpu().numpy(), cl)
    np.testing.assert_array_equal(d["ciwp_stoch"].cpu().numpy(), ci)
    np.testing.assert_array_equal(d["clwp_stoch"].cpu().numpy(), cw)
    assert cl.any() and not cl.all()
    # RRTMG_LW / RRTMG_SW
    from bench import LW_IN2D, SW_IN, SW_AER, SW_OUT1
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat", "tauaer"] + LW_IN2D + SW_IN + SW_AER
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in names}
    for k in FLUX + ("swuflx", "swdflx", "swuflxc", "swdflxc"):
        t[k] = torch.zeros((nlay + 1, ncol), device=dev)
    for k in SW_OUT1:
        t[k] = torch.zeros(ncol, device=dev)
    t["fswband"] = torch.zeros((14, ncol), device=dev)
    t["clearCounts"] = torch.zeros((4, ncol), dtype=torch.int32, device=dev)
    t["clearCounts_sw"] = torch.zeros((4, ncol), dtype=torch.int32, device=dev)
    ptr = {k: v.data_ptr() for k, v in t.items()}
    doy, lm, mh = int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])
    ctx.rrtmg_lw_dev(st, ncol, nlay, True, ptr, 3, 1, doy, lm, mh)
    ctx.rrtmg_sw_dev(st, ncol, nlay, 1361.0, 1.0, 0, ptr, 3, 1, doy, 10, lm, mh, normFlx=1)
    ctx.check(st)
    h_lw = ctx.rrtmg_lw_columns(inp)
    h_sw = ctx.rrtmg_sw_columns(inp, iaer=10, normFlx=1)
    ctx.set_inhomogeneity(0)
    for k in FLUX:
        np.testing.assert_array_equal(t[k].cpu().numpy(), h_lw[k], err_msg=k)
    for k in ("swuflx", "swdflx", "swuflxc", "swdflxc", "nirr", "parf", "fswband", "cotdtp"):
        np.testing.assert_array_equal(t[k].cpu().numpy(), h_sw[k], err_msg=k)
    np.testing.assert_array_equal(t["clearCounts"].cpu().numpy(), h_lw["clearCounts"])
    np.testing.assert_array_equal(t["clearCounts_sw"].cpu().numpy(), h_sw["clearCounts"])


@pytest.mark.parametrize("rk", [8, 4])
def test_rats_passes_equal_separate_calls_with_the_gas_removed(gpu_ctx, rk):
    """RATS loop of LW_Driver (GEOS_IrradGridComp.F90:3405-3468): rrtmg_lw once more per listed gas with that gas zeroed.  Here the
    gases ride on ONE call (shared input checks, partition, overlap, McICA sub-columns): bitwise equal to separate calls, equal
    to the pinned oracle run on the zeroed inputs within the flux tolerance; chunked batches included."""
    import torch
    from oracle import clib
    from geosradiation_gridcomp_amd import synth, gridcomp as G
    ctx = gpu_ctx[rk]; dt = ctx.dtype
    ncol, nlay = 300, 72
    inp = synth.make_columns(ncol, nlay, start=31337, cloudy_frac=0.6, aerosol=True)
    gases = ["H2O", "CO2", "O3", "CH4", "N2O", "CFC11", "CFC12", "HCFC22"]
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat", "tauaer"] + list(G.RAT_VMR.values()) + \
            ["o2vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel"]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=dt)).cuda() for k in names}
    tdt = t["play"].dtype
    for k in FLUX:
        t[k] = torch.zeros((nlay + 1, ncol), dtype=tdt, device="cuda")
    for k in ("uflx_rat", "dflx_rat", "duflx_dTs_rat"):
        t[k] = torch.full((len(gases), nlay + 1, ncol), -7.0, dtype=tdt, device="cuda")
    t["clearCounts"] = torch.zeros((4, ncol), dtype=torch.int32, device="cuda")
    ptr = {k: v.data_ptr() for k, v in t.items()}
    st = torch.cuda.current_stream().cuda_stream
    doy, lm, mh = int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])
    ctx.set_inhomogeneity(1)
    try:
        got = {}
        for chunk in (131072, 128):          # one batch; three batches, the last ragged
            ctx.set_chunk(chunk)
            ctx.rrtmg_lw_rats_dev(st, ncol, nlay, True, ptr, 3, 1, doy, lm, mh, gases)
            ctx.check(st)
            got[chunk] = {k: t[k].cpu().numpy().copy() for k in FLUX + ("uflx_rat", "dflx_rat", "duflx_dTs_rat", "clearCounts")}
        ctx.set_chunk(131072)
        for k in got[128]:
            np.testing.assert_array_equal(got[128][k], got[131072][k], err_msg=k)
        g = got[131072]
        full = ctx.rrtmg_lw_columns(inp)
        for k in FLUX + ("clearCounts",):
            np.testing.assert_array_equal(g[k], full[k], err_msg=k)
        for r, gas in enumerate(gases):
            z = dict(inp); z[G.RAT_VMR[gas]] = np.zeros_like(inp[G.RAT_VMR[gas]])
            sep = ctx.rrtmg_lw_columns(z)
            for k in ("uflx", "dflx", "duflx_dTs"):
                np.testing.assert_array_equal(g[k + "_rat"][r], sep[k], err_msg=(gas, k))
            assert np.abs(sep["uflx"][-1] - full["uflx"][-1]).max() > (1e-3 if gas.startswith(("CFC", "HCFC")) else 0.1), gas
            if gas in ("H2O", "CO2"):        # the dry column (pwvcm = 0) and a key species of nine bands, against the oracle
                clib.set_inhomogeneity(1, _kind(rk))
                o = clib.rrtmg_lw(z, _kind(rk))
                clib.set_inhomogeneity(0, _kind(rk))
                # fp64: the north-star bar.  fp32 against the fp32 oracle: 4e-3 (measured 2.1e-3 on the CO2-free columns, 70 ulp of
                # a 360 W m-2 flux: the two fp32 evaluations order their sums differently; the bitwise check above is the strict one)
                tol = TOL_FLUX[rk] if rk == 8 else 4e-3
                for k in ("uflx", "dflx"):
                    assert np.abs(g[k + "_rat"][r].astype(np.float64) - o[k].astype(np.float64)).max() <= tol, (gas, k)
    finally:
        ctx.set_chunk(131072)
        ctx.set_inhomogeneity(0)


@pytest.mark.parametrize("rk", [4, 8])
@pytest.mark.parametrize("ih", [0, 1])
def test_standalone_generator_pair_mapping_equals_column_mapping(gpu_ctx, rk, ih):
    """generate_stochastic_clouds stand-alone: k_mcica_sa (lane = (column, sub-column) pair, outputs written row-major through an LDS
    tile) against k_mcica<R, 1> (lane = column), which test_mcica_generator_matches_reference pins to the reference's masks:
    identical bits for sub-column counts below, at and above a wavefront's 64 pairs, ragged last waves included."""
    import os
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[rk]
    ncol, nlay = 37, 72
    inp = synth.make_columns(ncol, nlay, start=777, cloudy_frac=0.7, aerosol=False)
    a = (inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"], inp["ciwp"], inp["clwp"], 1e-20)
    ctx.set_inhomogeneity(ih)
    try:
        for nsub in (1, 8, 64, 140, 200, 333):
            new = ctx.generate_stochastic_clouds(ncol, nsub, nlay, *a)
            os.environ["GEOSRAD_MCICA_LANE_COLUMN"] = "1"
            try:
                old = ctx.generate_stochastic_clouds(ncol, nsub, nlay, *a)
            finally:
                del os.environ["GEOSRAD_MCICA_LANE_COLUMN"]
            for x, y, nm in zip(new, old, ("cldy_stoch", "ciwp_stoch", "clwp_stoch")):
                np.testing.assert_array_equal(x, y, err_msg=f"{nm} nsubcol={nsub}")
            assert nsub < 8 or (new[0].any() and not new[0].all())
    finally:
        ctx.set_inhomogeneity(0)


def test_rats_argument_errors_and_empty_list(gpu_ctx):
    """no gas listed = the plain call; unknown gas codes, missing output arrays and input errors are refused (negative values in the
    inputs are reported by the shared input check, as the reference's first rrtmg_lw call of the loop would)"""
    import torch
    from geosradiation_gridcomp_amd.api import GeosradError, GeosradInputError
    from geosradiation_gridcomp_amd import synth, gridcomp as G
    ctx = gpu_ctx[4]
    ncol, nlay = 70, 72
    inp = synth.make_columns(ncol, nlay, start=99, cloudy_frac=0.5, aerosol=False)
    names = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "zm", "alat"] + list(G.RAT_VMR.values()) + \
            ["o2vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel"]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).cuda() for k in names}
    for k in FLUX:
        t[k] = torch.zeros((nlay + 1, ncol), device="cuda")
    t["clearCounts"] = torch.zeros((4, ncol), dtype=torch.int32, device="cuda")
    for k in ("uflx_rat", "dflx_rat", "duflx_dTs_rat"):
        t[k] = torch.zeros((1, nlay + 1, ncol), device="cuda")
    ptr = {k: v.data_ptr() for k, v in t.items()}
    st = torch.cuda.current_stream().cuda_stream
    a = (st, ncol, nlay, True, ptr, 3, 1, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]))
    ctx.rrtmg_lw_rats_dev(*a, [])
    ctx.check(st)
    want = ctx.rrtmg_lw_columns(inp)
    for k in FLUX:
        np.testing.assert_array_equal(t[k].cpu().numpy(), want[k], err_msg=k)
    with pytest.raises(GeosradError, match="unknown gas code"):
        ctx.rrtmg_lw_rats_dev(*a, [8])
    with pytest.raises(GeosradError, match="at most 8 gases|bad RATS"):
        ctx.rrtmg_lw_rats_dev(*a, list(range(8)) + [0])
    p2 = dict(ptr); p2["dflx_rat"] = 0
    with pytest.raises(GeosradError, match="must not be null"):
        ctx.rrtmg_lw_rats_dev(st, ncol, nlay, True, p2, 3, 1, 180, int(inp["cloudLM"]), int(inp["cloudMH"]), ["CO2"])
    t["tlay"][5, 3] = -1.0
    ctx.rrtmg_lw_rats_dev(*a, ["CO2"])
    with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
        ctx.check(st)


def test_standalone_generator_full_size_is_batch_independent(gpu_ctx):
    """BASELINE configs[2] at full size (100 000 columns x 200 sub-columns x 72 layers, 17.3 GB of outputs, element offsets beyond
    2^32 bytes): a column's sub-columns do not depend on the batch it is generated in - the first, a middle and the last 70 columns
    of the big batch equal the same columns generated alone - and the clear columns are clear."""
    import torch
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[4]
    ncol, nlay, nsub, m = 100_000, 72, 200, 70
    ctx.set_inhomogeneity(1)
    try:
        inp = synth.make_columns(ncol, nlay, start=0, cloudy_frac=0.6, aerosol=False)
        keys = ("zm", "alat", "play", "cldf", "ciwp", "clwp")
        d = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).cuda() for k in keys}
        d["cldy_stoch"] = torch.full((ncol, nsub, nlay), -1, dtype=torch.int32, device="cuda")
        d["ciwp_stoch"] = torch.full((ncol, nsub, nlay), -1.0, device="cuda")
        d["clwp_stoch"] = torch.full((ncol, nsub, nlay), -1.0, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        ctx.generate_stochastic_clouds_dev(st, ncol, nsub, nlay, {k: v.data_ptr() for k, v in d.items()}, int(inp["dyofyr"]), 1e-20)
        ctx.check(st)
        assert int((d["cldy_stoch"] < 0).sum()) == 0 and float(d["ciwp_stoch"].min()) >= 0.0      # every cell written
        for c0 in (0, 54_321, ncol - m):
            sub = {k: (v[..., c0:c0 + m] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v) for k, v in inp.items()}
            cl, ci, cw = ctx.generate_stochastic_clouds(m, nsub, nlay, sub["zm"], sub["alat"], int(inp["dyofyr"]), sub["play"],
                                                        sub["cldf"], sub["ciwp"], sub["clwp"], 1e-20)
            np.testing.assert_array_equal(d["cldy_stoch"][c0:c0 + m].cpu().numpy(), cl)
            np.testing.assert_array_equal(d["ciwp_stoch"][c0:c0 + m].cpu().numpy(), ci)
            np.testing.assert_array_equal(d["clwp_stoch"][c0:c0 + m].cpu().numpy(), cw)
            assert cl.any()
        clearcol = torch.from_numpy((inp["cldf"] > 0).any(axis=0) == False).cuda()      # noqa: E712
        assert int(d["cldy_stoch"][clearcol].sum()) == 0 and int(clearcol.sum()) > 30_000
        cloudy_cells = d["cldy_stoch"] != 0
        assert bool(((d["ciwp_stoch"] > 0) | (d["clwp_stoch"] > 0))[cloudy_cells].all())
        assert float(d["ciwp_stoch"][~cloudy_cells].max()) == 0.0
        del d, cloudy_cells
        torch.cuda.empty_cache()
    finally:
        ctx.set_inhomogeneity(0)


def test_fp32_discrete_flip_rate(gpu_ctx):
    """SURVEY 7 hard part 1: the fp32 instantiation cannot be bit-identical to the reference's r4 build where a DISCRETE decision hangs
    on the last bit of a transcendental (ocml vs libm).  The McICA decisions are the ones with visible consequences (`cdf2 < alpha`,
    `cdf2 < rcorr` with alpha, rcorr = exp(-dz/L), cloud_subcol_gen.F90:314-321,411,424): measured here over 2 000 cloudy columns
    against the reference-pinned r4 oracle, for both seedings, and bounded.  (fp64: zero flips, asserted in
    test_mcica_generator_matches_reference.)"""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[4]
    n = 2000
    inp = synth.make_columns(n, 72, start=250_000, cloudy_frac=1.0)
    ctx.set_inhomogeneity(1); clib.set_inhomogeneity(1, "r4")
    try:
        for nsub, so in ((140, (1, 2, 3, 4)), (112, (4, 3, 2, 1))):
            cl, ci, cw = ctx.generate_stochastic_clouds(n, nsub, 72, inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"],
                                                        inp["ciwp"], inp["clwp"], 1e-20, seed_order=so)
            rl, ri, rw = clib.mcica(inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"], inp["ciwp"], inp["clwp"], nsub,
                                    seed_order=so, prec="r4")
            cells = (inp["cldf"] > 0).sum() * nsub                       # cells in which a decision is taken at all
            mask_flips = int((cl != rl.astype(np.int32)).sum())
            same = cl == rl.astype(np.int32)
            wp_flips = int((np.abs(cw[same] - rw[same]) > 1e-5 * np.maximum(np.abs(rw[same]), 1e-30)).sum()
                           + (np.abs(ci[same] - ri[same]) > 1e-5 * np.maximum(np.abs(ri[same]), 1e-30)).sum())
            cols = int((cl != rl.astype(np.int32)).any(axis=(1, 2)).sum())
            print(f"fp32 flip rate, nsubcol {nsub}: {mask_flips} mask flips and {wp_flips} condensate-rank flips in {cells} cloudy-layer "
                  f"cells ({mask_flips / cells:.2e}, {wp_flips / cells:.2e}); {cols} of {n} columns have a flipped mask cell")
            assert mask_flips <= 2e-5 * cells and wp_flips <= 2e-4 * cells and cols <= 0.01 * n
    finally:
        ctx.set_inhomogeneity(0); clib.set_inhomogeneity(0, "r4")


def test_lw_fp32_is_as_accurate_as_the_reference_precision(gpu_ctx):
    """The fp32 instantiation against the r8 oracle (pinned bit for bit to the reference's -fdefault-real-8 build) as the truth, next to
    the r4 oracle (pinned to the reference's default-real build, GEOS's production precision) on the same columns.  r4 and r8 builds
    seed McICA from different pressure bits, so the comparison uses the clear-sky fluxes of every column and the total-sky fluxes of
    the cloud-free ones."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    n, nlay = 1024, 72
    inp = synth.make_columns(n, nlay, start=710_000, aerosol=True, cloudy_frac=0.6)
    ctx = gpu_ctx[4]
    ctx.set_inhomogeneity(1)
    try:
        g4 = ctx.rrtmg_lw_columns(inp)
    finally:
        ctx.set_inhomogeneity(0)
    o = {}
    for kind in ("r4", "r8"):
        clib.set_inhomogeneity(1, kind)
        try:
            o[kind] = clib.rrtmg_lw(inp, prec=kind)
        finally:
            clib.set_inhomogeneity(0, kind)
    free = ~(inp["cldf"] > 0).any(axis=0)
    assert free.sum() >= 300

    def err(x):
        e = np.zeros(n)
        for k in ("uflxc", "dflxc"):
            e = np.maximum(e, np.abs(x[k].astype(np.float64) - o["r8"][k].astype(np.float64)).max(axis=0))
        for k in ("uflx", "dflx"):
            e = np.maximum(e, np.where(free, np.abs(x[k].astype(np.float64) - o["r8"][k].astype(np.float64)).max(axis=0), 0.0))
        return e

    eg, eo = err(g4), err(o["r4"])
    q = lambda e: (np.median(e), np.percentile(e, 99), e.max())
    print("fp32 error vs r8 oracle, W m-2 (median, 99 %%, max): GPU %.2e %.2e %.2e | r4 oracle %.2e %.2e %.2e" % (q(eg) + q(eo)))
    assert np.median(eg) <= 1.25 * np.median(eo) + 1e-6
    assert np.percentile(eg, 99) <= 1.25 * np.percentile(eo, 99) + 1e-5
    assert eg.max() <= max(1.5 * eo.max(), 2e-3)      # measured: GPU 7.7e-5 / 1.1e-3 / 2.2e-3 W m-2, r4 oracle 1.7e-4 / 1.1e-3 / 2.2e-3


@pytest.mark.parametrize("nsub,nlay,ncol", [(1, 72, 9), (7, 4, 70), (513, 72, 3), (200, 203, 5), (65, 137, 130)])
def test_mcica_generator_odd_shapes(gpu_ctx, nsub, nlay, ncol):
    """The stand-alone generator on shapes that straddle its tiling (k_mcica_sa: 64 (column, sub-column) pairs per chunk, 8 chunks per
    wavefront, an LDS tile of 64 x nlay; beyond the LDS the lane = column kernel): one sub-column, sub-column counts that are no multiple
    of anything, the smallest and the largest layer count - fp64 against the pinned oracle, masks bit for bit."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    ctx = gpu_ctx[8]
    inp = synth.make_columns(ncol, nlay, start=9000 + nsub, cloudy_frac=0.8)
    for ih in (0, 1):
        ctx.set_inhomogeneity(ih); clib.set_inhomogeneity(ih, "r8")
        try:
            cl, ci, cw = ctx.generate_stochastic_clouds(ncol, nsub, nlay, inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"],
                                                        inp["ciwp"], inp["clwp"], 1e-20, seed_order=(4, 3, 2, 1))
            rl, ri, rw = clib.mcica(inp["zm"], inp["alat"], int(inp["dyofyr"]), inp["play"], inp["cldf"], inp["ciwp"], inp["clwp"], nsub,
                                    seed_order=(4, 3, 2, 1), prec="r8")
        finally:
            ctx.set_inhomogeneity(0); clib.set_inhomogeneity(0, "r8")
        np.testing.assert_array_equal(cl, rl.astype(np.int32))
        np.testing.assert_allclose(ci, ri, rtol=1e-13, atol=0); np.testing.assert_allclose(cw, rw, rtol=1e-13, atol=0)
        assert cl.sum() > 0


def test_multi_device_context_is_bitwise_the_single_device_one(gpu_ctx):
    """geosrad_create_multi: the host-array entry points cut the columns into one contiguous shard per listed device and run
    the shards concurrently (in place in the caller's arrays: leading dimension = the full column count).  With device_ids = {0, 0}
    on the one GPU there is: RRTMG_LW, RRTMG_SW, irrad and sorad must return the bits of the single-device context, on a ragged
    column count that straddles the 16 384-column chunks of the host pipeline."""
    from geosradiation_gridcomp_amd import synth
    from geosradiation_gridcomp_amd.api import Context, GeosradError
    n, nlay = 33_001, 72
    inp = synth.make_columns(n, nlay, start=910_000, cloudy_frac=0.5, aerosol=True)
    one = gpu_ctx[4]
    two = Context(4, devices=[0, 0])
    try:
        one.set_inhomogeneity(1); two.set_inhomogeneity(1)
        a = one.rrtmg_lw_columns(inp, band_output=np.ones(16, np.int32)); b = two.rrtmg_lw_columns(inp, band_output=np.ones(16, np.int32))
        for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs", "clearCounts", "olrb", "dolrb_dTs"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        a = one.rrtmg_sw_columns(inp, iaer=10, normFlx=1, do_drfband=True); b = two.rrtmg_sw_columns(inp, iaer=10, normFlx=1, do_drfband=True)
        for k in ("swuflx", "swdflx", "swuflxc", "swdflxc", "nirr", "parf", "fswband", "drband", "dfband", "clearCounts", "cotdtp"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        m = 20_011
        small = {k: (np.ascontiguousarray(v[..., :m]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == n else v) for k, v in inp.items()}
        ch = synth.chou_lw_inputs(small, aerosol=True); cs = synth.chou_sw_inputs(small, aerosol=True)
        a = one.irrad_columns(ch); b = two.irrad_columns(ch)
        for k in ("flxu", "flxd", "flcu", "flcd", "dfdts", "sfcem", "taudiag", "taua"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        a = one.sorad_columns(cs, do_drfband=True); b = two.sorad_columns(cs, do_drfband=True)
        for k in ("flx", "flc", "flxu", "flcu", "fdiruv", "flx_sfc_band", "drband"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        # input assertions still surface, with the reference's message
        bad = dict(inp); bad["tlay"] = inp["tlay"].copy(); bad["tlay"][3, n - 5] = -1.0
        from geosradiation_gridcomp_amd.api import GeosradInputError
        with pytest.raises(GeosradInputError, match="negative values in input: tlay"):
            two.rrtmg_lw_columns(bad)
        # device-pointer entry points belong to one device
        with pytest.raises(GeosradError, match="single-device"):
            two.rrtmg_lw_dev(0, 8, nlay, True, {"clearCounts": 64}, 3, 1, 1, 10, 20)      # (never dereferenced)
    finally:
        one.set_inhomogeneity(0)
        two.close()
