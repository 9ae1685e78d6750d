"""GPU parity tests of the GridComp data path: the HIP kernels behind the C ABI against the plain-C
restatement oracle/gridcomp_oracle_impl.h.  The kernels perform the reference's statements operation by operation (no FMA
contraction), so the comparison is bitwise in both precisions."""
import numpy as np
import pytest

from geosradiation_gridcomp_amd import gridcomp as G
from geosradiation_gridcomp_amd import synth
from geosradiation_gridcomp_amd.api import GeosradError

pytestmark = pytest.mark.gpu
PREC = {4: "f32", 8: "f64"}


def _dev(arrs, dt):
    """numpy dict -> (torch tensors on cuda:0, name -> device address); None stays absent"""
    import torch
    t = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=dt)).cuda() for k, v in arrs.items() if isinstance(v, np.ndarray)}
    return t, {k: v.data_ptr() for k, v in t.items()}


def _zeros(shapes, dt):
    import torch
    tdt = torch.float32 if dt == np.float32 else torch.float64
    t = {k: torch.full(s, -7.0, dtype=tdt, device="cuda") for k, s in shapes.items()}
    return t, {k: v.data_ptr() for k, v in t.items()}


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rk", [4, 8])
def test_lw_driver_equals_oracle_prep_gpu_solver_oracle_post(gpu_ctx, rk):
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    ncol, lm = 300, 72          # two 256-column blocks, the second ragged
    inp = synth.make_columns(ncol, lm, start=900, cloudy_frac=0.6, aerosol=True)
    f = synth.geos_lw_fields(inp)
    consts = G.lwd_consts()
    ctx.set_inhomogeneity(1)
    tin, ptr = _dev(f, dt)
    shapes = {k: ((lm + 1, ncol) if k in G.LWD_OUT_3D else ((ncol, 16) if k in ("OLRB", "DOLRB") else (ncol,))) for k in G.LWD_OUT}
    tout, pout = _zeros(shapes, dt)
    ptr.update(pout)
    bo = np.zeros(16, dtype=np.int32); bo[[5, 9]] = 1
    ctx.lw_driver_rrtmg_dev(_stream(), ncol, lm, 16, ptr, consts, 3, 1, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"], band_output=bo)
    ctx.check(_stream())
    # the same three steps with the oracle either side of the GPU solver
    rr = clib.lwd_prep(f, consts, 3, 1, prec)
    inp2 = dict(rr); inp2.update(dyofyr=inp["dyofyr"], cloudLM=inp["cloudLM"], cloudMH=inp["cloudMH"])
    h = ctx.rrtmg_lw_columns(inp2, dudTs=True, band_output=bo)
    o = clib.lwd_post(h, h["clearCounts"], f["EMIS"], f["TS"], prec)
    ctx.set_inhomogeneity(0)
    for k in G.LWD_OUT[:16]:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)
    got = tout["OLRB"].cpu().numpy()
    np.testing.assert_array_equal(got[:, [5, 9]], h["olrb"][:, [5, 9]])
    assert (o["CLDTTLW"] > 0).any() and (o["CLDTTLW"] == 0).any()
    # GEOS conventions: upward negative, surface emission positive, OLR = -FLX_INT at the top
    assert (o["FLXU_INT"] < 0).all() and (o["SFCEM_INT"] > 0).all() and (o["FLX_INT"][0] < 0).all()


@pytest.mark.parametrize("rk", [4, 8])
def test_lw_driver_without_aerosols_and_exports_not_associated(gpu_ctx, rk):
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    ncol, lm = 64, 72
    inp = synth.make_columns(ncol, lm, start=40, cloudy_frac=0.0, aerosol=False)
    f = synth.geos_lw_fields(inp)
    f["TAUA"] = None; f["SSAA"] = None
    f["CO2_3D"] = f["T"] * 0 + 4.1e-4
    consts = G.lwd_consts()
    tin, ptr = _dev(f, dt)
    want = ["FLX_INT", "SFCEM_INT", "DFDTS"]
    tout, pout = _zeros({"FLX_INT": (lm + 1, ncol), "SFCEM_INT": (ncol,), "DFDTS": (lm + 1, ncol)}, dt)
    ptr.update(pout)
    ctx.lw_driver_rrtmg_dev(_stream(), ncol, lm, 0, ptr, consts, 3, 1, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"])
    ctx.check(_stream())
    rr = clib.lwd_prep(f, consts, 3, 1, prec)
    inp2 = dict(rr); inp2.update(dyofyr=inp["dyofyr"], cloudLM=inp["cloudLM"], cloudMH=inp["cloudMH"])
    h = ctx.rrtmg_lw_columns(inp2, dudTs=True)
    o = clib.lwd_post(h, h["clearCounts"], f["EMIS"], f["TS"], prec, want=want)
    for k in want:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)
    # a missing required field is refused
    bad = dict(ptr); bad.pop("Q")
    with pytest.raises(GeosradError):
        ctx.lw_driver_rrtmg_dev(_stream(), ncol, lm, 0, bad, consts, 3, 1, 180, f["LCLDLM"], f["LCLDMH"])


@pytest.mark.parametrize("rk", [4, 8])
def test_sw_driver_equals_oracle_prep_gpu_solver_oracle_post(gpu_ctx, rk):
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    ncol, lm = 300, 72
    inp = synth.make_columns(ncol, lm, start=5000, cloudy_frac=0.6, aerosol=True)
    f = synth.geos_sw_fields(inp)
    consts = G.swd_consts()
    ctx.set_inhomogeneity(1)
    tin, ptr = _dev(f, dt)
    shapes = {k: (ncol,) for k in G.SWD_OUT if not k.endswith("NA")}        # (the no-aerosol flavour has a test of its own)
    shapes.update({k: (lm + 1, ncol) for k in ("FSW", "FSC", "FSWU", "FSCU")}); shapes["FSWBAND"] = (14, ncol)
    tout, pout = _zeros(shapes, dt)
    ptr.update(pout)
    ctx.sw_driver_rrtmg_dev(_stream(), ncol, lm, 14, ptr, consts, 3, 1, 1361.0, 1.0, 0, int(inp["dyofyr"]), True, f["LCLDLM"], f["LCLDMH"], 1)
    ctx.check(_stream())
    rr, aer = clib.swd_prep(f, consts, 3, 1, prec)
    inp2 = dict(rr)
    for k in ("coszen", "alat", "asdir", "asdif", "aldir", "aldif", "dyofyr", "cloudLM", "cloudMH"):
        inp2[k] = inp[k]
    h = ctx.rrtmg_sw_columns(inp2, scon=1361.0, adjes=1.0, isolvar=0, iaer=10, normFlx=1)
    o = clib.swd_post(h, h["clearCounts"], h, True, consts[G.SWD_CONST.index("UNDEF")], prec)
    ctx.set_inhomogeneity(0)
    for k in o:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)
    for k, hk in (("NIRR", "nirr"), ("NIRF", "nirf"), ("PARR", "parr"), ("PARF", "parf"), ("UVRR", "uvrr"), ("UVRF", "uvrf"), ("FSWBAND", "fswband")):
        np.testing.assert_array_equal(tout[k].cpu().numpy(), h[hk], err_msg=k)
    # the aerosol triplet was normalised in place like the reference does (SOL:6116-6125)
    for k in ("TAUA", "SSAA", "ASYA"):
        np.testing.assert_array_equal(tin[k].cpu().numpy(), aer[k], err_msg=k)
    assert (o["COTLP"] == consts[-1]).any() and (o["COTLP"] != consts[-1]).any()


@pytest.mark.parametrize("n", [776, 777])       # 16-byte accesses (4 / 2 columns per thread) and the scalar fall-back
@pytest.mark.parametrize("rk", [4, 8])
def test_update_flx_export_and_tendencies_match_the_oracle(gpu_ctx, rk, n):
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    rng = np.random.default_rng(31)
    lm, lmh, llm = 72, 30, 47
    st = {k: rng.uniform(-400, 400, (lm + 1, n)) for k in G.LWU_IN if k not in ("TSINST", "TS_INT", "SFCEM_INT", "FCLD")}
    st["TSINST"] = rng.uniform(270, 300, n); st["TS_INT"] = st["TSINST"] + rng.uniform(-3, 3, n); st["SFCEM_INT"] = rng.uniform(300, 450, n)
    st["FCLD"] = rng.uniform(0, 1, (lm, n)) * (rng.uniform(0, 1, (lm, n)) < 0.02)
    undef = G.MAPL["UNDEF"]
    for rrtmg in (False, True):
        s2 = {k: v for k, v in st.items() if not (rrtmg and k in G.LWU_IN_NA)}
        want = G.LWU_OUT if not rrtmg else [k for k in G.LWU_OUT if k not in ("FLXD", "LWS", "OLC")]      # some exports not associated
        tin, ptr = _dev(s2, dt)
        tout, pout = _zeros({k: ((lm + 1, n) if k in G.LWU_OUT_3D else (n,)) for k in want}, dt)
        ptr.update(pout)
        ctx.lw_update_flx_dev(_stream(), n, lm, rrtmg, lmh, llm, undef, ptr)
        ctx.check(_stream())
        o = clib.lw_update_flx(s2, lm, rrtmg, lmh, llm, undef, prec, want=want)
        assert set(o) == set(want)
        for k in want:
            np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=f"{k} rrtmg={rrtmg}")
    # UPDATE_EXPORT
    sw = {k: rng.uniform(0, 1, (lm + 1, n)) for k in G.SWU_IN[1:9]}
    sw["SLR"] = rng.uniform(0, 1300, n); sw["FSWBANDN"] = rng.uniform(0, 1, (14, n)); sw["FSWBANDNAN"] = rng.uniform(0, 1, (14, n))
    tin, ptr = _dev(sw, dt)
    shp = lambda k: (lm + 1, n) if k in G.SWU_OUT_3D else ((14, n) if k in G.SWU_OUT_BAND else (n,))
    want = [k for k in G.SWU_OUT if k not in ("FSC", "OSRNA")]
    tout, pout = _zeros({k: shp(k) for k in want}, dt)
    ptr.update(pout)
    ctx.sw_update_export_dev(_stream(), n, lm, 14, ptr)
    ctx.check(_stream())
    o = clib.sw_update_export(sw, lm, 14, prec, want=want)
    for k in want:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)
    # an export without its internal is refused
    bad = dict(ptr); bad.pop("FSWUN")
    with pytest.raises(GeosradError):
        ctx.sw_update_export_dev(_stream(), n, lm, 14, bad)
    # heating rates
    rt = {k: rng.uniform(-300, 300, (lm + 1, n)) for k in G.RT_IN[1:8]}
    rt["PLE"] = np.cumsum(rng.uniform(100, 3000, (lm + 1, n)), axis=0)
    rt["DSFDTS"] = rng.uniform(4, 6, n); rt["SFCEM"] = rng.uniform(300, 450, n); rt["TRD"] = rng.uniform(270, 300, n)
    tin, ptr = _dev(rt, dt)
    tout, pout = _zeros({k: ((lm, n) if k in G.RT_OUT_3D else (n,)) for k in G.RT_OUT}, dt)
    ptr.update(pout)
    ctx.rad_tendencies_dev(_stream(), n, lm, G.MAPL["GRAV"], G.MAPL["CP"], ptr)
    ctx.check(_stream())
    o = clib.rad_tendencies(rt, lm, G.MAPL["GRAV"], G.MAPL["CP"], prec)
    for k in G.RT_OUT:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)


def test_update_flx_full_size_properties(gpu_ctx):
    """BASELINE configs[3] per-GPU size: no surface-temperature change -> the exports are the internals, bit for bit; the
    linearisation is exactly linear in the temperature change for a power-of-two step"""
    import torch
    ctx = gpu_ctx[4]
    n, lm = 97_200, 72
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    mk = lambda *s: torch.rand(*s, device="cuda", generator=g) * 800 - 400
    t = {k: mk(lm + 1, n) for k in G.LWU_IN if k not in ("TSINST", "TS_INT", "SFCEM_INT", "FCLD") and k not in G.LWU_IN_NA}
    t["TS_INT"] = torch.rand(n, device="cuda", generator=g) * 30 + 270
    t["TSINST"] = t["TS_INT"].clone(); t["SFCEM_INT"] = mk(n); t["FCLD"] = torch.zeros(lm, n, device="cuda")
    out = {k: torch.empty(lm + 1, n, device="cuda") for k in ("FLX", "FLCU", "FLXD")}
    out.update({k: torch.empty(n, device="cuda") for k in ("OLR", "SFCEM", "OLCC5", "CLDTT")})
    ptr = {k: v.data_ptr() for k, v in {**t, **out}.items()}
    ctx.lw_update_flx_dev(_stream(), n, lm, True, 30, 47, 1e15, ptr)
    ctx.check(_stream())
    assert torch.equal(out["FLX"], t["FLX_INT"]) and torch.equal(out["FLCU"], t["FLCU_INT"]) and torch.equal(out["FLXD"], t["FLXD_INT"])
    assert torch.equal(out["OLR"], -t["FLX_INT"][0]) and torch.equal(out["SFCEM"], t["SFCEM_INT"])
    assert torch.equal(out["CLDTT"], torch.zeros(n, device="cuda")) and torch.equal(out["OLCC5"], -t["FLC_INT"][0])
    t["TSINST"] = t["TS_INT"] + 2.0
    ptr["TSINST"] = t["TSINST"].data_ptr()
    ctx.lw_update_flx_dev(_stream(), n, lm, True, 30, 47, 1e15, ptr)
    ctx.check(_stream())
    delt = t["TSINST"] - t["TS_INT"]
    assert torch.equal(out["FLX"], t["FLX_INT"] + t["DFDTS"] * delt)


@pytest.mark.parametrize("rk", [4, 8])
def test_sw_driver_no_aerosol_flavour_shares_the_cloud_generator(gpu_ctx, rk):
    """FSWNA ... requested from the with-aerosol call (band sweeps repeated without aerosol terms; validation, setcoef, McICA shared)
    == a separate call of the driver without aerosols, which is what the GridComp does (GEOS_SolarGridComp.F90:3249-3259)"""
    ctx = gpu_ctx[rk]; dt = ctx.dtype
    ncol, lm = 200, 72
    inp = synth.make_columns(ncol, lm, start=777, cloudy_frac=0.6, aerosol=True)
    f = synth.geos_sw_fields(inp)
    consts = G.swd_consts()
    ctx.set_inhomogeneity(1)
    shapes = {k: (lm + 1, ncol) for k in ("FSW", "FSC", "FSWU", "FSCU", "FSWNA", "FSCNA", "FSWUNA", "FSCUNA")}
    shapes["FSWBAND"] = (14, ncol); shapes["FSWBANDNA"] = (14, ncol)
    tin, ptr = _dev(f, dt)
    tout, pout = _zeros(shapes, dt)
    ptr.update(pout)
    args = (3, 1, 1361.0, 1.0, 0, int(inp["dyofyr"]))
    ctx.sw_driver_rrtmg_dev(_stream(), ncol, lm, 14, ptr, consts, *args, True, f["LCLDLM"], f["LCLDMH"], 1)
    ctx.check(_stream())
    f0 = dict(f); f0["TAUA"] = None; f0["SSAA"] = None; f0["ASYA"] = None
    tin0, ptr0 = _dev(f0, dt)
    tout0, pout0 = _zeros({k: shapes[k] for k in ("FSW", "FSC", "FSWU", "FSCU", "FSWBAND")}, dt)
    ptr0.update(pout0)
    ctx.sw_driver_rrtmg_dev(_stream(), ncol, lm, 0, ptr0, consts, *args, False, f["LCLDLM"], f["LCLDMH"], 1)
    ctx.check(_stream())
    ctx.set_inhomogeneity(0)
    for a, b in (("FSWNA", "FSW"), ("FSCNA", "FSC"), ("FSWUNA", "FSWU"), ("FSCUNA", "FSCU"), ("FSWBANDNA", "FSWBAND")):
        np.testing.assert_array_equal(tout[a].cpu().numpy(), tout0[b].cpu().numpy(), err_msg=a)
    # and the aerosols do matter in the regular flavour
    assert np.abs(tout["FSC"].cpu().numpy() - tout["FSCNA"].cpu().numpy()).max() > 1e-3


def test_lw_driver_137_layers_single_column_and_chunked_batches(gpu_ctx):
    """edge sizes: BASELINE configs[4] layer count, one column, and a batch the solver splits into several chunks -- every column
    must come out the same whatever the batching (bitwise)"""
    from oracle import clib
    ctx = gpu_ctx[8]; dt = ctx.dtype
    consts = G.lwd_consts()

    def run(inp, ncol, lm):
        f = synth.geos_lw_fields(inp)
        tin, ptr = _dev(f, dt)
        tout, pout = _zeros({"FLX_INT": (lm + 1, ncol), "FLC_INT": (lm + 1, ncol), "DFDTS": (lm + 1, ncol), "SFCEM_INT": (ncol,),
                             "CLDTTLW": (ncol,)}, dt)
        ptr.update(pout)
        ctx.lw_driver_rrtmg_dev(_stream(), ncol, lm, 16, ptr, consts, 3, 1, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"])
        ctx.check(_stream())
        return f, {k: v.cpu().numpy() for k, v in tout.items()}

    ctx.set_inhomogeneity(2)
    inp = synth.make_columns(70, 137, start=31, cloudy_frac=0.5, aerosol=True)
    f, full = run(inp, 70, 137)
    rr = clib.lwd_prep(f, consts, 3, 1, "f64")
    inp2 = dict(rr); inp2.update(dyofyr=inp["dyofyr"], cloudLM=inp["cloudLM"], cloudMH=inp["cloudMH"])
    h = ctx.rrtmg_lw_columns(inp2, dudTs=True)
    o = clib.lwd_post(h, h["clearCounts"], f["EMIS"], f["TS"], "f64", want=list(full))
    for k in full:
        np.testing.assert_array_equal(full[k], o[k], err_msg=k)
    # one column (the first of the batch)
    one = conftest_sub(inp, 1)
    _, single = run(one, 1, 137)
    for k in full:
        np.testing.assert_array_equal(single[k][..., 0], full[k][..., 0], err_msg=k)
    # several chunks
    ctx.set_chunk(64)             # 64 + 6
    try:
        _, chunked = run(inp, 70, 137)
    finally:
        ctx.set_chunk(131072)
        ctx.set_inhomogeneity(0)
    for k in full:
        np.testing.assert_array_equal(chunked[k], full[k], err_msg=k)


def conftest_sub(inp, n):
    from tests.conftest import sub_columns
    return sub_columns(inp, n)


@pytest.mark.parametrize("rk", [4, 8])
def test_chou_branch_of_lw_driver_then_heartbeat(gpu_ctx, rk):
    """irrad on device fields, the driver's post step (derivative copies, net fluxes, SFCEM sign, TS_INT), then Update_Flx in its
    Chou-Suarez flavour (all four flux flavours) -- against numpy on the irrad outputs and the plain-C restatement of Update_Flx"""
    import torch
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    m, lm = 96, 72
    inp = synth.make_columns(m, lm, start=1500, cloudy_frac=0.6, aerosol=True)
    ch = synth.chou_lw_inputs(inp, aerosol=True)
    names_in = ("ple", "ta", "wa", "oa", "tb", "n2o", "ch4", "cfc11", "cfc12", "cfc22", "cwc", "fcld", "reff", "fs", "tg", "eg", "tv", "ev", "rv",
                "taua", "ssaa", "asya")
    t = {k: torch.from_numpy(np.ascontiguousarray(ch[k], dtype=dt)).cuda() for k in names_in}
    for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts"):
        t[k] = torch.zeros((lm + 1, m), dtype=t["ta"].dtype, device="cuda")
    t["sfcem"] = torch.zeros(m, dtype=t["ta"].dtype, device="cuda")
    t["taudiag"] = torch.zeros((10, lm, m), dtype=t["ta"].dtype, device="cuda")
    st = _stream()
    ctx.irrad_dev(st, m, lm, {k: v.data_ptr() for k, v in t.items()}, ch["co2"], True, ch["ict"], ch["icb"], ch["ns"], ch["na"], ch["nb"])
    g = {"FLXU_INT": t["flxu"], "FLCU_INT": t["flcu"], "FLAU_INT": t["flau"], "FLXAU_INT": t["flxau"], "FLXD_INT": t["flxd"], "FLCD_INT": t["flcd"],
         "FLAD_INT": t["flad"], "FLXAD_INT": t["flxad"], "DFDTS": t["dfdts"], "SFCEM_INT": t["sfcem"]}
    g["TS"] = torch.from_numpy(np.ascontiguousarray(inp["tsfc"], dtype=dt)).cuda()
    sf0 = t["sfcem"].clone()
    for k in ("FLX_INT", "FLXA_INT", "FLC_INT", "FLA_INT", "DFDTSC", "DFDTSNA", "DFDTSCNA"):
        g[k] = torch.full((lm + 1, m), -7.0, dtype=t["ta"].dtype, device="cuda")
    g["TS_INT"] = torch.zeros(m, dtype=t["ta"].dtype, device="cuda")
    ctx.lw_chou_post_dev(st, m, lm, {k: v.data_ptr() for k, v in g.items()})
    ctx.check(st)
    assert torch.equal(g["FLX_INT"], t["flxd"] + t["flxu"]) and torch.equal(g["FLA_INT"], t["flad"] + t["flau"])
    assert torch.equal(g["FLXA_INT"], t["flxad"] + t["flxau"]) and torch.equal(g["FLC_INT"], t["flcd"] + t["flcu"])
    assert torch.equal(g["DFDTSNA"], t["dfdts"]) and not g["DFDTSC"].any() and not g["DFDTSCNA"].any()
    assert torch.equal(g["SFCEM_INT"], -sf0) and (g["SFCEM_INT"] > 0).all() and torch.equal(g["TS_INT"], g["TS"])
    # heartbeat, Chou flavour
    u = {k: g[k] for k in G.LWU_IN if k in g}
    u["FCLD"] = t["fcld"]
    u["TSINST"] = g["TS"] + 0.5
    want = ["FLX", "FLXA", "FLC", "FLA", "FLXU", "FLAU", "FLXAD", "OLR", "OLRA", "OLA", "SFCEM", "LWSA", "FLNSNA", "FLNSA", "CLDTT"]
    tout, pout = _zeros({k: ((lm + 1, m) if k in G.LWU_OUT_3D else (m,)) for k in want}, dt)
    ptr = {k: v.data_ptr() for k, v in u.items()}; ptr.update(pout)
    lmh, llm = int(ch["ict"]), int(ch["icb"])
    ctx.lw_update_flx_dev(st, m, lm, False, lmh, llm, 1e15, ptr)
    ctx.check(st)
    o = clib.lw_update_flx({k: v.cpu().numpy() for k, v in u.items()}, lm, False, lmh, llm, 1e15, prec, want=want)
    for k in want:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)
    # clear sky emits at least as much as all sky (to the 2 W m-2 the scheme itself allows, see tests/test_gpu_chou.py)
    assert (o["OLR"] > 100).all() and (o["OLA"] >= o["OLRA"] - 2.0).all()


@pytest.mark.parametrize("rk", [4, 8])
def test_lw_driver_rats_internals(gpu_ctx, rk):
    """LW_Driver with RATS_DIAGNOSTICS (IRR:3389-3469, :3522-3530, :3614): FLXU_RAT / FLXD_RAT / FLX_RAT / DFDTS_RAT / SFCEM_RAT
    from the driver call = the reference's statements applied to separate solver calls with the gas removed (bitwise)."""
    from oracle import clib
    ctx = gpu_ctx[rk]; dt = ctx.dtype; prec = PREC[rk]
    ncol, lm = 300, 72
    inp = synth.make_columns(ncol, lm, start=4100, cloudy_frac=0.6, aerosol=True)
    f = synth.geos_lw_fields(inp)
    consts = G.lwd_consts()
    gases = ["CO2", "H2O", "CH4"]
    ctx.set_inhomogeneity(1)
    tin, ptr = _dev(f, dt)
    shapes = {k: ((lm + 1, ncol) if k in G.LWD_OUT_3D else ((ncol, 16) if k in ("OLRB", "DOLRB") else (ncol,))) for k in G.LWD_OUT}
    shapes.update({k: (len(gases), lm + 1, ncol) for k in G.LWD_RAT_OUT[:4]})
    shapes["SFCEM_RAT"] = (len(gases), ncol)
    tout, pout = _zeros(shapes, dt)
    ptr.update(pout)
    ctx.lw_driver_rrtmg_rats_dev(_stream(), ncol, lm, 16, ptr, consts, 3, 1, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"], gases)
    ctx.check(_stream())
    rr = clib.lwd_prep(f, consts, 3, 1, prec)
    inp2 = dict(rr); inp2.update(dyofyr=inp["dyofyr"], cloudLM=inp["cloudLM"], cloudMH=inp["cloudMH"])
    h = ctx.rrtmg_lw_columns(inp2, dudTs=True)
    o = clib.lwd_post(h, h["clearCounts"], f["EMIS"], f["TS"], prec)
    for k in G.LWD_OUT[:16]:
        np.testing.assert_array_equal(tout[k].cpu().numpy(), o[k], err_msg=k)      # the main call is untouched by the RATS passes
    one = dt(1.0)
    for r, gas in enumerate(gases):
        z = dict(inp2); key = G.RAT_VMR[gas]
        z[key] = np.zeros_like(z[key])
        s = ctx.rrtmg_lw_columns(z, dudTs=True)
        fu, fd = -s["uflx"][::-1], s["dflx"][::-1]
        np.testing.assert_array_equal(tout["FLXU_RAT"][r].cpu().numpy(), fu, err_msg=gas)
        np.testing.assert_array_equal(tout["FLXD_RAT"][r].cpu().numpy(), fd, err_msg=gas)
        np.testing.assert_array_equal(tout["DFDTS_RAT"][r].cpu().numpy(), -s["duflx_dTs"][::-1], err_msg=gas)
        np.testing.assert_array_equal(tout["FLX_RAT"][r].cpu().numpy(), fd + fu, err_msg=gas)
        sf = s["uflx"][0] - s["dflx"][0] * (one - np.asarray(f["EMIS"], dtype=dt))
        np.testing.assert_array_equal(tout["SFCEM_RAT"][r].cpu().numpy(), sf.astype(dt), err_msg=gas)
        assert (tout["FLX_RAT"][r][0] != tout["FLX_INT"][0]).any()
    ctx.set_inhomogeneity(0)


@pytest.mark.parametrize("rk", [4, 8])
def test_update_flx_rats_exports(gpu_ctx, rk):
    """RATS exports of Update_Flx (IRR:4036-4120) against the reference's statements evaluated with numpy in the same precision, operation
    by operation (numpy does not fuse): bitwise; exports not associated are left alone."""
    import torch
    ctx = gpu_ctx[rk]; dt = ctx.dtype
    n, lm, nr = 300, 72, 3
    rng = np.random.default_rng(11)
    st = {"FLX_INT": rng.uniform(-400, 100, (lm + 1, n)), "SFCEM_INT": rng.uniform(300, 450, n), "DFDTS": rng.uniform(-6, 0, (lm + 1, n)),
          "FLX_RAT": rng.uniform(-400, 100, (nr, lm + 1, n)), "SFCEM_RAT": rng.uniform(300, 450, (nr, n)), "DFDTS_RAT": rng.uniform(-6, 0, (nr, lm + 1, n))}
    st = {k: np.ascontiguousarray(v, dtype=dt) for k, v in st.items()}
    t = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    shp = {"COLTRAP": (nr, lm, n), "FLX": (nr, lm + 1, n), "DFDTS_OUT": (nr, lm + 1, n)}
    want = [k for k in G.LWR_OUT if k != "dFLNS"]              # dFLNS_<gas> not associated
    for k in want:
        t[k] = torch.full(shp.get(k, (nr, n)), 7.0, dtype=t["FLX_INT"].dtype, device="cuda")
    ctx.lw_update_rats_dev(_stream(), n, lm, nr, {k: v.data_ptr() for k, v in t.items()})
    ctx.check(_stream())
    F, FR, S, SR = st["FLX_INT"], st["FLX_RAT"], st["SFCEM_INT"], st["SFCEM_RAT"]
    ref = {}
    a = -(FR[:, 0]); ref["dOLR"] = (-(F[0])) - a
    ref["dLWS"] = (F[lm] + S) - (FR[:, lm] + SR)
    ref["dSFCEM"] = S - SR
    x = F[lm] - F[0]; ref["NETTRAP"] = x - (FR[:, lm] - FR[:, 0])
    y = F[1:] - F[:-1]; ref["COLTRAP"] = y[None] - (FR[:, 1:] - FR[:, :-1])
    ref["FLX"] = F[None] - FR
    ref["DFDTS_OUT"] = st["DFDTS"][None] - st["DFDTS_RAT"]
    for k in want:
        assert ref[k].dtype == dt
        np.testing.assert_array_equal(t[k].cpu().numpy(), ref[k], err_msg=k)


@pytest.mark.parametrize("rk", [4, 8])
def test_update_flx_band_olr_and_brightness_temperature(gpu_ctx, rk):
    """OLRBbbRG / TBRBbbRG of Update_Flx (IRR:3993-4021, Tbr_from_band_flux :4132-4208) against the reference's statements in numpy (same
    kinds, same order): the band flux bitwise; the brightness temperature - a double-precision log - to 1 ulp of the export's kind;
    a band whose flux is zero everywhere gives MAPL_UNDEF; bands not selected are not touched."""
    import torch
    ctx = gpu_ctx[rk]; dt = ctx.dtype
    n = 1000
    rng = np.random.default_rng(5)
    olrb = rng.uniform(0.5, 40.0, (n, 16)).astype(dt); dolrb = rng.uniform(0.0, 0.3, (n, 16)).astype(dt)
    olrb[:, 9] = 0; dolrb[:, 9] = 0                       # band 10: before the first full calculation
    ts = rng.uniform(270, 300, n).astype(dt); tsinst = (ts + rng.uniform(-3, 3, n)).astype(dt)
    bo = np.zeros(16, dtype=np.int32); bo[[0, 5, 9, 15]] = 1
    t = {"TSINST": tsinst, "TS_INT": ts, "OLRB": olrb, "DOLRB": dolrb}
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in t.items()}
    t["OLRB_EXP"] = torch.full((16, n), -5.0, dtype=t["OLRB"].dtype, device="cuda")
    t["TBRB_EXP"] = torch.full((16, n), -5.0, dtype=t["OLRB"].dtype, device="cuda")
    undef = G.MAPL["UNDEF"]
    ctx.lw_update_bands_dev(_stream(), n, bo, undef, {k: v.data_ptr() for k, v in t.items()})
    ctx.check(_stream())
    got_f, got_t = t["OLRB_EXP"].cpu().numpy(), t["TBRB_EXP"].cpu().numpy()
    h, c, kB, pi = 6.626070040e-34, 2.99792458e8, 1.38064852e-23, 3.14159265358979323846
    alT, bigC = h * c / kB, 2.0 * h * (c * c)
    delt = tsinst - ts
    for ib in range(16):
        if not bo[ib]:
            assert (got_f[ib] == -5.0).all() and (got_t[ib] == -5.0).all()
            continue
        f = olrb[:, ib] + dolrb[:, ib] * delt
        assert f.dtype == dt
        np.testing.assert_array_equal(got_f[ib], f)
        if ib == 9:
            assert (got_t[ib] == dt(undef)).all()
            continue
        wn1, wn2 = dt(G.LW_WAVENUM1[ib]) * dt(100.), dt(G.LW_WAVENUM2[ib]) * dt(100.)
        bmean = f.astype(np.float64) / (pi * np.float64(wn2 - wn1))
        wnmid = dt(np.float64(wn1 + wn2) / 2.0)
        wn3 = (wnmid * wnmid) * wnmid
        tbr = (alT * np.float64(wnmid) / np.log(bigC * np.float64(wn3) / bmean + 1.0)).astype(dt)
        np.testing.assert_allclose(got_t[ib], tbr, rtol=float(np.finfo(dt).eps), atol=0)
        assert (tbr > 20).all() and (tbr < 2000).all() and tbr.std() > 1


@pytest.mark.parametrize("rk", [4, 8])
def test_update_export_surface_block(gpu_ctx, rk):
    """2-D block of UPDATE_EXPORT (SOL:7403-7533) against the reference's statements in numpy (same kind, same order): bitwise; night
    columns (SLR = 0) give MAPL_UNDEF albedos and zero surface fluxes; exports not associated stay untouched."""
    import torch
    ctx = gpu_ctx[rk]; dt = ctx.dtype
    n, lm = 3000, 72
    rng = np.random.default_rng(21)
    U = lambda lo, hi, *s: rng.uniform(lo, hi, s).astype(dt)
    st = {"SLR": U(0, 1300, n), "ZTH": U(-0.2, 1.0, n), "ALBVF": U(0.02, 0.5, n), "ALBVR": U(0.02, 0.5, n), "ALBNF": U(0.02, 0.5, n),
          "ALBNR": U(0.02, 0.5, n), "FSWN": U(0, 1, lm + 1, n), "FSCN": U(0, 1, lm + 1, n), "FSWNAN": U(0, 1, lm + 1, n), "FSCNAN": U(0, 1, lm + 1, n)}
    for k in G.SWS_IN[6:12]:
        st[k] = U(0, 0.2, n)
    night = rng.uniform(0, 1, n) < 0.4
    st["SLR"][night] = 0
    st["FSWN"][lm][rng.uniform(0, 1, n) < 0.05] = 2.0          # 1 - FSWN/sum < .01 -> clamped; also exercises the 0.9 cap below
    st["FSWN"][lm][rng.uniform(0, 1, n) < 0.05] = 0.0
    dark = rng.uniform(0, 1, n) < 0.05
    for k in G.SWS_IN[6:12]:
        st[k][dark] = 0                                        # no surface flux at all: ALB undefined
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    want = [k for k in G.SWS_OUT if k not in ("DFPAR", "SLRSUFNA")]
    for k in want:
        t[k] = torch.full((n,), -9.0, dtype=t["SLR"].dtype, device="cuda")
    undef = G.MAPL["UNDEF"]
    ctx.sw_update_surface_dev(_stream(), n, lm, undef, {k: v.data_ptr() for k, v in t.items()})
    ctx.check(_stream())
    slr = st["SLR"]; ud = dt(undef); one = dt(1.0)
    d = [st[k] for k in G.SWS_IN[6:12]]
    sum6 = d[0] + d[1] + d[2] + d[3] + d[4] + d[5]
    ref = {}
    for k in ("ALBVF", "ALBVR", "ALBNF", "ALBNR"):
        ref[k + "_X"] = np.where(slr > 0, st[k] * one, ud)
    ok = (slr > 0) & (sum6 > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = one - st["FSWN"][lm] / sum6
        alb = np.where(ok, np.minimum(np.maximum(x, dt(.01)), dt(0.9)), ud).astype(dt)
        ref["ALBEDO"] = alb
        ref["SLRTP"] = slr
        for k, nm in enumerate(("DRUVR", "DFUVR", "DRPAR", "DFPAR", "DRNIR", "DFNIR")):
            ref[nm] = d[k] * slr
        zth = np.maximum(st["ZTH"], dt(0.0))
        sln = np.where(zth > 0, slr / np.where(zth > 0, zth, one), dt(0.0)).astype(dt)
        ref["DRNUVR"] = d[0] * sln; ref["DRNPAR"] = d[2] * sln; ref["DRNNIR"] = d[4] * sln
        ref["SLRSF"] = sum6 * slr
        defd = alb != ud
        for nm, f in (("SLRSFC", "FSCN"), ("SLRSFNA", "FSWNAN"), ("SLRSFCNA", "FSCNAN")):
            ref[nm] = np.where(defd, (st[f][lm] * slr) / (one - alb), dt(0.0)).astype(dt)
        ref["SLRSUF"] = alb * sum6 * slr
        for nm, f in (("SLRSUFC", "FSCN"), ("SLRSUFNA", "FSWNAN"), ("SLRSUFCNA", "FSCNAN")):
            ref[nm] = np.where(defd, alb * (st[f][lm] / (one - alb)) * slr, dt(0.0)).astype(dt)
    for k in want:
        assert ref[k].dtype == dt, k
        np.testing.assert_array_equal(t[k].cpu().numpy(), ref[k], err_msg=k)
    assert (ref["ALBEDO"] == ud).sum() > n // 3 and (ref["ALBEDO"] == dt(.01)).any() and (ref["ALBEDO"] == dt(0.9)).any()


@pytest.mark.parametrize("rk", [4, 8])
def test_lit_column_compaction_roundtrip_and_sw_on_packed_columns(gpu_ctx, rk):
    """SW first compacts the lit columns (`daytime = ZTH > 0.`, PackIt, GEOS_SolarGridComp.F90:3686,7753-7773), runs the solver on
    the packed batch and scatters the results back (UnPackIt :7776-7799).  Pack / unpack against numpy, and RRTMG_SW on the packed
    columns + unpack equal to the same columns of an unpacked call, bit for bit; dark columns receive UnPackIt's DEFAULT."""
    import torch
    from geosradiation_gridcomp_amd import synth
    ctx = gpu_ctx[rk]
    tdt = torch.float32 if rk == 4 else torch.float64
    n, nlay = 1500, 72
    inp = synth.make_columns(n, nlay, start=12_000, aerosol=True, cloudy_frac=0.5)
    rng = np.random.default_rng(3)
    day = rng.uniform(size=n) < 0.47
    zth = np.where(day, inp["coszen"], -rng.uniform(0.0, 1.0, n)).astype(ctx.dtype)      # night: cos(zenith) <= 0
    zth[5] = 0.0
    day[5] = False
    st = torch.cuda.current_stream().cuda_stream
    t_zth = torch.from_numpy(zth).cuda()
    idx = torch.zeros(n, dtype=torch.int32, device="cuda"); pos = torch.zeros(n, dtype=torch.int32, device="cuda")
    nl = torch.zeros(1, dtype=torch.int32, device="cuda")
    nlit = ctx.lit_index_dev(st, n, t_zth.data_ptr(), idx.data_ptr(), pos.data_ptr(), nl.data_ptr())
    want_idx = np.flatnonzero(day)
    assert nlit == want_idx.size == int(nl.item())
    np.testing.assert_array_equal(idx.cpu().numpy()[:nlit], want_idx)
    want_pos = np.full(n, -1, dtype=np.int32); want_pos[want_idx] = np.arange(nlit)
    np.testing.assert_array_equal(pos.cpu().numpy(), want_pos)
    pdim = nlit + 9                                       # NumMax may exceed NumLit (the balanced buffer is larger)

    def pack(a):      # numpy (nlev, n) -> device (nlev, pdim)
        a = np.ascontiguousarray(a, dtype=ctx.dtype)
        a2 = a.reshape(-1, n)
        u = torch.from_numpy(a2).cuda()
        p = torch.full((a2.shape[0], pdim), -5.0, dtype=tdt, device="cuda")
        ctx.lit_pack_dev(st, pdim, n, a2.shape[0], idx.data_ptr(), nl.data_ptr(), u.data_ptr(), p.data_ptr())
        np.testing.assert_array_equal(p.cpu().numpy()[:, :nlit], a2[:, want_idx])
        assert (p.cpu().numpy()[:, nlit:] == -5.0).all()
        return p
    names = ["play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel", "zm", "alat",
             "tauaer_sw", "ssaaer_sw", "asmaer_sw", "coszen", "asdir", "asdif", "aldir", "aldif"]
    d = {k: pack(inp[k]) for k in names}
    for k in ("swuflx", "swdflx", "swuflxc", "swdflxc"):
        d[k] = torch.zeros((nlay + 1, pdim), dtype=tdt, device="cuda")
    for k in ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf", "cotdtp", "cotdhp", "cotdmp", "cotdlp", "cotntp", "cotnhp", "cotnmp", "cotnlp"):
        d[k] = torch.zeros(pdim, dtype=tdt, device="cuda")
    d["fswband"] = torch.zeros((14, pdim), dtype=tdt, device="cuda")
    d["clearCounts_sw"] = torch.zeros((4, pdim), dtype=torch.int32, device="cuda")
    ctx.set_inhomogeneity(1)
    try:
        # the packed arrays have leading dimension pdim: the solver is called on nlit columns of a (pdim, .) batch -> pass pdim-wide arrays
        # through a contiguous nlit-wide view (what SORADCORE's Num2do-wide dummies are)
        dn = {k: v[..., :nlit].contiguous() for k, v in d.items()}
        ctx.rrtmg_sw_dev(st, nlit, nlay, 1361.0, 1.0, 0, {k: v.data_ptr() for k, v in dn.items()}, 3, 1, int(inp["dyofyr"]), 10,
                         int(inp["cloudLM"]), int(inp["cloudMH"]), normFlx=1)
        ctx.check(st)
        full = ctx.rrtmg_sw_columns(inp, iaer=10, normFlx=1)
    finally:
        ctx.set_inhomogeneity(0)
    for k in ("swuflx", "swdflx", "swuflxc", "swdflxc", "fswband", "nirr", "parf"):
        src = dn[k].reshape(-1, nlit).contiguous()
        nlev = src.shape[0]
        out = torch.full((nlev, n), 123.0, dtype=tdt, device="cuda")
        ctx.lit_unpack_dev(st, nlit, n, nlev, pos.data_ptr(), src.data_ptr(), out.data_ptr(), default=0.0)
        o = out.cpu().numpy().reshape(full[k].shape)
        np.testing.assert_array_equal(o[..., day], full[k][..., day], err_msg=k)       # bitwise: a column never depends on its batch
        assert (o[..., ~day] == 0.0).all()
        keep = torch.full((nlev, n), 123.0, dtype=tdt, device="cuda")
        ctx.lit_unpack_dev(st, nlit, n, nlev, pos.data_ptr(), src.data_ptr(), keep.data_ptr())     # no DEFAULT: dark columns untouched
        assert (keep.cpu().numpy().reshape(full[k].shape)[..., ~day] == 123.0).all()


@pytest.mark.parametrize("rk", [4, 8])
@pytest.mark.parametrize("aer", [True, False])
def test_chou_branch_of_soradcore(gpu_ctx, rk, aer):
    """geosrad_sw_driver_chou_dev on GEOS fields (model ordering, Pa, odd oxygen as volume mixing ratio, radii in metres with MAPL_UNDEF
    cells, top layers above 1 hPa) = the oracle's statement of SORADCORE's preparation (SOL:4484-4528) followed by the oracle's sorad;
    with no aerosol arrays the driver supplies the zeroes (SOL:4543-4546).  Also: the driver = geosrad_sorad_dev on the oracle-prepared
    arrays (fp64: bitwise wherever the prepared ozone is)."""
    import torch
    from oracle import clib
    ctx = gpu_ctx[rk]; prec = PREC[rk]; dt = np.float32 if rk == 4 else np.float64
    inp = synth.make_columns(300, 72, start=8080, cloudy_frac=0.6, aerosol=True)
    f = synth.geos_chou_sw_fields(inp, aerosol=aer)
    assert (f["PLE"][:3] < 100.0).all()                       # the odd-oxygen scaling above 1 hPa is exercised
    lm, m = f["T"].shape
    consts = G.swc_consts(co2=f["CO2"])
    t, ptr = _dev({k: f[k] for k in G.SWC_IN if k in f}, dt)
    shapes = {k: (lm + 1, m) for k in ("FSW", "FSC", "FSWU", "FSCU")}
    shapes.update({k: (m,) for k in ("NIRR", "NIRF", "PARR", "PARF", "UVRR", "UVRF")})
    shapes.update({k: (8, m) for k in ("FSWBAND", "DRBAND", "DFBAND")})
    tout, pout = _zeros(shapes, dt)
    ptr.update(pout)
    st = _stream()
    ctx.sw_driver_chou_dev(st, m, lm, ptr, consts, f["LCLDMH"], f["LCLDLM"], f["HK_UV"], f["HK_IR"], do_drfband=True)
    ctx.check(st)
    g = {k: v.cpu().numpy() for k, v in tout.items()}
    # the oracle's way
    fr = {k: np.asarray(v, dtype=dt) for k, v in f.items() if isinstance(v, np.ndarray)}       # what the device was given
    pr = clib.swc_prep(fr, (G.MAPL["O3MW"], G.MAPL["AIRMW"], G.MAPL["UNDEF"]), prec)
    z = np.zeros((8, lm, m), dtype=dt)
    cs = dict(cosz=fr["ZT"], pl=pr["PLhPa"], ta=fr["T"], wa=fr["Q"], oa=pr["O3"], cwc=pr["QQ3"], fcld=fr["CL"], reff=pr["RR3"],
              hk_uv=f["HK_UV"], hk_ir=f["HK_IR"], taua=fr["TAUA"] if aer else z, ssaa=fr["SSAA"] if aer else z, asya=fr["ASYA"] if aer else z,
              rsuvbm=fr["ALBVR"], rsuvdf=fr["ALBVF"], rsirbm=fr["ALBNR"], rsirdf=fr["ALBNF"], co2=f["CO2"], ict=f["LCLDMH"], icb=f["LCLDLM"])
    o = clib.sorad(cs, prec, do_drfband=True)
    assert o["rc"] == 0
    names = dict(FSW="flx", FSC="flc", FSWU="flxu", FSCU="flcu", NIRR="fdirir", NIRF="fdifir", PARR="fdirpar", PARF="fdifpar", UVRR="fdiruv",
                 UVRF="fdifuv", FSWBAND="flx_sfc_band", DRBAND="drband", DFBAND="dfband")
    tol = 1e-9 if rk == 8 else 2e-5
    for k, ok in names.items():
        assert np.isfinite(g[k]).all(), k
        assert np.abs(g[k].astype(np.float64) - o[ok].astype(np.float64)).max() <= tol, k
    assert (g["FSW"][0] > 0.3).all() and (g["FSWU"][0] > 0).all()
    # the same solver on the oracle-prepared arrays
    sd, sp = _dev({k: v for k, v in cs.items() if isinstance(v, np.ndarray) and k not in ("hk_uv", "hk_ir")}, dt)
    so, sop = _zeros({"flx": (lm + 1, m), "flc": (lm + 1, m), "flxu": (lm + 1, m), "flcu": (lm + 1, m), "fdiruv": (m,), "fdifuv": (m,),
                      "fdirpar": (m,), "fdifpar": (m,), "fdirir": (m,), "fdifir": (m,), "flx_sfc_band": (8, m), "drband": (8, m),
                      "dfband": (8, m)}, dt)
    sp.update(sop)
    ctx.sorad_dev(st, m, lm, 8, sp, f["CO2"], f["LCLDMH"], f["LCLDLM"], f["HK_UV"], f["HK_IR"], do_drfband=True)
    torch.cuda.synchronize()
    for k, ok in names.items():
        a, b = g[k], so[ok].cpu().numpy()
        if rk == 8:
            assert np.abs(a - b).max() <= 1e-13, k
        else:
            assert np.abs(a.astype(np.float64) - b).max() <= 2e-6, k
    # argument errors
    bad = dict(ptr); bad.pop("OX")
    with pytest.raises(GeosradError):
        ctx.sw_driver_chou_dev(st, m, lm, bad, consts, f["LCLDMH"], f["LCLDLM"], f["HK_UV"], f["HK_IR"])
    if aer:
        bad = dict(ptr); bad.pop("SSAA")
        with pytest.raises(GeosradError):
            ctx.sw_driver_chou_dev(st, m, lm, bad, consts, f["LCLDMH"], f["LCLDLM"], f["HK_UV"], f["HK_IR"])
