"""GPU: the Fortran drop-in boundary (LW and SW).  lw_driver.F90 calls `rrtmg_lw_ini` / `rrtmg_lw` / `set_inhomogeneity` with
the reference's own module names and signatures (as GEOS_IrradGridComp's LW_Driver does) but is linked against
the ISO_C_BINDING shim modules over libgeosrad.so; its fluxes must match the reference's golden vectors."""
import os
import subprocess
import numpy as np
import pytest
from tests.conftest import ROOT, load_golden, FLUX

pytestmark = pytest.mark.gpu
FDIR = os.path.join(ROOT, "geosradiation_gridcomp_amd", "fortran")
ORDER = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr",
         "cfc12vmr", "cfc22vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel", "tauaer", "zm", "alat"]


@pytest.mark.parametrize("name", ["lw_aer_72", "lw_cloudy_ih1_72"])
@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_caller_gets_reference_fluxes(tmp_path, name, kind):
    exe = os.path.join(FDIR, "bin", f"lw_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    inp, g, ih = load_golden(name)
    nlay, ncol = inp["play"].shape
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([ncol, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])], dtype=np.int32).tofile(f)
        for k in ORDER:
            np.ascontiguousarray(inp[k], dtype=np.float32).tofile(f)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    subprocess.check_call([exe, str(fin), str(fout)], env=env)
    raw = np.fromfile(fout, dtype=np.uint8)
    nflux = 6 * (nlay + 1) * ncol
    flux = raw[: nflux * 8].view(np.float64).reshape(6, nlay + 1, ncol)
    cc = raw[nflux * 8:].view(np.int32).reshape(4, ncol)
    for i, k in enumerate(FLUX):
        tol = {"r8": 1e-8, "r4": 2e-5}[kind] if "dTs" in k else {"r8": 1e-6, "r4": 2e-3}[kind]
        assert np.abs(flux[i] - g[f"{kind}_{k}"].astype(np.float64)).max() <= tol, k
    assert np.abs(cc - g[f"{kind}_clearCounts"]).max() <= (0 if kind == "r8" else 1)


SW_ORDER = ["coszen", "play", "plev", "tlay", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel",
            "zm", "alat", "tauaer_sw", "ssaaer_sw", "asmaer_sw", "asdir", "asdif", "aldir", "aldif"]


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_sw_caller_matches_oracle(tmp_path, kind):
    """sw_driver.F90 calls rrtmg_sw_ini / rrtmg_sw(MAPL, rpart, ncol, nlay, scon, ...) with the reference's module names,
    argument order and RC convention (as GEOS_SolarGridComp.F90:6225,6331 does), linked against the shim modules."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    exe = os.path.join(FDIR, "bin", f"sw_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    ncol, nlay, ih = 40, 72, 1
    inp = synth.make_columns(ncol, nlay, start=606, aerosol=True, cloudy_frac=0.6)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"

    def run(isolvar, scon, solcycfrac=None):
        with open(fin, "wb") as f:
            np.array([ncol, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]), 10, 1, isolvar], dtype=np.int32).tofile(f)
            np.array([scon], dtype=np.float32).tofile(f)
            for k in SW_ORDER:
                np.ascontiguousarray(inp[k], dtype=np.float32).tofile(f)
        env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
        subprocess.check_call([exe, str(fin), str(fout)] + ([] if solcycfrac is None else ["0", repr(solcycfrac)]), env=env)
        raw = np.fromfile(fout, dtype=np.uint8)
        rc = int(raw[:4].view(np.int32)[0])
        return rc, raw[4:]

    rc, raw = run(0, 1361.0)
    assert rc == 0
    nl = (nlay + 1) * ncol
    off = 0

    def take(n, shape):
        nonlocal off
        a = raw[off: off + n * 8].view(np.float64).reshape(shape); off += n * 8
        return a
    got = {k: take(nl, (nlay + 1, ncol)) for k in ("swuflx", "swdflx", "swuflxc", "swdflxc")}
    got["nirr"] = take(ncol, (ncol,)); got["parf"] = take(ncol, (ncol,))
    for k in ("fswband", "drband", "dfband"):
        got[k] = take(14 * ncol, (14, ncol))
    cot0 = take(ncol, (ncol,))
    cc = raw[off:].view(np.int32).reshape(4, ncol)
    clib.set_inhomogeneity(ih, kind)
    o = clib.rrtmg_sw(inp, prec=kind, iaer=10, normFlx=1, do_drfband=True)
    clib.set_inhomogeneity(0, kind)
    same = (cc == o["clearCounts"]).all(axis=0)
    assert same.all() if kind == "r8" else same.mean() > 0.9
    tol = 1e-9 if kind == "r8" else 5e-3          # normalised fluxes (TOA down = 1); fp32: see tests/test_gpu_sw.py
    for k, v in got.items():
        assert np.abs(v - o[k].astype(np.float64))[..., same].max() <= tol, k
    assert np.abs(cot0 - o["cot"][0])[same].max() <= (1e-9 if kind == "r8" else 1e-4) * max(1.0, o["cot"][0].max())
    # RC convention: the reference's _FAIL paths return a non-zero RC instead of stopping: isolvar 1 without SOLCYCFRAC is one
    # (rrtmg_sw_rad.F90:909-911)
    rc, _ = run(1, 1361.0)
    assert rc != 0
    # isolvar = 1 with the optional INDSOLVAR / SOLCYCFRAC arguments by keyword
    rc, raw = run(1, 1361.0, solcycfrac=0.3)
    assert rc == 0
    off = 0
    got1 = {k: take(nl, (nlay + 1, ncol)) for k in ("swuflx", "swdflx")}
    clib.set_inhomogeneity(ih, kind)
    o1 = clib.rrtmg_sw(inp, prec=kind, iaer=10, normFlx=1, do_drfband=True, isolvar=1, solcycfrac=0.3, indsolvar=(1.15, 0.9))
    clib.set_inhomogeneity(0, kind)
    assert o1["rc"] == 0
    for k, v in got1.items():
        assert np.abs(v - o1[k].astype(np.float64))[..., same].max() <= tol, k


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_gridcomp_use_lists_and_pointer_dummies(tmp_path, kind):
    """uselist_driver.F90 carries the `use ... only:` lines of GEOS_IrradGridComp.F90:67-75,1472-1474, GEOS_SolarGridComp.F90:175-183,
    3346-3348,6678-6679 and GEOS_RadiationGridComp.F90:481-484 and is compiled against the shim modules; it calls rrtmg_sw once with
    DRBAND / DFBAND disassociated (do_drfband false, the default GEOS run) and once with them pointing at the non-contiguous
    section ptr2(1:ncol,:) of a larger array (SOLAR_TO_OBIO, GEOS_SolarGridComp.F90:4148-4151)."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    exe = os.path.join(FDIR, "bin", f"uselist_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    ncol, nlay, ih = 40, 72, 1
    inp = synth.make_columns(ncol, nlay, start=909, aerosol=True, cloudy_frac=0.6)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([ncol, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]), 10, 1, 0], dtype=np.int32).tofile(f)
        np.array([1361.0], dtype=np.float32).tofile(f)
        for k in SW_ORDER:
            np.ascontiguousarray(inp[k], dtype=np.float32).tofile(f)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    subprocess.check_call([exe, str(fin), str(fout)], env=env)
    raw = np.fromfile(fout, dtype=np.uint8)
    nout = int(raw[:4].view(np.int32)[0])
    v = raw[4: 4 + 8 * nout].view(np.float64)
    off = 0

    def take(n, shape=None):
        nonlocal off
        a = v[off: off + n]; off += n
        return a.reshape(shape) if shape else a
    adl, rdl, z = take(ncol), take(ncol), take(3)
    assert list(take(4)) == [140.0, 16.0, 10.0, 3250.0]                     # ngptlw, nbndlw, wavenum1(1), wavenum2(16)
    nl = (nlay + 1) * ncol
    npad = ncol + 7
    rc1, u1, d1 = take(1), take(nl, (nlay + 1, ncol)), take(nl, (nlay + 1, ncol))
    rc2, u2, d2 = take(1), take(nl, (nlay + 1, ncol)), take(nl, (nlay + 1, ncol))
    dr, df = take(14 * npad, (14, npad)), take(14 * npad, (14, npad))
    assert list(take(5)) == [112.0, 16.0, 29.0, 2600.0, 2600.0]             # ngptsw, jpb1, jpb2, wavenum1(16), wavenum2(29)
    hk = take(2)
    assert off == nout and rc1[0] == 0 and rc2[0] == 0
    assert (hk > 0).all() or (hk == 0).all()                                # sorad_constants compiled from the reference tree, or absent
    # the two calls differ only in do_drfband: identical fluxes
    np.testing.assert_array_equal(u1, u2); np.testing.assert_array_equal(d1, d2)
    # rows beyond the section are the parent array's own: untouched
    assert (dr[:, ncol:] == -777.0).all() and (df[:, ncol:] == -777.0).all()
    clib.set_inhomogeneity(ih, kind)
    o = clib.rrtmg_sw(inp, prec=kind, iaer=10, normFlx=1, do_drfband=True)
    clib.set_inhomogeneity(0, kind)
    tol = 1e-9 if kind == "r8" else 5e-3
    cc_same = np.abs(u1 - o["swuflx"].astype(np.float64)).max(axis=0) <= tol      # fp32: a McICA decision may flip in a few columns
    assert cc_same.all() if kind == "r8" else cc_same.mean() > 0.9
    for got, k in ((u1, "swuflx"), (d1, "swdflx"), (dr[:, :ncol], "drband"), (df[:, :ncol], "dfband")):
        assert np.abs(got - o[k].astype(np.float64))[..., cc_same].max() <= tol, k
    # host functions of the RRTMGP branch, through the same shim modules (bitwise check against the reference: tests/test_host.py)
    dt = np.float32 if kind == "r4" else np.float64
    lat = np.asarray(inp["alat"], dtype=np.float32).astype(np.float64) * (180.0 / np.pi)
    doy = int(inp["dyofyr"])
    for got, (a1, a2, a30, a4) in ((adl, (1.4315, 2.1219, 7.0, -25.584)), (rdl, (0.72192, 0.78996, 8.5, 40.404))):
        a3 = -4.0 * a30 / 365.0 * (doy - 272) if doy > 181 else 4.0 * a30 / 365.0 * (doy - 91)
        want = (a1 + a2 * np.exp(-(lat - a3) ** 2 / a4 ** 2)) * 1.0e3
        np.testing.assert_allclose(got, want, rtol=2e-5 if kind == "r4" else 1e-12)
    assert z[2] == 1.0 and 0 < z[0] < z[1]


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_chou_caller_matches_oracle(tmp_path, kind):
    """chou_driver.F90 calls irrad(...) and sorad(...) with the reference's module names and argument lists
    (GEOS_IrradGridComp.F90:2093-2101), linked against chou_shims.F90."""
    from geosradiation_gridcomp_amd import synth
    from oracle import clib
    exe = os.path.join(FDIR, "bin", f"chou_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    m, nlay = 24, 72
    inp = synth.make_columns(m, nlay, start=808, aerosol=True, cloudy_frac=0.6)
    ch = synth.chou_lw_inputs(inp, aerosol=True)
    cs = synth.chou_sw_inputs(inp, aerosol=True)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([m, nlay, ch["ict"], ch["icb"], ch["na"]], dtype=np.int32).tofile(f)
        np.array([ch["co2"]], dtype=np.float32).tofile(f)
        for k in ("ple", "ta", "wa", "oa", "tb", "n2o", "ch4", "cfc11", "cfc12", "cfc22", "cwc", "fcld", "reff", "fs", "tg", "eg", "tv", "ev", "rv",
                  "taua", "ssaa", "asya"):
            np.ascontiguousarray(ch[k], dtype=np.float32).tofile(f)
        for k in ("cosz", "pl", "taua", "ssaa", "asya", "rsuvbm", "rsuvdf", "rsirbm", "rsirdf"):
            np.ascontiguousarray(cs[k], dtype=np.float32).tofile(f)
        np.concatenate([cs["hk_uv"].ravel(), cs["hk_ir"].ravel()]).astype(np.float32).tofile(f)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    subprocess.check_call([exe, str(fin), str(fout)], env=env)
    raw = np.fromfile(fout, dtype=np.float64)
    n1 = (nlay + 1) * m
    off = 0

    def take(n, shape):
        nonlocal off
        a = raw[off: off + n].reshape(shape); off += n
        return a
    got = {k: take(n1, (nlay + 1, m)) for k in ("flxu", "flxd", "flcu", "dfdts")}
    got["sfcem"] = take(m, (m,))
    sgot = {k: take(n1, (nlay + 1, m)) for k in ("flx", "flc", "flxu")}
    sgot["fdirpar"] = take(m, (m,)); sgot["flx_sfc_band"] = take(8 * m, (8, m)); sgot["drband"] = take(8 * m, (8, m))
    # the driver reads float32 inputs: feed the oracle the same rounded values
    ch32 = {k: (np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) else v) for k, v in ch.items()}
    ch32["co2"] = float(np.float32(ch["co2"]))
    cs32 = {k: (np.asarray(v, dtype=np.float32) if isinstance(v, np.ndarray) else v) for k, v in cs.items()}
    cs32["co2"] = ch32["co2"]
    o = clib.irrad(ch32, kind)
    s = clib.sorad(cs32, kind)
    tol = 1e-6 if kind == "r8" else 2e-2
    for k, v in got.items():
        assert np.abs(v - o[k].astype(np.float64)).max() <= (tol if k != "dfdts" else tol * 1e-2), k
    tol = 1e-9 if kind == "r8" else 2e-5
    for k, v in sgot.items():
        assert np.abs(v - s[k].astype(np.float64)).max() <= tol, k


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_gridcomp_path_on_device_fields(tmp_path, kind, gpu_ctx):
    """gridcomp_driver.F90: LW_Driver's RRTMG branch + one heartbeat Update_Flx called from Fortran on device-resident GEOS fields
    (module geosrad_gridcomp); same library, same inputs -> the same bits as the Python mirror of the entry points."""
    import torch
    from geosradiation_gridcomp_amd import gridcomp as G
    from geosradiation_gridcomp_amd import synth
    exe = os.path.join(FDIR, "bin", f"gridcomp_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    ncol, lm, nb, ih = 48, 72, 16, 1
    inp = synth.make_columns(ncol, lm, start=2024, aerosol=True, cloudy_frac=0.6)
    f = synth.geos_lw_fields(inp)
    f32 = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in f.items() if isinstance(v, np.ndarray)}
    consts = G.lwd_consts()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as fh:
        np.array([ncol, lm, nb, ih, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"]], dtype=np.int32).tofile(fh)
        np.array(consts, dtype=np.float64).tofile(fh)
        for k in G.LWD_IN:
            if k != "CO2_3D":
                f32[k].tofile(fh)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    subprocess.check_call([exe, str(fin), str(fout)], env=env)
    raw = np.fromfile(fout, dtype=np.float64)
    n3p = (lm + 1) * ncol
    parts = np.split(raw, np.cumsum([n3p, n3p, ncol, ncol, n3p, ncol, ncol, ncol, 2 * n3p]))
    got = dict(zip(["FLX_INT", "DFDTS", "SFCEM_INT", "CLDTTLW", "FLX", "OLR", "FLNS", "SFCEM", "FLX_RAT", "SFCEM_RAT"], parts))
    # the same two calls through the Python mirror (inputs rounded to float32 first, as the file holds them)
    ctx = gpu_ctx[4 if kind == "r4" else 8]
    dt = ctx.dtype
    tdt = torch.float32 if kind == "r4" else torch.float64
    st = torch.cuda.current_stream().cuda_stream
    t = {k: torch.from_numpy(v.astype(dt)).cuda() for k, v in f32.items()}
    for k in G.LWD_OUT[:16]:
        t[k] = torch.zeros((lm + 1, ncol) if k in G.LWD_OUT_3D else (ncol,), dtype=tdt, device="cuda")
    ptr = {k: v.data_ptr() for k, v in t.items()}
    ctx.set_inhomogeneity(ih)
    t["FLX_RAT"] = torch.zeros((2, lm + 1, ncol), dtype=tdt, device="cuda"); t["SFCEM_RAT"] = torch.zeros((2, ncol), dtype=tdt, device="cuda")
    ptr = {k: v.data_ptr() for k, v in t.items()}
    ctx.lw_driver_rrtmg_rats_dev(st, ncol, lm, nb, ptr, consts, 3, 1, int(inp["dyofyr"]), f["LCLDLM"], f["LCLDMH"], ["CO2", "H2O"])
    u = {k: t[k] for k in ("TS_INT", "SFCEM_INT", "FCLD", "FLX_INT", "FLC_INT", "FLXU_INT", "FLCU_INT", "FLXD_INT", "FLCD_INT", "DFDTS", "DFDTSC")}
    u["TSINST"] = t["TS"] + 1.0
    for k in ("FLX", "OLR", "FLNS", "SFCEM"):
        u[k] = torch.zeros((lm + 1, ncol) if k == "FLX" else (ncol,), dtype=tdt, device="cuda")
    ctx.lw_update_flx_dev(st, ncol, lm, True, f["LCLDMH"], f["LCLDLM"], 1.0e15, {k: v.data_ptr() for k, v in u.items()})
    ctx.check(st)
    ctx.set_inhomogeneity(0)
    for k in ("FLX_INT", "DFDTS", "SFCEM_INT", "CLDTTLW", "FLX_RAT", "SFCEM_RAT"):
        np.testing.assert_array_equal(got[k], t[k].cpu().numpy().astype(np.float64).ravel(), err_msg=k)
    # without CO2 / without water vapour more leaves at the top (FLX is net downward: more negative)
    assert (got["FLX_RAT"].reshape(2, lm + 1, ncol)[:, 0] < got["FLX_INT"].reshape(lm + 1, ncol)[0]).all()
    for k in ("FLX", "OLR", "FLNS", "SFCEM"):
        np.testing.assert_array_equal(got[k], u[k].cpu().numpy().astype(np.float64).ravel(), err_msg=k)
    assert (got["OLR"] > 100).all() and (got["SFCEM"] > got["SFCEM_INT"]).all()      # a warmer surface emits more


@pytest.mark.parametrize("kind", ["r8", "r4"])
def test_fortran_chou_branch_of_soradcore_on_device_fields(tmp_path, kind, gpu_ctx):
    """swchou_driver.F90: `call sw_driver_chou` (module geosrad_gridcomp) on device-resident GEOS fields - the Chou-Suarez branch of SORADCORE,
    GEOS_SolarGridComp.F90:4484-4572 / SHRTWAVE :6597-6672; same library, same inputs -> the same bits as the Python mirror of the entry point."""
    import torch
    from geosradiation_gridcomp_amd import gridcomp as G
    from geosradiation_gridcomp_amd import synth
    exe = os.path.join(FDIR, "bin", f"swchou_driver_{kind}")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", FDIR])
    ncol, lm = 70, 72
    inp = synth.make_columns(ncol, lm, start=909, aerosol=True, cloudy_frac=0.6)
    f = synth.geos_chou_sw_fields(inp, aerosol=True)
    f32 = {k: np.ascontiguousarray(f[k], dtype=np.float32) for k in G.SWC_IN}
    # (the file holds float32: MAPL_UNDEF as the float32 the fields carry, so that the real(8) build recognises it too)
    consts = G.swc_consts(co2=f["CO2"], UNDEF=float(np.float32(G.MAPL["UNDEF"])))
    hk = np.concatenate([np.asarray(f["HK_UV"], dtype=np.float32).ravel(), np.asarray(f["HK_IR"], dtype=np.float32).ravel()])
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as fh:
        np.array([ncol, lm, f["LCLDMH"], f["LCLDLM"]], dtype=np.int32).tofile(fh)
        np.array(consts, dtype=np.float64).tofile(fh)
        for k in G.SWC_IN:
            f32[k].tofile(fh)
        hk.tofile(fh)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    subprocess.check_call([exe, str(fin), str(fout)], env=env)
    raw = np.fromfile(fout, dtype=np.float64)
    n3p = (lm + 1) * ncol
    got = dict(zip(["FSW", "FSWU", "NIRR", "FSWBAND", "DRBAND"], np.split(raw, np.cumsum([n3p, n3p, ncol, 8 * ncol]))))
    ctx = gpu_ctx[4 if kind == "r4" else 8]
    dt = ctx.dtype
    tdt = torch.float32 if kind == "r4" else torch.float64
    st = torch.cuda.current_stream().cuda_stream
    t = {k: torch.from_numpy(v.astype(dt)).cuda() for k, v in f32.items()}
    for k in G.SWC_OUT:
        shp = (lm + 1, ncol) if k in ("FSW", "FSC", "FSWU", "FSCU") else ((8, ncol) if k in ("FSWBAND", "DRBAND", "DFBAND") else (ncol,))
        t[k] = torch.zeros(shp, dtype=tdt, device="cuda")
    ctx.sw_driver_chou_dev(st, ncol, lm, {k: v.data_ptr() for k, v in t.items()}, consts, f["LCLDMH"], f["LCLDLM"], hk[:5], hk[5:], do_drfband=True)
    ctx.check(st)
    for k, v in got.items():
        np.testing.assert_array_equal(v, t[k].cpu().numpy().astype(np.float64).ravel(), err_msg=k)
    assert (got["FSW"].reshape(lm + 1, ncol)[0] > 0.3).all() and (got["DRBAND"] >= 0).all()


def test_fortran_dropin_rate_at_full_size(tmp_path, capsys, gpu_ctx):
    """The drop-in path as a GEOS maintainer gets it: Fortran callers (lw_driver / sw_driver, default real, host arrays) on a C360 tile's
    per-GPU share, 97 200 columns x 72 layers, cloudy with aerosols, three calls each - the caller-side time of one call includes
    the transfers both ways.  Prints the times (INTEGRATION.md quotes them; no wall-clock bound is asserted: the box is shared) and
    holds every flux profile of a 700-column slice, which straddles two 16 384-column chunks of the host pipeline, to the bits the
    Python binding returns for those columns alone."""
    import re
    from geosradiation_gridcomp_amd import synth
    ncol, nlay, ih = 97_200, 72, 1
    inp = synth.make_columns(ncol, nlay, start=0, aerosol=True, cloudy_frac=0.6)
    env = dict(os.environ, GEOSRAD_DATA=os.path.join(ROOT, "geosradiation_gridcomp_amd", "data"))
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([ncol, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"])], dtype=np.int32).tofile(f)
        for k in ORDER:
            np.ascontiguousarray(inp[k], dtype=np.float32).tofile(f)
    out = subprocess.run([os.path.join(FDIR, "bin", "lw_driver_r4"), str(fin), str(fout), "3"], env=env, check=True, capture_output=True, text=True).stdout
    lw_ms = float(re.search(r"ms per call\s+([0-9.]+)", out).group(1))
    flux = np.fromfile(fout, dtype=np.float64, count=6 * (nlay + 1) * ncol).reshape(6, nlay + 1, ncol)
    assert np.isfinite(flux).all() and (flux[0, 0] > 100).all()          # upward flux at the surface
    sl = slice(16_384 - 350, 16_384 + 350)
    shard = {k: (np.ascontiguousarray(v[..., sl]) if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol else v) for k, v in inp.items()}
    ctx = gpu_ctx[4]
    ctx.set_inhomogeneity(ih)
    try:
        pl = ctx.rrtmg_lw_columns(shard)
        ps = ctx.rrtmg_sw_columns(shard, iaer=10, normFlx=1, do_drfband=True)
    finally:
        ctx.set_inhomogeneity(0)
    for i, k in enumerate(FLUX):
        np.testing.assert_array_equal(flux[i][:, sl], pl[k].astype(np.float64), err_msg=k)
    with open(fin, "wb") as f:
        np.array([ncol, nlay, ih, int(inp["dyofyr"]), int(inp["cloudLM"]), int(inp["cloudMH"]), 10, 1, 0], dtype=np.int32).tofile(f)
        np.array([1361.0], dtype=np.float32).tofile(f)
        for k in SW_ORDER:
            np.ascontiguousarray(inp[k], dtype=np.float32).tofile(f)
    out = subprocess.run([os.path.join(FDIR, "bin", "sw_driver_r4"), str(fin), str(fout), "3"], env=env, check=True, capture_output=True, text=True).stdout
    sw_ms = float(re.search(r"ms per call\s+([0-9.]+)", out).group(1))
    raw = np.fromfile(fout, dtype=np.uint8)
    assert int(raw[:4].view(np.int32)[0]) == 0
    swf = raw[4: 4 + 4 * (nlay + 1) * ncol * 8].view(np.float64).reshape(4, nlay + 1, ncol)
    for i, k in enumerate(("swuflx", "swdflx", "swuflxc", "swdflxc")):
        np.testing.assert_array_equal(swf[i][:, sl], ps[k].astype(np.float64), err_msg=k)
    assert np.isfinite(swf).all() and np.abs(swf[1, nlay] - 1.0).max() <= 2e-6          # normalised: TOA down = 1
    with capsys.disabled():
        print(f"\nFortran drop-in, {ncol} columns x {nlay} layers, host arrays: rrtmg_lw {lw_ms:.1f} ms, rrtmg_sw {sw_ms:.1f} ms per call "
              f"= {ncol / (lw_ms + sw_ms) * 1e3:.3g} columns/s for the pair")
    assert lw_ms > 0 and sw_ms > 0


def test_ranks_sharing_one_gpu(capsys):
    """bench.py --ranks-per-gpu: K Fortran caller processes (the drop-in's host-array entry points, device chosen from the node-local
    rank) share the GPU on 1/K of a batch each, as the MPI ranks of a node do.  A small batch here; the full-size figures are in
    profiles/ (r03_ranks_per_gpu.md)."""
    import json
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ranks-per-gpu", "1,2,3", "--ncol", "12000", "--steps", "3"],
                         check=True, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    assert set(d["ranks_per_gpu"]) == {"1", "2", "3"}
    for k, v in d["ranks_per_gpu"].items():
        assert v["columns_per_rank"] == -(-12000 // int(k)) and v["aggregate_columns_per_s"] > 1e4
    with capsys.disabled():
        print("\nranks sharing one GPU, 12 000 columns: " + ", ".join(f"K={k}: {v['aggregate_columns_per_s']:.3g} col/s" for k, v in d["ranks_per_gpu"].items()))
