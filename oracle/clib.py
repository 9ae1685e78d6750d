"""oracle/clib.py -- TEST INFRASTRUCTURE.  ctypes driver for oracle/liboracle.so, the plain-C restatement
of RRTMG_LW + McICA (oracle/lw_oracle_impl.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product path never does.

Same array convention as oracle/reflib.py (numpy C-order == reversed Fortran shape).
"""
import ctypes
import os
import subprocess
import numpy as np
from geosradiation_gridcomp_amd.tableblob import read_blob

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "geosradiation_gridcomp_amd", "data")
_lib = None
_keep = {}     # keeps table arrays alive
_state = {"f32": None, "f64": None}

LW_IN2D = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr",
           "cldf", "ciwp", "clwp", "rei", "rel"]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _lib = ctypes.CDLL(so)
        for sfx in ("f32", "f64"):
            getattr(_lib, f"oracle_lw_set_table_{sfx}").argtypes = [ctypes.c_char_p, ctypes.c_void_p]
            getattr(_lib, f"oracle_sw_set_table_{sfx}").argtypes = [ctypes.c_char_p, ctypes.c_void_p]
            getattr(_lib, f"oracle_kiss_to_real_{sfx}").restype = ctypes.c_float if sfx == "f32" else ctypes.c_double
            _load_tables(sfx)
            set_inhomogeneity(0, sfx)
            set_corr_lengths(None, None, sfx)
    return _lib


def _sfx(prec):
    return {"r4": "f32", "r8": "f64", "f32": "f32", "f64": "f64"}[prec]


def dtype_of(prec):
    return np.float32 if _sfx(prec) == "f32" else np.float64


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _load_tables(sfx):
    kind = "r4" if sfx == "f32" else "r8"
    _, t = read_blob(os.path.join(DATA, f"rrtmg_lw_{kind}.grtb"))
    setter = getattr(_lib, f"oracle_lw_set_table_{sfx}")
    for name, a in t.items():
        a = np.asfortranarray(a)
        flat = np.ascontiguousarray(a.ravel(order="F"))
        _keep[(sfx, name)] = flat
        setter(name.encode(), _p(flat))
    _, t = read_blob(os.path.join(DATA, f"rrtmg_sw_{kind}.grtb"))
    setter = getattr(_lib, f"oracle_sw_set_table_{sfx}")
    for name, a in t.items():
        flat = np.ascontiguousarray(np.asfortranarray(a).ravel(order="F"))
        _keep[(sfx, "sw_" + name)] = flat
        setter(name.encode(), _p(flat))
    _, t = read_blob(os.path.join(DATA, f"chou_sw_{kind}.grtb"))
    setter = getattr(_lib, f"oracle_chou_sw_set_table_{sfx}")
    setter.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    for name, a in t.items():
        flat = np.ascontiguousarray(np.asfortranarray(a).ravel(order="F"))
        _keep[(sfx, "chsw_" + name)] = flat
        setter(name.encode(), _p(flat))
    _, t = read_blob(os.path.join(DATA, f"chou_lw_{kind}.grtb"))
    setter = getattr(_lib, f"oracle_chou_set_table_{sfx}")
    setter.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    for name, a in t.items():
        flat = np.ascontiguousarray(np.asfortranarray(a).ravel(order="F"))
        _keep[(sfx, "chou_" + name)] = flat
        setter(name.encode(), _p(flat))


def set_inhomogeneity(ih, prec="f32"):
    """0 = homogeneous, 1 = beta, 2 = gamma (cloud_condensate_inhomogeneity.F90:45)."""
    sfx = _sfx(prec)
    L = _lib if _lib is not None else lib()
    setter = getattr(L, f"oracle_lw_set_table_{sfx}")
    if ih == 0:
        setter(b"xcw", None)
        return
    kind = "r4" if sfx == "f32" else "r8"
    _, t = read_blob(os.path.join(DATA, f"xcw_{'beta' if ih == 1 else 'gamma'}_{kind}.grtb"))
    flat = np.ascontiguousarray(t["xcw"].ravel(order="F"))
    _keep[(sfx, "xcw")] = flat
    setter(b"xcw", _p(flat))


# Oreopoulos et al. (2012) defaults (cloud_subcol_gen.F90:51-59)
DEF_ADL = (1.4315, 2.1219, 7.0, -25.584)
DEF_RDL = (0.72192, 0.78996, 8.5, 40.404)


def set_corr_lengths(adl=None, rdl=None, prec="f32"):
    sfx = _sfx(prec)
    L = _lib if _lib is not None else lib()
    dt = dtype_of(sfx)
    a = np.array(DEF_ADL if adl is None else adl, dtype=dt)
    r = np.array(DEF_RDL if rdl is None else rdl, dtype=dt)
    getattr(L, f"oracle_set_corr_lengths_{sfx}")(_p(a), _p(r))


def rrtmg_lw(inp, prec="f32", dudTs=True, iceflg=3, liqflg=1, band_output=None, intermediates=False):
    L = lib()
    sfx = _sfx(prec)
    dt = dtype_of(sfx)
    nlay, ncol = inp["play"].shape
    a = {k: _c(inp[k], dt) for k in ["play", "plev", "tlay", "tlev", "tsfc", "emis", "tauaer", "zm", "alat"] + LW_IN2D}
    bo = np.zeros(16, dtype=np.int32) if band_output is None else _c(band_output, np.int32)
    out = {k: np.zeros((nlay + 1, ncol), dtype=dt) for k in ["uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs"]}
    out["clearCounts"] = np.zeros((4, ncol), dtype=np.int32)
    out["olrb"] = np.zeros((ncol, 16), dtype=dt)
    out["dolrb_dTs"] = np.zeros((ncol, 16), dtype=dt)
    inter = [None, None, None]
    if intermediates:
        for k, nm in enumerate(("taug", "pfracs", "taucmc")):
            out[nm] = np.zeros((ncol, 140, nlay), dtype=dt)
            inter[k] = _p(out[nm])
    ci = ctypes.c_int
    rc = getattr(L, f"oracle_rrtmg_lw_{sfx}")(
        ci(ncol), ci(nlay), ci(1 if dudTs else 0), _p(a["play"]), _p(a["plev"]), _p(a["tlay"]), _p(a["tlev"]),
        _p(a["tsfc"]), _p(a["emis"]), *[_p(a[k]) for k in LW_IN2D], ci(iceflg), ci(liqflg), _p(a["tauaer"]),
        _p(a["zm"]), _p(a["alat"]), ci(int(inp["dyofyr"])), ci(int(inp["cloudLM"])), ci(int(inp["cloudMH"])),
        _p(out["clearCounts"]), _p(out["uflx"]), _p(out["dflx"]), _p(out["uflxc"]), _p(out["dflxc"]),
        _p(out["duflx_dTs"]), _p(out["duflxc_dTs"]), _p(bo), _p(out["olrb"]), _p(out["dolrb_dTs"]), *inter)
    out["rc"] = rc
    return out


def mcica(zmid, alat, doy, play, cldfrac, ciwp, clwp, nsubcol, seed_order=(1, 2, 3, 4), cwp_tiny=1e-20, prec="f32"):
    L = lib()
    sfx = _sfx(prec)
    dt = dtype_of(sfx)
    nlay, ncol = play.shape
    cldy = np.zeros((ncol, nsubcol, nlay), dtype=np.int32)
    ci_s = np.zeros((ncol, nsubcol, nlay), dtype=dt); cl_s = np.zeros_like(ci_s)
    so = np.array(seed_order, dtype=np.int32)
    ci = ctypes.c_int
    tiny = ctypes.c_float(cwp_tiny) if sfx == "f32" else ctypes.c_double(cwp_tiny)
    rc = getattr(L, f"oracle_mcica_{sfx}")(ci(ncol), ci(nsubcol), ci(nlay), _p(_c(zmid, dt)), _p(_c(alat, dt)), ci(int(doy)),
                                           _p(_c(play, dt)), _p(_c(cldfrac, dt)), _p(_c(ciwp, dt)), _p(_c(clwp, dt)), tiny,
                                           _p(so), _p(cldy), _p(ci_s), _p(cl_s))
    if rc:
        raise RuntimeError(f"oracle_mcica rc={rc}")
    return cldy, ci_s, cl_s


def clearcounts(cldy, cloudLM, cloudMH):
    L = lib()
    ncol, nsubcol, nlay = cldy.shape
    cnt = np.zeros((ncol, 4), dtype=np.int32)
    ci = ctypes.c_int
    rc = L.oracle_clearcounts_f32(ci(ncol), ci(nsubcol), ci(nlay), ci(int(cloudLM)), ci(int(cloudMH)), _p(_c(cldy, np.int32)), _p(cnt))
    if rc:
        raise RuntimeError("invalid pressure super-layers!")
    return cnt


def kiss_stream(seed4, n, prec="f32"):
    L = lib()
    sfx = _sfx(prec)
    out = np.zeros(n, dtype=dtype_of(sfx))
    getattr(L, f"oracle_kiss_stream_{sfx}")(_p(np.array(seed4, dtype=np.int32)), ctypes.c_int(n), _p(out))
    return out


def kiss_to_real(kiss, prec="f32"):
    return getattr(lib(), f"oracle_kiss_to_real_{_sfx(prec)}")(ctypes.c_int32(kiss))


def zcw_lookup(cdf, sigma, prec="f32"):
    L = lib()
    sfx = _sfx(prec)
    dt = dtype_of(sfx)
    c = _c(cdf, dt); s = _c(sigma, dt); z = np.zeros_like(c)
    getattr(L, f"oracle_zcw_lookup_{sfx}")(ctypes.c_int(c.size), _p(c), _p(s), _p(z))
    return z


# ---- RRTMG_SW ---------------------------------------------------------------------------------------------------
NG_SW = 112
SW_IN2D = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr"]


def sw_setcoef_taumol(inp, isolvar=0, svar=(1.0, 1.0, 1.0), svar_bnd=None, prec="f32"):
    L = lib()
    sfx = _sfx(prec); dt = dtype_of(sfx)
    nlay, ncol = inp["play"].shape
    T = lambda k: _c(np.asarray(inp[k], dtype=dt).T, dt)
    a = [T(k) for k in ("play", "tlay", "plev", "h2ovmr", "co2vmr", "o3vmr", "ch4vmr", "o2vmr")]
    sv = np.array(svar, dtype=dt)
    sb = np.ones((3, 29), dtype=dt) if svar_bnd is None else _c(svar_bnd, dt)
    taug = np.zeros((ncol, NG_SW, nlay), dtype=dt); taur = np.zeros_like(taug)
    ssi = np.zeros((ncol, NG_SW), dtype=dt); sfz = np.zeros_like(ssi)
    colmol = np.zeros((ncol, nlay), dtype=dt); lt = np.zeros(ncol, dtype=np.int32)
    getattr(L, f"oracle_sw_setcoef_taumol_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(nlay), *[_p(x) for x in a], ctypes.c_int(isolvar),
                                                   _p(sv), _p(sb), _p(taug), _p(taur), _p(ssi), _p(sfz), _p(colmol), _p(lt))
    return dict(taug=taug, taur=taur, ssi=ssi, sfluxzen=sfz, colmol=colmol, laytrop=lt)


def sw_cldprmc(cldy, ciwpmc, clwpmc, rei, rel, iceflag=3, liqflag=1, prec="f32"):
    L = lib()
    sfx = _sfx(prec); dt = dtype_of(sfx)
    ncol, ng, nlay = cldy.shape
    T = lambda x: _c(np.asarray(x, dtype=dt).T, dt)
    out = [np.zeros((ncol, ng, nlay), dtype=dt) for _ in range(4)]
    rc = getattr(L, f"oracle_sw_cldprmc_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(nlay), ctypes.c_int(iceflag), ctypes.c_int(liqflag),
                                                _p(_c(cldy, np.int32)), _p(_c(ciwpmc, dt)), _p(_c(clwpmc, dt)), _p(T(rei)), _p(T(rel)),
                                                *[_p(o) for o in out])
    if rc:
        raise RuntimeError(f"oracle_sw_cldprmc rc={rc}")
    return out


def rrtmg_sw(inp, prec="f32", scon=1361.0, adjes=1.0, isolvar=0, iceflg=3, liqflg=1, iaer=0, normFlx=0, do_drfband=False,
             indsolvar=None, bndscl=None, solcycfrac=None):
    """Full rrtmg_sw restatement (rrtmg_sw_rad.F90:68).  `inp` from synth.make_columns (needs coszen, albedos)."""
    L = lib()
    sfx = _sfx(prec); dt = dtype_of(sfx)
    nlay, ncol = inp["play"].shape
    c = lambda k: _c(inp[k], dt)
    plev = c("plev")
    aer = [(_c(inp[k], dt) if iaer == 10 else np.zeros(1, dtype=dt)) for k in ("tauaer_sw", "ssaaer_sw", "asmaer_sw")]
    out = {k: np.zeros((nlay + 1, ncol), dtype=dt) for k in ("swuflx", "swdflx", "swuflxc", "swdflxc")}
    for k in ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf"):
        out[k] = np.zeros(ncol, dtype=dt)
    out["fswband"] = np.zeros((14, ncol), dtype=dt)
    out["cot"] = np.zeros((8, ncol), dtype=dt)
    out["drband"] = np.zeros((14, ncol), dtype=dt); out["dfband"] = np.zeros((14, ncol), dtype=dt)
    out["clearCounts"] = np.zeros((4, ncol), dtype=np.int32)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    ci = ctypes.c_int
    ind = None if indsolvar is None else _p(np.array(indsolvar, dtype=dt))
    bs = None if bndscl is None else _p(np.array(bndscl, dtype=dt))
    scf = None if solcycfrac is None else _p(np.array([solcycfrac], dtype=dt))
    rc = getattr(L, f"oracle_rrtmg_sw_{sfx}")(
        ci(ncol), ci(nlay), R(scon), R(adjes), _p(c("coszen")), ci(isolvar), _p(c("play")), _p(plev), _p(c("tlay")),
        _p(c("h2ovmr")), _p(c("o3vmr")), _p(c("co2vmr")), _p(c("ch4vmr")), _p(c("o2vmr")), ci(iceflg), ci(liqflg),
        _p(c("cldf")), _p(c("ciwp")), _p(c("clwp")), _p(c("rei")), _p(c("rel")), ci(int(inp["dyofyr"])), _p(c("zm")), _p(c("alat")),
        ci(iaer), _p(aer[0]), _p(aer[1]), _p(aer[2]), _p(c("asdir")), _p(c("asdif")), _p(c("aldir")), _p(c("aldif")),
        ci(int(inp["cloudLM"])), ci(int(inp["cloudMH"])), ci(normFlx), _p(out["clearCounts"]), _p(out["swuflx"]), _p(out["swdflx"]),
        _p(out["swuflxc"]), _p(out["swdflxc"]), _p(out["nirr"]), _p(out["nirf"]), _p(out["parr"]), _p(out["parf"]), _p(out["uvrr"]),
        _p(out["uvrf"]), _p(out["fswband"]), _p(out["cot"]), ci(1 if do_drfband else 0), _p(out["drband"]), _p(out["dfband"]), bs, ind, scf)
    out["rc"] = rc
    return out


def nrlssi2_adjust(solcycfr, indsolvar, prec="f32"):
    """adjust_solcyc_amplitudes restatement (NRLSSI2.F90:236-271) -> indsolvar_scl(2); raises where the reference error-stops"""
    sfx = _sfx(prec); dt = dtype_of(sfx)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    o = np.zeros(2, dtype=dt)
    if getattr(lib(), f"oracle_nrlssi2_adjust_{sfx}")(R(dt(solcycfr)), _p(np.array(indsolvar, dtype=dt)), _p(o)):
        raise ValueError("RRTMG_SW: solcycfr must be in [0,1]")
    return o


def nrlssi2_interp(solcycfr, prec="f32"):
    """interpolate_indices restatement (NRLSSI2.F90:277-332) -> (Mg, SB)"""
    sfx = _sfx(prec); dt = dtype_of(sfx)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    o = np.zeros(2, dtype=dt)
    if getattr(lib(), f"oracle_nrlssi2_interp_{sfx}")(R(dt(solcycfr)), _p(o)):
        raise ValueError("RRTMG_SW: solcycfr must be in [0,1]")
    return o[0], o[1]


def nrlssi2_means(indsolvar=None, prec="f32"):
    """initialize_NRLSSI2's isolvar = 1 cycle means (NRLSSI2.F90:160-232) -> (<svar_f>, <svar_s>)"""
    sfx = _sfx(prec); dt = dtype_of(sfx)
    o = np.zeros(2, dtype=dt)
    getattr(lib(), f"oracle_nrlssi2_means_{sfx}")(None if indsolvar is None else _p(np.array(indsolvar, dtype=dt)), _p(o))
    return o[0], o[1]


def sw_solar_isolvar1(scon, solcycfrac, indsolvar=None, prec="f32"):
    """svar_f, svar_s, svar_i of the isolvar = 1 branch of the solar-variability block (rrtmg_sw_rad.F90:906-930,994-1008,1060-1079)"""
    sfx = _sfx(prec); dt = dtype_of(sfx)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    o = np.zeros(3, dtype=dt)
    rc = getattr(lib(), f"oracle_sw_solar_isolvar1_{sfx}")(R(dt(scon)), None if indsolvar is None else _p(np.array(indsolvar, dtype=dt)),
                                                          _p(np.array([solcycfrac], dtype=dt)), _p(o))
    if rc:
        raise ValueError(f"isolvar 1: rc {rc}")
    return o


def irrad(ch, prec="f32", trace=True):
    """Chou-Suarez LW (irrad.F90:27).  `ch` from synth.chou_lw_inputs.  Returns the 8 flux arrays + dfdts (np+1, m), sfcem (m),
    taudiag (10, np, m); upward fluxes are negative as in the reference."""
    L = lib()
    sfx = _sfx(prec); dt = dtype_of(sfx)
    n1, m = ch["ple"].shape
    npl = n1 - 1
    c = lambda k: _c(ch[k], dt)
    a = {k: c(k) for k in ("ple", "ta", "wa", "oa", "tb", "n2o", "ch4", "cfc11", "cfc12", "cfc22", "cwc", "fcld", "reff", "fs", "tg", "eg",
                           "tv", "ev", "rv")}
    aer = {k: _c(ch[k], dt).copy() for k in ("taua", "ssaa", "asya")}        # INOUT in the reference
    out = {k: np.zeros((n1, m), dtype=dt) for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts")}
    out["sfcem"] = np.zeros(m, dtype=dt)
    out["taudiag"] = np.zeros((10, npl, m), dtype=dt)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    ci = ctypes.c_int
    rc = getattr(L, f"oracle_irrad_{sfx}")(
        ci(m), ci(npl), _p(a["ple"]), _p(a["ta"]), _p(a["wa"]), _p(a["oa"]), _p(a["tb"]), R(ch["co2"]), ci(1 if trace else 0),
        _p(a["n2o"]), _p(a["ch4"]), _p(a["cfc11"]), _p(a["cfc12"]), _p(a["cfc22"]), _p(a["cwc"]), _p(a["fcld"]), ci(int(ch["ict"])),
        ci(int(ch["icb"])), _p(a["reff"]), ci(int(ch["ns"])), _p(a["fs"]), _p(a["tg"]), _p(a["eg"]), _p(a["tv"]), _p(a["ev"]), _p(a["rv"]),
        ci(int(ch["na"])), ci(int(ch["nb"])), _p(aer["taua"]), _p(aer["ssaa"]), _p(aer["asya"]),
        *[_p(out[k]) for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts", "sfcem", "taudiag")])
    out["rc"] = rc
    out.update({k + "_out": v for k, v in aer.items()})
    return out


def sorad(cs, prec="f32", do_drfband=True):
    """Chou-Suarez SW (sorad.F90:43).  `cs` from synth.chou_sw_inputs.  Fluxes are fractions of the TOA insolation."""
    L = lib()
    sfx = _sfx(prec); dt = dtype_of(sfx)
    n1, m = cs["pl"].shape
    npl = n1 - 1
    a = {k: _c(cs[k], dt) for k in ("cosz", "pl", "ta", "wa", "oa", "cwc", "fcld", "reff", "hk_uv", "hk_ir", "taua", "ssaa", "asya", "rsuvbm",
                                    "rsuvdf", "rsirbm", "rsirdf")}
    out = {k: np.zeros((n1, m), dtype=dt) for k in ("flx", "flc", "flxu", "flcu")}
    for k in ("fdiruv", "fdifuv", "fdirpar", "fdifpar", "fdirir", "fdifir"):
        out[k] = np.zeros(m, dtype=dt)
    out["flx_sfc_band"] = np.zeros((8, m), dtype=dt); out["drband"] = np.zeros((8, m), dtype=dt); out["dfband"] = np.zeros((8, m), dtype=dt)
    R = ctypes.c_float if sfx == "f32" else ctypes.c_double
    ci = ctypes.c_int
    rc = getattr(L, f"oracle_sorad_{sfx}")(
        ci(m), ci(npl), ci(8), _p(a["cosz"]), _p(a["pl"]), _p(a["ta"]), _p(a["wa"]), _p(a["oa"]), R(cs["co2"]), _p(a["cwc"]), _p(a["fcld"]),
        ci(int(cs["ict"])), ci(int(cs["icb"])), _p(a["reff"]), _p(a["hk_uv"]), _p(a["hk_ir"]), _p(a["taua"]), _p(a["ssaa"]), _p(a["asya"]),
        _p(a["rsuvbm"]), _p(a["rsuvdf"]), _p(a["rsirbm"]), _p(a["rsirdf"]), _p(out["flx"]), _p(out["flc"]), _p(out["fdiruv"]),
        _p(out["fdifuv"]), _p(out["fdirpar"]), _p(out["fdifpar"]), _p(out["fdirir"]), _p(out["fdifir"]), _p(out["flxu"]), _p(out["flcu"]),
        _p(out["flx_sfc_band"]), ci(1 if do_drfband else 0), _p(out["drband"]), _p(out["dfband"]))
    out["rc"] = rc
    return out


# ---- GridComp data path either side of the solvers (gridcomp_oracle_impl.h; PARITY UNPINNED, see its header) -------------------
LWD_RR = ["play", "plev", "tlay", "tlev", "tsfc", "emis", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr",
          "cfc22vmr", "ccl4vmr", "cldf", "ciwp", "clwp", "rei", "rel", "tauaer", "zm", "alat"]
SWD_RR = ["play", "plev", "tlay", "tlev", "h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr", "cldf", "ciwp", "clwp", "rei", "rel", "zm",
          "tauaer_sw", "ssaaer_sw", "asmaer_sw"]


def _parr(arrays):
    """(ctypes array of pointers, keep-alive list); None entries -> NULL"""
    arr = (ctypes.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = None if a is None else a.ctypes.data
    return arr


def lwd_prep(f, consts, iceflg=3, liqflg=1, prec="f32"):
    """RRTMG branch of LW_Driver, prep / flip (IRR:3188-3372).  `f`: GEOS-side fields by gridcomp.LWD_IN name, numpy [k][ij]."""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    lm, ncol = f["T"].shape
    nb = 0 if f.get("TAUA") is None else f["TAUA"].shape[0]
    ins = [None if f.get(k) is None else _c(f[k], dt) for k in G.LWD_IN]
    shp = {"plev": (lm + 1, ncol), "tlev": (lm + 1, ncol), "tsfc": (ncol,), "alat": (ncol,), "emis": (16, ncol), "tauaer": (16, lm, ncol)}
    out = {k: np.zeros(shp.get(k, (lm, ncol)), dtype=dt) for k in LWD_RR}
    cs = (ctypes.c_double * len(consts))(*consts)
    getattr(L, f"oracle_lwd_prep_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_int(nb), _parr(ins), cs, ctypes.c_int(iceflg),
                                         ctypes.c_int(liqflg), _parr([out[k] for k in LWD_RR]))
    return out


def lwd_post(flux, clearCounts, emis, ts, prec="f32", want=None):
    """un-flip, sign conventions, SFCEM, net fluxes, cloud fractions (IRR:3487-3615).  flux: uflx dflx uflxc dflxc duflx_dTs duflxc_dTs"""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    n1, ncol = flux["uflx"].shape
    lm = n1 - 1
    fl = [_c(flux[k], dt) for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs")]
    names = G.LWD_OUT[:16]
    want = names if want is None else want
    out = {k: np.zeros((n1, ncol) if k in G.LWD_OUT_3D else (ncol,), dtype=dt) for k in names if k in want}
    cc = np.ascontiguousarray(clearCounts, dtype=np.int32)
    getattr(L, f"oracle_lwd_post_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_int(140), _parr(fl), _p(cc), _p(_c(emis, dt)),
                                         _p(_c(ts, dt)), _parr([out.get(k) for k in names]))
    return out


def lw_update_flx(st, lm, rrtmg, lev_mid_high, lev_low_mid, undef, prec="f32", want=None):
    """Update_Flx (IRR:3796-3999).  `st`: internals by gridcomp.LWU_IN name."""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    ncol = st["TSINST"].shape[0]
    ins = [None if st.get(k) is None else _c(st[k], dt) for k in G.LWU_IN]
    want = G.LWU_OUT if want is None else want
    out = {k: np.zeros((lm + 1, ncol) if k in G.LWU_OUT_3D else (ncol,), dtype=dt) for k in G.LWU_OUT if k in want}
    getattr(L, f"oracle_lw_update_flx_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_int(1 if rrtmg else 0), ctypes.c_int(lev_mid_high),
                                              ctypes.c_int(lev_low_mid), ctypes.c_double(undef), _parr(ins),
                                              _parr([out.get(k) for k in G.LWU_OUT]))
    return out


def swd_prep(f, consts, iceflg=3, liqflg=1, prec="f32"):
    """RRTMG branch of SORADCORE, prep / flip (SOL:6113-6212).  Returns (rrtmg-side arrays, normalised TAUA/SSAA/ASYA)."""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    lm, ncol = f["T"].shape
    nb = 14
    ins = [None if f.get(k) is None else _c(f[k], dt).copy() for k in G.SWD_IN[:15]]
    shp = {"plev": (lm + 1, ncol), "tlev": (lm + 1, ncol), "tauaer_sw": (14, lm, ncol), "ssaaer_sw": (14, lm, ncol), "asmaer_sw": (14, lm, ncol)}
    out = {k: np.zeros(shp.get(k, (lm, ncol)), dtype=dt) for k in SWD_RR}
    cs = (ctypes.c_double * len(consts))(*consts)
    getattr(L, f"oracle_swd_prep_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_int(nb), _parr(ins), cs, ctypes.c_int(iceflg),
                                         ctypes.c_int(liqflg), _parr([out[k] for k in SWD_RR]))
    return out, {"TAUA": ins[12], "SSAA": ins[13], "ASYA": ins[14]}


def swc_prep(f, consts, prec="f32"):
    """Chou-Suarez branch of SORADCORE, prep (SOL:4484-4528).  `f`: GEOS fields PLE OX QI QL QR QS RI RL RR RS, [k][ij];
    consts = (O3MW, AIRMW, UNDEF).  Returns PLhPa (LM+1,ncol), O3 (LM,ncol), QQ3 (4,LM,ncol), RR3 (4,LM,ncol)."""
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    lm, ncol = f["OX"].shape
    ins = [_c(f[k], dt) for k in ("PLE", "OX", "QI", "QL", "QR", "QS", "RI", "RL", "RR", "RS")]
    out = dict(PLhPa=np.zeros((lm + 1, ncol), dtype=dt), O3=np.zeros((lm, ncol), dtype=dt), QQ3=np.zeros((4, lm, ncol), dtype=dt),
               RR3=np.zeros((4, lm, ncol), dtype=dt))
    cs = (ctypes.c_double * 3)(*consts)
    getattr(L, f"oracle_swc_prep_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), _parr(ins), cs,
                                         _parr([out[k] for k in ("PLhPa", "O3", "QQ3", "RR3")]))
    return out


def swd_post(flux, clearCounts, cot8, aerosols, undef, prec="f32"):
    """SOL:6395-6450.  flux: swuflx swdflx swuflxc swdflxc; cot8: cotdtp cotdhp cotdmp cotdlp cotntp cotnhp cotnmp cotnlp"""
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    n1, ncol = flux["swuflx"].shape
    fl = [_c(flux[k], dt) for k in ("swuflx", "swdflx", "swuflxc", "swdflxc")]
    c8 = [_c(cot8[k], dt) for k in ("cotdtp", "cotdhp", "cotdmp", "cotdlp", "cotntp", "cotnhp", "cotnmp", "cotnlp")]
    names = ["FSW", "FSC", "FSWU", "FSCU", "CLDTS", "CLDHS", "CLDMS", "CLDLS", "COTTP", "COTHP", "COTMP", "COTLP"]
    out = {k: np.zeros((n1, ncol) if k.startswith("FS") else (ncol,), dtype=dt) for k in names}
    cc = np.ascontiguousarray(clearCounts, dtype=np.int32)
    getattr(L, f"oracle_swd_post_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(n1 - 1), ctypes.c_int(112), ctypes.c_int(1 if aerosols else 0),
                                         ctypes.c_double(undef), _parr(fl), _p(cc), _parr(c8), _parr([out[k] for k in names]))
    return out


def sw_update_export(st, lm, nbands, prec="f32", want=None):
    """UPDATE_EXPORT flux part (SOL:7540-7579).  `st`: gridcomp.SWU_IN internals."""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    ncol = st["SLR"].shape[0]
    ins = [None if st.get(k) is None else _c(st[k], dt) for k in G.SWU_IN]
    want = G.SWU_OUT if want is None else want
    shape = lambda k: (lm + 1, ncol) if k in G.SWU_OUT_3D else ((nbands, ncol) if k in G.SWU_OUT_BAND else (ncol,))
    out = {k: np.zeros(shape(k), dtype=dt) for k in G.SWU_OUT if k in want}
    getattr(L, f"oracle_sw_update_export_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_int(nbands), _parr(ins),
                                                 _parr([out.get(k) for k in G.SWU_OUT]))
    return out


def rad_tendencies(st, lm, grav, cp, prec="f32", want=None):
    """RAD:798-819.  `st`: gridcomp.RT_IN fields."""
    from geosradiation_gridcomp_amd import gridcomp as G
    L = lib(); sfx = _sfx(prec); dt = dtype_of(sfx)
    ncol = st["DSFDTS"].shape[0]
    ins = [None if st.get(k) is None else _c(st[k], dt) for k in G.RT_IN]
    want = G.RT_OUT if want is None else want
    out = {k: np.zeros((lm, ncol) if k in G.RT_OUT_3D else (ncol,), dtype=dt) for k in G.RT_OUT if k in want}
    getattr(L, f"oracle_rad_tendencies_{sfx}")(ctypes.c_int(ncol), ctypes.c_int(lm), ctypes.c_double(grav), ctypes.c_double(cp), _parr(ins),
                                               _parr([out.get(k) for k in G.RT_OUT]))
    return out
