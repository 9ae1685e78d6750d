"""oracle/reflib.py -- TEST INFRASTRUCTURE.  ctypes driver for oracle/_ref/libref_{r4,r8}.so, i.e. the
reference's own Fortran (compiled unmodified from /root/reference by oracle/Makefile) behind the
C-callable glue in oracle/ref_glue.F90.  Used only by tests/, golden-vector generation, smoke() and
bench.py's cpu_baseline leg -- never by the product path.

Array convention: numpy C-order arrays whose *reversed* shape is the Fortran shape, e.g. Fortran
``play(ncol,nlay)`` <-> numpy ``(nlay, ncol)``.
"""
import ctypes
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}

NG_LW = 140
NB_LW = 16


def available(kind="r4"):
    return os.path.exists(os.path.join(HERE, "_ref", f"libref_{kind}.so"))


def lib(kind="r4"):
    if kind not in _libs:
        L = ctypes.CDLL(os.path.join(HERE, "_ref", f"libref_{kind}.so"))
        L.ref_lw_ini()
        L.ref_sw_ini()
        _libs[kind] = L
    return _libs[kind]


def dtype_of(kind):
    return np.float32 if kind == "r4" else np.float64


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


LW_IN2D = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr",
           "cldf", "ciwp", "clwp", "rei", "rel"]


def rrtmg_lw(inp, kind="r4", psize=4, dudTs=True, iceflg=3, liqflg=1, band_output=None):
    """Call the reference rrtmg_lw (rrtmg_lw_rad.F90:15).  `inp`: dict from synth.make_columns."""
    L = lib(kind)
    dt = dtype_of(kind)
    nlay, ncol = inp["play"].shape
    a = {k: _c(inp[k], dt) for k in ["play", "plev", "tlay", "tlev", "tsfc", "emis", "tauaer", "zm", "alat"] + LW_IN2D}
    bo = np.zeros(NB_LW, dtype=np.int32) if band_output is None else _c(band_output, np.int32)
    out = {k: np.zeros((nlay + 1, ncol), dtype=dt) for k in ["uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs"]}
    out["clearCounts"] = np.zeros((4, ncol), dtype=np.int32)
    out["olrb"] = np.zeros((ncol, NB_LW), dtype=dt)
    out["dolrb_dTs"] = np.zeros((ncol, NB_LW), dtype=dt)
    ci = ctypes.c_int
    L.ref_rrtmg_lw(ci(ncol), ci(nlay), ci(psize), ci(1 if dudTs else 0),
                   _p(a["play"]), _p(a["plev"]), _p(a["tlay"]), _p(a["tlev"]), _p(a["tsfc"]), _p(a["emis"]),
                   *[_p(a[k]) for k in LW_IN2D[:10]],
                   *[_p(a[k]) for k in LW_IN2D[10:]],
                   ci(iceflg), ci(liqflg), _p(a["tauaer"]), _p(a["zm"]), _p(a["alat"]),
                   ci(int(inp["dyofyr"])), ci(int(inp["cloudLM"])), ci(int(inp["cloudMH"])),
                   _p(out["clearCounts"]), _p(out["uflx"]), _p(out["dflx"]), _p(out["uflxc"]), _p(out["dflxc"]),
                   _p(out["duflx_dTs"]), _p(out["duflxc_dTs"]), _p(bo), _p(out["olrb"]), _p(out["dolrb_dTs"]))
    return out


def lw_setcoef_taumol(inp, kind="r4", dudTs=True):
    """setcoef + taumol intermediates (taug, pfracs, Planck terms) for ALL columns of `inp`."""
    L = lib(kind)
    dt = dtype_of(kind)
    nlay, ncol = inp["play"].shape
    T = lambda k: _c(np.asarray(inp[k], dtype=dt).T, dt)          # (ncol, nlay) C == Fortran (nlay, ncol)
    pavel, tavel, pz, tz = T("play"), T("tlay"), T("plev"), T("tlev")
    semiss = T("emis")
    g = [T(k) for k in LW_IN2D[:10]]
    taua = _c(np.transpose(np.asarray(inp["tauaer"], dtype=dt), (2, 0, 1)), dt)   # (ncol, 16, nlay)
    tb = _c(inp["tsfc"], dt)
    taug = np.zeros((ncol, NG_LW, nlay), dtype=dt); pfr = np.zeros_like(taug)
    pl = np.zeros((ncol, nlay, NB_LW), dtype=dt); pv = np.zeros((ncol, nlay + 1, NB_LW), dtype=dt)
    pb = np.zeros((ncol, NB_LW), dtype=dt); dpb = np.zeros_like(pb); pw = np.zeros(ncol, dtype=dt)
    lt = np.zeros(ncol, dtype=np.int32)
    ci = ctypes.c_int
    L.ref_lw_setcoef_taumol(ci(ncol), ci(nlay), ci(1 if dudTs else 0), _p(pavel), _p(tavel), _p(pz), _p(tz), _p(tb),
                            _p(semiss), *[_p(x) for x in g], _p(taua), _p(taug), _p(pfr), _p(pl), _p(pv), _p(pb),
                            _p(dpb), _p(pw), _p(lt))
    return dict(taug=taug, pfracs=pfr, planklay=pl, planklev=pv, plankbnd=pb, dplankbnd_dTs=dpb, pwvcm=pw, laytrop=lt)


def set_inhomogeneity(ih, kind="r4"):
    lib(kind).ref_set_inhomogeneity(ctypes.c_int(ih))


def subcol_defaults(kind="r4"):
    dt = dtype_of(kind)
    adl = np.zeros(4, dtype=dt); rdl = np.zeros(4, dtype=dt)
    lib(kind).ref_subcol_defaults(_p(adl), _p(rdl))
    return adl, rdl


def init_subcol_gen(adl, rdl, kind="r4"):
    dt = dtype_of(kind)
    lib(kind).ref_init_subcol_gen(_p(_c(adl, dt)), _p(_c(rdl, dt)))


def mcica(zmid, alat, doy, play, cldfrac, ciwp, clwp, nsubcol, seed_order=(1, 2, 3, 4), cwp_tiny=1e-20, kind="r4"):
    """generate_stochastic_clouds (cloud_subcol_gen.F90:132).  Profile inputs numpy (nlay, ncol) [API layout];
    returns cldy,ciwp_s,clwp_s as numpy (ncol, nsubcol, nlay) == Fortran (nlay,nsubcol,ncol)."""
    L = lib(kind)
    dt = dtype_of(kind)
    nlay, ncol = play.shape
    T = lambda x: _c(np.asarray(x, dtype=dt).T, dt)
    cldy = np.zeros((ncol, nsubcol, nlay), dtype=np.int32)
    ci_s = np.zeros((ncol, nsubcol, nlay), dtype=dt); cl_s = np.zeros_like(ci_s)
    tiny = np.array([cwp_tiny], dtype=dt)
    so = np.array(seed_order, dtype=np.int32)
    ci = ctypes.c_int
    L.ref_mcica(ci(ncol), ci(ncol), ci(nsubcol), ci(nlay), _p(T(zmid)), _p(_c(alat, dt)), ci(int(doy)), _p(T(play)),
                _p(T(cldfrac)), _p(T(ciwp)), _p(T(clwp)), _p(tiny), _p(so), _p(cldy), _p(ci_s), _p(cl_s))
    return cldy, ci_s, cl_s


def clearcounts(cldy, cloudLM, cloudMH, kind="r4"):
    L = lib(kind)
    ncol, nsubcol, nlay = cldy.shape
    cnt = np.zeros((ncol, 4), dtype=np.int32)
    ci = ctypes.c_int
    L.ref_clearcounts(ci(ncol), ci(ncol), ci(nsubcol), ci(nlay), ci(int(cloudLM)), ci(int(cloudMH)),
                      _p(_c(cldy, np.int32)), _p(cnt))
    return cnt


def zcw_lookup(cdf, sigma, kind="r4"):
    dt = dtype_of(kind)
    c = _c(cdf, dt); s = _c(sigma, dt); z = np.zeros_like(c)
    lib(kind).ref_zcw_lookup(ctypes.c_int(c.size), _p(c), _p(s), _p(z))
    return z


def corr_lengths(alat, doy, kind="r4"):
    """correlation_length_cloud_fraction / _condensate (cloud_subcol_gen.F90:491-542) -> (adl, rdl) in metres."""
    dt = dtype_of(kind)
    a = _c(alat, dt); adl = np.zeros_like(a); rdl = np.zeros_like(a)
    lib(kind).ref_corr_lengths(ctypes.c_int(a.size), ctypes.c_int(int(doy)), _p(a), _p(adl), _p(rdl))
    return adl, rdl


def lw_cldprmc(cldy, ciwpmc, clwpmc, reice, reliq, iceflag=3, liqflag=1, kind="r4"):
    """cldprmc (rrtmg_lw_cldprmc.F90:24).  cldy etc. numpy (ncol, 140, nlay); reice/reliq numpy (nlay, ncol)."""
    L = lib(kind)
    dt = dtype_of(kind)
    ncol, ng, nlay = cldy.shape
    T = lambda x: _c(np.asarray(x, dtype=dt).T, dt)
    tau = np.zeros((ncol, ng, nlay), dtype=dt); cloudy = np.zeros((ncol, nlay), dtype=np.int32)
    ci = ctypes.c_int
    L.ref_lw_cldprmc(ci(ncol), ci(nlay), _p(_c(cldy, np.int32)), _p(_c(ciwpmc, dt)), _p(_c(clwpmc, dt)),
                     _p(T(reice)), _p(T(reliq)), ci(iceflag), ci(liqflag), _p(tau), _p(cloudy))
    return tau, cloudy


def dump_lw_tables(path, kind="r4"):
    p = os.fsencode(path)
    lib(kind).ref_lw_dump_tables(p, ctypes.c_int(len(p)))


def dump_xcw(path, ih, kind="r4"):
    p = os.fsencode(path)
    lib(kind).ref_dump_xcw(p, ctypes.c_int(len(p)), ctypes.c_int(ih))


# ---- RRTMG_SW parts (setcoef_sw / taumol_sw / cldprmc_sw: the reference files free of ESMF/MAPL) -------------------
NG_SW = 112


def dump_sw_tables(path, kind="r4"):
    p = os.fsencode(path)
    lib(kind).ref_sw_dump_tables(p, ctypes.c_int(len(p)))


def dump_chou_lw_tables(path, kind="r4"):
    """irrad_constants + rad_constants (IR part): data modules of the Chou-Suarez LW scheme"""
    p = os.fsencode(path)
    lib(kind).ref_chou_lw_dump_tables(p, ctypes.c_int(len(p)))


def dump_chou_sw_tables(path, kind="r4"):
    """sorad_constants + rad_constants (UV / NIR part): data modules of the Chou-Suarez SW scheme"""
    p = os.fsencode(path)
    lib(kind).ref_chou_sw_dump_tables(p, ctypes.c_int(len(p)))


def sw_setcoef_taumol(inp, isolvar=0, svar=(1.0, 1.0, 1.0), svar_bnd=None, kind="r4"):
    """setcoef_sw + taumol_sw.  Returns taug,taur numpy (ncol,112,nlay); ssi,sfluxzen (ncol,112); colmol (ncol,nlay)."""
    L = lib(kind)
    dt = dtype_of(kind)
    nlay, ncol = inp["play"].shape
    T = lambda k: _c(np.asarray(inp[k], dtype=dt).T, dt)
    play, tlay, plev = T("play"), T("tlay"), T("plev")
    g = [T(k) for k in ("h2ovmr", "co2vmr", "o3vmr", "ch4vmr", "o2vmr")]
    sv = np.array(svar, dtype=dt)
    sb = np.ones((3, 29), dtype=dt) if svar_bnd is None else _c(svar_bnd, dt)
    taug = np.zeros((ncol, NG_SW, nlay), dtype=dt); taur = np.zeros_like(taug)
    ssi = np.zeros((ncol, NG_SW), dtype=dt); sfz = np.zeros_like(ssi)
    colmol = np.zeros((ncol, nlay), dtype=dt); lt = np.zeros(ncol, dtype=np.int32)
    ci = ctypes.c_int
    L.ref_sw_setcoef_taumol(ci(ncol), ci(nlay), _p(play), _p(tlay), _p(plev), *[_p(x) for x in g], ci(isolvar), _p(sv), _p(sb),
                            _p(taug), _p(taur), _p(ssi), _p(sfz), _p(colmol), _p(lt))
    return dict(taug=taug, taur=taur, ssi=ssi, sfluxzen=sfz, colmol=colmol, laytrop=lt)


def sw_cldprmc(cldy, ciwpmc, clwpmc, rei, rel, iceflag=3, liqflag=1, kind="r4"):
    """cldprmc_sw.  cldy etc. numpy (ncol,112,nlay); rei/rel numpy (nlay,ncol) [API layout].  Returns
    taormc,taucmc,ssacmc,asmcmc numpy (ncol,112,nlay)."""
    L = lib(kind)
    dt = dtype_of(kind)
    ncol, ng, nlay = cldy.shape
    T = lambda x: _c(np.asarray(x, dtype=dt).T, dt)
    out = [np.zeros((ncol, ng, nlay), dtype=dt) for _ in range(4)]
    ci = ctypes.c_int
    reiT, relT = T(rei), T(rel)
    # the reference keeps ~8 automatic (nlay,ngptsw,pncol) arrays on the stack: call it in small partitions, as
    # rrtmg_sw itself does (pncol = 2 by default, rrtmg_sw_rad.F90:386-390)
    for c0 in range(0, ncol, 4):
        c1 = min(ncol, c0 + 4)
        o = [np.zeros((c1 - c0, ng, nlay), dtype=dt) for _ in range(4)]
        L.ref_sw_cldprmc(ci(c1 - c0), ci(nlay), ci(iceflag), ci(liqflag), _p(_c(cldy[c0:c1], np.int32)), _p(_c(ciwpmc[c0:c1], dt)),
                         _p(_c(clwpmc[c0:c1], dt)), _p(_c(reiT[c0:c1], dt)), _p(_c(relT[c0:c1], dt)), *[_p(x) for x in o])
        for k in range(4):
            out[k][c0:c1] = o[k]
    return out


# ---- NRLSSI2 host routines of the isolvar = 1 branch (NRLSSI2.F90:160-332) -----------------------------------------
def nrlssi2_adjust(solcycfr, indsolvar, kind="r4"):
    """adjust_solcyc_amplitudes (NRLSSI2.F90:236-271) -> indsolvar_scl(2)"""
    dt = dtype_of(kind)
    f = np.array([solcycfr], dtype=dt); i = _c(indsolvar, dt); o = np.zeros(2, dtype=dt)
    lib(kind).ref_nrlssi2_adjust(_p(f), _p(i), _p(o))
    return o


def nrlssi2_interp(solcycfr, kind="r4"):
    """interpolate_indices (NRLSSI2.F90:277-332) -> (Mg, SB)"""
    dt = dtype_of(kind)
    f = np.array([solcycfr], dtype=dt); mg = np.zeros(1, dtype=dt); sb = np.zeros(1, dtype=dt)
    lib(kind).ref_nrlssi2_interp(_p(f), _p(mg), _p(sb))
    return mg[0], sb[0]


def nrlssi2_means(indsolvar=None, kind="r4"):
    """initialize_NRLSSI2(1 [, indsolvar]) (NRLSSI2.F90:160-232) -> (isolvar_1_mean_svar_f, isolvar_1_mean_svar_s)"""
    dt = dtype_of(kind)
    i = np.ones(2, dtype=dt) if indsolvar is None else _c(indsolvar, dt)
    mf = np.zeros(1, dtype=dt); ms = np.zeros(1, dtype=dt)
    lib(kind).ref_nrlssi2_means(ctypes.c_int(0 if indsolvar is None else 1), _p(i), _p(mf), _p(ms))
    return mf[0], ms[0]
