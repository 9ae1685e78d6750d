"""Deterministic synthetic atmospheric columns.

Every quantity of column ``i`` is a pure function of ``(seed, i)`` (counter-based hash), so a shard
``[start, start+ncol)`` generated on any rank is byte-identical to the same columns generated as part
of a bigger batch.  All arrays are float32 (GEOS feeds the solvers default-real = fp32 data) and are
laid out exactly as the reference's solver API wants them: Fortran ``(ncol, nlay)`` == C-order
``(nlay, ncol)``, i.e. column index fastest.  RRTMG ordering: layer 0 is the lowest model layer.

The unit conversions / level-temperature / mid-layer-height formulas mirror what the GEOS driver does
before calling ``rrtmg_lw`` (GEOS_IrradGridComp.F90:3243-3358) so the inputs are representative.
"""
import numpy as np

SEED = 20250220
NBNDLW = 16
NBNDSW = 14

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    # splitmix64 finaliser (public-domain constant set)
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def _u(seed, col, stream, n=1):
    """uniform [0,1) doubles, shape (n, ncol): hash of (seed, column, stream, k)."""
    with np.errstate(over="ignore"):
        col = col.astype(np.uint64)[None, :]
        k = np.arange(n, dtype=np.uint64)[:, None]
        x = _mix(np.uint64(seed) ^ _mix(col * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(0x9E3779B1)))
        x = _mix(x + k * np.uint64(0xD6E8FEB86659FD93))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _n(seed, col, stream, n=1):
    u1 = _u(seed, col, stream, n)
    u2 = _u(seed, col, stream + 7919, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def make_columns(ncol, nlay=72, seed=SEED, start=0, cloudy_frac=0.0, aerosol=False, lit=True):
    """Return a dict of float32/int32 arrays (C-order, column index fastest).

    cloudy_frac : fraction of columns that get 1-3 cloud decks (SURVEY 8(d) cfg 3/4)
    aerosol     : fill tauaer (LW absorption tau) / SW tau, ssa, g per band
    """
    col = np.arange(start, start + ncol)
    f32 = np.float32

    # ---- pressure grid: surface -> 0.01 hPa, finer near the surface (hybrid-like) ---------------
    ps = 950.0 + 85.0 * _u(seed, col, 1)[0]                               # hPa
    k = np.arange(nlay + 1, dtype=np.float64)[:, None] / nlay
    eta = 0.35 * k + 0.65 * k ** 1.6                                      # monotone 0..1
    plev = (ps[None, :] - 0.01) * np.exp(-11.5 * eta) + 0.01              # (nlay+1, ncol), [0]=surface
    plev = plev.astype(f32)
    play = (0.5 * (plev[:-1].astype(np.float64) + plev[1:].astype(np.float64))).astype(f32)

    # ---- temperature: MLS-like piecewise-linear in ln p + per-column offset -----------------------
    lp = np.log(play.astype(np.float64))
    xp = np.log(np.array([0.005, 0.1, 1.0, 60.0, 200.0, 1100.0]))
    fp = np.array([190.0, 230.0, 270.0, 215.0, 215.0, 297.0])
    tlay = np.interp(lp, xp, fp) + 2.0 * _n(seed, col, 2)[0][None, :]
    tlay = np.clip(tlay, 165.0, 335.0).astype(f32)
    # level temperatures: pressure-thickness weighted (IRR:3255-3261), surface from 2 m proxy
    dp = (plev[:-1].astype(np.float64) - plev[1:].astype(np.float64))    # layer thickness, bottom-up
    tl = tlay.astype(np.float64)
    tlev = np.empty((nlay + 1, ncol))
    tlev[1:nlay] = (tl[1:] * dp[:-1] + tl[:-1] * dp[1:]) / (dp[:-1] + dp[1:])
    tlev[nlay] = tlev[nlay - 1]                                            # model top
    tlev[0] = tl[0] + 0.6                                                  # "T2M"
    tlev = tlev.astype(f32)
    tsfc = (tlev[0].astype(np.float64) - 2.0 + 6.0 * _u(seed, col, 3)[0]).astype(f32)

    # ---- gases (vmr wrt dry air) -----------------------------------------------------------------
    sig = play.astype(np.float64) / ps[None, :]
    h2o = np.maximum(4e-6, 0.02 * sig ** 3 * (0.5 + _u(seed, col, 4)[0][None, :]))
    o3 = 3e-8 + 8e-6 * np.exp(-0.5 * ((lp - np.log(10.0)) / 1.1) ** 2) * (0.7 + 0.6 * _u(seed, col, 5)[0][None, :])
    ones = np.ones((nlay, ncol))
    out = dict(
        play=play, plev=plev, tlay=tlay, tlev=tlev, tsfc=tsfc,
        h2ovmr=h2o.astype(f32), o3vmr=o3.astype(f32),
        co2vmr=(4.0e-4 * ones).astype(f32), ch4vmr=(1.8e-6 * ones).astype(f32),
        n2ovmr=(3.2e-7 * ones).astype(f32), o2vmr=(0.209 * ones).astype(f32),
        cfc11vmr=(2e-10 * ones).astype(f32), cfc12vmr=(5e-10 * ones).astype(f32),
        cfc22vmr=(2e-10 * ones).astype(f32), ccl4vmr=(1.105e-10 * ones).astype(f32),
        emis=np.full((NBNDLW, ncol), 0.98, dtype=f32),
    )

    # ---- mid-layer heights (IRR:3343-3351): ZM(1)=0, dz = R T/g * dp/p -----------------------------
    rgas, grav = 287.04, 9.80665
    zm = np.zeros((nlay, ncol))
    pl64, ple64, tlev64 = play.astype(np.float64), plev.astype(np.float64), tlev.astype(np.float64)
    for kk in range(1, nlay):
        zm[kk] = zm[kk - 1] + rgas * tlev64[kk] / grav * (pl64[kk - 1] - pl64[kk]) / ple64[kk]
    out["zm"] = zm.astype(f32)
    out["alat"] = ((_u(seed, col, 6)[0] - 0.5) * np.pi).astype(f32)

    # ---- clouds ------------------------------------------------------------------------------------
    cldf = np.zeros((nlay, ncol)); ciwp = np.zeros((nlay, ncol)); clwp = np.zeros((nlay, ncol))
    rel = np.full((nlay, ncol), 10.0); rei = np.full((nlay, ncol), 40.0)
    if cloudy_frac > 0.0:
        is_cldy = _u(seed, col, 10)[0] < cloudy_frac
        ndeck = 1 + (_u(seed, col, 11)[0] * 3).astype(int)
        pc = play.astype(np.float64)
        for d in range(3):
            act = is_cldy & (ndeck > d)
            # deck centre pressure: low (850), mid (550), high (250) +- jitter; thickness 2-6 layers
            pcen = np.array([850.0, 550.0, 250.0])[d] * (0.85 + 0.3 * _u(seed, col, 20 + d)[0])
            half = 20.0 + 60.0 * _u(seed, col, 30 + d)[0]
            inside = (np.abs(pc - pcen[None, :]) < half[None, :]) & act[None, :]
            frac = 0.05 + 0.95 * _u(seed, col, 40 + d)[0]
            # a few overcast decks to exercise cldfrac==1
            frac = np.where(_u(seed, col, 45 + d)[0] < 0.1, 1.0, frac)
            lw = np.exp(np.log(1.0) + np.log(200.0) * _u(seed, col, 50 + d)[0])
            iw = np.exp(np.log(0.5) + np.log(100.0) * _u(seed, col, 60 + d)[0])
            warm = tlay.astype(np.float64) > 255.0
            cldf = np.where(inside, np.maximum(cldf, frac[None, :]), cldf)
            clwp = np.where(inside & warm, lw[None, :] / 3.0, clwp)
            ciwp = np.where(inside & ~warm, iw[None, :] / 3.0, ciwp)
            # mixed phase in a band around 255 K
            mixed = inside & (np.abs(tlay.astype(np.float64) - 255.0) < 8.0)
            clwp = np.where(mixed, lw[None, :] / 6.0, clwp)
            ciwp = np.where(mixed, iw[None, :] / 6.0, ciwp)
        rel = 4.0 + 16.0 * _u(seed, col, 70, nlay)
        rei = 15.0 + 105.0 * _u(seed, col, 71, nlay)
    out.update(cldf=cldf.astype(f32), ciwp=ciwp.astype(f32), clwp=clwp.astype(f32),
               rel=rel.astype(f32), rei=rei.astype(f32))

    # ---- aerosols ------------------------------------------------------------------------------------
    tauaer = np.zeros((NBNDLW, nlay, ncol), dtype=f32)
    if aerosol:
        for b in range(NBNDLW):
            t0 = np.exp(np.log(1e-4) + np.log(3e3) * _u(seed, col, 100 + b)[0]) / nlay * 4.0
            ssa = 0.8 + 0.19 * _u(seed, col, 130 + b)[0]
            tauaer[b] = (t0[None, :] * sig ** 2 * (1.0 - ssa[None, :])).astype(f32)   # absorption tau
    out["tauaer"] = tauaer

    # ---- surface / sun (SW) ---------------------------------------------------------------------------
    a_vis = 0.05 + 0.25 * _u(seed, col, 200)[0]
    a_nir = 0.10 + 0.40 * _u(seed, col, 201)[0]
    out.update(asdir=a_vis.astype(f32), asdif=a_vis.astype(f32), aldir=a_nir.astype(f32), aldif=a_nir.astype(f32))
    cz = 0.05 + 0.95 * _u(seed, col, 202)[0]
    out["coszen"] = cz.astype(f32) if lit else np.where(_u(seed, col, 203)[0] < 0.5, cz, 0.0).astype(f32)
    if aerosol:
        tsw = np.zeros((NBNDSW, nlay, ncol), dtype=f32); ssw = np.zeros_like(tsw); gsw = np.zeros_like(tsw)
        for b in range(NBNDSW):
            t0 = np.exp(np.log(1e-4) + np.log(3e3) * _u(seed, col, 300 + b)[0]) / nlay * 4.0
            tsw[b] = (t0[None, :] * sig ** 2).astype(f32)
            ssw[b] = (0.8 + 0.19 * _u(seed, col, 330 + b)[0])[None, :].astype(f32)
            gsw[b] = (0.5 + 0.3 * _u(seed, col, 360 + b)[0])[None, :].astype(f32)
        out.update(tauaer_sw=tsw, ssaaer_sw=ssw, asmaer_sw=gsw)

    # pressure super-layer interfaces (700 / 400 hPa; IRR:1819-1820), RRTMG (bottom-up, 1-based) sense
    pref = plev[:, 0].astype(np.float64)
    out["cloudLM"] = np.int32(max(1, int(np.sum(pref[1:] >= 700.0))))
    out["cloudMH"] = np.int32(max(int(out["cloudLM"]) + 1, int(np.sum(pref[1:] >= 400.0))))
    out["dyofyr"] = np.int32(180)
    return out


def chou_lw_inputs(inp, aerosol=False):
    """Inputs of the Chou-Suarez `irrad` for the columns of `make_columns` (the unit conversions and the vertical flip that
    GEOS_IrradGridComp does before the call, GEOS_IrradGridComp.F90:1870-2101): layers from the TOP down, pressure in Pa,
    specific humidity, O3 mass mixing ratio, hydrometeor mixing ratios of the 4 species (ice, liquid, rain, snow).
    Arrays are numpy C-order with the reversed Fortran shape, e.g. ple (np+1, m), cwc (4, np, m), eg (10, ns, m)."""
    f32 = np.float32
    nlay, m = inp["play"].shape
    flip = lambda a: np.ascontiguousarray(a[::-1])
    ple = (flip(inp["plev"]).astype(np.float64) * 100.0)                       # Pa, top -> surface
    dp = np.diff(ple, axis=0)
    w = flip(inp["h2ovmr"]).astype(np.float64) * (18.016 / 28.966)             # mass mixing ratio
    grav = 9.80665
    out = dict(
        ple=ple.astype(f32), ta=flip(inp["tlay"]), wa=(w / (1.0 + w)).astype(f32),
        oa=(flip(inp["o3vmr"]).astype(np.float64) * (47.998 / 28.966)).astype(f32), tb=inp["tlev"][0].astype(f32),
        co2=float(inp["co2vmr"][0, 0]), n2o=flip(inp["n2ovmr"]), ch4=flip(inp["ch4vmr"]), cfc11=flip(inp["cfc11vmr"]),
        cfc12=flip(inp["cfc12vmr"]), cfc22=flip(inp["cfc22vmr"]), fcld=flip(inp["cldf"]),
    )
    # in-cloud water paths (g m-2) -> grid-mean mixing ratios (kg/kg): wp = cwc * dp / g * 1e3 (getirtau.code)
    cf = flip(inp["cldf"]).astype(np.float64)
    ice = flip(inp["ciwp"]).astype(np.float64) * cf * grav / (dp * 1.0e3)
    liq = flip(inp["clwp"]).astype(np.float64) * cf * grav / (dp * 1.0e3)
    zero = np.zeros_like(ice)
    out["cwc"] = np.stack([ice, liq, 0.05 * liq, 0.05 * ice]).astype(f32)
    out["reff"] = np.stack([flip(inp["rei"]), flip(inp["rel"]), np.full_like(zero, 100.0), np.full_like(zero, 140.0)]).astype(f32)
    # level indices separating high/middle (400 hPa) and middle/low (700 hPa) clouds, top-down numbering
    pref = ple.mean(axis=1) * 0.01
    out["ict"] = int(max(2, np.sum(pref < 400.0)))
    out["icb"] = int(max(out["ict"] + 1, np.sum(pref < 700.0)))
    out["ns"] = 1
    out["fs"] = np.ones((1, m), dtype=f32)
    out["tg"] = inp["tsfc"].reshape(1, m).astype(f32)
    out["tv"] = out["tg"].copy()
    out["eg"] = np.broadcast_to(inp["emis"][:10].reshape(10, 1, m), (10, 1, m)).astype(f32).copy()
    out["ev"] = np.zeros((10, 1, m), dtype=f32)
    out["rv"] = np.zeros((10, 1, m), dtype=f32)
    out["nb"] = 10
    if aerosol and "tauaer_sw" in inp:
        tau = inp["tauaer_sw"][:10, ::-1].astype(np.float64)
        ssa = inp["ssaaer_sw"][:10, ::-1].astype(np.float64)
        g = inp["asmaer_sw"][:10, ::-1].astype(np.float64)
        out["na"] = 3
        out["taua"] = np.ascontiguousarray(tau).astype(f32)
        out["ssaa"] = np.ascontiguousarray(tau * ssa).astype(f32)          # the reference takes tau, tau*ssa, tau*ssa*g
        out["asya"] = np.ascontiguousarray(tau * ssa * g).astype(f32)
    else:
        out["na"] = 0
        out["taua"] = np.zeros((10, nlay, m), dtype=f32); out["ssaa"] = np.zeros_like(out["taua"]); out["asya"] = np.zeros_like(out["taua"])
    return out


def chou_sw_inputs(inp, aerosol=False):
    """Inputs of the Chou-Suarez `sorad` for the columns of `make_columns` (same conventions as chou_lw_inputs; pl in hPa;
    hk_uv / hk_ir = the band weights the GridComp passes, here the defaults of sorad_constants)."""
    import os
    from . import _lib
    from .tableblob import read_blob
    f32 = np.float32
    ch = chou_lw_inputs(inp)
    nlay, m = inp["play"].shape
    _, t = read_blob(os.path.join(_lib.DATA, "chou_sw_r8.grtb"))
    out = dict(cosz=np.maximum(inp["coszen"], 1e-4).astype(f32), pl=(ch["ple"].astype(np.float64) * 0.01).astype(f32), ta=ch["ta"], wa=ch["wa"],
               oa=ch["oa"], co2=ch["co2"], cwc=ch["cwc"], fcld=ch["fcld"], ict=ch["ict"], icb=ch["icb"], reff=ch["reff"],
               hk_uv=np.asarray(t["hk_uv_old"], dtype=np.float64), hk_ir=np.ascontiguousarray(np.asarray(t["hk_ir_old"], dtype=np.float64).T),
               rsuvbm=inp["asdir"], rsuvdf=inp["asdif"], rsirbm=inp["aldir"], rsirdf=inp["aldif"], nb=8)
    if aerosol and "tauaer_sw" in inp:
        tau = inp["tauaer_sw"][:8, ::-1].astype(np.float64)
        ssa = inp["ssaaer_sw"][:8, ::-1].astype(np.float64)
        g = inp["asmaer_sw"][:8, ::-1].astype(np.float64)
        out["taua"] = np.ascontiguousarray(tau).astype(f32)
        out["ssaa"] = np.ascontiguousarray(tau * ssa).astype(f32)
        out["asya"] = np.ascontiguousarray(tau * ssa * g).astype(f32)
    else:
        out["taua"] = np.zeros((8, nlay, m), dtype=f32); out["ssaa"] = np.zeros_like(out["taua"]); out["asya"] = np.zeros_like(out["taua"])
    return out


def geos_chou_sw_fields(inp, aerosol=True, undef_every=7):
    """GEOS-side fields (gridcomp.SWC_IN; model ordering, SI units, [k][ij]) that lead the Chou-Suarez branch of SORADCORE
    (GEOS_SolarGridComp.F90:4484-4553) back to `chou_sw_inputs(inp)`: odd oxygen as a VOLUME mixing ratio, effective radii in metres with
    every `undef_every`-th cell MAPL_UNDEF (the GridComp then substitutes 36 / 14 / 50 / 50 microns).  float64 numpy; plus the
    scalars sorad needs."""
    from . import gridcomp as G
    cs = chou_sw_inputs(inp, aerosol=aerosol)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    f = {"PLE": f64(cs["pl"]) * 100.0, "T": f64(cs["ta"]), "Q": f64(cs["wa"]), "OX": f64(cs["oa"]) * (G.MAPL["AIRMW"] / G.MAPL["O3MW"]),
         "CL": f64(cs["fcld"]), "ZT": f64(cs["cosz"]), "ALBVR": f64(cs["rsuvbm"]), "ALBVF": f64(cs["rsuvdf"]), "ALBNR": f64(cs["rsirbm"]),
         "ALBNF": f64(cs["rsirdf"])}
    for s, (q, r) in enumerate((("QI", "RI"), ("QL", "RL"), ("QR", "RR"), ("QS", "RS"))):
        f[q] = f64(cs["cwc"][s])
        rr = f64(cs["reff"][s]) * 1.0e-6
        if undef_every:
            flat = rr.reshape(-1)
            flat[s::undef_every] = G.MAPL["UNDEF"]
        f[r] = rr
    if aerosol:
        f["TAUA"] = f64(cs["taua"]); f["SSAA"] = f64(cs["ssaa"]); f["ASYA"] = f64(cs["asya"])
    f["LCLDMH"] = int(cs["ict"]); f["LCLDLM"] = int(cs["icb"]); f["CO2"] = float(cs["co2"])
    f["HK_UV"] = cs["hk_uv"]; f["HK_IR"] = cs["hk_ir"]
    return f


def geos_lw_fields(inp):
    """GEOS-side (model ordering, 1 = top, SI units) fields of gridcomp.LWD_IN that lead LW_Driver's prep
    (GEOS_IrradGridComp.F90:3243-3371) back to (nearly) the RRTMG-side columns `inp` of make_columns; plus the model-ordering
    super-layer interface indices.  float64 numpy; [k][ij]."""
    from . import gridcomp as G
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    flip = lambda a: f64(a)[::-1].copy()
    lm = inp["play"].shape[0]
    f = {}
    f["PLE"] = flip(inp["plev"]) * 100.0
    f["PL"] = flip(inp["play"]) * 100.0
    f["T"] = flip(inp["tlay"])
    x = flip(inp["h2ovmr"]) * (G.MAPL["H2OMW"] / G.MAPL["AIRMW"])
    f["Q"] = x / (1.0 + x)
    f["O3"] = flip(inp["o3vmr"]) * (G.MAPL["O3MW"] / G.MAPL["AIRMW"])
    f["CH4"] = flip(inp["ch4vmr"]); f["N2O"] = flip(inp["n2ovmr"]); f["CO2_3D"] = None
    f["CFC11"] = flip(inp["cfc11vmr"]); f["CFC12"] = flip(inp["cfc12vmr"]); f["HCFC22"] = flip(inp["cfc22vmr"])
    f["FCLD"] = flip(inp["cldf"])
    dp = f["PLE"][1:] - f["PLE"][:-1]
    f["CWC_LIQ"] = flip(inp["clwp"]) / (1.02 * 100 * dp)
    f["CWC_ICE"] = flip(inp["ciwp"]) / (1.02 * 100 * dp)
    f["REFF_LIQ"] = flip(inp["rel"]); f["REFF_ICE"] = flip(inp["rei"])
    ta = f64(inp["tauaer"])[:, ::-1, :]
    f["TAUA"] = 10.0 * ta; f["SSAA"] = 9.0 * ta            # extinction and scattering: absorption = TAUA - SSAA = tauaer
    f["TS"] = f64(inp["tsfc"]); f["EMIS"] = f64(inp["emis"])[0]; f["LATS"] = f64(inp["alat"]); f["T2M"] = f64(inp["tlev"])[0]
    # RRTMG's cloudLM = LM - LCLDLM + 1 (IRR:3238-3239)
    f["LCLDLM"] = lm - int(inp["cloudLM"]) + 1; f["LCLDMH"] = lm - int(inp["cloudMH"]) + 1
    return f


def geos_sw_fields(inp):
    """same for SORADCORE's RRTMG branch (GEOS_SolarGridComp.F90:6113-6219): gridcomp.SWD_IN on packed daytime columns, aerosol
    triplet in the un-normalised (tau, tau*ssa, tau*ssa*g) form the aerosol bundle delivers."""
    from . import gridcomp as G
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    flip = lambda a: f64(a)[::-1].copy()
    lm = inp["play"].shape[0]
    f = {}
    f["PLE"] = flip(inp["plev"]) * 100.0
    f["PL"] = flip(inp["play"]) * 100.0
    f["T"] = flip(inp["tlay"])
    x = flip(inp["h2ovmr"]) * (G.MAPL["H2OMW"] / G.MAPL["AIRMW"])
    f["Q"] = x / (1.0 + x)
    f["O3"] = flip(inp["o3vmr"]) * (G.MAPL["O3MW"] / G.MAPL["AIRMW"])
    f["CH4"] = flip(inp["ch4vmr"]); f["CL"] = flip(inp["cldf"]); f["TS"] = f64(inp["tsfc"])
    dp = f["PLE"][1:] - f["PLE"][:-1]
    f["QQ_ICE"] = flip(inp["ciwp"]) / (1.02 * 100 * dp); f["QQ_LIQ"] = flip(inp["clwp"]) / (1.02 * 100 * dp)
    f["RR_ICE"] = flip(inp["rei"]); f["RR_LIQ"] = flip(inp["rel"])
    ta = f64(inp["tauaer_sw"])[:, ::-1, :]; ss = f64(inp["ssaaer_sw"])[:, ::-1, :]; g = f64(inp["asmaer_sw"])[:, ::-1, :]
    f["TAUA"] = ta.copy(); f["SSAA"] = ta * ss; f["ASYA"] = ta * ss * g
    f["ZT"] = f64(inp["coszen"]); f["ALAT"] = f64(inp["alat"])
    f["ALBVR"] = f64(inp["asdir"]); f["ALBVF"] = f64(inp["asdif"]); f["ALBNR"] = f64(inp["aldir"]); f["ALBNF"] = f64(inp["aldif"])
    f["LCLDLM"] = lm - int(inp["cloudLM"]) + 1; f["LCLDMH"] = lm - int(inp["cloudMH"]) + 1
    return f
