"""tests/golden/make_golden.py -- regenerate the committed golden vectors by RUNNING THE REFERENCE
(oracle/_ref = the reference's Fortran compiled unmodified from /root/reference, see oracle/Makefile).

    python -m tests.golden.make_golden

Each .npz holds the generator parameters (inputs are re-created bit-identically by
geosradiation_gridcomp_amd.synth.make_columns), the reference outputs in both precisions (r4 = GEOS
default real, r8 = -fdefault-real-8) and a few intermediates.  Data only: no reference source text.
"""
import os
import numpy as np
from geosradiation_gridcomp_amd import synth
from oracle import reflib

HERE = os.path.dirname(os.path.abspath(__file__))
FLUX = ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs")

CASES = {
    # name: (make_columns kwargs, ih, band_output?)
    "lw_clear_72": (dict(ncol=16, nlay=72, aerosol=False, cloudy_frac=0.0), 0, False),
    "lw_aer_72": (dict(ncol=16, nlay=72, aerosol=True, cloudy_frac=0.0), 0, True),
    "lw_cloudy_ih1_72": (dict(ncol=24, nlay=72, aerosol=True, cloudy_frac=0.8), 1, True),
    "lw_cloudy_ih0_72": (dict(ncol=16, nlay=72, aerosol=False, cloudy_frac=0.8, start=1000), 0, False),
    "lw_cloudy_ih2_137": (dict(ncol=8, nlay=137, aerosol=True, cloudy_frac=0.9, start=5000), 2, False),
}


SW_CASES = {
    # name: (make_columns kwargs, ih)
    "sw_stages_72": (dict(ncol=6, nlay=72, aerosol=True, cloudy_frac=0.9, start=300), 1),
    "sw_stages_137": (dict(ncol=4, nlay=137, aerosol=False, cloudy_frac=0.9, start=7000), 2),
}
SW_SCON = 1361.0
SW_INDSOLVAR = (0.1580, 80.0)
SW_BNDSCL = np.linspace(0.9, 1.1, 14)


def sw_solar_scalars(isolvar, dt, blob, scon=SW_SCON, indsolvar=SW_INDSOLVAR, bndscl=SW_BNDSCL):
    """svar (f,s,i) and svar_bnd(3,29) exactly as the driver forms them (SW/rrtmg_sw_rad.F90:893-1127) in precision dt."""
    R = dt
    Fint, Sint, Iint = R(blob["Fint"]), R(blob["Sint"]), R(blob["Iint"])
    svar = [R(1), R(1), R(1)]
    sb = np.ones((3, 29), dtype=dt)
    scon = R(scon)
    scon_int = R(R(Fint + Sint) + Iint)
    if isolvar == 0:
        svar = [R(scon / scon_int)] * 3
    elif isolvar == 2:
        f = R(R(R(indsolvar[0]) - R(blob["Mg_0"])) / R(R(blob["Mg_avg"]) - R(blob["Mg_0"])))
        s_ = R(R(R(indsolvar[1]) - R(blob["SB_0"])) / R(R(blob["SB_avg"]) - R(blob["SB_0"])))
        i_ = R(R(scon - R(R(f * Fint) + R(s_ * Sint))) / Iint)
        svar = [f, s_, i_]
    elif isolvar == 3:
        for b in range(16, 30):
            sb[:, b - 1] = R(R(scon / scon_int) * R(bndscl[b - 16]))
    return svar, sb


def main_sw():
    from geosradiation_gridcomp_amd.tableblob import read_blob
    from geosradiation_gridcomp_amd import _lib
    for name, (kw, ih) in SW_CASES.items():
        inp = synth.make_columns(**kw)
        out = {"kw_json": np.array(repr(kw)), "ih": np.int32(ih), "scon": np.float64(SW_SCON), "indsolvar": np.array(SW_INDSOLVAR),
               "bndscl": SW_BNDSCL}
        for kind in ("r4", "r8"):
            dt = reflib.dtype_of(kind)
            _, blob = read_blob(os.path.join(_lib.DATA, f"rrtmg_sw_{kind}.grtb"))
            # setcoef_sw + taumol_sw: optical depths once, solar source for every isolvar GEOS accepts
            for isol in (0, -1, 2, 3):
                svar, sb = sw_solar_scalars(isol, dt, blob)
                t = reflib.sw_setcoef_taumol(inp, isolvar=isol, svar=svar, svar_bnd=sb, kind=kind)
                if isol == 0:
                    out[f"{kind}_taug"] = t["taug"]; out[f"{kind}_taur"] = t["taur"]; out[f"{kind}_laytrop"] = t["laytrop"]
                out[f"{kind}_ssi_isolvar{isol}".replace("-", "m")] = t["sfluxzen"] if isol < 0 else t["ssi"]
            # McICA sub-columns with the SW seeding, then cldprmc_sw for every ice parameterisation (first 2 columns)
            reflib.set_inhomogeneity(ih, kind)
            n2 = 2
            sub = {k: (v[..., :n2] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == kw["ncol"] else v) for k, v in inp.items()}
            cl, ci_s, cl_s = reflib.mcica(sub["zm"], sub["alat"], int(inp["dyofyr"]), sub["play"], sub["cldf"], sub["ciwp"], sub["clwp"], 112,
                                          seed_order=(4, 3, 2, 1), kind=kind)
            reflib.set_inhomogeneity(0, kind)
            out[f"{kind}_mc_cldy"] = cl.astype(np.uint8); out[f"{kind}_mc_ciwp"] = ci_s; out[f"{kind}_mc_clwp"] = cl_s
            for iceflag in (1, 2, 3, 4):
                r = reflib.sw_cldprmc(cl, ci_s, cl_s, sub["rei"], sub["rel"], iceflag=iceflag, kind=kind)
                for nm, a in zip(("taormc", "taucmc", "ssacmc", "asmcmc"), r):
                    out[f"{kind}_ice{iceflag}_{nm}"] = a
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")))


# isolvar = 1 (rrtmg_sw_rad.F90:906-930,994-1008,1060-1079): the NRLSSI2 host routines it calls (NRLSSI2.F90:160-332, no ESMF/MAPL
# dependency) run from the reference over a grid of cycle positions x amplitude scalings, and the spectral solar source taumol_sw
# returns for the resulting svar_f / svar_s / svar_i on the columns of sw_stages_72
NRL_INDSOLVAR = [(1.0, 1.0), (1.2, 0.8), (0.5, 1.0), (1.0, 1.7), (2.0, 0.3), (0.9, 1.1), (1.0001, 0.9999), (3.0, 3.0)]
NRL_CASES = [(0.30, (1.15, 0.9), 1361.0), (0.0189, (1.2, 0.8), 1361.0), (0.85, None, 1360.0), (0.5, (1.0, 1.0), 0.0), (0.0, (0.7, 1.3), 1365.5),
             (1.0, (1.0, 1.7), 1361.0), (0.375, (2.0, 0.3), 0.0)]


def nrl_fracs(dt):
    il = dt(1.0) / dt(132); ilh = dt(0.5) * il
    rng = np.random.default_rng(1360)
    f = [dt(0), dt(1), ilh, dt(1) - ilh, dt(0.0189), dt(0.3750), np.nextafter(dt(0.0189), dt(0)), np.nextafter(dt(0.3750), dt(1)),
         np.nextafter(dt(0), dt(1)), np.nextafter(dt(1), dt(0))]
    f += [dt(n - 2) * il + ilh for n in (2, 3, 17, 66, 67, 131, 132, 133)]
    f += [np.nextafter(dt(n - 2) * il + ilh, dt(d)) for n in (2, 40, 99, 133) for d in (0, 1)]
    f += list(rng.uniform(0, 1, 200 - len(f)).astype(dt))
    return np.array(f, dtype=dt)


def isolvar1_svar(kind, scon, solcycfrac, indsolvar, blob, get=None):
    """svar_f, svar_s, svar_i as the driver's isolvar = 1 branch forms them (rrtmg_sw_rad.F90:906-930,994-1008,1060-1079) in the
    kind's precision from the NRLSSI2 routines' results (`get` = reflib by default)."""
    get = get or reflib
    R = reflib.dtype_of(kind)
    fr = R(solcycfrac)
    scl = np.ones(2, dtype=R)
    if indsolvar is not None and (R(indsolvar[0]) != 1 or R(indsolvar[1]) != 1):
        scl = get.nrlssi2_adjust(fr, indsolvar, kind)
    mf, ms = get.nrlssi2_means(indsolvar, kind)
    mg, sb = get.nrlssi2_interp(fr, kind)
    Mg0, Mga, SB0, SBa = (R(blob[k]) for k in ("Mg_0", "Mg_avg", "SB_0", "SB_avg"))
    Fint, Sint, Iint = R(blob["Fint"]), R(blob["Sint"]), R(blob["Iint"])
    f = R(R(scl[0] * R(mg - Mg0)) / R(Mga - Mg0))
    s_ = R(R(scl[1] * R(sb - SB0)) / R(SBa - SB0))
    i_ = R(1) if scon == 0 else R(R(R(scon) - R(R(mf * Fint) + R(ms * Sint))) / Iint)
    return np.array([f, s_, i_], dtype=R)


def main_nrlssi2():
    from geosradiation_gridcomp_amd.tableblob import read_blob
    from geosradiation_gridcomp_amd import _lib
    kw, _ = SW_CASES["sw_stages_72"]
    inp = synth.make_columns(**kw)
    out = {"indsolvar": np.array(NRL_INDSOLVAR), "kw_json": np.array(repr(kw)),
           "case_solcycfrac": np.array([c[0] for c in NRL_CASES]), "case_scon": np.array([c[2] for c in NRL_CASES]),
           "case_indsolvar": np.array([(np.nan, np.nan) if c[1] is None else c[1] for c in NRL_CASES])}
    for kind in ("r4", "r8"):
        dt = reflib.dtype_of(kind)
        _, blob = read_blob(os.path.join(_lib.DATA, f"rrtmg_sw_{kind}.grtb"))
        fr = nrl_fracs(dt)
        out[f"{kind}_solcycfr"] = fr
        out[f"{kind}_MgSB"] = np.array([reflib.nrlssi2_interp(f, kind) for f in fr], dtype=dt)
        out[f"{kind}_scl"] = np.array([[reflib.nrlssi2_adjust(f, ind, kind) for f in fr] for ind in NRL_INDSOLVAR], dtype=dt)
        out[f"{kind}_means"] = np.array([reflib.nrlssi2_means(ind, kind) for ind in NRL_INDSOLVAR], dtype=dt)
        out[f"{kind}_means_absent"] = np.array(reflib.nrlssi2_means(None, kind), dtype=dt)
        sv = np.array([isolvar1_svar(kind, scon, f, ind, blob) for f, ind, scon in NRL_CASES], dtype=dt)
        out[f"{kind}_case_svar"] = sv
        # the spectral solar source of taumol_sw (isolvar 1 shares the isolvar <= 2 formula, rrtmg_sw_taumol.F90:325-347) for case 0
        t = reflib.sw_setcoef_taumol(inp, isolvar=1, svar=list(sv[0]), svar_bnd=np.ones((3, 29), dtype=dt), kind=kind)
        out[f"{kind}_ssi_case0"] = t["ssi"]
    np.savez_compressed(os.path.join(HERE, "nrlssi2_isolvar1.npz"), **out)
    print("nrlssi2_isolvar1", os.path.getsize(os.path.join(HERE, "nrlssi2_isolvar1.npz")))


# LW cldprmc with the ice parameterisations GEOS does not default to (rrtmg_lw_cldprmc.F90:138-226,270-316): one batch of cloudy
# columns, the fluxes of the reference for each iceflag (the generator keeps rei inside every parameterisation's valid range)
ICE_KW = dict(ncol=16, nlay=72, aerosol=True, cloudy_frac=1.0, start=4400)
ICE_FLAGS = (0, 1, 2, 4)


def main_iceflags():
    inp = synth.make_columns(**ICE_KW)
    out = {"kw_json": np.array(repr(ICE_KW)), "ih": np.int32(1)}
    for kind in ("r4", "r8"):
        reflib.set_inhomogeneity(1, kind)
        for ice in ICE_FLAGS:
            r = reflib.rrtmg_lw(inp, kind, iceflg=ice)
            for k in FLUX:
                out[f"{kind}_ice{ice}_{k}"] = r[k]
            out[f"{kind}_ice{ice}_clearCounts"] = r["clearCounts"]
        reflib.set_inhomogeneity(0, kind)
    np.savez_compressed(os.path.join(HERE, "lw_iceflags_72.npz"), **out)
    print("lw_iceflags_72", os.path.getsize(os.path.join(HERE, "lw_iceflags_72.npz")))


def main():
    for name, (kw, ih, bo) in CASES.items():
        inp = synth.make_columns(**kw)
        out = {"kw_json": np.array(repr(kw)), "ih": np.int32(ih)}
        band_output = np.ones(16, dtype=np.int32) if bo else None
        for kind in ("r4", "r8"):
            reflib.set_inhomogeneity(ih, kind)
            r = reflib.rrtmg_lw(inp, kind, band_output=band_output)
            for k in FLUX:
                out[f"{kind}_{k}"] = r[k]
            out[f"{kind}_clearCounts"] = r["clearCounts"]
            if bo:
                out[f"{kind}_olrb"] = r["olrb"]; out[f"{kind}_dolrb_dTs"] = r["dolrb_dTs"]
            # intermediates for the first 2 columns
            sub = {k: (v[..., :2] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == kw["ncol"] else v) for k, v in inp.items()}
            t = reflib.lw_setcoef_taumol(sub, kind)
            out[f"{kind}_taug2"] = t["taug"]; out[f"{kind}_pfracs2"] = t["pfracs"]
            # McICA sub-columns of the first 4 columns (LW seeding + SW-like seeding with 112 sub-columns)
            sub4 = {k: (v[..., :4] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == kw["ncol"] else v) for k, v in inp.items()}
            for tag, nsub, so in (("lw", 140, (1, 2, 3, 4)), ("sw", 112, (4, 3, 2, 1))):
                cl, ci_s, cl_s = reflib.mcica(sub4["zm"], sub4["alat"], int(inp["dyofyr"]), sub4["play"], sub4["cldf"], sub4["ciwp"],
                                              sub4["clwp"], nsub, seed_order=so, kind=kind)
                out[f"{kind}_mc_{tag}_cldy"] = cl.astype(np.uint8)
                out[f"{kind}_mc_{tag}_ciwp"] = ci_s; out[f"{kind}_mc_{tag}_clwp"] = cl_s
            reflib.set_inhomogeneity(0, kind)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    import sys
    if "ice" in sys.argv[1:]:
        main_iceflags()
        sys.exit(0)
    if "nrlssi2" in sys.argv[1:]:
        main_nrlssi2()
        sys.exit(0)
    if "sw" not in sys.argv[1:]:
        main()
        main_iceflags()
    if "lw" not in sys.argv[1:]:
        main_sw()
        main_nrlssi2()
