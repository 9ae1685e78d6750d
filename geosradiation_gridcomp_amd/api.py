"""Host-side mirror of the reference's solver interface for the hot path, over the C ABI.

Names, argument order and meaning follow the reference's Fortran entry points so parity tests read like
calls into the reference:

    rrtmg_lw_ini()                          rrtmg_lw_init.F90:22
    rrtmg_lw(ncol, nlay, psize, dudTs, ...) rrtmg_lw_rad.F90:15-23
    rrtmg_sw_ini()                          SW rrtmg_sw_init.F90:23
    rrtmg_sw(rpart, ncol, nlay, scon, ...)  SW rrtmg_sw_rad.F90:68-124
    irrad(m, np, ple, ta, wa, oa, tb, ...)  GEOSirrad_GridComp/irrad.F90:27-35 (Chou-Suarez LW)
    sorad(m, np, nb, cosz, pl, ta, ...)     GEOSsolar_GridComp/sorad.F90:43-51 (Chou-Suarez SW)
    generate_stochastic_clouds(...)         cloud_subcol_gen.F90:132-137
    clearCounts_threeBand(...)              cloud_subcol_gen.F90:611-614
    set_inhomogeneity / unset_inhomogeneity cloud_condensate_inhomogeneity.F90:45,75
    initialize_cloud_subcol_gen(...)        cloud_subcol_gen.F90:109-111

Array convention: numpy C-order arrays whose reversed shape is the Fortran shape (Fortran
``play(ncol,nlay)`` <-> numpy ``(nlay, ncol)``), i.e. byte-identical to what the Fortran caller passes.
Where the reference would ``error stop`` a ``GeosradInputError`` carrying the reference's message is raised.
"""
import ctypes
import os
import numpy as np
from . import _lib

NBNDLW = 16
NGPTLW = 140
NBNDSW = 14
NGPTSW = 112
_SW_GAS = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "o2vmr"]
_SW_COT = ["cotdtp", "cotdhp", "cotdmp", "cotdlp", "cotntp", "cotnhp", "cotnmp", "cotnlp"]

_IN2D = ["h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr", "ccl4vmr",
         "cldf", "ciwp", "clwp", "rei", "rel"]


class GeosradError(RuntimeError):
    pass


class GeosradInputError(GeosradError):
    """The reference would `error stop` on these inputs."""


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class Context:
    """One (GPU, precision) instance of the solver: replaces the reference's module-level state."""

    def __init__(self, real_kind=4, device=0, tables=True, devices=None):
        """device: HIP device index, or -1 = the library's choice for this process (GEOSRAD_DEVICE, else the launcher's node-local MPI
        rank modulo the device count: geosrad_pick_device).  devices = [ids]: a multi-device context (geosrad_create_multi) - the
        host-array entry points then split their columns into one contiguous shard per entry, processed concurrently."""
        self.L = _lib.lib()
        self.real_kind = int(real_kind)
        self.dtype = np.float32 if self.real_kind == 4 else np.float64
        self.h = ctypes.c_void_p()
        if devices is not None:
            ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            rc = self.L.geosrad_create_multi(ctypes.byref(self.h), ids, len(devices), self.real_kind)
        else:
            rc = self.L.geosrad_create(ctypes.byref(self.h), int(device), self.real_kind)
        if rc:
            raise GeosradError({2: "no usable HIP device (the product has no CPU fallback)"}.get(rc, f"geosrad_create rc={rc}"))
        if tables:
            self.rrtmg_lw_ini()
            self.rrtmg_sw_ini()
            self.irrad_ini()
            self.sorad_ini()

    def close(self):
        if self.h:
            self.L.geosrad_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            msg = self.L.geosrad_last_error(self.h).decode()
            raise (GeosradInputError if rc == 5 else GeosradError)(msg)

    def _kind(self):
        return "r4" if self.real_kind == 4 else "r8"

    # ---- initialisation ------------------------------------------------------------------------------
    def rrtmg_lw_ini(self, path=None):
        path = path or os.path.join(_lib.DATA, f"rrtmg_lw_{self._kind()}.grtb")
        self._chk(self.L.geosrad_load_tables_lw(self.h, os.fsencode(path)))

    def rrtmg_sw_ini(self, path=None):
        path = path or os.path.join(_lib.DATA, f"rrtmg_sw_{self._kind()}.grtb")
        self._chk(self.L.geosrad_load_tables_sw(self.h, os.fsencode(path)))

    def irrad_ini(self, path=None):
        """uploads irrad_constants / rad_constants (the reference keeps them as module data: no init routine there)"""
        path = path or os.path.join(_lib.DATA, f"chou_lw_{self._kind()}.grtb")
        self._chk(self.L.geosrad_load_tables_chou_lw(self.h, os.fsencode(path)))

    def sorad_ini(self, path=None):
        path = path or os.path.join(_lib.DATA, f"chou_sw_{self._kind()}.grtb")
        self._chk(self.L.geosrad_load_tables_chou_sw(self.h, os.fsencode(path)))

    def set_inhomogeneity(self, ih, path=None):
        if ih and path is None:
            path = os.path.join(_lib.DATA, f"xcw_{'beta' if ih == 1 else 'gamma'}_{self._kind()}.grtb")
        self._chk(self.L.geosrad_load_inhomogeneity(self.h, int(ih), os.fsencode(path) if path else None))

    def unset_inhomogeneity(self):
        self.set_inhomogeneity(0)

    def initialize_cloud_subcol_gen(self, adl=None, rdl=None):
        a = (ctypes.c_double * 4)(*adl) if adl is not None else None
        r = (ctypes.c_double * 4)(*rdl) if rdl is not None else None
        self._chk(self.L.geosrad_set_corr_lengths(self.h, a, r))

    def set_chunk(self, n):
        self._chk(self.L.geosrad_set_chunk(self.h, int(n)))

    def workspace_bytes(self):
        return int(self.L.geosrad_workspace_bytes(self.h))

    # ---- RRTMG_LW, host arrays -------------------------------------------------------------------------
    def rrtmg_lw(self, ncol, nlay, psize, dudTs, play, plev, tlay, tlev, tsfc, emis,
                 h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr,
                 cldf, ciwp, clwp, rei, rel, iceflglw, liqflglw, tauaer, zm, alat, dyofyr, cloudLM, cloudMH,
                 band_output=None, out=None):
        """Returns dict(uflx,dflx,uflxc,dflxc,duflx_dTs,duflxc_dTs (nlay+1,ncol); clearCounts (4,ncol);
        olrb,dolrb_dTs (ncol,16))."""
        dt = self.dtype
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        args2 = [c(x) for x in (h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr,
                                cldf, ciwp, clwp, rei, rel)]
        play, plev, tlay, tlev, tsfc, emis, tauaer, zm, alat = map(c, (play, plev, tlay, tlev, tsfc, emis, tauaer, zm, alat))
        assert play.shape == (nlay, ncol) and plev.shape == (nlay + 1, ncol)
        if out is None:       # (a caller that keeps its output arrays, like a Fortran caller does, passes the dict of an earlier call)
            out = {k: np.zeros((nlay + 1, ncol), dtype=dt) for k in ("uflx", "dflx", "uflxc", "dflxc", "duflx_dTs", "duflxc_dTs")}
            out["clearCounts"] = np.zeros((4, ncol), dtype=np.int32)
            out["olrb"] = np.zeros((ncol, NBNDLW), dtype=dt)
            out["dolrb_dTs"] = np.zeros((ncol, NBNDLW), dtype=dt)
        bo = np.zeros(NBNDLW, dtype=np.int32) if band_output is None else np.ascontiguousarray(band_output, dtype=np.int32)
        ci = ctypes.c_int
        rc = self.L.geosrad_rrtmg_lw(
            self.h, ci(ncol), ci(nlay), ci(psize), ci(1 if dudTs else 0), _p(play), _p(plev), _p(tlay), _p(tlev), _p(tsfc),
            _p(emis), *[_p(a) for a in args2[:10]], *[_p(a) for a in args2[10:]], ci(iceflglw), ci(liqflglw), _p(tauaer),
            _p(zm), _p(alat), ci(int(dyofyr)), ci(int(cloudLM)), ci(int(cloudMH)), _p(out["clearCounts"]), _p(out["uflx"]),
            _p(out["dflx"]), _p(out["uflxc"]), _p(out["dflxc"]), _p(out["duflx_dTs"]), _p(out["duflxc_dTs"]), _p(bo),
            _p(out["olrb"]), _p(out["dolrb_dTs"]))
        self._chk(rc)
        return out

    def rrtmg_lw_columns(self, inp, psize=4, dudTs=True, iceflg=3, liqflg=1, band_output=None, out=None):
        """Convenience: `inp` as produced by synth.make_columns."""
        nlay, ncol = inp["play"].shape
        return self.rrtmg_lw(ncol, nlay, psize, dudTs, inp["play"], inp["plev"], inp["tlay"], inp["tlev"], inp["tsfc"],
                             inp["emis"], *[inp[k] for k in _IN2D], iceflg, liqflg, inp.get("tauaer"), inp["zm"],
                             inp["alat"], inp["dyofyr"], inp["cloudLM"], inp["cloudMH"], band_output=band_output, out=out)

    def rrtmg_lw_taumol(self, inp):
        """(taug, pfracs) numpy (ncol,140,nlay) == Fortran (nlay,140,ncol), as left by the reference's taumol."""
        dt = self.dtype
        nlay, ncol = inp["play"].shape
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        a = {k: c(inp[k]) for k in ["play", "plev", "tlay", "tlev", "tsfc", "emis", "tauaer"] + _IN2D[:10]}
        taug = np.zeros((ncol, NGPTLW, nlay), dtype=dt); pfr = np.zeros_like(taug)
        rc = self.L.geosrad_rrtmg_lw_taumol(self.h, ctypes.c_int(ncol), ctypes.c_int(nlay), _p(a["play"]), _p(a["plev"]),
                                            _p(a["tlay"]), _p(a["tlev"]), _p(a["tsfc"]), _p(a["emis"]),
                                            *[_p(a[k]) for k in _IN2D[:10]], _p(a["tauaer"]), _p(taug), _p(pfr))
        self._chk(rc)
        return taug, pfr

    # ---- RRTMG_SW, host arrays -------------------------------------------------------------------------
    def rrtmg_sw(self, rpart, ncol, nlay, scon, adjes, coszen, isolvar, play, plev, tlay, h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr,
                 iceflgsw, liqflgsw, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer,
                 asdir, asdif, aldir, aldif, cloudLM, cloudMH, normFlx, do_drfband=False, bndscl=None, indsolvar=None, out=None,
                 solcycfrac=None):
        """rrtmg_sw (SW/rrtmg_sw_rad.F90:68).  Returns dict(swuflx,swdflx,swuflxc,swdflxc (nlay+1,ncol); nirr..uvrf,
        cotdtp..cotnlp (ncol); fswband[,drband,dfband] (14,ncol); clearCounts (4,ncol))."""
        dt = self.dtype
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        gases = [c(x) for x in (h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr)]
        cl5 = [c(x) for x in (cld, ciwp, clwp, rei, rel)]
        coszen, play, plev, tlay, zm, alat, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif = map(
            c, (coszen, play, plev, tlay, zm, alat, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif))
        assert play.shape == (nlay, ncol) and plev.shape == (nlay + 1, ncol)
        if out is None:
            out = {k: np.zeros((nlay + 1, ncol), dtype=dt) for k in ("swuflx", "swdflx", "swuflxc", "swdflxc")}
            for k in ["nirr", "nirf", "parr", "parf", "uvrr", "uvrf"] + _SW_COT:
                out[k] = np.zeros(ncol, dtype=dt)
            out["fswband"] = np.zeros((NBNDSW, ncol), dtype=dt)
            if do_drfband:
                out["drband"] = np.zeros((NBNDSW, ncol), dtype=dt); out["dfband"] = np.zeros((NBNDSW, ncol), dtype=dt)
            out["clearCounts"] = np.zeros((4, ncol), dtype=np.int32)
        bs = None if bndscl is None else np.ascontiguousarray(bndscl, dtype=dt)
        ind = None if indsolvar is None else np.ascontiguousarray(indsolvar, dtype=dt)
        scf = None if solcycfrac is None else np.array([solcycfrac], dtype=dt)
        ci, cd = ctypes.c_int, ctypes.c_double
        rc = self.L.geosrad_rrtmg_sw(
            self.h, ci(rpart), ci(ncol), ci(nlay), cd(scon), cd(adjes), _p(coszen), ci(isolvar), _p(play), _p(plev), _p(tlay),
            *[_p(a) for a in gases], ci(iceflgsw), ci(liqflgsw), *[_p(a) for a in cl5], ci(int(dyofyr)), _p(zm), _p(alat), ci(iaer),
            _p(tauaer), _p(ssaaer), _p(asmaer), _p(asdir), _p(asdif), _p(aldir), _p(aldif), ci(int(cloudLM)), ci(int(cloudMH)),
            ci(int(normFlx)), _p(out["clearCounts"]), _p(out["swuflx"]), _p(out["swdflx"]), _p(out["swuflxc"]), _p(out["swdflxc"]),
            *[_p(out[k]) for k in ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf", "fswband")], *[_p(out[k]) for k in _SW_COT],
            ci(1 if do_drfband else 0), _p(out.get("drband")), _p(out.get("dfband")), _p(bs), _p(ind), _p(scf))
        self._chk(rc)
        return out

    def rrtmg_sw_columns(self, inp, scon=1361.0, adjes=1.0, isolvar=0, iceflg=3, liqflg=1, iaer=0, normFlx=0, do_drfband=False,
                         bndscl=None, indsolvar=None, rpart=4, out=None, solcycfrac=None):
        """Convenience: `inp` as produced by synth.make_columns."""
        nlay, ncol = inp["play"].shape
        aer = [inp.get(k) if iaer == 10 else None for k in ("tauaer_sw", "ssaaer_sw", "asmaer_sw")]
        return self.rrtmg_sw(rpart, ncol, nlay, scon, adjes, inp["coszen"], isolvar, inp["play"], inp["plev"], inp["tlay"],
                             *[inp[k] for k in _SW_GAS], iceflg, liqflg, inp["cldf"], inp["ciwp"], inp["clwp"], inp["rei"], inp["rel"],
                             inp["dyofyr"], inp["zm"], inp["alat"], iaer, *aer, inp["asdir"], inp["asdif"], inp["aldir"], inp["aldif"],
                             inp["cloudLM"], inp["cloudMH"], normFlx, do_drfband=do_drfband, bndscl=bndscl, indsolvar=indsolvar, out=out,
                             solcycfrac=solcycfrac)

    def rrtmg_sw_taumol(self, inp, scon=1361.0, isolvar=0, bndscl=None, indsolvar=None, solcycfrac=None):
        """(taug, taur) numpy (ncol,112,nlay) and ssi (ncol,112) as left by the reference's taumol_sw."""
        dt = self.dtype
        nlay, ncol = inp["play"].shape
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        a = {k: c(inp[k]) for k in ["play", "plev", "tlay"] + _SW_GAS}
        taug = np.zeros((ncol, NGPTSW, nlay), dtype=dt); taur = np.zeros_like(taug); ssi = np.zeros((ncol, NGPTSW), dtype=dt)
        bs = None if bndscl is None else np.ascontiguousarray(bndscl, dtype=dt)
        ind = None if indsolvar is None else np.ascontiguousarray(indsolvar, dtype=dt)
        scf = None if solcycfrac is None else np.array([solcycfrac], dtype=dt)
        rc = self.L.geosrad_rrtmg_sw_taumol(self.h, ctypes.c_int(ncol), ctypes.c_int(nlay), ctypes.c_double(scon), ctypes.c_int(isolvar),
                                            _p(a["play"]), _p(a["plev"]), _p(a["tlay"]), *[_p(a[k]) for k in _SW_GAS], _p(bs), _p(ind),
                                            _p(scf), _p(taug), _p(taur), _p(ssi))
        self._chk(rc)
        return taug, taur, ssi

    def rrtmg_sw_cldprmc(self, inp, iceflg=3, liqflg=1):
        """(taucmc, ssacmc, asmcmc) numpy (ncol,112,nlay) == Fortran (nlay,112,ncol): the cloud optics of the solver's own McICA
        sub-columns as the reference's cldprmc_sw leaves them."""
        dt = self.dtype
        nlay, ncol = inp["play"].shape
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        a = {k: c(inp[k]) for k in ["play", "plev", "tlay", "cldf", "ciwp", "clwp", "rei", "rel", "zm", "alat"] + _SW_GAS}
        o = [np.zeros((ncol, NGPTSW, nlay), dtype=dt) for _ in range(3)]
        ci = ctypes.c_int
        rc = self.L.geosrad_rrtmg_sw_cldprmc(self.h, ci(ncol), ci(nlay), _p(a["play"]), _p(a["plev"]), _p(a["tlay"]),
                                             *[_p(a[k]) for k in _SW_GAS], ci(iceflg), ci(liqflg), _p(a["cldf"]), _p(a["ciwp"]),
                                             _p(a["clwp"]), _p(a["rei"]), _p(a["rel"]), ci(int(inp["dyofyr"])), _p(a["zm"]),
                                             _p(a["alat"]), ci(int(inp["cloudLM"])), ci(int(inp["cloudMH"])), *[_p(x) for x in o])
        self._chk(rc)
        return o

    def rrtmg_sw_dev(self, stream, ncol, nlay, scon, adjes, isolvar, ptr, iceflg, liqflg, dyofyr, iaer, cloudLM, cloudMH, normFlx=0,
                     do_drfband=False, bndscl=None, indsolvar=None, rpart=4, solcycfrac=None):
        """`ptr`: dict name -> device address (int) for every argument array of rrtmg_sw (inputs and outputs)."""
        dt = self.dtype
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        bs = None if bndscl is None else np.ascontiguousarray(bndscl, dtype=dt)
        ind = None if indsolvar is None else np.ascontiguousarray(indsolvar, dtype=dt)
        scf = None if solcycfrac is None else np.array([solcycfrac], dtype=dt)
        ci, cd = ctypes.c_int, ctypes.c_double
        rc = self.L.geosrad_rrtmg_sw_dev(
            self.h, ctypes.c_void_p(stream), ci(rpart), ci(ncol), ci(nlay), cd(scon), cd(adjes), v("coszen"), ci(isolvar), v("play"),
            v("plev"), v("tlay"), *[v(k) for k in _SW_GAS], ci(iceflg), ci(liqflg), v("cldf"), v("ciwp"), v("clwp"), v("rei"), v("rel"),
            ci(int(dyofyr)), v("zm"), v("alat"), ci(iaer), v("tauaer_sw"), v("ssaaer_sw"), v("asmaer_sw"), v("asdir"), v("asdif"),
            v("aldir"), v("aldif"), ci(int(cloudLM)), ci(int(cloudMH)), ci(int(normFlx)), v("clearCounts_sw"), v("swuflx"), v("swdflx"),
            v("swuflxc"), v("swdflxc"), *[v(k) for k in ("nirr", "nirf", "parr", "parf", "uvrr", "uvrf", "fswband")],
            *[v(k) for k in _SW_COT], ci(1 if do_drfband else 0), v("drband"), v("dfband"), _p(bs), _p(ind), _p(scf))
        self._chk(rc)

    # ---- Chou-Suarez LW, host arrays ---------------------------------------------------------------------
    def irrad(self, m, np_, ple, ta, wa, oa, tb, co2, trace, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, ict, icb, reff,
              ns, fs, tg, eg, tv, ev, rv, na, nb, taua, ssaa, asya):
        """irrad (irrad.F90:27).  Returns dict(flxu, flcu, flau, flxau, flxd, flcd, flad, flxad, dfdts (np+1, m); sfcem (m);
        taudiag (10, np, m); taua, ssaa, asya = the in-out aerosol arrays after the call)."""
        dt = self.dtype
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        ple, ta, wa, oa, tb, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, reff, fs, tg, eg, tv, ev, rv = map(
            c, (ple, ta, wa, oa, tb, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, reff, fs, tg, eg, tv, ev, rv))
        aer = [None if a is None else np.array(a, dtype=dt, order="C", copy=True) for a in (taua, ssaa, asya)]
        assert ple.shape == (np_ + 1, m) and ta.shape == (np_, m)
        out = {k: np.zeros((np_ + 1, m), dtype=dt) for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts")}
        out["sfcem"] = np.zeros(m, dtype=dt)
        out["taudiag"] = np.zeros((10, np_, m), dtype=dt)
        ci = ctypes.c_int
        rc = self.L.geosrad_irrad(
            self.h, ci(m), ci(np_), _p(ple), _p(ta), _p(wa), _p(oa), _p(tb), ctypes.c_double(co2), ci(1 if trace else 0), _p(n2o), _p(ch4),
            _p(cfc11), _p(cfc12), _p(cfc22), _p(cwc), _p(fcld), ci(int(ict)), ci(int(icb)), _p(reff), ci(int(ns)), _p(fs), _p(tg), _p(eg),
            _p(tv), _p(ev), _p(rv), ci(int(na)), ci(int(nb)), _p(aer[0]), _p(aer[1]), _p(aer[2]),
            *[_p(out[k]) for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts", "sfcem", "taudiag")])
        self._chk(rc)
        out.update(taua=aer[0], ssaa=aer[1], asya=aer[2])
        return out

    def irrad_columns(self, ch, trace=True):
        """Convenience: `ch` as produced by synth.chou_lw_inputs."""
        n1, m = ch["ple"].shape
        return self.irrad(m, n1 - 1, ch["ple"], ch["ta"], ch["wa"], ch["oa"], ch["tb"], ch["co2"], trace, ch["n2o"], ch["ch4"], ch["cfc11"],
                          ch["cfc12"], ch["cfc22"], ch["cwc"], ch["fcld"], ch["ict"], ch["icb"], ch["reff"], ch["ns"], ch["fs"], ch["tg"],
                          ch["eg"], ch["tv"], ch["ev"], ch["rv"], ch["na"], ch["nb"], ch["taua"], ch["ssaa"], ch["asya"])

    def irrad_dev(self, stream, m, np_, ptr, co2, trace, ict, icb, ns, na, nb):
        """`ptr`: dict name -> device address for every array argument of irrad (inputs, in-out aerosols, outputs)."""
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        ci = ctypes.c_int
        rc = self.L.geosrad_irrad_dev(
            self.h, ctypes.c_void_p(stream), ci(m), ci(np_), v("ple"), v("ta"), v("wa"), v("oa"), v("tb"), ctypes.c_double(co2),
            ci(1 if trace else 0), v("n2o"), v("ch4"), v("cfc11"), v("cfc12"), v("cfc22"), v("cwc"), v("fcld"), ci(int(ict)), ci(int(icb)),
            v("reff"), ci(int(ns)), v("fs"), v("tg"), v("eg"), v("tv"), v("ev"), v("rv"), ci(int(na)), ci(int(nb)), v("taua"), v("ssaa"),
            v("asya"), *[v(k) for k in ("flxu", "flcu", "flau", "flxau", "flxd", "flcd", "flad", "flxad", "dfdts", "sfcem", "taudiag")])
        self._chk(rc)

    # ---- Chou-Suarez SW, host arrays ---------------------------------------------------------------------
    def sorad(self, m, np_, nb, cosz, pl, ta, wa, oa, co2, cwc, fcld, ict, icb, reff, hk_uv, hk_ir, taua, ssaa, asya,
              rsuvbm, rsuvdf, rsirbm, rsirdf, do_drfband=False):
        """sorad (sorad.F90:43).  Returns dict(flx, flc, flxu, flcu (np+1, m); fdiruv ... fdifir (m); flx_sfc_band[, drband, dfband] (8, m))."""
        dt = self.dtype
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        cosz, pl, ta, wa, oa, cwc, fcld, reff, hk_uv, hk_ir, taua, ssaa, asya, rsuvbm, rsuvdf, rsirbm, rsirdf = map(
            c, (cosz, pl, ta, wa, oa, cwc, fcld, reff, hk_uv, hk_ir, taua, ssaa, asya, rsuvbm, rsuvdf, rsirbm, rsirdf))
        assert pl.shape == (np_ + 1, m) and hk_uv.shape == (5,) and hk_ir.shape == (10, 3)
        out = {k: np.zeros((np_ + 1, m), dtype=dt) for k in ("flx", "flc", "flxu", "flcu")}
        for k in ("fdiruv", "fdifuv", "fdirpar", "fdifpar", "fdirir", "fdifir"):
            out[k] = np.zeros(m, dtype=dt)
        out["flx_sfc_band"] = np.zeros((8, m), dtype=dt)
        if do_drfband:
            out["drband"] = np.zeros((8, m), dtype=dt); out["dfband"] = np.zeros((8, m), dtype=dt)
        ci = ctypes.c_int
        rc = self.L.geosrad_sorad(
            self.h, ci(m), ci(np_), ci(nb), _p(cosz), _p(pl), _p(ta), _p(wa), _p(oa), ctypes.c_double(co2), _p(cwc), _p(fcld), ci(int(ict)),
            ci(int(icb)), _p(reff), _p(hk_uv), _p(hk_ir), _p(taua), _p(ssaa), _p(asya), _p(rsuvbm), _p(rsuvdf), _p(rsirbm), _p(rsirdf),
            _p(out["flx"]), _p(out["flc"]), _p(out["fdiruv"]), _p(out["fdifuv"]), _p(out["fdirpar"]), _p(out["fdifpar"]), _p(out["fdirir"]),
            _p(out["fdifir"]), _p(out["flxu"]), _p(out["flcu"]), _p(out["flx_sfc_band"]), ci(1 if do_drfband else 0), _p(out.get("drband")),
            _p(out.get("dfband")))
        self._chk(rc)
        return out

    def sorad_columns(self, cs, do_drfband=False):
        """Convenience: `cs` as produced by synth.chou_sw_inputs."""
        n1, m = cs["pl"].shape
        return self.sorad(m, n1 - 1, cs["nb"], cs["cosz"], cs["pl"], cs["ta"], cs["wa"], cs["oa"], cs["co2"], cs["cwc"], cs["fcld"], cs["ict"],
                          cs["icb"], cs["reff"], cs["hk_uv"], cs["hk_ir"], cs["taua"], cs["ssaa"], cs["asya"], cs["rsuvbm"], cs["rsuvdf"],
                          cs["rsirbm"], cs["rsirdf"], do_drfband=do_drfband)

    def sorad_dev(self, stream, m, np_, nb, ptr, co2, ict, icb, hk_uv, hk_ir, do_drfband=False):
        """`ptr`: dict name -> device address for every array argument of sorad; hk_uv / hk_ir are host arrays."""
        dt = self.dtype
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        hu = np.ascontiguousarray(hk_uv, dtype=dt); hi = np.ascontiguousarray(hk_ir, dtype=dt)
        ci = ctypes.c_int
        rc = self.L.geosrad_sorad_dev(
            self.h, ctypes.c_void_p(stream), ci(m), ci(np_), ci(nb), v("cosz"), v("pl"), v("ta"), v("wa"), v("oa"), ctypes.c_double(co2),
            v("cwc"), v("fcld"), ci(int(ict)), ci(int(icb)), v("reff"), _p(hu), _p(hi), v("taua"), v("ssaa"), v("asya"), v("rsuvbm"),
            v("rsuvdf"), v("rsirbm"), v("rsirdf"), v("flx"), v("flc"), v("fdiruv"), v("fdifuv"), v("fdirpar"), v("fdifpar"), v("fdirir"),
            v("fdifir"), v("flxu"), v("flcu"), v("flx_sfc_band"), ci(1 if do_drfband else 0), v("drband"), v("dfband"))
        self._chk(rc)

    # ---- RRTMG_LW, device pointers (bench / drivers that keep data in HBM) -------------------------------------
    def rrtmg_lw_dev(self, stream, ncol, nlay, dudTs, ptr, iceflg, liqflg, dyofyr, cloudLM, cloudMH, band_output=None):
        """`ptr`: dict name -> device address (int) for every argument array of rrtmg_lw (inputs and outputs)."""
        bo = np.zeros(NBNDLW, dtype=np.int32) if band_output is None else np.ascontiguousarray(band_output, dtype=np.int32)
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        ci = ctypes.c_int
        rc = self.L.geosrad_rrtmg_lw_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(nlay), ci(4), ci(1 if dudTs else 0), v("play"), v("plev"), v("tlay"),
            v("tlev"), v("tsfc"), v("emis"), *[v(k) for k in _IN2D[:10]], *[v(k) for k in _IN2D[10:]], ci(iceflg), ci(liqflg),
            v("tauaer"), v("zm"), v("alat"), ci(int(dyofyr)), ci(int(cloudLM)), ci(int(cloudMH)), v("clearCounts"), v("uflx"),
            v("dflx"), v("uflxc"), v("dflxc"), v("duflx_dTs"), v("duflxc_dTs"), _p(bo), v("olrb"), v("dolrb_dTs"))
        self._chk(rc)

    def rrtmg_lw_rats_dev(self, stream, ncol, nlay, dudTs, ptr, iceflg, liqflg, dyofyr, cloudLM, cloudMH, rat_gas, band_output=None):
        """rrtmg_lw_dev + the RATS loop of LW_Driver (GEOS_IrradGridComp.F90:3405-3468) from the same call: `rat_gas` = names from
        gridcomp.RAT_GAS; ptr["uflx_rat"], ["dflx_rat"], ["duflx_dTs_rat"] = device arrays [len(rat_gas)][nlay+1][ncol]."""
        from . import gridcomp as G
        bo = np.zeros(NBNDLW, dtype=np.int32) if band_output is None else np.ascontiguousarray(band_output, dtype=np.int32)
        rg = np.ascontiguousarray([G.RAT_GAS.index(g) if isinstance(g, str) else int(g) for g in rat_gas], dtype=np.int32)
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        ci = ctypes.c_int
        rc = self.L.geosrad_rrtmg_lw_rats_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(nlay), ci(4), ci(1 if dudTs else 0), v("play"), v("plev"), v("tlay"),
            v("tlev"), v("tsfc"), v("emis"), *[v(k) for k in _IN2D[:10]], *[v(k) for k in _IN2D[10:]], ci(iceflg), ci(liqflg),
            v("tauaer"), v("zm"), v("alat"), ci(int(dyofyr)), ci(int(cloudLM)), ci(int(cloudMH)), v("clearCounts"), v("uflx"),
            v("dflx"), v("uflxc"), v("dflxc"), v("duflx_dTs"), v("duflxc_dTs"), _p(bo), v("olrb"), v("dolrb_dTs"),
            ci(len(rg)), _p(rg), v("uflx_rat"), v("dflx_rat"), v("duflx_dTs_rat"))
        self._chk(rc)

    # ---- GridComp data path either side of the solvers (device pointers, GEOS layout) ---------------------------------------
    @staticmethod
    def _ptr_array(names, ptr):
        arr = (ctypes.c_void_p * len(names))()
        for i, k in enumerate(names):
            arr[i] = ptr.get(k) or None
        return arr

    def lw_driver_rrtmg_dev(self, stream, ncol, lm, nb_aer, ptr, consts, iceflg, liqflg, doy, lcldlm, lcldmh, band_output=None):
        """RRTMG branch of LW_Driver (GEOS_IrradGridComp.F90:3188-3615).  `ptr`: name -> device address for gridcomp.LWD_IN and
        gridcomp.LWD_OUT (missing / 0 = not associated); `consts` in gridcomp.LWD_CONST order; lcldlm / lcldmh in MODEL ordering."""
        from . import gridcomp as G
        bo = np.zeros(NBNDLW, dtype=np.int32) if band_output is None else np.ascontiguousarray(band_output, dtype=np.int32)
        cs = (ctypes.c_double * len(G.LWD_CONST))(*consts)
        ci = ctypes.c_int
        self._chk(self.L.geosrad_lw_driver_rrtmg_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(nb_aer), self._ptr_array(G.LWD_IN, ptr), cs, ci(iceflg), ci(liqflg),
            ci(int(doy)), ci(int(lcldlm)), ci(int(lcldmh)), _p(bo), self._ptr_array(G.LWD_OUT, ptr)))

    def lw_driver_rrtmg_rats_dev(self, stream, ncol, lm, nb_aer, ptr, consts, iceflg, liqflg, doy, lcldlm, lcldmh, rat_gas,
                                 band_output=None):
        """lw_driver_rrtmg_dev with the RATS loop (IRR:3389-3469, :3522-3530, :3614): `rat_gas` = names from gridcomp.RAT_GAS,
        `ptr` additionally holds gridcomp.LWD_RAT_OUT ((nrats, LM+1, ncol), SFCEM_RAT (nrats, ncol); missing = not associated)."""
        from . import gridcomp as G
        bo = np.zeros(NBNDLW, dtype=np.int32) if band_output is None else np.ascontiguousarray(band_output, dtype=np.int32)
        rg = np.ascontiguousarray([G.RAT_GAS.index(g) if isinstance(g, str) else int(g) for g in rat_gas], dtype=np.int32)
        cs = (ctypes.c_double * len(G.LWD_CONST))(*consts)
        ci = ctypes.c_int
        self._chk(self.L.geosrad_lw_driver_rrtmg_rats_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(nb_aer), self._ptr_array(G.LWD_IN, ptr), cs, ci(iceflg), ci(liqflg),
            ci(int(doy)), ci(int(lcldlm)), ci(int(lcldmh)), _p(bo), self._ptr_array(G.LWD_OUT, ptr), ci(len(rg)), _p(rg),
            self._ptr_array(G.LWD_RAT_OUT, ptr)))

    def lw_update_rats_dev(self, stream, ncol, lm, nrats, ptr):
        """RATS exports of Update_Flx (GEOS_IrradGridComp.F90:4036-4120).  `ptr`: name -> device address for gridcomp.LWR_IN and
        gridcomp.LWR_OUT (missing = export not associated)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_lw_update_rats_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(nrats),
                                                     self._ptr_array(G.LWR_IN, ptr), self._ptr_array(G.LWR_OUT, ptr)))

    def lw_update_bands_dev(self, stream, ncol, band_output, undef, ptr):
        """Band OLR / brightness-temperature exports of Update_Flx (GEOS_IrradGridComp.F90:3993-4021).  `ptr`: TSINST, TS_INT, OLRB,
        DOLRB (internals, (ncol,16) C order) and the exports OLRB_EXP, TBRB_EXP ((16,ncol) C order; missing = not associated)."""
        from . import gridcomp as G
        bo = np.ascontiguousarray(band_output, dtype=np.int32)
        w1 = (ctypes.c_double * 16)(*G.LW_WAVENUM1); w2 = (ctypes.c_double * 16)(*G.LW_WAVENUM2)
        v = lambda k: ctypes.c_void_p(ptr[k]) if ptr.get(k) else None
        self._chk(self.L.geosrad_lw_update_bands_dev(self.h, ctypes.c_void_p(stream), ctypes.c_int(ncol), _p(bo), w1, w2, ctypes.c_double(undef),
                                                      v("TSINST"), v("TS_INT"), v("OLRB"), v("DOLRB"), v("OLRB_EXP"), v("TBRB_EXP")))

    def sw_update_surface_dev(self, stream, ncol, lm, undef, ptr):
        """2-D block of UPDATE_EXPORT (GEOS_SolarGridComp.F90:7403-7533): `ptr` name -> device address for gridcomp.SWS_IN / SWS_OUT
        (the albedo exports are named ALBVF_X ...; missing = not associated)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_sw_update_surface_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ctypes.c_double(undef),
                                                        self._ptr_array(G.SWS_IN, ptr), self._ptr_array(G.SWS_OUT, ptr)))

    def sw_driver_rrtmg_dev(self, stream, ncol, lm, nb_aer, ptr, consts, iceflg, liqflg, sc, dist, isolvar, dyofyr, include_aerosols,
                            lcldlm, lcldmh, normflx=1, bndsolvar=None, indsolvar=None):
        """RRTMG branch of SORADCORE on the packed daytime columns (GEOS_SolarGridComp.F90:6113-6450)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        cs = (ctypes.c_double * len(G.SWD_CONST))(*consts)
        bs = None if bndsolvar is None else np.ascontiguousarray(bndsolvar, dtype=self.dtype)
        ins = None if indsolvar is None else np.ascontiguousarray(indsolvar, dtype=self.dtype)
        self._chk(self.L.geosrad_sw_driver_rrtmg_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(nb_aer), self._ptr_array(G.SWD_IN, ptr), cs, ci(iceflg), ci(liqflg),
            ctypes.c_double(sc), ctypes.c_double(dist), ci(isolvar), ci(int(dyofyr)), ci(1 if include_aerosols else 0), ci(int(lcldlm)),
            ci(int(lcldmh)), ci(normflx), None if bs is None else _p(bs), None if ins is None else _p(ins),
            self._ptr_array(G.SWD_OUT, ptr)))

    def sw_driver_chou_dev(self, stream, ncol, lm, ptr, consts, lcldmh, lcldlm, hk_uv, hk_ir, do_drfband=False):
        """Chou-Suarez branch of SORADCORE on the packed daytime columns (GEOS_SolarGridComp.F90:4484-4572, SHRTWAVE :6597-6672):
        `ptr` holds device pointers of gridcomp.SWC_IN (TAUA / SSAA / ASYA may be missing: no aerosols) and gridcomp.SWC_OUT;
        consts in the order of gridcomp.SWC_CONST; hk_uv (5) and hk_ir (memory order of the Fortran (3,10), as for sorad_dev) host arrays
        (HK_UV_TEMP, HK_IR_TEMP)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        cs = (ctypes.c_double * len(G.SWC_CONST))(*consts)
        hu = np.ascontiguousarray(hk_uv, dtype=self.dtype)
        hi = np.ascontiguousarray(hk_ir, dtype=self.dtype)
        self._chk(self.L.geosrad_sw_driver_chou_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), self._ptr_array(G.SWC_IN, ptr), cs,
                                                    ci(int(lcldmh)), ci(int(lcldlm)), _p(hu), _p(hi), ci(1 if do_drfband else 0),
                                                    self._ptr_array(G.SWC_OUT, ptr)))

    def lw_chou_post_dev(self, stream, ncol, lm, ptr):
        """after irrad in the Chou-Suarez branch of LW_Driver (GEOS_IrradGridComp.F90:2101-2108, :3601-3616): gridcomp.LWC_IN / LWC_OUT"""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_lw_chou_post_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), self._ptr_array(G.LWC_IN, ptr),
                                                  self._ptr_array(G.LWC_OUT, ptr)))

    def lw_update_flx_dev(self, stream, ncol, lm, rrtmg, lev_mid_high, lev_low_mid, undef, ptr):
        """Update_Flx (GEOS_IrradGridComp.F90:3796-3999): `ptr` holds gridcomp.LWU_IN internals and the requested gridcomp.LWU_OUT."""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_lw_update_flx_dev(
            self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(1 if rrtmg else 0), ci(int(lev_mid_high)), ci(int(lev_low_mid)),
            ctypes.c_double(undef), self._ptr_array(G.LWU_IN, ptr), self._ptr_array(G.LWU_OUT, ptr)))

    def sw_update_export_dev(self, stream, ncol, lm, nbands, ptr):
        """flux part of UPDATE_EXPORT (GEOS_SolarGridComp.F90:7540-7579)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_sw_update_export_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ci(nbands),
                                                      self._ptr_array(G.SWU_IN, ptr), self._ptr_array(G.SWU_OUT, ptr)))

    def rad_tendencies_dev(self, stream, ncol, lm, grav, cp, ptr):
        """heating rates of the parent GridComp (GEOS_RadiationGridComp.F90:798-819)."""
        from . import gridcomp as G
        ci = ctypes.c_int
        self._chk(self.L.geosrad_rad_tendencies_dev(self.h, ctypes.c_void_p(stream), ci(ncol), ci(lm), ctypes.c_double(grav),
                                                    ctypes.c_double(cp), self._ptr_array(G.RT_IN, ptr), self._ptr_array(G.RT_OUT, ptr)))

    def profile(self, enable=True):
        self._chk(self.L.geosrad_profile(self.h, ctypes.c_int(1 if enable else 0)))

    def profile_read(self):
        """{kernel name: (total ms, launches)} measured with HIP events on the launch stream; the names are those of the kernels that ran
        (a rocprofv3 kernel trace of the same run shows them as geosrad::<name><...>)."""
        out = {}
        for k in range(14):
            ms = ctypes.c_double(); n = ctypes.c_long()
            self._chk(self.L.geosrad_profile_read(self.h, ctypes.c_int(k), ctypes.byref(ms), ctypes.byref(n)))
            out[self.L.geosrad_kernel_label(self.h, ctypes.c_int(k)).decode()] = (ms.value, n.value)
        return out

    def check(self, stream=0):
        self._chk(self.L.geosrad_check(self.h, ctypes.c_void_p(stream)))

    # ---- McICA ---------------------------------------------------------------------------------------------
    def generate_stochastic_clouds(self, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny,
                                   seed_order=(1, 2, 3, 4)):
        """Inputs in the solver-API layout, numpy (nlay,ncol).  Returns cldy (int32), ciwp_stoch, clwp_stoch as numpy
        (ncol,nsubcol,nlay) == Fortran (nlay,nsubcol,ncol)."""
        dt = self.dtype
        c = lambda a: np.ascontiguousarray(a, dtype=dt)
        zmid, alat, play, cldfrac, ciwp, clwp = map(c, (zmid, alat, play, cldfrac, ciwp, clwp))
        cldy = np.zeros((ncol, nsubcol, nlay), dtype=np.int32)
        ci_s = np.zeros((ncol, nsubcol, nlay), dtype=dt); cl_s = np.zeros_like(ci_s)
        so = (ctypes.c_int32 * 4)(*[int(s) for s in seed_order])
        rc = self.L.geosrad_mcica(self.h, ctypes.c_int(ncol), ctypes.c_int(nsubcol), ctypes.c_int(nlay), _p(zmid), _p(alat),
                                  ctypes.c_int(int(doy)), _p(play), _p(cldfrac), _p(ciwp), _p(clwp), ctypes.c_double(cwp_tiny),
                                  so, _p(cldy), _p(ci_s), _p(cl_s))
        self._chk(rc)
        return cldy, ci_s, cl_s

    def generate_stochastic_clouds_dev(self, stream, ncol, nsubcol, nlay, ptr, doy, cwp_tiny, seed_order=(1, 2, 3, 4)):
        """device-pointer variant: `ptr` maps zm, alat, play, cldf, ciwp, clwp (inputs, solver-API layout) and
        cldy_stoch (int32), ciwp_stoch, clwp_stoch (outputs, Fortran (nlay,nsubcol,ncol)) to device addresses."""
        v = lambda k: ctypes.c_void_p(ptr[k])
        so = (ctypes.c_int32 * 4)(*[int(s) for s in seed_order])
        rc = self.L.geosrad_mcica_dev(self.h, ctypes.c_void_p(stream), ctypes.c_int(ncol), ctypes.c_int(nsubcol), ctypes.c_int(nlay),
                                      v("zm"), v("alat"), ctypes.c_int(int(doy)), v("play"), v("cldf"), v("ciwp"), v("clwp"),
                                      ctypes.c_double(cwp_tiny), so, v("cldy_stoch"), v("ciwp_stoch"), v("clwp_stoch"))
        self._chk(rc)

    # ---- lit-column compaction (GEOS_SolarGridComp.F90:3686, PackIt / UnPackIt :7753-7799); device addresses ------------------
    def lit_index_dev(self, stream, ncol, zth, lit_index, lit_pos, nlit_dev, want_count=True):
        """zth: (ncol,) reals; lit_index, lit_pos: (ncol,) int32; nlit_dev: (1,) int32.  Returns NumLit when want_count (synchronises)."""
        n = ctypes.c_int(0)
        rc = self.L.geosrad_lit_index_dev(self.h, ctypes.c_void_p(stream), ctypes.c_int(ncol), ctypes.c_void_p(zth), ctypes.c_void_p(lit_index),
                                          ctypes.c_void_p(lit_pos), ctypes.c_void_p(nlit_dev), ctypes.byref(n) if want_count else None)
        self._chk(rc)
        return n.value if want_count else None

    def lit_pack_dev(self, stream, pdim, udim, nlev, lit_index, nlit_dev, unpacked, packed):
        self._chk(self.L.geosrad_lit_pack_dev(self.h, ctypes.c_void_p(stream), ctypes.c_int(pdim), ctypes.c_int(udim), ctypes.c_int(nlev),
                                              ctypes.c_void_p(lit_index), ctypes.c_void_p(nlit_dev), ctypes.c_void_p(unpacked),
                                              ctypes.c_void_p(packed)))

    def lit_unpack_dev(self, stream, pdim, udim, nlev, lit_pos, packed, unpacked, default=None):
        self._chk(self.L.geosrad_lit_unpack_dev(self.h, ctypes.c_void_p(stream), ctypes.c_int(pdim), ctypes.c_int(udim), ctypes.c_int(nlev),
                                                ctypes.c_void_p(lit_pos), ctypes.c_void_p(packed), ctypes.c_void_p(unpacked),
                                                ctypes.c_int(0 if default is None else 1), ctypes.c_double(0.0 if default is None else default)))

    def clearCounts_threeBand(self, ncol, nsubcol, nlay, cloudLM, cloudMH, cldy_stoch):
        cldy = np.ascontiguousarray(cldy_stoch, dtype=np.int32)
        cnt = np.zeros((ncol, 4), dtype=np.int32)
        rc = self.L.geosrad_clearcounts(self.h, ctypes.c_int(ncol), ctypes.c_int(nsubcol), ctypes.c_int(nlay),
                                        ctypes.c_int(int(cloudLM)), ctypes.c_int(int(cloudMH)), _p(cldy), _p(cnt))
        self._chk(rc)
        return cnt
