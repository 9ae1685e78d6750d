// gridcomp_kernels.hpp -- the data path of the GridComp drivers either side of the RRTMG solvers,
// as HBM-streaming kernels (gfx950):
//   k_lwd_prep / k_lwd_zm / k_lwd_post   the RRTMG branch of LW_Driver (GEOSirrad_GridComp/GEOS_IrradGridComp.F90:3188-3372 and
//                                        :3487-3533, :3601-3615): unit conversions, level temperatures, effective-radius limits,
//                                        absorption aerosol optical depth, vertical flip; after the solver: flip back, sign
//                                        conventions, SFCEM, net fluxes, super-layer cloud fractions
//   k_swd_prep / k_swd_zm / k_swd_post   the same for SORADCORE's RRTMG branch (GEOSsolar_GridComp/GEOS_SolarGridComp.F90:6113-6219,
//                                        :6395-6454)
//   k_lw_update_flx                      Update_Flx (IRR:3861-3999): the heartbeat linearisation of the LW fluxes in the surface
//                                        temperature, every model step between two full calculations
//   k_sw_update_export                   the 3-D / TOA / surface flux part of UPDATE_EXPORT (SOL:7540-7579): normalised fluxes x SLR
//   k_rad_tendencies                     the parent's heating rates (GEOS_RadiationGridComp.F90:798-819)
// GEOS fields are (IM,JM,levels): column index fastest, which is the [level][column] layout of every kernel here (lane = column,
// blockIdx.y = level): all accesses are coalesced, nothing is transposed, the vertical flip is an index calculation.
// A null output pointer = Fortran "not associated" (export not requested).
#pragma once
#include "lw_device.hpp"

namespace geosrad {

// ---------------------------------------------------------------------------------------------------------------------------
// LW_Driver, RRTMG branch
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct LwdArgs {
    int ncol, lm, nb;                 // columns (IM*JM), layers, aerosol bands of TAUA/SSAA (16)
    int iceflg, liqflg;
    // GEOS side, model ordering (k = 1 top .. LM bottom); ple has LM+1 levels (0..LM)
    const R *ple, *pl, *t, *q, *o3, *ch4, *n2o, *co2_3d, *cfc11, *cfc12, *hcfc22, *fcld;
    const R *cwc_liq, *cwc_ice, *reff_liq, *reff_ice;      // CWC(:,:,:,KLIQUID|KICE), REFF(...)
    const R *taua, *ssaa;                                  // (IM,JM,LM,nb) or null
    const R *ts, *emis, *lats, *t2m;
    R co2_fixed, o2, ccl4, airmw_over_h2omw, airmw_over_o3mw, rgas, grav;
    // RRTMG side, 1 = bottom layer; [K][ncol]
    R *play, *plev, *tlay, *tlev, *tsfc, *emis_r, *h2o, *o3_r, *co2_r, *ch4_r, *n2o_r, *o2_r, *cfc11_r, *cfc12_r, *cfc22_r, *ccl4_r,
        *cldf, *ciwp, *clwp, *rei, *rel, *tauaer, *zm, *alat;
};

// interface temperature of model level k (1..LM+1), IRR:3248-3256
template <typename R> GR_DEV R lwd_tlev(const LwdArgs<R> &A, int k, int ij)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int n = A.ncol, lm = A.lm;
    if (k == lm + 1) return A.t2m[ij];
    if (k == 1) k = 2;
    auto PLE = [&](int l) { return A.ple[(size_t)l * n + ij]; };
    auto T = [&](int l) { return A.t[(size_t)(l - 1) * n + ij]; };
    const R dpk = PLE(k) - PLE(k - 1), dpm = PLE(k - 1) - PLE(k - 2);
    return (T(k - 1) * dpk + T(k) * dpm) / (dpm + dpk);
}

template <typename R> GR_DEV R clampr(R x, R lo, R hi) { x = x > lo ? x : lo; return x < hi ? x : hi; }      // min(max(x,lo),hi)

// effective-radius limits RRTMG assumes (IRR:3272-3291, SOL:6144-6170)
template <typename R> GR_DEV void rrtmg_reff_limits(int iceflg, int liqflg, R &reice, R &reliq)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    if (liqflg == 0) reliq = clampr<R>(reliq, (R)5.0, (R)10.0);
    else if (liqflg == 1) reliq = clampr<R>(reliq, (R)2.5, (R)60.0);
    if (iceflg == 0) reice = clampr<R>(reice, (R)10.0, (R)30.0);
    else if (iceflg == 1) reice = clampr<R>(reice, (R)13.0, (R)130.0);
    else if (iceflg == 2) reice = clampr<R>(reice, (R)5.0, (R)131.0);
    else if (iceflg == 3) reice = clampr<R>(reice, (R)5.0, (R)140.0);
    else if (iceflg == 4) reice = clampr<R>(reice * (R)2., (R)1.0, (R)200.0);
}

// one thread per (column, RRTMG layer K = 1..LM); blockIdx.y = K - 1, and K = LM additionally writes the top level
template <typename R> __global__ void __launch_bounds__(256) k_lwd_prep(LwdArgs<R> A)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= A.ncol) return;
    const int n = A.ncol, lm = A.lm, K = blockIdx.y + 1, LV = lm - K + 1;
    const size_t o = (size_t)(K - 1) * n + ij;          // RRTMG layer K
    const size_t g = (size_t)(LV - 1) * n + ij;         // GEOS layer LV
    const R pleb = A.ple[(size_t)LV * n + ij], plet = A.ple[(size_t)(LV - 1) * n + ij];
    const R dp = pleb - plet;
    // content [kg/kg] -> path [g/m2]: 1000*dp/g ~ 1.02*100*dp (IRR:3262-3267)
    const R xx = (R)1.02 * (R)100 * dp;
    A.clwp[o] = xx * A.cwc_liq[g];
    A.ciwp[o] = xx * A.cwc_ice[g];
    R reliq = A.reff_liq[g], reice = A.reff_ice[g];
    rrtmg_reff_limits<R>(A.iceflg, A.liqflg, reice, reliq);
    A.rel[o] = reliq; A.rei[o] = reice;
    // levels: PLE_R(0:LM) = PLE(LM:0)/100, TLEV_R(0:LM) = TLEV(LM+1:1) (IRR:3293-3299, :3340-3342)
    A.plev[o] = pleb / (R)100.;
    A.tlev[o] = lwd_tlev<R>(A, LV + 1, ij);
    if (K == lm) {
        A.plev[(size_t)lm * n + ij] = A.ple[ij] / (R)100.;
        A.tlev[(size_t)lm * n + ij] = lwd_tlev<R>(A, 1, ij);
    }
    // layers (IRR:3301-3322) with the clean-up of negatives (IRR:3361-3371)
    auto pos = [](R x) { return x < 0 ? (R)0 : x; };
    A.play[o] = A.pl[g] / (R)100.;
    A.tlay[o] = A.t[g];
    const R q = A.q[g];
    A.h2o[o] = pos(q / ((R)1. - q) * A.airmw_over_h2omw);
    A.o3_r[o] = pos(A.o3[g] * A.airmw_over_o3mw);
    A.ch4_r[o] = pos(A.ch4[g]);
    A.n2o_r[o] = pos(A.n2o[g]);
    A.co2_r[o] = pos(A.co2_3d ? A.co2_3d[g] : A.co2_fixed);
    A.o2_r[o] = pos(A.o2);
    A.ccl4_r[o] = pos(A.ccl4);
    A.cfc11_r[o] = pos(A.cfc11[g]);
    A.cfc12_r[o] = pos(A.cfc12[g]);
    A.cfc22_r[o] = pos(A.hcfc22[g]);
    A.cldf[o] = pos(A.fcld[g]);
    // absorption aerosol optical thickness (IRR:3324-3335); tauaer is (ncol,nlay,16)
    for (int b = 0; b < 16; b++) {
        R v = 0;
        if (A.taua && b < A.nb) {
            const size_t ga = ((size_t)b * lm + (LV - 1)) * n + ij;
            v = A.taua[ga] - A.ssaa[ga];
            v = v > 0 ? v : (R)0;
        }
        A.tauaer[((size_t)b * lm + (K - 1)) * n + ij] = v;
    }
    if (K == 1) {
        A.tsfc[ij] = A.ts[ij];
        A.alat[ij] = A.lats[ij];
        const R e = A.emis[ij];
        for (int b = 0; b < 16; b++) A.emis_r[(size_t)b * n + ij] = e;      // all bands get the same emissivity (IRR:3244)
    }
}

// layer mid-point heights, a running sum up the column (IRR:3344-3356): one thread per column
template <typename R> __global__ void __launch_bounds__(256) k_lwd_zm(LwdArgs<R> A)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= A.ncol) return;
    const int n = A.ncol, lm = A.lm;
    R z = 0;
    A.zm[ij] = 0;
    R plm = A.play[ij];
    for (int K = 2; K <= lm; K++) {
        const R plk = A.play[(size_t)(K - 1) * n + ij];
        // dz ~ RT/g x dp/p; the jump from LAYER K-1 to K is centred on LEVEL K-1 (0-based levels)
        z = z + A.rgas * A.tlev[(size_t)(K - 1) * n + ij] / A.grav * (plm - plk) / A.plev[(size_t)(K - 1) * n + ij];
        A.zm[(size_t)(K - 1) * n + ij] = z;
        plm = plk;
    }
}

template <typename R> struct LwdPost {
    int ncol, lm, ngpt;
    const R *uflx, *dflx, *uflxc, *dflxc, *duflx, *duflxc;      // RRTMG (ncol, LM+1), 1 = surface
    const int32_t *clearCounts;                                 // (ncol,4)
    const R *emis, *ts;
    // GEOS internal state (IM,JM,0:LM); any may be null
    R *flxu_int, *flxd_int, *flcu_int, *flcd_int, *dfdts, *dfdtsc, *dfdtsna, *dfdtscna, *flx_int, *flc_int;
    R *sfcem_int, *ts_int, *cldttlw, *cldhilw, *cldmdlw, *cldlolw;
};

// one thread per (column, GEOS level K = 0..LM) (IRR:3487-3533, :3601-3615, :3560-3565)
template <typename R> __global__ void __launch_bounds__(256) k_lwd_post(LwdPost<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const int n = P.ncol, lm = P.lm, K = blockIdx.y, LV = lm - K + 1;
    const size_t s = (size_t)(LV - 1) * n + ij, o = (size_t)K * n + ij;
    // upward negative in the GEOS convention
    const R fu = -P.uflx[s], fd = P.dflx[s], cu = -P.uflxc[s], cd = P.dflxc[s], du = -P.duflx[s], dc = -P.duflxc[s];
    if (P.flxu_int) P.flxu_int[o] = fu;
    if (P.flxd_int) P.flxd_int[o] = fd;
    if (P.flcu_int) P.flcu_int[o] = cu;
    if (P.flcd_int) P.flcd_int[o] = cd;
    if (P.dfdts) P.dfdts[o] = du;
    if (P.dfdtsc) P.dfdtsc[o] = dc;
    if (P.dfdtsna) P.dfdtsna[o] = du;        // RRTMG has no no-aerosol derivatives (IRR:3560-3565)
    if (P.dfdtscna) P.dfdtscna[o] = dc;
    if (P.flx_int) P.flx_int[o] = fd + fu;   // net downward (IRR:3601-3604)
    if (P.flc_int) P.flc_int[o] = cd + cu;
    if (K == lm) {
        // surface emitted: reflected LW is not counted; first negative (Chou-Suarez convention), then reverted (IRR:3512, :3608)
        R sf = -(P.uflx[ij] - P.dflx[ij] * ((R)1. - P.emis[ij]));
        sf = -sf;
        if (P.sfcem_int) P.sfcem_int[ij] = sf;
        if (P.ts_int) P.ts_int[ij] = P.ts[ij];
        const R ng = (R)P.ngpt;
        if (P.cldttlw) P.cldttlw[ij] = (R)1.0 - (R)P.clearCounts[(size_t)0 * n + ij] / ng;
        if (P.cldhilw) P.cldhilw[ij] = (R)1.0 - (R)P.clearCounts[(size_t)1 * n + ij] / ng;
        if (P.cldmdlw) P.cldmdlw[ij] = (R)1.0 - (R)P.clearCounts[(size_t)2 * n + ij] / ng;
        if (P.cldlolw) P.cldlolw[ij] = (R)1.0 - (R)P.clearCounts[(size_t)3 * n + ij] / ng;
    }
}

// RATS share of the un-flip (IRR:3522-3530) and of the net fluxes (IRR:3614): one thread per (column, GEOS level K, gas)
template <typename R> struct LwdRatPost {
    int ncol, lm, nrats;
    const R *uflx, *dflx, *duflx;               // RRTMG (ncol, LM+1, nrats), 1 = surface
    const R *emis;                              // EMIS(I,J) = EMISS(IJ,1): all bands share it (IRR:3249)
    R *flxu_rat, *flxd_rat, *flx_rat, *dfdts_rat, *sfcem_rat;     // (ncol,0:LM,nrats), SFCEM_RAT (ncol,nrats); any may be null
};
template <typename R> __global__ void __launch_bounds__(256) k_lwd_rat_post(LwdRatPost<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const int n = P.ncol, lm = P.lm, K = blockIdx.y, r = blockIdx.z, LV = lm - K + 1;
    const size_t plane = (size_t)(lm + 1) * n * r;
    const size_t s = plane + (size_t)(LV - 1) * n + ij, o = plane + (size_t)K * n + ij;
    const R fu = -P.uflx[s], fd = P.dflx[s];
    if (P.flxu_rat) P.flxu_rat[o] = fu;
    if (P.flxd_rat) P.flxd_rat[o] = fd;
    if (P.dfdts_rat) P.dfdts_rat[o] = -P.duflx[s];
    if (P.flx_rat) P.flx_rat[o] = fd + fu;
    if (K == lm && P.sfcem_rat) P.sfcem_rat[(size_t)r * n + ij] = P.uflx[plane + ij] - P.dflx[plane + ij] * ((R)1.0 - P.emis[ij]);
}

// RATS exports of Update_Flx (IRR:4036-4120): one thread per (column, level K = 0..LM, gas)
template <typename R> struct LwRatUpd {
    int ncol, lm, nrats;
    const R *flx_int, *sfcem_int, *dfdts;                 // (ncol,0:LM), (ncol), (ncol,0:LM)
    const R *flx_rat, *sfcem_rat, *dfdts_rat;             // (ncol,0:LM,nrats), (ncol,nrats), (ncol,0:LM,nrats)
    R *dolr, *dlws, *dflns, *dsfcem, *nettrap;            // (ncol,nrats)
    R *coltrap;                                           // (ncol,LM,nrats)
    R *flx, *dfdts_out;                                   // (ncol,0:LM,nrats)
};
template <typename R> __global__ void __launch_bounds__(256) k_lw_update_rats(LwRatUpd<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const int n = P.ncol, lm = P.lm, K = blockIdx.y, r = blockIdx.z;
    const size_t pl = (size_t)(lm + 1) * n * r, o = (size_t)K * n + ij;
    const R f = P.flx_int[o], fr = P.flx_rat[pl + o];
    if (P.flx) P.flx[pl + o] = f - fr;
    if (P.dfdts_out) P.dfdts_out[pl + o] = P.dfdts[o] - P.dfdts_rat[pl + o];
    if (K >= 1 && P.coltrap) {
        R t = f - P.flx_int[o - n];
        t = t - (fr - P.flx_rat[pl + o - n]);
        P.coltrap[(size_t)lm * n * r + (size_t)(K - 1) * n + ij] = t;
    }
    if (K == lm) {
        const size_t q = (size_t)r * n + ij;
        const R f0 = P.flx_int[ij], fr0 = P.flx_rat[pl + ij], se = P.sfcem_int[ij], ser = P.sfcem_rat[q];
        if (P.dolr) { const R a = -(fr0); P.dolr[q] = (-(f0)) - a; }
        if (P.dlws) P.dlws[q] = (f + se) - (fr + ser);
        if (P.dflns) P.dflns[q] = f - fr;
        if (P.dsfcem) P.dsfcem[q] = se - ser;
        if (P.nettrap) { R t = f - f0; t = t - (fr - fr0); P.nettrap[q] = t; }
    }
}

// band OLR and brightness temperature of Update_Flx (IRR:3993-4021, Tbr_from_band_flux :4132-4186, invert_Planck_for_T :4188-4208):
// one thread per (column, band).  PASS 0 writes the updated band flux and records whether any column of the band is non-zero
// (the reference tests `all(Fband_ == 0.0)` on the whole field); PASS 1 inverts the Planck function.
template <typename R> struct LwBandUpd {
    int ncol;
    int band_output[16];
    R wn1[16], wn2[16];                 // [m-1], in the caller's real kind like wavenum1(ibnd)*100.
    R undef;
    const R *tsinst, *ts_int, *olrb_int, *dolrb_int;      // (ncol), (ncol), (16,ncol), (16,ncol)
    R *olrb_exp, *tbrb_exp;             // (ncol,16); either may be null
    int *nonzero;                       // [16]
};
template <typename R, int PASS> __global__ void __launch_bounds__(256) k_lw_update_bands(LwBandUpd<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x, ib = blockIdx.y;
    if (ij >= P.ncol || !P.band_output[ib]) return;
    const R delt = P.tsinst[ij] - P.ts_int[ij];
    const R f = P.olrb_int[(size_t)ij * 16 + ib] + P.dolrb_int[(size_t)ij * 16 + ib] * delt;
    const size_t o = (size_t)ib * P.ncol + ij;
    if (PASS == 0) {
        if (P.olrb_exp) P.olrb_exp[o] = f;
        if (f != (R)0) P.nonzero[ib] = 1;           // every writer writes 1
    } else {
        if (!P.nonzero[ib]) { P.tbrb_exp[o] = P.undef; return; }
        const double h = 6.626070040e-34, c = 2.99792458e8, kB = 1.38064852e-23, pi = 3.14159265358979323846;
        const double alT = h * c / kB, bigC = 2.0 * h * (c * c);
        const double Fband = (double)f;
        const double Bmean = Fband / (pi * (double)(P.wn2[ib] - P.wn1[ib]));
        const R wnMid = (R)((double)(P.wn1[ib] + P.wn2[ib]) / 2.0);
        const R wn3 = (wnMid * wnMid) * wnMid;       // wn**3 in the real kind of wn
        const double T = alT * (double)wnMid / log(bigC * (double)wn3 / Bmean + 1.0);
        P.tbrb_exp[o] = (R)T;
    }
}

// Chou-Suarez branch of LW_Driver: `irrad` takes the GEOS fields as they are (no flip, no unit conversion) and fills the INTERNAL
// fluxes itself; what the driver adds (IRR:2101-2108, :3601-3616): the derivatives irrad does not provide, the net fluxes of the four
// flavours, the sign of SFCEM, TS_INT.
template <typename R> struct LwcPost {
    int ncol, lm;
    const R *flxu, *flcu, *flau, *flxau, *flxd, *flcd, *flad, *flxad, *dfdts, *ts;
    R *sfcem_int;                                                         // in: as irrad leaves it (negative); out: positive
    R *flx_int, *flxa_int, *flc_int, *fla_int, *dfdtsc, *dfdtsna, *dfdtscna, *ts_int;
};
template <typename R> __global__ void __launch_bounds__(256) k_lwd_chou_post(LwcPost<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const size_t o = (size_t)blockIdx.y * P.ncol + ij;
    if (P.flx_int) P.flx_int[o] = P.flxd[o] + P.flxu[o];
    if (P.flxa_int) P.flxa_int[o] = P.flxad[o] + P.flxau[o];
    if (P.flc_int) P.flc_int[o] = P.flcd[o] + P.flcu[o];
    if (P.fla_int) P.fla_int[o] = P.flad[o] + P.flau[o];
    if (P.dfdtsc) P.dfdtsc[o] = 0;                       // Chou-Suarez has no clear-sky derivative (IRR:2104-2108)
    if (P.dfdtsna) P.dfdtsna[o] = P.dfdts[o];
    if (P.dfdtscna) P.dfdtscna[o] = 0;
    if (blockIdx.y == 0) {
        if (P.sfcem_int) P.sfcem_int[ij] = -P.sfcem_int[ij];
        if (P.ts_int) P.ts_int[ij] = P.ts[ij];
    }
}

// Chou-Suarez branch of SORADCORE (SOL:4484-4528): `sorad` takes the packed GEOS fields in their own layout and ordering; what the
// driver prepares for it: interface pressures in hPa, ozone as a non-negative MASS fraction from the odd-oxygen prognostic (above 1 hPa
// scaled by exp(-1.5 (log10 p - 2)^2), p the mid-layer pressure in Pa), the four condensate species and their effective radii in
// microns as (ncol, LM, 4) arrays with MAPL_UNDEF radii replaced by 36 / 14 / 50 / 50 microns.  (The reference overwrites the undefined
// radii in its packed import buffer, which it discards afterwards; here the imports stay untouched.)  The relative humidity it also
// forms (:4489) is not an argument of SHRTWAVE.
template <typename R> struct SwcPrep {
    int ncol, lm;
    const R *ple, *ox, *q[4], *r[4];      // q / r: ice, liquid, rain, snow
    R o3fac, undef;                       // MAPL_O3MW / MAPL_AIRMW
    R *plhpa, *o3, *qq3, *rr3;            // (ncol, LM+1), (ncol, LM), (ncol, LM, 4) x 2
};
template <typename R> __global__ void __launch_bounds__(256) k_swc_prep(SwcPrep<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const int k = blockIdx.y;                                      // 0 .. LM
    const size_t o = (size_t)k * P.ncol + ij;
    const R pe = P.ple[o];
    P.plhpa[o] = pe * (R)0.01;                                     // SOL:4490
    if (k == P.lm) return;
    const R pl = (R)0.5 * (pe + P.ple[o + P.ncol]);                // SOL:4488
    R o3 = P.ox[o];                                                // SOL:4523-4533
    if (pl < (R)100.) {
        const R x = gr_log10<R>(pl) - (R)2.;
        o3 = o3 * gr_exp<R>((R)-1.5 * (x * x));
    }
    o3 = o3 * P.o3fac;
    P.o3[o] = o3 > (R)0. ? o3 : (R)0.;
    const R dflt[4] = {(R)36.e-6, (R)14.e-6, (R)50.e-6, (R)50.e-6};   // SOL:4508-4511
    const size_t sp = (size_t)P.lm * P.ncol;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        P.qq3[s * sp + o] = P.q[s][o];                             // SOL:4502-4505
        R r = P.r[s][o];
        if (r == P.undef) r = dflt[s];
        P.rr3[s * sp + o] = r * (R)1.e6;                           // SOL:4512-4515
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Update_Flx (IRR:3796-3999)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct LwUpd {
    int ncol, lm, rrtmg;                 // rrtmg != 0: the no-aerosol flavours are MAPL_UNDEF (IRR:3927-3990)
    int lev_mid_high, lev_low_mid;       // 1-based model levels separating the cloud super-layers (IRR:3811-3829)
    R undef;
    // internal state
    const R *tsinst, *ts_int, *sfcem_int, *fcld;
    const R *flx_int, *flxa_int, *flc_int, *fla_int, *flxu_int, *flxau_int, *flcu_int, *flau_int, *flxd_int, *flxad_int, *flcd_int,
        *flad_int, *dfdts, *dfdtsna, *dfdtsc, *dfdtscna;
    // exports, (IM,JM,0:LM)
    R *flx, *flxa, *flc, *fla, *flxu, *flxau, *flcu, *flau, *flxd, *flxad, *flcd, *flad;
    // exports, (IM,JM)
    R *olr, *olra, *olc, *ola, *olcc5, *dsfdts, *sfcem, *lws, *lwsa, *lcs, *las, *lcsc5, *flns, *flnsna, *flnsc, *flnsa, *dsfdts0,
        *sfcem0, *tsreff, *cldtt;
};

// V consecutive columns per thread (V = 4 floats / 2 doubles = one 16-byte access when the column count and the field
// addresses allow, else V = 1): these kernels do nothing but stream, and wide accesses are what keeps enough bytes in flight
template <typename R, int V> GR_DEV void ldv(const R *__restrict__ p, size_t o, R (&x)[V])
{
    if constexpr (V == 1) x[0] = p[o];
    else if constexpr (sizeof(R) == 4) { const float4 t = *reinterpret_cast<const float4 *>(p + o); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
    else { const double2 t = *reinterpret_cast<const double2 *>(p + o); x[0] = t.x; x[1] = t.y; }
}
template <typename R, int V> GR_DEV void stv(R *__restrict__ p, size_t o, const R (&x)[V])
{
    if constexpr (V == 1) p[o] = x[0];
    else if constexpr (sizeof(R) == 4) { float4 t; t.x = x[0]; t.y = x[1]; t.z = x[2]; t.w = x[3]; *reinterpret_cast<float4 *>(p + o) = t; }
    else { double2 t; t.x = x[0]; t.y = x[1]; *reinterpret_cast<double2 *>(p + o) = t; }
}
#define VFOR for (int v = 0; v < V; v++)
// out = expr, V columns at a time; `expr` may use the per-column arrays declared before it, indexed [v]
#define VSET(out, off, expr) do { if (out) { R r_[V]; VFOR r_[v] = (expr); stv<R, V>(out, off, r_); } } while (0)

// one thread per (V columns, level K = 0..LM); the 2-D exports are written by the K = 0 / K = LM threads
template <typename R, int V> __global__ void __launch_bounds__(256) k_lw_update_flx(LwUpd<R> U)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (ij >= U.ncol) return;
    const int n = U.ncol, lm = U.lm, K = blockIdx.y;
    const size_t o = (size_t)K * n + ij;
    const bool rr = U.rrtmg != 0;
    const R undef = U.undef;
    R delt[V], tsi[V], d[V], dc[V], dna[V], dcna[V], a[V], b[V];
    ldv<R, V>(U.tsinst, ij, tsi); ldv<R, V>(U.ts_int, ij, a);
    VFOR delt[v] = tsi[v] - a[v];                    // surface temperature change since the last full calculation
    ldv<R, V>(U.dfdts, o, d); ldv<R, V>(U.dfdtsc, o, dc);
    VFOR { dna[v] = 0; dcna[v] = 0; }
    if (!rr) { ldv<R, V>(U.dfdtsna, o, dna); ldv<R, V>(U.dfdtscna, o, dcna); }
    const bool edge = K == 0 || K == lm;
    R fx[V], fc[V], fxa[V], fca[V];
    VFOR { fx[v] = 0; fc[v] = 0; fxa[v] = 0; fca[v] = 0; }
    if (U.flx || edge) ldv<R, V>(U.flx_int, o, fx);
    if (U.flc || edge) ldv<R, V>(U.flc_int, o, fc);
    if (!rr && (U.flxa || edge)) ldv<R, V>(U.flxa_int, o, fxa);
    if (!rr && (U.fla || edge)) ldv<R, V>(U.fla_int, o, fca);
    // net downward, negated upward (linearised), downward (not linearised)
    VSET(U.flx, o, fx[v] + d[v] * delt[v]);
    VSET(U.flc, o, fc[v] + dc[v] * delt[v]);
    if (U.flxu) { ldv<R, V>(U.flxu_int, o, a); VSET(U.flxu, o, a[v] + d[v] * delt[v]); }
    if (U.flcu) { ldv<R, V>(U.flcu_int, o, a); VSET(U.flcu, o, a[v] + dc[v] * delt[v]); }
    if (U.flxd) { ldv<R, V>(U.flxd_int, o, a); VSET(U.flxd, o, a[v]); }
    if (U.flcd) { ldv<R, V>(U.flcd_int, o, a); VSET(U.flcd, o, a[v]); }
    VSET(U.flxa, o, rr ? undef : fxa[v] + dna[v] * delt[v]);
    VSET(U.fla, o, rr ? undef : fca[v] + dcna[v] * delt[v]);
    if (U.flxau) { if (!rr) ldv<R, V>(U.flxau_int, o, a); VSET(U.flxau, o, rr ? undef : a[v] + dna[v] * delt[v]); }
    if (U.flau) { if (!rr) ldv<R, V>(U.flau_int, o, a); VSET(U.flau, o, rr ? undef : a[v] + dcna[v] * delt[v]); }
    if (U.flxad) { if (!rr) ldv<R, V>(U.flxad_int, o, a); VSET(U.flxad, o, rr ? undef : a[v]); }
    if (U.flad) { if (!rr) ldv<R, V>(U.flad_int, o, a); VSET(U.flad, o, rr ? undef : a[v]); }
    if (!edge) return;
    // 2-D total cloud fraction, max overlap within / random between the super-layers (IRR:3831-3846)
    R cldtt[V];
    VFOR cldtt[v] = 0;
    if (U.cldtt || U.olcc5 || U.lcsc5) {
        R m1[V], m2[V], m3[V];
        VFOR { m1[v] = 0; m2[v] = 0; m3[v] = 0; }
        for (int k = 1; k <= U.lev_mid_high - 1; k++) { ldv<R, V>(U.fcld, (size_t)(k - 1) * n + ij, b); VFOR m1[v] = m1[v] > b[v] ? m1[v] : b[v]; }
        for (int k = U.lev_mid_high; k <= U.lev_low_mid - 1; k++) { ldv<R, V>(U.fcld, (size_t)(k - 1) * n + ij, b); VFOR m2[v] = m2[v] > b[v] ? m2[v] : b[v]; }
        for (int k = U.lev_low_mid; k <= lm; k++) { ldv<R, V>(U.fcld, (size_t)(k - 1) * n + ij, b); VFOR m3[v] = m3[v] > b[v] ? m3[v] : b[v]; }
        VFOR {
            R x = ((R)1 - m1[v]);
            x = x * ((R)1 - m2[v]);
            cldtt[v] = (R)1.0 - x * ((R)1 - m3[v]);
        }
    }
    if (K == 0) {
        // TOA: outgoing longwave radiation (IRR:3879-3893)
        VSET(U.olr, ij, -(fx[v] + d[v] * delt[v]));
        VSET(U.olc, ij, -(fc[v] + dc[v] * delt[v]));
        VSET(U.olra, ij, rr ? undef : -(fxa[v] + dna[v] * delt[v]));
        VSET(U.ola, ij, rr ? undef : -(fca[v] + dcna[v] * delt[v]));
        VSET(U.olcc5, ij, cldtt[v] <= (R)0.05 ? -(fc[v] + dc[v] * delt[v]) : undef);
        VSET(U.cldtt, ij, cldtt[v]);
    }
    if (K == lm) {
        // surface (IRR:3895-3925, :3991-3999)
        R se[V];
        ldv<R, V>(U.sfcem_int, ij, se);
        VSET(U.dsfdts, ij, -d[v]);
        VSET(U.sfcem, ij, se[v] - d[v] * delt[v]);
        VSET(U.lws, ij, fx[v] + se[v]);
        VSET(U.lcs, ij, fc[v] + se[v]);
        VSET(U.lwsa, ij, rr ? undef : fxa[v] + se[v]);
        VSET(U.las, ij, rr ? undef : fca[v] + se[v]);
        VSET(U.lcsc5, ij, cldtt[v] <= (R)0.05 ? fc[v] + se[v] : undef);
        VSET(U.flns, ij, fx[v] + d[v] * delt[v]);
        VSET(U.flnsc, ij, fc[v] + dc[v] * delt[v]);
        VSET(U.flnsna, ij, rr ? undef : fxa[v] + dna[v] * delt[v]);
        VSET(U.flnsa, ij, rr ? undef : fca[v] + dcna[v] * delt[v]);
        VSET(U.dsfdts0, ij, -d[v]);
        VSET(U.sfcem0, ij, se[v] - d[v] * delt[v]);
        VSET(U.tsreff, ij, tsi[v]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// SORADCORE, RRTMG branch (works on the packed daytime columns; ple has LM+1 levels 1..LM+1)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct SwdArgs {
    int ncol, lm, nb;                 // nb = 14 aerosol bands
    int iceflg, liqflg;
    const R *ple, *pl, *t, *q, *o3, *ch4, *cl, *ts;
    const R *qq_ice, *qq_liq, *rr_ice, *rr_liq;      // QQ3(:,:,1|2), RR3(:,:,1|2)
    R *taua, *ssaa, *asya;                           // (ncol,LM,nb) GEOS side: normalised IN PLACE like the reference (SOL:6116-6125)
    R co2, o2, airmw_over_h2omw, airmw_over_o3mw, rgas, grav;
    R *play, *plev, *tlay, *tlev, *h2o, *o3_r, *co2_r, *ch4_r, *o2_r, *cldf, *ciwp, *clwp, *rei, *rel, *zl, *tauaer, *ssaaer, *asmaer;
};

template <typename R> GR_DEV R swd_tlev(const SwdArgs<R> &A, int k, int ij)      // TLEV(1..LM+1), SOL:6172-6176
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int n = A.ncol, lm = A.lm;
    if (k == lm + 1) return A.ts[ij];
    if (k == 1) k = 2;
    auto PLE = [&](int l) { return A.ple[(size_t)(l - 1) * n + ij]; };
    auto T = [&](int l) { return A.t[(size_t)(l - 1) * n + ij]; };
    const R dpk = PLE(k + 1) - PLE(k), dpm = PLE(k) - PLE(k - 1);        // DPR(k), DPR(k-1)
    return (T(k - 1) * dpk + T(k) * dpm) / (dpm + dpk);
}

template <typename R> __global__ void __launch_bounds__(256) k_swd_prep(SwdArgs<R> A)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= A.ncol) return;
    const int n = A.ncol, lm = A.lm, K = blockIdx.y + 1, LV = lm - K + 1;
    const size_t o = (size_t)(K - 1) * n + ij, g = (size_t)(LV - 1) * n + ij;
    const R dpr = A.ple[(size_t)LV * n + ij] - A.ple[(size_t)(LV - 1) * n + ij];      // DPR(LV) = PLE(LV+1) - PLE(LV)
    const R xx = (R)1.02 * (R)100 * dpr;
    A.ciwp[o] = xx * A.qq_ice[g];
    A.clwp[o] = xx * A.qq_liq[g];
    R reice = A.rr_ice[g], reliq = A.rr_liq[g];
    // SW limits (SOL:6144-6170): liqflg 0 is 10..30 here
    if (A.liqflg == 0) reliq = clampr<R>(reliq, (R)10., (R)30.);
    else if (A.liqflg == 1) reliq = clampr<R>(reliq, (R)2.5, (R)60.);
    if (A.iceflg == 0) reice = clampr<R>(reice, (R)10., (R)30.);
    else if (A.iceflg == 1) reice = clampr<R>(reice, (R)13., (R)130.);
    else if (A.iceflg == 2) reice = clampr<R>(reice, (R)5., (R)131.);
    else if (A.iceflg == 3) reice = clampr<R>(reice, (R)5., (R)140.);
    else if (A.iceflg == 4) reice = clampr<R>(reice * (R)2., (R)1., (R)200.);
    A.rei[o] = reice; A.rel[o] = reliq;
    // PLE_R(1:LM+1) = PLE(LM+1:1)/100, TLEV_R likewise (SOL:6180-6181)
    A.plev[o] = A.ple[(size_t)LV * n + ij] / (R)100.;      // PLE(LV+1)
    A.tlev[o] = swd_tlev<R>(A, LV + 1, ij);
    if (K == lm) {
        A.plev[(size_t)lm * n + ij] = A.ple[ij] / (R)100.;
        A.tlev[(size_t)lm * n + ij] = swd_tlev<R>(A, 1, ij);
    }
    auto pos = [](R x) { return x < 0 ? (R)0 : x; };
    A.play[o] = A.pl[g] / (R)100.;
    A.tlay[o] = A.t[g];
    const R q = A.q[g];
    A.h2o[o] = pos(q / ((R)1. - q) * A.airmw_over_h2omw);
    A.o3_r[o] = pos(A.o3[g] * A.airmw_over_o3mw);
    A.ch4_r[o] = pos(A.ch4[g]);
    A.co2_r[o] = pos(A.co2);
    A.o2_r[o] = pos(A.o2);
    A.cldf[o] = pos(A.cl[g]);
    // aerosols: normalise (SOL:6116-6125) and flip (SOL:6210-6212)
    for (int b = 0; b < A.nb; b++) {
        R ta = 0, ss = 0, as = 0;
        if (A.taua) {
            const size_t ga = ((size_t)b * lm + (LV - 1)) * n + ij;
            ta = A.taua[ga]; ss = A.ssaa[ga]; as = A.asya[ga];
            if (ta > 0 && ss > 0) { as = as / ss; ss = ss / ta; }
            else { ta = 0; ss = 0; as = 0; }
            A.taua[ga] = ta; A.ssaa[ga] = ss; A.asya[ga] = as;
        }
        const size_t oa = ((size_t)b * lm + (K - 1)) * n + ij;
        A.tauaer[oa] = ta; A.ssaaer[oa] = ss; A.asmaer[oa] = as;
    }
}

// ZL_R (SOL:6200-6207): note the level index differs from LW (levels are 1-based here)
template <typename R> __global__ void __launch_bounds__(256) k_swd_zm(SwdArgs<R> A)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= A.ncol) return;
    const int n = A.ncol, lm = A.lm;
    R z = 0;
    A.zl[ij] = 0;
    R plm = A.play[ij];
    for (int k = 2; k <= lm; k++) {
        const R plk = A.play[(size_t)(k - 1) * n + ij];
        z = z + A.rgas * A.tlev[(size_t)(k - 1) * n + ij] / A.grav * (plm - plk) / A.plev[(size_t)(k - 1) * n + ij];
        A.zl[(size_t)(k - 1) * n + ij] = z;
        plm = plk;
    }
}

template <typename R> struct SwdPost {
    int ncol, lm, ngpt, aerosols;
    R undef;
    const R *swuflx, *swdflx, *swuflxc, *swdflxc;          // RRTMG (ncol, LM+1), 1 = surface
    const int32_t *clearCounts;
    const R *cotn[4], *cotd[4];                            // COTN?P / COTD?P: T, H, M, L
    R *fsw, *fsc, *fswu, *fscu;                            // (ncol, LM+1) model ordering
    R *cldts, *cldhs, *cldms, *cldls, *cot[4];
};

template <typename R> __global__ void __launch_bounds__(256) k_swd_post(SwdPost<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= P.ncol) return;
    const int n = P.ncol, lm = P.lm, L = blockIdx.y;      // model level index 0..LM of the un-flipped arrays
    const size_t s = (size_t)(lm - L) * n + ij, o = (size_t)L * n + ij;
    const R u = P.swuflx[s], d = P.swdflx[s], uc = P.swuflxc[s], dcl = P.swdflxc[s];
    if (P.fsw) P.fsw[o] = d - u;                          // SOL:6447-6450
    if (P.fsc) P.fsc[o] = dcl - uc;
    if (P.fswu) P.fswu[o] = u;
    if (P.fscu) P.fscu[o] = uc;
    if (L == 0) {
        const R ng = (R)P.ngpt;
        if (P.aerosols) {                                  // SOL:6405-6410
            if (P.cldts) P.cldts[ij] = (R)1. - (R)P.clearCounts[(size_t)0 * n + ij] / ng;
            if (P.cldhs) P.cldhs[ij] = (R)1. - (R)P.clearCounts[(size_t)1 * n + ij] / ng;
            if (P.cldms) P.cldms[ij] = (R)1. - (R)P.clearCounts[(size_t)2 * n + ij] / ng;
            if (P.cldls) P.cldls[ij] = (R)1. - (R)P.clearCounts[(size_t)3 * n + ij] / ng;
        }
        for (int k = 0; k < 4; k++)                        // SOL:6416-6438
            if (P.cot[k]) {
                const R a = P.cotn[k][ij], b = P.cotd[k][ij];
                P.cot[k][ij] = (a > 0 && b > 0) ? a / b : P.undef;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// UPDATE_EXPORT, 2-D block (SOL:7403-7533): albedo exports, total surface albedo, incident / surface fluxes.  One thread per column.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct SwSfc {
    int ncol, lm;
    R undef;
    const R *slr, *zth, *alb_imp[4], *dn[6];          // ALBVF ALBVR ALBNF ALBNR; DRUVRN DFUVRN DRPARN DFPARN DRNIRN DFNIRN
    const R *fswn, *fscn, *fswnan, *fscnan;           // (ncol,0:LM)
    R *alb_exp[4], *albedo, *slrtp, *dx[6], *drn[3], *slrsf, *slrsfc, *slrsfna, *slrsfcna, *slrsuf, *slrsufc, *slrsufna, *slrsufcna;
};
template <typename R> __global__ void __launch_bounds__(256) k_sw_update_surface(SwSfc<R> U)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= U.ncol) return;
    const size_t sfc = (size_t)U.lm * U.ncol + ij;
    const R slr = U.slr[ij], undef = U.undef;
    for (int k = 0; k < 4; k++)
        if (U.alb_exp[k]) U.alb_exp[k][ij] = slr > 0 ? U.alb_imp[k][ij] * (R)1. : undef;      // FAC = 1. (SOL:7406)
    R d[6];
    for (int k = 0; k < 6; k++) d[k] = U.dn[k][ij];
    const R sum6 = d[0] + d[1] + d[2] + d[3] + d[4] + d[5];      // DRUVRN+DFUVRN+DRPARN+DFPARN+DRNIRN+DFNIRN, left to right
    R alb = sum6;
    if (slr > (R)0.0 && alb > (R)0.0) {
        R x = (R)1.0 - U.fswn[sfc] / alb;
        x = x > (R).01 ? x : (R).01;                 // max(x, .01)
        alb = x < (R)0.9 ? x : (R)0.9;               // min(.., 0.9)
    } else alb = undef;
    if (U.albedo) U.albedo[ij] = alb;
    if (U.slrtp) U.slrtp[ij] = slr;
    for (int k = 0; k < 6; k++)
        if (U.dx[k]) U.dx[k][ij] = d[k] * slr;
    if (U.drn[0] || U.drn[1] || U.drn[2]) {
        const R zth = U.zth[ij] > (R)0.0 ? U.zth[ij] : (R)0.0;          // ZTH = max(ZTH,0.0)
        const R sln = zth > (R)0.0 ? slr / zth : (R)0.0;
        if (U.drn[0]) U.drn[0][ij] = d[0] * sln;
        if (U.drn[1]) U.drn[1][ij] = d[2] * sln;
        if (U.drn[2]) U.drn[2][ij] = d[4] * sln;
    }
    if (U.slrsf) U.slrsf[ij] = sum6 * slr;
    const bool def = alb != undef;
    if (U.slrsfc) U.slrsfc[ij] = def ? (U.fscn[sfc] * slr) / ((R)1. - alb) : (R)0.0;
    if (U.slrsfna) U.slrsfna[ij] = def ? (U.fswnan[sfc] * slr) / ((R)1. - alb) : (R)0.0;
    if (U.slrsfcna) U.slrsfcna[ij] = def ? (U.fscnan[sfc] * slr) / ((R)1. - alb) : (R)0.0;
    if (U.slrsuf) U.slrsuf[ij] = alb * sum6 * slr;
    if (U.slrsufc) U.slrsufc[ij] = def ? alb * (U.fscn[sfc] / ((R)1. - alb)) * slr : (R)0.0;
    if (U.slrsufna) U.slrsufna[ij] = def ? alb * (U.fswnan[sfc] / ((R)1. - alb)) * slr : (R)0.0;
    if (U.slrsufcna) U.slrsufcna[ij] = def ? alb * (U.fscnan[sfc] / ((R)1. - alb)) * slr : (R)0.0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// UPDATE_EXPORT, flux part (SOL:7540-7579): exports = normalised internals x SLR
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct SwUpd {
    int ncol, lm, nbands;
    const R *slr;
    const R *fswn, *fscn, *fswnan, *fscnan, *fswun, *fscun, *fswunan, *fscunan;      // (IM,JM,0:LM)
    const R *fswbandn, *fswbandnan;                                                  // (IM,JM,nbands)
    R *fsw, *fsc, *fswna, *fscna, *fswu, *fscu, *fswuna, *fscuna, *fswd, *fscd, *fswdna, *fscdna;
    R *fswband, *fswbandna;
    R *rsr, *rsc, *rsrna, *rscna, *rsrs, *rscs, *rsrsna, *rscsna, *osr, *osrclr, *osrna, *osrcna;
};

// blockIdx.y = level 0..LM, then the bands; V columns per thread
template <typename R, int V> __global__ void __launch_bounds__(256) k_sw_update_export(SwUpd<R> U)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (ij >= U.ncol) return;
    const int n = U.ncol, lm = U.lm;
    R slr[V], a[V];
    ldv<R, V>(U.slr, ij, slr);
    if ((int)blockIdx.y > lm) {
        const size_t o = (size_t)(blockIdx.y - lm - 1) * n + ij;
        if (U.fswband) { ldv<R, V>(U.fswbandn, o, a); VSET(U.fswband, o, a[v] * slr[v]); }
        if (U.fswbandna) { ldv<R, V>(U.fswbandnan, o, a); VSET(U.fswbandna, o, a[v] * slr[v]); }
        return;
    }
    const int L = blockIdx.y;
    const size_t o = (size_t)L * n + ij;
    const bool edge = L == 0 || L == lm;
    R w[V], c[V], wna[V], cna[V], wu[V], cu[V], wuna[V], cuna[V];
    VFOR { w[v] = 0; c[v] = 0; wna[v] = 0; cna[v] = 0; wu[v] = 0; cu[v] = 0; wuna[v] = 0; cuna[v] = 0; }
    if (U.fsw || U.fswd || (edge && (U.rsr || U.rsrs || U.osr))) ldv<R, V>(U.fswn, o, w);
    if (U.fsc || U.fscd || (edge && (U.rsc || U.rscs || U.osrclr))) ldv<R, V>(U.fscn, o, c);
    if (U.fswna || U.fswdna || (edge && (U.rsrna || U.rsrsna || U.osrna))) ldv<R, V>(U.fswnan, o, wna);
    if (U.fscna || U.fscdna || (edge && (U.rscna || U.rscsna || U.osrcna))) ldv<R, V>(U.fscnan, o, cna);
    if (U.fswu || U.fswd) ldv<R, V>(U.fswun, o, wu);
    if (U.fscu || U.fscd) ldv<R, V>(U.fscun, o, cu);
    if (U.fswuna || U.fswdna) ldv<R, V>(U.fswunan, o, wuna);
    if (U.fscuna || U.fscdna) ldv<R, V>(U.fscunan, o, cuna);
    VSET(U.fsw, o, w[v] * slr[v]);
    VSET(U.fsc, o, c[v] * slr[v]);
    VSET(U.fswna, o, wna[v] * slr[v]);
    VSET(U.fscna, o, cna[v] * slr[v]);
    VSET(U.fswu, o, wu[v] * slr[v]);
    VSET(U.fscu, o, cu[v] * slr[v]);
    VSET(U.fswuna, o, wuna[v] * slr[v]);
    VSET(U.fscuna, o, cuna[v] * slr[v]);
    VSET(U.fswd, o, (w[v] + wu[v]) * slr[v]);
    VSET(U.fscd, o, (c[v] + cu[v]) * slr[v]);
    VSET(U.fswdna, o, (wna[v] + wuna[v]) * slr[v]);
    VSET(U.fscdna, o, (cna[v] + cuna[v]) * slr[v]);
    if (L == 0) {
        VSET(U.rsr, ij, w[v] * slr[v]);
        VSET(U.rsc, ij, c[v] * slr[v]);
        VSET(U.rsrna, ij, wna[v] * slr[v]);
        VSET(U.rscna, ij, cna[v] * slr[v]);
        VSET(U.osr, ij, ((R)1. - w[v]) * slr[v]);
        VSET(U.osrclr, ij, ((R)1. - c[v]) * slr[v]);
        VSET(U.osrna, ij, ((R)1. - wna[v]) * slr[v]);
        VSET(U.osrcna, ij, ((R)1. - cna[v]) * slr[v]);
    }
    if (L == lm) {
        VSET(U.rsrs, ij, w[v] * slr[v]);
        VSET(U.rscs, ij, c[v] * slr[v]);
        VSET(U.rsrsna, ij, wna[v] * slr[v]);
        VSET(U.rscsna, ij, cna[v] * slr[v]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// parent RUN: heating rates from the net fluxes (GEOS_RadiationGridComp.F90:798-819)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R> struct RadTend {
    int ncol, lm;
    R grav, cp;
    const R *ple, *flw, *fsw, *flwclr, *fswclr, *fswna, *fla, *fscna;      // (IM,JM,0:LM); any flux may be null with its export
    const R *dsfdts, *sfcem, *trd;
    R *dtdt, *radlw, *radsw, *radlwc, *radswc, *radswna, *radlwcna, *radswcna;      // (IM,JM,LM)
    R *blw, *alw, *radsrf;
};

template <typename R, int V> __global__ void __launch_bounds__(256) k_rad_tendencies(RadTend<R> P)
{
#pragma clang fp contract(off)      // the statements below are the reference's, operation by operation
    const int ij = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (ij >= P.ncol) return;
    const int n = P.ncol, lm = P.lm, k = blockIdx.y;      // layer k+1 lies between levels k and k+1
    const size_t t = (size_t)k * n + ij, b = (size_t)(k + 1) * n + ij;
    const R gcp = P.grav / P.cp;
    R lt[V], lb[V], st[V], sb[V], x[V], y[V], dmi[V];
    VFOR { lt[v] = 0; lb[v] = 0; st[v] = 0; sb[v] = 0; dmi[v] = 0; }
    const bool last = k == lm - 1;
    if (P.dtdt || P.radlw || (last && P.radsrf)) { ldv<R, V>(P.flw, t, lt); ldv<R, V>(P.flw, b, lb); }
    if (P.dtdt || P.radsw || (last && P.radsrf)) { ldv<R, V>(P.fsw, t, st); ldv<R, V>(P.fsw, b, sb); }
    VSET(P.dtdt, t, ((lt[v] - lb[v]) + (st[v] - sb[v])) * gcp);
    if (P.radlw || P.radsw || P.radlwc || P.radswc || P.radswna || P.radlwcna || P.radswcna) {
        ldv<R, V>(P.ple, t, x); ldv<R, V>(P.ple, b, y);
        VFOR dmi[v] = P.grav / (P.cp * (y[v] - x[v]));
    }
    VSET(P.radlw, t, (lt[v] - lb[v]) * dmi[v]);
    VSET(P.radsw, t, (st[v] - sb[v]) * dmi[v]);
    if (P.radlwc) { ldv<R, V>(P.flwclr, t, x); ldv<R, V>(P.flwclr, b, y); VSET(P.radlwc, t, (x[v] - y[v]) * dmi[v]); }
    if (P.radswc) { ldv<R, V>(P.fswclr, t, x); ldv<R, V>(P.fswclr, b, y); VSET(P.radswc, t, (x[v] - y[v]) * dmi[v]); }
    if (P.radswna) { ldv<R, V>(P.fswna, t, x); ldv<R, V>(P.fswna, b, y); VSET(P.radswna, t, (x[v] - y[v]) * dmi[v]); }
    if (P.radlwcna) { ldv<R, V>(P.fla, t, x); ldv<R, V>(P.fla, b, y); VSET(P.radlwcna, t, (x[v] - y[v]) * dmi[v]); }
    if (P.radswcna) { ldv<R, V>(P.fscna, t, x); ldv<R, V>(P.fscna, b, y); VSET(P.radswcna, t, (x[v] - y[v]) * dmi[v]); }
    if (last) {
        if (P.blw) { ldv<R, V>(P.dsfdts, ij, x); VSET(P.blw, ij, x[v]); }
        if (P.alw) { R se[V], tr[V]; ldv<R, V>(P.dsfdts, ij, x); ldv<R, V>(P.sfcem, ij, se); ldv<R, V>(P.trd, ij, tr); VSET(P.alw, ij, se[v] - x[v] * tr[v]); }
        VSET(P.radsrf, ij, sb[v] + lb[v]);
    }
}

#undef VFOR
#undef VSET

// ---------------------------------------------------------------------------------------------------
// Lit-column compaction of the solar component (GEOS_SolarGridComp.F90:3686 `daytime = ZTH > 0.`, PackIt / UnPackIt :7753-7799):
// SORADCORE only sees the NumLit daytime columns, packed to the front of every field in (i, j) order.
//   k_lit_index: stable list of the lit columns (one 1024-thread block, like k_partition): idx[m] = column of packed position m,
//                pos[column] = packed position or -1, *nlit = NumLit
//   k_lit_pack  : Packed(m, l) = UnPacked(idx[m], l)
//   k_lit_unpack: UnPacked(idx[m], l) = Packed(m, l); dark columns get DEFAULT when one is given (PRESENT(DEFAULT)), else keep their value
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(1024) k_lit_index(int ncol, const R *__restrict__ zth, int32_t *__restrict__ idx, int32_t *__restrict__ pos,
                                                    int32_t *__restrict__ nlit)
{
    __shared__ int cnt[1024];
    const int t = threadIdx.x;
    const int chunk = (ncol + 1023) / 1024;
    const int b = t * chunk, e = (b + chunk < ncol) ? b + chunk : ncol;
    int c = 0;
    for (int i = b; i < e; i++) c += zth[i] > (R)0;
    cnt[t] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // inclusive Hillis-Steele scan
        const int v = t >= d ? cnt[t - d] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    int m = cnt[t] - c;                              // lit columns before this thread's chunk
    for (int i = b; i < e; i++) {
        if (zth[i] > (R)0) { idx[m] = i; pos[i] = m; m++; }
        else pos[i] = -1;
    }
    if (t == 0) *nlit = cnt[1023];
}
template <typename R>
__global__ void __launch_bounds__(256) k_lit_pack(int pdim, int udim, const int32_t *__restrict__ idx, const int32_t *__restrict__ nlit,
                                                  const R *__restrict__ unpacked, R *__restrict__ packed)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (m >= *nlit || m >= pdim) return;             // a packed buffer shorter than NumLit holds the first pdim lit columns
    packed[(size_t)l * pdim + m] = unpacked[(size_t)l * udim + idx[m]];
}
template <typename R>
__global__ void __launch_bounds__(256) k_lit_unpack(int pdim, int udim, const int32_t *__restrict__ pos, const R *__restrict__ packed,
                                                    R *__restrict__ unpacked, int use_default, R dflt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (i >= udim) return;
    const int m = pos[i];
    if (m >= 0 && m < pdim) unpacked[(size_t)l * udim + i] = packed[(size_t)l * pdim + m];
    else if (m < 0 && use_default) unpacked[(size_t)l * udim + i] = dflt;
}

}  // namespace geosrad
