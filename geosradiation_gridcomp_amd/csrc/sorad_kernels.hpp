// sorad_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the Chou-Suarez shortwave scheme `sorad`.
//
// Reference behaviour: GEOSsolar_GridComp/sorad.F90:43-1588 (SOLUV / SOLIR / CLDFLX inlined, O2 + CO2 flux reductions),
// deledd :1592-1706; cloud optics GEOS_RadiationShared/getvistau.code, getnirtau.code.  Non-OVERCAST build.
//
// sorad is 35 independent spectral passes (5 UV/PAR bands + 3 NIR bands x 10 k-values), each a set of delta-Eddington layers (no
// vertical dependence, fp64, the expensive part) + first-order vertical recurrences (adding over up to 8 sky situations):
//   k_sorad_prep   : per column   - scaled absorber amounts, cloud-group covers, cloud top
//   k_sorad_cloud  : per (column, optics group: UV/PAR + 3 NIR bands) - getvistau / getnirtau
//   k_sorad_pass   : per (column, pass), lane = column - deledd of the clear / cloudy portion of every layer, CLDFLX; the per-level arrays of
//                    the pass in HBM scratch planes [array][level][column] (coalesced); k_sorad_sum adds the passes up (default path)
//   k_sorad_col    : per column, lanes = (pass, level), the 35 passes on chip: no scratch (GEOSRAD_SORAD_PATH=col)
//   k_sorad_reduce : per column   - weighted sum over the passes (hk_uv, hk_ir), flux reductions, surface rescaling
// Sky situations of zero weight (ct = 0: a cloud group without cloud) are skipped: their contribution is `+ x * 0`.
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {

constexpr int SO_NPASS = 35;      // 5 + 3 * 10
constexpr int SO_NPLANE = 30;     // scratch planes of a pass in k_sorad_pass
constexpr int SO_NGRP = 4;        // cloud optics groups: 0 = UV/PAR, 1..3 = NIR bands

template <typename R> struct SoradDev {
    R zk_uv[5], wk_uv[5], ry_uv[5], xk_ir[10], ry_ir[3];
    const R *coa, *cah, *caib, *caif;     // Fortran (62,101), (43,37), (11,9,11), (9,11)
    R aig_uv[3], awg_uv[3], arg_uv[3], aib_uv, awb_uv[2], arb_uv[2], aib_nir, awb_nir[6], arb_nir[6], aia_nir[9], awa_nir[9], ara_nir[9],
        aig_nir[9], awg_nir[9], arg_nir[9];
};

template <typename R> struct SoradArgs {
    int m, ld, np, ict, icb, do_drfband;
    R co2;
    R hk[SO_NPASS];                       // hk_uv(1..5), then hk_ir(ib, ik) for ib = 1..3, ik = 1..10
    const R *cosz, *pl, *ta, *wa, *oa, *cwc, *fcld, *reff, *taua, *ssaa, *asya, *rsuvbm, *rsuvdf, *rsirbm, *rsirdf;
    // workspace
    R *lay;          // [4][K2][m]: dp, wh, oh, scal
    R *swh;          // [K2][m]    cumulative scaled water vapour (index k = 1..np+1)
    R *colv;         // [8][m]: cc1, cc2, cc3, wvtoa, o3toa, scal0, ntop (as real), spare
    R *cld;          // [SO_NGRP][4][K2][m]: tauclb, tauclf, asycl, ssacl
    R *scr;          // [SO_NPASS][SO_NPLANE][K2][m]: per-pass planes of k_sorad_pass (null on the k_sorad_col path)
    R *psum;         // [SO_NPASS][3][m]: fsdir, fsdif and the all-sky net flux at the surface of the pass
};
template <typename R> struct SoradOut { R *flx, *flc, *fdiruv, *fdifuv, *fdirpar, *fdifpar, *fdirir, *fdifir, *flxu, *flcu, *flx_sfc_band, *drband, *dfband; };

// deledd (:1592-1706) -- fp64 internally whatever the default real kind, as in the reference
template <typename R> GR_DEV void so_deledd(R tau1, R ssc1, R g01, R cza1, R &rr1, R &tt1, R &td1)
{
    double zth = (double)cza1;
    const double g0 = (double)g01, tau = (double)tau1, ssc = (double)ssc1;
    const double ff = g0 * g0;
    double xx = 1.0 - ff * ssc;
    const double taup = tau * xx, sscp = ssc * (1.0 - ff) / xx, gp = g0 / (1.0 + g0);
    xx = 3.0 * gp;
    const double gm1 = (7.0 - sscp * (4.0 + xx)) * 0.25, gm2 = -(1.0 - sscp * (4.0 - xx)) * 0.25;
    const double akk = sqrt((gm1 + gm2) * (gm1 - gm2));
    xx = akk * zth;
    double st7 = 1.0 - xx, st8 = 1.0 + xx, st3 = st7 * st8;
    if (fabs(st3) < 1.e-8) {
        zth = zth + 0.0010;
        if (zth > 1.0) zth = zth - 0.0020;
        xx = akk * zth; st7 = 1.0 - xx; st8 = 1.0 + xx; st3 = st7 * st8;
    }
    const double td = exp(-taup / zth);
    const double gm3 = (2.0 - zth * 3.0 * gp) * 0.25;
    xx = gm1 - gm2;
    const double alf1 = gm1 - gm3 * xx, alf2 = gm2 + gm3 * xx;
    xx = akk * 2.0;
    const double all = (gm3 - alf2 * zth) * xx * td, bll = (1.0 - gm3 + alf1 * zth) * xx;
    xx = akk * gm3;
    const double cll = (alf2 + xx) * st7, dll = (alf2 - xx) * st8;
    xx = akk * (1.0 - gm3);
    const double fll = (alf1 + xx) * st8, ell = (alf1 - xx) * st7;
    const double st2 = exp(-akk * taup), st4 = st2 * st2;
    const double st1 = sscp / ((akk + gm1 + (akk - gm1) * st4) * st3);
    double rr = (cll - dll * st4 - all * st2) * st1;
    double tt = -((fll - ell * st4) * td - bll * st2) * st1;
    rr = rr > 0 ? rr : 0;
    tt = tt > 0 ? tt : 0;
    tt = tt + td;
    td1 = (R)td; rr1 = (R)rr; tt1 = (R)tt;
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_prep (:271-357, 416-431, 1534-1541)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_prep(SoradArgs<R> A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.m) return;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2;
#define AP(a, k) a[(size_t)((k) - 1) * ld + i]
#define LAY(f, k) A.lay[((size_t)(f) * K2 + (k)) * m + i]
    const R xtoa = AP(A.pl, 1) > (R)1.e-3 ? AP(A.pl, 1) : (R)1.e-3;
    const R scal0 = xtoa * gr_pow<R>((R)0.5 * xtoa / (R)300., (R).8);
    const R o3toa = (R)1.02 * AP(A.oa, 1) * xtoa * (R)466.7 + (R)1.0e-8;
    const R wvtoa = (R)1.02 * AP(A.wa, 1) * scal0 * ((R)1.0 + (R)0.00135 * (AP(A.ta, 1) - (R)240.)) + (R)1.0e-9;
    R sw = wvtoa, cc1 = 0, cc2 = 0, cc3 = 0;
    int ntop = np + 1; bool found = false;
    A.swh[(size_t)1 * m + i] = sw;
    for (int k = 1; k <= np; k++) {
        const R dp = AP(A.pl, k + 1) - AP(A.pl, k);
        const R pa = (R)0.5 * (AP(A.pl, k) + AP(A.pl, k + 1));
        const R scal = dp * gr_pow<R>(pa / (R)300., (R).8);
        const R wh = (R)1.02 * AP(A.wa, k) * scal * ((R)1. + (R)0.00135 * (AP(A.ta, k) - (R)240.)) + (R)1.e-9;
        sw = sw + wh;
        A.swh[(size_t)(k + 1) * m + i] = sw;
        LAY(0, k) = dp; LAY(1, k) = wh; LAY(2, k) = (R)1.02 * AP(A.oa, k) * dp * (R)466.7 + (R)1.e-8; LAY(3, k) = scal;
        const R fc = AP(A.fcld, k);
        if (k < A.ict) cc1 = cc1 > fc ? cc1 : fc; else if (k < A.icb) cc2 = cc2 > fc ? cc2 : fc; else cc3 = cc3 > fc ? cc3 : fc;
        if (fc > (R)0.02 && !found) { found = true; ntop = k; }
    }
    R *cv = A.colv + i;
    cv[0 * (size_t)m] = cc1; cv[1 * (size_t)m] = cc2; cv[2 * (size_t)m] = cc3; cv[3 * (size_t)m] = wvtoa; cv[4 * (size_t)m] = o3toa;
    cv[5 * (size_t)m] = scal0; cv[6 * (size_t)m] = (R)ntop;
#undef AP
#undef LAY
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_cloud: getvistau (grp 0) / getnirtau (grp = NIR band 1..3), one thread per (column, group)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_cloud(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ib = blockIdx.y;
    if (i >= A.m) return;
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2, ict = A.ict, icb = A.icb;
    const R dm = (R)0.1, dt = (R)0.30103, da = (R)0.1, t1 = (R)-0.9031;
    const R cosz = A.cosz[i];
    const R cc[4] = {0, A.colv[0 * (size_t)m + i], A.colv[1 * (size_t)m + i], A.colv[2 * (size_t)m + i]};
#define CLD(f, k) A.cld[(((size_t)ib * 4 + (f)) * K2 + (k)) * m + i]
#define CAIB(a, b, c) T.caib[(((c) - 1) * 9 + ((b) - 1)) * 11 + ((a) - 1)]
#define CAIF(a, b) T.caif[((b) - 1) * 9 + ((a) - 1)]
#define N2(tab, j) tab[((j) - 1) * 3 + (ib - 1)]
    for (int k = 1; k <= np; k++) {
        const R dp_pa = A.lay[((size_t)0 * K2 + k) * m + i] * (R)100.;
        const R wp = (dp_pa * (R)1.0e3) / (R)9.80665;                 // MAPL_GRAV
        const R r1 = A.reff[((size_t)0 * np + (k - 1)) * ld + i], r2 = A.reff[((size_t)1 * np + (k - 1)) * ld + i],
                r4 = A.reff[((size_t)3 * np + (k - 1)) * ld + i];
        const R h1 = A.cwc[((size_t)0 * np + (k - 1)) * ld + i], h2 = A.cwc[((size_t)1 * np + (k - 1)) * ld + i],
                h3 = A.cwc[((size_t)2 * np + (k - 1)) * ld + i], h4 = A.cwc[((size_t)3 * np + (k - 1)) * ld + i];
        const R fc = A.fcld[(size_t)(k - 1) * ld + i];
        const R rs = r4 < (R)112.0 ? r4 : (R)112.0;
        R tc1, tc2, tc3, tc4;
        if (ib == 0) {
            tc1 = r1 <= 0 ? (R)0 : (wp * h1) * T.aib_uv / r1;
            tc2 = r2 <= 0 ? (R)0 : (wp * h2) * (T.awb_uv[0] + T.awb_uv[1] / r2);
            tc3 = (wp * h3) * T.arb_uv[0];
            tc4 = rs <= 0 ? (R)0 : (wp * h4) * T.aib_uv / rs;
        } else {
            tc1 = r1 <= 0 ? (R)0 : (wp * h1) * T.aib_nir / r1;
            tc2 = r2 <= 0 ? (R)0 : (wp * h2) * (N2(T.awb_nir, 1) + N2(T.awb_nir, 2) / r2);
            tc3 = (wp * h3) * N2(T.arb_nir, 1);
            tc4 = rs <= 0 ? (R)0 : (wp * h4) * T.aib_nir / rs;
        }
        R tb = 0, tf = 0;                                             // sums of the 4 species (taubeam / taudiff)
        const int kk = k < ict ? 1 : (k < icb ? 2 : 3);
        R tauc = tc1 + tc2 + tc3 + tc4;
        const bool cloudy = tauc > (R)0.02 && fc > (R)0.01;
        if (cloudy) {
            R fa = ib == 0 ? fc / cc[kk] : (cc[kk] != 0 ? fc / cc[kk] : (R)0);
            R tcap = tauc < (R)32. ? tauc : (R)32.;
            R fm = cosz / dm, ft = (gr_log10<R>(tcap) - t1) / dt;
            fa = fa / da;
            int im = (int)(fm + (R)1.5), it = (int)(ft + (R)1.5), ia = (int)(fa + (R)1.5);
            im = im > 2 ? im : 2; it = it > 2 ? it : 2; ia = ia > 2 ? ia : 2;
            im = im < 10 ? im : 10; it = it < 8 ? it : 8; ia = ia < 10 ? ia : 10;
            fm = fm - (R)(im - 1); ft = ft - (R)(it - 1); fa = fa - (R)(ia - 1);
            const R c0 = CAIB(im, it, ia);
            R xai = (-CAIB(im - 1, it, ia) * ((R)1. - fm) + CAIB(im + 1, it, ia) * ((R)1. + fm)) * fm * (R).5 + c0 * ((R)1. - fm * fm);
            xai = xai + (-CAIB(im, it - 1, ia) * ((R)1. - ft) + CAIB(im, it + 1, ia) * ((R)1. + ft)) * ft * (R).5 + c0 * ((R)1. - ft * ft);
            xai = xai + (-CAIB(im, it, ia - 1) * ((R)1. - fa) + CAIB(im, it, ia + 1) * ((R)1. + fa)) * fa * (R).5 + c0 * ((R)1. - fa * fa);
            xai = xai - (R)2. * c0;
            xai = xai > 0 ? xai : (R)0; xai = xai < 1 ? xai : (R)1;
            tb = tc1 * xai + tc2 * xai + tc3 * xai + tc4 * xai;
            const R f0 = CAIF(it, ia);
            xai = (-CAIF(it - 1, ia) * ((R)1. - ft) + CAIF(it + 1, ia) * ((R)1. + ft)) * ft * (R).5 + f0 * ((R)1. - ft * ft);
            xai = xai + (-CAIF(it, ia - 1) * ((R)1. - fa) + CAIF(it, ia + 1) * ((R)1. + fa)) * fa * (R).5 + f0 * ((R)1. - fa * fa);
            xai = xai - f0;
            xai = xai > 0 ? xai : (R)0; xai = xai < 1 ? xai : (R)1;
            tf = tc1 * xai + tc2 * xai + tc3 * xai + tc4 * xai;
        }
        R asy = 1, ssa = (R)0.99999;
        if (cloudy) {
            if (ib == 0) {
                const R g1 = (T.aig_uv[0] + (T.aig_uv[1] + T.aig_uv[2] * r1) * r1) * tc1;
                const R g2 = (T.awg_uv[0] + (T.awg_uv[1] + T.awg_uv[2] * r2) * r2) * tc2;
                const R g3 = T.arg_uv[0] * tc3;
                const R g4 = (T.aig_uv[0] + (T.aig_uv[1] + T.aig_uv[2] * rs) * rs) * tc4;
                asy = (g1 + g2 + g3 + g4) / tauc;
            } else {
                const R w1 = ((R)1. - (N2(T.aia_nir, 1) + (N2(T.aia_nir, 2) + N2(T.aia_nir, 3) * r1) * r1)) * tc1;
                const R w2 = ((R)1. - (N2(T.awa_nir, 1) + (N2(T.awa_nir, 2) + N2(T.awa_nir, 3) * r2) * r2)) * tc2;
                const R w3 = ((R)1. - N2(T.ara_nir, 1)) * tc3;
                const R w4 = ((R)1. - (N2(T.aia_nir, 1) + (N2(T.aia_nir, 2) + N2(T.aia_nir, 3) * rs) * rs)) * tc4;
                ssa = (w1 + w2 + w3 + w4) / tauc;
                const R g1 = (N2(T.aig_nir, 1) + (N2(T.aig_nir, 2) + N2(T.aig_nir, 3) * r1) * r1) * w1;
                const R g2 = (N2(T.awg_nir, 1) + (N2(T.awg_nir, 2) + N2(T.awg_nir, 3) * r2) * r2) * w2;
                const R g3 = N2(T.arg_nir, 1) * w3;
                const R g4 = (N2(T.aig_nir, 1) + (N2(T.aig_nir, 2) + N2(T.aig_nir, 3) * r4) * r4) * w4;      // reff(k,4), not the capped value
                if (w1 + w2 + w3 + w4 != 0) asy = (g1 + g2 + g3 + g4) / (w1 + w2 + w3 + w4);
            }
        }
        CLD(0, k) = tb; CLD(1, k) = tf; CLD(2, k) = asy; CLD(3, k) = ssa;
    }
#undef CLD
#undef CAIB
#undef CAIF
#undef N2
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_pass: one thread per (column, spectral pass).  Scratch planes of the pass (index q, level k):
//   0..9  : rr, tt, td, rs, ts of the clear (q = 2 f) and cloudy (q = 2 f + 1) portion of layer k (k = 0 above the model top,
//           np+1 = surface)
//   10..25: rra, rxa composites from the surface of the sky situations' variants at the level (q = 10 + 2 v, 11 + 2 v; see CLDFLX below)
//   26..29: fall, fclr, fupa, fupc
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_pass(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int pass = blockIdx.y;
    if (i >= A.m) return;
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2, ict = A.ict, icb = A.icb;
    const bool uv = pass < 5;
    const int ib = uv ? pass + 1 : (pass - 5) / 10 + 1, ik = uv ? 0 : (pass - 5) % 10 + 1;   // band in its region, k-value
    const int iv = uv ? ib : ib + 5;                                                         // aerosol band 1..8
    const int grp = uv ? 0 : ib;
    const R cz = A.cosz[i], dsm = (R)0.602;
    const R cc1 = A.colv[0 * (size_t)m + i], cc2 = A.colv[1 * (size_t)m + i], cc3 = A.colv[2 * (size_t)m + i];
    const R wvtoa = A.colv[3 * (size_t)m + i], o3toa = A.colv[4 * (size_t)m + i];
    R *S = A.scr + (size_t)pass * SO_NPLANE * K2 * m + i;
#define P(q, k) S[((size_t)(q) * K2 + (k)) * m]
#define LY(f, j, k) P(2 * (f) + (j) - 1, k)              // f: 0 rr 1 tt 2 td 3 rs 4 ts;  j: 1 clear, 2 cloudy
    // The sweeps below are first-order recurrences over the levels: each step loads a layer's five properties and stores the new
    // composites.  gfx9 tracks loads and stores with one in-order counter, so a load issued after a store cannot be consumed before
    // that store is acknowledged: every sweep therefore requests the NEXT level's properties before it stores the current results.
    struct L5 { R rr, tt, td, rs, ts; };
    auto ld5 = [&](int j, int k) { L5 l; l.rr = LY(0, j, k); l.tt = LY(1, j, k); l.td = LY(2, j, k); l.rs = LY(3, j, k); l.ts = LY(4, j, k); return l; };
    // boundary "layers": surface (np+1) and the layer above the model top (0)  (:365-387, 914-936)
    {
        const R rb = uv ? A.rsuvbm[i] : A.rsirbm[i], rd = uv ? A.rsuvdf[i] : A.rsirdf[i];
        const R td0 = uv ? gr_exp<R>(-(wvtoa * T.wk_uv[ib - 1] + o3toa * T.zk_uv[ib - 1]) / cz) : gr_exp<R>(-wvtoa * T.xk_ir[ik - 1] / cz);
        // the cloudy-portion planes (j = 2) of a group are only read when the group has cloud
        for (int j = 1; j <= 2; j++) {
            if (j == 1 || cc3 > 0) { LY(0, j, np + 1) = rb; LY(3, j, np + 1) = rd; LY(2, j, np + 1) = 0; LY(1, j, np + 1) = 0; LY(4, j, np + 1) = 0; }
            if (j == 1 || cc1 > 0) { LY(0, j, 0) = 0; LY(3, j, 0) = 0; LY(1, j, 0) = 1; LY(4, j, 0) = 1; LY(2, j, 0) = td0; }
        }
    }
    // ---- layers: clear and cloudy portion (:436-520, 996-1068) ---------------------------------------------------------
    struct In6 { R dp, wh, oh, ta, sa, as; };
    auto ldin = [&](int k) {
        In6 v;
        v.dp = A.lay[((size_t)0 * K2 + k) * m + i]; v.wh = A.lay[((size_t)1 * K2 + k) * m + i]; v.oh = A.lay[((size_t)2 * K2 + k) * m + i];
        const size_t ja = ((size_t)(iv - 1) * np + (k - 1)) * ld + i;
        v.ta = A.taua[ja]; v.sa = A.ssaa[ja]; v.as = A.asya[ja];
        return v;
    };
    In6 nin = ldin(1);
    for (int k = 1; k <= np; k++) {
        const In6 cin = nin;
        if (k + 1 <= np) nin = ldin(k + 1);      // before this level's stores (see the note on the in-order memory counter below)
        const R dp = cin.dp, wh = cin.wh, oh = cin.oh, ta_ = cin.ta, sa_ = cin.sa, as_ = cin.as;
        R taurs, tausto, ssatau;
        if (uv) {
            taurs = T.ry_uv[ib - 1] * dp;
            tausto = taurs + T.zk_uv[ib - 1] * oh + T.wk_uv[ib - 1] * wh + ta_ + (R)1.0e-7;
            ssatau = sa_ + taurs;
        } else {
            taurs = T.ry_ir[ib - 1] * dp;
            tausto = taurs + T.xk_ir[ik - 1] * wh + ta_ + (R)1.0e-7;
            ssatau = sa_ + taurs + (R)1.0e-8;
        }
        const R asysto = as_;
        R tautob = tausto, asytob = asysto / ssatau, ssatob = ssatau / tautob + (R)1.0e-8;
        ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
        R rrt, ttt, tdt, rst, tst, dum;
        so_deledd<R>(tautob, ssatob, asytob, cz, rrt, ttt, tdt);
        so_deledd<R>(tautob, ssatob, asytob, dsm, rst, tst, dum);
        LY(0, 1, k) = rrt; LY(1, 1, k) = ttt; LY(2, 1, k) = tdt; LY(3, 1, k) = rst; LY(4, 1, k) = tst;
        // the cloudy portion only matters in sky situations of non-zero weight, i.e. when the layer's group has cloud
        const R ccg = k < ict ? cc1 : (k < icb ? cc2 : cc3);
        if (ccg > 0) {
            const R tcb = A.cld[(((size_t)grp * 4 + 0) * K2 + k) * m + i], tcf = A.cld[(((size_t)grp * 4 + 1) * K2 + k) * m + i],
                    asyc = A.cld[(((size_t)grp * 4 + 2) * K2 + k) * m + i];
            const R ssac = uv ? (R)1 : A.cld[(((size_t)grp * 4 + 3) * K2 + k) * m + i];
            tautob = tausto + tcb;
            ssatob = (uv ? (ssatau + tcb) : (ssatau + ssac * tcb)) / tautob + (R)1.0e-8;
            ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
            asytob = (uv ? (asysto + asyc * tcb) : (asysto + asyc * ssac * tcb)) / (ssatob * tautob);
            const R tautof = tausto + tcf;
            R ssatof = (uv ? (ssatau + tcf) : (ssatau + ssac * tcf)) / tautof + (R)1.0e-8;
            ssatof = ssatof < (R)0.999999 ? ssatof : (R)0.999999;
            const R asytof = (uv ? (asysto + asyc * tcf) : (asysto + asyc * ssac * tcf)) / (ssatof * tautof);
            so_deledd<R>(tautob, ssatob, asytob, cz, rrt, ttt, tdt);
            so_deledd<R>(tautof, ssatof, asytof, dsm, rst, tst, dum);
            LY(0, 2, k) = rrt; LY(1, 2, k) = ttt; LY(2, 2, k) = tdt; LY(3, 2, k) = rst; LY(4, 2, k) = tst;
        }
    }

    // ---- CLDFLX (:689-872) ---------------------------------------------------------------------------------------------
    // The reference builds the composites piecewise (from the top through the high group per ih, on through the middle group per
    // (ih, im), ...) and re-walks the low / high groups for every sky situation.  Here every sky situation of non-zero weight
    // s = (ih, im, is) is ONE chain each way over all levels - the same operations on the same values in the same order (the same
    // results up to the compiler's choice of fused multiply-adds: 3e-16 / 4e-7 of the insolation against the first version) - and the
    // level is the outer loop:
    //   sweep U (surface -> top): the composites from the surface (rra, rxa) of all situations, parked per level.  Situations that
    //     share the portions below a level share the values: 2 variants in the low group, 4 in the middle, 8 in the high group are
    //     stored (planes 10 + 2 v, 11 + 2 v).
    //   sweep D (top -> surface): the composites from the top (tda, tta, rsa) stay in registers; at every level the fluxes of all
    //     situations are formed and summed in the reference's order (ih, im, is); the four flux planes are written once.
    // Per level and pass: 10 + 4..16 loads and 4..16 stores in U, 10 + 4..16 loads and 4 stores in D - the first version, which walked
    // situation by situation, read ~80 and wrote ~48 values per level (profiles/r01_v7_chou_pmc_traffic.md: 138 x the algorithmic bytes).
    // (Computing the layers inside sweep U, which would save their read-back there, is slower: 21.4 instead of 19.4 ms per 100 000
    // columns, with 214 VGPRs as well as - one portion at a time, scheduling barriers between the deledd calls - with 115.)
    const int nh = cc1 > 0 ? 2 : 1, nm = cc2 > 0 ? 2 : 1, ns = cc3 > 0 ? 2 : 1;       // portions of non-zero weight
    // situation s = 4 (ih-1) + 2 (im-1) + (is-1); the portion it uses in layer k; the variant of its surface-side composites at level k
    auto portion = [&](int s, int k) { return 1 + (k < ict ? (s >> 2) : (k < icb ? ((s >> 1) & 1) : (s & 1))); };
    auto variant = [&](int s, int k) { return k < ict ? s : (k < icb ? (s & 3) : (s & 1)); };
    uint32_t act = 0;
    for (int ih = 1; ih <= nh; ih++) for (int im = 1; im <= nm; im++) for (int is = 1; is <= ns; is++) act |= 1u << (4 * (ih - 1) + 2 * (im - 1) + (is - 1));
#define RRAV(k, v) P(10 + 2 * (v), k)
#define RXAV(k, v) P(11 + 2 * (v), k)
    // ---- sweep U ----
    {
        R rra[8], rxa[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { rra[s] = 0; rxa[s] = 0; }
#pragma unroll
        for (int s = 0; s < 8; s++)
            if ((act >> s) & 1u) { rra[s] = LY(0, 1 + (s & 1), np + 1); rxa[s] = LY(3, 1 + (s & 1), np + 1); }
#pragma unroll
        for (int v = 0; v < 2; v++)
            if ((act >> v) & 1u) { RRAV(np + 1, v) = rra[v]; RXAV(np + 1, v) = rxa[v]; }
        L5 n1 = ld5(1, np), n2 = ld5(2, np);
        for (int k = np; k >= 1; k--) {
            const L5 l1 = n1, l2 = n2;
            if (k - 1 >= 1) { n1 = ld5(1, k - 1); n2 = ld5(2, k - 1); }      // (before this level's stores: in-order memory counter)
#pragma unroll
            for (int s = 0; s < 8; s++) {
                if (!((act >> s) & 1u)) continue;
                const L5 &l = portion(s, k) == 1 ? l1 : l2;
                const R denm = l.ts / ((R)1. - l.rs * rxa[s]);
                const R nrra = l.rr + (l.td * rra[s] + (l.tt - l.td) * rxa[s]) * denm;
                rxa[s] = l.rs + l.ts * rxa[s] * denm; rra[s] = nrra;
            }
            const int nv = k < ict ? 8 : (k < icb ? 4 : 2);
#pragma unroll
            for (int v = 0; v < 8; v++)
                if (v < nv && ((act >> v) & 1u)) { RRAV(k, v) = rra[v]; RXAV(k, v) = rxa[v]; }
        }
    }
    // ---- sweep D ----
    R fsdir = 0, fsdif = 0, fall_sfc = 0;
    {
        R tda[8], tta[8], rsa[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { tda[s] = 0; tta[s] = 0; rsa[s] = 0; }
#pragma unroll
        for (int s = 0; s < 8; s++)
            if ((act >> s) & 1u) { const int j = 1 + (s >> 2); tda[s] = LY(2, j, 0); tta[s] = LY(1, j, 0); rsa[s] = LY(3, j, 0); }
        for (int k = 1; k <= np + 1; k++) {
            // everything this level needs is requested before anything is stored
            L5 l1{}, l2{};
            if (k <= np) { l1 = ld5(1, k); l2 = ld5(2, k); }
            const int nv = k < ict ? 8 : (k < icb ? 4 : 2);
            R bra[8], bxa[8];
#pragma unroll
            for (int v = 0; v < 8; v++) { bra[v] = 0; bxa[v] = 0; if (v < nv && ((act >> v) & 1u)) { bra[v] = RRAV(k, v); bxa[v] = RXAV(k, v); } }
            R fall = 0, fclr = 0, fupa = 0, fupc = 0;
#pragma unroll
            for (int s = 0; s < 8; s++) {      // the reference's order: ih outermost, is innermost
                if (!((act >> s) & 1u)) continue;
                const R ch = (s >> 2) ? cc1 : (R)1.0 - cc1;
                const R cm = ((s >> 1) & 1) ? ch * cc2 : ch * ((R)1.0 - cc2);
                const R ct = (s & 1) ? cm * cc3 : cm * ((R)1.0 - cc3);
                const int v = variant(s, k);
                const R rra = bra[v], rxa = bxa[v];
                const R denm = (R)1. / ((R)1. - rsa[s] * rxa);
                const R fdndir = tda[s];
                const R xx4 = tda[s] * rra, yy = tta[s] - tda[s];
                const R fdndif = (xx4 * rsa[s] + yy) * denm;
                const R fupdif = (xx4 + yy * rxa) * denm;
                const R flxdn = fdndir + fdndif - fupdif;
                // the first sky situation (all-clear portions) starts the weighted sums: 0 + x * ct, as the reference's zeroed arrays give
                if (s == 0) { fupc = fupdif; fclr = flxdn; fupa = (R)0 + fupdif * ct; fall = (R)0 + flxdn * ct; }
                else { fupa = fupa + fupdif * ct; fall = fall + flxdn * ct; }
                if (k == np + 1) { fsdir = fsdir + fdndir * ct; fsdif = fsdif + fdndif * ct; }
            }
            P(26, k) = fall; P(27, k) = fclr; P(28, k) = fupa; P(29, k) = fupc;
            if (k == np + 1) fall_sfc = fall;
            if (k <= np) {
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    if (!((act >> s) & 1u)) continue;
                    const L5 &l = portion(s, k) == 1 ? l1 : l2;
                    const R denm = l.ts / ((R)1. - rsa[s] * l.rs);
                    // (the reference writes the product in this term as tda * rsa * rr above the low group and tda * rr * rsa inside it)
                    const R ntta = k < icb ? tda[s] * l.tt + (tda[s] * rsa[s] * l.rr + tta[s] - tda[s]) * denm
                                           : tda[s] * l.tt + (tda[s] * l.rr * rsa[s] + tta[s] - tda[s]) * denm;
                    const R nrsa = l.rs + l.ts * rsa[s] * denm;
                    tda[s] = tda[s] * l.td; tta[s] = ntta; rsa[s] = nrsa;
                }
            }
        }
    }
    A.psum[((size_t)pass * 3 + 0) * m + i] = fsdir;
    A.psum[((size_t)pass * 3 + 1) * m + i] = fsdif;
    A.psum[((size_t)pass * 3 + 2) * m + i] = fall_sfc;      // all-sky net flux at the surface
#undef RRAV
#undef RXAV
#undef P
#undef LY
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_sum: per (column, level) -- flux integration over the 35 passes in pass order (Eq. 6.1)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_sum(SoradArgs<R> A, SoradOut<R> O)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y + 1;
    if (i >= A.m) return;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2;
    R s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 5
    for (int p = 0; p < SO_NPASS; p++) {
        const R hk = A.hk[p];
        const R *q = A.scr + (((size_t)p * SO_NPLANE + 26) * K2 + k) * m + i;
        s0 = s0 + q[0] * hk; s1 = s1 + q[(size_t)K2 * m] * hk; s2 = s2 + q[(size_t)2 * K2 * m] * hk; s3 = s3 + q[(size_t)3 * K2 * m] * hk;
    }
    const size_t o = (size_t)(k - 1) * ld + i;
    O.flx[o] = s0; O.flc[o] = s1; O.flxu[o] = s2; O.flcu[o] = s3;
    (void)np;
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_col: one block per column, lanes = (pass slot, LEVEL); the 35 spectral passes (5 UV/PAR + 3 x 10 NIR) three at a time with every
// per-level array of a pass in LDS - nothing of a pass touches HBM (the first version kept 34 planes of scratch per (column, pass)
// there: 1.6 MB of traffic per column).  Per pass:
//   phase A (lane = layer k): delta-Eddington R / T of the clear and the cloudy portion of the layer (4 x deledd in fp64, no vertical
//           dependence; sorad.F90:436-520, 996-1068); lanes 0 and np+1: the layer above the model top and the surface (:365-387, 914-936)
//   phase B (lane = one of 16 chains): CLDFLX's adding recurrences (:689-872) are first-order in the level index; each of the up to 8
//           sky situations (high, middle, low group clear | cloudy) has one chain from the top (direct / total transmittance and diffuse
//           reflectance of the layers above a level) and one from the surface (reflectances of the layers below): 16 lanes walk them
//           side by side.  (The reference shares a chain's first groups between situations; the values are the same.)
//   phase C (lane = level k): fluxes of every situation at the level, Eqs. (6.15)-(6.16), weighted by the situation's probability,
//           then the pass's share hk of the four level fluxes - accumulated over the passes in registers, in pass order (Eq. 6.1).
// LDS (reals): the column's inputs staged once [3 K2 + 3*8*np + 16 K2]; per pass slot: layer properties [2][5][K2], composites [8][5][K2].
// ---------------------------------------------------------------------------------------------------
// Q passes are worked on at a time (lane = (pass slot, level)): Q * (np + 2) lanes of a 256-thread block, 3 for 72 layers
__host__ __device__ constexpr int sorad_col_q(int np) { return 256 / (np + 2) < 3 ? (256 / (np + 2) < 1 ? 1 : 256 / (np + 2)) : 3; }
__host__ __device__ constexpr int sorad_col_threads(int np) { return (sorad_col_q(np) * (np + 2) + 63) / 64 * 64; }
template <typename R> __host__ __device__ constexpr size_t sorad_col_lds_reals(int np)
{
    return (size_t)19 * (np + 2) + (size_t)24 * np + (size_t)sorad_col_q(np) * 50 * (np + 2);
}

template <typename R>
__global__ void __launch_bounds__(256) k_sorad_col(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp, SoradOut<R> O)
{
    extern __shared__ __align__(16) unsigned char so_lds_raw[];
    R *const lds = reinterpret_cast<R *>(so_lds_raw);
    // consecutive blocks go to the 8 XCDs in turn: give each XCD a contiguous range of columns (neighbouring columns share the
    // sectors of the column-fastest input arrays, and so meet in one L2)
    const int per = (A.m + 7) / 8;
    const int i = (int)(blockIdx.x % 8u) * per + (int)(blockIdx.x / 8u);
    if (i >= A.m || (int)(blockIdx.x / 8u) >= per) return;
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2, ict = A.ict, icb = A.icb;
    const int t = (int)threadIdx.x, nt = (int)blockDim.x;
    const int Q = sorad_col_q(np);
    const int q = t / K2, k = t - q * K2;          // pass slot and level of this lane (q >= Q: idle lanes of the last wavefront)
    R *const s_lay = lds;                          // [3][K2]     dp, wh, oh
    R *const s_aer = s_lay + 3 * K2;               // [3][8][np]  taua, ssaa, asya
    R *const s_cld = s_aer + 24 * np;              // [4 groups][4][K2]
    R *const s_lyq = s_cld + 16 * K2;              // [Q][2 portions][5][K2]  rr, tt, td, rs, ts; after phase B: [Q][4][K2] level fluxes of the pass
    R *const s_cpq = s_lyq + (size_t)Q * 10 * K2;  // [Q][8 situations][5][K2] tda, tta, rsa (level = lower boundary of k), rra, rxa
    for (int e = t; e < 3 * K2; e += nt) s_lay[e] = A.lay[(size_t)e * m + i];
    for (int e = t; e < 8 * np; e += nt) {
        const size_t ja = (size_t)e * ld + i;
        s_aer[e] = A.taua[ja]; s_aer[8 * np + e] = A.ssaa[ja]; s_aer[16 * np + e] = A.asya[ja];
    }
    for (int e = t; e < 16 * K2; e += nt) s_cld[e] = A.cld[(size_t)e * m + i];
    const R cz = A.cosz[i], dsm = (R)0.602;
    const R cc1 = A.colv[0 * (size_t)m + i], cc2 = A.colv[1 * (size_t)m + i], cc3 = A.colv[2 * (size_t)m + i];
    const R wvtoa = A.colv[3 * (size_t)m + i], o3toa = A.colv[4 * (size_t)m + i];
    const int nh = cc1 > 0 ? 2 : 1, nm = cc2 > 0 ? 2 : 1, ns = cc3 > 0 ? 2 : 1;       // portions of non-zero weight
    __syncthreads();
    R *const s_ly = s_lyq + (size_t)(q < Q ? q : 0) * 10 * K2;
    R *const s_cp = s_cpq + (size_t)(q < Q ? q : 0) * 40 * K2;
#define LY(f, j, kk) s_ly[(((j) - 1) * 5 + (f)) * K2 + (kk)]              // f: 0 rr 1 tt 2 td 3 rs 4 ts;  j: 1 clear, 2 cloudy portion
#define CP(s_, f, kk) s_cp[((s_) * 5 + (f)) * K2 + (kk)]                  // f: 0 tda 1 tta 2 rsa 3 rra 4 rxa
    R flx = 0, flc = 0, flxu = 0, flcu = 0;      // lanes of slot 0: this level's Eq. (6.1) sums over the passes, in pass order
    const int nround = (SO_NPASS + Q - 1) / Q;
    for (int rnd = 0; rnd < nround; rnd++) {
        const int pass = rnd * Q + q;
        const bool live = q < Q && pass < SO_NPASS;
        const bool uv = pass < 5;
        const int ib = uv ? pass + 1 : (pass - 5) / 10 + 1, ik = uv ? 0 : (pass - 5) % 10 + 1;   // band in its region, k-value
        const int iv = uv ? ib : ib + 5;                                                         // aerosol band 1..8
        const int grp = uv ? 0 : ib;
        // ---- phase A: lane = (pass slot, layer) ---------------------------------------------------------------------------------
        if (live) {
            if (k == 0) {
                const R td0 = uv ? gr_exp<R>(-(wvtoa * T.wk_uv[ib - 1] + o3toa * T.zk_uv[ib - 1]) / cz) : gr_exp<R>(-wvtoa * T.xk_ir[ik - 1] / cz);
                for (int j = 1; j <= 2; j++) { LY(0, j, 0) = 0; LY(3, j, 0) = 0; LY(1, j, 0) = 1; LY(4, j, 0) = 1; LY(2, j, 0) = td0; }
            } else if (k == np + 1) {
                const R rb = uv ? A.rsuvbm[i] : A.rsirbm[i], rd = uv ? A.rsuvdf[i] : A.rsirdf[i];
                for (int j = 1; j <= 2; j++) { LY(0, j, np + 1) = rb; LY(3, j, np + 1) = rd; LY(2, j, np + 1) = 0; LY(1, j, np + 1) = 0; LY(4, j, np + 1) = 0; }
            } else {
                const R dp = s_lay[0 * K2 + k], wh = s_lay[1 * K2 + k], oh = s_lay[2 * K2 + k];
                const int ja = (iv - 1) * np + (k - 1);
                const R ta_ = s_aer[ja], sa_ = s_aer[8 * np + ja], as_ = s_aer[16 * np + ja];
                R taurs, tausto, ssatau;
                if (uv) {
                    taurs = T.ry_uv[ib - 1] * dp;
                    tausto = taurs + T.zk_uv[ib - 1] * oh + T.wk_uv[ib - 1] * wh + ta_ + (R)1.0e-7;
                    ssatau = sa_ + taurs;
                } else {
                    taurs = T.ry_ir[ib - 1] * dp;
                    tausto = taurs + T.xk_ir[ik - 1] * wh + ta_ + (R)1.0e-7;
                    ssatau = sa_ + taurs + (R)1.0e-8;
                }
                const R asysto = as_;
                R tautob = tausto, asytob = asysto / ssatau, ssatob = ssatau / tautob + (R)1.0e-8;
                ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
                R rrt, ttt, tdt, rst, tst, dum;
                so_deledd<R>(tautob, ssatob, asytob, cz, rrt, ttt, tdt);
                so_deledd<R>(tautob, ssatob, asytob, dsm, rst, tst, dum);
                LY(0, 1, k) = rrt; LY(1, 1, k) = ttt; LY(2, 1, k) = tdt; LY(3, 1, k) = rst; LY(4, 1, k) = tst;
                // the cloudy portion only matters in sky situations of non-zero weight, i.e. when the layer's group has cloud
                const R ccg = k < ict ? cc1 : (k < icb ? cc2 : cc3);
                if (ccg > 0) {
                    const R tcb = s_cld[(grp * 4 + 0) * K2 + k], tcf = s_cld[(grp * 4 + 1) * K2 + k], asyc = s_cld[(grp * 4 + 2) * K2 + k];
                    const R ssac = uv ? (R)1 : s_cld[(grp * 4 + 3) * K2 + k];
                    tautob = tausto + tcb;
                    ssatob = (uv ? (ssatau + tcb) : (ssatau + ssac * tcb)) / tautob + (R)1.0e-8;
                    ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
                    asytob = (uv ? (asysto + asyc * tcb) : (asysto + asyc * ssac * tcb)) / (ssatob * tautob);
                    const R tautof = tausto + tcf;
                    R ssatof = (uv ? (ssatau + tcf) : (ssatau + ssac * tcf)) / tautof + (R)1.0e-8;
                    ssatof = ssatof < (R)0.999999 ? ssatof : (R)0.999999;
                    const R asytof = (uv ? (asysto + asyc * tcf) : (asysto + asyc * ssac * tcf)) / (ssatof * tautof);
                    so_deledd<R>(tautob, ssatob, asytob, cz, rrt, ttt, tdt);
                    so_deledd<R>(tautof, ssatof, asytof, dsm, rst, tst, dum);
                    LY(0, 2, k) = rrt; LY(1, 2, k) = ttt; LY(2, 2, k) = tdt; LY(3, 2, k) = rst; LY(4, 2, k) = tst;
                }
            }
        }
        __syncthreads();
        // ---- phase B: lane = (pass slot, chain): situation s = ((ih-1)*2 + (im-1))*2 + (is-1), from the top (chains 0-7) / from the surface (8-15)
        if (t < 16 * Q && rnd * Q + t / 16 < SO_NPASS) {
            const int qb = t / 16, c = t & 15;
            R *const b_ly = s_lyq + (size_t)qb * 10 * K2;
            R *const b_cp = s_cpq + (size_t)qb * 40 * K2;
#define BLY(f, j, kk) b_ly[(((j) - 1) * 5 + (f)) * K2 + (kk)]
#define BCP(s_, f, kk) b_cp[((s_) * 5 + (f)) * K2 + (kk)]
            const int s_ = c & 7, ih = 1 + ((s_ >> 2) & 1), im = 1 + ((s_ >> 1) & 1), is = 1 + (s_ & 1);
            if (ih <= nh && im <= nm && is <= ns) {
                if (c < 8) {
                    R tda = BLY(2, ih, 0), tta = BLY(1, ih, 0), rsa = BLY(3, ih, 0);
                    BCP(s_, 0, 0) = tda; BCP(s_, 1, 0) = tta; BCP(s_, 2, 0) = rsa;
                    for (int kk = 1; kk <= np; kk++) {
                        const int j = kk < ict ? ih : (kk < icb ? im : is);
                        const R rr = BLY(0, j, kk), tt = BLY(1, j, kk), td = BLY(2, j, kk), rs = BLY(3, j, kk), ts = BLY(4, j, kk);
                        const R denm = ts / ((R)1. - rsa * rs);
                        // (the reference writes tda*rsa*rr in the high and middle groups and tda*rr*rsa in the low one)
                        const R x3 = kk < icb ? tda * rsa * rr : tda * rr * rsa;
                        const R ntta = tda * tt + (x3 + tta - tda) * denm;
                        const R nrsa = rs + ts * rsa * denm;
                        tda = tda * td; tta = ntta; rsa = nrsa;
                        BCP(s_, 0, kk) = tda; BCP(s_, 1, kk) = tta; BCP(s_, 2, kk) = rsa;
                    }
                } else {
                    R rra = BLY(0, is, np + 1), rxa = BLY(3, is, np + 1);
                    BCP(s_, 3, np + 1) = rra; BCP(s_, 4, np + 1) = rxa;
                    for (int kk = np; kk >= 0; kk--) {
                        const int j = kk >= icb ? is : (kk >= ict ? im : ih);
                        const R rr = BLY(0, j, kk), tt = BLY(1, j, kk), td = BLY(2, j, kk), rs = BLY(3, j, kk), ts = BLY(4, j, kk);
                        const R denm = ts / ((R)1. - rs * rxa);
                        const R nrra = rr + (td * rra + (tt - td) * rxa) * denm;
                        rxa = rs + ts * rxa * denm; rra = nrra;
                        BCP(s_, 3, kk) = rra; BCP(s_, 4, kk) = rxa;
                    }
                }
            }
#undef BLY
#undef BCP
        }
        __syncthreads();
        // ---- phase C: lane = (pass slot, level k = 1 .. np+1): integration over the sky situations of non-zero weight -----------
        if (live && k >= 1) {
            R fall = 0, fclr = 0, fupa = 0, fupc = 0, fsdir = 0, fsdif = 0;
            for (int ih = 1; ih <= nh; ih++) {
                const R ch = ih == 1 ? (R)1.0 - cc1 : cc1;
                for (int im = 1; im <= nm; im++) {
                    const R cm = im == 1 ? ch * ((R)1.0 - cc2) : ch * cc2;
                    for (int is = 1; is <= ns; is++) {
                        const R ct = is == 1 ? cm * ((R)1.0 - cc3) : cm * cc3;
                        const int s_ = ((ih - 1) * 2 + (im - 1)) * 2 + (is - 1);
                        const R tda = CP(s_, 0, k - 1), tta = CP(s_, 1, k - 1), rsa = CP(s_, 2, k - 1), rra = CP(s_, 3, k), rxa = CP(s_, 4, k);
                        const R denm = (R)1. / ((R)1. - rsa * rxa);      // Eqs. (6.15), (6.16)
                        const R fdndir = tda;
                        const R xx4 = tda * rra, yy = tta - tda;
                        const R fdndif = (xx4 * rsa + yy) * denm;
                        const R fupdif = (xx4 + yy * rxa) * denm;
                        const R flxdn = fdndir + fdndif - fupdif;
                        // the first sky situation (all-clear portions) starts the weighted sums: 0 + x * ct, as the reference's zeroed arrays give
                        if (s_ == 0) { fupc = fupdif; fclr = flxdn; fupa = (R)0 + fupdif * ct; fall = (R)0 + flxdn * ct; }
                        else { fupa = fupa + fupdif * ct; fall = fall + flxdn * ct; }
                        fsdir = fsdir + fdndir * ct;
                        fsdif = fsdif + fdndif * ct;
                    }
                }
            }
            // the pass's level fluxes, for the lanes of slot 0 to add up in pass order (the layer properties are not needed any more)
            s_ly[0 * K2 + k] = fall; s_ly[1 * K2 + k] = fclr; s_ly[2 * K2 + k] = fupa; s_ly[3 * K2 + k] = fupc;
            if (k == np + 1) {      // surface: what k_sorad_reduce needs of the pass
                A.psum[((size_t)pass * 3 + 0) * m + i] = fsdir;
                A.psum[((size_t)pass * 3 + 1) * m + i] = fsdif;
                A.psum[((size_t)pass * 3 + 2) * m + i] = fall;
            }
        }
        __syncthreads();
        if (q == 0 && k >= 1) {
            for (int qq = 0; qq < Q && rnd * Q + qq < SO_NPASS; qq++) {
                const R hk = A.hk[rnd * Q + qq];
                const R *const f = s_lyq + (size_t)qq * 10 * K2;
                flx = flx + f[0 * K2 + k] * hk; flc = flc + f[1 * K2 + k] * hk; flxu = flxu + f[2 * K2 + k] * hk; flcu = flcu + f[3 * K2 + k] * hk;
            }
        }
        __syncthreads();       // the next round rewrites the layer properties and the composites
    }
#undef LY
#undef CP
    if (q == 0 && k >= 1) {
        const size_t o = (size_t)(k - 1) * ld + i;
        O.flx[o] = flx; O.flc[o] = flc; O.flxu[o] = flxu; O.flcu[o] = flcu;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_reduce: per column -- flux integration over the passes (Eq. 6.1), O2 / CO2 reductions (:1425-1552), surface rescaling
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_reduce(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp, SoradOut<R> O)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.m) return;
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2;
#define OUT2(a, k) a[(size_t)((k) - 1) * ld + i]
    // (the level fluxes were summed over the passes by k_sorad_col)
    // surface band fluxes and direct / diffuse partition
    R fdiruv = 0, fdifuv = 0, fdirpar = 0, fdifpar = 0, fdirir = 0, fdifir = 0, band[8], drb[8], dfb[8];
    for (int b = 0; b < 8; b++) { band[b] = 0; drb[b] = 0; dfb[b] = 0; }
    for (int p = 0; p < SO_NPASS; p++) {
        const R hk = A.hk[p], fs = A.psum[((size_t)p * 3 + 0) * m + i], fd = A.psum[((size_t)p * 3 + 1) * m + i];
        const int b = p < 5 ? p : 5 + (p - 5) / 10;
        band[b] = band[b] + A.psum[((size_t)p * 3 + 2) * m + i] * hk; drb[b] = drb[b] + fs * hk; dfb[b] = dfb[b] + fd * hk;
        if (p < 4) { fdiruv = fdiruv + fs * hk; fdifuv = fdifuv + fd * hk; }
        else if (p == 4) { fdirpar = fs * hk; fdifpar = fd * hk; }
        else { fdirir = fdirir + fs * hk; fdifir = fdifir + fd * hk; }
    }
    // flux reductions: running column amounts, two table look-ups per level
    const R snt = (R)1.0 / A.cosz[i];
    const R scal0 = A.colv[5 * (size_t)m + i];
    const int ntop = (int)A.colv[6 * (size_t)m + i];
    const R cnt = (R)165.22 * snt;
    R so2o = scal0 * cnt, so2c = ((R)789. * A.co2) * scal0;
    const R flx_top = OUT2(O.flx, ntop);
    R dftop = 0, dfsfc = 0;
    for (int k = 1; k <= np + 1; k++) {
        if (k > 1) { const R sc = A.lay[((size_t)3 * K2 + (k - 1)) * m + i]; so2o = so2o + sc * cnt; so2c = so2c + ((R)789. * A.co2) * sc; }
        R df = (R)0.0633 * ((R)1. - gr_exp<R>((R)-0.000155 * sqrt(so2o)));
        {
            const R u1 = (R)-3.0, du = (R)0.15, w1 = (R)-4.0, dw = (R)0.15;
            const R x0 = u1 + (R)43 * du, y0 = w1 + (R)37 * dw, x1 = u1 - (R)0.5 * du, y1 = w1 - (R)0.5 * dw;
            R ulog = gr_log10<R>(so2c * snt); ulog = ulog < x0 ? ulog : x0;
            R wlog = gr_log10<R>(A.swh[(size_t)k * m + i] * snt); wlog = wlog < y0 ? wlog : y0;
            int ic = (int)((ulog - x1) / du + (R)1.), iw = (int)((wlog - y1) / dw + (R)1.);
            ic = ic > 2 ? ic : 2; iw = iw > 2 ? iw : 2; ic = ic < 43 ? ic : 43; iw = iw < 37 ? iw : 37;
            const R dc = ulog - (R)(ic - 2) * du - u1, dd = wlog - (R)(iw - 2) * dw - w1;
#define CAH(a, b) T.cah[((b) - 1) * 43 + ((a) - 1)]
            const R x2 = CAH(ic - 1, iw - 1) + (CAH(ic - 1, iw) - CAH(ic - 1, iw - 1)) / dw * dd;
            R y2 = x2 + (CAH(ic, iw - 1) - CAH(ic - 1, iw - 1)) / du * dc;
#undef CAH
            y2 = y2 > 0 ? y2 : (R)0;
            df = df + (R)1.5 * y2;
        }
        {
            const R u1 = (R)0.000250, du = (R)0.000050, w1 = (R)-2.0, dw = (R)0.05;
            const R x0 = u1 + (R)62 * du, y0 = w1 + (R)101 * dw, x1 = u1 - (R)0.5 * du, y1 = w1 - (R)0.5 * dw;
            R ulog = A.co2 * snt; ulog = ulog < x0 ? ulog : x0;
            R wlog = gr_log10<R>(A.pl[(size_t)(k - 1) * ld + i]); wlog = wlog < y0 ? wlog : y0;
            int ic = (int)((ulog - x1) / du + (R)1.), iw = (int)((wlog - y1) / dw + (R)1.);
            ic = ic > 2 ? ic : 2; iw = iw > 2 ? iw : 2; ic = ic < 62 ? ic : 62; iw = iw < 101 ? iw : 101;
            const R dc = ulog - (R)(ic - 2) * du - u1, dd = wlog - (R)(iw - 2) * dw - w1;
#define COA(a, b) T.coa[((b) - 1) * 62 + ((a) - 1)]
            const R x2 = COA(ic - 1, iw - 1) + (COA(ic - 1, iw) - COA(ic - 1, iw - 1)) / dw * dd;
            R y2 = x2 + (COA(ic, iw - 1) - COA(ic - 1, iw - 1)) / du * dc;
#undef COA
            y2 = y2 > 0 ? y2 : (R)0;
            df = df + (R)1.5 * y2;
        }
        // below the cloud top the reduction scales with the all-sky flux, Eq. (6.18)
        if (k == ntop) dftop = df;
        R fl = OUT2(O.flx, k);
        if (k > ntop) { const R xx4 = fl / flx_top; df = dftop + xx4 * (df - dftop); }
        df = df < fl - (R)1.0e-8 ? df : fl - (R)1.0e-8;
        OUT2(O.flx, k) = fl - df;
        OUT2(O.flc, k) = OUT2(O.flc, k) - df;
        if (k == np + 1) dfsfc = df;
    }
    R xx4 = OUT2(O.flx, np + 1) + dfsfc;
    const R eps = sizeof(R) == 4 ? (R)1.1920929e-07 : (R)2.220446049250313e-16;
    if (fabs(xx4) > eps) { xx4 = (R)1.0 - dfsfc / xx4; xx4 = xx4 < 1 ? xx4 : (R)1; xx4 = xx4 > 0 ? xx4 : (R)0; }
    else xx4 = 0;
    O.fdirir[i] = xx4 * fdirir; O.fdifir[i] = xx4 * fdifir; O.fdiruv[i] = xx4 * fdiruv; O.fdifuv[i] = xx4 * fdifuv;
    O.fdirpar[i] = xx4 * fdirpar; O.fdifpar[i] = xx4 * fdifpar;
    for (int b = 0; b < 8; b++) {
        O.flx_sfc_band[(size_t)b * ld + i] = xx4 * band[b];
        if (A.do_drfband) { O.drband[(size_t)b * ld + i] = xx4 * drb[b]; O.dfband[(size_t)b * ld + i] = xx4 * dfb[b]; }
    }
#undef OUT2
}

}  // namespace geosrad
