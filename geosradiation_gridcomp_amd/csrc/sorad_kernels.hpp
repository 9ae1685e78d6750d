// sorad_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the Chou-Suarez shortwave scheme `sorad`.
//
// Reference behaviour: GEOSsolar_GridComp/sorad.F90:43-1588 (SOLUV / SOLIR / CLDFLX inlined, O2 + CO2 flux reductions),
// deledd :1592-1706; cloud optics GEOS_RadiationShared/getvistau.code, getnirtau.code.  Non-OVERCAST build.
//
// sorad is 35 independent spectral passes (5 UV/PAR bands + 3 NIR bands x 10 k-values), each a set of delta-Eddington layers (no
// vertical dependence, fp64, the expensive part) + first-order vertical recurrences (adding over up to 8 sky situations):
//   k_sorad_class  : per column   - which of the three cloud groups (high / middle / low) hold cloud: the column's CLASS (0..7)
//   k_partition8   : one block    - stable counting sort of the columns by class -> perm (position -> column), class offsets.  Every later
//                    kernel works on POSITIONS: workspace arrays are indexed by the position, API arrays by perm[position], so 256-position
//                    blocks are class-homogeneous (at most 7 mixed blocks) whatever the spatial distribution of the clouds
//   k_sorad_gather : per (position, aerosol row) - the three aerosol arrays copied into position order (each is read by up to 10 passes)
//   k_sorad_prep   : per position - scaled absorber amounts, cloud-group covers, cloud top
//   k_sorad_cloud  : per (position, layer) - getvistau / getnirtau for the four optics groups (UV/PAR + 3 NIR bands)
//   k_sorad_pass<CLS> : per (position, pass), lane = column, one instantiation per class - deledd of the clear / cloudy portion of every
//                    layer, CLDFLX over exactly the class's sky situations; the per-level arrays of the pass in HBM scratch planes
//                    [array][level][position] (coalesced); k_sorad_sum adds the passes up (default path)
//   k_sorad_col    : per column, lanes = (pass, level), the 35 passes on chip: no scratch (GEOSRAD_SORAD_PATH=col)
//   k_sorad_reduce : per column   - weighted sum over the passes (hk_uv, hk_ir), flux reductions, surface rescaling
// Sky situations of zero weight (ct = 0: a cloud group without cloud) are skipped: their contribution is `+ x * 0`.
#pragma once
#include <type_traits>
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
    R *aer;          // [3][SO_NGATHER][np][m]: taua, ssaa, asya of the NIR bands in position order (k_sorad_gather)
    uint8_t *cls;    // [m] class of the column: 4 (high group has cloud) + 2 (middle) + 1 (low); in column order
    int32_t *perm;   // [m] position -> column (stable within a class)
    int32_t *cls_off;// [9] first position of every class, cls_off[8] = m
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
// k_sorad_class: which cloud groups of the column hold cloud (cc1, cc2, cc3 > 0, :416-431): the sky situations of non-zero weight are the
// 2^(groups with cloud) combinations of their clear / cloudy portions.  `one_class`: every column gets class 7 (identity permutation: the
// on-chip path k_sorad_col takes the situations per column at run time)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_class(SoradArgs<R> A, int one_class)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.m) return;
    int c = 7;
    if (!one_class) {
        bool h = false, mid = false, low = false;
        for (int k = 1; k <= A.np; k++) {
            const bool cl = A.fcld[(size_t)(k - 1) * A.ld + i] > 0;
            if (k < A.ict) h = h || cl; else if (k < A.icb) mid = mid || cl; else low = low || cl;
        }
        c = (h ? 4 : 0) + (mid ? 2 : 0) + (low ? 1 : 0);
    }
    A.cls[i] = (uint8_t)c;
}

// stable counting sort of the columns by class (one 1024-thread block; cf. k_partition of the RRTMG solvers)
static __global__ void __launch_bounds__(1024) k_partition8(int ncol, const uint8_t *__restrict__ cls, int32_t *__restrict__ perm,
                                                     int32_t *__restrict__ off)
{
    __shared__ int cnt[8][1024];
    __shared__ int first[9];
    const int t = threadIdx.x;
    const int chunk = (ncol + 1023) / 1024;
    const int b = t * chunk < ncol ? t * chunk : ncol, e = (b + chunk < ncol) ? b + chunk : ncol;
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = b; i < e; i++) {
        const int q = cls[i] & 7;
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] += q == k;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) cnt[k][t] = c[k];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // inclusive Hillis-Steele scan of the eight rows
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = t >= d ? cnt[k][t - d] : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) cnt[k][t] += v[k];
        __syncthreads();
    }
    if (t == 0) {
        first[0] = 0;
        for (int k = 0; k < 8; k++) first[k + 1] = first[k] + cnt[k][1023];
        for (int k = 0; k <= 8; k++) off[k] = first[k];
    }
    __syncthreads();
    int at[8];
#pragma unroll
    for (int k = 0; k < 8; k++) at[k] = first[k] + cnt[k][t] - c[k];
    for (int i = b; i < e; i++) {
        const int q = cls[i] & 7;
        int dst = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) if (q == k) { dst = at[k]; at[k]++; }
        perm[dst] = i;
    }
}

// the aerosol arrays (m, np, nb >= 8) of the last SO_NGATHER of the 8 bands in position order: row r = (array 0..2, band, layer)
// -> aer[r][position]
#ifndef SO_NGATHER
#define SO_NGATHER 3          // the NIR bands 6..8
#endif
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_gather(SoradArgs<R> A)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;                                  // 0 .. 3 * SO_NGATHER * np - 1
    if (p >= A.m) return;
    const int rows = SO_NGATHER * A.np, a = r / rows, q = r - a * rows;
    const R *src = a == 0 ? A.taua : (a == 1 ? A.ssaa : A.asya);
    A.aer[(size_t)r * A.m + p] = src[((size_t)(8 - SO_NGATHER) * A.np + q) * A.ld + A.perm[p]];
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_prep (:271-357, 416-431, 1534-1541); workspace by position, API arrays by column
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_prep(SoradArgs<R> A)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= A.m) return;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2;
    const int i = A.perm[pos];
#define AP(a, k) a[(size_t)((k) - 1) * ld + i]
#define LAY(f, k) A.lay[((size_t)(f) * K2 + (k)) * m + pos]
    const R xtoa = AP(A.pl, 1) > (R)1.e-3 ? AP(A.pl, 1) : (R)1.e-3;
    const R scal0 = xtoa * gr_pow<R>((R)0.5 * xtoa / (R)300., (R).8);
    const R o3toa = (R)1.02 * AP(A.oa, 1) * xtoa * (R)466.7 + (R)1.0e-8;
    const R wvtoa = (R)1.02 * AP(A.wa, 1) * scal0 * ((R)1.0 + (R)0.00135 * (AP(A.ta, 1) - (R)240.)) + (R)1.0e-9;
    R sw = wvtoa, cc1 = 0, cc2 = 0, cc3 = 0;
    int ntop = np + 1; bool found = false;
    A.swh[(size_t)1 * m + pos] = sw;
    for (int k = 1; k <= np; k++) {
        const R dp = AP(A.pl, k + 1) - AP(A.pl, k);
        const R pa = (R)0.5 * (AP(A.pl, k) + AP(A.pl, k + 1));
        const R scal = dp * gr_pow<R>(pa / (R)300., (R).8);
        const R wh = (R)1.02 * AP(A.wa, k) * scal * ((R)1. + (R)0.00135 * (AP(A.ta, k) - (R)240.)) + (R)1.e-9;
        sw = sw + wh;
        A.swh[(size_t)(k + 1) * m + pos] = sw;
        LAY(0, k) = dp; LAY(1, k) = wh; LAY(2, k) = (R)1.02 * AP(A.oa, k) * dp * (R)466.7 + (R)1.e-8; LAY(3, k) = scal;
        const R fc = AP(A.fcld, k);
        if (k < A.ict) cc1 = cc1 > fc ? cc1 : fc; else if (k < A.icb) cc2 = cc2 > fc ? cc2 : fc; else cc3 = cc3 > fc ? cc3 : fc;
        if (fc > (R)0.02 && !found) { found = true; ntop = k; }
    }
    R *cv = A.colv + pos;
    cv[0 * (size_t)m] = cc1; cv[1 * (size_t)m] = cc2; cv[2 * (size_t)m] = cc3; cv[3 * (size_t)m] = wvtoa; cv[4 * (size_t)m] = o3toa;
    cv[5 * (size_t)m] = scal0; cv[6 * (size_t)m] = (R)ntop;
#undef AP
#undef LAY
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_cloud: getvistau (group 0) / getnirtau (groups 1..3 = NIR bands), one thread per (position, layer): the layer's condensate and
// radii are read once (through the permutation) for the four optics groups; layers of cloud groups the column's class lacks are skipped
// (no pass reads their planes)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_cloud(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp)
{
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = (int)blockIdx.y + 1;
    if (pos >= A.m) return;
    if (pos < A.cls_off[1]) return;          // class 0: no cloud group holds cloud, no pass reads the cloud planes
    const int i = A.perm[pos];
    {
        const int c = A.cls[i];
        if (!(k < A.ict ? (c & 4) : (k < A.icb ? (c & 2) : (c & 1)))) return;
    }
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2, ict = A.ict, icb = A.icb;
    const R dm = (R)0.1, dt = (R)0.30103, da = (R)0.1, t1 = (R)-0.9031;
    const R cosz = A.cosz[i];
    const R cc[4] = {0, A.colv[0 * (size_t)m + pos], A.colv[1 * (size_t)m + pos], A.colv[2 * (size_t)m + pos]};
#define CLD(f, k) A.cld[(((size_t)ib * 4 + (f)) * K2 + (k)) * m + pos]
#define CAIB(a, b, c) T.caib[(((c) - 1) * 9 + ((b) - 1)) * 11 + ((a) - 1)]
#define CAIF(a, b) T.caif[((b) - 1) * 9 + ((a) - 1)]
#define N2(tab, j) tab[((j) - 1) * 3 + (ib - 1)]
    {
        const R dp_pa = A.lay[((size_t)0 * K2 + k) * m + pos] * (R)100.;
        const R wp = (dp_pa * (R)1.0e3) / (R)9.80665;                 // MAPL_GRAV
        const R r1 = A.reff[((size_t)0 * np + (k - 1)) * ld + i], r2 = A.reff[((size_t)1 * np + (k - 1)) * ld + i],
                r4 = A.reff[((size_t)3 * np + (k - 1)) * ld + i];
        const R h1 = A.cwc[((size_t)0 * np + (k - 1)) * ld + i], h2 = A.cwc[((size_t)1 * np + (k - 1)) * ld + i],
                h3 = A.cwc[((size_t)2 * np + (k - 1)) * ld + i], h4 = A.cwc[((size_t)3 * np + (k - 1)) * ld + i];
        const R fc = A.fcld[(size_t)(k - 1) * ld + i];
        const R rs = r4 < (R)112.0 ? r4 : (R)112.0;
        for (int ib = 0; ib < SO_NGRP; ib++) {
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
    }
#undef CLD
#undef CAIB
#undef CAIF
#undef N2
}

// ---------------------------------------------------------------------------------------------------
// k_sorad_pass<R, CLS>: one thread per (position, spectral pass); every lane of a block belongs to class CLS (which cloud groups hold
// cloud: 4 high + 2 middle + 1 low), so the sky situations of non-zero weight - the NS = 2^(groups with cloud) combinations of the groups'
// clear / cloudy portions - and everything indexed by them is compile-time: register arrays of exactly NS entries, no predication.
// Scratch planes of the pass (index q, level k; [q][k][position]):
//   0..9  : rr, tt, td, rs, ts of the clear (q = 2 f) and cloudy (q = 2 f + 1) portion of layer k (cloudy: only in groups with cloud)
//   10..25: rra, rxa composites from the surface of the distinct VARIANTS at the level (q = 10 + 2 v, 11 + 2 v; see CLDFLX below)
//   26..29: fall, fclr, fupa, fupc (class 0: fall, fupa only - one sky situation of weight 1, the clear-sky fluxes are the same numbers)
// ---------------------------------------------------------------------------------------------------
template <typename R> struct SoL5 { R rr, tt, td, rs, ts; };

// the divisions of the adding equations and of the layers' mixed optical properties (~20 per level and pass in a class-7 column): hardware
// reciprocal (1 ulp) in the fp32 instantiation, v_rcp_f64 + Newton steps (gr_div64) in the fp64 one, instead of the correctly rounded
// expansions (10 / 12 instructions each).  deledd keeps the reference's fp64 arithmetic as it is.
#ifndef SO_EXACT_DIV
GR_DEV float so_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
GR_DEV double so_div(double a, double b) { return gr_div64(a, b); }
#else
template <typename R> GR_DEV R so_div(R a, R b) { return a / b; }
#endif

// one adding step from the surface: layer l above the composite (rra, rxa) (sorad.F90:751-768, 789-806, 825-842)
template <typename R> GR_DEV void so_add_up(const SoL5<R> &l, R &rra, R &rxa)
{
    const R denm = so_div(l.ts, (R)1. - l.rs * rxa);
    const R nrra = l.rr + (l.td * rra + (l.tt - l.td) * rxa) * denm;
    rxa = l.rs + l.ts * rxa * denm; rra = nrra;
}
// one adding step from the top: layer l below the composite (tda, tta, rsa) (:700-745); LOW: the reference writes the product of this
// term as tda * rsa * rr above the low group and tda * rr * rsa inside it
template <typename R, bool LOW> GR_DEV void so_add_down(const SoL5<R> &l, R &tda, R &tta, R &rsa)
{
    const R denm = so_div(l.ts, (R)1. - rsa * l.rs);
    const R ntta = LOW ? tda * l.tt + (tda * l.rr * rsa + tta - tda) * denm : tda * l.tt + (tda * rsa * l.rr + tta - tda) * denm;
    const R nrsa = l.rs + l.ts * rsa * denm;
    tda = tda * l.td; tta = ntta; rsa = nrsa;
}

template <typename R, int CLS>
__global__ void __launch_bounds__(256) k_sorad_pass(SoradArgs<R> A, const SoradDev<R> *__restrict__ Tp)
{
    constexpr bool GH = (CLS & 4) != 0, GM = (CLS & 2) != 0, GL = (CLS & 1) != 0;       // groups with cloud
    constexpr int NH = GH ? 2 : 1, NM = GM ? 2 : 1, NL = GL ? 2 : 1, NS = NH * NM * NL;
    int bstart, pass;
    if (!band_block(A.m, SO_NPASS, bstart, pass)) return;
    const int lo = A.cls_off[CLS], hi = A.cls_off[CLS + 1];
    if (bstart >= hi || bstart + (int)blockDim.x <= lo) return;                         // no position of this class in the block
    const int i = bstart + (int)threadIdx.x;                                            // position
    if (i < lo || i >= hi) return;
    const SoradDev<R> &T = *Tp;
    const int np = A.np, m = A.m, K2 = np + 2, ict = A.ict, icb = A.icb;
    const bool uv = pass < 5;
    const int ib = uv ? pass + 1 : (pass - 5) / 10 + 1, ik = uv ? 0 : (pass - 5) % 10 + 1;   // band in its region, k-value
    const int iv = uv ? ib : ib + 5;                                                         // aerosol band 1..8
    const int grp = uv ? 0 : ib;
    const int col = A.perm[i];
    const R cz = A.cosz[col], dsm = (R)0.602;
    const R cc1 = A.colv[0 * (size_t)m + i], cc2 = A.colv[1 * (size_t)m + i], cc3 = A.colv[2 * (size_t)m + i];
    const R wvtoa = A.colv[3 * (size_t)m + i], o3toa = A.colv[4 * (size_t)m + i];
    R *S = A.scr + (size_t)pass * SO_NPLANE * K2 * m + i;
    using L5 = SoL5<R>;
#define P(q, k) S[((size_t)(q) * K2 + (k)) * m]
#define LY(f, j, k) P(2 * (f) + (j) - 1, k)              // f: 0 rr 1 tt 2 td 3 rs 4 ts;  j: 1 clear, 2 cloudy
    // The sweeps below are first-order recurrences over the levels: each step loads a layer's five properties and stores the new
    // composites.  gfx9 tracks loads and stores with one in-order counter, so a load issued after a store cannot be consumed before
    // that store is acknowledged: every sweep therefore requests the NEXT level's properties before it stores the current results.
    auto ld5 = [&](int j, int k) { L5 l; l.rr = LY(0, j, k); l.tt = LY(1, j, k); l.td = LY(2, j, k); l.rs = LY(3, j, k); l.ts = LY(4, j, k); return l; };
    // boundary "layers" (:365-387, 914-936): the surface (np + 1: rr = rb, rs = rd, no transmission) and the layer above the model top
    // (0: no reflection, tt = ts = 1, td = td0) are the same for both portions and stay in registers
    const R rb = uv ? A.rsuvbm[col] : A.rsirbm[col], rd = uv ? A.rsuvdf[col] : A.rsirdf[col];
    const R td0 = uv ? gr_exp<R>(-(wvtoa * T.wk_uv[ib - 1] + o3toa * T.zk_uv[ib - 1]) / cz) : gr_exp<R>(-wvtoa * T.xk_ir[ik - 1] / cz);
    // the pass's absorption / Rayleigh coefficients, read once (inside the layer loop they were scalar loads from *Tp per layer)
    const R k_ry = uv ? T.ry_uv[ib - 1] : T.ry_ir[ib - 1], k_zk = uv ? T.zk_uv[ib - 1] : (R)0, k_wk = uv ? T.wk_uv[ib - 1] : T.xk_ir[ik - 1];
    // ---- layers: clear and cloudy portion (:436-520, 996-1068) ---------------------------------------------------------
    struct In6 { R dp, wh, oh, ta, sa, as; };
    // aerosols: the three NIR bands are read by ten passes each - from the position-ordered copies (k_sorad_gather); the five UV / PAR
    // bands by one pass each - from the caller's arrays, through the permutation
    const size_t aer_rows = (size_t)SO_NGATHER * np;
    const bool from_copy = SO_NGATHER == 8 || !uv;
    const R *a0 = from_copy ? A.aer + ((size_t)(iv - 1 - (8 - SO_NGATHER)) * np) * m + i : A.taua + ((size_t)(iv - 1) * np) * A.ld + col;
    const R *a1 = from_copy ? a0 + aer_rows * m : A.ssaa + ((size_t)(iv - 1) * np) * A.ld + col;
    const R *a2 = from_copy ? a1 + aer_rows * m : A.asya + ((size_t)(iv - 1) * np) * A.ld + col;
    const size_t astr = from_copy ? (size_t)m : (size_t)A.ld;
    auto ldin = [&](int k) {
        In6 v;
        v.dp = A.lay[((size_t)0 * K2 + k) * m + i]; v.wh = A.lay[((size_t)1 * K2 + k) * m + i]; v.oh = A.lay[((size_t)2 * K2 + k) * m + i];
        const size_t ja = (size_t)(k - 1) * astr;
        v.ta = a0[ja]; v.sa = a1[ja]; v.as = a2[ja];
        return v;
    };
    struct Cl4 { R tcb, tcf, asyc, ssac; };
    auto ldcl = [&](int k) {
        Cl4 c;
        c.tcb = A.cld[(((size_t)grp * 4 + 0) * K2 + k) * m + i]; c.tcf = A.cld[(((size_t)grp * 4 + 1) * K2 + k) * m + i];
        c.asyc = A.cld[(((size_t)grp * 4 + 2) * K2 + k) * m + i];
        c.ssac = uv ? (R)1 : A.cld[(((size_t)grp * 4 + 3) * K2 + k) * m + i];
        return c;
    };
    // the cloudy portion only matters in sky situations of non-zero weight, i.e. when the layer's group has cloud (wave-uniform)
    auto gcld = [&](int k) { return k < ict ? GH : (k < icb ? GM : GL); };
    // delta-Eddington properties of layer k: clear portion l1, cloudy portion l2 (only when its group has cloud)
    auto layer = [&](const In6 &cin, const Cl4 &cl, bool cloudy, L5 &l1, L5 &l2) {
        const R dp = cin.dp, wh = cin.wh, oh = cin.oh, ta_ = cin.ta, sa_ = cin.sa, as_ = cin.as;
        R taurs, tausto, ssatau;
        if (uv) {
            taurs = k_ry * dp;
            tausto = taurs + k_zk * oh + k_wk * wh + ta_ + (R)1.0e-7;
            ssatau = sa_ + taurs;
        } else {
            taurs = k_ry * dp;
            tausto = taurs + k_wk * wh + ta_ + (R)1.0e-7;
            ssatau = sa_ + taurs + (R)1.0e-8;
        }
        const R asysto = as_;
        R tautob = tausto, asytob = so_div(asysto, ssatau), ssatob = so_div(ssatau, tautob) + (R)1.0e-8;
        ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
        R dum;
        so_deledd<R>(tautob, ssatob, asytob, cz, l1.rr, l1.tt, l1.td);
        so_deledd<R>(tautob, ssatob, asytob, dsm, l1.rs, l1.ts, dum);
        if (cloudy) {
            const R tcb = cl.tcb, tcf = cl.tcf, asyc = cl.asyc, ssac = cl.ssac;
            tautob = tausto + tcb;
            ssatob = so_div(uv ? (ssatau + tcb) : (ssatau + ssac * tcb), tautob) + (R)1.0e-8;
            ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
            asytob = so_div(uv ? (asysto + asyc * tcb) : (asysto + asyc * ssac * tcb), ssatob * tautob);
            const R tautof = tausto + tcf;
            R ssatof = so_div(uv ? (ssatau + tcf) : (ssatau + ssac * tcf), tautof) + (R)1.0e-8;
            ssatof = ssatof < (R)0.999999 ? ssatof : (R)0.999999;
            const R asytof = so_div(uv ? (asysto + asyc * tcf) : (asysto + asyc * ssac * tcf), ssatof * tautof);
            so_deledd<R>(tautob, ssatob, asytob, cz, l2.rr, l2.tt, l2.td);
            so_deledd<R>(tautof, ssatof, asytof, dsm, l2.rs, l2.ts, dum);
        }
    };
    auto st5 = [&](int j, int k, const L5 &l) { LY(0, j, k) = l.rr; LY(1, j, k) = l.tt; LY(2, j, k) = l.td; LY(3, j, k) = l.rs; LY(4, j, k) = l.ts; };

    // ---- CLDFLX (:689-872) ---------------------------------------------------------------------------------------------
    // A sky situation (ih, im, is) takes the clear (1) or cloudy (2) portion of every layer of the high / middle / low group.  The
    // reference builds the composites piecewise per group and combines them per situation; here each DISTINCT composite is one chain:
    //   sweep U (surface -> top): composites from the surface (rra, rxa).  Inside the low group they depend on `is` only (NL chains), in
    //     the middle group on (im, is) (NM NL chains), in the high group on all three (NS chains); a chain forks where the next group
    //     begins.  Parked per level and variant v (planes 10 + 2 v, 11 + 2 v).
    //   sweep D (top -> surface): composites from the top (tda, tta, rsa) in registers: NH chains in the high group, NH NM in the middle,
    //     NS in the low group; at every level the fluxes of all NS situations are formed from (top chain, parked variant) and summed in
    //     the reference's order (ih outermost, is innermost); the flux planes are written once.
    // The operations on a chain are the reference's, in its order (the same results up to the compiler's choice of fused multiply-adds).
    // Index of situation (ihx, imx, isx), each 0 = clear / 1 = cloudy: t = (ihx NM + imx) NL + isx.
#define RRAV(k, v) P(10 + 2 * (v), k)
#define RXAV(k, v) P(11 + 2 * (v), k)
    // ---- sweep U ----
    {
        R rra[NS], rxa[NS];
#pragma unroll
        for (int v = 0; v < NL; v++) { rra[v] = rb; rxa[v] = rd; RRAV(np + 1, v) = rb; RXAV(np + 1, v) = rd; }
        // the layers' properties are formed here, on the way up (each layer's inputs requested a level ahead), and parked for sweep D
        In6 nin = ldin(np);
        Cl4 ncl{};
        if (gcld(np)) ncl = ldcl(np);
        L5 l1, l2;
#define SO_NEXT_LAYER(k)                                                                                 \
        {                                                                                                \
            const In6 cin = nin; const Cl4 ccl = ncl;                                                    \
            if ((k) - 1 >= 1) { nin = ldin((k) - 1); if (gcld((k) - 1)) ncl = ldcl((k) - 1); }           \
            layer(cin, ccl, gcld(k), l1, l2);                                                            \
            st5(1, k, l1);                                                                               \
            if (gcld(k)) st5(2, k, l2);                                                                  \
        }
        // low group: layers np .. icb, NL chains
        int k = np;
        for (; k >= icb; k--) {
            SO_NEXT_LAYER(k)
#pragma unroll
            for (int v = 0; v < NL; v++) so_add_up<R>(v ? l2 : l1, rra[v], rxa[v]);
#pragma unroll
            for (int v = 0; v < NL; v++) { RRAV(k, v) = rra[v]; RXAV(k, v) = rxa[v]; }
        }
        // fork: (imx, isx) <- isx
        if constexpr (GM) {
#pragma unroll
            for (int v = 0; v < NL; v++) { rra[NL + v] = rra[v]; rxa[NL + v] = rxa[v]; }
        }
        for (; k >= ict; k--) {
            SO_NEXT_LAYER(k)
#pragma unroll
            for (int v = 0; v < NM * NL; v++) so_add_up<R>(v / NL ? l2 : l1, rra[v], rxa[v]);
#pragma unroll
            for (int v = 0; v < NM * NL; v++) { RRAV(k, v) = rra[v]; RXAV(k, v) = rxa[v]; }
        }
        if constexpr (GH) {
#pragma unroll
            for (int v = 0; v < NM * NL; v++) { rra[NM * NL + v] = rra[v]; rxa[NM * NL + v] = rxa[v]; }
        }
        for (; k >= 1; k--) {
            SO_NEXT_LAYER(k)
#pragma unroll
            for (int v = 0; v < NS; v++) so_add_up<R>(v / (NM * NL) ? l2 : l1, rra[v], rxa[v]);
#pragma unroll
            for (int v = 0; v < NS; v++) { RRAV(k, v) = rra[v]; RXAV(k, v) = rxa[v]; }
        }
#undef SO_NEXT_LAYER
    }
    // ---- sweep D ----
    // weights of the situations (:706-712, 778-781, 816-818)
    R ct[NS];
#pragma unroll
    for (int t = 0; t < NS; t++) {
        const int ihx = t / (NM * NL), imx = (t / NL) % NM, isx = t % NL;
        const R ch = ihx ? cc1 : (R)1.0 - cc1;
        const R cm = imx ? ch * cc2 : ch * ((R)1.0 - cc2);
        ct[t] = isx ? cm * cc3 : cm * ((R)1.0 - cc3);
    }
    R fsdir = 0, fsdif = 0, fall_sfc = 0;
    {
        R tda[NS], tta[NS], rsa[NS];
#pragma unroll
        for (int v = 0; v < NH; v++) { tda[v] = td0; tta[v] = 1; rsa[v] = 0; }
        // fluxes of level k: the top chain of situation t is TOP(t), its surface-side variant BOT(t) - both depend on the group of k
        auto level = [&](int k, auto grp_c, const R *bra, const R *bxa) {
            constexpr int G = decltype(grp_c)::value;                  // 0 high, 1 middle, 2 low group (and the surface)
            R fall = 0, fclr = 0, fupa = 0, fupc = 0;
#pragma unroll
            for (int t = 0; t < NS; t++) {
                const int top = G == 0 ? t / (NM * NL) : (G == 1 ? t / NL : t);
                const int bot = G == 0 ? t : (G == 1 ? t % (NM * NL) : t % NL);
                const R a_ = bra[bot], x_ = bxa[bot];
                const R denm = so_div((R)1., (R)1. - rsa[top] * x_);
                const R fdndir = tda[top];
                const R xx4 = tda[top] * a_, yy = tta[top] - tda[top];
                const R fdndif = (xx4 * rsa[top] + yy) * denm;
                const R fupdif = (xx4 + yy * x_) * denm;
                const R flxdn = fdndir + fdndif - fupdif;
                // the first sky situation (all-clear portions) starts the weighted sums: 0 + x * ct, as the reference's zeroed arrays give
                if (t == 0) { fupc = fupdif; fclr = flxdn; fupa = (R)0 + fupdif * ct[0]; fall = (R)0 + flxdn * ct[0]; }
                else { fupa = fupa + fupdif * ct[t]; fall = fall + flxdn * ct[t]; }
                if (k == np + 1) { fsdir = fsdir + fdndir * ct[t]; fsdif = fsdif + fdndif * ct[t]; }
            }
            P(26, k) = fall; P(28, k) = fupa;
            if constexpr (CLS != 0) { P(27, k) = fclr; P(29, k) = fupc; }
            if (k == np + 1) fall_sfc = fall;
        };
        using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>; using G2 = std::integral_constant<int, 2>;
        int k = 1;
        for (; k < ict; k++) {                     // high group: NH chains, NS variants below
            const L5 l1 = ld5(1, k); L5 l2 = l1;   // everything this level needs is requested before anything is stored
            if constexpr (GH) l2 = ld5(2, k);
            R bra[NS], bxa[NS];
#pragma unroll
            for (int v = 0; v < NS; v++) { bra[v] = RRAV(k, v); bxa[v] = RXAV(k, v); }
            level(k, G0{}, bra, bxa);
#pragma unroll
            for (int v = 0; v < NH; v++) so_add_down<R, false>(v ? l2 : l1, tda[v], tta[v], rsa[v]);
        }
        if constexpr (GM) {                        // fork: (ihx, imx) <- ihx
#pragma unroll
            for (int v = NH - 1; v >= 0; v--) { tda[v * NM + 1] = tda[v]; tta[v * NM + 1] = tta[v]; rsa[v * NM + 1] = rsa[v];
                                               tda[v * NM] = tda[v]; tta[v * NM] = tta[v]; rsa[v * NM] = rsa[v]; }
        }
        for (; k < icb; k++) {                     // middle group: NH NM chains, NM NL variants below
            const L5 l1 = ld5(1, k); L5 l2 = l1;
            if constexpr (GM) l2 = ld5(2, k);
            R bra[NM * NL], bxa[NM * NL];
#pragma unroll
            for (int v = 0; v < NM * NL; v++) { bra[v] = RRAV(k, v); bxa[v] = RXAV(k, v); }
            level(k, G1{}, bra, bxa);
#pragma unroll
            for (int v = 0; v < NH * NM; v++) so_add_down<R, false>(v % NM ? l2 : l1, tda[v], tta[v], rsa[v]);
        }
        if constexpr (GL) {                        // fork: t <- (ihx, imx)
#pragma unroll
            for (int v = NH * NM - 1; v >= 0; v--) { tda[v * NL + 1] = tda[v]; tta[v * NL + 1] = tta[v]; rsa[v * NL + 1] = rsa[v];
                                                    tda[v * NL] = tda[v]; tta[v * NL] = tta[v]; rsa[v * NL] = rsa[v]; }
        }
        for (; k <= np + 1; k++) {                 // low group and the surface: NS chains, NL variants below
            L5 l1{}, l2{};
            if (k <= np) { l1 = ld5(1, k); l2 = l1; if constexpr (GL) l2 = ld5(2, k); }
            R bra[NL], bxa[NL];
#pragma unroll
            for (int v = 0; v < NL; v++) { bra[v] = RRAV(k, v); bxa[v] = RXAV(k, v); }
            level(k, G2{}, bra, bxa);
            if (k <= np) {
#pragma unroll
                for (int v = 0; v < NS; v++) so_add_down<R, true>(v % NL ? l2 : l1, tda[v], tta[v], rsa[v]);
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
// k_sorad_sum: per (position, level) -- flux integration over the 35 passes in pass order (Eq. 6.1)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sorad_sum(SoradArgs<R> A, SoradOut<R> O)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y + 1;
    if (i >= A.m) return;
    const int ld = A.ld, m = A.m, K2 = A.np + 2;
    const bool one = i < A.cls_off[1];         // class 0: one sky situation of weight 1 - total-sky = clear-sky, two planes written
    R s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 5
    for (int p = 0; p < SO_NPASS; p++) {
        const R hk = A.hk[p];
        const R *q = A.scr + (((size_t)p * SO_NPLANE + 26) * K2 + k) * m + i;
        const R f0 = q[0], f2 = q[(size_t)2 * K2 * m];
        const R f1 = one ? f0 : q[(size_t)K2 * m], f3 = one ? f2 : q[(size_t)3 * K2 * m];
        s0 = s0 + f0 * hk; s1 = s1 + f1 * hk; s2 = s2 + f2 * hk; s3 = s3 + f3 * hk;
    }
    const size_t o = (size_t)(k - 1) * ld + A.perm[i];
    O.flx[o] = s0; O.flc[o] = s1; O.flxu[o] = s2; O.flcu[o] = s3;
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
                R tautob = tausto, asytob = so_div(asysto, ssatau), ssatob = so_div(ssatau, tautob) + (R)1.0e-8;
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
                    ssatob = so_div(uv ? (ssatau + tcb) : (ssatau + ssac * tcb), tautob) + (R)1.0e-8;
                    ssatob = ssatob < (R)0.999999 ? ssatob : (R)0.999999;
                    asytob = so_div(uv ? (asysto + asyc * tcb) : (asysto + asyc * ssac * tcb), ssatob * tautob);
                    const R tautof = tausto + tcf;
                    R ssatof = so_div(uv ? (ssatau + tcf) : (ssatau + ssac * tcf), tautof) + (R)1.0e-8;
                    ssatof = ssatof < (R)0.999999 ? ssatof : (R)0.999999;
                    const R asytof = so_div(uv ? (asysto + asyc * tcf) : (asysto + asyc * ssac * tcf), ssatof * tautof);
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
                        const R denm = so_div(ts, (R)1. - rsa * rs);
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
                        const R denm = so_div(ts, (R)1. - rs * rxa);
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
                        const R denm = so_div((R)1., (R)1. - rsa * rxa);      // Eqs. (6.15), (6.16)
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
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= A.m) return;
    const int i = A.perm[pos];                  // API arrays by column, workspace by position
    const SoradDev<R> &T = *Tp;
    const int np = A.np, ld = A.ld, m = A.m, K2 = np + 2;
#define OUT2(a, k) a[(size_t)((k) - 1) * ld + i]
    // (the level fluxes were summed over the passes by k_sorad_col)
    // surface band fluxes and direct / diffuse partition
    R fdiruv = 0, fdifuv = 0, fdirpar = 0, fdifpar = 0, fdirir = 0, fdifir = 0, band[8], drb[8], dfb[8];
    for (int b = 0; b < 8; b++) { band[b] = 0; drb[b] = 0; dfb[b] = 0; }
    for (int p = 0; p < SO_NPASS; p++) {
        const R hk = A.hk[p], fs = A.psum[((size_t)p * 3 + 0) * m + pos], fd = A.psum[((size_t)p * 3 + 1) * m + pos];
        const int b = p < 5 ? p : 5 + (p - 5) / 10;
        band[b] = band[b] + A.psum[((size_t)p * 3 + 2) * m + pos] * hk; drb[b] = drb[b] + fs * hk; dfb[b] = dfb[b] + fd * hk;
        if (p < 4) { fdiruv = fdiruv + fs * hk; fdifuv = fdifuv + fd * hk; }
        else if (p == 4) { fdirpar = fs * hk; fdifpar = fd * hk; }
        else { fdirir = fdirir + fs * hk; fdifir = fdifir + fd * hk; }
    }
    // flux reductions: running column amounts, two table look-ups per level
    const R snt = (R)1.0 / A.cosz[i];
    const R scal0 = A.colv[5 * (size_t)m + pos];
    const int ntop = (int)A.colv[6 * (size_t)m + pos];
    const R cnt = (R)165.22 * snt;
    R so2o = scal0 * cnt, so2c = ((R)789. * A.co2) * scal0;
    const R flx_top = OUT2(O.flx, ntop);
    R dftop = 0, dfsfc = 0;
    for (int k = 1; k <= np + 1; k++) {
        if (k > 1) { const R sc = A.lay[((size_t)3 * K2 + (k - 1)) * m + pos]; so2o = so2o + sc * cnt; so2c = so2c + ((R)789. * A.co2) * sc; }
        R df = (R)0.0633 * ((R)1. - gr_exp<R>((R)-0.000155 * sqrt(so2o)));
        {
            const R u1 = (R)-3.0, du = (R)0.15, w1 = (R)-4.0, dw = (R)0.15;
            const R x0 = u1 + (R)43 * du, y0 = w1 + (R)37 * dw, x1 = u1 - (R)0.5 * du, y1 = w1 - (R)0.5 * dw;
            R ulog = gr_log10<R>(so2c * snt); ulog = ulog < x0 ? ulog : x0;
            R wlog = gr_log10<R>(A.swh[(size_t)k * m + pos] * snt); wlog = wlog < y0 ? wlog : y0;
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
