// lw_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the RRTMG_LW column solver.
//
// What is computed follows the reference (file:line cited at each kernel; LW = GEOSirrad_GridComp/RRTMG/
// rrtmg_lw/gcm_model/src); how it is computed is ours:
//   lane = column, all HBM traffic coalesced over the column dimension, no cross-lane traffic, no LDS
//   dependence, no MFMA (the path is table interpolation + first-order vertical recurrences).
//   k_validate_pwv : per column  - input checks, precipitable water, "any cloud" flag
//   k_setcoef      : per (layer,column) - p/T interpolation record shared by all 16 bands
//   k_lw_bands     : per (column, band): fused taumol -> rtrnmc; blockIdx.y selects the band body
//                    (heaviest bands first); down sweep keeps the band's g-point radiances in
//                    registers, up sweep re-reads the (absorptivity, source) pairs it parked in HBM
//   k_lw_reduce    : per (level,column) - fixed-order sum of the 16 band partials (bitwise reproducible)
#pragma once
#include "lw_device.hpp"
#include <type_traits>

namespace geosrad {

#define GR_DEV __device__ __forceinline__

template <typename R> GR_DEV R gr_exp(R x);
template <> GR_DEV float gr_exp<float>(float x) { return expf(x); }
template <> GR_DEV double gr_exp<double>(double x) { return exp(x); }
template <typename R> GR_DEV R gr_log(R x);
template <> GR_DEV float gr_log<float>(float x) { return logf(x); }
template <> GR_DEV double gr_log<double>(double x) { return log(x); }
template <typename R> GR_DEV R gr_pow(R x, R y);
template <> GR_DEV float gr_pow<float>(float x, float y) { return powf(x, y); }
template <> GR_DEV double gr_pow<double>(double x, double y) { return pow(x, y); }

GR_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

constexpr int pad4(int n) { return (n + 3) & ~3; }

// ---------------------------------------------------------------------------------------------------
// k_validate_pwv: one thread per column.
//   - the reference's input assertions (LW/rrtmg_lw_rad.F90:209-318) -> error bits
//   - pwvcm (LW/rrtmg_lw_setcoef.F90:206-272), same summation order (bottom-up)
//   - colcloudy = any(cldf > 0): lets later kernels skip McICA work for clear columns without
//     changing results (all masks false, clearCounts = ngpt)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_validate_pwv(LwArgs<R> A, const LwDev<R> *__restrict__ T)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const int ld = A.ld, nlay = A.nlay;
    const R amd = (R)28.9660, amw = (R)18.0160;
    const R grav = T->grav, avogad = T->avogad;
    uint32_t err = 0;
    const R *chk[17] = {A.play, A.tlay, A.h2o, A.o3, A.co2, A.ch4, A.n2o, A.o2, A.cfc11, A.cfc12, A.cfc22, A.ccl4,
                        A.cldf, A.ciwp, A.clwp, A.rei, A.rel};
    R amttl = 0, wvttl = 0;
    bool cloudy = false;
    R pprev = A.plev[col];
    if (pprev < 0 || A.tlev[col] < 0) err |= 1u << 17;
    for (int lay = 0; lay < nlay; lay++) {
        const size_t i = (size_t)lay * ld + col;
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (chk[k][i] < 0) err |= 1u << k;
        const R pup = A.plev[i + ld];
        if (pup < 0 || A.tlev[i + ld] < 0) err |= 1u << 17;
        if (A.tauaer)
            for (int ib = 0; ib < NB_LW; ib++)
                if (A.tauaer[((size_t)ib * nlay + lay) * ld + col] < 0) err |= 1u << 20;
        const R h2o = A.h2o[i];
        const R amm = ((R)1. - h2o) * amd + h2o * amw;
        const R coldry = (pprev - pup) * (R)1.e3 * avogad / ((R)1.e2 * grav * amm * ((R)1. + h2o));
        const R btemp = h2o * coldry;
        amttl = amttl + coldry + btemp;
        wvttl = wvttl + btemp;
        if (A.cldf[i] > 0) cloudy = true;
        // pressure ordering (LW/rrtmg_lw_setcoef.F90:443-453): lower-atmosphere layer above an upper one
        pprev = pup;
    }
    if (A.tsfc[col] < 0) err |= 1u << 18;
    for (int ib = 0; ib < NB_LW; ib++)
        if (A.emis[(size_t)ib * ld + col] < 0) err |= 1u << 19;
    const R wvsh = (amw * wvttl) / (amd * amttl);
    A.pwvcm[col] = wvsh * ((R)1.e3 * A.plev[col]) / ((R)1.e2 * grav);
    A.colcloudy[col] = cloudy ? 1 : 0;
    if (!cloudy) {
        // clear column: all sub-columns clear in every super-layer (cloud_subcol_gen.F90:649-659)
        for (int k = 0; k < 4; k++) A.clearCounts[(size_t)k * ld + col] = NG_LW;
    }
    // pressure misordering: plog > 4.56 somewhere above a layer with plog <= 4.56
    {
        bool upper = false, bad = false;
        for (int lay = 0; lay < nlay; lay++) {
            const bool lower = gr_log<R>(A.play[(size_t)lay * ld + col]) > (R)4.56;
            if (lower && upper) bad = true;
            if (!lower) upper = true;
        }
        if (bad) err |= 1u << ERR_PRESSURE_ORDER;
    }
    if (err) atomicOr(A.err, err);
}

// ---------------------------------------------------------------------------------------------------
// k_setcoef: one thread per (layer, column); blockIdx.y = layer.  LW/rrtmg_lw_setcoef.F90:401-579
// (everything that does not depend on the band; Planck terms are interpolated inside the band kernel).
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_setcoef(LwArgs<R> A, const LwDev<R> *__restrict__ T)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lay = blockIdx.y;
    if (col >= A.ncol) return;
    const int ld = A.ld, n = A.ncol, nlay = A.nlay;
    const size_t i = (size_t)lay * ld + col;
    const R amd = (R)28.9660, amw = (R)18.0160;
    const R stpfac = (R)296. / (R)1013.;
    const R pavel = A.play[i], tavel = A.tlay[i], h2o = A.h2o[i];
    const R amm = ((R)1. - h2o) * amd + h2o * amw;
    const R coldry = (A.plev[i] - A.plev[i + ld]) * (R)1.e3 * T->avogad / ((R)1.e2 * T->grav * amm * ((R)1. + h2o));
    const R summol = A.co2[i] + A.o3[i] + A.n2o[i] + A.ch4[i] + A.o2[i];
    const R wbroad = coldry * ((R)1. - summol);
    const R wv = h2o * coldry;

    const R plog = gr_log<R>(pavel);
    const int jp = clampi((int)((R)36. - (R)5 * (plog + (R)0.04)), 1, 58);
    const int jp1 = jp + 1;
    const R fp = (R)5. * (T->preflog[jp - 1] - plog);
    const R dt0 = (tavel - T->tref[jp - 1]) / (R)15.;
    const int jt = clampi((int)((R)3. + dt0), 1, 4);
    const R ft = dt0 - (R)(jt - 3);
    const R dt1 = (tavel - T->tref[jp1 - 1]) / (R)15.;
    const int jt1 = clampi((int)((R)3. + dt1), 1, 4);
    const R ft1 = dt1 - (R)(jt1 - 3);
    const R water = wv / coldry;
    const R scalefac = pavel * stpfac / tavel;
    const bool lower = plog > (R)4.56;
    R forfac = scalefac / ((R)1. + water), forfrac, selffac, selffrac = 0;
    int indfor, indself = 1;
    if (lower) {
        R factor = ((R)332. - tavel) / (R)36.;
        indfor = clampi((int)factor, 1, 2);
        forfrac = factor - (R)indfor;
        selffac = water * forfac;
        factor = (tavel - (R)188.) / (R)7.2;
        indself = clampi((int)factor - 7, 1, 9);
        selffrac = factor - (R)(indself + 7);
    } else {
        R factor = (tavel - (R)188.) / (R)36.;
        indfor = 3;
        forfrac = factor - (R)1.;
        selffac = 0;
    }
    const R scaleminor = pavel / tavel;
    const R scaleminorn2 = (pavel / tavel) * (wbroad / (coldry + wv));
    const R factor = (tavel - (R)180.8) / (R)7.2;
    const int indminor = clampi((int)factor, 1, 18);
    const R minorfrac = factor - (R)indminor;
    const R colh2o = (R)1.e-20 * h2o * coldry;
    const R compfp = (R)1. - fp;

    R *sc = A.sc + (size_t)lay * n + col;
    const size_t fs = (size_t)nlay * n;
    sc[SC_FAC10 * fs] = compfp * ft;
    sc[SC_FAC00 * fs] = compfp * ((R)1. - ft);
    sc[SC_FAC11 * fs] = fp * ft1;
    sc[SC_FAC01 * fs] = fp * ((R)1. - ft1);
    sc[SC_COLDRY * fs] = coldry;
    sc[SC_FORFAC * fs] = colh2o * forfac;
    sc[SC_FORFRAC * fs] = forfrac;
    sc[SC_SELFFAC * fs] = colh2o * selffac;
    sc[SC_SELFFRAC * fs] = selffrac;
    sc[SC_MINORFRAC * fs] = minorfrac;
    sc[SC_SCALEMINOR * fs] = scaleminor;
    sc[SC_SCALEMINORN2 * fs] = scaleminorn2;
    sc[SC_COLBRD * fs] = (R)1.e-20 * wbroad;
    A.scidx[(size_t)lay * n + col] = pack_idx(jp, jt, jt1, indfor, indself, indminor, lower ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------
// band bodies: gas optical depth tau[g] and Planck fraction pf[g] of ONE layer of ONE column for all
// g-points of a band (LW/rrtmg_lw_taumol.F90:155-3126, one struct per taugbN).
// ---------------------------------------------------------------------------------------------------
template <typename R> struct Layer {
    R fac00, fac01, fac10, fac11, coldry, forfac, forfrac, selffac, selffrac, minorfrac, scaleminor, scaleminorn2,
        colbrd, pavel;
    int jp, jt, jt1, indfor, indself, indminor;
    bool lower;
    size_t i;  // API index (lay*ld + col) of this cell, for on-demand gas loads
};

template <typename R> GR_DEV R colamt(const R *__restrict__ vmr, const Layer<R> &L)
{
    return (R)1.e-20 * vmr[L.i] * L.coldry;
}
// "require some minor absorbers to be non-zero" (LW/rrtmg_lw_setcoef.F90:560-564)
template <typename R> GR_DEV R colamt_nz(const R *__restrict__ vmr, const Layer<R> &L)
{
    R c = (R)1.e-20 * vmr[L.i] * L.coldry;
    return c == (R)0 ? (R)1.e-32 * L.coldry : c;
}

// row fetch: NGP contiguous reals, 16-byte aligned -> dwordx4 loads
template <typename R, int NG> GR_DEV void ldrow(const R *__restrict__ p, R (&o)[NG])
{
    constexpr int NGP = pad4(NG);
    constexpr int VW = 16 / sizeof(R);
    using V = typename std::conditional<sizeof(R) == 4, float4, double2>::type;
    const V *q = reinterpret_cast<const V *>(__builtin_assume_aligned(p, 16));
    R tmp[NGP];
#pragma unroll
    for (int k = 0; k < NGP / VW; k++) {
        V v = q[k];
        if constexpr (sizeof(R) == 4) {
            tmp[4 * k] = v.x; tmp[4 * k + 1] = v.y; tmp[4 * k + 2] = v.z; tmp[4 * k + 3] = v.w;
        } else {
            tmp[2 * k] = v.x; tmp[2 * k + 1] = v.y;
        }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) o[g] = tmp[g];
}
// acc[g] (+)= c * row[g]
template <typename R, int NG, bool INIT> GR_DEV void axrow(R (&acc)[NG], R c, const R *__restrict__ tab, int row)
{
    R r[NG];
    ldrow<R, NG>(tab + (size_t)row * pad4(NG), r);
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = INIT ? c * r[g] : acc[g] + c * r[g];
}
// acc[g] += s * (t[i][g] + f * (t[i+1][g] - t[i][g]))      (linear interpolation between two rows)
template <typename R, int NG> GR_DEV void add_lin(R (&acc)[NG], R s, R f, const R *__restrict__ tab, int row)
{
    R a[NG], b[NG];
    ldrow<R, NG>(tab + (size_t)row * pad4(NG), a);
    ldrow<R, NG>(tab + (size_t)(row + 1) * pad4(NG), b);
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = acc[g] + s * (a[g] + f * (b[g] - a[g]));
}
template <typename R, int NG> GR_DEV void lin(R (&o)[NG], R f, const R *__restrict__ tab, int row)
{
    R a[NG], b[NG];
    ldrow<R, NG>(tab + (size_t)row * pad4(NG), a);
    ldrow<R, NG>(tab + (size_t)(row + 1) * pad4(NG), b);
#pragma unroll
    for (int g = 0; g < NG; g++) o[g] = a[g] + f * (b[g] - a[g]);
}
// minor gas on a (species parameter, T) grid, table rows [indm][jm] with NSP species rows per T
// (e.g. LW/rrtmg_lw_taumol.F90:546-551); 0-based row = (indm-1)*NSP + (jm-1)
template <typename R, int NG, int NSP>
GR_DEV void minor2(R (&o)[NG], const R *__restrict__ tab, int jm, int indm, R fm, R minorfrac)
{
    R m1[NG], m2[NG];
    lin<R, NG>(m1, fm, tab, (indm - 1) * NSP + (jm - 1));
    lin<R, NG>(m2, fm, tab, indm * NSP + (jm - 1));
#pragma unroll
    for (int g = 0; g < NG; g++) o[g] = m1[g] + minorfrac * (m2[g] - m1[g]);
}

template <typename R> struct Spec { R speccomb, specparm, fs; int js; };
// binary-species parameter (e.g. LW/rrtmg_lw_taumol.F90:435-441)
template <typename R> GR_DEV Spec<R> spec(R cola, R rat, R colb, R mult, R oneminus)
{
    Spec<R> s;
    s.speccomb = cola + rat * colb;
    s.specparm = cola / s.speccomb;
    if (s.specparm >= oneminus) s.specparm = oneminus;
    const R sm = mult * s.specparm;
    const int j = (int)sm;
    s.js = 1 + j;
    s.fs = sm - (R)j;
    return s;
}

// key-species term of one reference-pressure side of a binary (9-species-row) lower-atmosphere band,
// with the cubic treatment near specparm -> 0 / 1 (LW/rrtmg_lw_taumol.F90:482-606).
// ind = 1-based row of (jp|jp+1, jt|jt1, js); facA/facB = (fac00,fac10) or (fac01,fac11).
template <typename R, int NG, bool INIT>
GR_DEV void major_a(R (&acc)[NG], const R *__restrict__ absa, int ind, const Spec<R> &sp, R facA, R facB)
{
    R c0, c1, c2 = 0;
    int base = ind - 1;  // 0-based
    const bool lo = sp.specparm < (R)0.125, hi = sp.specparm > (R)0.875;
    if (lo || hi) {
        const R p = lo ? sp.fs - (R)1 : -sp.fs;
        const R p4 = ((p * p) * p) * p;
        const R fk0 = p4, fk1 = (R)1 - p - (R)2.0 * p4, fk2 = p + p4;
        if (lo) { c0 = fk0; c1 = fk1; c2 = fk2; }
        else { c0 = fk2; c1 = fk1; c2 = fk0; base -= 1; }
    } else {
        c0 = (R)1. - sp.fs; c1 = sp.fs;
    }
    R t[NG];
    axrow<R, NG, true>(t, c0 * facA, absa, base);
    axrow<R, NG, false>(t, c1 * facA, absa, base + 1);
    if (lo || hi) axrow<R, NG, false>(t, c2 * facA, absa, base + 2);
    axrow<R, NG, false>(t, c0 * facB, absa, base + 9);
    axrow<R, NG, false>(t, c1 * facB, absa, base + 10);
    if (lo || hi) axrow<R, NG, false>(t, c2 * facB, absa, base + 11);
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = INIT ? sp.speccomb * t[g] : acc[g] + sp.speccomb * t[g];
}
// upper-atmosphere binary side, 5 species rows (e.g. :706-716)
template <typename R, int NG, bool INIT>
GR_DEV void major_b5(R (&acc)[NG], const R *__restrict__ absb, int ind, const Spec<R> &sp, R facA, R facB)
{
    R t[NG];
    const R c0 = (R)1. - sp.fs, c1 = sp.fs;
    axrow<R, NG, true>(t, c0 * facA, absb, ind - 1);
    axrow<R, NG, false>(t, c1 * facA, absb, ind);
    axrow<R, NG, false>(t, c0 * facB, absb, ind + 4);
    axrow<R, NG, false>(t, c1 * facB, absb, ind + 5);
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = INIT ? sp.speccomb * t[g] : acc[g] + sp.speccomb * t[g];
}
// single key species: col * 4-point (p,T) interpolation (e.g. :240-244); ind0/ind1 1-based
template <typename R, int NG>
GR_DEV void major1(R (&acc)[NG], const R *__restrict__ tab, int ind0, int ind1, const Layer<R> &L, R col)
{
    R t[NG];
    axrow<R, NG, true>(t, L.fac00, tab, ind0 - 1);
    axrow<R, NG, false>(t, L.fac10, tab, ind0);
    axrow<R, NG, false>(t, L.fac01, tab, ind1 - 1);
    axrow<R, NG, false>(t, L.fac11, tab, ind1);
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = col * t[g];
}
// "too much of a minor gas" column adjustment (e.g. :461-468)
template <typename R> GR_DEV R adjcol(R colx, R coldry, R chiref, R thresh, R a, R pw)
{
    const R rat = (R)1.e20 * (colx / coldry) / chiref;
    if (rat > thresh) return (a + gr_pow<R>(rat - a, pw)) * chiref * coldry * (R)1.e-20;
    return colx;
}

#define CHI(m, j) (T.chi_mls[((j) - 1) * 7 + ((m) - 1)])
#define RAT(pair, j) (T.rat[(pair) * 60 + (j)])
#define IND0A(n) (((L.jp - 1) * 5 + (L.jt - 1)) * (n))
#define IND1A(n) ((L.jp * 5 + (L.jt1 - 1)) * (n))
#define IND0B(n) (((L.jp - 13) * 5 + (L.jt - 1)) * (n))
#define IND1B(n) (((L.jp - 12) * 5 + (L.jt1 - 1)) * (n))
#define ADD_SELF() add_lin<R, NG>(tau, L.selffac, L.selffrac, B.selfref, L.indself - 1)
#define ADD_FOR() add_lin<R, NG>(tau, L.forfac, L.forfrac, B.forref, L.indfor - 1)
#define BAND_HEAD(ib, ng, g0)                                                  \
    static constexpr int IB = ib, NG = ng, G0 = g0;                            \
    template <typename R>                                                      \
    GR_DEV static void tau_pf(const LwDev<R> &T, const LwArgs<R> &A, const Layer<R> &L, R (&tau)[NG], R (&pf)[NG])

// Planck fraction helpers
template <typename R, int NG> GR_DEV void pf_const(R (&pf)[NG], const R *__restrict__ frac) { ldrow<R, NG>(frac, pf); }
template <typename R, int NG> GR_DEV void pf_interp(R (&pf)[NG], const R *__restrict__ frac, const Spec<R> &s)
{
    lin<R, NG>(pf, s.fs, frac, s.js - 1);
}

struct Band1 {  // 10-350 cm-1: h2o; minor n2 (:214-291)
    BAND_HEAD(1, 10, 0)
    {
        const BandTab<R> &B = T.b[IB];
        const R colh2o = colamt(A.h2o, L);
        const R scalen2 = L.colbrd * L.scaleminorn2;
        R corradj;
        if (L.lower) {
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            add_lin<R, NG>(tau, scalen2, L.minorfrac, B.m[0], L.indminor - 1);
            corradj = L.pavel < (R)250. ? (R)1. - (R)0.15 * ((R)250. - L.pavel) / (R)154.4 : (R)1;
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colh2o);
            ADD_FOR();
            add_lin<R, NG>(tau, scalen2, L.minorfrac, B.m[1], L.indminor - 1);
            corradj = (R)1. - (R)0.15 * (L.pavel / (R)95.6);
            pf_const<R, NG>(pf, B.fracrefb);
        }
#pragma unroll
        for (int g = 0; g < NG; g++) tau[g] = corradj * tau[g];
    }
};

struct Band2 {  // 350-500: h2o (:296-363)
    BAND_HEAD(2, 12, 10)
    {
        const BandTab<R> &B = T.b[IB];
        const R colh2o = colamt(A.h2o, L);
        if (L.lower) {
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            const R corradj = (R)1. - (R).05 * (L.pavel - (R)100.) / (R)900.;
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = corradj * tau[g];
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colh2o);
            ADD_FOR();
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band3 {  // 500-630: h2o,co2; minor n2o (:368-727)
    BAND_HEAD(3, 16, 22)
    {
        const BandTab<R> &B = T.b[IB];
        const R colh2o = colamt(A.h2o, L), colco2 = colamt_nz(A.co2, L), coln2o = colamt_nz(A.n2o, L);
        const R adjn2o = adjcol<R>(coln2o, L.coldry, CHI(4, L.jp + 1), (R)1.5, (R)0.5, (R)0.65);
        R m[NG];
        if (L.lower) {
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            const Spec<R> sm = spec<R>(colh2o, CHI(1, 3) / CHI(2, 3), colco2, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 9) / CHI(2, 9), colco2, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            minor2<R, NG, 9>(m, B.m[0], sm.js, L.indminor, sm.fs, L.minorfrac);
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            const R rp = CHI(1, 13) / CHI(2, 13);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 4, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 4, T.oneminus);
            const Spec<R> sm = spec<R>(colh2o, rp, colco2, 4, T.oneminus);
            major_b5<R, NG, true>(tau, B.absb, IND0B(5) + sp.js, sp, L.fac00, L.fac10);
            major_b5<R, NG, false>(tau, B.absb, IND1B(5) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_FOR();
            minor2<R, NG, 5>(m, B.m[1], sm.js, L.indminor, sm.fs, L.minorfrac);
            pf_interp<R, NG>(pf, B.fracrefb, sm);  // refrat_m_b == refrat_planck_b (:421-422)
        }
#pragma unroll
        for (int g = 0; g < NG; g++) tau[g] = tau[g] + adjn2o * m[g];
    }
};

struct Band4 {  // 630-700: h2o,co2 | o3,co2 (:732-962)
    BAND_HEAD(4, 14, 38)
    {
        const BandTab<R> &B = T.b[IB];
        const R colco2 = colamt_nz(A.co2, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 11) / CHI(2, 11), colco2, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            const R colo3 = colamt_nz(A.o3, L);
            const Spec<R> sp = spec<R>(colo3, RAT(RAT_O3CO2, L.jp), colco2, 4, T.oneminus);
            const Spec<R> sp1 = spec<R>(colo3, RAT(RAT_O3CO2, L.jp + 1), colco2, 4, T.oneminus);
            const Spec<R> spl = spec<R>(colo3, CHI(3, 13) / CHI(2, 13), colco2, 4, T.oneminus);
            major_b5<R, NG, true>(tau, B.absb, IND0B(5) + sp.js, sp, L.fac00, L.fac10);
            major_b5<R, NG, false>(tau, B.absb, IND1B(5) + sp1.js, sp1, L.fac01, L.fac11);
            pf_interp<R, NG>(pf, B.fracrefb, spl);
            // empirical stratospheric-cooling tweak (:951-957)
            tau[7] *= (R)0.92; tau[8] *= (R)0.88; tau[9] *= (R)1.07; tau[10] *= (R)1.1;
            tau[11] *= (R)0.99; tau[12] *= (R)0.88; tau[13] *= (R)0.943;
        }
    }
};

struct Band5 {  // 700-820: h2o,co2 | o3,co2; minor o3, ccl4 (:967-1229)
    BAND_HEAD(5, 16, 52)
    {
        const BandTab<R> &B = T.b[IB];
        const R colco2 = colamt_nz(A.co2, L), colo3 = colamt_nz(A.o3, L), colccl4 = colamt(A.ccl4, L);
        R c4[NG];
        ldrow<R, NG>(B.m[1], c4);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            const Spec<R> sm = spec<R>(colh2o, CHI(1, 7) / CHI(2, 7), colco2, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 5) / CHI(2, 5), colco2, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            R m[NG];
            minor2<R, NG, 9>(m, B.m[0], sm.js, L.indminor, sm.fs, L.minorfrac);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + m[g] * colo3 + colccl4 * c4[g];
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            const Spec<R> sp = spec<R>(colo3, RAT(RAT_O3CO2, L.jp), colco2, 4, T.oneminus);
            const Spec<R> sp1 = spec<R>(colo3, RAT(RAT_O3CO2, L.jp + 1), colco2, 4, T.oneminus);
            const Spec<R> spl = spec<R>(colo3, CHI(3, 43) / CHI(2, 43), colco2, 4, T.oneminus);
            major_b5<R, NG, true>(tau, B.absb, IND0B(5) + sp.js, sp, L.fac00, L.fac10);
            major_b5<R, NG, false>(tau, B.absb, IND1B(5) + sp1.js, sp1, L.fac01, L.fac11);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + colccl4 * c4[g];
            pf_interp<R, NG>(pf, B.fracrefb, spl);
        }
    }
};

struct Band6 {  // 820-980: h2o; minor co2, cfc11, cfc12 (:1234-1322)
    BAND_HEAD(6, 8, 68)
    {
        const BandTab<R> &B = T.b[IB];
        const R colcfc11 = colamt(A.cfc11, L), colcfc12 = colamt(A.cfc12, L);
        R c11[NG], c12[NG];
        ldrow<R, NG>(B.m[1], c11);
        ldrow<R, NG>(B.m[2], c12);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L), colco2 = colamt_nz(A.co2, L);
            const R adjco2 = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.77);
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            add_lin<R, NG>(tau, adjco2, L.minorfrac, B.m[0], L.indminor - 1);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + colcfc11 * c11[g] + colcfc12 * c12[g];
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = (R)0.0 + colcfc11 * c11[g] + colcfc12 * c12[g];
        }
        pf_const<R, NG>(pf, B.fracrefa);
    }
};

struct Band7 {  // 980-1080: h2o,o3 | o3; minor co2 (:1327-1601)
    BAND_HEAD(7, 12, 76)
    {
        const BandTab<R> &B = T.b[IB];
        const R colco2 = colamt_nz(A.co2, L), colo3 = colamt_nz(A.o3, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            const R rp = CHI(1, 3) / CHI(3, 3);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OO3, L.jp), colo3, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OO3, L.jp + 1), colo3, 8, T.oneminus);
            const Spec<R> sm = spec<R>(colh2o, rp, colo3, 8, T.oneminus);  // refrat_m_a == refrat_planck_a
            const R adjco2 = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)3.0, (R)0.79);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            R m[NG];
            minor2<R, NG, 9>(m, B.m[0], sm.js, L.indminor, sm.fs, L.minorfrac);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + adjco2 * m[g];
            pf_interp<R, NG>(pf, B.fracrefa, sm);
        } else {
            const R adjco2 = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.79);
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colo3);
            add_lin<R, NG>(tau, adjco2, L.minorfrac, B.m[1], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefb);
            tau[5] *= (R)0.92; tau[6] *= (R)0.88; tau[7] *= (R)1.07; tau[8] *= (R)1.1; tau[9] *= (R)0.99; tau[10] *= (R)0.855;
        }
    }
};

struct Band8 {  // 1080-1180: h2o | o3; minor co2, o3, n2o, cfc12, cfc22 (:1606-1733)
    BAND_HEAD(8, 8, 88)
    {
        const BandTab<R> &B = T.b[IB];
        const R colco2 = colamt_nz(A.co2, L), colo3 = colamt_nz(A.o3, L), coln2o = colamt_nz(A.n2o, L);
        const R colcfc12 = colamt(A.cfc12, L), colcfc22 = colamt(A.cfc22, L);
        const R adjco2 = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.65);
        R c12[NG], c22[NG];
        ldrow<R, NG>(B.m[5], c12);
        ldrow<R, NG>(B.m[6], c22);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            add_lin<R, NG>(tau, adjco2, L.minorfrac, B.m[0], L.indminor - 1);
            add_lin<R, NG>(tau, colo3, L.minorfrac, B.m[2], L.indminor - 1);
            add_lin<R, NG>(tau, coln2o, L.minorfrac, B.m[3], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colo3);
            add_lin<R, NG>(tau, adjco2, L.minorfrac, B.m[1], L.indminor - 1);
            add_lin<R, NG>(tau, coln2o, L.minorfrac, B.m[4], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefb);
        }
#pragma unroll
        for (int g = 0; g < NG; g++) tau[g] = tau[g] + colcfc12 * c12[g] + colcfc22 * c22[g];
    }
};

struct Band9 {  // 1180-1390: h2o,ch4 | ch4; minor n2o (:1738-2001)
    BAND_HEAD(9, 12, 96)
    {
        const BandTab<R> &B = T.b[IB];
        const R colch4 = colamt_nz(A.ch4, L), coln2o = colamt_nz(A.n2o, L);
        const R adjn2o = adjcol<R>(coln2o, L.coldry, CHI(4, L.jp + 1), (R)1.5, (R)0.5, (R)0.65);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp), colch4, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp + 1), colch4, 8, T.oneminus);
            const Spec<R> sm = spec<R>(colh2o, CHI(1, 3) / CHI(6, 3), colch4, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 9) / CHI(6, 9), colch4, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            R m[NG];
            minor2<R, NG, 9>(m, B.m[0], sm.js, L.indminor, sm.fs, L.minorfrac);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + adjn2o * m[g];
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colch4);
            add_lin<R, NG>(tau, adjn2o, L.minorfrac, B.m[1], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band10 {  // 1390-1480: h2o (:2006-2072)
    BAND_HEAD(10, 6, 108)
    {
        const BandTab<R> &B = T.b[IB];
        const R colh2o = colamt(A.h2o, L);
        if (L.lower) {
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colh2o);
            ADD_FOR();
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band11 {  // 1480-1800: h2o; minor o2 (:2077-2160)
    BAND_HEAD(11, 8, 114)
    {
        const BandTab<R> &B = T.b[IB];
        const R colh2o = colamt(A.h2o, L);
        const R scaleo2 = colamt(A.o2, L) * L.scaleminor;
        if (L.lower) {
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colh2o);
            ADD_SELF(); ADD_FOR();
            add_lin<R, NG>(tau, scaleo2, L.minorfrac, B.m[0], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colh2o);
            ADD_FOR();
            add_lin<R, NG>(tau, scaleo2, L.minorfrac, B.m[1], L.indminor - 1);
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band12 {  // 1800-2080: h2o,co2 | nothing (:2165-2345)
    BAND_HEAD(12, 8, 122)
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L), colco2 = colamt_nz(A.co2, L);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 10) / CHI(2, 10), colco2, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) { tau[g] = 0; pf[g] = 0; }
        }
    }
};

struct Band13 {  // 2080-2250: h2o,n2o | (o3 minor); minor co2, co (:2350-2585)
    BAND_HEAD(13, 4, 130)
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L), coln2o = colamt_nz(A.n2o, L), colco2 = colamt_nz(A.co2, L);
            const R colco = (R)1.e-32 * L.coldry;  // covmr == 0 in GEOS (LW/rrtmg_lw_rad.F90:518, setcoef :564)
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2ON2O, L.jp), coln2o, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2ON2O, L.jp + 1), coln2o, 8, T.oneminus);
            const Spec<R> smco2 = spec<R>(colh2o, CHI(1, 1) / CHI(4, 1), coln2o, 8, T.oneminus);
            const Spec<R> smco = spec<R>(colh2o, CHI(1, 3) / CHI(4, 3), coln2o, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 5) / CHI(4, 5), coln2o, 8, T.oneminus);
            const R adjco2 = adjcol<R>(colco2, L.coldry, (R)3.55e-4, (R)3.0, (R)2.0, (R)0.68);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            R m[NG], mc[NG];
            minor2<R, NG, 9>(m, B.m[0], smco2.js, L.indminor, smco2.fs, L.minorfrac);
            minor2<R, NG, 9>(mc, B.m[1], smco.js, L.indminor, smco.fs, L.minorfrac);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + adjco2 * m[g] + colco * mc[g];
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            const R colo3 = colamt_nz(A.o3, L);
            R m[NG];
            lin<R, NG>(m, L.minorfrac, B.m[2], L.indminor - 1);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = colo3 * m[g];
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band14 {  // 2250-2380: co2 (:2590-2653)
    BAND_HEAD(14, 2, 134)
    {
        const BandTab<R> &B = T.b[IB];
        const R colco2 = colamt_nz(A.co2, L);
        if (L.lower) {
            major1<R, NG>(tau, B.absa, IND0A(1) + 1, IND1A(1) + 1, L, colco2);
            ADD_SELF(); ADD_FOR();
            pf_const<R, NG>(pf, B.fracrefa);
        } else {
            major1<R, NG>(tau, B.absb, IND0B(1) + 1, IND1B(1) + 1, L, colco2);
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

struct Band15 {  // 2380-2600: n2o,co2 | nothing; minor n2 (:2658-2866)
    BAND_HEAD(15, 2, 136)
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            const R coln2o = colamt_nz(A.n2o, L), colco2 = colamt_nz(A.co2, L);
            const R rp = CHI(4, 1) / CHI(2, 1);
            const Spec<R> sp = spec<R>(coln2o, RAT(RAT_N2OCO2, L.jp), colco2, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(coln2o, RAT(RAT_N2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            const Spec<R> sm = spec<R>(coln2o, rp, colco2, 8, T.oneminus);  // refrat_m_a == refrat_planck_a
            const R scalen2 = L.colbrd * L.scaleminor;
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            R m[NG];
            minor2<R, NG, 9>(m, B.m[0], sm.js, L.indminor, sm.fs, L.minorfrac);
#pragma unroll
            for (int g = 0; g < NG; g++) tau[g] = tau[g] + scalen2 * m[g];
            pf_interp<R, NG>(pf, B.fracrefa, sm);
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) { tau[g] = 0; pf[g] = 0; }
        }
    }
};

struct Band16 {  // 2600-3250: h2o,ch4 | ch4 (:2871-3126)
    BAND_HEAD(16, 2, 138)
    {
        const BandTab<R> &B = T.b[IB];
        const R colch4 = colamt_nz(A.ch4, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            const Spec<R> sp = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp), colch4, 8, T.oneminus);
            const Spec<R> sp1 = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp + 1), colch4, 8, T.oneminus);
            const Spec<R> spl = spec<R>(colh2o, CHI(1, 6) / CHI(6, 6), colch4, 8, T.oneminus);
            major_a<R, NG, true>(tau, B.absa, IND0A(9) + sp.js, sp, L.fac00, L.fac10);
            major_a<R, NG, false>(tau, B.absa, IND1A(9) + sp1.js, sp1, L.fac01, L.fac11);
            ADD_SELF(); ADD_FOR();
            pf_interp<R, NG>(pf, B.fracrefa, spl);
        } else {
            // reference quirk kept: nspb(16) = 0 (rrtmg_lw_init.F90:195) so ind0 = ind1 = 1 (:3110-3111)
            major1<R, NG>(tau, B.absb, 1, 1, L, colch4);
            pf_const<R, NG>(pf, B.fracrefb);
        }
    }
};

#undef BAND_HEAD
#undef ADD_SELF
#undef ADD_FOR

// ---------------------------------------------------------------------------------------------------
// Fused taumol + rtrnmc for one (column, band): LW/rrtmg_lw_rtrnmc.F90:164-388.
// ---------------------------------------------------------------------------------------------------
template <typename R> GR_DEV void load_layer(const LwArgs<R> &A, int lay, int col, Layer<R> &L)
{
    const size_t w = (size_t)lay * A.ncol + col;
    const size_t fs = (size_t)A.nlay * A.ncol;
    const R *sc = A.sc + w;
    L.fac00 = sc[SC_FAC00 * fs]; L.fac01 = sc[SC_FAC01 * fs]; L.fac10 = sc[SC_FAC10 * fs]; L.fac11 = sc[SC_FAC11 * fs];
    L.coldry = sc[SC_COLDRY * fs]; L.forfac = sc[SC_FORFAC * fs]; L.forfrac = sc[SC_FORFRAC * fs];
    L.selffac = sc[SC_SELFFAC * fs]; L.selffrac = sc[SC_SELFFRAC * fs]; L.minorfrac = sc[SC_MINORFRAC * fs];
    L.scaleminor = sc[SC_SCALEMINOR * fs]; L.scaleminorn2 = sc[SC_SCALEMINORN2 * fs]; L.colbrd = sc[SC_COLBRD * fs];
    const uint32_t p = A.scidx[w];
    L.jp = p & 63; L.jt = (p >> 6) & 7; L.jt1 = (p >> 9) & 7; L.indfor = (p >> 12) & 3; L.indself = (p >> 14) & 15;
    L.indminor = (p >> 18) & 31; L.lower = (p >> 23) & 1;
    L.i = (size_t)lay * A.ld + col;
    L.pavel = A.play[L.i];
}

// Planck function of band IB at temperature t by linear interpolation in totplnk(181,16)
// (LW/rrtmg_lw_setcoef.F90:300-343)
template <typename R> GR_DEV R planck_at(const R *__restrict__ totplnk, int ib, R t)
{
    const int ind = clampi((int)(t - (R)159.), 1, 180);
    const R frac = t - (R)159. - (R)ind;
    const R *p = totplnk + (size_t)(ib - 1) * 181 + (ind - 1);
    const R d = p[1] - p[0];
    return p[0] + frac * d;
}

template <typename R, typename BAND>
GR_DEV void band_body(const LwArgs<R> &A, const LwDev<R> &T, int col)
{
    constexpr int NG = BAND::NG, IB = BAND::IB, G0 = BAND::G0;
    using R2 = typename Vec2<R>::T;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool dudTs = A.dudTs != 0;
    const R bpade = T.bpade, tblint = (R)NTBL;
    const R sumfac = (R)0.5 * T.delwave[IB] * T.fluxfac;

    // diffusivity angle (:177-186)
    R secdiff = (R)1.66;
    if (!(IB == 1 || IB == 4 || IB >= 10)) {
        constexpr double a0[17] = {0, 1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
        constexpr double a1[17] = {0, 0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
        constexpr double a2[17] = {0, 0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
        secdiff = (R)a0[IB] + (R)a1[IB] * gr_exp<R>((R)a2[IB] * A.pwvcm[col]);
        secdiff = secdiff > (R)1.80 ? (R)1.80 : (secdiff < (R)1.50 ? (R)1.50 : secdiff);
    }
    const bool ccol = A.colcloudy[col] != 0;
    const size_t cell0 = (size_t)G0 * nlay * n + col;   // + (g*nlay + lay)*n
    R *part = A.part + (size_t)(IB - 1) * (nlay + 1) * n + col;   // + (q*16*(nlay+1) + lev)*n
    const size_t qs = (size_t)NB_LW * (nlay + 1) * n;

    R radld[NG], radclrd[NG], pfsfc[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) { radld[g] = 0; radclrd[g] = 0; pfsfc[g] = 0; }
    bool diverge = false;
    int ltop = -1;   // highest optically cloudy layer: where the clear/total streams part (:297-307)

    // ---- downward sweep, top layer -> surface ------------------------------------------------------
    R plk_up = planck_at<R>(T.totplnk, IB, A.tlev[(size_t)nlay * ld + col]);   // level above the current layer
    for (int lay = nlay - 1; lay >= 0; lay--) {
        Layer<R> L;
        load_layer<R>(A, lay, col, L);
        R tau[NG], pf[NG];
        BAND::template tau_pf<R>(T, A, L, tau, pf);
        const R ta = A.tauaer ? A.tauaer[((size_t)(IB - 1) * nlay + lay) * ld + col] : (R)0;
        const R blay = planck_at<R>(T.totplnk, IB, A.tlay[L.i]);
        const R plk_dn = planck_at<R>(T.totplnk, IB, A.tlev[L.i]);
        const R dplankup = plk_up - blay, dplankdn = plk_dn - blay;
        plk_up = plk_dn;
        if (A.dbg_taug) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const size_t o = ((size_t)col * NG_LW + (G0 + g)) * nlay + lay;   // Fortran (nlay,140,ncol)
                A.dbg_taug[o] = tau[g] + ta;
                A.dbg_pfracs[o] = pf[g];
            }
        }
        const bool laycld = ccol && A.laycloudy[(size_t)lay * n + col] != 0;
        if (laycld && !diverge) { diverge = true; ltop = lay; }   // (:297-299) set before the clear-sky update of this layer
        R dsum = 0, dcsum = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            R odepth = secdiff * (tau[g] + ta);
            if (odepth < 0) odepth = 0;
            const R tblind = odepth / (bpade + odepth);
            const int itgas = (int)(tblint * tblind + (R)0.5);
            const R2 e = T.lut[itgas];
            const R agas = (R)1. - e.x, tfacgas = e.y;
            const R bbdgas = pf[g] * (blay + tfacgas * dplankdn);
            const R bbugas = pf[g] * (blay + tfacgas * dplankup);
            const size_t c = cell0 + ((size_t)g * nlay + lay) * n;
            R atot = agas, bbutot = bbugas;
            const R radprev = radld[g];
            bool cldcell = false;
            if (laycld) {
                const R tc = A.taucmc[c];
                if (tc > 0) {
                    cldcell = true;
                    const R odcld = secdiff * tc;
                    const R odtot = T.tau_tbl[itgas] + odcld;   // add cloud to the DISCRETISED gas tau (:264-268)
                    const R tb2 = odtot / (bpade + odtot);
                    const int ittot = (int)(tblint * tb2 + (R)0.5);
                    const R2 e2 = T.lut[ittot];
                    atot = (R)1. - e2.x;
                    const R bbdtot = pf[g] * (blay + e2.y * dplankdn);
                    bbutot = pf[g] * (blay + e2.y * dplankup);
                    radld[g] = radprev + (bbdtot - radprev) * atot;
                }
            }
            if (!cldcell) radld[g] = radprev + (bbdgas - radprev) * agas;
            R2 s; s.x = atot; s.y = bbutot;
            A.s1[c] = s;
            dsum = dsum + sumfac * radld[g];
            if (ccol) {
                if (diverge) {
                    radclrd[g] = radclrd[g] + (bbdgas - radclrd[g]) * agas;
                    R2 sg; sg.x = agas; sg.y = bbugas;
                    A.s2[c] = sg;
                } else {
                    radclrd[g] = radld[g];
                }
                dcsum = dcsum + sumfac * radclrd[g];
            }
            if (lay == 0) pfsfc[g] = pf[g];
        }
        part[(0 * qs) + (size_t)lay * n] = dsum;
        if (ccol) part[(1 * qs) + (size_t)lay * n] = dcsum;
    }
    // TOA downward flux is zero (level nlay); written so the reduce kernel can sum unconditionally
    part[(0 * qs) + (size_t)nlay * n] = 0;
    if (ccol) part[(1 * qs) + (size_t)nlay * n] = 0;

    // ---- surface (:319-333): emission + reflection --------------------------------------------------
    const R semis = A.emis[(size_t)(IB - 1) * ld + col];
    const R tb = A.tsfc[col];
    const R plankbnd = semis * planck_at<R>(T.totplnk, IB, tb);
    const R dplankbnd = dudTs ? semis * planck_at<R>(T.totplnkderiv, IB, tb) : (R)0;
    const R reflect = (R)1. - semis;
    R radlu[NG], radclru[NG], dlu[NG], dclru[NG];
    R usum = 0, ucsum = 0, dusum = 0, ducsum = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const R rad0 = pfsfc[g] * plankbnd;
        radlu[g] = rad0 + reflect * radld[g];
        radclru[g] = rad0 + reflect * radclrd[g];
        usum = usum + sumfac * radlu[g];
        ucsum = ucsum + sumfac * radclru[g];
        dlu[g] = pfsfc[g] * dplankbnd;
        dclru[g] = dlu[g];
        dusum = dusum + sumfac * dlu[g];
        ducsum = ducsum + sumfac * dclru[g];
    }
    part[2 * qs] = usum;
    if (ccol) part[3 * qs] = ucsum;
    if (dudTs) { part[4 * qs] = dusum; if (ccol) part[5 * qs] = ducsum; }

    // ---- upward sweep, surface -> top (:336-379) -----------------------------------------------------
    for (int lay = 0; lay < nlay; lay++) {
        usum = 0; ucsum = 0; dusum = 0; ducsum = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const size_t c = cell0 + ((size_t)g * nlay + lay) * n;
            const R2 s = A.s1[c];
            radlu[g] = radlu[g] + (s.y - radlu[g]) * s.x;
            dlu[g] = dlu[g] - dlu[g] * s.x;
            usum = usum + sumfac * radlu[g];
            dusum = dusum + sumfac * dlu[g];
            if (ccol) {
                if (diverge) {
                    const R2 sg = (lay <= ltop) ? A.s2[c] : s;   // above ltop the layer is clear: gas == total
                    radclru[g] = radclru[g] + (sg.y - radclru[g]) * sg.x;
                    dclru[g] = dclru[g] - dclru[g] * sg.x;
                } else {
                    radclru[g] = radlu[g];
                    dclru[g] = dlu[g];
                }
                ucsum = ucsum + sumfac * radclru[g];
                ducsum = ducsum + sumfac * dclru[g];
            }
        }
        const size_t o = (size_t)(lay + 1) * n;
        part[2 * qs + o] = usum;
        if (ccol) part[3 * qs + o] = ucsum;
        if (dudTs) { part[4 * qs + o] = dusum; if (ccol) part[5 * qs + o] = ducsum; }
    }
}

// band launch order: heaviest first (binary-species bands with 16/14/12 g-points, then the rest)
__constant__ const int LW_BAND_ORDER[NB_LW] = {3, 5, 4, 7, 9, 2, 1, 8, 12, 6, 11, 13, 10, 15, 16, 14};

template <typename R>
__global__ void __launch_bounds__(256) k_lw_bands(LwArgs<R> A, const LwDev<R> *__restrict__ Tp)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const LwDev<R> &T = *Tp;
    switch (LW_BAND_ORDER[blockIdx.y]) {
        case 1: band_body<R, Band1>(A, T, col); break;
        case 2: band_body<R, Band2>(A, T, col); break;
        case 3: band_body<R, Band3>(A, T, col); break;
        case 4: band_body<R, Band4>(A, T, col); break;
        case 5: band_body<R, Band5>(A, T, col); break;
        case 6: band_body<R, Band6>(A, T, col); break;
        case 7: band_body<R, Band7>(A, T, col); break;
        case 8: band_body<R, Band8>(A, T, col); break;
        case 9: band_body<R, Band9>(A, T, col); break;
        case 10: band_body<R, Band10>(A, T, col); break;
        case 11: band_body<R, Band11>(A, T, col); break;
        case 12: band_body<R, Band12>(A, T, col); break;
        case 13: band_body<R, Band13>(A, T, col); break;
        case 14: band_body<R, Band14>(A, T, col); break;
        case 15: band_body<R, Band15>(A, T, col); break;
        default: band_body<R, Band16>(A, T, col); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_lw_reduce: one thread per (level, column): sum the 16 band partials in band order and write the
// API outputs (LW/rrtmg_lw_rad.F90:587-605).  olrb/dolrb (16,ncol) from the TOA partials (:382-385).
// ---------------------------------------------------------------------------------------------------
template <typename R> struct LwOut {
    R *uflx, *dflx, *uflxc, *dflxc, *duflx_dTs, *duflxc_dTs, *olrb, *dolrb_dTs;
    int band_output[NB_LW];
    long col0;   // global index of the batch's first column (for olrb)
};

template <typename R>
__global__ void __launch_bounds__(256) k_lw_reduce(LwArgs<R> A, LwOut<R> O)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lev = blockIdx.y;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool ccol = A.colcloudy[col] != 0;
    const size_t qs = (size_t)NB_LW * (nlay + 1) * n;
    const R *p = A.part + (size_t)lev * n + col;
    R s[6] = {0, 0, 0, 0, 0, 0};
    for (int ib = 0; ib < NB_LW; ib++) {
        const size_t o = (size_t)ib * (nlay + 1) * n;
        s[0] += p[0 * qs + o];
        s[2] += p[2 * qs + o];
        if (A.dudTs) s[4] += p[4 * qs + o];
        if (ccol) {
            s[1] += p[1 * qs + o];
            s[3] += p[3 * qs + o];
            if (A.dudTs) s[5] += p[5 * qs + o];
        }
    }
    if (!ccol) { s[1] = s[0]; s[3] = s[2]; s[5] = s[4]; }
    const size_t i = (size_t)lev * ld + col;
    O.dflx[i] = s[0]; O.dflxc[i] = s[1]; O.uflx[i] = s[2]; O.uflxc[i] = s[3];
    if (A.dudTs) { O.duflx_dTs[i] = s[4]; O.duflxc_dTs[i] = s[5]; }
    if (lev == nlay) {
        for (int ib = 0; ib < NB_LW; ib++) {
            if (O.band_output[ib]) {
                const size_t o = (size_t)ib * (nlay + 1) * n;
                O.olrb[(size_t)(O.col0 + col) * NB_LW + ib] = p[2 * qs + o];
                if (A.dudTs) O.dolrb_dTs[(size_t)(O.col0 + col) * NB_LW + ib] = p[4 * qs + o];
            }
        }
    }
}

}  // namespace geosrad
