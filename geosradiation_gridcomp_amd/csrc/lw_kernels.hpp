// lw_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the RRTMG_LW column solver.
//
// What is computed follows the reference (file:line cited at each kernel; LW = GEOSirrad_GridComp/RRTMG/
// rrtmg_lw/gcm_model/src); how it is computed is ours:
//   lane = column, all HBM traffic coalesced over the column dimension, no cross-lane traffic, LDS only as a
//   read-only copy of the transmittance table, no MFMA (the path is table interpolation + first-order vertical
//   recurrences).  (The other mapping - lanes = layers, intermediates in LDS - is k_lw_cols, lw_cols_kernels.hpp.)
//   k_validate_pwv : per column  - input checks, precipitable water, "any cloud" flag
//   k_setcoef      : per (layer,column) - p/T interpolation record shared by all 16 bands
//   k_lw_bands     : per (256-column block, band) (band_block: XCD-aware grid): fused taumol -> rtrnmc; down sweep
//                    keeps the band's g-point radiances in registers and evaluates the k-distribution 4 or 8
//                    g-points at a time, parking the 2-byte table index of every cell in HBM; the up sweep
//                    re-forms (absorptivity, source) from the parked indices
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

// Double-precision division / reciprocal / square root for operands in the normal range (the optics: optical depths, albedos, cosines):
// v_rcp_f64 / v_rsq_f64 refined by Newton steps and one residual correction - the sequence the IEEE expansion wraps in v_div_scale /
// v_div_fmas / v_div_fixup (12 -> 8 instructions) and the library sqrt in range scaling and class tests (18 -> 9).  Not correctly rounded:
// <= 1 ulp (tests/test_gpu_fastmath.py), against a parity bound of 1e-6 W m-2 that the fp64 instantiations meet at ~1e-9.
#ifndef GR_FAST64
#define GR_FAST64 1
#endif
GR_DEV double gr_rcp64(double b)
{
#if GR_FAST64
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
#else
    return 1.0 / b;
#endif
}
GR_DEV double gr_div64(double a, double b)
{
#if GR_FAST64
    const double r = gr_rcp64(b), q = a * r;
    return fma(fma(-b, q, a), r, q);
#else
    return a / b;
#endif
}
GR_DEV double gr_sqrt64(double x)
{
#if GR_FAST64
    if (x == 0.0) return 0.0;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return fma(fma(-g, g, x), h, g);
#else
    return sqrt(x);
#endif
}

GR_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

constexpr int pad4(int n) { return (n + 3) & ~3; }

// Addressing idiom of the hot kernels:  wave-uniform base pointer (SGPR pair)  +  32-bit per-lane BYTE offset.
// Written on bytes so the compiler sees base + zext(u32) and emits `global_load v, voff, s[base:base+1]`
// instead of materialising a 64-bit per-lane address (2 VGPRs + 64-bit VALU adds) for every array.
template <typename T> GR_DEV T ldg(const T *base, uint32_t byteoff)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byteoff);
}
template <typename T> GR_DEV void stg(T *base, uint32_t byteoff, T v)
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byteoff) = v;
}
// streaming variants for scratch that is written once and read once much later (parked cells): non-temporal, so that it does not
// displace the k-distribution rows and look-up tables from the caches
template <typename T> GR_DEV T ldg_nt(const T *base, uint32_t byteoff)
{
    return __builtin_nontemporal_load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byteoff));
}
template <typename T> GR_DEV void stg_nt(T *base, uint32_t byteoff, T v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byteoff));
}
typedef float gr_v2f __attribute__((ext_vector_type(2)));
typedef double gr_v2d __attribute__((ext_vector_type(2)));
template <> GR_DEV float2 ldg_nt<float2>(const float2 *base, uint32_t byteoff)
{
    const gr_v2f t = __builtin_nontemporal_load(reinterpret_cast<const gr_v2f *>(reinterpret_cast<const char *>(base) + byteoff));
    float2 r; r.x = t.x; r.y = t.y; return r;
}
template <> GR_DEV double2 ldg_nt<double2>(const double2 *base, uint32_t byteoff)
{
    const gr_v2d t = __builtin_nontemporal_load(reinterpret_cast<const gr_v2d *>(reinterpret_cast<const char *>(base) + byteoff));
    double2 r; r.x = t.x; r.y = t.y; return r;
}
template <> GR_DEV void stg_nt<float2>(float2 *base, uint32_t byteoff, float2 v)
{
    gr_v2f t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<gr_v2f *>(reinterpret_cast<char *>(base) + byteoff));
}
template <> GR_DEV void stg_nt<double2>(double2 *base, uint32_t byteoff, double2 v)
{
    gr_v2d t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<gr_v2d *>(reinterpret_cast<char *>(base) + byteoff));
}

// ---------------------------------------------------------------------------------------------------
// k_validate_pwv: one thread per column.
//   - the reference's input assertions (LW/rrtmg_lw_rad.F90:209-318) -> error bits
//   - pwvcm (LW/rrtmg_lw_setcoef.F90:206-272), same summation order (bottom-up)
//   - colcloudy = 1 + the highest layer with cldf > 0 (0: none; nlay <= mxlay = 203 fits a byte): lets later kernels skip McICA
//     work for clear columns, and the generator stop above a wave's highest cloud, without changing results (all
//     masks false, clearCounts = ngpt)
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
    int cftop = 0;
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
        if (A.cldf[i] > 0) cftop = lay + 1;
        // pressure ordering (LW/rrtmg_lw_setcoef.F90:443-453): lower-atmosphere layer above an upper one
        pprev = pup;
    }
    if (A.tsfc[col] < 0) err |= 1u << 18;
    for (int ib = 0; ib < NB_LW; ib++)
        if (A.emis[(size_t)ib * ld + col] < 0) err |= 1u << 19;
    const R wvsh = (amw * wvttl) / (amd * amttl);
    A.pwvcm[col] = wvsh * ((R)1.e3 * A.plev[col]) / ((R)1.e2 * grav);
    const bool cloudy = cftop > 0;
    A.colcloudy[col] = (uint8_t)cftop;
    // clear column: all sub-columns clear in every super-layer (cloud_subcol_gen.F90:649-659);
    // cloudy column: k_mcica's (column, band) threads add their counts
    for (int k = 0; k < 4; k++) A.clearCounts[(size_t)k * ld + col] = cloudy ? 0 : NG_LW;
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
// k_partition: stable partition of the batch's columns into clear | cloudy (one 1024-thread block).
// Every later kernel works on COMPACTED positions: workspace arrays are indexed by the position, API arrays
// by perm[position].  256-column blocks are then homogeneous (at most one mixed block), so clear blocks run the
// cheaper clear-sky instantiation whatever the spatial distribution of the cloudy columns, and no wave
// carries idle cloudy-only work for its clear lanes.  Columns are independent: the permutation does not touch results.
// (Ordering the cloudy columns by cloud top as well, so that the per-wave decisions of the cloudy instantiations - the generator's
// walk up to the wave's highest cloud, the band kernels' cloud path per layer - see similar neighbours, was measured both batch-wide
// and inside 256-column blocks: k_mcica gains 0.45 ms batch-wide, but every array read in the caller's column order then costs
// more cache lines per wave and k_sw_reform loses 0.2-0.25 ms; the step does not move - profiles/r03_mcica_cloudtop.md.)
// ---------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(1024) k_partition(int ncol, const uint8_t *__restrict__ colcloudy, int32_t *__restrict__ perm,
                                                    int32_t *__restrict__ nclear)
{
    __shared__ int cnt[1024];
    const int t = threadIdx.x;
    const int chunk = (ncol + 1023) / 1024;
    const int b = t * chunk, e = (b + chunk < ncol) ? b + chunk : ncol;
    int c = 0;
    for (int i = b; i < e; i++) c += colcloudy[i] != 0;
    cnt[t] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // inclusive Hillis-Steele scan
        const int v = t >= d ? cnt[t - d] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    const int ncloudy = cnt[1023];
    const int ncl = ncol - ncloudy;
    int cbefore = cnt[t] - c;                       // cloudy columns before this thread's chunk
    for (int i = b; i < e; i++) {
        if (colcloudy[i]) { perm[ncl + cbefore] = i; cbefore++; }
        else perm[i - cbefore] = i;
    }
    if (t == 0) *nclear = ncl;
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
    const size_t i = (size_t)lay * ld + A.perm[col];          // API arrays: original column; workspace: compacted position
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
// band bodies (LW/rrtmg_lw_taumol.F90:155-3126, one struct per taugbN).
//
// Each band is split into
//   prep():  everything of a (layer, column) that does not depend on the g-point: column amounts,
//            binary-species parameters, "too much of a minor gas" adjustments, table row numbers;
//   eval<W>(): gas optical depth tau[] and Planck fraction pf[] of W consecutive g-points starting at
//            `go`, as a short list of  coefficient x table-row  products, each row fetched with one
//            16-byte load per lane.
// The band kernel walks a band's g-points W = 4 at a time with a scheduling barrier between groups, so
// only one group's rows are in flight: that keeps the VGPR count low enough for several waves per SIMD
// (the first version evaluated all 16 g-points at once and needed 256 VGPRs + AGPR spills).
// ---------------------------------------------------------------------------------------------------
template <typename R> struct Layer {
    R fac00, fac01, fac10, fac11, coldry, forfac, forfrac, selffac, selffrac, minorfrac, scaleminor, scaleminorn2,
        colbrd, pavel;
    int jp, jt, jt1, indfor, indself, indminor;
    bool lower;
    uint32_t ab;  // BYTE offset of this (layer, column) cell in an API array of reals: (lay*ld + col)*sizeof(R)
};

template <typename R> GR_DEV R colamt(const R *__restrict__ vmr, const Layer<R> &L)
{
    return (R)1.e-20 * ldg(vmr, L.ab) * L.coldry;
}
// "require some minor absorbers to be non-zero" (LW/rrtmg_lw_setcoef.F90:560-564)
template <typename R> GR_DEV R colamt_nz(const R *__restrict__ vmr, const Layer<R> &L)
{
    R c = (R)1.e-20 * ldg(vmr, L.ab) * L.coldry;
    return c == (R)0 ? (R)1.e-32 * L.coldry : c;
}

// W consecutive reals of a table row (W = 2 or a multiple of 4), 16-byte aligned: uniform table base + per-lane byte offset,
// 16 bytes per load instruction (the later pieces of a row at immediate offsets from the same address register)
template <typename R, int W> GR_DEV void ldw(const R *__restrict__ tab, uint32_t byteoff, R (&o)[W])
{
    if constexpr (sizeof(R) == 4 && W == 2) {
        const float2 v = ldg(reinterpret_cast<const float2 *>(tab), byteoff);
        o[0] = v.x; o[1] = v.y;
    } else if constexpr (sizeof(R) == 4) {
        static_assert(W % 4 == 0, "row pieces are whole 16-byte groups");
#pragma unroll
        for (int k = 0; k < W / 4; k++) {
            const float4 v = ldg(reinterpret_cast<const float4 *>(tab), byteoff + 16u * (uint32_t)k);
            o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        }
    } else {
        static_assert(W % 2 == 0, "row pieces are whole 16-byte groups");
#pragma unroll
        for (int k = 0; k < W / 2; k++) {
            const double2 a = ldg(reinterpret_cast<const double2 *>(tab), byteoff + 16u * (uint32_t)k);
            o[2 * k] = a.x; o[2 * k + 1] = a.y;
        }
    }
}
// byte offset of row r (0-based) of a [rows][S] table, columns go..go+W-1
#define ROWB(r) (((uint32_t)(r) * (uint32_t)S + (uint32_t)go) * (uint32_t)sizeof(R))

template <typename R, int W, int S, bool INIT> GR_DEV void axw(R (&acc)[W], R c, const R *__restrict__ tab, int row, int go, uint32_t byteoff = 0)
{
    R r[W];
    ldw<R, W>(tab, ROWB(row) + byteoff, r);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = INIT ? c * r[j] : acc[j] + c * r[j];
}
// o[j] = a[j] + f (b[j] - a[j]) between rows `row` and `row + 1`
template <typename R, int W, int S> GR_DEV void linw(R (&o)[W], R f, const R *__restrict__ tab, int row, int go)
{
    R a[W], b[W];
    ldw<R, W>(tab, ROWB(row), a);
    ldw<R, W>(tab, ROWB(row + 1), b);
#pragma unroll
    for (int j = 0; j < W; j++) o[j] = a[j] + f * (b[j] - a[j]);
}
template <typename R, int W, int S> GR_DEV void add_linw(R (&acc)[W], R s, R f, const R *__restrict__ tab, int row, int go)
{
    R t[W];
    linw<R, W, S>(t, f, tab, row, go);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = acc[j] + s * t[j];
}
// minor gas on a (species parameter, T) grid, rows [indm][jm], NSP species rows per T (e.g. :546-551)
template <typename R, int W, int S, int NSP>
GR_DEV void minor2w(R (&o)[W], const R *__restrict__ tab, int jm, int indm, R fm, R minorfrac, int go)
{
    R m1[W], m2[W];
    linw<R, W, S>(m1, fm, tab, (indm - 1) * NSP + (jm - 1), go);
    linw<R, W, S>(m2, fm, tab, indm * NSP + (jm - 1), go);
#pragma unroll
    for (int j = 0; j < W; j++) o[j] = m1[j] + minorfrac * (m2[j] - m1[j]);
}

// binary-species parameter (e.g. LW/rrtmg_lw_taumol.F90:435-441) with the interpolation weights of the
// species dimension: linear in the interior, cubic towards specparm -> 0 / 1 (:482-541)
template <typename R> struct Spec { R speccomb, fs, c0, c1, c2; int js, boff; bool edge; };
template <typename R> GR_DEV Spec<R> spec(R cola, R rat, R colb, R mult, R oneminus)
{
    Spec<R> s;
    s.speccomb = cola + rat * colb;
    R specparm = cola / s.speccomb;
    if (specparm >= oneminus) specparm = oneminus;
    const R sm = mult * specparm;
    const int j = (int)sm;
    s.js = 1 + j;
    s.fs = sm - (R)j;
    const bool lo = specparm < (R)0.125, hi = specparm > (R)0.875;
    s.edge = lo || hi;
    s.boff = hi ? -1 : 0;
    if (s.edge) {
        const R p = lo ? s.fs - (R)1 : -s.fs;
        const R p4 = ((p * p) * p) * p;
        const R fk0 = p4, fk1 = (R)1 - p - (R)2.0 * p4, fk2 = p + p4;
        s.c0 = lo ? fk0 : fk2; s.c1 = fk1; s.c2 = lo ? fk2 : fk0;
    } else {
        s.c0 = (R)1. - s.fs; s.c1 = s.fs; s.c2 = 0;
    }
    return s;
}

// key-species term of one reference-pressure side of a 9-species-row lower-atmosphere band (:553-606);
// ind = 1-based row of (jp|jp+1, jt|jt1, js); (facA, facB) = (fac00, fac10) or (fac01, fac11)
template <typename R, int W, int S, bool INIT>
GR_DEV void major_a(R (&acc)[W], const R *__restrict__ absa, int ind, const Spec<R> &sp, R facA, R facB, int go)
{
    const int base = ind - 1 + sp.boff;
    R t[W];
    axw<R, W, S, true>(t, sp.c0 * facA, absa, base, go);
    axw<R, W, S, false>(t, sp.c1 * facA, absa, base + 1, go);
    if (sp.edge) axw<R, W, S, false>(t, sp.c2 * facA, absa, base + 2, go);
    axw<R, W, S, false>(t, sp.c0 * facB, absa, base + 9, go);
    axw<R, W, S, false>(t, sp.c1 * facB, absa, base + 10, go);
    if (sp.edge) axw<R, W, S, false>(t, sp.c2 * facB, absa, base + 11, go);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = INIT ? sp.speccomb * t[j] : acc[j] + sp.speccomb * t[j];
}
// upper-atmosphere binary side, 5 species rows, always linear (e.g. :706-716)
template <typename R, int W, int S, bool INIT>
GR_DEV void major_b5(R (&acc)[W], const R *__restrict__ absb, int ind, const Spec<R> &sp, R facA, R facB, int go)
{
    R t[W];
    const R c0 = (R)1. - sp.fs, c1 = sp.fs;
    axw<R, W, S, true>(t, c0 * facA, absb, ind - 1, go);
    axw<R, W, S, false>(t, c1 * facA, absb, ind, go);
    axw<R, W, S, false>(t, c0 * facB, absb, ind + 4, go);
    axw<R, W, S, false>(t, c1 * facB, absb, ind + 5, go);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = INIT ? sp.speccomb * t[j] : acc[j] + sp.speccomb * t[j];
}
// single key species: col * 4-point (p,T) interpolation (e.g. :240-244); ind0/ind1 1-based
template <typename R, int W, int S>
GR_DEV void major1(R (&acc)[W], const R *__restrict__ tab, int ind0, int ind1, const Layer<R> &L, R col, int go)
{
    R t[W];
    axw<R, W, S, true>(t, L.fac00, tab, ind0 - 1, go);
    axw<R, W, S, false>(t, L.fac10, tab, ind0, go);
    axw<R, W, S, false>(t, L.fac01, tab, ind1 - 1, go);
    axw<R, W, S, false>(t, L.fac11, tab, ind1, go);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = col * t[j];
}
// "too much of a minor gas" column adjustment (e.g. :461-468)
template <typename R> GR_DEV R adjcol(R colx, R coldry, R chiref, R thresh, R a, R pw)
{
    const R rat = (R)1.e20 * (colx / coldry) / chiref;
    if (rat > thresh) return (a + gr_pow<R>(rat - a, pw)) * chiref * coldry * (R)1.e-20;
    return colx;
}

// Planck fractions of a (layer, column, band): kind 0 none (pf = 0), 1 / 2 fracrefa / fracrefb row 1, 3 / 4 linear between rows js and
// js + 1 of fracrefa / fracrefb with weight fs
template <typename R> struct PfSel { int kind, js; R fs; };

// g-independent state of one (layer, column, band); every band uses a subset
template <typename R> struct Prep {
    Spec<R> sp, sp1, sm, sm2, spl;
    R ca, cb, cc, cd, ce, cf;     // column amounts / scale factors, meaning per band
    int ind0, ind1;
};

#define CHI(m, j) ldg(T.chi_mls, (uint32_t)(((j) - 1) * 7 + ((m) - 1)) * (uint32_t)sizeof(R))
#define RAT(pair, j) ldg(T.rat, (uint32_t)((pair) * 60 + (j)) * (uint32_t)sizeof(R))
#define IND0A(n) (((L.jp - 1) * 5 + (L.jt - 1)) * (n))
#define IND1A(n) ((L.jp * 5 + (L.jt1 - 1)) * (n))
#define IND0B(n) (((L.jp - 13) * 5 + (L.jt - 1)) * (n))
#define IND1B(n) (((L.jp - 12) * 5 + (L.jt1 - 1)) * (n))
#define ADD_SELF() add_linw<R, W, S>(tau, L.selffac, L.selffrac, B.selfref, L.indself - 1, go)
#define ADD_FOR() add_linw<R, W, S>(tau, L.forfac, L.forfrac, B.forref, L.indfor - 1, go)
// (sel, when asked for: WHICH Planck fractions the layer takes - table and interpolation weights; k_lw_cells parks it for k_lw_sweep)
#define PF_CONST(frac) do { ldw<R, W>(frac, (uint32_t)go * (uint32_t)sizeof(R), pf);                                \
                            if (sel) { sel->kind = (frac) == B.fracrefb ? 2 : 1; sel->js = 1; sel->fs = 0; } } while (0)
#define PF_INTERP(frac, s) do { linw<R, W, S>(pf, (s).fs, frac, (s).js - 1, go);                                    \
                                if (sel) { sel->kind = (frac) == B.fracrefb ? 4 : 3; sel->js = (s).js; sel->fs = (s).fs; } } while (0)
#define BAND_DECL(ib, ng, g0)                                                                                   \
    static constexpr int IB = ib, NG = ng, G0 = g0, S = pad4(ng);                                             \
    template <typename R> GR_DEV static void prep(const LwDev<R> &T, const LwArgs<R> &A, const Layer<R> &L, Prep<R> &P)
#define BAND_EVAL()                                                                                             \
    template <typename R, int W>                                                                                \
    GR_DEV static void eval(const LwDev<R> &T, const Layer<R> &L, const Prep<R> &P, int go, R (&tau)[W], R (&pf)[W], PfSel<R> *sel = nullptr)

struct Band1 {  // 10-350 cm-1: h2o; minor n2 (:214-291)
    BAND_DECL(1, 10, 0)
    {
        P.ca = colamt(A.h2o, L);
        P.cb = L.colbrd * L.scaleminorn2;   // scalen2
        if (L.lower) {
            P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1;
            P.cc = L.pavel < (R)250. ? (R)1. - (R)0.15 * ((R)250. - L.pavel) / (R)154.4 : (R)1;
        } else {
            P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1;
            P.cc = (R)1. - (R)0.15 * (L.pavel / (R)95.6);
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[0], L.indminor - 1, go);
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            ADD_FOR();
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[1], L.indminor - 1, go);
            PF_CONST(B.fracrefb);
        }
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = P.cc * tau[j];
    }
};

struct Band2 {  // 350-500: h2o (:296-363)
    BAND_DECL(2, 12, 10)
    {
        P.ca = colamt(A.h2o, L);
        if (L.lower) { P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1; P.cc = (R)1. - (R).05 * (L.pavel - (R)100.) / (R)900.; }
        else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; P.cc = 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = P.cc * tau[j];
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            ADD_FOR();
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band3 {  // 500-630: h2o,co2; minor n2o (:368-727)
    BAND_DECL(3, 16, 22)
    {
        const R colh2o = colamt(A.h2o, L), colco2 = colamt_nz(A.co2, L), coln2o = colamt_nz(A.n2o, L);
        P.ca = adjcol<R>(coln2o, L.coldry, CHI(4, L.jp + 1), (R)1.5, (R)0.5, (R)0.65);
        if (L.lower) {
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 3) / CHI(2, 3), colco2, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 9) / CHI(2, 9), colco2, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else {
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 4, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 4, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 13) / CHI(2, 13), colco2, 4, T.oneminus);   // refrat_m_b == refrat_planck_b
            P.spl = P.sm;
            P.ind0 = IND0B(5) + P.sp.js; P.ind1 = IND1B(5) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        R m[W];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            major_b5<R, W, S, true>(tau, B.absb, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_b5<R, W, S, false>(tau, B.absb, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_FOR();
            minor2w<R, W, S, 5>(m, B.m[1], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
            PF_INTERP(B.fracrefb, P.spl);
        }
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = tau[j] + P.ca * m[j];
    }
};

struct Band4 {  // 630-700: h2o,co2 | o3,co2 (:732-962)
    BAND_DECL(4, 14, 38)
    {
        const R colco2 = colamt_nz(A.co2, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 11) / CHI(2, 11), colco2, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else {
            const R colo3 = colamt_nz(A.o3, L);
            P.sp = spec<R>(colo3, RAT(RAT_O3CO2, L.jp), colco2, 4, T.oneminus);
            P.sp1 = spec<R>(colo3, RAT(RAT_O3CO2, L.jp + 1), colco2, 4, T.oneminus);
            P.spl = spec<R>(colo3, CHI(3, 13) / CHI(2, 13), colco2, 4, T.oneminus);
            P.ind0 = IND0B(5) + P.sp.js; P.ind1 = IND1B(5) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            major_b5<R, W, S, true>(tau, B.absb, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_b5<R, W, S, false>(tau, B.absb, P.ind1, P.sp1, L.fac01, L.fac11, go);
            PF_INTERP(B.fracrefb, P.spl);
            // empirical stratospheric-cooling tweak on g-points 8..14 (:951-957)
            constexpr double f[16] = {1, 1, 1, 1, 1, 1, 1, 0.92, 0.88, 1.07, 1.1, 0.99, 0.88, 0.943, 1, 1};
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = go + j;
                const R s = g == 7 ? (R)f[7] : g == 8 ? (R)f[8] : g == 9 ? (R)f[9] : g == 10 ? (R)f[10] : g == 11 ? (R)f[11] :
                            g == 12 ? (R)f[12] : g == 13 ? (R)f[13] : (R)1;
                if (g >= 7 && g <= 13) tau[j] = tau[j] * s;
            }
        }
    }
};

struct Band5 {  // 700-820: h2o,co2 | o3,co2; minor o3, ccl4 (:967-1229)
    BAND_DECL(5, 16, 52)
    {
        const R colco2 = colamt_nz(A.co2, L);
        P.ca = colamt_nz(A.o3, L);
        P.cb = colamt(A.ccl4, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 7) / CHI(2, 7), colco2, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 5) / CHI(2, 5), colco2, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else {
            P.sp = spec<R>(P.ca, RAT(RAT_O3CO2, L.jp), colco2, 4, T.oneminus);
            P.sp1 = spec<R>(P.ca, RAT(RAT_O3CO2, L.jp + 1), colco2, 4, T.oneminus);
            P.spl = spec<R>(P.ca, CHI(3, 43) / CHI(2, 43), colco2, 4, T.oneminus);
            P.ind0 = IND0B(5) + P.sp.js; P.ind1 = IND1B(5) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        R c4[W];
        ldw<R, W>(B.m[1], (uint32_t)go * (uint32_t)sizeof(R), c4);
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            R m[W];
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + m[j] * P.ca + P.cb * c4[j];
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            major_b5<R, W, S, true>(tau, B.absb, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_b5<R, W, S, false>(tau, B.absb, P.ind1, P.sp1, L.fac01, L.fac11, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cb * c4[j];
            PF_INTERP(B.fracrefb, P.spl);
        }
    }
};

struct Band6 {  // 820-980: h2o; minor co2, cfc11, cfc12 (:1234-1322)
    BAND_DECL(6, 8, 68)
    {
        P.cb = colamt(A.cfc11, L); P.cc = colamt(A.cfc12, L);
        if (L.lower) {
            P.ca = colamt(A.h2o, L);
            P.cd = adjcol<R>(colamt_nz(A.co2, L), L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.77);
            P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        R c11[W], c12[W];
        ldw<R, W>(B.m[1], (uint32_t)go * (uint32_t)sizeof(R), c11);
        ldw<R, W>(B.m[2], (uint32_t)go * (uint32_t)sizeof(R), c12);
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            add_linw<R, W, S>(tau, P.cd, L.minorfrac, B.m[0], L.indminor - 1, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cb * c11[j] + P.cc * c12[j];
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = (R)0.0 + P.cb * c11[j] + P.cc * c12[j];
        }
        PF_CONST(B.fracrefa);
    }
};

struct Band7 {  // 980-1080: h2o,o3 | o3; minor co2 (:1327-1601)
    BAND_DECL(7, 12, 76)
    {
        const R colco2 = colamt_nz(A.co2, L);
        P.ca = colamt_nz(A.o3, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OO3, L.jp), P.ca, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OO3, L.jp + 1), P.ca, 8, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 3) / CHI(3, 3), P.ca, 8, T.oneminus);   // refrat_m_a == refrat_planck_a
            P.cb = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)3.0, (R)0.79);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else {
            P.cb = adjcol<R>(colco2, L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.79);
            P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            R m[W];
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cb * m[j];
            PF_INTERP(B.fracrefa, P.sm);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[1], L.indminor - 1, go);
            PF_CONST(B.fracrefb);
            // (:1591-1596) g-points 6..11
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = go + j;
                const R s = g == 5 ? (R)0.92 : g == 6 ? (R)0.88 : g == 7 ? (R)1.07 : g == 8 ? (R)1.1 : g == 9 ? (R)0.99 :
                            g == 10 ? (R)0.855 : (R)1;
                if (g >= 5 && g <= 10) tau[j] = tau[j] * s;
            }
        }
    }
};

struct Band8 {  // 1080-1180: h2o | o3; minor co2, o3, n2o, cfc12, cfc22 (:1606-1733)
    BAND_DECL(8, 8, 88)
    {
        P.cb = colamt_nz(A.o3, L); P.cc = colamt_nz(A.n2o, L);
        P.cd = colamt(A.cfc12, L); P.ce = colamt(A.cfc22, L);
        P.cf = adjcol<R>(colamt_nz(A.co2, L), L.coldry, CHI(2, L.jp + 1), (R)3.0, (R)2.0, (R)0.65);   // adjcolco2
        if (L.lower) { P.ca = colamt(A.h2o, L); P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1; }
        else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        const R adjco2 = P.cf;
        R c12[W], c22[W];
        ldw<R, W>(B.m[5], (uint32_t)go * (uint32_t)sizeof(R), c12);
        ldw<R, W>(B.m[6], (uint32_t)go * (uint32_t)sizeof(R), c22);
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            add_linw<R, W, S>(tau, adjco2, L.minorfrac, B.m[0], L.indminor - 1, go);
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[2], L.indminor - 1, go);
            add_linw<R, W, S>(tau, P.cc, L.minorfrac, B.m[3], L.indminor - 1, go);
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.cb, go);
            add_linw<R, W, S>(tau, adjco2, L.minorfrac, B.m[1], L.indminor - 1, go);
            add_linw<R, W, S>(tau, P.cc, L.minorfrac, B.m[4], L.indminor - 1, go);
            PF_CONST(B.fracrefb);
        }
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cd * c12[j] + P.ce * c22[j];
    }
};

struct Band9 {  // 1180-1390: h2o,ch4 | ch4; minor n2o (:1738-2001)
    BAND_DECL(9, 12, 96)
    {
        P.ca = colamt_nz(A.ch4, L);
        P.cb = adjcol<R>(colamt_nz(A.n2o, L), L.coldry, CHI(4, L.jp + 1), (R)1.5, (R)0.5, (R)0.65);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp), P.ca, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp + 1), P.ca, 8, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 3) / CHI(6, 3), P.ca, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 9) / CHI(6, 9), P.ca, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            R m[W];
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cb * m[j];
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[1], L.indminor - 1, go);
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band10 {  // 1390-1480: h2o (:2006-2072)
    BAND_DECL(10, 6, 108)
    {
        P.ca = colamt(A.h2o, L);
        if (L.lower) { P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1; } else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            ADD_FOR();
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band11 {  // 1480-1800: h2o; minor o2 (:2077-2160)
    BAND_DECL(11, 8, 114)
    {
        P.ca = colamt(A.h2o, L);
        P.cb = colamt(A.o2, L) * L.scaleminor;   // scaleo2
        if (L.lower) { P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1; } else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[0], L.indminor - 1, go);
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            ADD_FOR();
            add_linw<R, W, S>(tau, P.cb, L.minorfrac, B.m[1], L.indminor - 1, go);
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band12 {  // 1800-2080: h2o,co2 | nothing (:2165-2345)
    BAND_DECL(12, 8, 122)
    {
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L), colco2 = colamt_nz(A.co2, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp), colco2, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 10) / CHI(2, 10), colco2, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            PF_INTERP(B.fracrefa, P.spl);
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) { tau[j] = 0; pf[j] = 0; }
            if (sel) { sel->kind = 0; sel->js = 1; sel->fs = 0; }
        }
    }
};

struct Band13 {  // 2080-2250: h2o,n2o | (o3 minor); minor co2, co (:2350-2585)
    BAND_DECL(13, 4, 130)
    {
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L), coln2o = colamt_nz(A.n2o, L);
            P.ca = (R)1.e-32 * L.coldry;   // colco: covmr == 0 in GEOS (LW/rrtmg_lw_rad.F90:518, setcoef :564)
            P.sp = spec<R>(colh2o, RAT(RAT_H2ON2O, L.jp), coln2o, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2ON2O, L.jp + 1), coln2o, 8, T.oneminus);
            P.sm = spec<R>(colh2o, CHI(1, 1) / CHI(4, 1), coln2o, 8, T.oneminus);     // co2 minor
            P.sm2 = spec<R>(colh2o, CHI(1, 3) / CHI(4, 3), coln2o, 8, T.oneminus);    // co minor
            P.spl = spec<R>(colh2o, CHI(1, 5) / CHI(4, 5), coln2o, 8, T.oneminus);
            P.cb = adjcol<R>(colamt_nz(A.co2, L), L.coldry, (R)3.55e-4, (R)3.0, (R)2.0, (R)0.68);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        } else {
            P.ca = colamt_nz(A.o3, L);
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            R m[W], mc[W];
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
            minor2w<R, W, S, 9>(mc, B.m[1], P.sm2.js, L.indminor, P.sm2.fs, L.minorfrac, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.cb * m[j] + P.ca * mc[j];
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            R m[W];
            linw<R, W, S>(m, L.minorfrac, B.m[2], L.indminor - 1, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = P.ca * m[j];
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band14 {  // 2250-2380: co2 (:2590-2653)
    BAND_DECL(14, 2, 134)
    {
        P.ca = colamt_nz(A.co2, L);
        if (L.lower) { P.ind0 = IND0A(1) + 1; P.ind1 = IND1A(1) + 1; } else { P.ind0 = IND0B(1) + 1; P.ind1 = IND1B(1) + 1; }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major1<R, W, S>(tau, B.absa, P.ind0, P.ind1, L, P.ca, go);
            ADD_SELF(); ADD_FOR();
            PF_CONST(B.fracrefa);
        } else {
            major1<R, W, S>(tau, B.absb, P.ind0, P.ind1, L, P.ca, go);
            PF_CONST(B.fracrefb);
        }
    }
};

struct Band15 {  // 2380-2600: n2o,co2 | nothing; minor n2 (:2658-2866)
    BAND_DECL(15, 2, 136)
    {
        if (L.lower) {
            const R coln2o = colamt_nz(A.n2o, L), colco2 = colamt_nz(A.co2, L);
            P.sp = spec<R>(coln2o, RAT(RAT_N2OCO2, L.jp), colco2, 8, T.oneminus);
            P.sp1 = spec<R>(coln2o, RAT(RAT_N2OCO2, L.jp + 1), colco2, 8, T.oneminus);
            P.sm = spec<R>(coln2o, CHI(4, 1) / CHI(2, 1), colco2, 8, T.oneminus);   // refrat_m_a == refrat_planck_a
            P.ca = L.colbrd * L.scaleminor;   // scalen2
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            R m[W];
            minor2w<R, W, S, 9>(m, B.m[0], P.sm.js, L.indminor, P.sm.fs, L.minorfrac, go);
#pragma unroll
            for (int j = 0; j < W; j++) tau[j] = tau[j] + P.ca * m[j];
            PF_INTERP(B.fracrefa, P.sm);
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) { tau[j] = 0; pf[j] = 0; }
            if (sel) { sel->kind = 0; sel->js = 1; sel->fs = 0; }
        }
    }
};

struct Band16 {  // 2600-3250: h2o,ch4 | ch4 (:2871-3126)
    BAND_DECL(16, 2, 138)
    {
        P.ca = colamt_nz(A.ch4, L);
        if (L.lower) {
            const R colh2o = colamt(A.h2o, L);
            P.sp = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp), P.ca, 8, T.oneminus);
            P.sp1 = spec<R>(colh2o, RAT(RAT_H2OCH4, L.jp + 1), P.ca, 8, T.oneminus);
            P.spl = spec<R>(colh2o, CHI(1, 6) / CHI(6, 6), P.ca, 8, T.oneminus);
            P.ind0 = IND0A(9) + P.sp.js; P.ind1 = IND1A(9) + P.sp1.js;
        }
    }
    BAND_EVAL()
    {
        const BandTab<R> &B = T.b[IB];
        if (L.lower) {
            major_a<R, W, S, true>(tau, B.absa, P.ind0, P.sp, L.fac00, L.fac10, go);
            major_a<R, W, S, false>(tau, B.absa, P.ind1, P.sp1, L.fac01, L.fac11, go);
            ADD_SELF(); ADD_FOR();
            PF_INTERP(B.fracrefa, P.spl);
        } else {
            // reference quirk kept: nspb(16) = 0 (rrtmg_lw_init.F90:195) so ind0 = ind1 = 1 (:3110-3111)
            major1<R, W, S>(tau, B.absb, 1, 1, L, P.ca, go);
            PF_CONST(B.fracrefb);
        }
    }
};

#undef BAND_DECL
#undef BAND_EVAL
#undef ADD_SELF
#undef ADD_FOR
#undef PF_CONST
#undef PF_INTERP
#undef ROWB

// ---------------------------------------------------------------------------------------------------
// Fused taumol + rtrnmc for one (column, band): LW/rrtmg_lw_rtrnmc.F90:164-388.
// ---------------------------------------------------------------------------------------------------
template <typename R> GR_DEV void load_layer(const LwArgs<R> &A, int lay, int col, int pc, Layer<R> &L)
{
    // uniform (SGPR) field base + one shared 32-bit per-lane byte offset
    const uint32_t cell = (uint32_t)lay * (uint32_t)A.ncol + (uint32_t)col;
    const uint32_t wb = cell * (uint32_t)sizeof(R);
    const size_t fs = (size_t)A.nlay * A.ncol;
#define SCF(f) ldg(A.sc + (size_t)(f) * fs, wb)
    L.fac00 = SCF(SC_FAC00); L.fac01 = SCF(SC_FAC01); L.fac10 = SCF(SC_FAC10); L.fac11 = SCF(SC_FAC11);
    L.coldry = SCF(SC_COLDRY); L.forfac = SCF(SC_FORFAC); L.forfrac = SCF(SC_FORFRAC);
    L.selffac = SCF(SC_SELFFAC); L.selffrac = SCF(SC_SELFFRAC); L.minorfrac = SCF(SC_MINORFRAC);
    L.scaleminor = SCF(SC_SCALEMINOR); L.scaleminorn2 = SCF(SC_SCALEMINORN2); L.colbrd = SCF(SC_COLBRD);
#undef SCF
    const uint32_t p = ldg(A.scidx, cell * 4u);
    L.jp = p & 63; L.jt = (p >> 6) & 7; L.jt1 = (p >> 9) & 7; L.indfor = (p >> 12) & 3; L.indself = (p >> 14) & 15;
    L.indminor = (p >> 18) & 31; L.lower = (p >> 23) & 1;
    L.ab = ((uint32_t)lay * (uint32_t)A.ld + (uint32_t)pc) * (uint32_t)sizeof(R);      // API arrays: original column
    L.pavel = ldg(A.play, L.ab);
}

// Planck function of band IB at temperature t by linear interpolation in totplnk(181,16)
// (LW/rrtmg_lw_setcoef.F90:300-343)
template <typename R> GR_DEV R planck_at(const R *__restrict__ totplnk, int ib, R t)
{
    const int ind = clampi((int)(t - (R)159.), 1, 180);
    const R frac = t - (R)159. - (R)ind;
    const uint32_t o = (uint32_t)(ind - 1) * (uint32_t)sizeof(R);
    const R *tb = totplnk + (ib - 1) * 181;          // uniform: ib is a compile-time band number
    const R p0 = ldg(tb, o), p1 = ldg(tb, o + (uint32_t)sizeof(R));
    const R d = p1 - p0;
    return p0 + frac * d;
}

// Workspace layout of the per-cell planes taucmc / s1 / s2: band-major, then [layer][g-in-band][column]:
//   element offset = G0*nlay*n + ((lay*NG + g)*n + col)
// so one band's sub-array stays below 4 GiB and a cell is  uniform band base + 32-bit byte offset.
//
// CLD = false: 256-column block without any cloud (clear == total, one stream);
// CLD = true : general case, per-lane `ccol` predicate, separate clear-sky stream once the streams part.
// DBG = true : additionally dumps taug/pfracs in the reference's (nlay,140,ncol) layout (test hook only).
// the transmittance table lives in LDS for 4-byte reals: 2 blocks x 80 KB fill a CU's 160 KB, which is the 2 waves/SIMD the
// band kernels' register count allows anyway; the 8-byte table (160 KB) stays in L2
template <typename R> struct LwLutInLds { static constexpr bool value = sizeof(R) == 4; };
// behind it (fp32) the band's four small tables - self and foreign continuum, Planck fractions of the lower / upper atmosphere, at most
// (10 + 4 + 9 + 5) rows of 16 reals = 1 792 B: their rows are then LDS reads instead of 5-6 of the ~20 row gathers of a g-group
constexpr size_t LW_LDS_LUT = (((size_t)(NTBL + 1) * 2 * sizeof(float)) + 15) & ~(size_t)15;
constexpr size_t LW_LDS_SMALL = (size_t)(10 + 4 + 9 + 5) * 16 * sizeof(float);
template <typename R> constexpr size_t lw_bands_lds_bytes() { return LwLutInLds<R>::value ? LW_LDS_LUT + LW_LDS_SMALL : 0; }
__host__ __device__ constexpr int lw_nfraca(int ib) { constexpr int n[17] = {0, 1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9}; return n[ib]; }
__host__ __device__ constexpr int lw_nfracb(int ib) { constexpr int n[17] = {0, 1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 1, 1, 0, 1}; return n[ib]; }

// Pade variable of an optical depth, od / (bpade + od) (LW/rrtmg_lw_rtrnmc.F90:264-268): the index into the transmittance table.
// fp32: the hardware reciprocal (1 ulp) instead of the correctly rounded quotient's ten dependent instructions; a quotient within
// ~1e-7 of a rounding boundary of the 10 000-entry table then lands in the neighbouring entry, which evaluation-order differences
// between two fp32 builds of the reference do as well.  Measured at 97 200 columns against the fp64 instantiation (round 4, same call):
// hardware reciprocal median 7.72e-5 / 99 % 1.04e-3 / 99.9 % 2.16e-3 / worst 4.51e-3 W m-2, exact quotient 7.71e-5 / 1.03e-3 / 2.17e-3 / 4.45e-3 -
// the same distribution (tests/test_gpu_fullsize.py::test_lw_fp32_against_fp64_at_full_size holds it); k_lw_bands 5.85-5.94 against
// 6.00-6.02 ms
template <typename R> GR_DEV R lw_pade(R od, R bpade) { return od / (bpade + od); }
#ifndef LW_EXACT_DIV
template <> GR_DEV float lw_pade<float>(float od, float bpade) { return od * __builtin_amdgcn_rcpf(bpade + od); }
#endif

template <typename R, typename BAND, bool CLD, bool DBG, bool WIDE = true>
GR_DEV void band_body(const LwArgs<R> &A, const LwDev<R> &Tg, int col, int nclear, const typename Vec2<R>::T *luts)
{
    constexpr int NG = BAND::NG, IB = BAND::IB, G0 = BAND::G0;
    // g-points per evaluation of the k-distribution: 4 (one 16-byte piece of each table row); 8 in the cloud-free instantiation for the
    // bands whose g-point count is a multiple of 8 (half as many serial table-row round trips per layer, 227 instead of 213 VGPRs:
    // 4.98 -> 4.78 ms per 100 000 clear-sky columns; the cloudy instantiation has no registers for it)
    constexpr int W = (!CLD && WIDE && NG % 8 == 0) ? 8 : (NG >= 4 ? 4 : 2);
    constexpr int NQ = (NG + W - 1) / W;
    using R2 = typename Vec2<R>::T;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    // (1 - transmittance, tfac) of a discretised optical depth: from the block's LDS copy of the table when there is one
    // the band's small tables from the block's LDS copy (k_lw_bands): same layout as in HBM, [rows][S]
    LwDev<R> TL = Tg;
    if constexpr (LwLutInLds<R>::value) {
        constexpr int S = BAND::S;
        const R *sm = reinterpret_cast<const R *>(reinterpret_cast<const unsigned char *>(luts) + LW_LDS_LUT);
        TL.b[IB].selfref = sm; TL.b[IB].forref = sm + 10 * S; TL.b[IB].fracrefa = sm + 14 * S;
        if (lw_nfracb(IB) > 0) TL.b[IB].fracrefb = sm + (14 + lw_nfraca(IB)) * S;
    }
    const LwDev<R> &T = TL;
    auto lut_at = [&](int i) -> R2 {
        if constexpr (LwLutInLds<R>::value) return luts[i];
        else return ldg(T.lut, (uint32_t)i * (uint32_t)sizeof(R2));
    };
    const bool dudTs = A.dudTs != 0;
    const R bpade = T.bpade, tblint = (R)NTBL;
    const R sumfac = (R)0.5 * T.delwave[IB] * T.fluxfac;
    const uint32_t ucol = (uint32_t)col;
    const uint32_t cb = ucol * (uint32_t)sizeof(R);       // byte offset of this position in a workspace row of reals
    const int pc = ldg(A.perm, ucol * 4u);                // original column: index into the API arrays
    const uint32_t cba = (uint32_t)pc * (uint32_t)sizeof(R);

    // diffusivity angle (:177-186)
    R secdiff = (R)1.66;
    if (!(IB == 1 || IB == 4 || IB >= 10)) {
        constexpr double a0[17] = {0, 1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
        constexpr double a1[17] = {0, 0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
        constexpr double a2[17] = {0, 0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
        secdiff = (R)a0[IB] + (R)a1[IB] * gr_exp<R>((R)a2[IB] * ldg(A.pwvcm, cba));
        secdiff = secdiff > (R)1.80 ? (R)1.80 : (secdiff < (R)1.50 ? (R)1.50 : secdiff);
    }
    // CLD kernels only see cloudy columns (DBG: all columns), so this is true there; it is deliberately left a run-time
    // value: as a compile-time constant the if-converted code needs 290 registers and halves the occupancy
    int ncl_opaque = nclear;
    asm volatile("" : "+s"(ncl_opaque));        // hides the fact from the optimiser (the caller already tested col >= nclear)
    const bool ccol = CLD && col >= ncl_opaque;

    // uniform bases
    const size_t bandoff = (size_t)G0 * nlay * n;                         // this band's sub-array of the cell planes
    const R *const taucmc_b = A.taucmc + bandoff;
    // the parked cells are tiled by 256-column block, [block][layer][g][256]: a block's scratch is one contiguous run
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    // what is parked per cell between the sweeps: the Pade index of the cell's discretised optical depth, 2 bytes - the up sweep
    // re-forms (absorptivity, source) from it: table look-up + the layer's Planck terms + the Planck fraction, which is re-evaluated
    // there (a quarter of the bytes of the (a, B-up) pair the first version parked)
    using PK = uint16_t;
    PK *const s1_b = A.s1 + (size_t)G0 * nlay * npad;
    PK *const s2_b = A.s2 + (size_t)G0 * nlay * npad;
    const uint32_t tbase = (ucol >> 8) * (uint32_t)nlay * (uint32_t)NG * 256u + (ucol & 255u);
#define SCELL(lay, g) (tbase + ((uint32_t)(lay) * (uint32_t)NG + (uint32_t)(g)) * 256u)
    const size_t qs = (size_t)NB_LW * (nlay + 1) * n;                     // one flux kind of `part`
    R *const part = A.part + (size_t)(IB - 1) * (nlay + 1) * n;           // [(kind*qs) + lev*n + col]
#define PART(kind, lev, val) stg(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))

    // surface terms (:319-333), needed when the down sweep reaches layer 0
    const R semis = ldg(A.emis + (size_t)(IB - 1) * ld, cba);
    const R tb = ldg(A.tsfc, cba);
    const R plankbnd = semis * planck_at<R>(T.totplnk, IB, tb);
    const R dplankbnd = dudTs ? semis * planck_at<R>(T.totplnkderiv, IB, tb) : (R)0;
    const R reflect = (R)1. - semis;

    // radiances per g-point: downward during the first sweep, converted in place to upward at the surface
    // (the surface step is taken at the start of the upward sweep, where the Planck fractions of layer 0 are evaluated anyway: the
    // derivative radiances dlu / dclu then only live there, not across the downward sweep)
    R rad[NG], radc[NG];                      // radc is dead (eliminated) when CLD == false
#pragma unroll
    for (int g = 0; g < NG; g++) { rad[g] = 0; radc[g] = 0; }
    bool diverge = false;
    int ltop = -1;   // highest optically cloudy layer: where the clear/total streams part (:297-307)
    R usum = 0, ucsum = 0, dusum = 0, ducsum = 0;

    // parked cells of the g-group processed last, waiting to be written (see phase 2 below)
    constexpr bool DEFER = !CLD;       // (the cloudy body defers its stores the same way, in its own branch)
    PK pend1[W], pend2[W];
    uint32_t poff[W];
    int npend = 0;
    uint32_t pmask2 = 0;
    // ---- downward sweep, top layer -> surface ------------------------------------------------------
    R plk_up = planck_at<R>(T.totplnk, IB, ldg(A.tlev + (size_t)nlay * ld, cba));   // level above the current layer
    // the layer's cloud flag is read one layer ahead: it decides (by ballot) whether the layer's McICA optical depths are requested,
    // and behind a load of its own that request would start a memory round trip late
    int lc_next = CLD ? (int)ldg(A.laycloudy, (uint32_t)(nlay - 1) * (uint32_t)n + ucol) : 0;
#pragma nounroll
    for (int lay = nlay - 1; lay >= 0; lay--) {
        const int lc_cur = lc_next;
        if (CLD && lay > 0) lc_next = (int)ldg(A.laycloudy, (uint32_t)(lay - 1) * (uint32_t)n + ucol);
        const bool laycld0 = CLD && ccol && lc_cur != 0;
        const bool wlc0 = CLD && __ballot(laycld0) != 0;
        // McICA optical depths of ALL the layer's g-points, requested with the layer record: one HBM round trip per cloudy layer instead
        // of one per g-group
        R tcall[CLD ? NG : 1];
        if (CLD) {
#pragma unroll
            for (int g = 0; g < NG; g++) tcall[g] = 0;
            if (wlc0) {
#pragma unroll
                for (int g = 0; g < NG; g++)
                    tcall[g] = ldg(taucmc_b, (((uint32_t)lay * (uint32_t)NG + (uint32_t)g) * (uint32_t)n + ucol) * (uint32_t)sizeof(R));
            }
        }
        Layer<R> L;
        load_layer<R>(A, lay, col, pc, L);
        // the layer's aerosol and temperatures are requested, and the Planck look-ups they lead to made, BEFORE the band's prep: behind
        // it they were a memory round trip of their own in every layer (5.12 -> 4.85 ms per 100 000 clear-sky columns)
        const R ta = A.tauaer ? ldg(A.tauaer + (size_t)(IB - 1) * nlay * ld, L.ab) : (R)0;
        const R blay = planck_at<R>(T.totplnk, IB, ldg(A.tlay, L.ab));
        const R plk_dn = planck_at<R>(T.totplnk, IB, ldg(A.tlev, L.ab));
        Prep<R> P;
        BAND::template prep<R>(T, A, L, P);
        const R dplankup = plk_up - blay, dplankdn = plk_dn - blay;
        plk_up = plk_dn;
        bool laycld = false;
        if (CLD) {
            laycld = laycld0;
            if (laycld && !diverge) { diverge = true; ltop = lay; }   // (:297-299) before this layer's clear-sky update
        }
        const uint32_t cell0 = ((uint32_t)lay * (uint32_t)NG) * (uint32_t)n + ucol;   // g = 0 cell of this layer; + g*n
        R dsum = 0, dcsum = 0;
        if constexpr (!CLD) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R tau[W], pf[W];
            BAND::template eval<R, W>(T, L, P, q * W, tau, pf);
            // phase 1: Pade index of the gas optical depth + LUT gathers of the whole group
            int itg[W];
            R2 eg[W];
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) { itg[j] = 0; eg[j].x = 0; eg[j].y = 0; continue; }   // padding of the last group
                if (DBG) {
                    const size_t o = ((size_t)pc * NG_LW + (G0 + g)) * nlay + lay;   // Fortran (nlay,140,ncol)
                    A.dbg_taug[o] = tau[j] + ta;
                    A.dbg_pfracs[o] = pf[j];
                }
                R odepth = secdiff * (tau[j] + ta);
                if (odepth < 0) odepth = 0;
                const R tblind = lw_pade<R>(odepth, bpade);
                itg[j] = (int)(tblint * tblind + (R)0.5);
                eg[j] = lut_at(itg[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: only now write the PREVIOUS group's parked cells.  Loads and stores share one in-order counter
            // (vmcnt) on gfx9: a load issued after a store cannot be consumed before that store is acknowledged, so the
            // stores go behind this group's loads and get a whole group of arithmetic to drain.
            if (npend) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (j < npend) {
                        stg_nt(s1_b, poff[j] * (uint32_t)sizeof(PK), pend1[j]);
                        if (CLD && (pmask2 >> j) & 1u) stg_nt(s2_b, poff[j] * (uint32_t)sizeof(PK), pend2[j]);
                    }
            }
            npend = 0; pmask2 = 0;
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: radiance recurrences of the group
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const int itgas = itg[j];
                const R2 e = eg[j];
                const R agas = (R)1. - e.x, tfacgas = e.y;
                const R bbdgas = pf[j] * (blay + tfacgas * dplankdn);
                const R bbugas = pf[j] * (blay + tfacgas * dplankup);
                const uint32_t cell = cell0 + (uint32_t)g * (uint32_t)n;
                const uint32_t scell = SCELL(lay, g);
                R atot = agas, bbutot = bbugas;
                int itp = itgas;                 // index parked for the total-sky stream
                const R radprev = rad[g];
                bool cldcell = false;
                if (CLD && laycld) {
                    const R tc = ldg(taucmc_b, cell * (uint32_t)sizeof(R));
                    if (tc > 0) {
                        cldcell = true;
                        // cloud added to the DISCRETISED gas tau (:264-268)
                        const R odtot = ldg(T.tau_tbl, (uint32_t)itgas * (uint32_t)sizeof(R)) + secdiff * tc;
                        const R tb2 = lw_pade<R>(odtot, bpade);
                        const int ittot = (int)(tblint * tb2 + (R)0.5);
                        itp = ittot;
                        const R2 e2 = lut_at(ittot);
                        atot = (R)1. - e2.x;
                        const R bbdtot = pf[j] * (blay + e2.y * dplankdn);
                        bbutot = pf[j] * (blay + e2.y * dplankup);
                        rad[g] = radprev + (bbdtot - radprev) * atot;
                    }
                }
                if (!cldcell) rad[g] = radprev + (bbdgas - radprev) * agas;
                if (DEFER) { pend1[j] = (PK)(itp); poff[j] = scell; npend = j + 1; }
                else stg_nt(s1_b, scell * (uint32_t)sizeof(PK), (PK)(itp));
                dsum = dsum + sumfac * rad[g];
                if (CLD && ccol) {
                    if (diverge) {
                        radc[g] = radc[g] + (bbdgas - radc[g]) * agas;
                        if (DEFER) { pend2[j] = (PK)(itgas); pmask2 |= 1u << j; }
                        else stg_nt(s2_b, scell * (uint32_t)sizeof(PK), (PK)(itgas));
                    } else {
                        radc[g] = rad[g];
                    }
                    dcsum = dcsum + sumfac * radc[g];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one g-group's table rows in flight at a time
        }
        } else {
        // general (cloudy-column) body.  Control flow around memory operations is wave-uniform only (ballots): a per-lane branch
        // with a load inside makes the loads of a group wait for one another.  Lanes the test does not concern load / store
        // along and select afterwards (their values are never used: unwritten cloud cells, gas-only pairs above their own cloud top).
        const bool wlc = CLD && __ballot(laycld) != 0;       // some column of the wave has cloud in this layer
        const bool wdv = CLD && __ballot(diverge) != 0;      // some column of the wave keeps gas-only pairs from here down
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R tcv[W];
#pragma unroll
            for (int j = 0; j < W; j++) tcv[j] = 0;
            if (wlc) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (q * W + j < NG) tcv[j] = tcall[q * W + j];
#pragma unroll
                for (int j = 0; j < W; j++) tcv[j] = laycld ? tcv[j] : (R)0;
            }
            R tau[W], pf[W];
            BAND::template eval<R, W>(T, L, P, q * W, tau, pf);
            int itg[W];
            R2 eg[W];
            R ttb[W];
            bool anyc = false;
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                ttb[j] = 0;
                if (g >= NG) { itg[j] = 0; eg[j].x = 0; eg[j].y = 0; continue; }   // padding of the last group
                if (DBG) {
                    const size_t o = ((size_t)pc * NG_LW + (G0 + g)) * nlay + lay;   // Fortran (nlay,140,ncol)
                    A.dbg_taug[o] = tau[j] + ta;
                    A.dbg_pfracs[o] = pf[j];
                }
                R odepth = secdiff * (tau[j] + ta);
                if (odepth < 0) odepth = 0;
                const R tblind = lw_pade<R>(odepth, bpade);
                itg[j] = (int)(tblint * tblind + (R)0.5);
                eg[j] = lut_at(itg[j]);
                anyc = anyc || tcv[j] > 0;
            }
            const bool wcl = wlc && __ballot(anyc) != 0;         // some cell of the group is cloudy in some column of the wave
            if (wcl) {
#pragma unroll
                for (int j = 0; j < W; j++) ttb[j] = ldg(T.tau_tbl, (uint32_t)itg[j] * (uint32_t)sizeof(R));
            }
            // the PREVIOUS group's parked cells are written only now, behind this group's loads (see the cloud-free body)
            __builtin_amdgcn_sched_barrier(0);
            if (npend) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (j < npend) {
                        stg_nt(s1_b, poff[j] * (uint32_t)sizeof(PK), pend1[j]);
                        if ((pmask2 >> j) & 1u) stg_nt(s2_b, poff[j] * (uint32_t)sizeof(PK), pend2[j]);
                    }
            }
            npend = 0; pmask2 = 0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const R agas = (R)1. - eg[j].x, tfacgas = eg[j].y;
                const R bbdgas = pf[j] * (blay + tfacgas * dplankdn);
                const R bbugas = pf[j] * (blay + tfacgas * dplankup);
                const uint32_t scell = SCELL(lay, g);
                R atot = agas, bbutot = bbugas, bbd = bbdgas;
                int itp = itg[j];
                if (wcl) {
                    // cloud added to the DISCRETISED gas tau (:264-268); evaluated for the whole wave, kept for the cloudy cells
                    const bool cld = tcv[j] > 0;
                    const R odtot = ttb[j] + secdiff * tcv[j];
                    const R tb2 = lw_pade<R>(odtot, bpade);
                    const int ittot = (int)(tblint * tb2 + (R)0.5);
                    itp = cld ? ittot : itg[j];
                    const R2 e2 = lut_at(itp);
                    const R ac = (R)1. - e2.x;
                    const R bbdtot = pf[j] * (blay + e2.y * dplankdn), bbut = pf[j] * (blay + e2.y * dplankup);
                    atot = cld ? ac : agas; bbutot = cld ? bbut : bbugas; bbd = cld ? bbdtot : bbdgas;
                }
                rad[g] = rad[g] + (bbd - rad[g]) * atot;
                pend1[j] = (PK)(itp); poff[j] = scell; npend = j + 1;
                dsum = dsum + sumfac * rad[g];
                if (CLD) {
                    const R rc = radc[g] + (bbdgas - radc[g]) * agas;
                    radc[g] = diverge ? rc : rad[g];
                    if (wdv) { pend2[j] = (PK)(itg[j]); pmask2 |= 1u << j; }
                    dcsum = dcsum + sumfac * radc[g];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one g-group's table rows in flight at a time
        }
        }
        PART(0, lay, dsum);
        if (CLD && ccol) PART(1, lay, dcsum);
    }
#pragma unroll
    for (int j = 0; j < W; j++)
        if (j < npend) {
            stg_nt(s1_b, poff[j] * (uint32_t)sizeof(PK), pend1[j]);
            if (CLD && (pmask2 >> j) & 1u) stg_nt(s2_b, poff[j] * (uint32_t)sizeof(PK), pend2[j]);
        }
    // TOA downward flux is zero (level nlay); written so the reduce kernel can sum unconditionally
    PART(0, nlay, 0);
    if (CLD && ccol) PART(1, nlay, 0);

    // ---- upward sweep, surface -> top (:336-379) -----------------------------------------------------
    // highest layer in which a column of this wave keeps a gas-only pair of its own (wave-uniform; no memory access): above it the
    // parked total pairs serve both streams.  All pairs of a layer are requested together, without control flow between them.
    int wtop = -1;
    if (CLD) {
        for (int l = nlay - 1; l >= 0; l--)
            if (__ballot(ccol && diverge && ltop == l) != 0) { wtop = l; break; }
    }
    R dlu[NG], dclu[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) { dlu[g] = 0; dclu[g] = 0; }
#pragma nounroll
    for (int lay = 0; lay < nlay; lay++) {
        R u0 = 0, uc0 = 0, du0 = 0, duc0 = 0;       // level 0 (surface) sums, formed in the first trip
        usum = 0; ucsum = 0; dusum = 0; ducsum = 0;
        PK sv[NG], sg[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) sv[g] = ldg_nt(s1_b, SCELL(lay, g) * (uint32_t)sizeof(PK));
        if (CLD && lay <= wtop) {
            // (lanes that have no pair of their own here read what happens to be there and do not use it)
#pragma unroll
            for (int g = 0; g < NG; g++) sg[g] = ldg_nt(s2_b, SCELL(lay, g) * (uint32_t)sizeof(PK));
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) sg[g] = sv[g];
        }
        const bool own = CLD && ccol && diverge && lay <= ltop;      // above ltop the layer is clear for every g-point: gas == total
        // (absorptivity, upward source) of the cell from its parked index: the transmittance table, the layer's Planck terms and
        // the Planck fraction, which the band body evaluates again (its optical-depth terms are dead code here)
        Layer<R> L;
        load_layer<R>(A, lay, col, pc, L);
        const R blay = planck_at<R>(T.totplnk, IB, ldg(A.tlay, L.ab));
        const R dplankup = planck_at<R>(T.totplnk, IB, ldg(A.tlev + (size_t)ld, L.ab)) - blay;
        Prep<R> P;
        BAND::template prep<R>(T, A, L, P);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R tau[W], pf[W];
            BAND::template eval<R, W>(T, L, P, q * W, tau, pf);
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                if (lay == 0) {
                    // surface: emission + reflection turn the downward radiance into the upward one (:319-333)
                    const R rad0 = pf[j] * plankbnd;
                    rad[g] = rad0 + reflect * rad[g];
                    dlu[g] = pf[j] * dplankbnd;
                    u0 = u0 + sumfac * rad[g];
                    du0 = du0 + sumfac * dlu[g];
                    if (CLD) {
                        radc[g] = rad0 + reflect * radc[g];
                        dclu[g] = dlu[g];
                        uc0 = uc0 + sumfac * radc[g];
                        duc0 = duc0 + sumfac * dclu[g];
                    }
                }
                const R2 e1 = lut_at((int)sv[g]);
                const R a1 = (R)1. - e1.x, b1 = pf[j] * (blay + e1.y * dplankup);
                rad[g] = rad[g] + (b1 - rad[g]) * a1;
                dlu[g] = dlu[g] - dlu[g] * a1;
                usum = usum + sumfac * rad[g];
                dusum = dusum + sumfac * dlu[g];
                if (CLD) {
                    const R2 e2 = lut_at(own ? (int)sg[g] : (int)sv[g]);
                    const R gx = (R)1. - e2.x, gy = pf[j] * (blay + e2.y * dplankup);
                    const R rc = radc[g] + (gy - radc[g]) * gx, dc = dclu[g] - dclu[g] * gx;
                    radc[g] = diverge ? rc : rad[g];
                    dclu[g] = diverge ? dc : dlu[g];
                    ucsum = ucsum + sumfac * radc[g];
                    ducsum = ducsum + sumfac * dclu[g];
                }
            }
        }
        if (lay == 0) {
            PART(2, 0, u0);
            if (CLD && ccol) PART(3, 0, uc0);
            if (dudTs) { PART(4, 0, du0); if (CLD && ccol) PART(5, 0, duc0); }
        }
        PART(2, lay + 1, usum);
        if (CLD && ccol) PART(3, lay + 1, ucsum);
        if (dudTs) { PART(4, lay + 1, dusum); if (CLD && ccol) PART(5, lay + 1, ducsum); }
    }
#undef PART
}

// band launch order: heaviest first (binary-species bands with 16/14/12 g-points, then the rest)
__constant__ const int LW_BAND_ORDER[NB_LW] = {3, 5, 4, 7, 9, 2, 1, 8, 12, 6, 11, 13, 10, 15, 16, 14};
__host__ __device__ constexpr int lw_band_g0(int ib)
{
    constexpr int g0[17] = {0, 0, 10, 22, 38, 52, 68, 76, 88, 96, 108, 114, 122, 130, 134, 136, 138};
    return g0[ib];
}
__host__ __device__ constexpr int lw_band_ng(int ib)
{
    constexpr int ng[17] = {0, 10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
    return ng[ib];
}

// Grid of the band kernels (RRTMG_LW and RRTMG_SW): one block per (256-column block, band), 8 * ceil(gx / 8) * nb blocks in ONE
// dimension.  The hardware deals consecutive blocks to the 8 XCDs in turn; block id therefore takes XCD id % 8, and within that XCD
// the nb bands of a column block are consecutive: they run at about the same time on one L2, so the column block's setcoef record,
// gas arrays and indices - which every band reads - come from HBM once instead of once per band (the first version's
// (column block, band) = (blockIdx.x, blockIdx.y) grid ran all column blocks of a band before the next band's).
// A launch with gridDim.y == nb keeps the (column block, band) = (blockIdx.x, blockIdx.y) order: all column blocks of a band before the
// next band, heaviest bands first - every resident block then runs the same band body on the same tables, which the cloudy RRTMG_LW
// instantiation (the largest code) needs more than it needs the inputs from L2 (measured: 6.7 against 7.7 ms).  Round 3: the cloud-free
// RRTMG_LW instantiation takes this order too - its blocks differ 8x in length (2 to 16 g-points) and with a few thousand of them the
// one-dimensional order left a quarter of the slots empty in the launch's tail (1.45 of 2 wavefronts per SIMD resident); heaviest band
// first over ALL column blocks: step 18.97 -> 18.74 ms (two streams), 20.69 -> 20.35 ms (one); 100 000 clear-sky columns unchanged.
GR_DEV bool band_block(int ncol, int nb, int &bstart, int &bslot)
{
    if (gridDim.y > 1) { bstart = (int)(blockIdx.x * blockDim.x); bslot = (int)blockIdx.y; return true; }
    const unsigned id = blockIdx.x, xcd = id & 7u, slot = id >> 3;
    const unsigned cblk = (slot / (unsigned)nb) * 8u + xcd;
    bslot = (int)(slot % (unsigned)nb);
    bstart = (int)(cblk * blockDim.x);
    return bstart < ncol;
}
__host__ inline unsigned band_grid(int ncol, int nb) { const unsigned gx = (unsigned)((ncol + 255) / 256); return 8u * ((gx + 7u) / 8u) * (unsigned)nb; }
// the same grid for blocks of `bsz` columns and `ny` slots per column block (k_mcica: 64 columns, ny = segments of sub-columns)
__host__ inline unsigned xcd_grid(int ncol, int bsz, int ny) { const unsigned gx = (unsigned)((ncol + bsz - 1) / bsz); return 8u * ((gx + 7u) / 8u) * (unsigned)ny; }

// Two instantiations are launched back to back: CLD = false handles the
// 256-column blocks of compacted positions that hold clear columns only, CLD = true the others (k_partition);
// a block of the wrong kind exits immediately, so each kernel keeps the register budget of its own path.
// T is passed BY VALUE: table pointers that arrive as kernel arguments are known to be global-address-space
// and wave-uniform, so a table row fetch is `global_load_dwordx4 v, voff, s[base:base+1]`; behind a
// pointer-to-struct they degrade to flat loads with a 64-bit per-lane address each.
// BLK: threads per block.  The cloud-free fp32 instantiation exists twice: 256 threads with 8 g-points per k-distribution evaluation
// (181 VGPRs, two blocks = two wavefronts per SIMD on a CU), and 768 threads with 4 (155 VGPRs: one block = THREE wavefronts per
// SIMD sharing one copy of the transmittance table).  Measured (RRTMG_LW alone, cloud-free batches of 35 000 / 55 000 / 100 000 columns):
// 1.96 -> 1.78, 3.05 -> 2.71, 5.44 -> 4.93 ms per call with the wide blocks; in the default mixed batch (38 880 cloud-free columns
// picked out of 97 200: every array read in the caller's column order then touches 2.5x the cache lines, and twelve wavefronts share
// a CU's L1) 2.01 -> 2.15 ms.  Both are launched; the one the batch does not call for returns at once (6 us): wide blocks for a
// batch of at least LW_WIDE_FROM cloud-free columns that is at least four fifths cloud-free.
constexpr int LW_WIDE_BLOCK = 768;
#ifndef LW_WIDE_FROM
#define LW_WIDE_FROM 32768
#endif
GR_DEV bool lw_wide_blocks(int nclear, int ncol) { return nclear >= LW_WIDE_FROM && 5 * (long)nclear >= 4 * (long)ncol; }
// fp64: two wavefronts per SIMD asked for (256 VGPRs: the cloud-free bodies then spill ~100 registers over the 16 bands, the cloudy ones
// ~860) instead of the 292-404 registers of one wavefront per SIMD: 15.6 -> 14.55 ms, 97 200 clear-sky columns 11.4 -> 9.5 ms.  fp32 stays
// unconstrained: asked to fit 2 x 208 registers instead of the 240 it takes it loses (5.99 -> 6.40 ms).
template <typename R, bool CLD, bool DBG, int BLK = 256>
__global__ void __launch_bounds__(BLK, BLK == LW_WIDE_BLOCK ? 3 : (sizeof(R) == 8 ? 2 : 1)) k_lw_bands(LwArgs<R> A, LwDev<R> T)
{
    int bstart, bslot;
    if (!band_block(A.ncol, NB_LW, bstart, bslot)) return;
    if (!((A.band_mask >> LW_BAND_ORDER[bslot]) & 1u)) return;      // a RATS pass re-runs the bands its gas appears in
    const int nclear = *A.nclear;
    if constexpr (!CLD && !DBG && sizeof(R) == 4) {
        if (lw_wide_blocks(nclear, A.ncol) != (BLK == LW_WIDE_BLOCK)) return;
    }
    // every column runs the instantiation of its own class (the one mixed block is visited by both kernels, each
    // masking the other class's lanes): a column's arithmetic never depends on its neighbours -> bitwise column independence
    const int bend = bstart + (int)blockDim.x < A.ncol ? bstart + (int)blockDim.x : A.ncol;
    if (!DBG && (CLD ? bend <= nclear : bstart >= nclear)) return;
    // fp32: the (1 - exp(-tau), tfac) table (80 KB) is copied into LDS once per block - every cell looks it up at an index that
    // differs from lane to lane, which costs up to 64 cache lines per wave from L2 and a few LDS cycles from here
    using R2 = typename Vec2<R>::T;
    extern __shared__ __align__(16) unsigned char lw_lds[];
    const R2 *luts = nullptr;
    if constexpr (LwLutInLds<R>::value) {
        R2 *const l = reinterpret_cast<R2 *>(lw_lds);
        for (int i = threadIdx.x; i <= NTBL; i += (int)blockDim.x) l[i] = ldg(T.lut, (uint32_t)i * (uint32_t)sizeof(R2));
        {   // the band's small tables, [rows][S] each, in the order selfref (10), forref (4), fracrefa, fracrefb
#ifdef LW_ONLY_BAND
            const int ib = LW_ONLY_BAND, S = pad4(lw_band_ng(ib));        // timing experiment: every block runs this band's body
#else
            const int ib = LW_BAND_ORDER[bslot], S = pad4(lw_band_ng(ib));
#endif
            R *const sm = reinterpret_cast<R *>(lw_lds + LW_LDS_LUT);
            const BandTab<R> &B = T.b[ib];
            const int na = lw_nfraca(ib), nb = lw_nfracb(ib);
            for (int i = threadIdx.x; i < (14 + na + nb) * S; i += (int)blockDim.x) {
                const int r = i / S, c = i - r * S;
                const R *src = r < 10 ? B.selfref + r * S : (r < 14 ? B.forref + (r - 10) * S : (r < 14 + na ? B.fracrefa + (r - 14) * S : B.fracrefb + (r - 14 - na) * S));
                sm[i] = src[c];
            }
        }
        __syncthreads();
        luts = l;
    }
    const int col = bstart + threadIdx.x;
    if (col >= A.ncol) return;
    if (!DBG && (CLD ? col < nclear : col >= nclear)) return;
#ifdef LW_ONLY_BAND
    switch (LW_ONLY_BAND) {
#else
    switch (LW_BAND_ORDER[bslot]) {
#endif
        case 1: band_body<R, Band1, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 2: band_body<R, Band2, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 3: band_body<R, Band3, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 4: band_body<R, Band4, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 5: band_body<R, Band5, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 6: band_body<R, Band6, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 7: band_body<R, Band7, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 8: band_body<R, Band8, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 9: band_body<R, Band9, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 10: band_body<R, Band10, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 11: band_body<R, Band11, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 12: band_body<R, Band12, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 13: band_body<R, Band13, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 14: band_body<R, Band14, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        case 15: band_body<R, Band15, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
        default: band_body<R, Band16, CLD, DBG, BLK == 256>(A, T, col, nclear, luts); break;
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
    const R *part_alt;       // RATS pass: the partials of the bands in alt_mask (bit ib) come from here, the others from A.part
    uint32_t alt_mask;
};

template <typename R>
__global__ void __launch_bounds__(256) k_lw_reduce(LwArgs<R> A, LwOut<R> O)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lev = blockIdx.y;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool ccol = col >= *A.nclear;
    const int pc = A.perm[col];
    const size_t qs = (size_t)NB_LW * (nlay + 1) * n;
    const R *const pmain = A.part + (size_t)lev * n + col;
    const R *const palt = O.part_alt + (size_t)lev * n + col;      // only dereferenced for bands in alt_mask
    R s[6] = {0, 0, 0, 0, 0, 0};
    for (int ib = 0; ib < NB_LW; ib++) {
        const size_t o = (size_t)ib * (nlay + 1) * n;
        const R *const p = ((O.alt_mask >> (ib + 1)) & 1u) ? palt : pmain;
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
    const size_t i = (size_t)lev * ld + pc;
    // a RATS pass (geosrad.hip, lw_dev) asks for the total-sky profiles only: null clear-sky outputs are skipped
    O.dflx[i] = s[0]; O.uflx[i] = s[2];
    if (O.dflxc) { O.dflxc[i] = s[1]; O.uflxc[i] = s[3]; }
    if (A.dudTs) { O.duflx_dTs[i] = s[4]; if (O.duflxc_dTs) O.duflxc_dTs[i] = s[5]; }
    if (lev == nlay) {
        for (int ib = 0; ib < NB_LW; ib++) {
            if (O.band_output[ib]) {
                const size_t o = (size_t)ib * (nlay + 1) * n;
                O.olrb[(size_t)(O.col0 + pc) * NB_LW + ib] = pmain[2 * qs + o];
                if (A.dudTs) O.dolrb_dTs[(size_t)(O.col0 + pc) * NB_LW + ib] = pmain[4 * qs + o];
            }
        }
    }
}

}  // namespace geosrad
