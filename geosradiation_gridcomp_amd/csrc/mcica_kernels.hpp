// mcica_kernels.hpp -- McICA stochastic sub-column generator + LW cloud optics (gfx950).
//
// Reference behaviour: GEOS_RadiationShared/cloud_subcol_gen.F90:132-487 (generate_stochastic_clouds),
// :546-576 (rng_kiss), :611-769 (clearCounts_threeBand); cloud_condensate_inhomogeneity.F90:86-124
// (zcw_lookup); LW/rrtmg_lw_cldprmc.F90:24-385 (cldprmc).
//
// Mapping: lane = column (each column owns one KISS stream, seeded from its own lowest-layer pressures;
// the stream is sequential over (sub-column, layer), so a column is a natural lane).  The 32-bit
// wrap-around integer arithmetic of KISS is done in uint32_t; int -> real conversion and the affine map
// to (0,1) use explicitly un-fused round-to-nearest ops so the random numbers are bit-identical to the
// reference in either precision.
#pragma once
#include "lw_device.hpp"
#include "lw_kernels.hpp"

namespace geosrad {

// un-contracted arithmetic (hipcc contracts a*b+c into fma by default; discrete decisions below must not
// depend on that)
// (HIP's __fmul_rn/__dmul_rn are plain `a * b` and still contract after inlining; the pragma strips the
// `contract` flag from the emitted instruction itself.)
GR_DEV float nf_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b; }
GR_DEV double nf_mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b; }
GR_DEV float nf_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b; }
GR_DEV double nf_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b; }
GR_DEV float nf_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b; }
GR_DEV double nf_sub(double a, double b) {
#pragma clang fp contract(off)
    return a - b; }

struct Kiss { uint32_t s1, s2, s3, s4; };

// one step of a 16-bit multiply-with-carry generator, a * (s & 65535) + (s >> 16): V_MAD_U32_U16 multiplies the LOW halves of its
// first two operands, so the mask costs nothing (the compiler's own choice is v_and + v_mad_u32_u24)
GR_DEV uint32_t mwc16(uint32_t s, uint32_t a)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(s), "s"(a), "v"(s >> 16));
    return r;
}

// rng_kiss (cloud_subcol_gen.F90:570-575)
template <typename R> GR_DEV R kiss_next(Kiss &k)
{
    k.s1 = 69069u * k.s1 + 1327217885u;
    uint32_t x = k.s2;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    k.s2 = x;
    k.s3 = mwc16(k.s3, 18000u);
    k.s4 = mwc16(k.s4, 30903u);
    const int32_t kiss = (int32_t)(k.s1 + k.s2 + (k.s3 << 16) + k.s4);
    return nf_add(nf_mul((R)kiss, (R)2.328306e-10), (R)0.5);
}

// ---- KISS jump-ahead -----------------------------------------------------------------------------------
// The reference draws one stream per column, sequentially over (sub-column, layer): 2*nlay (homogeneous) or
// 4*nlay (inhomogeneous condensate) numbers per sub-column.  To give every (column, band) its own lane we
// advance a freshly seeded state by n = g0 * draws_per_subcolumn in O(1):
//   s1  (LCG mod 2^32)         x -> A1 x + C1              with (A1, C1) = (a^n, c (a^n - 1)/(a - 1))
//   s2  (3-shift xorshift)     GF(2)-linear                x -> M^n x, M^n given by its 32 columns
//   s3,s4 (16-bit multiply-with-carry, s' = a (s & 65535) + (s >> 16)):  s' * 2^16 == s (mod m), m = a 2^16 - 1,
//         and a * 2^16 == 1 (mod m), hence s_n == s_0 * a^n (mod m); from the 2nd step on the state is the
//         canonical residue in [0, m) (or the fixed points 0 / m), so the jump is one modular multiply.
// The constants are computed on the host (geosrad.hip: make_kiss_jump) and passed by value.
struct KissJump { uint32_t A1, C1, K3, K4; uint32_t M2[32]; };

GR_DEV uint32_t mwc_jump(uint32_t s, uint32_t K, uint32_t m)
{
    uint64_t r = s >= m ? s - m : s;
    r = (r * (uint64_t)K) % (uint64_t)m;
    return r == 0 ? s : (uint32_t)r;      // residue 0 <=> the fixed points s = 0 or s = m
}
GR_DEV void kiss_jump(Kiss &k, const KissJump &J)
{
    k.s1 = J.A1 * k.s1 + J.C1;
    uint32_t y = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) y ^= ((k.s2 >> i) & 1u) ? J.M2[i] : 0u;
    k.s2 = y;
    k.s3 = mwc_jump(k.s3, J.K3, 18000u * 65536u - 1u);
    k.s4 = mwc_jump(k.s4, J.K4, 30903u * 65536u - 1u);
}

// correlation_length (cloud_subcol_gen.F90:491-514)
template <typename R> GR_DEV R corr_length(const R *am, int doy, R alat)
{
    const R r2d = (R)(180.0 / 3.14159265358979323846);
    R am3;
    if (doy > 181) am3 = (R)-4. * am[2] / (R)365. * (R)(doy - 272);
    else am3 = (R)4. * am[2] / (R)365. * (R)(doy - 91);
    const R x = alat * r2d - am3;
    return (am[0] + am[1] * gr_exp<R>(-(x * x) / (am[3] * am[3]))) * (R)1.e3;
}

// tables reached through pointers that were read from *Tp / *Sp: loaded as GLOBAL (a generic pointer makes them flat loads with 64-bit lane
// address arithmetic)
template <typename T> GR_DEV T mc_ldg(const T *base, int idx)
{
    typedef const T __attribute__((address_space(1))) *gT;
    return ((gT)base)[idx];
}

// zcw_lookup (cloud_condensate_inhomogeneity.F90:86-124); xcw Fortran (1000,140)
template <typename R> GR_DEV R zcw_lookup(const R *__restrict__ xcw, R cdf, R sigma)
{
    constexpr int n1 = 1000, n2 = 140;
    R rind1 = nf_add(nf_mul(cdf, (R)(n1 - 1)), (R)1.);
    int ind1 = (int)rind1; ind1 = ind1 > n1 - 1 ? n1 - 1 : ind1; ind1 = ind1 < 1 ? 1 : ind1;
    rind1 = nf_sub(rind1, (R)ind1);
    R rind2 = nf_sub(nf_mul((R)40., sigma), (R)3.);
    int ind2 = (int)rind2; ind2 = ind2 > n2 - 1 ? n2 - 1 : ind2; ind2 = ind2 < 1 ? 1 : ind2;
    rind2 = nf_sub(rind2, (R)ind2);
    const int o = (ind2 - 1) * n1 + (ind1 - 1);
    const R p00 = mc_ldg(xcw, o), p01 = mc_ldg(xcw, o + 1), p10 = mc_ldg(xcw, o + n1), p11 = mc_ldg(xcw, o + n1 + 1);
    const R u1 = nf_sub((R)1.0, rind1), u2 = nf_sub((R)1.0, rind2);
    const R t1 = nf_mul(nf_mul(u1, u2), p00), t2 = nf_mul(nf_mul(u1, rind2), p10);
    const R t3 = nf_mul(nf_mul(rind1, u2), p01), t4 = nf_mul(nf_mul(rind1, rind2), p11);
    return nf_add(nf_add(nf_add(t1, t2), t3), t4);
}

// zcw_lookup in two halves, same operations in the same order: the four table values are requested in one place and combined in
// another, so that independent work can sit between the request and its first use (k_mcica_sa)
template <typename R> struct ZcwReq { R v0, v1, v2, v3, r1, r2; };
template <typename R> GR_DEV void zcw_request(const R *__restrict__ xcw, R cdf, R sigma, ZcwReq<R> &q)
{
    constexpr int n1 = 1000, n2 = 140;
    R rind1 = nf_add(nf_mul(cdf, (R)(n1 - 1)), (R)1.);
    int ind1 = (int)rind1; ind1 = ind1 > n1 - 1 ? n1 - 1 : ind1; ind1 = ind1 < 1 ? 1 : ind1;
    q.r1 = nf_sub(rind1, (R)ind1);
    R rind2 = nf_sub(nf_mul((R)40., sigma), (R)3.);
    int ind2 = (int)rind2; ind2 = ind2 > n2 - 1 ? n2 - 1 : ind2; ind2 = ind2 < 1 ? 1 : ind2;
    q.r2 = nf_sub(rind2, (R)ind2);
    const R *p = xcw + (size_t)(ind2 - 1) * n1 + (ind1 - 1);
    q.v0 = p[0]; q.v1 = p[n1]; q.v2 = p[1]; q.v3 = p[n1 + 1];
}
template <typename R> GR_DEV R zcw_combine(const ZcwReq<R> &q)
{
    const R u1 = nf_sub((R)1.0, q.r1), u2 = nf_sub((R)1.0, q.r2);
    const R t1 = nf_mul(nf_mul(u1, u2), q.v0), t2 = nf_mul(nf_mul(u1, q.r2), q.v1);
    const R t3 = nf_mul(nf_mul(q.r1, u2), q.v2), t4 = nf_mul(nf_mul(q.r1, q.r2), q.v3);
    return nf_add(nf_add(nf_add(t1, t2), t3), t4);
}

// KISS seeds from the fractional part of the four lowest-layer pressures (cloud_subcol_gen.F90:375-400)
template <typename R> GR_DEV Kiss kiss_seed(const R *__restrict__ play, int ld, int nlay, int col, bool surface_at_one,
                                            const int *so)
{
    R pseed[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lay = surface_at_one ? k : nlay - 1 - k;
        pseed[k] = nf_mul(play[(size_t)lay * ld + col], (R)100.);
    }
    uint32_t sd[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // pseed[so[k]-1] with a register-resident select (so is a permutation of 1..4)
        const int j = so[k];
        const R ps = j == 1 ? pseed[0] : (j == 2 ? pseed[1] : (j == 3 ? pseed[2] : pseed[3]));
        const R frac = nf_sub(ps, (R)(int32_t)ps);
        sd[k] = (uint32_t)(int32_t)nf_add(nf_mul(frac, (R)2147483646), (R)1);
    }
    Kiss k; k.s1 = sd[0]; k.s2 = sd[1]; k.s3 = sd[2]; k.s4 = sd[3];
    return k;
}

// ---------------------------------------------------------------------------------------------------
// k_overlap: per (layer, column): alpha / rcorr = exp(-|dz| / L) (cloud_subcol_gen.F90:310-321), computed
// once instead of once per sub-column.  Layer 0 entries are unused (set to 0).
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_overlap(int ncol, int ld, int nlay, int doy, const R *__restrict__ zmid,
                                                 const R *__restrict__ alat, const int32_t *__restrict__ perm,
                                                 const int32_t *__restrict__ nclear, const LwDev<R> *__restrict__ T,
                                                 R *__restrict__ alpha, R *__restrict__ rcorr, uint8_t *__restrict__ laycloudy)
{
    // perm / nclear (k_partition) are null for the stand-alone generator: identity, every column processed
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lay = blockIdx.y;
    if (col >= ncol) return;
    if (nclear && col < *nclear) return;
    const int pc = perm ? perm[col] : col;
    const size_t w = (size_t)lay * ncol + col;
    if (laycloudy) laycloudy[w] = 0;     // k_mcica's (column, band) threads OR their findings into it
    if (lay == 0) { alpha[w] = 0; if (rcorr) rcorr[w] = 0; return; }
    const R dz = fabs(zmid[(size_t)lay * ld + pc] - zmid[(size_t)(lay - 1) * ld + pc]);
    const R adl = corr_length<R>(T->aam, doy, alat[pc]);
    alpha[w] = gr_exp<R>(-dz / adl);
    if (T->xcw) {
        const R rdl = corr_length<R>(T->ram, doy, alat[pc]);
        rcorr[w] = gr_exp<R>(-dz / rdl);
    }
}

// LW cloud absorption coefficients of one (layer, band): tau = ciwp * kice + clwp * kliq (LW/rrtmg_lw_cldprmc.F90:84-360).  They
// depend on the effective radii only, so they are looked up once per layer for all sub-columns of the segment; error bits where the
// reference would `error stop` (it does so in cloudy cells only: the caller raises them only if a cell of the layer is cloudy).
template <typename R>
GR_DEV void lw_cloud_coef(const LwDev<R> &T, int iceflag, int ib, R reice, R reliq, R &kice, R &kliq, uint32_t &err)
{
    kice = 0; kliq = 0;
    if (iceflag == 0) kice = T.absice0[0] + T.absice0[1] / reice;
    else if (iceflag == 1) { const int i1 = T.ice1b[ib - 1]; kice = T.absice1[(i1 - 1) * 2] + T.absice1[(i1 - 1) * 2 + 1] / reice; }
    else {
        R factor; int nmax; const R *tab;
        if (iceflag == 2) { factor = (reice - (R)2.) / (R)3.; nmax = 43; tab = T.absice2; }
        else if (iceflag == 3) { factor = (reice - (R)2.) / (R)3.; nmax = 46; tab = T.absice3; }
        else { factor = reice; nmax = 200; tab = T.absice4; }
        int index = (int)factor;
        if (index >= nmax) { if (index == nmax) index = nmax - 1; else { err |= 1u << ERR_ICE_RADIUS_HI; index = nmax - 1; } }
        else if (index <= 0) { if (index == 0) index = 1; else { err |= 1u << ERR_ICE_RADIUS_LO; index = 1; } }
        const R fint = factor - (R)index;
        const int o = (ib - 1) * nmax + (index - 1);
        const R p0 = mc_ldg(tab, o), p1 = mc_ldg(tab, o + 1);
        kice = p0 + fint * (p1 - p0);
    }
    {
        const R factor = reliq - (R)1.5;
        int index = (int)factor;
        if (index >= 58) { if (index == 58) index = 57; else { err |= 1u << ERR_LIQ_RADIUS_HI; index = 57; } }
        else if (index <= 0) { if (index == 0) index = 1; else { err |= 1u << ERR_LIQ_RADIUS_LO; index = 1; } }
        const R fint = factor - (R)index;
        const int o = (ib - 1) * 58 + (index - 1);
        const R p0 = mc_ldg(T.absliq1, o), p1 = mc_ldg(T.absliq1, o + 1);
        kliq = p0 + fint * (p1 - p0);
    }
}

// cldprmc_sw (SW/rrtmg_sw_cldprmc.F90:131-411) in two steps: the extinction / single-scattering / asymmetry / forward-scattering
// coefficients of ice and liquid depend on the effective radii only (one look-up per (layer, band) for all sub-columns of the
// segment); the cell then combines them with its water paths: un-scaled tau (taormc) and the delta-scaled tau / single-scattering
// albedo / asymmetry of the liquid + ice cloud.  jb = 16..29.
template <typename R> struct SwCldCoef { R extcoice, ssacoice, gice, forwice, extcoliq, ssacoliq, gliq, forwliq; };

template <typename R>
GR_DEV SwCldCoef<R> sw_cloud_coef(const SwDev<R> &S, int iceflag, int jb, R radice, R radliq)
{
    const R epsg = (R)1.e-06;
    const int ib = jb - 16;      // 0-based column of the (n,16:29) tables
    SwCldCoef<R> c;
#define LIN_T(tab, nmax) (mc_ldg(tab, ib * (nmax) + index - 1) + fint * (mc_ldg(tab, ib * (nmax) + index) - mc_ldg(tab, ib * (nmax) + index - 1)))
    if (iceflag == 1) {
        const int ic = S.icxa[jb - 15] - 1;
        c.extcoice = S.abari[ic] + S.bbari[ic] / radice;
        c.ssacoice = (R)1. - S.cbari[ic] - S.dbari[ic] * radice;
        c.gice = S.ebari[ic] + S.fbari[ic] * radice;
        if (c.gice > (R)1. - epsg) c.gice = (R)1. - epsg;
        c.forwice = c.gice * c.gice;
    } else if (iceflag == 2) {
        const R factor = (radice - (R)2.) / (R)3.;
        int index = (int)factor; if (index == 43) index = 42;
        const R fint = factor - (R)index;
        index = clampi(index, 1, 42);      // memory safety only: the reference does not range-check here
        c.extcoice = LIN_T(S.extice2, 43); c.ssacoice = LIN_T(S.ssaice2, 43); c.gice = LIN_T(S.asyice2, 43);
        c.forwice = c.gice * c.gice;
    } else if (iceflag == 3) {
        const R factor = (radice - (R)2.) / (R)3.;
        int index = (int)factor; if (index == 46) index = 45;
        const R fint = factor - (R)index;
        index = clampi(index, 1, 45);
        c.extcoice = LIN_T(S.extice3, 46); c.ssacoice = LIN_T(S.ssaice3, 46); c.gice = LIN_T(S.asyice3, 46);
        const R fdelta = LIN_T(S.fdlice3, 46);
        c.forwice = fdelta + (R)0.5 / c.ssacoice;
        if (c.forwice > c.gice) c.forwice = c.gice;
    } else {
        const R factor = radice;
        int index = (int)factor;
        const R fint = factor - (R)index;
        index = clampi(index, 1, 199);
        c.extcoice = LIN_T(S.extice4, 200); c.ssacoice = LIN_T(S.ssaice4, 200); c.gice = LIN_T(S.asyice4, 200);
        c.forwice = c.gice * c.gice;
    }
    {
        int index = (int)(radliq - (R)1.5);
        if (index == 0) index = 1;
        if (index == 58) index = 57;
        const R fint = radliq - (R)1.5 - (R)index;
        index = clampi(index, 1, 57);
        c.extcoliq = LIN_T(S.extliq1, 58);
        c.ssacoliq = LIN_T(S.ssaliq1, 58);
        if (fint < 0 && c.ssacoliq > (R)1.) c.ssacoliq = mc_ldg(S.ssaliq1, ib * 58 + index - 1);
        c.gliq = LIN_T(S.asyliq1, 58);
        c.forwliq = c.gliq * c.gliq;
    }
#undef LIN_T
    return c;
}

template <typename R>
GR_DEV void sw_cloud_optics(const SwCldCoef<R> &c, int iceflag, R ciwp, R clwp, R &taor, R &tauc, R &ssac, R &asmc)
{
    const R cldmin = (R)1.e-20;
    // a phase without condensate enters with zero coefficients, as the reference leaves them (:131-133)
    const bool ice = ciwp != 0, liq = clwp != 0;
    const R extcoice = ice ? c.extcoice : (R)0, ssacoice = ice ? c.ssacoice : (R)0, gice = ice ? c.gice : (R)0, forwice = ice ? c.forwice : (R)0;
    const R extcoliq = liq ? c.extcoliq : (R)0, ssacoliq = liq ? c.ssacoliq : (R)0, gliq = liq ? c.gliq : (R)0, forwliq = liq ? c.forwliq : (R)0;
    const R tauliqorig = clwp * extcoliq, tauiceorig = ciwp * extcoice;
    taor = tauliqorig + tauiceorig;
    const R ssaliq = ssacoliq * ((R)1. - forwliq) / ((R)1. - forwliq * ssacoliq);
    const R ssaice = ssacoice * ((R)1. - forwice) / ((R)1. - forwice * ssacoice);
    const R tauliq = ((R)1. - forwliq * ssacoliq) * tauliqorig;
    const R tauice = ((R)1. - forwice * ssacoice) * tauiceorig;
    const R scatliq = ssaliq * tauliq;
    R scatice = ssaice * tauice;
    R tc = tauliq + tauice;
    if (tc == 0) tc = cldmin;
    if (scatice == 0) scatice = cldmin;
    tauc = tc;
    ssac = (scatliq + scatice) / tc;
    if (iceflag == 3)
        asmc = ((R)1. / (scatliq + scatice)) * (scatliq * (gliq - forwliq) / ((R)1. - forwliq) + scatice * ((gice - forwice) / ((R)1. - forwice)));
    else
        asmc = (scatliq * (gliq - forwliq) / ((R)1. - forwliq) + scatice * (gice - forwice) / ((R)1. - forwice)) / (scatliq + scatice);
}

// ---------------------------------------------------------------------------------------------------
// k_mcica: one thread per (column, segment of <= 4 consecutive sub-columns of one band).
// The reference draws one KISS stream per column, sequentially over (sub-column, layer): per sub-column 2*nlay numbers
// for the cloud-presence pass and, with inhomogeneous condensate, 2*nlay more for the condensate pass
// (cloud_subcol_gen.F90:402-466).  Here every sub-column of the segment gets its own two stream positions by
// jump-ahead (KissJump): the column's freshly seeded state is advanced to the segment's first sub-column, from there by
// one sub-column per step, and by 2*nlay to the condensate pass.  The thread then walks the LAYERS once, advancing all its
// streams together, so each layer's inputs (overlap correlations, cloud fraction, water paths, radii) are read once
// per segment instead of once per sub-column, and each output cell is written exactly once.
//   MODE 0 (RRTMG_LW): segments = quads of the 16 bands' g-points; fused generate_stochastic_clouds +
//          clearCounts_threeBand + cldprmc: writes taucmc (band-major plane layout), ORs laycloudy[lay][col],
//          adds clearCounts(ncol,4).
//   MODE 1 (stand-alone generator API): quads of sub-columns; writes cldy/ciwp_stoch/clwp_stoch Fortran (nlay,nsubcol,ncol).
//   MODE 2 (RRTMG_SW): quads of the 14 bands' g-points; fused generator + clearCounts + cldprmc_sw: writes
//          taucmc / ssacmc / asmcmc planes and the PAR bands' super-layer sums of the un-scaled tau (cotsum).
// ---------------------------------------------------------------------------------------------------
constexpr int MC_S = 4;      // sub-columns per thread
struct McSegDev { int start, count, band, pad; KissJump j; };      // j: jump to sub-column `start` (unused when start == 0)
struct McPlan { const McSegDev *seg; int nseg; KissJump jsub, jhalf; };      // jsub: one sub-column ahead; jhalf: 2*nlay ahead

template <typename R> struct McArgs {
    int ncol, ld, nlay, nsubcol, doy, cloudLM, cloudMH, iceflg, liqflg;
    int so[4];
    R cwp_tiny;
    const R *play, *cldf, *ciwp, *clwp, *rei, *rel;
    const R *alpha, *rcorr;            // [nlay][ncol]
    const int32_t *perm, *nclear;      // k_partition's compaction (null: identity, all columns)
    const uint8_t *cftop;              // [ncol] original order: 1 + the highest layer with cloud fraction (null: walk every layer)
    // MODE 0
    R *taucmc; uint8_t *laycloudy; int32_t *clearCounts; uint32_t *err;
    // MODE 1
    int32_t *cldy; R *ciwp_s, *clwp_s;
    // MODE 2
    R *ssacmc, *asmcmc, *cotsum;
};

template <typename R, int MODE>
__global__ void __launch_bounds__(64) k_mcica(McArgs<R> M, McPlan P, const LwDev<R> *__restrict__ Tp, const SwDev<R> *__restrict__ Sp)
{
    // one-dimensional grid (lw_kernels.hpp band_block): the segments of a 64-column block are consecutive blocks of one XCD, so the
    // layer fields every segment walks (overlap correlations, cloud fraction, water paths, radii: ~2 KB per column) come from HBM
    // once and from that XCD's L2 for the other segments; with (column block, segment) = (blockIdx.x, blockIdx.y) every segment
    // fetched them again (PMC: 158 KB per column read for 3 KB of inputs)
    int bstart, bseg;
    if (!band_block(M.ncol, P.nseg, bstart, bseg)) return;
    const int col = bstart + threadIdx.x;
    if (col >= M.ncol) return;
    if (M.nclear && col < *M.nclear) return;      // clear columns: nothing to generate
    const int pc = M.perm ? M.perm[col] : col;    // API arrays are indexed by the original column
    const LwDev<R> &T = *Tp;
    const McSegDev &SG = P.seg[bseg];
    const int s0 = SG.start, ns = SG.count, ib = SG.band;     // uniform over the block
    const int n = M.ncol, ld = M.ld, nlay = M.nlay;
    const bool inhomo = T.xcw != nullptr;
    constexpr bool PLANES = MODE == 0 || MODE == 2;
    // vertical ordering is detected from the first column of the call (cloud_subcol_gen.F90:266)
    const bool surface_at_one = M.play[0] > M.play[(size_t)(nlay - 1) * ld];

    // stream positions of the segment's sub-columns: k1 = presence pass, k2 = condensate pass
    Kiss k1[MC_S], k2[MC_S];
    k1[0] = kiss_seed<R>(M.play, ld, nlay, pc, surface_at_one, M.so);
    // n = 0 must stay the identity: a raw seed may exceed the MWC modulus (non-canonical), which mwc_jump would reduce
    if (s0 > 0) kiss_jump(k1[0], SG.j);
#pragma unroll
    for (int s = 1; s < MC_S; s++) { k1[s] = k1[s - 1]; if (s < ns) kiss_jump(k1[s], P.jsub); }
#pragma unroll
    for (int s = 0; s < MC_S; s++) { k2[s] = k1[s]; if (inhomo && s < ns) kiss_jump(k2[s], P.jhalf); }

    // pressure super-layer bounds, 0-based inclusive (cloud_subcol_gen.F90:617-632)
    int lo0, lo1, mi0, mi1, hi0, hi1;
    if (M.cloudLM < M.cloudMH) { lo0 = 0; lo1 = M.cloudLM - 1; mi0 = M.cloudLM; mi1 = M.cloudMH - 1; hi0 = M.cloudMH; hi1 = nlay - 1; }
    else { hi0 = 0; hi1 = M.cloudMH - 2; mi0 = M.cloudMH - 1; mi1 = M.cloudLM - 2; lo0 = M.cloudLM - 1; lo1 = nlay - 1; }

    // plane layout (see band_body): band-major, then [layer][g-in-band][column]
    const int bg0 = MODE == 0 ? lw_band_g0(ib) : (MODE == 2 ? sw_band_g0(ib) : 0);
    const int bng = MODE == 0 ? lw_band_ng(ib) : (MODE == 2 ? sw_band_ng(ib) : 0);
    // cell (band ib, layer il, sub-column s0 + s, column) = wave-uniform start of (band, first sub-column of the segment) + a 32-bit lane part
    // (a band's planes are nlay x ng x ncol reals: < 4 GB for every chunk the context launches)
    const size_t tbu = PLANES ? (size_t)bg0 * nlay * n + (size_t)(s0 - bg0) * n : 0;
    const uint32_t tbs = PLANES ? (uint32_t)bng * (uint32_t)n : 0u;

    R cprev[MC_S], c3prev[MC_S], cs_lo[MC_S], cs_mid[MC_S], cs_hi[MC_S];
#pragma unroll
    for (int s = 0; s < MC_S; s++) { cprev[s] = 0; c3prev[s] = 0; cs_lo[s] = 0; cs_mid[s] = 0; cs_hi[s] = 0; }
    uint32_t any_all = 0, any_hi = 0, any_mid = 0, any_lo = 0;      // bit s: sub-column s has a cloudy cell (in the super-layer)
    uint32_t err = 0;

    // The walk ends above the highest layer in which a column of the wave has cloud fraction: the reference walks on to the top
    // (the streams are consumed for every layer), but a layer without cloud fraction is clear in every sub-column whatever is drawn,
    // nothing above reads the streams again, and nothing is written for such a layer.  In RRTMG's ordering (surface first) that is
    // the whole stratosphere.
    int lend = nlay;
    if (PLANES && M.cftop) {
        const int t = (int)M.cftop[pc];
        int m = 0;                         // maximum over the wave's active lanes, bit by bit (ballots only see active lanes)
#pragma unroll
        for (int b = 7; b >= 0; b--)
            if (__ballot(t >= (m | (1 << b))) != 0) m |= 1 << b;
        lend = m < nlay ? m : nlay;
    }
    for (int il = 0; il < lend; il++) {
        const size_t w = (size_t)il * n + col, a = (size_t)il * ld + pc;
        const R al = il > 0 ? M.alpha[w] : (R)0;
        const R rc = (inhomo && il > 0) ? M.rcorr[w] : (R)0;
        const R cf = M.cldf[a], ciw = M.ciwp[a], clw = M.clwp[a];
        const R rei = PLANES ? M.rei[a] : (R)0, rel = PLANES ? M.rel[a] : (R)0;
        const R thr = nf_sub((R)1., cf);
        const R sigma = cf > (R)0.99 ? (R)0.5 : (cf > (R)0.9 ? (R)0.71 : (R)1.0);
        const bool in_hi = il >= hi0 && il <= hi1, in_mid = il >= mi0 && il <= mi1, in_lo = il >= lo0 && il <= lo1;
        // the draws of the segment's sub-columns first; then, if any sub-column of any column of the wave is cloudy here, the
        // condensate scaling factors of all of them together (four table values each) - not one look-up per cloudy sub-column
        // inside a per-lane branch, where each would wait for the one before
        bool cldy[MC_S];
        R zcws[MC_S];
        bool anyc = false;
#pragma unroll
        for (int s = 0; s < MC_S; s++) {
            cldy[s] = false; zcws[s] = 1;
            if (s >= ns) continue;
            // cloud presence with exponential overlap (:406-414)
            R cdf1 = kiss_next<R>(k1[s]);
            const R cdf2 = kiss_next<R>(k1[s]);
            if (il > 0 && cdf2 < al) cdf1 = cprev[s];
            cprev[s] = cdf1;
            cldy[s] = cdf1 >= thr;
            anyc = anyc || cldy[s];
            if (inhomo) {
                // condensate with exponential overlap + inhomogeneity (:416-466); the stream is consumed for every layer
                const R c2 = kiss_next<R>(k2[s]);
                R cdf3 = kiss_next<R>(k2[s]);
                if (il > 0 && c2 < rc) cdf3 = c3prev[s];
                c3prev[s] = cdf3;
            }
        }
        const bool wany = __ballot(anyc) != 0;
        if (inhomo && wany) {
#pragma unroll
            for (int s = 0; s < MC_S; s++)
                if (s < ns) zcws[s] = zcw_lookup<R>(T.xcw, c3prev[s], sigma);
        }
        R kice = 0, kliq = 0;
        uint32_t kerr = 0;
        if (MODE == 0 && wany) lw_cloud_coef<R>(T, M.iceflg, ib, rei, rel, kice, kliq, kerr);
        SwCldCoef<R> swc{};
        if (MODE == 2 && wany) swc = sw_cloud_coef<R>(*Sp, M.iceflg, ib, rei, rel);
#pragma unroll
        for (int s = 0; s < MC_S; s++) {
            if (s >= ns) continue;
            const bool cloudy = cldy[s];
            R ci = 0, cl = 0;
            if (cloudy) {
                if (inhomo) { ci = nf_mul(ciw, zcws[s]); cl = nf_mul(clw, zcws[s]); }
                else { ci = ciw; cl = clw; }      // homogeneous condensate (:438-443)
            }
            bool c = false;
            if (cloudy) {
                const bool cin = ci <= M.cwp_tiny, cln = cl <= M.cwp_tiny;
                if (cin) ci = 0;
                if (cln) cl = 0;
                c = !(cin && cln);
            }
            if (c) {
                any_all |= 1u << s;
                if (in_hi) any_hi |= 1u << s;
                if (in_mid) any_mid |= 1u << s;
                if (in_lo) any_lo |= 1u << s;
            }
            // cells of a layer without cloud fraction are clear in every sub-column: they are neither written here nor read by
            // the band kernels (which test laycloudy, set only where some sub-column has an optically non-zero cloud)
            if (MODE == 0) {
                if (cf > 0) {
                    R tau = 0;
                    if (c) {
                        if (ci > 0) tau = ci * kice;
                        if (cl > 0) tau = tau + cl * kliq;
                        err |= kerr;
                    }
                    stg(M.taucmc + tbu, ((uint32_t)col + (uint32_t)s * (uint32_t)n + (uint32_t)il * tbs) * (uint32_t)sizeof(R), tau);
                    if (tau > 0) M.laycloudy[w] = 1;
                }
            } else if (MODE == 2) {
                R taor = 0, tauc = 0, ssac = 1, asmc = 0;
                if (c) sw_cloud_optics<R>(swc, M.iceflg, ci, cl, taor, tauc, ssac, asmc);
                if (cf > 0) {
                    const uint32_t oc = ((uint32_t)col + (uint32_t)s * (uint32_t)n + (uint32_t)il * tbs) * (uint32_t)sizeof(R);
                    stg(M.taucmc + tbu, oc, tauc); stg(M.ssacmc + tbu, oc, ssac); stg(M.asmcmc + tbu, oc, asmc);
                    if (tauc > 0) M.laycloudy[w] = 1;
                }
                // super-layer sums of the un-scaled tau for the PAR diagnostics (SW/rrtmg_sw_spcvmc.F90:760-800)
                if (il < M.cloudLM) cs_lo[s] += taor; else if (il < M.cloudMH) cs_mid[s] += taor; else cs_hi[s] += taor;
            } else {
                const size_t o = ((size_t)col * M.nsubcol + (s0 + s)) * nlay + il;
                M.cldy[o] = c ? 1 : 0; M.ciwp_s[o] = ci; M.clwp_s[o] = cl;
            }
        }
    }
    if (MODE == 2 && ib >= 24 && ib <= 26) {
#pragma unroll
        for (int s = 0; s < MC_S; s++) {
            if (s >= ns) continue;
            M.cotsum[((size_t)0 * 112 + s0 + s) * n + col] = cs_lo[s];
            M.cotsum[((size_t)1 * 112 + s0 + s) * n + col] = cs_mid[s];
            M.cotsum[((size_t)2 * 112 + s0 + s) * n + col] = cs_hi[s];
        }
    }
    if (PLANES) {
        // integer adds: order-independent, bitwise reproducible (the validate kernel zeroed the cloudy columns' counts)
        const uint32_t act = (1u << ns) - 1u;
        atomicAdd(&M.clearCounts[(size_t)0 * ld + pc], __popc(~any_all & act));
        atomicAdd(&M.clearCounts[(size_t)1 * ld + pc], __popc(~any_hi & act));
        atomicAdd(&M.clearCounts[(size_t)2 * ld + pc], __popc(~any_mid & act));
        atomicAdd(&M.clearCounts[(size_t)3 * ld + pc], __popc(~any_lo & act));
        if (err) atomicOr(M.err, err);
    }
}

// ---------------------------------------------------------------------------------------------------
// k_mcica_sa: the stand-alone generator API (generate_stochastic_clouds, cloud_subcol_gen.F90:132-487) with every cell of
// cldy_stoch / ciwp_stoch / clwp_stoch, Fortran (nlay,nsubcol,ncol), materialised: 12 B out per (layer, sub-column) cell against
// ~60 integer operations - the HBM roofline is the bound that matters here, so the mapping follows the OUTPUT.
//   lane = one (column, sub-column) pair, pairs numbered f = col * nsubcol + isub: the 64 pairs of a wavefront own the contiguous
//   block [f0 * nlay, (f0 + 64) * nlay) of each output array whatever nsubcol is (a wave may straddle two columns).
//   Each lane reaches its sub-column's two stream positions by jump-ahead (table of nsubcol jumps, built once on the host) and walks
//   the layers once; the only per-cell state the outputs need - the condensate scaling factor, or "clear" - goes to an LDS tile
//   [64 pairs][nlay] (row stride odd: conflict-free both ways).  The tile is then written out ROW-MAJOR: consecutive lanes write
//   consecutive addresses (256 B per store instruction), the water paths re-formed from the layer's ciwp / clwp exactly as
//   the walk would have (one multiplication, cloud_subcol_gen.F90:438-466).  k_mcica<R, 1> (lane = column) wrote 4 B per lane at a
//   stride of nsubcol * nlay * 4 B.
// A pair whose column has no cloud fraction anywhere is clear in every cell (the generator's numbers are < 1): its lanes skip the
// walk, a wave of such pairs skips it altogether.
// ---------------------------------------------------------------------------------------------------
constexpr int MC_SA_K = 8;          // chunks of 64 pairs per wavefront: the staged layer fields and the prologue are shared by the chunks
// LDS reals per block (one wavefront): the tile, five staged layer fields of the block's columns, their "has cloud fraction" flags
__host__ __device__ constexpr int mc_sa_ncw(int nsub) { return (64 * MC_SA_K - 1) / nsub + 2; }      // columns MC_SA_K chunks can touch
__host__ __device__ constexpr size_t mc_sa_lds_reals(int nlay, int nsub)
{
    return (size_t)64 * (nlay | 1) + (size_t)5 * mc_sa_ncw(nsub) * nlay + (size_t)mc_sa_ncw(nsub);
}
template <typename R>
__global__ void __launch_bounds__(64) k_mcica_sa(McArgs<R> M, const KissJump *__restrict__ jsubs, KissJump jhalf,
                                                 const LwDev<R> *__restrict__ Tp)
{
    extern __shared__ __align__(16) unsigned char mc_lds[];
    const LwDev<R> &T = *Tp;
    const int nlay = M.nlay, ld = M.ld, nsub = M.nsubcol, n = M.ncol;
    const int rs = nlay | 1;
    const long ntot = (long)n * nsub;
    const int lane = threadIdx.x;
    const int ncw = mc_sa_ncw(nsub);
    R *const tile = reinterpret_cast<R *>(mc_lds);
    const long fb = (long)blockIdx.x * (64 * MC_SA_K);      // first pair of the block
    const bool inhomo = T.xcw != nullptr;
    const bool surface_at_one = M.play[0] > M.play[(size_t)(nlay - 1) * ld];

    // the layer inputs of the block's columns, staged once: the walk and the write-out read them from LDS (a loop with a run-time trip
    // count is not unrolled, so a global load per layer inside it would cost a full memory round trip per layer)
    const int col0 = (int)(fb / nsub), nst = ncw * nlay;
    R *const cwi = tile + 64 * rs, *const cwl = cwi + nst, *const cfs = cwl + nst, *const als = cfs + nst, *const rcs = als + nst;
    int *const colflag = reinterpret_cast<int *>(rcs + nst);
    for (int i = lane; i < ncw; i += 64) colflag[i] = 0;
    __builtin_amdgcn_wave_barrier();
    {
        int c = 0, l = lane;
        while (l >= nlay) { l -= nlay; c++; }
        for (int i = lane; i < nst; i += 64) {
            const int cc = col0 + c < n ? col0 + c : n - 1;
            const size_t a = (size_t)l * ld + cc, w = (size_t)l * n + cc;
            const R v0 = M.ciwp[a], v1 = M.clwp[a], v2 = M.cldf[a], v3 = M.alpha[w], v4 = inhomo ? M.rcorr[w] : (R)0;
            cwi[i] = v0; cwl[i] = v1; cfs[i] = v2; als[i] = v3; rcs[i] = v4;
            if (v2 > 0) atomicMax(&colflag[c], l + 1);   // 1 + the column's highest layer with cloud fraction (0: none)
            l += 64;
            while (l >= nlay) { l -= nlay; c++; }
        }
    }
    __syncthreads();
  for (int ch = 0; ch < MC_SA_K; ch++) {
    const long f0 = fb + (long)ch * 64;
    if (f0 >= ntot) break;                           // uniform
    const bool active = f0 + lane < ntot;
    const long f = active ? f0 + lane : ntot - 1;
    const int col = (int)(f / nsub), isub = (int)(f - (long)col * nsub);
    const int cofs = (col - col0) * nlay;
    const int cftop = active ? colflag[col - col0] : 0;
    const bool colcloudy = cftop != 0;
    const bool wave_cloudy = __ballot(colcloudy) != 0;
    if (wave_cloudy) {
        // the walk ends above the wave's highest layer with cloud fraction (k_mcica): the cells above are clear whatever is drawn
        int wtop = 0;
#pragma unroll
        for (int b = 15; b >= 0; b--)
            if (__ballot(cftop >= (wtop | (1 << b))) != 0) wtop |= 1 << b;
        constexpr int MC_G = 4;
        int nl = (wtop + MC_G - 1) & ~(MC_G - 1);
        nl = nl < nlay ? nl : nlay;
        for (int il = nl; il < nlay; il++) tile[lane * rs + il] = (R)-1;
        Kiss k1 = kiss_seed<R>(M.play, ld, nlay, col, surface_at_one, M.so);
        if (isub > 0) {       // n = 0 must stay the identity (a raw seed may be a non-canonical MWC residue)
            KissJump J = jsubs[isub];
            kiss_jump(k1, J);
        }
        Kiss k2 = k1;
        if (inhomo) kiss_jump(k2, jhalf);
        R cprev = 0, c3prev = 0;
        // layers in groups of MC_G: the group's draws (integer work only), then the previous group's scaling factors are combined
        // and stored, then this group's table values are requested - a request has a whole group's arithmetic to arrive in;
        // the layer inputs of the next group are requested a group ahead as well
        ZcwReq<R> pq[MC_G];
        bool pcl[MC_G];
#pragma unroll
        for (int k = 0; k < MC_G; k++) { pcl[k] = false; pq[k].v0 = pq[k].v1 = pq[k].v2 = pq[k].v3 = pq[k].r1 = pq[k].r2 = 0; }
        bool ppend = false;
        for (int g0 = 0; g0 < nl + MC_G; g0 += MC_G) {        // one extra trip stores the last group
            R al[MC_G], rc[MC_G], cf[MC_G];
#pragma unroll
            for (int k = 0; k < MC_G; k++) {
                int il = g0 + k; il = il < nlay ? il : nlay - 1;
                al[k] = als[cofs + il]; rc[k] = rcs[cofs + il]; cf[k] = cfs[cofs + il];
            }
            bool cl[MC_G];
            R c3[MC_G], sg[MC_G];
            bool anyc = false;
#pragma unroll
            for (int k = 0; k < MC_G; k++) {
                const int il = g0 + k;
                cl[k] = false; c3[k] = 0; sg[k] = 1;
                if (il >= nl) continue;
                const R thr = nf_sub((R)1., cf[k]);
                sg[k] = cf[k] > (R)0.99 ? (R)0.5 : (cf[k] > (R)0.9 ? (R)0.71 : (R)1.0);
                // cloud presence with exponential overlap (:406-414)
                R cdf1 = kiss_next<R>(k1);
                const R cdf2 = kiss_next<R>(k1);
                if (il > 0 && cdf2 < al[k]) cdf1 = cprev;
                cprev = cdf1;
                cl[k] = colcloudy && cdf1 >= thr;
                anyc = anyc || cl[k];
                if (inhomo) {     // condensate with exponential overlap (:416-466); the stream is consumed for every layer
                    const R c2 = kiss_next<R>(k2);
                    R cdf3 = kiss_next<R>(k2);
                    if (il > 0 && c2 < rc[k]) cdf3 = c3prev;
                    c3prev = cdf3;
                    c3[k] = cdf3;
                }
            }
            // the previous group's cells
#pragma unroll
            for (int k = 0; k < MC_G; k++) {
                const int il = g0 - MC_G + k;
                if (il < 0 || il >= nl) continue;
                const R z = ppend ? zcw_combine<R>(pq[k]) : (R)1;
                tile[lane * rs + il] = pcl[k] ? z : (R)-1;
            }
            // this group's requests (wave-uniform test around the gathers)
            ppend = inhomo && __ballot(anyc) != 0;
            if (ppend) {
#pragma unroll
                for (int k = 0; k < MC_G; k++) zcw_request<R>(T.xcw, c3[k], sg[k], pq[k]);
            }
#pragma unroll
            for (int k = 0; k < MC_G; k++) pcl[k] = cl[k];
        }
    }
    __syncthreads();
    const long rows = (ntot - f0) < 64 ? (ntot - f0) : 64;
    const int total = (int)rows * nlay;
    const size_t o0 = (size_t)f0 * nlay;
    // row r of the tile = pair f0 + r; its column, relative to col0, is tracked with r (no division in the loop)
    const int c0rel = (int)(f0 / nsub) - col0;
    if constexpr (sizeof(R) == 4) {
        if ((nlay & 3) == 0) {
            // four consecutive layers of a row per lane: 16-byte stores, 1 KB per store instruction, a quarter of the loop trips
            int r = 0, j = lane * 4, rq = (int)(f0 - (long)(col0 + c0rel) * nsub), crel = c0rel;
            while (j >= nlay) { j -= nlay; r++; if (++rq >= nsub) { rq -= nsub; crel++; } }
            for (int idx = lane * 4; idx < total; idx += 256) {
                R z[4], ci[4], cl[4];
                int c[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { z[k] = wave_cloudy ? tile[r * rs + j + k] : (R)-1; ci[k] = 0; cl[k] = 0; c[k] = 0; }
                if (z[0] >= (R)0 || z[1] >= (R)0 || z[2] >= (R)0 || z[3] >= (R)0) {
                    const int a = crel * nlay + j;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (z[k] >= (R)0) {
                            R x = nf_mul(cwi[a + k], z[k]), y = nf_mul(cwl[a + k], z[k]);
                            const bool cin = x <= M.cwp_tiny, cln = y <= M.cwp_tiny;
                            ci[k] = cin ? (R)0 : x; cl[k] = cln ? (R)0 : y;
                            c[k] = !(cin && cln);
                        }
                    }
                }
                *reinterpret_cast<int4 *>(M.cldy + o0 + idx) = make_int4(c[0], c[1], c[2], c[3]);
                *reinterpret_cast<float4 *>(M.ciwp_s + o0 + idx) = make_float4(ci[0], ci[1], ci[2], ci[3]);
                *reinterpret_cast<float4 *>(M.clwp_s + o0 + idx) = make_float4(cl[0], cl[1], cl[2], cl[3]);
                j += 256;
                while (j >= nlay) { j -= nlay; r++; if (++rq >= nsub) { rq -= nsub; crel++; } }
            }
            __syncthreads();                         // the tile is reused by the next chunk
            continue;
        }
    }
    int r = 0, j = lane, rq = (int)(f0 - (long)(col0 + c0rel) * nsub), crel = c0rel;
    while (j >= nlay) { j -= nlay; r++; if (++rq >= nsub) { rq -= nsub; crel++; } }
#pragma unroll 4
    for (int idx = lane; idx < total; idx += 64) {
        const R z = wave_cloudy ? tile[r * rs + j] : (R)-1;
        R ci = 0, cl = 0;
        bool c = false;
        if (z >= (R)0) {
            const int a = crel * nlay + j;
            // homogeneous condensate: z = 1 and x * 1 = x (:438-443)
            ci = nf_mul(cwi[a], z); cl = nf_mul(cwl[a], z);
            const bool cin = ci <= M.cwp_tiny, cln = cl <= M.cwp_tiny;
            if (cin) ci = 0;
            if (cln) cl = 0;
            c = !(cin && cln);
        }
        M.cldy[o0 + idx] = c ? 1 : 0; M.ciwp_s[o0 + idx] = ci; M.clwp_s[o0 + idx] = cl;
        j += 64;
        while (j >= nlay) { j -= nlay; r++; if (++rq >= nsub) { rq -= nsub; crel++; } }
    }
    __syncthreads();                                 // the tile is reused by the next chunk
  }
}

// clearCounts_threeBand stand-alone (cloud_subcol_gen.F90:611-769): cldy Fortran (nlay,nsubcol,ncol)
static __global__ void __launch_bounds__(64) k_clearcounts(int ncol, int nsubcol, int nlay, int cloudLM, int cloudMH,
                                                    const int32_t *__restrict__ cldy, int32_t *__restrict__ cnt /*(4,ncol)*/)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    int lo0, lo1, mi0, mi1, hi0, hi1;
    if (cloudLM < cloudMH) { lo0 = 0; lo1 = cloudLM - 1; mi0 = cloudLM; mi1 = cloudMH - 1; hi0 = cloudMH; hi1 = nlay - 1; }
    else { hi0 = 0; hi1 = cloudMH - 2; mi0 = cloudMH - 1; mi1 = cloudLM - 2; lo0 = cloudLM - 1; lo1 = nlay - 1; }
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int is = 0; is < nsubcol; is++) {
        bool a = false, h = false, m = false, l = false;
        const int32_t *p = cldy + ((size_t)col * nsubcol + is) * nlay;
        for (int il = 0; il < nlay; il++)
            if (p[il]) { a = true; if (il >= hi0 && il <= hi1) h = true; if (il >= mi0 && il <= mi1) m = true; if (il >= lo0 && il <= lo1) l = true; }
        c0 += !a; c1 += !h; c2 += !m; c3 += !l;
    }
    cnt[4 * col] = c0; cnt[4 * col + 1] = c1; cnt[4 * col + 2] = c2; cnt[4 * col + 3] = c3;
}

}  // namespace geosrad
