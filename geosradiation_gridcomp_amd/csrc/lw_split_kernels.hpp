// lw_split_kernels.hpp -- RRTMG_LW band sweeps as TWO kernels (GEOSRAD_LW_PATH=split):
//   k_lw_cells : one thread per (column, band, chunk of layers): the k-distribution of the layers, with no vertical dependence - what the
//                fused k_lw_bands does inside its downward sweep (taumol, LW/rrtmg_lw_taumol.F90:155-3126, + the Pade index of the
//                discretised optical depth, rrtmg_lw_rtrnmc.F90:245-283).  Parks per cell the 2-byte index of the total-sky optical depth
//                (and the gas-only one below a cloudy column's cloud top), per (band, layer, column) which Planck fractions the layer takes.
//   k_lw_sweep : lane = column, block = 1 024 / 768 columns of one band: the two vertical recurrences of rtrnmc (:245-379) from the parked
//                indices - no table-row gathers, no layer record, 121-168 VGPRs: four / three wavefronts per SIMD.
// Why it was built: k_lw_bands waits on its per-layer chain (record -> table rows -> look-up) at the two wavefronts per SIMD its 220-240 VGPRs
// allow (profiles/r03_lw_units.md); the k-distribution has no vertical dependence, so as a kernel of its own it has many times the threads
// and no adding state to carry.  The arithmetic per cell is k_lw_bands' (same operations, same order; the compiler's choice of fused
// multiply-adds aside).  MEASURED SLOWER than the fused kernel - 4.1 + 3.4 against 5.9 ms per 97 200 columns (profiles/r04_lw_split.md:
// the k-distribution alone is 70 % of the fused kernel, which overlaps the two halves inside one wavefront) - selectable, not the default.
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {

constexpr int LWS_CHUNKS = 8;          // layer chunks of k_lw_cells (threads per (column, band))
// columns per block of k_lw_sweep: fp32 - the wavefronts of a block share one copy of the transmittance table in LDS (82 KB): 1 024
// cloud-free columns (<= 128 VGPRs: four wavefronts per SIMD), 768 cloudy ones (two skies: <= 168 VGPRs, three); fp64 reads the table from
// L2 and keeps 256-column blocks
template <typename R, bool CLD> constexpr int lws_block = sizeof(R) == 4 ? (CLD ? 768 : 1024) : 256;

// parked indices of the split path: per band [256-column block][layer][column in block][g] - a lane's NG indices of a layer are contiguous
// (2 NG bytes: one or two 16-byte accesses)
template <int NG> GR_DEV size_t lws_cell(int ncol_pad, int nlay, int g0, uint32_t ucol, int lay)
{
    return (size_t)g0 * nlay * ncol_pad + (((size_t)(ucol >> 8) * nlay + lay) * 256u + (ucol & 255u)) * NG;
}
// (a lane's run is 4-byte aligned: 2 NG bytes per lane, NG even)
typedef uint32_t lws_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t lws_u2 __attribute__((ext_vector_type(2), aligned(4)));
template <int NW> GR_DEV void lws_store(uint16_t *p, const uint32_t (&w)[NW])
{
    uint32_t *q = reinterpret_cast<uint32_t *>(p);
#pragma unroll
    for (int k = 0; k + 4 <= NW; k += 4) { lws_u4 v; v.x = w[k]; v.y = w[k + 1]; v.z = w[k + 2]; v.w = w[k + 3]; __builtin_nontemporal_store(v, reinterpret_cast<lws_u4 *>(q + k)); }
    if constexpr (NW % 4 >= 2) { lws_u2 v; v.x = w[NW / 4 * 4]; v.y = w[NW / 4 * 4 + 1]; __builtin_nontemporal_store(v, reinterpret_cast<lws_u2 *>(q + NW / 4 * 4)); }
    if constexpr (NW % 2 == 1) __builtin_nontemporal_store(w[NW - 1], q + NW - 1);
}
template <int NW> GR_DEV void lws_load(const uint16_t *p, uint32_t (&w)[NW])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int k = 0; k + 4 <= NW; k += 4) {
        const lws_u4 v = __builtin_nontemporal_load(reinterpret_cast<const lws_u4 *>(q + k));
        w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
    if constexpr (NW % 4 >= 2) { const lws_u2 v = __builtin_nontemporal_load(reinterpret_cast<const lws_u2 *>(q + NW / 4 * 4)); w[NW / 4 * 4] = v.x; w[NW / 4 * 4 + 1] = v.y; }
    if constexpr (NW % 2 == 1) w[NW - 1] = __builtin_nontemporal_load(q + NW - 1);
}

// ---------------------------------------------------------------------------------------------------
// k_lw_cells
// ---------------------------------------------------------------------------------------------------
template <typename R, typename BAND, bool CLD>
GR_DEV void lws_cells_body(const LwArgs<R> &A, const LwDev<R> &T, int col, int nclear, int lay0, int lay1)
{
    constexpr int NG = BAND::NG, IB = BAND::IB, G0 = BAND::G0, W = NG >= 4 ? 4 : 2, NQ = (NG + W - 1) / W, NW = NG / 2;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const R bpade = T.bpade, tblint = (R)NTBL;
    const uint32_t ucol = (uint32_t)col;
    const int pc = ldg(A.perm, ucol * 4u);
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
    const bool ccol = CLD && col >= nclear;
    // the gas-only index is parked next to the total-sky one wherever the clear-sky stream of a cloudy column can have parted from the
    // total-sky one: at and below the column's highest layer with cloud fraction (the optically cloudy layers are among them)
    const int ctop = CLD && ccol ? (int)A.colcloudy[pc] - 1 : -1;
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    const R *const taucmc_b = A.taucmc + (size_t)G0 * nlay * n;
    const size_t selb = (size_t)(IB - 1) * nlay * n + ucol;
    for (int lay = lay0; lay < lay1; lay++) {
        Layer<R> L;
        load_layer<R>(A, lay, col, pc, L);
        const R ta = A.tauaer ? ldg(A.tauaer + (size_t)(IB - 1) * nlay * ld, L.ab) : (R)0;
        const bool laycld = CLD && ccol && A.laycloudy[(size_t)lay * n + ucol] != 0;
        Prep<R> P;
        BAND::template prep<R>(T, A, L, P);
        PfSel<R> sel{0, 1, (R)0};
        uint32_t wt[NW], wg[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) { wt[k] = 0; wg[k] = 0; }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R tau[W], pf[W];
            BAND::template eval<R, W>(T, L, P, q * W, tau, pf, &sel);
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                R odepth = secdiff * (tau[j] + ta);
                if (odepth < 0) odepth = 0;
                const R tblind = lw_pade<R>(odepth, bpade);
                const int itg = (int)(tblint * tblind + (R)0.5);
                int itp = itg;
                if (CLD && laycld) {
                    const R tc = ldg(taucmc_b, (((uint32_t)lay * (uint32_t)NG + (uint32_t)g) * (uint32_t)n + ucol) * (uint32_t)sizeof(R));
                    if (tc > 0) {      // cloud added to the DISCRETISED gas tau (:264-268)
                        const R odtot = ldg(T.tau_tbl, (uint32_t)itg * (uint32_t)sizeof(R)) + secdiff * tc;
                        const R tb2 = lw_pade<R>(odtot, bpade);
                        itp = (int)(tblint * tb2 + (R)0.5);
                    }
                }
                wt[g / 2] |= (uint32_t)itp << (16 * (g & 1));
                wg[g / 2] |= (uint32_t)itg << (16 * (g & 1));
            }
        }
        const size_t cell = lws_cell<NG>((int)npad, nlay, G0, ucol, lay);
        lws_store<NW>(A.s1 + cell, wt);
        if (CLD && lay <= ctop) lws_store<NW>(A.s2 + cell, wg);
        A.pfcode[selb + (size_t)lay * n] = (uint32_t)sel.kind | ((uint32_t)sel.js << 8);
        A.pffs[selb + (size_t)lay * n] = sel.fs;
    }
}

template <typename R, bool CLD>
__global__ void __launch_bounds__(256) k_lw_cells(LwArgs<R> A, LwDev<R> T)
{
    int bstart, bslot;
    if (!band_block(A.ncol, NB_LW * LWS_CHUNKS, bstart, bslot)) return;
    const int ib = LW_BAND_ORDER[bslot % NB_LW], chunk = bslot / NB_LW;
    if (!((A.band_mask >> ib) & 1u)) return;
    const int nclear = *A.nclear;
    const int bend = bstart + (int)blockDim.x < A.ncol ? bstart + (int)blockDim.x : A.ncol;
    if (CLD ? bend <= nclear : bstart >= nclear) return;
    const int col = bstart + threadIdx.x;
    if (col >= A.ncol) return;
    if (CLD ? col < nclear : col >= nclear) return;
    const int per = (A.nlay + LWS_CHUNKS - 1) / LWS_CHUNKS;
    const int lay0 = chunk * per, lay1 = lay0 + per < A.nlay ? lay0 + per : A.nlay;
    switch (ib) {
        case 1: lws_cells_body<R, Band1, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 2: lws_cells_body<R, Band2, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 3: lws_cells_body<R, Band3, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 4: lws_cells_body<R, Band4, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 5: lws_cells_body<R, Band5, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 6: lws_cells_body<R, Band6, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 7: lws_cells_body<R, Band7, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 8: lws_cells_body<R, Band8, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 9: lws_cells_body<R, Band9, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 10: lws_cells_body<R, Band10, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 11: lws_cells_body<R, Band11, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 12: lws_cells_body<R, Band12, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 13: lws_cells_body<R, Band13, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 14: lws_cells_body<R, Band14, CLD>(A, T, col, nclear, lay0, lay1); break;
        case 15: lws_cells_body<R, Band15, CLD>(A, T, col, nclear, lay0, lay1); break;
        default: lws_cells_body<R, Band16, CLD>(A, T, col, nclear, lay0, lay1); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_lw_sweep: rtrnmc's two recurrences of one band (NG g-points) of one column (LW/rrtmg_lw_rtrnmc.F90:245-379)
// ---------------------------------------------------------------------------------------------------
template <typename R, int NG, bool CLD>
GR_DEV void lws_sweep_body(const LwArgs<R> &A, const LwDev<R> &T, int ib, int col, int nclear, const typename Vec2<R>::T *luts,
                           const R *fra, const R *frb)
{
    constexpr int NW = NG / 2, S = pad4(NG);
    using R2 = typename Vec2<R>::T;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld, G0 = lw_band_g0(ib);
    auto lut_at = [&](int i) -> R2 {
        if constexpr (LwLutInLds<R>::value) return luts[i];
        else return ldg(T.lut, (uint32_t)i * (uint32_t)sizeof(R2));
    };
    const bool dudTs = A.dudTs != 0;
    const R sumfac = (R)0.5 * T.delwave[ib] * T.fluxfac;
    const uint32_t ucol = (uint32_t)col;
    const uint32_t cb = ucol * (uint32_t)sizeof(R);
    const int pc = ldg(A.perm, ucol * 4u);
    const uint32_t cba = (uint32_t)pc * (uint32_t)sizeof(R);
    const bool ccol = CLD && col >= nclear;
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    const size_t qs = (size_t)NB_LW * (nlay + 1) * n;
    R *const part = A.part + (size_t)(ib - 1) * (nlay + 1) * n;
#define PART(kind, lev, val) stg(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))
    const R semis = ldg(A.emis + (size_t)(ib - 1) * ld, cba);
    const R tb = ldg(A.tsfc, cba);
    const R plankbnd = semis * planck_at<R>(T.totplnk, ib, tb);
    const R dplankbnd = dudTs ? semis * planck_at<R>(T.totplnkderiv, ib, tb) : (R)0;
    const R reflect = (R)1. - semis;
    const size_t selb = (size_t)(ib - 1) * nlay * n + ucol;
    R rad[NG], radc[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) { rad[g] = 0; radc[g] = 0; }
    bool diverge = false;
    int ltop = -1;      // highest optically cloudy layer: where the clear / total streams part (:297-307)
    // ---- downward sweep, top layer -> surface ----
    // everything a layer needs from memory - its parked indices, its Planck-fraction selector, its two temperatures and the Planck
    // look-ups they lead to - is requested one layer ahead: the recurrence of a layer then waits for nothing
    struct LayIn { uint32_t wt[NW], wg[NW]; uint32_t code; R fs, blay, plk; bool cld; };
    auto request = [&](int lay, bool down, bool want_g, LayIn &b) {
        const size_t cell = lws_cell<NG>((int)npad, nlay, G0, ucol, lay);
        lws_load<NW>(A.s1 + cell, b.wt);
        if (want_g) lws_load<NW>(A.s2 + cell, b.wg);
        b.code = A.pfcode[selb + (size_t)lay * n];
        b.fs = A.pffs[selb + (size_t)lay * n];
        const uint32_t ab = ((uint32_t)lay * (uint32_t)ld + (uint32_t)pc) * (uint32_t)sizeof(R);
        b.blay = planck_at<R>(T.totplnk, ib, ldg(A.tlay, ab));
        b.plk = planck_at<R>(T.totplnk, ib, down ? ldg(A.tlev, ab) : ldg(A.tlev + (size_t)ld, ab));      // lower (down) / upper (up) level
        b.cld = CLD && ccol && A.laycloudy[(size_t)lay * n + ucol] != 0;
    };
    auto fracs = [&](const LayIn &b, R (&pf)[NG]) {
        const int kind = (int)(b.code & 255u), js = (int)(b.code >> 8);
        const R *tab = (kind == 2 || kind == 4) ? frb : fra;
        if (kind == 0) {
#pragma unroll
            for (int g = 0; g < NG; g++) pf[g] = 0;
        } else if (kind <= 2) {
#pragma unroll
            for (int g = 0; g < NG; g++) pf[g] = tab[g];
        } else {
            const R *r0 = tab + (js - 1) * S, *r1 = r0 + S;
#pragma unroll
            for (int g = 0; g < NG; g++) { const R a = r0[g], b_ = r1[g]; pf[g] = a + b.fs * (b_ - a); }
        }
    };
    LayIn nx;
    // (a column's gas-only indices are read from its first optically cloudy layer down: the wave asks for them with the layer's other
    // requests once one of its columns has diverged - the layer in which that happens fetches its own behind the test)
    request(nlay - 1, true, false, nx);
    bool wdv_prev = false;
#pragma nounroll
    for (int lay = nlay - 1; lay >= 0; lay--) {
        LayIn cur = nx;
        const bool laycld = cur.cld;
        if (CLD && laycld && !diverge) { diverge = true; ltop = lay; }      // before this layer's clear-sky update
        const bool wdv = CLD && __ballot(diverge) != 0;
        if (CLD && wdv && !wdv_prev) lws_load<NW>(A.s2 + lws_cell<NG>((int)npad, nlay, G0, ucol, lay), cur.wg);
        wdv_prev = wdv;
        if (lay > 0) request(lay - 1, true, wdv, nx);
        const R blay = cur.blay, plk_dn = cur.plk;
        const R dplankdn = plk_dn - blay;
        R pf[NG];
        fracs(cur, pf);
        R dsum = 0, dcsum = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int itp = (int)((cur.wt[g / 2] >> (16 * (g & 1))) & 0xFFFFu);
            const R2 e = lut_at(itp);
            const R atot = (R)1. - e.x;
            const R bbd = pf[g] * (blay + e.y * dplankdn);
            rad[g] = rad[g] + (bbd - rad[g]) * atot;
            dsum = dsum + sumfac * rad[g];
            if (CLD) {
                R rc = rad[g];
                if (wdv) {
                    const int itg = (int)((cur.wg[g / 2] >> (16 * (g & 1))) & 0xFFFFu);
                    const R2 eg = lut_at(diverge ? itg : itp);
                    const R agas = (R)1. - eg.x, bbdgas = pf[g] * (blay + eg.y * dplankdn);
                    rc = radc[g] + (bbdgas - radc[g]) * agas;
                }
                radc[g] = diverge ? rc : rad[g];
                dcsum = dcsum + sumfac * radc[g];
            }
        }
        PART(0, lay, dsum);
        if (CLD && ccol) PART(1, lay, dcsum);
    }
    PART(0, nlay, 0);      // TOA downward flux is zero; written so the reduce kernel can sum unconditionally
    if (CLD && ccol) PART(1, nlay, 0);

    // ---- upward sweep, surface -> top (:336-379) ----
    int wtop = -1;         // highest layer in which a column of this wave keeps a gas-only index of its own (wave-uniform)
    if (CLD) {
        for (int l = nlay - 1; l >= 0; l--)
            if (__ballot(ccol && diverge && ltop == l) != 0) { wtop = l; break; }
    }
    R dlu[NG], dclu[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) { dlu[g] = 0; dclu[g] = 0; }
    request(0, false, CLD && 0 <= wtop, nx);
#pragma nounroll
    for (int lay = 0; lay < nlay; lay++) {
        R u0 = 0, uc0 = 0, du0 = 0, duc0 = 0;
        R usum = 0, ucsum = 0, dusum = 0, ducsum = 0;
        const LayIn cur = nx;
        const bool rdg = CLD && lay <= wtop;
        if (lay + 1 < nlay) request(lay + 1, false, CLD && lay + 1 <= wtop, nx);
        const bool own = CLD && ccol && diverge && lay <= ltop;      // above ltop the layer is clear for every g-point: gas == total
        const R blay = cur.blay;
        const R dplankup = cur.plk - blay;
        R pf[NG];
        fracs(cur, pf);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (lay == 0) {      // surface: emission + reflection turn the downward radiance into the upward one (:319-333)
                const R rad0 = pf[g] * plankbnd;
                rad[g] = rad0 + reflect * rad[g];
                dlu[g] = pf[g] * dplankbnd;
                u0 = u0 + sumfac * rad[g];
                du0 = du0 + sumfac * dlu[g];
                if (CLD) {
                    radc[g] = rad0 + reflect * radc[g];
                    dclu[g] = dlu[g];
                    uc0 = uc0 + sumfac * radc[g];
                    duc0 = duc0 + sumfac * dclu[g];
                }
            }
            const int itp = (int)((cur.wt[g / 2] >> (16 * (g & 1))) & 0xFFFFu);
            const R2 e1 = lut_at(itp);
            const R a1 = (R)1. - e1.x, b1 = pf[g] * (blay + e1.y * dplankup);
            rad[g] = rad[g] + (b1 - rad[g]) * a1;
            dlu[g] = dlu[g] - dlu[g] * a1;
            usum = usum + sumfac * rad[g];
            dusum = dusum + sumfac * dlu[g];
            if (CLD) {
                int itc = itp;
                if (rdg) { const int itg = (int)((cur.wg[g / 2] >> (16 * (g & 1))) & 0xFFFFu); itc = own ? itg : itp; }
                const R2 e2 = lut_at(itc);
                const R gx = (R)1. - e2.x, gy = pf[g] * (blay + e2.y * dplankup);
                const R rc = radc[g] + (gy - radc[g]) * gx, dc = dclu[g] - dclu[g] * gx;
                radc[g] = diverge ? rc : rad[g];
                dclu[g] = diverge ? dc : dlu[g];
                ucsum = ucsum + sumfac * radc[g];
                ducsum = ducsum + sumfac * dclu[g];
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

template <typename R, bool CLD>
__global__ void __launch_bounds__((lws_block<R, CLD>)) k_lw_sweep(LwArgs<R> A, LwDev<R> T)
{
    // (column block, band) = (blockIdx.x, blockIdx.y), heaviest band first (band_block, lw_kernels.hpp)
    const int bstart = (int)(blockIdx.x * blockDim.x), ib = LW_BAND_ORDER[blockIdx.y];
    if (!((A.band_mask >> ib) & 1u)) return;
    const int nclear = *A.nclear;
    const int bend = bstart + (int)blockDim.x < A.ncol ? bstart + (int)blockDim.x : A.ncol;
    if (CLD ? bend <= nclear : bstart >= nclear) return;
    using R2 = typename Vec2<R>::T;
    extern __shared__ __align__(16) unsigned char lws_lds[];
    const R2 *luts = nullptr;
    const BandTab<R> &B = T.b[ib];
    const R *fra = B.fracrefa, *frb = B.fracrefb;
    if constexpr (LwLutInLds<R>::value) {
        R2 *const l = reinterpret_cast<R2 *>(lws_lds);
        for (int i = threadIdx.x; i <= NTBL; i += (int)blockDim.x) l[i] = ldg(T.lut, (uint32_t)i * (uint32_t)sizeof(R2));
        const int S = pad4(lw_band_ng(ib)), na = lw_nfraca(ib), nb = lw_nfracb(ib);
        R *const sm = reinterpret_cast<R *>(lws_lds + LW_LDS_LUT);
        for (int i = threadIdx.x; i < (na + nb) * S; i += (int)blockDim.x) sm[i] = i < na * S ? B.fracrefa[i] : B.fracrefb[i - na * S];
        __syncthreads();
        luts = l; fra = sm; frb = sm + na * S;
    }
    const int col = bstart + threadIdx.x;
    if (col >= A.ncol) return;
    if (CLD ? col < nclear : col >= nclear) return;
    switch (lw_band_ng(ib)) {
        case 2: lws_sweep_body<R, 2, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 4: lws_sweep_body<R, 4, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 6: lws_sweep_body<R, 6, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 8: lws_sweep_body<R, 8, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 10: lws_sweep_body<R, 10, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 12: lws_sweep_body<R, 12, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        case 14: lws_sweep_body<R, 14, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
        default: lws_sweep_body<R, 16, CLD>(A, T, ib, col, nclear, luts, fra, frb); break;
    }
}

}  // namespace geosrad
