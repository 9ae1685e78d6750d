// lw_cols_kernels.hpp -- RRTMG_LW band sweeps with every (layer, g-point) intermediate ON CHIP (gfx950 / CDNA4).
//
// Mapping (the north-star's: the vertical dimension goes across lanes, the per-cell state of a column lives in LDS):
//   block        = C atmospheric columns of one class (cloud-free | cloudy, k_partition), ALL 16 bands, one after the other
//   worker t     = (c = t % C, lay = t / C): one (layer, column) pair; its setcoef record stays in registers for the whole kernel,
//                  and it owns the flux accumulators of level `lay` (downward) and `lay + 1` (upward)
//   sweep wave   = one extra wavefront: lane = (column, g-point of the stage[, sky])
// The 16 bands are cut into STAGES of GR g-points (8 cloud-free, 4 cloudy: two skies per g-point).  A stage goes through
//   phase 1 (workers, no vertical dependence): taumol of the worker's layer for the stage's g-points (rows of the k-distribution
//            tables, 4 g-points per fetch), Pade index, transmittance look-up -> the layer's (absorptivity, source down, source up)
//            per g-point, written to LDS planes [g][layer][column]
//   phase 2 (sweep wave): the two first-order recurrences of rtrnmc down and up the column, read from and written back to the
//            same planes (now: radiance at every level)
//   phase 3 (workers): each worker adds the stage's radiances at its own two levels to its band sums, g-point by g-point
// and the stages are software-pipelined over two plane buffers with ONE barrier per stage: between two barriers the workers run
// phase 3 of stage i-2 and then phase 1 of stage i on buffer i % 2 (a worker reads and rewrites only its own cells), while the
// sweep wave runs phase 2 of stage i-1 on the other buffer.  At the end each worker writes the fluxes of its levels: no per-cell
// scratch, no band partials and no reduction kernel in HBM.  HBM traffic = the API inputs (re-read per band from L2) + the API
// outputs.  What is computed is rtrnmc / taumol statement by statement as in lw_kernels.hpp (whose band bodies `prep` / `eval`
// are used unchanged); the order of every floating-point sum is that of k_lw_bands + k_lw_reduce.
//   LW/rrtmg_lw_rtrnmc.F90:164-388, LW/rrtmg_lw_taumol.F90:155-3126.
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {

template <bool CLD> struct LwcK { static constexpr int GR = CLD ? 4 : 8, NST = CLD ? 2 : 1, NPL = 3 * NST; };

// LDS layout of a plane buffer: [plane][g-point][column][layer], the layers of one (plane, g-point, column) contiguous and padded to
// LS = 4 x odd reals: the sweep wave (lane i = column + C x g-point, i x LS reals apart) reads four layers per 16-byte access without
// bank conflicts, and so do the workers (lane = column fastest, then layer) with their 4-byte accesses.  Padding layers hold a zero
// absorptivity: the recurrences pass through them unchanged.
__host__ __device__ constexpr int lwc_layer_stride(int nlay) { return 4 * (((nlay + 3) / 4) | 1); }
// one plane buffer (reals): NPL x GR x C layer runs, then per (g-point, column): Planck fraction of the lowest layer, upward radiance at
// the surface per sky, its derivative
template <bool CLD> __host__ __device__ constexpr size_t lwc_buf_reals(int nlay, int C)
{
    return (size_t)LwcK<CLD>::NPL * LwcK<CLD>::GR * C * lwc_layer_stride(nlay) + (((size_t)LwcK<CLD>::GR * C * (2 + LwcK<CLD>::NST) + 3) & ~(size_t)3);
}
// + (cloudy instantiation) the workers' ten flux totals [10][workers]: touched once per band, they would cost ten registers each
template <typename R, bool CLD> __host__ __device__ constexpr size_t lwc_lds_bytes(int nlay, int C)
{
    return (2 * lwc_buf_reals<CLD>(nlay, C) + (CLD ? (size_t)10 * ((C * nlay + 63) / 64 * 64) : 0)) * sizeof(R);
}
// columns per block: C x nlay workers must fit 576 threads and the two plane buffers 112 KiB of LDS
#ifndef LWC_CMAX
#define LWC_CMAX 8
#endif
template <typename R> __host__ __device__ constexpr int lwc_columns_per_block(int nlay)
{
    return sizeof(R) == 4 ? (nlay <= 72 ? LWC_CMAX : (nlay <= 144 ? LWC_CMAX / 2 : LWC_CMAX / 4))
                          : (nlay <= 72 ? LWC_CMAX / 2 : (nlay <= 144 ? LWC_CMAX / 4 : 1));
}
constexpr int LWC_MAXT = 640;        // 576 workers + the sweep wave

// a stage's band-level facts, uniform over the block (carried from the interval that produced the stage to the two that finish it)
template <typename R> struct LwcStage { R sumfac; int ib, nvalid; bool band_end; };
template <typename R> struct LwcAcc { R d, u, u0, du, du0, dc, uc, uc0, duc, duc0; };

#define LWC_CELL(buf, plane, gl, lay_, c_) ((buf) + ((size_t)(((plane) * GR + (gl)) * C + (c_)) * PS + (size_t)(lay_)))

// ---- phase 2: lane = (column c, g-point gl, sky s): down the column, turn at the surface, up again (:245-379) ----------------------
// planes of sky s: 3s + 0 absorptivity (after the up sweep: d(upward radiance)/dTs), 3s + 1 source down (then: downward radiance at
// the layer's lower level), 3s + 2 source up (then: upward radiance at the layer's upper level).
// A lane's layers are contiguous: the sweeps read and write them four at a time (one 16-byte LDS access), two such vectors per
// step of a ping-pong that requests the next eight layers before it works on the current eight (a step of the recurrence is two
// dependent operations, an LDS read fifty times that).  No bounds tests: padding layers have zero absorptivity.
template <typename R> struct alignas(4 * sizeof(R)) LwcVec4 { R v[4]; };
template <typename R, bool CLD, int C>
GR_DEV void lwc_sweep(const LwArgs<R> &A, const LwDev<R> &T, R *__restrict__ buf, int PS, int nlay, const LwcStage<R> &st, int c, int gl,
                      int s, int pc)
{
    using K = LwcK<CLD>;
    using V4 = LwcVec4<R>;
    constexpr int GR = K::GR, NST = K::NST;
    if (gl >= st.nvalid) return;
    const bool dudTs = A.dudTs != 0;
    // surface terms of this band (:319-333)
    const R semis = A.emis[(size_t)(st.ib - 1) * A.ld + pc];
    const R tb = A.tsfc[pc];
    const R plankbnd = semis * planck_at<R>(T.totplnk, st.ib, tb);
    const R dplankbnd = dudTs ? semis * planck_at<R>(T.totplnkderiv, st.ib, tb) : (R)0;
    const R reflect = (R)1. - semis;
    V4 *const pa = reinterpret_cast<V4 *>(LWC_CELL(buf, 3 * s + 0, gl, 0, c)), *const pd = reinterpret_cast<V4 *>(LWC_CELL(buf, 3 * s + 1, gl, 0, c)),
       *const pu = reinterpret_cast<V4 *>(LWC_CELL(buf, 3 * s + 2, gl, 0, c));
    R *const pf0 = buf + (size_t)K::NPL * GR * C * PS, *const u0 = pf0 + GR * C, *const dl0 = u0 + NST * GR * C;
    const int nv = (nlay + 3) / 4;           // 4-layer vectors, the last one possibly with padding layers
    R rad = 0;
    // one 4-layer vector of the downward sweep (layers 4v+3 .. 4v) / of the upward sweep (4v .. 4v+3)
    auto down4 = [&](V4 &a, V4 &b) {
#pragma unroll
        for (int i = 3; i >= 0; i--) { rad = rad + (b.v[i] - rad) * a.v[i]; b.v[i] = rad; }
    };
    R dl = 0;
    auto up4 = [&](V4 &a, V4 &b) {
#pragma unroll
        for (int i = 0; i < 4; i++) { rad = rad + (b.v[i] - rad) * a.v[i]; b.v[i] = rad; if (dudTs) { dl = dl - dl * a.v[i]; a.v[i] = dl; } }
    };
    // ---- downward: vectors nv-1 .. 0 ----
    {
        int v = nv - 1;
        if (nv & 1) { V4 a = pa[v], b = pd[v]; down4(a, b); pd[v] = b; v--; }
        V4 xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1;
        if (v >= 1) { xa0 = pa[v]; xb0 = pd[v]; xa1 = pa[v - 1]; xb1 = pd[v - 1]; }
        while (v >= 1) {
            const bool more = v >= 3;
            if (more) { ya0 = pa[v - 2]; yb0 = pd[v - 2]; ya1 = pa[v - 3]; yb1 = pd[v - 3]; }
            down4(xa0, xb0); down4(xa1, xb1);
            pd[v] = xb0; pd[v - 1] = xb1;
            v -= 2;
            if (!more) break;
            if (v >= 3) { xa0 = pa[v - 2]; xb0 = pd[v - 2]; xa1 = pa[v - 3]; xb1 = pd[v - 3]; }
            down4(ya0, yb0); down4(ya1, yb1);
            pd[v] = yb0; pd[v - 1] = yb1;
            v -= 2;
        }
    }
    // ---- surface: emission + reflection turn the downward radiance into the upward one (:319-333) ----
    const R pfs = pf0[gl * C + c];
    const R rad0 = pfs * plankbnd;
    rad = rad0 + reflect * rad;
    dl = pfs * dplankbnd;
    u0[s * GR * C + gl * C + c] = rad;
    if (s == 0) dl0[gl * C + c] = dl;
    // ---- upward: vectors 0 .. nv-1 ----
    {
        int v = 0;
        const int nv2 = nv & ~1;
        V4 xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1;
        if (nv2 >= 2) { xa0 = pa[0]; xb0 = pu[0]; xa1 = pa[1]; xb1 = pu[1]; }
        while (v < nv2) {
            const bool more = v + 2 < nv2;
            if (more) { ya0 = pa[v + 2]; yb0 = pu[v + 2]; ya1 = pa[v + 3]; yb1 = pu[v + 3]; }
            up4(xa0, xb0); up4(xa1, xb1);
            pu[v] = xb0; pu[v + 1] = xb1;
            if (dudTs) { pa[v] = xa0; pa[v + 1] = xa1; }
            v += 2;
            if (!more) break;
            if (v + 2 < nv2) { xa0 = pa[v + 2]; xb0 = pu[v + 2]; xa1 = pa[v + 3]; xb1 = pu[v + 3]; }
            up4(ya0, yb0); up4(ya1, yb1);
            pu[v] = yb0; pu[v + 1] = yb1;
            if (dudTs) { pa[v] = ya0; pa[v + 1] = ya1; }
            v += 2;
        }
        if (nv & 1) {
            V4 a = pa[v], b = pu[v];
            up4(a, b);
            pu[v] = b;
            if (dudTs) pa[v] = a;
        }
    }
    // the padding layers' absorptivity stays zero for the next stage that uses this buffer
    if (dudTs && (nlay & 3)) {
        R *const z = reinterpret_cast<R *>(pa);
        for (int l = nlay; l < 4 * nv; l++) z[l] = 0;
    }
}

// ---- phase 3: the worker's two levels: lower level of its layer (downward flux), upper level (upward); layer 0 also the surface ----
template <typename R> struct LwcSums { R d, u, u0, du, du0, dc, uc, uc0, duc, duc0; };
template <typename R, bool CLD, int C>
GR_DEV void lwc_collect(const LwArgs<R> &A, const LwOut<R> &O, const R *__restrict__ buf, int PS, int nlay, const LwcStage<R> &st, int c,
                        int lay, int pc, bool valid, LwcSums<R> &b, LwcAcc<R> &acc, R *__restrict__ accl, int accs)
{
    using K = LwcK<CLD>;
    constexpr int GR = K::GR, NST = K::NST;
    const bool dudTs = A.dudTs != 0;
    const R sumfac = st.sumfac;
    const R *const pf0 = buf + (size_t)K::NPL * GR * C * PS, *const u0 = pf0 + GR * C, *const dl0 = u0 + NST * GR * C;
#pragma unroll
    for (int gl = 0; gl < GR; gl++) {
        if (gl < st.nvalid) {
            b.d = b.d + sumfac * *LWC_CELL(buf, 1, gl, lay, c);
            b.u = b.u + sumfac * *LWC_CELL(buf, 2, gl, lay, c);
            if (dudTs) b.du = b.du + sumfac * *LWC_CELL(buf, 0, gl, lay, c);
            if (CLD) {
                b.dc = b.dc + sumfac * *LWC_CELL(buf, 4, gl, lay, c);
                b.uc = b.uc + sumfac * *LWC_CELL(buf, 5, gl, lay, c);
                if (dudTs) b.duc = b.duc + sumfac * *LWC_CELL(buf, 3, gl, lay, c);
            }
            if (lay == 0) {
                b.u0 = b.u0 + sumfac * u0[gl * C + c];
                b.du0 = b.du0 + sumfac * dl0[gl * C + c];
                if (CLD) { b.uc0 = b.uc0 + sumfac * u0[GR * C + gl * C + c]; b.duc0 = b.duc0 + sumfac * dl0[gl * C + c]; }
            }
        }
    }
    if (st.band_end) {
        // band OLR (:382-385): the upward band flux at the top level
        if (valid && lay == nlay - 1 && O.band_output[st.ib - 1]) {
            O.olrb[(size_t)(O.col0 + pc) * NB_LW + (st.ib - 1)] = b.u;
            if (dudTs) O.dolrb_dTs[(size_t)(O.col0 + pc) * NB_LW + (st.ib - 1)] = b.du;
        }
        // band sums -> totals, in band order like k_lw_reduce
        if (CLD) {
            accl[0 * accs] += b.d; accl[1 * accs] += b.u; accl[2 * accs] += b.u0; accl[5 * accs] += b.dc; accl[6 * accs] += b.uc; accl[7 * accs] += b.uc0;
            if (dudTs) { accl[3 * accs] += b.du; accl[4 * accs] += b.du0; accl[8 * accs] += b.duc; accl[9 * accs] += b.duc0; }
        } else {
            acc.d += b.d; acc.u += b.u; acc.u0 += b.u0;
            if (dudTs) { acc.du += b.du; acc.du0 += b.du0; }
        }
        b = LwcSums<R>{};
    }
}

// what a worker keeps for the whole kernel
template <typename R> struct LwcWorker {
    int c, lay, pos, pc;                // the worker's column: compacted position (workspace arrays) and API column
    bool act, valid;                    // act: lay < nlay (the workers are padded to whole wavefronts); valid: the column exists
    Layer<R> L;
};

// ---- one band: the worker's layer; every stage of the band, each followed by the block's barrier --------------------------------------
// I0 = index of the band's first stage in the kernel's stage sequence (sets the plane buffer of each stage)
template <typename R, typename BAND, bool CLD, bool DBG, int C, int I0>
GR_DEV void lwc_band(const LwArgs<R> &A, const LwOut<R> &O, const LwDev<R> &T, R *__restrict__ lds, int PS, int nlay, bool worker,
                     const LwcWorker<R> &S, int p2c, int p2gl, int p2s, int p2pc, LwcStage<R> (&ring)[2], LwcSums<R> &bsum, LwcAcc<R> &acc,
                     R *__restrict__ accl, int accs)
{
    using K = LwcK<CLD>;
    using R2 = typename Vec2<R>::T;
    constexpr int NG = BAND::NG, IB = BAND::IB, G0 = BAND::G0;
    constexpr int GR = K::GR;
    constexpr int NR = (NG + GR - 1) / GR;
    const int ld = A.ld, n = A.ncol;
    const size_t bufsz = lwc_buf_reals<CLD>(nlay, C);
    const R bpade = T.bpade, tblint = (R)NTBL;
    auto lut_at = [&](int i) -> R2 { return ldg(T.lut, (uint32_t)i * (uint32_t)sizeof(R2)); };

    // ---- the worker's layer, this band --------------------------------------------------------------------------------------
    Prep<R> P;
    R secdiff = 0, ta = 0, blay = 0, dplankup = 0, dplankdn = 0;
    bool laycld = false;
    if (worker && S.act) {
        BAND::template prep<R>(T, A, S.L, P);
        // diffusivity angle (:177-186)
        secdiff = (R)1.66;
        if (!(IB == 1 || IB == 4 || IB >= 10)) {
            constexpr double a0[17] = {0, 1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
            constexpr double a1[17] = {0, 0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
            constexpr double a2[17] = {0, 0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
            secdiff = (R)a0[IB] + (R)a1[IB] * gr_exp<R>((R)a2[IB] * A.pwvcm[S.pc]);
            secdiff = secdiff > (R)1.80 ? (R)1.80 : (secdiff < (R)1.50 ? (R)1.50 : secdiff);
        }
        ta = A.tauaer ? ldg(A.tauaer + (size_t)(IB - 1) * nlay * ld, S.L.ab) : (R)0;
        blay = planck_at<R>(T.totplnk, IB, ldg(A.tlay, S.L.ab));
        const R plk_dn = planck_at<R>(T.totplnk, IB, ldg(A.tlev, S.L.ab));
        const R plk_up = planck_at<R>(T.totplnk, IB, ldg(A.tlev + ld, S.L.ab));
        dplankup = plk_up - blay; dplankdn = plk_dn - blay;
        if (CLD) laycld = A.laycloudy[(size_t)S.lay * n + S.pos] != 0;
    }
    const R *const taucmc_b = CLD ? A.taucmc + (size_t)G0 * nlay * n : nullptr;
    // gas optical depth and Planck fraction of ALL the band's g-points at once: a worker fetches each of its table rows whole and once
    // (16-byte pieces at immediate offsets of one address), instead of a quarter of the row per group of four g-points
    constexpr int W = NG >= 4 ? 4 : 2;          // g-points per evaluation of the k-distribution: one 16-byte piece of each table row

#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int I = I0 + r;                                   // compile-time after unrolling
        R *const bcur = lds + (size_t)(I & 1) * bufsz;          // stage I (phase 1 now) and stage I-2 (phase 3 now)
        R *const bprev = lds + (size_t)((I + 1) & 1) * bufsz;   // stage I-1 (phase 2 now)
        if (worker) {
            if (S.act) {
                if (I >= 2) lwc_collect<R, CLD, C>(A, O, bcur, PS, nlay, ring[I & 1], S.c, S.lay, S.pc, S.valid, bsum, acc, accl, accs);
                R *const pf0 = bcur + (size_t)K::NPL * GR * C * PS;
                // one group of W g-points at a time, table rows to LDS cells, before the next group's rows are requested
#pragma unroll
                for (int q = 0; q < GR / W; q++) {
                    const int go = r * GR + q * W;
                    if (go >= NG) continue;
                    R tau[W], pf[W];
                    BAND::template eval<R, W>(T, S.L, P, go, tau, pf);
                    __builtin_amdgcn_sched_barrier(0);
                    int itg[W];
                    R2 eg[W];
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const int g = go + j;
                        itg[j] = 0; eg[j].x = 0; eg[j].y = 0;
                        if (g >= NG) continue;          // padding of the band's last group
                        if (DBG) {
                            const size_t o = ((size_t)S.pc * NG_LW + (G0 + g)) * nlay + S.lay;   // Fortran (nlay,140,ncol)
                            if (S.valid) { A.dbg_taug[o] = tau[j] + ta; A.dbg_pfracs[o] = pf[j]; }
                        }
                        R odepth = secdiff * (tau[j] + ta);
                        if (odepth < 0) odepth = 0;
                        const R tblind = lw_pade<R>(odepth, bpade);
                        itg[j] = (int)(tblint * tblind + (R)0.5);
                        eg[j] = lut_at(itg[j]);
                    }
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const int g = go + j, gl = q * W + j;
                        if (g >= NG) continue;
                        const R agas = (R)1. - eg[j].x, tfacgas = eg[j].y;
                        const R bbdgas = pf[j] * (blay + tfacgas * dplankdn);
                        const R bbugas = pf[j] * (blay + tfacgas * dplankup);
                        *LWC_CELL(bcur, 0, gl, S.lay, S.c) = agas; *LWC_CELL(bcur, 1, gl, S.lay, S.c) = bbdgas; *LWC_CELL(bcur, 2, gl, S.lay, S.c) = bbugas;
                        if (CLD) { *LWC_CELL(bcur, 3, gl, S.lay, S.c) = agas; *LWC_CELL(bcur, 4, gl, S.lay, S.c) = bbdgas; *LWC_CELL(bcur, 5, gl, S.lay, S.c) = bbugas; }
                        if (S.lay == 0) pf0[gl * C + S.c] = pf[j];
                    }
                    // cloudy cells (few layers of a column have any): the total-sky values replace the gas-only ones just written
                    if (CLD && laycld) {
#pragma unroll
                        for (int j = 0; j < W; j++) {
                            const int g = go + j, gl = q * W + j;
                            if (g >= NG) continue;
                            const R tc = taucmc_b[((size_t)S.lay * NG + g) * n + S.pos];
                            if (tc > 0) {
                                // cloud added to the DISCRETISED gas optical depth (:264-268)
                                const R odtot = ldg(T.tau_tbl, (uint32_t)itg[j] * (uint32_t)sizeof(R)) + secdiff * tc;
                                const R tb2 = lw_pade<R>(odtot, bpade);
                                const int ittot = (int)(tblint * tb2 + (R)0.5);
                                const R2 e2 = lut_at(ittot);
                                *LWC_CELL(bcur, 0, gl, S.lay, S.c) = (R)1. - e2.x;
                                *LWC_CELL(bcur, 1, gl, S.lay, S.c) = pf[j] * (blay + e2.y * dplankdn);
                                *LWC_CELL(bcur, 2, gl, S.lay, S.c) = pf[j] * (blay + e2.y * dplankup);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (I >= 1) {
            lwc_sweep<R, CLD, C>(A, T, bprev, PS, nlay, ring[(I + 1) & 1], p2c, p2gl, p2s, p2pc);
        }
        // this stage's facts replace those of stage I-2 (whose last user, the collect above, is done)
        ring[I & 1].sumfac = (R)0.5 * T.delwave[IB] * T.fluxfac;
        ring[I & 1].ib = IB;
        ring[I & 1].nvalid = (NG - r * GR) < GR ? (NG - r * GR) : GR;
        ring[I & 1].band_end = r == NR - 1;
        __syncthreads();
    }
}

// number of stages of bands 1 .. ib - 1
template <bool CLD> __host__ __device__ constexpr int lwc_stages_before(int ib)
{
    int s = 0;
    for (int b = 1; b < ib; b++) s += (lw_band_ng(b) + LwcK<CLD>::GR - 1) / LwcK<CLD>::GR;
    return s;
}

// grid: 8 * ceil(ceil(ncol / C) / 8) blocks.  The hardware deals consecutive blocks to the 8 XCDs in turn; block b therefore takes the
// column group (b % 8) * per + b / 8, so that one XCD (one L2) works through a contiguous range of columns: neighbouring groups,
// which share the 128-byte lines of the API arrays, meet in the same L2 at about the same time.
// blockDim = NWT + 64: NWT = C * nlay workers rounded up to whole wavefronts, then the sweep wave.
template <typename R, bool CLD, bool DBG, int C>
__global__ void __launch_bounds__(LWC_MAXT) k_lw_cols(LwArgs<R> A, LwOut<R> O, LwDev<R> T)
{
    extern __shared__ __align__(16) unsigned char lwc_lds_raw[];
    R *const lds = reinterpret_cast<R *>(lwc_lds_raw);
    using K = LwcK<CLD>;
    constexpr int GR = K::GR;
    const int nclear = *A.nclear;
    // this class's range of compacted positions (the DBG instantiation runs every column through the general body)
    const int base = (CLD && !DBG) ? nclear : 0, end = CLD ? A.ncol : nclear;
    const int ngroups = (end - base + C - 1) / C;
    const int per = (ngroups + 7) / 8;
    const int grp = (int)(blockIdx.x % 8u) * per + (int)(blockIdx.x / 8u);
    if (grp >= ngroups || (int)(blockIdx.x / 8u) >= per) return;              // uniform: before any barrier
    const int nlay = A.nlay;
    const int PS = lwc_layer_stride(nlay);            // reals between the layer runs of neighbouring (g-point, column) pairs
    const int t = (int)threadIdx.x;
    const int nwt = (int)blockDim.x - 64;
    const bool worker = t < nwt;

    LwcWorker<R> S{};
    int p2c = 0, p2gl = 0, p2s = 0, p2pc = 0;
    if (worker) {
        S.c = t % C; S.lay = t / C;
        S.act = S.lay < nlay;
        int pos = base + grp * C + S.c;
        S.valid = pos < end;
        if (!S.valid) pos = end - 1;                // a stand-in column for the tail of the last group: computed, never written
        S.pos = pos;
        S.pc = A.perm[pos];
        if (S.act) load_layer<R>(A, S.lay, pos, S.pc, S.L);
    } else {
        const int p = t - nwt;
        p2c = p % C; p2gl = (p / C) % GR; p2s = p / (C * GR);
        if (p2s >= K::NST) p2gl = GR;               // lanes beyond the (column, g-point, sky) triples: never valid
        int pos = base + grp * C + p2c;
        if (pos >= end) pos = end - 1;
        p2pc = A.perm[pos];
    }
    // padding layers (zero absorptivity) and everything else start from zero
    {
        const size_t tot = 2 * lwc_buf_reals<CLD>(nlay, C) + (CLD ? (size_t)10 * nwt : 0);
        for (size_t i = (size_t)t; i < tot; i += blockDim.x) lds[i] = 0;
        __syncthreads();
    }
    LwcAcc<R> acc{};
    R *const accl = lds + 2 * lwc_buf_reals<CLD>(nlay, C) + t;      // cloudy instantiation: this worker's totals, [10][nwt] (zeroed above)
    const int accs = nwt;
    LwcSums<R> bsum{};
    LwcStage<R> ring[2] = {};
#define LWC_BAND(B, ib) lwc_band<R, B, CLD, DBG, C, lwc_stages_before<CLD>(ib)>(A, O, T, lds, PS, nlay, worker, S, p2c, p2gl, p2s, p2pc, ring, bsum, acc, accl, nwt)
    LWC_BAND(Band1, 1); LWC_BAND(Band2, 2); LWC_BAND(Band3, 3); LWC_BAND(Band4, 4);
    LWC_BAND(Band5, 5); LWC_BAND(Band6, 6); LWC_BAND(Band7, 7); LWC_BAND(Band8, 8);
    LWC_BAND(Band9, 9); LWC_BAND(Band10, 10); LWC_BAND(Band11, 11); LWC_BAND(Band12, 12);
    LWC_BAND(Band13, 13); LWC_BAND(Band14, 14); LWC_BAND(Band15, 15); LWC_BAND(Band16, 16);
#undef LWC_BAND
    // drain the pipeline: phase 2 of the last stage, phase 3 of the last two
    constexpr int N = lwc_stages_before<CLD>(17);
    const size_t bufsz = lwc_buf_reals<CLD>(nlay, C);
    if (worker) { if (S.act) lwc_collect<R, CLD, C>(A, O, lds + (size_t)(N & 1) * bufsz, PS, nlay, ring[N & 1], S.c, S.lay, S.pc, S.valid, bsum, acc, accl, accs); }
    else lwc_sweep<R, CLD, C>(A, T, lds + (size_t)((N + 1) & 1) * bufsz, PS, nlay, ring[(N + 1) & 1], p2c, p2gl, p2s, p2pc);
    __syncthreads();
    if (!worker || !S.act) return;
    lwc_collect<R, CLD, C>(A, O, lds + (size_t)((N + 1) & 1) * bufsz, PS, nlay, ring[(N + 1) & 1], S.c, S.lay, S.pc, S.valid, bsum, acc, accl, accs);

    // ---- the worker's levels of the API outputs (LW/rrtmg_lw_rad.F90:587-605) --------------------------------------------------
    if (!S.valid) return;
    const int ld = A.ld, lay = S.lay, pc = S.pc;
    const bool dudTs = A.dudTs != 0;
    const size_t lo = (size_t)lay * ld + pc, up = lo + ld;
    if (CLD) {
        acc.d = accl[0 * accs]; acc.u = accl[1 * accs]; acc.u0 = accl[2 * accs]; acc.du = accl[3 * accs]; acc.du0 = accl[4 * accs];
        acc.dc = accl[5 * accs]; acc.uc = accl[6 * accs]; acc.uc0 = accl[7 * accs]; acc.duc = accl[8 * accs]; acc.duc0 = accl[9 * accs];
    }
    O.dflx[lo] = acc.d; O.uflx[up] = acc.u;
    O.dflxc[lo] = CLD ? acc.dc : acc.d; O.uflxc[up] = CLD ? acc.uc : acc.u;
    if (dudTs) { O.duflx_dTs[up] = acc.du; O.duflxc_dTs[up] = CLD ? acc.duc : acc.du; }
    if (lay == 0) {
        O.uflx[pc] = acc.u0; O.uflxc[pc] = CLD ? acc.uc0 : acc.u0;
        if (dudTs) { O.duflx_dTs[pc] = acc.du0; O.duflxc_dTs[pc] = CLD ? acc.duc0 : acc.du0; }
    }
    if (lay == nlay - 1) { O.dflx[up] = 0; O.dflxc[up] = 0; }      // no downward longwave at the top of the atmosphere
}
#undef LWC_CELL

}  // namespace geosrad
