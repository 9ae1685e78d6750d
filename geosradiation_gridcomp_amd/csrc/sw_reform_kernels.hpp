// sw_reform_kernels.hpp -- RRTMG_SW band sweeps, lane = (column, unit of a band's g-points), with (fp32) the second sweep
// RE-FORMING every cell's optics and layer properties instead of reading them back:
//   sweep A (surface -> TOA): k-distribution, delta scaling, reftra_sw, upward adding (vrtqdr_sw :1453-1505); parks per cell the gas
//     optical depth and the two upward reflectances at the cell's upper boundary - 12 bytes (fp32) instead of k_sw_bands' 28;
//   sweep B (TOA -> surface): Rayleigh and aerosol terms re-read per layer (4 values per lane and layer), cloud optics from the McICA
//     planes a second time; delta scaling + reftra_sw again, downward adding (:1530-1572), fluxes (:1576-1586), band sums (:467-502).
// Reference: SW/rrtmg_sw_taumol.F90:27-2084, rrtmg_sw_spcvmc.F90:413-671, :1115-1370 (reftra_sw), :1374-1588 (vrtqdr_sw).
// Why this wins now and lost in round 2 (profiles/r02_sw_parked_cells.md): the build no longer lets the SLP vectorizer pair the
// g-points' arithmetic into half-rate v_pk_* instructions (a third fewer instructions, 30-50 fewer VGPRs), reftra shares its
// direct-beam exponential, and the layer loops are not unrolled - the second two-stream per cell then costs less than the 16 parked
// bytes (32 bytes of HBM traffic) it replaces.
#pragma once
#include "sw_kernels.hpp"

namespace geosrad {

template <typename B> struct SwrRaylPerG { static constexpr bool value = B::JB >= 23 && B::JB <= 27; };

// A band's g-points are dealt to lanes in UNITS of at most SWR_U g-points (lane = (column, unit)): the adding state a lane carries is
// that of its unit, which sets the register count and with it the wavefronts per SIMD (measured per unit size, cloud-free / cloudy
// instantiation: 4: 113 / 168, 6: 112 / 164, 8: 128 / 193, 10: 148 / 225, 12: 161 / 256 VGPRs); the layer records, aerosol terms and
// partial fluxes are touched once per unit.  SWR_U = 12: one unit per band.
#ifndef SWR_U
#define SWR_U 6
#endif
// fp64: every real is a register pair - units of at most 4 g-points, evaluated 2 at a time (SWR_W64), keep the cloudy instantiation at
// 247 VGPRs = two wavefronts per SIMD (units of 6 evaluated 4 at a time: 256 + 72 spilled into AGPRs, one wavefront)
#ifndef SWR_U64
#define SWR_U64 4
#endif
template <typename R> constexpr int swr_u = sizeof(R) == 8 ? SWR_U64 : SWR_U;
// sizes of the units of a band with ng g-points: first unit's size, number of units (the units after the first share the rest evenly in
// multiples of 2: 8 -> 4 + 4, 10 -> 6 + 4, 12 -> 6 + 6 for SWR_U = 6)
__host__ __device__ constexpr int swr_nunit(int U, int ng) { return (ng + U - 1) / U; }
__host__ __device__ constexpr int swr_usize(int U, int ng, int u)
{
    const int nu = swr_nunit(U, ng);
    if (nu == 1) return ng;
    // nu >= 2: sizes in multiples of 2, as even as possible, larger units first
    const int pairs = ng / 2, base = pairs / nu, extra = pairs % nu;
    return 2 * (base + (u < extra ? 1 : 0));
}
__host__ __device__ constexpr int swr_ustart(int U, int ng, int u) { int s = 0; for (int k = 0; k < u; k++) s += swr_usize(U, ng, k); return s; }
// slots of the partial fluxes: units in band order (16..29), ascending inside a band
__host__ __device__ constexpr int swr_slot0(int U, int jb) { int s = 0; for (int b = 16; b < jb; b++) s += swr_nunit(U, sw_band_ng(b)); return s; }
template <typename R> constexpr int swr_nslot = swr_slot0(swr_u<R>, 30);                                   // 23 (fp32), 32 (fp64)
template <typename R> constexpr int swr_ncot = swr_slot0(swr_u<R>, 27) - swr_slot0(swr_u<R>, 24);          // units of the PAR bands 24-26

// one unit (g-points GO .. GO + NGU - 1 of band B) of one column
template <typename R, typename B, bool CLD, int GO, int NGU, int SLOT>
GR_DEV void swr_body(const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV, int col, int nclear)
{
    constexpr int NG = NGU, NGB = B::NG, IBM = B::JB - 15, G0 = B::G0;        // NG: g-points of this lane; NGB: of the band (strides)
    // fp32 re-forms a cell's optics in the second sweep; in fp64 the second two-stream (IEEE divisions, double-precision exp / sqrt) costs
    // more than the traffic it saves, so there the five layer properties are parked as well (planes 5 .. 9; the total sky's in cloudy
    // cells: 10 .. 14) and read back - the unit mapping (small state, no spills, two wavefronts per SIMD instead of one) is what fp64 gains
    constexpr bool REFORM = sizeof(R) == 4;
#ifndef SWR_W64
#define SWR_W64 2
#endif
    constexpr int W = sizeof(R) == 8 ? (NG >= SWR_W64 ? SWR_W64 : 2) : (NG >= 4 ? 4 : 2);
    constexpr int NQ = (NG + W - 1) / W;
    constexpr int S = pad4(NGB);
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const uint32_t ucol = (uint32_t)col;
    const uint32_t cb = ucol * (uint32_t)sizeof(R);
    const int pc = ldg(A.perm, ucol * 4u);
    const uint32_t cba = (uint32_t)pc * (uint32_t)sizeof(R);
    const SwBandTab<R> &Bt = T.b[IBM];
    int ncl_opaque = nclear;
    asm volatile("" : "+s"(ncl_opaque));        // (see sw_band_body: a compile-time "true" here costs registers)
    const bool ccol = CLD && col >= ncl_opaque;
    R prmu0 = ldg(A.coszen, cba);
    prmu0 = prmu0 > (R)1.e-10 ? prmu0 : (R)1.e-10;                     // zepzen (SW/rrtmg_sw_rad.F90:1365)
    const R rmu0 = (R)1. / prmu0;

    // surface albedo of this band (:1230-1248)
    R albp, albd;
    if (IBM <= 8 || IBM == 14) { albp = ldg(A.aldir, cba); albd = ldg(A.aldif, cba); }
    else if (IBM >= 10) { albp = ldg(A.asdir, cba); albd = ldg(A.asdif, cba); }
    else { albp = (ldg(A.asdir, cba) + ldg(A.aldir, cba)) / (R)2.; albd = (ldg(A.asdif, cba) + ldg(A.aldif, cba)) / (R)2.; }

    // ---- solar source of the band's g-points (taumolNN tail sections) -----------------------------------
    R zinc[NG];          // adjflux * ssi (without the cosine)
    {
        int js = 1; R fs = 0;
        if constexpr (B::SRC != 0) {
            int laytrop = 0;
            for (int lay = 0; lay < nlay; lay++) laytrop += (int)((ldg(A.scidx, ((uint32_t)lay * (uint32_t)n + ucol) * 4u) >> 23) & 1u);
            int lsol;
            if constexpr (B::SRC == 1) {
                lsol = laytrop - 1;
                for (int lay = 0; lay < laytrop; lay++) {
                    const int jp0 = (int)(ldg(A.scidx, ((uint32_t)lay * (uint32_t)n + ucol) * 4u) & 63u);
                    const int jp1 = lay + 1 < nlay ? (int)(ldg(A.scidx, ((uint32_t)(lay + 1) * (uint32_t)n + ucol) * 4u) & 63u) : 99;
                    if (jp0 < B::LREF && jp1 >= B::LREF) { lsol = (lay + 1 < laytrop - 1) ? lay + 1 : laytrop - 1; break; }
                }
                if (lsol < 0) lsol = 0;
            } else {
                lsol = nlay - 1;
                for (int lay = laytrop; lay < nlay; lay++) {
                    const int jpm = lay > 0 ? (int)(ldg(A.scidx, ((uint32_t)(lay - 1) * (uint32_t)n + ucol) * 4u) & 63u) : 0;
                    const int jp0 = (int)(ldg(A.scidx, ((uint32_t)lay * (uint32_t)n + ucol) * 4u) & 63u);
                    if (jpm < B::LREF && jp0 >= B::LREF) { lsol = lay; break; }
                }
            }
            SwLayer<R> Ls;
            sw_load_layer<R>(A, lsol, col, Ls);
            const SwSpec<R> sp = (B::SRC == 1) ? sw_spec<R>(Ls.col[B::LOA], (R)B::STR, Ls.col[B::LOB], 8, T.oneminus)
                                               : sw_spec<R>(Ls.col[B::UPA], (R)B::STR, Ls.col[B::UPB], 4, T.oneminus);
            js = sp.js; fs = sp.fs;
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R sf[W], fb[W], sd[W], ir[W];
            const int go = GO + q * W;
            if constexpr (B::NSRC == 1) {
                ldw<R, W>(Bt.sflux, (uint32_t)go * (uint32_t)sizeof(R), sf); ldw<R, W>(Bt.facb, (uint32_t)go * (uint32_t)sizeof(R), fb);
                ldw<R, W>(Bt.snsp, (uint32_t)go * (uint32_t)sizeof(R), sd); ldw<R, W>(Bt.irrad, (uint32_t)go * (uint32_t)sizeof(R), ir);
            } else {
                linw<R, W, S>(sf, fs, Bt.sflux, js - 1, go); linw<R, W, S>(fb, fs, Bt.facb, js - 1, go);
                linw<R, W, S>(sd, fs, Bt.snsp, js - 1, go); linw<R, W, S>(ir, fs, Bt.irrad, js - 1, go);
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                R src;
                if (SV.isolvar < 0) src = sf[j];
                else if (SV.isolvar <= 2) src = SV.svar_f * fb[j] + SV.svar_s * sd[j] + SV.svar_i * ir[j];
                else src = SV.svar_bnd[IBM] * fb[j] + SV.svar_bnd[IBM] * sd[j] + SV.svar_bnd[IBM] * ir[j];
                zinc[g] = SV.adjflux[IBM] * src;
            }
        }
    }

    // parked planes of this band, tiled by 256-column block: [block][layer][g][256]
    //   0 tau_gas   1 prup   2 prupd (upward adding reflectances of the clear sky at the cell's upper boundary)
    //   3, 4 the same two of the total sky, from the sub-column's lowest cloudy cell upwards (cloudy columns)
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    const size_t plane = (size_t)NG_SW * nlay * npad;
    R *const cellb = A.cell + (size_t)G0 * nlay * npad;
    const uint32_t tbase = (ucol >> 8) * (uint32_t)nlay * (uint32_t)NGB * 256u + (ucol & 255u);
    const size_t bandoff = (size_t)G0 * nlay * n;
    const R *const tcb = A.taucmc + bandoff, *const ocb = A.ssacmc + bandoff, *const gcb = A.asmcmc + bandoff;
#define CELL(q) (cellb + (size_t)(q) * plane)
#define PST(q, off, v) stg_nt(CELL(q), off, v)
#define PLD(q, off) ldg_nt(CELL(q), off)
    // (g_: index inside the unit)
#define CT4(lay_, g_) ((tbase + ((uint32_t)(lay_) * (uint32_t)NGB + (uint32_t)(GO + (g_))) * 256u) * (uint32_t)sizeof(R))
#define MC4(lay_, g_) ((((uint32_t)(lay_) * (uint32_t)NGB + (uint32_t)(GO + (g_))) * (uint32_t)n + ucol) * (uint32_t)sizeof(R))
    const size_t aerb = (size_t)(IBM - 1) * nlay * ld;

    // ---- sweep A: surface -> TOA --------------------------------------------------------------------------------------
    R prup[NG], prupd[NG], prupT[NG], prupdT[NG];
    int lowc[NG];            // lowest cloudy layer of sub-column g: the parked total-sky planes hold values from there upwards
#pragma unroll
    for (int g = 0; g < NG; g++) { prup[g] = albp; prupd[g] = albd; prupT[g] = albp; prupdT[g] = albd; lowc[g] = 0x7fffffff; }
    int lcA = CLD ? (int)ldg(A.laycloudy, ucol) : 0;       // the layer's cloud flag, read one layer ahead (it gates the layer's McICA requests)
#pragma nounroll
    for (int lay = 0; lay < nlay; lay++) {
        const int lcA_cur = lcA;
        if (CLD && lay + 1 < nlay) lcA = (int)ldg(A.laycloudy, (uint32_t)(lay + 1) * (uint32_t)n + ucol);
        SwLayer<R> L;
        sw_load_layer<R>(A, lay, col, L);
        R ta = 0, om = 1, as = 0;
        if (A.iaer == 10) {
            const uint32_t ab = ((uint32_t)lay * (uint32_t)ld + (uint32_t)pc) * (uint32_t)sizeof(R);
            ta = ldg(A.tauaer + aerb, ab); om = ldg(A.ssaaer + aerb, ab); as = ldg(A.asmaer + aerb, ab);
        }
        const bool laycld = CLD && ccol && lcA_cur != 0;
        const bool wlc = CLD && __ballot(laycld) != 0;      // some column of the wave has cloud in this layer (wave-uniform)
        SwPrep<R> P;
        sw_prep<R, B>(T, L, P);
        R tr0 = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            R tcv[W], ocv[W], gcv[W];
#pragma unroll
            for (int j = 0; j < W; j++) { tcv[j] = 0; ocv[j] = 0; gcv[j] = 0; }
            if (wlc) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (q * W + j < NG) {
                        // (single-scattering albedo and asymmetry with the optical depth: behind a test of the optical depths they were a
                        // second memory round trip per group)
                        tcv[j] = ldg(tcb, MC4(lay, q * W + j));
                        ocv[j] = ldg(ocb, MC4(lay, q * W + j)); gcv[j] = ldg(gcb, MC4(lay, q * W + j));
                    }
            }
            R tg[W], tr[W];
            sw_eval<R, B, W>(T, L, P, GO + q * W, tg, tr);
            if constexpr (!SwrRaylPerG<B>::value) {      // one Rayleigh optical depth per layer: what depends on it alone is formed once
                if (q == 0) tr0 = tr[0];
#pragma unroll
                for (int j = 0; j < W; j++) tr[j] = tr0;
            }
            if (wlc) {
#pragma unroll
                for (int j = 0; j < W; j++) tcv[j] = laycld ? tcv[j] : (R)0;
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const uint32_t ct4 = CT4(lay, g);
                SwCell<R> c;
                sw_cell_clear<R>(tg[j], tr[j], ta, om, as, prmu0, rmu0, c);
                if constexpr (REFORM) PST(0, ct4, tg[j]);
                else { PST(5, ct4, c.ref); PST(6, ct4, c.refd); PST(7, ct4, c.tra); PST(8, ct4, c.trad); PST(9, ct4, c.dbt); }
                {   // upward adding (:1453-1505): reflectances of everything below the cell's upper boundary
                    const R zrj = f_rcp<R>((R)1. - prupd[g] * c.refd);
                    const R pu = c.ref + (c.trad * ((c.tra - c.dbt) * prupd[g] + c.dbt * prup[g])) * zrj;
                    const R pd = c.refd + c.trad * c.trad * prupd[g] * zrj;
                    prup[g] = pu; prupd[g] = pd;
                }
                PST(1, ct4, prup[g]); PST(2, ct4, prupd[g]);
                if constexpr (CLD) {
                    const bool cellcld = tcv[j] > 0;
                    // Below the lowest cloudy cell of a sub-column the total-sky upward state IS the clear-sky one
                    if (cellcld && lowc[g] > lay) lowc[g] = lay;
                    const bool divg = ccol && lowc[g] <= lay;
                    if (ccol && !divg) { prupT[g] = prup[g]; prupdT[g] = prupd[g]; }
                    if (divg) {
                        SwCell<R> t = c;
                        if (cellcld) {
                            sw_cell_cloud<R>(c, tcv[j], ocv[j], gcv[j], prmu0, rmu0, t);
                            if constexpr (!REFORM) { PST(10, ct4, t.ref); PST(11, ct4, t.refd); PST(12, ct4, t.tra); PST(13, ct4, t.trad); PST(14, ct4, t.dbt); }
                        }
                        const R zrj = f_rcp<R>((R)1. - prupdT[g] * t.refd);
                        const R pu = t.ref + (t.trad * ((t.tra - t.dbt) * prupdT[g] + t.dbt * prupT[g])) * zrj;
                        const R pd = t.refd + t.trad * t.trad * prupdT[g] * zrj;
                        prupT[g] = pu; prupdT[g] = pd;
                        PST(3, ct4, pu); PST(4, ct4, pd);
                    }
                }
            }
#ifndef SWR_NO_GROUP_BARRIER
            __builtin_amdgcn_sched_barrier(0);      // one group's arithmetic at a time
#endif
        }
    }

    // ---- sweep B: TOA -> surface ---------------------------------------------------------------------------------------
    const size_t qs = (size_t)swr_nslot<R> * (nlay + 1) * n;
    R *const part = A.part + (size_t)SLOT * (nlay + 1) * n;
#ifdef SWR_PART_NT
#define PART(kind, lev, val) stg_nt(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))
#else
#define PART(kind, lev, val) stg(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))
#endif
    R tdbt[NG], ztdn[NG], prdnd[NG], tdbtT[NG], ztdnT[NG], prdndT[NG];
    uint32_t dmask = 0;      // bit g: a cloudy cell has been met in sub-column g (total-sky downward state diverged from clear sky)
    {   // level nlay (TOA): ptdbt = ztdn = 1, prdnd = 0; the upward reflectances there are still in registers
        R cu = 0, cd = 0, fu = 0, fd = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            tdbt[g] = 1; ztdn[g] = 1; prdnd[g] = 0; tdbtT[g] = 1; ztdnT[g] = 1; prdndT[g] = 0;
            const R zi = zinc[g] * prmu0;
            {
                const R pu = prup[g], pd = prupd[g];
                const R zr = f_rcp<R>((R)1. - prdnd[g] * pd);
                cu = cu + zi * ((tdbt[g] * pu + (ztdn[g] - tdbt[g]) * pd) * zr);
                cd = cd + zi * (tdbt[g] + (ztdn[g] - tdbt[g] + tdbt[g] * pu * prdnd[g]) * zr);
            }
            if (CLD && ccol) {
                const R puT = prupT[g], pdT = prupdT[g];
                const R zr = f_rcp<R>((R)1. - prdndT[g] * pdT);
                fu = fu + zi * ((tdbtT[g] * puT + (ztdnT[g] - tdbtT[g]) * pdT) * zr);
                fd = fd + zi * (tdbtT[g] + (ztdnT[g] - tdbtT[g] + tdbtT[g] * puT * prdndT[g]) * zr);
            }
        }
        PART(0, nlay, cu); PART(1, nlay, cd);
        if (CLD && ccol) { PART(2, nlay, fu); PART(3, nlay, fd); }
    }
    // a group's parked values (gas optical depth; the clear sky's upward reflectances at the lower boundary) are requested one group
    // ahead of their use, a layer's Rayleigh / aerosol terms one layer ahead
    struct Park { R tg[W], pu[W], pd[W], lp[REFORM ? 1 : 5][W]; };
    auto request = [&](int lay, int q, Park &b) {
        const int lu = lay > 0 ? lay - 1 : 0;      // (surface layer: a harmless repeat; replaced by the albedo on use)
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int g = q * W + j;
            b.tg[j] = 0; b.pu[j] = 0; b.pd[j] = 0;
            if constexpr (!REFORM) { for (int k = 0; k < 5; k++) b.lp[k][j] = 0; }
            if (g >= NG) continue;
            if constexpr (REFORM) b.tg[j] = PLD(0, CT4(lay, g));
            else { for (int k = 0; k < 5; k++) b.lp[k][j] = PLD(5 + k, CT4(lay, g)); }
            b.pu[j] = PLD(1, CT4(lu, g)); b.pd[j] = PLD(2, CT4(lu, g));
        }
    };
    struct Lay { R colmol, ta, om, as; uint32_t idx; R ca, cb2; };
    auto request_layer = [&](int lay, Lay &b) {
        const uint32_t wb = ((uint32_t)lay * (uint32_t)n + ucol) * (uint32_t)sizeof(R);
        const size_t fs_ = (size_t)nlay * n;
        b.colmol = ldg(A.sc + (size_t)SW_COLMOL * fs_, wb);
        b.ta = 0; b.om = 1; b.as = 0; b.idx = 0; b.ca = 0; b.cb2 = 0;
        if (A.iaer == 10) {
            const uint32_t ab = ((uint32_t)lay * (uint32_t)ld + (uint32_t)pc) * (uint32_t)sizeof(R);
            b.ta = ldg(A.tauaer + aerb, ab); b.om = ldg(A.ssaaer + aerb, ab); b.as = ldg(A.asmaer + aerb, ab);
        }
        if constexpr (B::JB == 24) {       // rayla is interpolated in the binary-species parameter below the tropopause (:1467)
            b.idx = ldg(A.scidx, ((uint32_t)lay * (uint32_t)n + ucol) * 4u);
            b.ca = ldg(A.sc + (size_t)(SW_COLH2O + B::LOA) * fs_, wb); b.cb2 = ldg(A.sc + (size_t)(SW_COLH2O + B::LOB) * fs_, wb);
        }
    };
    Park nx; Lay nl;
    request_layer(nlay - 1, nl);
    request(nlay - 1, 0, nx);
    R sdir = 0, sfd = 0, sfu = 0;
    int lcB = CLD ? (int)ldg(A.laycloudy, (uint32_t)(nlay - 1) * (uint32_t)n + ucol) : 0;
#pragma nounroll
    for (int lay = nlay - 1; lay >= 0; lay--) {      // cross layer `lay`; its lower boundary is API level `lay`
        const int lcB_cur = lcB;
        if (CLD && lay > 0) lcB = (int)ldg(A.laycloudy, (uint32_t)(lay - 1) * (uint32_t)n + ucol);
        const int jk = nlay - 1 - lay;
        const Lay cl = nl;
        request_layer(lay > 0 ? lay - 1 : 0, nl);
        const bool laycld = CLD && ccol && lcB_cur != 0;
        const bool wlc = CLD && __ballot(laycld) != 0;
        bool lower = true; int js = 1; R fs = 0;
        if constexpr (B::JB == 24) {
            lower = (cl.idx >> 23) & 1u;
            const SwSpec<R> sp = sw_spec<R>(cl.ca, (R)B::STR, cl.cb2, 8, T.oneminus);
            js = sp.js; fs = sp.fs;
        }
        R tr0 = 0;
        R cu = 0, cd = 0, fu = 0, fd = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            Park cur = nx;
            if (q + 1 < NQ) request(lay, q + 1, nx);
            else request(lay > 0 ? lay - 1 : 0, 0, nx);      // (after the surface layer: a harmless repeat)
            // cloud optics of the group's cells / diverged upward reflectances at the lower boundary, behind wave-uniform tests
            R tcv[W], ocv[W], gcv[W], puT[W], pdT[W];
            bool dv[W], anydv = false;
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                tcv[j] = 0; ocv[j] = 0; gcv[j] = 0;
                cur.pu[j] = lay > 0 ? cur.pu[j] : albp; cur.pd[j] = lay > 0 ? cur.pd[j] : albd;
                puT[j] = cur.pu[j]; pdT[j] = cur.pd[j];
                dv[j] = false;
                if constexpr (CLD) { if (g < NG) dv[j] = ccol && lay > 0 && lowc[g] <= lay - 1; }
                anydv = anydv || dv[j];
            }
            if constexpr (CLD) {
                if (wlc) {
#pragma unroll
                    for (int j = 0; j < W; j++)
                        if (q * W + j < NG) {
                            tcv[j] = ldg(tcb, MC4(lay, q * W + j));
                            if constexpr (REFORM) { ocv[j] = ldg(ocb, MC4(lay, q * W + j)); gcv[j] = ldg(gcb, MC4(lay, q * W + j)); }
                        }
                }
                if (__ballot(anydv) != 0) {
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        if (q * W + j >= NG) continue;
                        const uint32_t u4 = CT4(lay > 0 ? lay - 1 : 0, q * W + j);
                        const R b0 = PLD(3, u4), b1 = PLD(4, u4);
                        if (dv[j]) { puT[j] = b0; pdT[j] = b1; }
                    }
                }
            }
            // Rayleigh optical depth of the group (taur = colmol * rayl)
            R tr[W];
            if constexpr (REFORM) {
                if (SwrRaylPerG<B>::value || q == 0) sw_rayl<R, B, W>(T, lower, cl.colmol, js, fs, GO + q * W, tr);
                if constexpr (!SwrRaylPerG<B>::value) {
                    if (q == 0) tr0 = tr[0];
#pragma unroll
                    for (int j = 0; j < W; j++) tr[j] = tr0;
                }
            }
            // fp64: the total sky's parked layer properties of the group's cloudy cells, behind one wave-uniform test
            R tp[REFORM ? 1 : 5][W];
            if constexpr (!REFORM && CLD) {
                bool anycm = false;
#pragma unroll
                for (int j = 0; j < W; j++) anycm = anycm || (ccol && laycld && tcv[j] > 0);
                if (__ballot(anycm) != 0) {
#pragma unroll
                    for (int j = 0; j < W; j++)
                        if (q * W + j < NG) { for (int k = 0; k < 5; k++) tp[k][j] = PLD(10 + k, CT4(lay, q * W + j)); }
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const R zi = zinc[g] * prmu0;
                SwCell<R> c;
                if constexpr (REFORM) sw_cell_clear<R>(cur.tg[j], tr[j], cl.ta, cl.om, cl.as, prmu0, rmu0, c);
                else { c.tau = 0; c.om = 0; c.g = 0; c.ref = cur.lp[0][j]; c.refd = cur.lp[1][j]; c.tra = cur.lp[2][j]; c.trad = cur.lp[3][j]; c.dbt = cur.lp[4][j]; }
                // downward adding recurrences (:1530-1572): values at the lower boundary of this layer
                {
                    R zt, pr;
                    if (jk == 0) { zt = c.tra; pr = c.refd; }
                    else {
                        const R zreflect = f_rcp<R>((R)1. - c.refd * prdnd[g]);
                        zt = tdbt[g] * c.tra + (c.trad * ((ztdn[g] - tdbt[g]) + tdbt[g] * c.ref * prdnd[g])) * zreflect;
                        pr = c.refd + c.trad * c.trad * prdnd[g] * zreflect;
                    }
                    tdbt[g] = c.dbt * tdbt[g]; ztdn[g] = zt; prdnd[g] = pr;
                }
                R u, d;
                {
                    const R zr = f_rcp<R>((R)1. - prdnd[g] * cur.pd[j]);
                    u = (tdbt[g] * cur.pu[j] + (ztdn[g] - tdbt[g]) * cur.pd[j]) * zr;
                    d = tdbt[g] + (ztdn[g] - tdbt[g] + tdbt[g] * cur.pu[j] * prdnd[g]) * zr;
                    cu = cu + zi * u; cd = cd + zi * d;
                }
                if constexpr (CLD) {
                    const bool cm = ccol && laycld && tcv[j] > 0;
                    // Above the highest cloudy cell of a sub-column the total-sky downward state IS the clear-sky one
                    const bool divg = ccol && (((dmask >> g) & 1u) || cm);
                    if (ccol && !divg) { tdbtT[g] = tdbt[g]; ztdnT[g] = ztdn[g]; prdndT[g] = prdnd[g]; }
                    if (divg) {
                        dmask |= 1u << g;
                        SwCell<R> t = c;
                        if constexpr (REFORM) { if (cm) sw_cell_cloud<R>(c, tcv[j], ocv[j], gcv[j], prmu0, rmu0, t); }
                        else { if (cm) { t.ref = tp[0][j]; t.refd = tp[1][j]; t.tra = tp[2][j]; t.trad = tp[3][j]; t.dbt = tp[4][j]; } }
                        R zt, pr;
                        if (jk == 0) { zt = t.tra; pr = t.refd; }
                        else {
                            const R zreflect = f_rcp<R>((R)1. - t.refd * prdndT[g]);
                            zt = tdbtT[g] * t.tra + (t.trad * ((ztdnT[g] - tdbtT[g]) + tdbtT[g] * t.ref * prdndT[g])) * zreflect;
                            pr = t.refd + t.trad * t.trad * prdndT[g] * zreflect;
                        }
                        tdbtT[g] = t.dbt * tdbtT[g]; ztdnT[g] = zt; prdndT[g] = pr;
                    }
                    if (ccol) {
                        const R zr = f_rcp<R>((R)1. - prdndT[g] * pdT[j]);
                        u = (tdbtT[g] * puT[j] + (ztdnT[g] - tdbtT[g]) * pdT[j]) * zr;
                        d = tdbtT[g] + (ztdnT[g] - tdbtT[g] + tdbtT[g] * puT[j] * prdndT[g]) * zr;
                        fu = fu + zi * u; fd = fd + zi * d;
                    }
                }
                // surface: direct, total downward and upward flux of the sky that counts as total (:624-671)
                if (lay == 0) { sdir = sdir + zi * ((CLD && ccol) ? tdbtT[g] : tdbt[g]); sfd = sfd + zi * d; sfu = sfu + zi * u; }
            }
#ifndef SWR_NO_GROUP_BARRIER
            __builtin_amdgcn_sched_barrier(0);      // one group's arithmetic at a time
#endif
        }
        PART(0, lay, cu); PART(1, lay, cd);
        if (CLD && ccol) { PART(2, lay, fu); PART(3, lay, fd); }
    }
    stg(A.bsfc + (size_t)(0 * swr_nslot<R> + SLOT) * n, cb, sdir);
    stg(A.bsfc + (size_t)(1 * swr_nslot<R> + SLOT) * n, cb, sfd);
    stg(A.bsfc + (size_t)(2 * swr_nslot<R> + SLOT) * n, cb, sfu);
#undef PART
#undef PST
#undef PLD
#undef CT4
#undef CELL
#undef MC4

    // ---- PAR in-cloud optical thickness diagnostics (SW/rrtmg_sw_spcvmc.F90:749-1109), bands 24-26 ---------------
    if constexpr (IBM >= 9 && IBM <= 11) {
        R d[4] = {0, 0, 0, 0}, nn[4] = {0, 0, 0, 0};
        if (CLD && ccol) {
            const R w0 = IBM == 9 ? (R)0.5 : (R)1.0;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const R wgt = w0 * zinc[g];
                const R sl = ldg(A.cotsum + (size_t)(0 * NG_SW + G0 + GO + g) * n, cb), sm = ldg(A.cotsum + (size_t)(1 * NG_SW + G0 + GO + g) * n, cb),
                        sh = ldg(A.cotsum + (size_t)(2 * NG_SW + G0 + GO + g) * n, cb);
                if (sl > 0) { d[3] += wgt; nn[3] += wgt * sl; }
                if (sm > 0) { d[2] += wgt; nn[2] += wgt * sm; }
                if (sh > 0) { d[1] += wgt; nn[1] += wgt * sh; }
                const R st = sl + sm + sh;
                if (st > 0) { d[0] += wgt; nn[0] += wgt * st; }
            }
        }
        constexpr int cs = SLOT - swr_slot0(swr_u<R>, 24);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            stg(A.cot + (size_t)(k * swr_ncot<R> + cs) * n, cb, d[k]);
            stg(A.cot + (size_t)((4 + k) * swr_ncot<R> + cs) * n, cb, nn[k]);
        }
    }
}

#ifndef SWR_OCC_CLR
#define SWR_OCC_CLR 4
#endif
#ifndef SWR_OCC_CLD
#define SWR_OCC_CLD 3
#endif
#ifndef SWR_OCC_CLD64          // fp64 cloudy instantiation: two wavefronts per SIMD with units of 4 evaluated 2 g-points at a time
#define SWR_OCC_CLD64 2
#endif
// block slot -> (band, unit): the units of the 14 bands in SW_BAND_ORDER (heaviest bands first)
struct SwrUnit { int jb, u; };
__host__ __device__ constexpr SwrUnit swr_unit_of(int U, int s)
{
    constexpr int order[NB_SW] = {17, 29, 20, 21, 23, 18, 19, 24, 27, 16, 25, 26, 28, 22};
    for (int k = 0; k < NB_SW; k++) {
        const int nu = swr_nunit(U, sw_band_ng(order[k]));
        if (s < nu) return SwrUnit{order[k], s};
        s -= nu;
    }
    return SwrUnit{22, 0};
}
template <typename R, typename B, bool CLD, int U>
GR_DEV void swr_unit(const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV, int col, int nclear)
{
    constexpr int UU = swr_u<R>;
    if constexpr (U < swr_nunit(UU, B::NG))
        swr_body<R, B, CLD, swr_ustart(UU, B::NG, U), swr_usize(UU, B::NG, U), swr_slot0(UU, B::JB) + U>(A, T, SV, col, nclear);
}

template <typename R, bool CLD>
__global__ void __launch_bounds__(256, (sizeof(R) == 4 ? (CLD ? SWR_OCC_CLD : SWR_OCC_CLR) : (CLD ? SWR_OCC_CLD64 : 2))) k_sw_reform(SwArgs<R> A, SwDev<R> T, SwSolar<R> SV)
{
    int bstart, bslot;       // one-dimensional grid: the units of a column block run together on one XCD (lw_kernels.hpp band_block)
    if (!band_block(A.ncol, swr_nslot<R>, bstart, bslot)) return;
    const int nclear = *A.nclear;
    // every column runs the instantiation of its own class (the one mixed block is visited by both kernels, each
    // masking the other class's lanes): a column's arithmetic never depends on its neighbours -> bitwise column independence
    const int bend = bstart + (int)blockDim.x < A.ncol ? bstart + (int)blockDim.x : A.ncol;
    if (CLD ? bend <= nclear : bstart >= nclear) return;
    const int col = bstart + threadIdx.x;
    if (col >= A.ncol) return;
    if (CLD ? col < nclear : col >= nclear) return;
    const SwrUnit un = swr_unit_of(swr_u<R>, bslot);
#ifdef SWR_ONLY_BAND       // register census of one band's unit bodies (profiles/tools/swr_regs.sh)
#define SWR_ON(B_) (B_::JB == SWR_ONLY_BAND)
#else
#define SWR_ON(B_) true
#endif
#define SWR_CASE(B_) case B_::JB: if (SWR_ON(B_)) { if (un.u == 0) swr_unit<R, B_, CLD, 0>(A, T, SV, col, nclear); else if (un.u == 1) swr_unit<R, B_, CLD, 1>(A, T, SV, col, nclear); \
                                                    else swr_unit<R, B_, CLD, 2>(A, T, SV, col, nclear); } break;
    switch (un.jb) {
        SWR_CASE(SwB16) SWR_CASE(SwB17) SWR_CASE(SwB18) SWR_CASE(SwB19) SWR_CASE(SwB20) SWR_CASE(SwB21) SWR_CASE(SwB22)
        SWR_CASE(SwB23) SWR_CASE(SwB24) SWR_CASE(SwB25) SWR_CASE(SwB26) SWR_CASE(SwB27) SWR_CASE(SwB28) SWR_CASE(SwB29)
        default: break;
    }
#undef SWR_CASE
#undef SWR_ON
}

// ---------------------------------------------------------------------------------------------------
// k_swr_reduce: one thread per (column, level) for the flux profiles, blockIdx.y = nlay + 1 for the per-column part
// (SW/rrtmg_sw_rad.F90:1515-1798): unit partials summed in slot order (band order, units ascending: fixed, so results are bitwise
// reproducible), surface broadband / band diagnostics (spcvmc :624-671), clear == total for cloud-free columns, normFlx.
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_swr_reduce(SwArgs<R> A, SwOut<R> O)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool ccol = col >= *A.nclear;
    const int pc = A.perm[col];
    const size_t qs = (size_t)swr_nslot<R> * (nlay + 1) * n;
    R top = 0;
    for (int s = 0; s < swr_nslot<R>; s++) top += A.part[(size_t)(ccol ? 3 : 1) * qs + ((size_t)s * (nlay + 1) + nlay) * n + col];
    R scale = 1;
    if (A.normFlx == 1) scale = top > (R)1e-7 ? top : (R)1e-7;
    if ((int)blockIdx.y <= nlay) {
        const int lev = blockIdx.y;
        R s4[4] = {0, 0, 0, 0};
        for (int s = 0; s < swr_nslot<R>; s++) {
            const size_t o = ((size_t)s * (nlay + 1) + lev) * n + col;
            s4[0] += A.part[o]; s4[1] += A.part[qs + o];
            if (ccol) { s4[2] += A.part[2 * qs + o]; s4[3] += A.part[3 * qs + o]; }
        }
        if (!ccol) { s4[2] = s4[0]; s4[3] = s4[1]; }
        const size_t i = (size_t)lev * ld + pc;
        if (A.normFlx == 1) { O.swuflxc[i] = s4[0] / scale; O.swdflxc[i] = s4[1] / scale; O.swuflx[i] = s4[2] / scale; O.swdflx[i] = s4[3] / scale; }
        else { O.swuflxc[i] = s4[0]; O.swdflxc[i] = s4[1]; O.swuflx[i] = s4[2]; O.swdflx[i] = s4[3]; }
        return;
    }
    R znirr = 0, znirf = 0, zparr = 0, zparf = 0, zuvrr = 0, zuvrf = 0;
    int s = 0;
    for (int ibm = 1; ibm <= NB_SW; ibm++) {
        R dir = 0, fd = 0, fu = 0;
        const int ns = swr_nunit(swr_u<R>, sw_band_ng(ibm + 15));
        for (int k = 0; k < ns; k++, s++) {
            dir += A.bsfc[(size_t)(0 * swr_nslot<R> + s) * n + col]; fd += A.bsfc[(size_t)(1 * swr_nslot<R> + s) * n + col];
            fu += A.bsfc[(size_t)(2 * swr_nslot<R> + s) * n + col];
        }
        if (ibm == 14 || ibm <= 8) { znirr += dir; znirf += fd; }
        else if (ibm >= 10 && ibm <= 11) { zparr += dir; zparf += fd; }
        else if (ibm >= 12 && ibm <= 13) { zuvrr += dir; zuvrf += fd; }
        else { zparr += (R)0.5 * dir; zparf += (R)0.5 * fd; znirr += (R)0.5 * dir; znirf += (R)0.5 * fd; }
        R fnet = fd - fu, dr = dir, df = fd - dir;
        if (A.normFlx == 1) { fnet = fnet / scale; dr = dr / scale; df = df / scale; }
        O.fswband[(size_t)(ibm - 1) * ld + pc] = fnet;
        if (A.do_drfband) { O.drband[(size_t)(ibm - 1) * ld + pc] = dr; O.dfband[(size_t)(ibm - 1) * ld + pc] = df; }
    }
    R o6[6] = {znirr, znirf - znirr, zparr, zparf - zparr, zuvrr, zuvrf - zuvrr};
    if (A.normFlx == 1) for (int k = 0; k < 6; k++) o6[k] = o6[k] / scale;
    O.nirr[pc] = o6[0]; O.nirf[pc] = o6[1]; O.parr[pc] = o6[2]; O.parf[pc] = o6[3]; O.uvrr[pc] = o6[4]; O.uvrf[pc] = o6[5];
    for (int k = 0; k < 8; k++) {
        R sum = 0;
        if (ccol) for (int b = 0; b < swr_ncot<R>; b++) sum += A.cot[(size_t)(k * swr_ncot<R> + b) * n + col];
        O.cot[k][pc] = sum;
    }
}

}  // namespace geosrad
