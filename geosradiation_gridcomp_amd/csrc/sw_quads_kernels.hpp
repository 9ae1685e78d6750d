// sw_quads_kernels.hpp -- RRTMG_SW band sweeps, second mapping: one lane per (column, QUAD of g-points).
//
// Same fused path as k_sw_bands (sw_kernels.hpp: taumol_sw -> delta scaling -> reftra_sw -> vrtqdr_sw; reference
// SW/rrtmg_sw_taumol.F90:27-2084, rrtmg_sw_spcvmc.F90:413-671, :1115-1370 reftra_sw, :1374-1588 vrtqdr_sw), re-cut for the two things
// that bound k_sw_bands:
//   * HBM traffic of the parked cells.  k_sw_bands parks 7 values per (layer, g-point) cell between its sweeps (28 B, fp32) and runs
//     at the HBM rate of exactly those bytes.  Here sweep B RE-FORMS a cell's optics and layer properties from the one quantity that
//     is expensive to get again - the gas optical depth of the k-distribution - so a cell parks 3 values: tau_gas, and the two upward
//     adding reflectances at its upper boundary (12 B).  Rayleigh and aerosol terms are re-read per layer (one value / three values
//     per lane and layer), cloud optics come from the McICA planes a second time.
//   * registers.  A lane of k_sw_bands carries the adding state of all 6..12 g-points of its band for two skies (up to 256 VGPRs: two
//     wavefronts per SIMD, so a wavefront's arithmetic does not overlap its own memory waits - which is why re-forming lost there,
//     profiles/r02_sw_parked_cells.md).  Here a band's g-points are dealt to wavefronts in quads: a wavefront = 64 columns x one quad,
//     its state that of <= 4 g-points; the per-layer set-up (setcoef record, binary-species parameter, aerosol) is evaluated once per
//     quad instead of once per band.  The wavefronts of a block are the quads of ONE band for the same 64-column groups, so the
//     layer records they all read come from HBM once and from the CU's L1 / the XCD's L2 for the other quads.
// Partial fluxes go to one slot per quad ([kind][32 slots][level][column]); k_swq_reduce sums them in slot order (band order, quads
// ascending inside a band: fixed, so results are bitwise reproducible run to run).
#pragma once
#include "sw_kernels.hpp"

namespace geosrad {

constexpr int SWQ_NSLOT = 32;
struct SwqSlot { int jb, go, wg; };
// quads of the 14 bands in band order (16..29; NG = 6,12,8,8,10,10,2,10,8,6,6,8,6,12): full quads, then the band's tail of 2
__host__ __device__ constexpr SwqSlot swq_slot(int s)
{
    constexpr SwqSlot t[SWQ_NSLOT] = {
        {16, 0, 4}, {16, 4, 2}, {17, 0, 4}, {17, 4, 4}, {17, 8, 4}, {18, 0, 4}, {18, 4, 4}, {19, 0, 4}, {19, 4, 4}, {20, 0, 4}, {20, 4, 4},
        {20, 8, 2}, {21, 0, 4}, {21, 4, 4}, {21, 8, 2}, {22, 0, 2}, {23, 0, 4}, {23, 4, 4}, {23, 8, 2}, {24, 0, 4}, {24, 4, 4}, {25, 0, 4},
        {25, 4, 2}, {26, 0, 4}, {26, 4, 2}, {27, 0, 4}, {27, 4, 4}, {28, 0, 4}, {28, 4, 2}, {29, 0, 4}, {29, 4, 4}, {29, 8, 4}};
    return t[s];
}
// first slot of band jb (16..29) and one past its last
__host__ __device__ constexpr int swq_band_slot0(int jb)
{
    int s = 0;
    while (s < SWQ_NSLOT && swq_slot(s).jb != jb) s++;
    return s;
}
__host__ __device__ constexpr int swq_band_nslot(int jb) { return (sw_band_ng(jb) + 3) / 4; }
// PAR diagnostics (bands 24-26): their quads' slots, numbered 0..5
constexpr int SWQ_NCOT = 6;

// does the band's Rayleigh optical depth depend on the g-point?  (rayl is a band constant in taumol16-22, 28, 29)
template <typename B> struct SwqRaylPerG { static constexpr bool value = B::JB >= 23 && B::JB <= 27; };

// ---------------------------------------------------------------------------------------------------
// one quad of one band of one column.  go = first g-point of the quad inside the band, WG = its width (4, or 2 for a band's tail),
// slot = its partial-flux slot.  Vertical index as in sw_band_body: API layer `lay` = 0 at the surface.
// Parked planes (A.cell, plane stride = NG_SW * nlay * npad), tiled [band][64-column group][quad][layer][j < WG][64] (a wavefront's
// cells are one contiguous run per plane):
//   0 tau_gas   1 prup   2 prupd (upward adding reflectances of the clear sky at the cell's upper boundary)
//   3, 4 the same two of the total sky, from the sub-column's lowest cloudy cell upwards (cloudy columns)
// ---------------------------------------------------------------------------------------------------
template <typename R, typename B, bool CLD, int WG>
GR_DEV void swq_body(const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV, int col, int nclear, int go, int slot)
{
    constexpr int NG = B::NG, IBM = B::JB - 15, G0 = B::G0;
    constexpr int S = pad4(NG);
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const uint32_t ucol = (uint32_t)col;
    const uint32_t cb = ucol * (uint32_t)sizeof(R);
    const int pc = ldg(A.perm, ucol * 4u);
    const uint32_t cba = (uint32_t)pc * (uint32_t)sizeof(R);
    const SwBandTab<R> &Bt = T.b[IBM];
    int ncl_opaque = nclear;
    asm volatile("" : "+s"(ncl_opaque));
    const bool ccol = CLD && col >= ncl_opaque;
    R prmu0 = ldg(A.coszen, cba);
    prmu0 = prmu0 > (R)1.e-10 ? prmu0 : (R)1.e-10;                     // zepzen (SW/rrtmg_sw_rad.F90:1365)
    const R rmu0 = (R)1. / prmu0;

    // surface albedo of this band (:1230-1248)
    R albp, albd;
    if (IBM <= 8 || IBM == 14) { albp = ldg(A.aldir, cba); albd = ldg(A.aldif, cba); }
    else if (IBM >= 10) { albp = ldg(A.asdir, cba); albd = ldg(A.asdif, cba); }
    else { albp = (ldg(A.asdir, cba) + ldg(A.aldir, cba)) / (R)2.; albd = (ldg(A.asdif, cba) + ldg(A.aldif, cba)) / (R)2.; }

    // ---- solar source of the quad's g-points (taumolNN tail sections), as in sw_band_body -------------------
    R zinc[WG], zi[WG];            // adjflux * ssi; the same times the cosine
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
        R sf[WG], fb[WG], sd[WG], ir[WG];
        if constexpr (B::NSRC == 1) {
            ldw<R, WG>(Bt.sflux, (uint32_t)go * (uint32_t)sizeof(R), sf); ldw<R, WG>(Bt.facb, (uint32_t)go * (uint32_t)sizeof(R), fb);
            ldw<R, WG>(Bt.snsp, (uint32_t)go * (uint32_t)sizeof(R), sd); ldw<R, WG>(Bt.irrad, (uint32_t)go * (uint32_t)sizeof(R), ir);
        } else {
            linw<R, WG, S>(sf, fs, Bt.sflux, js - 1, go); linw<R, WG, S>(fb, fs, Bt.facb, js - 1, go);
            linw<R, WG, S>(sd, fs, Bt.snsp, js - 1, go); linw<R, WG, S>(ir, fs, Bt.irrad, js - 1, go);
        }
#pragma unroll
        for (int j = 0; j < WG; j++) {
            R src;
            if (SV.isolvar < 0) src = sf[j];
            else if (SV.isolvar <= 2) src = SV.svar_f * fb[j] + SV.svar_s * sd[j] + SV.svar_i * ir[j];
            else src = SV.svar_bnd[IBM] * fb[j] + SV.svar_bnd[IBM] * sd[j] + SV.svar_bnd[IBM] * ir[j];
            zinc[j] = SV.adjflux[IBM] * src;
            zi[j] = zinc[j] * prmu0;
        }
    }

    // parked planes of this (band, block, quad)
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    const size_t plane = (size_t)NG_SW * nlay * npad;
    R *const cellb = A.cell + (size_t)G0 * nlay * npad + ((size_t)(ucol >> 6) * (uint32_t)NG + (uint32_t)go) * (size_t)nlay * 64u;
    const uint32_t tbase = ucol & 63u;
    const size_t bandoff = (size_t)G0 * nlay * n;
    const R *const tcb = A.taucmc + bandoff, *const ocb = A.ssacmc + bandoff, *const gcb = A.asmcmc + bandoff;
#define CELL(q) (cellb + (size_t)(q) * plane)
#define PST(q, off, v) stg_nt(CELL(q), off, v)
#define PLD(q, off) ldg_nt(CELL(q), off)
#define CT4(lay_, j_) ((tbase + ((uint32_t)(lay_) * (uint32_t)WG + (uint32_t)(j_)) * 64u) * (uint32_t)sizeof(R))
    // McICA planes: [band][lay][g][col]
#define MC4(lay_, j_) ((((uint32_t)(lay_) * (uint32_t)NG + (uint32_t)(go + (j_))) * (uint32_t)n + ucol) * (uint32_t)sizeof(R))
    const size_t aerb = (size_t)(IBM - 1) * nlay * ld;

    // ---- sweep A: surface -> TOA: k-distribution, cell optics, reftra, upward adding ------------------------------------
    R prup[WG], prupd[WG], prupT[WG], prupdT[WG];
    int lowc[WG];
#pragma unroll
    for (int j = 0; j < WG; j++) { prup[j] = albp; prupd[j] = albd; prupT[j] = albp; prupdT[j] = albd; lowc[j] = 0x7fffffff; }
#pragma nounroll
    for (int lay = 0; lay < nlay; lay++) {
        SwLayer<R> L;
        sw_load_layer<R>(A, lay, col, L);
        R ta = 0, om = 1, as = 0;
        if (A.iaer == 10) {
            const uint32_t ab = ((uint32_t)lay * (uint32_t)ld + (uint32_t)pc) * (uint32_t)sizeof(R);
            ta = ldg(A.tauaer + aerb, ab); om = ldg(A.ssaaer + aerb, ab); as = ldg(A.asmaer + aerb, ab);
        }
        const bool laycld = CLD && ccol && ldg(A.laycloudy, (uint32_t)lay * (uint32_t)n + ucol) != 0;
        const bool wlc = CLD && __ballot(laycld) != 0;
        R tcv[WG], ocv[WG], gcv[WG];
#pragma unroll
        for (int j = 0; j < WG; j++) { tcv[j] = 0; ocv[j] = 0; gcv[j] = 0; }
        if (wlc) {
#pragma unroll
            for (int j = 0; j < WG; j++) tcv[j] = ldg(tcb, MC4(lay, j));
        }
        SwPrep<R> P;
        sw_prep<R, B>(T, L, P);
        R tg[WG], tr[WG];
        sw_eval<R, B, WG>(T, L, P, go, tg, tr);
        if constexpr (!SwqRaylPerG<B>::value) {      // one Rayleigh optical depth per layer: what depends on it alone is formed once for the quad
#pragma unroll
            for (int j = 1; j < WG; j++) tr[j] = tr[0];
        }
        if (wlc) {
            bool anyc = false;
#pragma unroll
            for (int j = 0; j < WG; j++) { tcv[j] = laycld ? tcv[j] : (R)0; anyc = anyc || tcv[j] > 0; }
            if (__ballot(anyc) != 0) {
#pragma unroll
                for (int j = 0; j < WG; j++) { ocv[j] = ldg(ocb, MC4(lay, j)); gcv[j] = ldg(gcb, MC4(lay, j)); }
            }
        }
#pragma unroll
        for (int j = 0; j < WG; j++) {
            const uint32_t ct4 = CT4(lay, j);
            SwCell<R> c;
            sw_cell_clear<R>(tg[j], tr[j], ta, om, as, prmu0, rmu0, c);
            PST(0, ct4, tg[j]);
            {
                const R zrj = f_rcp<R>((R)1. - prupd[j] * c.refd);
                const R pu = c.ref + (c.trad * ((c.tra - c.dbt) * prupd[j] + c.dbt * prup[j])) * zrj;
                const R pd = c.refd + c.trad * c.trad * prupd[j] * zrj;
                prup[j] = pu; prupd[j] = pd;
            }
            PST(1, ct4, prup[j]); PST(2, ct4, prupd[j]);
            if constexpr (CLD) {
                const bool cellcld = tcv[j] > 0;
                if (cellcld && lowc[j] > lay) lowc[j] = lay;
                const bool divg = ccol && lowc[j] <= lay;
                if (ccol && !divg) { prupT[j] = prup[j]; prupdT[j] = prupd[j]; }
                if (divg) {
                    SwCell<R> t = c;
                    if (cellcld) sw_cell_cloud<R>(c, tcv[j], ocv[j], gcv[j], prmu0, rmu0, t);
                    const R zrj = f_rcp<R>((R)1. - prupdT[j] * t.refd);
                    const R pu = t.ref + (t.trad * ((t.tra - t.dbt) * prupdT[j] + t.dbt * prupT[j])) * zrj;
                    const R pd = t.refd + t.trad * t.trad * prupdT[j] * zrj;
                    prupT[j] = pu; prupdT[j] = pd;
                    PST(3, ct4, pu); PST(4, ct4, pd);
                }
            }
#ifdef SWQ_CELL_BARRIER
            __builtin_amdgcn_sched_barrier(0);      // one cell's arithmetic at a time: the quad's four two-streams interleaved cost registers
#endif
        }
    }

    // ---- sweep B: TOA -> surface: cell optics re-formed from the parked gas optical depth, downward adding, fluxes ---------------
    const size_t qs = (size_t)SWQ_NSLOT * (nlay + 1) * n;
    R *const part = A.part + (size_t)slot * (nlay + 1) * n;
#define PART(kind, lev, val) stg(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))
    R tdbt[WG], ztdn[WG], prdnd[WG], tdbtT[WG], ztdnT[WG], prdndT[WG];
    uint32_t dmask = 0;
    {   // level nlay (TOA): ptdbt = ztdn = 1, prdnd = 0: up = prup, down = 1
        R cu = 0, cd = 0, fu = 0, fd = 0;
#pragma unroll
        for (int j = 0; j < WG; j++) {
            tdbt[j] = 1; ztdn[j] = 1; prdnd[j] = 0; tdbtT[j] = 1; ztdnT[j] = 1; prdndT[j] = 0;
            {   // the expressions of the general level with the TOA state, as in sw_band_body (same rounding)
                const R pu = prup[j], pd = prupd[j];
                const R zr = f_rcp<R>((R)1. - prdnd[j] * pd);
                cu = cu + zi[j] * ((tdbt[j] * pu + (ztdn[j] - tdbt[j]) * pd) * zr);
                cd = cd + zi[j] * (tdbt[j] + (ztdn[j] - tdbt[j] + tdbt[j] * pu * prdnd[j]) * zr);
            }
            if (CLD && ccol) {
                const R puT = prupT[j], pdT = prupdT[j];
                const R zr = f_rcp<R>((R)1. - prdndT[j] * pdT);
                fu = fu + zi[j] * ((tdbtT[j] * puT + (ztdnT[j] - tdbtT[j]) * pdT) * zr);
                fd = fd + zi[j] * (tdbtT[j] + (ztdnT[j] - tdbtT[j] + tdbtT[j] * puT * prdndT[j]) * zr);
            }
        }
        PART(0, nlay, cu); PART(1, nlay, cd);
        if (CLD && ccol) { PART(2, nlay, fu); PART(3, nlay, fd); }
    }
    // a layer's inputs are requested one layer ahead of their use
    struct Req { R tg[WG], pu[WG], pd[WG], colmol, ta, om, as; uint32_t idx; R ca, cb2; };
    auto request = [&](int lay, Req &b) {
        const int lu = lay > 0 ? lay - 1 : 0;      // (surface layer: a harmless repeat; replaced by the albedo on use)
#pragma unroll
        for (int j = 0; j < WG; j++) { b.tg[j] = PLD(0, CT4(lay, j)); b.pu[j] = PLD(1, CT4(lu, j)); b.pd[j] = PLD(2, CT4(lu, j)); }
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
    Req nx;
    request(nlay - 1, nx);
    R sdir = 0, sfd = 0, sfu = 0;
#pragma nounroll
    for (int lay = nlay - 1; lay >= 0; lay--) {      // cross layer `lay`; its lower boundary is API level `lay`
        const int jk = nlay - 1 - lay;
        Req cur = nx;
        request(lay > 0 ? lay - 1 : 0, nx);
        // cloud optics of the layer's cells / diverged upward reflectances at its lower boundary, behind wave-uniform tests
        const bool laycld = CLD && ccol && ldg(A.laycloudy, (uint32_t)lay * (uint32_t)n + ucol) != 0;
        const bool wlc = CLD && __ballot(laycld) != 0;
        R tcv[WG], ocv[WG], gcv[WG], puT[WG], pdT[WG];
        bool dv[WG], anydv = false;
#pragma unroll
        for (int j = 0; j < WG; j++) {
            tcv[j] = 0; ocv[j] = 0; gcv[j] = 0;
            cur.pu[j] = lay > 0 ? cur.pu[j] : albp; cur.pd[j] = lay > 0 ? cur.pd[j] : albd;
            puT[j] = cur.pu[j]; pdT[j] = cur.pd[j];
            dv[j] = CLD && ccol && lay > 0 && lowc[j] <= lay - 1;
            anydv = anydv || dv[j];
        }
        if constexpr (CLD) {
            if (wlc) {
#pragma unroll
                for (int j = 0; j < WG; j++) { tcv[j] = ldg(tcb, MC4(lay, j)); ocv[j] = ldg(ocb, MC4(lay, j)); gcv[j] = ldg(gcb, MC4(lay, j)); }
            }
            if (__ballot(anydv) != 0) {
#pragma unroll
                for (int j = 0; j < WG; j++) {
                    const uint32_t u4 = CT4(lay > 0 ? lay - 1 : 0, j);
                    const R b0 = PLD(3, u4), b1 = PLD(4, u4);
                    if (dv[j]) { puT[j] = b0; pdT[j] = b1; }
                }
            }
        }
        // Rayleigh optical depth of the quad (taur = colmol * rayl, :e.g. 383-420)
        R tr[WG];
        {
            bool lower = true; int js = 1; R fs = 0;
            if constexpr (B::JB == 24) {
                lower = (cur.idx >> 23) & 1u;
                const SwSpec<R> sp = sw_spec<R>(cur.ca, (R)B::STR, cur.cb2, 8, T.oneminus);
                js = sp.js; fs = sp.fs;
            }
            sw_rayl<R, B, WG>(T, lower, cur.colmol, js, fs, go, tr);
            if constexpr (!SwqRaylPerG<B>::value) {
#pragma unroll
                for (int j = 1; j < WG; j++) tr[j] = tr[0];
            }
        }
        R cu = 0, cd = 0, fu = 0, fd = 0;
#pragma unroll
        for (int j = 0; j < WG; j++) {
            SwCell<R> c;
            sw_cell_clear<R>(cur.tg[j], tr[j], cur.ta, cur.om, cur.as, prmu0, rmu0, c);
            // downward adding recurrences (:1530-1572): values at the lower boundary of this layer
            {
                R zt, pr;
                if (jk == 0) { zt = c.tra; pr = c.refd; }
                else {
                    const R zreflect = f_rcp<R>((R)1. - c.refd * prdnd[j]);
                    zt = tdbt[j] * c.tra + (c.trad * ((ztdn[j] - tdbt[j]) + tdbt[j] * c.ref * prdnd[j])) * zreflect;
                    pr = c.refd + c.trad * c.trad * prdnd[j] * zreflect;
                }
                tdbt[j] = c.dbt * tdbt[j]; ztdn[j] = zt; prdnd[j] = pr;
            }
            R u, d;
            {
                const R zr = f_rcp<R>((R)1. - prdnd[j] * cur.pd[j]);
                u = (tdbt[j] * cur.pu[j] + (ztdn[j] - tdbt[j]) * cur.pd[j]) * zr;
                d = tdbt[j] + (ztdn[j] - tdbt[j] + tdbt[j] * cur.pu[j] * prdnd[j]) * zr;
                cu = cu + zi[j] * u; cd = cd + zi[j] * d;
            }
            if constexpr (CLD) {
                const bool cm = ccol && laycld && tcv[j] > 0;
                // Above the highest cloudy cell of a sub-column the total-sky downward state IS the clear-sky one
                const bool divg = ccol && (((dmask >> j) & 1u) || cm);
                if (ccol && !divg) { tdbtT[j] = tdbt[j]; ztdnT[j] = ztdn[j]; prdndT[j] = prdnd[j]; }
                if (divg) {
                    dmask |= 1u << j;
                    SwCell<R> t = c;
                    if (cm) sw_cell_cloud<R>(c, tcv[j], ocv[j], gcv[j], prmu0, rmu0, t);
                    R zt, pr;
                    if (jk == 0) { zt = t.tra; pr = t.refd; }
                    else {
                        const R zreflect = f_rcp<R>((R)1. - t.refd * prdndT[j]);
                        zt = tdbtT[j] * t.tra + (t.trad * ((ztdnT[j] - tdbtT[j]) + tdbtT[j] * t.ref * prdndT[j])) * zreflect;
                        pr = t.refd + t.trad * t.trad * prdndT[j] * zreflect;
                    }
                    tdbtT[j] = t.dbt * tdbtT[j]; ztdnT[j] = zt; prdndT[j] = pr;
                }
                if (ccol) {
                    const R zr = f_rcp<R>((R)1. - prdndT[j] * pdT[j]);
                    u = (tdbtT[j] * puT[j] + (ztdnT[j] - tdbtT[j]) * pdT[j]) * zr;
                    d = tdbtT[j] + (ztdnT[j] - tdbtT[j] + tdbtT[j] * puT[j] * prdndT[j]) * zr;
                    fu = fu + zi[j] * u; fd = fd + zi[j] * d;
                }
            }
            // surface: direct, total downward and upward flux of the sky that counts as total (:624-671)
            if (lay == 0) { sdir = sdir + zi[j] * ((CLD && ccol) ? tdbtT[j] : tdbt[j]); sfd = sfd + zi[j] * d; sfu = sfu + zi[j] * u; }
#ifdef SWQ_CELL_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        PART(0, lay, cu); PART(1, lay, cd);
        if (CLD && ccol) { PART(2, lay, fu); PART(3, lay, fd); }
    }
    stg(A.bsfc + (size_t)(0 * SWQ_NSLOT + slot) * n, cb, sdir);
    stg(A.bsfc + (size_t)(1 * SWQ_NSLOT + slot) * n, cb, sfd);
    stg(A.bsfc + (size_t)(2 * SWQ_NSLOT + slot) * n, cb, sfu);
#undef PART
#undef PST
#undef PLD
#undef CT4
#undef CELL

    // ---- PAR in-cloud optical thickness diagnostics (SW/rrtmg_sw_spcvmc.F90:749-1109), bands 24-26: this quad's share ---------------
    if constexpr (IBM >= 9 && IBM <= 11) {
        R d[4] = {0, 0, 0, 0}, nn[4] = {0, 0, 0, 0};
        if (CLD && ccol) {
            const R w0 = IBM == 9 ? (R)0.5 : (R)1.0;
#pragma unroll
            for (int j = 0; j < WG; j++) {
                const R wgt = w0 * zinc[j];
                const int g = G0 + go + j;
                const R sl = ldg(A.cotsum + (size_t)(0 * NG_SW + g) * n, cb), sm = ldg(A.cotsum + (size_t)(1 * NG_SW + g) * n, cb),
                        sh = ldg(A.cotsum + (size_t)(2 * NG_SW + g) * n, cb);
                if (sl > 0) { d[3] += wgt; nn[3] += wgt * sl; }
                if (sm > 0) { d[2] += wgt; nn[2] += wgt * sm; }
                if (sh > 0) { d[1] += wgt; nn[1] += wgt * sh; }
                const R st = sl + sm + sh;
                if (st > 0) { d[0] += wgt; nn[0] += wgt * st; }
            }
        }
        const int cs = slot - swq_band_slot0(24);          // 0..5
#pragma unroll
        for (int k = 0; k < 4; k++) {
            stg(A.cot + (size_t)(k * SWQ_NCOT + cs) * n, cb, d[k]);
            stg(A.cot + (size_t)((4 + k) * SWQ_NCOT + cs) * n, cb, nn[k]);
        }
    }
#undef MC4
}

// Blocks: 256 threads = 4 wavefronts; per 256 columns (4 groups of 64) a band with NQ quads takes
//   NQ = 1 (band 22): 1 block  = 4 column groups x 1 quad
//   NQ = 2          : 2 blocks = 2 column groups x 2 quads each
//   NQ = 3          : 4 blocks = 1 column group x 3 quads each (the fourth wavefront exits)
// i.e. 1 + 8 * 2 + 5 * 4 = 37 block slots per 256 columns, in band order.
constexpr int SWQ_NBLK = 37;
struct SwqBlk { int jb, sub; };
__host__ __device__ constexpr int swq_nq(int jb) { return (sw_band_ng(jb) + 3) / 4; }
__host__ __device__ constexpr SwqBlk swq_blk(int s)
{
    int jb = 16;
    while (true) {
        const int nq = swq_nq(jb), nb = nq == 1 ? 1 : (nq == 2 ? 2 : 4);
        if (s < nb) return SwqBlk{jb, s};
        s -= nb; jb++;
    }
}

#ifndef SWQ_OCC_CLR
#define SWQ_OCC_CLR 4
#endif
#ifndef SWQ_OCC_CLD
#define SWQ_OCC_CLD 3
#endif
template <typename R, typename B, bool CLD>
GR_DEV void swq_band(const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV, int bstart, int sub, int nclear)
{
    constexpr int NQ = (B::NG + 3) / 4, TAIL = B::NG % 4;
    const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    int q, cg;
    if constexpr (NQ == 3) { if (w == 3) return; q = w; cg = sub; }
    else if constexpr (NQ == 2) { q = w & 1; cg = sub * 2 + (w >> 1); }
    else { q = 0; cg = w; }
    const int w0 = bstart + cg * 64;          // the wavefront's 64 compacted positions
    if (w0 >= A.ncol) return;
    // every column runs the instantiation of its own class (the one mixed wavefront is visited by both kernels, each masking the
    // other class's lanes): a column's arithmetic never depends on its neighbours -> bitwise column independence
    const int wend = w0 + 64 < A.ncol ? w0 + 64 : A.ncol;
    if (CLD ? wend <= nclear : w0 >= nclear) return;
    const int col = w0 + lane;
    if (col >= A.ncol) return;
    if (CLD ? col < nclear : col >= nclear) return;
    const int slot = swq_band_slot0(B::JB) + q;
    if (TAIL != 0 && q == NQ - 1) swq_body<R, B, CLD, (TAIL != 0 ? TAIL : 4)>(A, T, SV, col, nclear, q * 4, slot);
    else swq_body<R, B, CLD, (B::NG >= 4 ? 4 : B::NG)>(A, T, SV, col, nclear, q * 4, slot);
}

template <typename R, bool CLD>
__global__ void __launch_bounds__(256, (sizeof(R) == 4 ? (CLD ? SWQ_OCC_CLD : SWQ_OCC_CLR) : 2)) k_sw_quads(SwArgs<R> A, SwDev<R> T, SwSolar<R> SV)
{
    int bstart, bslot;       // one-dimensional grid: the blocks of a 256-column group run together on one XCD (lw_kernels.hpp band_block)
    if (!band_block(A.ncol, SWQ_NBLK, bstart, bslot)) return;
    const int nclear = *A.nclear;
    const int bend = bstart + 256 < A.ncol ? bstart + 256 : A.ncol;
    if (CLD ? bend <= nclear : bstart >= nclear) return;
    const SwqBlk blk = swq_blk(bslot);
#ifdef SWQ_ONLY_BAND       // register census of one band's quad bodies (profiles/tools/swq_regs.sh)
#define SWQ_CASE(B_) case B_::JB: if (B_::JB == SWQ_ONLY_BAND) swq_band<R, B_, CLD>(A, T, SV, bstart, blk.sub, nclear); break;
#else
#define SWQ_CASE(B_) case B_::JB: swq_band<R, B_, CLD>(A, T, SV, bstart, blk.sub, nclear); break;
#endif
    switch (blk.jb) {
        SWQ_CASE(SwB16) SWQ_CASE(SwB17) SWQ_CASE(SwB18) SWQ_CASE(SwB19) SWQ_CASE(SwB20) SWQ_CASE(SwB21) SWQ_CASE(SwB22)
        SWQ_CASE(SwB23) SWQ_CASE(SwB24) SWQ_CASE(SwB25) SWQ_CASE(SwB26) SWQ_CASE(SwB27) SWQ_CASE(SwB28) SWQ_CASE(SwB29)
        default: break;
    }
#undef SWQ_CASE
}

// ---------------------------------------------------------------------------------------------------
// k_swq_reduce: k_sw_reduce for the per-quad partials: slots summed in slot order (= band order, quads ascending)
// (SW/rrtmg_sw_rad.F90:1515-1798; surface broadband / band diagnostics spcvmc :624-671; normFlx)
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_swq_reduce(SwArgs<R> A, SwOut<R> O)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool ccol = col >= *A.nclear;
    const int pc = A.perm[col];
    const size_t qs = (size_t)SWQ_NSLOT * (nlay + 1) * n;
    R top = 0;
    for (int s = 0; s < SWQ_NSLOT; s++) top += A.part[(size_t)(ccol ? 3 : 1) * qs + ((size_t)s * (nlay + 1) + nlay) * n + col];
    R scale = 1;
    if (A.normFlx == 1) scale = top > (R)1e-7 ? top : (R)1e-7;
    if ((int)blockIdx.y <= nlay) {
        const int lev = blockIdx.y;
        R s4[4] = {0, 0, 0, 0};
        for (int s = 0; s < SWQ_NSLOT; s++) {
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
        const int ns = (sw_band_ng(ibm + 15) + 3) / 4;
        for (int k = 0; k < ns; k++, s++) {
            dir += A.bsfc[(size_t)(0 * SWQ_NSLOT + s) * n + col]; fd += A.bsfc[(size_t)(1 * SWQ_NSLOT + s) * n + col];
            fu += A.bsfc[(size_t)(2 * SWQ_NSLOT + s) * n + col];
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
        if (ccol) for (int b = 0; b < SWQ_NCOT; b++) sum += A.cot[(size_t)(k * SWQ_NCOT + b) * n + col];
        O.cot[k][pc] = sum;
    }
}

}  // namespace geosrad
