// sw_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the RRTMG_SW column solver.
//
// Reference behaviour (SW = GEOSsolar_GridComp/RRTMG/rrtmg_sw/gcm_model/src): rrtmg_sw_rad.F90:68-1801 (driver,
// solar variability, albedo->band map, normFlx), rrtmg_sw_setcoef.F90:23-241, rrtmg_sw_taumol.F90:27-2084,
// rrtmg_sw_cldprmc.F90:36-418, rrtmg_sw_spcvmc.F90:34-1112 (+ reftra_sw :1115-1370, vrtqdr_sw :1374-1588).
//
// Same mapping as the LW path (lane = column, one block per (256-column block, band) on the XCD-aware one-dimensional grid of
// lw_kernels.hpp band_block, uniform base + 32-bit byte offset addressing):
//   k_sw_validate : per column  - input asserts, cloudy flags, clearCounts of clear columns
//   k_sw_setcoef  : per (layer,column) - column amounts + p/T interpolation record shared by all 14 bands
//   k_mcica<.,2>  : per (column, band) - McICA sub-columns + cldprmc_sw (delta-scaled tau, ssa, g) [mcica_kernels.hpp]
//   k_sw_bands    : per (column, band) - fused taumol_sw -> delta-scaling -> reftra_sw -> vrtqdr_sw:
//                   sweep A (surface -> TOA) evaluates the layer R/T and the upward adding recurrences and parks 7 values
//                   per cell; sweep B (TOA -> surface) runs the downward recurrences in registers and forms the fluxes
//                   (see sw_band_body).  Cloudy columns carry the clear-sky and the total-sky problem through the same two sweeps.
//   k_sw_reduce   : per column - fixed-order sum over bands, surface band diagnostics, optional normalisation
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {

constexpr int NB_SW = 14;
constexpr int NG_SW = 112;

enum SwGas { G_H2O = 0, G_CO2, G_O3, G_CH4, G_O2, G_NONE };
enum SwScField {
    SW_FAC00 = 0, SW_FAC01, SW_FAC10, SW_FAC11, SW_COLH2O, SW_COLCO2, SW_COLO3, SW_COLCH4, SW_COLO2, SW_COLMOL,
    SW_SELFFAC, SW_SELFFRAC, SW_FORFAC, SW_FORFRAC, SW_NFIELD
};

template <typename R> struct SwBandTab {
    const R *absa, *absb, *selfref, *forref;       // [rows][NGP]
    const R *sflux, *irrad, *facb, *snsp;          // [nsrc][NGP]
    const R *rayl;                                 // [NGP] (scalar bands: value replicated), band 24: rayla [9][NGP]
    const R *x0, *x1;                              // extras: absch4 | abso3a, abso3b | absco2, absh2o | raylb (x... see sw tables)
    const R *raylb;                                // band 24 upper
};

template <typename R> struct SwDev {
    SwBandTab<R> b[NB_SW + 1];                     // index = band - 15 (1..14)
    const R *preflog, *tref;
    const R *extliq1, *ssaliq1, *asyliq1, *extice2, *ssaice2, *asyice2, *extice3, *ssaice3, *asyice3, *fdlice3, *extice4,
        *ssaice4, *asyice4;                        // Fortran (n, 16:29)
    R abari[5], bbari[5], cbari[5], dbari[5], ebari[5], fbari[5];
    int icxa[NB_SW + 1];
    R oneminus, grav, avogad, rrsw_scon, Iint, Fint, Sint, Mg_avg, Mg_0, SB_avg, SB_0;
};

// per-call scalars of the solar-variability block (SW/rrtmg_sw_rad.F90:893-1127), evaluated on the host
template <typename R> struct SwSolar {
    int isolvar;
    R svar_f, svar_s, svar_i;
    R svar_bnd[NB_SW + 1];      // isolvar == 3: one multiplier per band (f = s = i)
    R adjflux[NB_SW + 1];
};

template <typename R> struct SwArgs {
    int ncol, ld, nlay, iceflg, liqflg, doy, cloudLM, cloudMH, iaer, normFlx, do_drfband;
    const R *play, *plev, *tlay, *h2o, *o3, *co2, *ch4, *o2, *cld, *ciwp, *clwp, *rei, *rel, *zm, *alat;
    const R *tauaer, *ssaaer, *asmaer, *coszen, *asdir, *asdif, *aldir, *aldif;
    // workspace
    R *sc; uint32_t *scidx;              // setcoef record [SW_NFIELD][nlay][ncol] + packed indices
    uint8_t *colcloudy;                  // [ncol] 1 + highest layer with cld > 0 (0: none), original column order
    int32_t *perm, *nclear;              // k_partition: compacted position -> column; number of clear columns
    R *alpha, *rcorr;
    R *taucmc, *ssacmc, *asmcmc;         // McICA cloud optics, band-major planes [band][lay][g][col]
    uint8_t *laycloudy;                  // [lay][col] some sub-column of the layer has cloud (only those layers' planes are valid)
    R *cotsum;                           // [3][NG_SW][ncol]  per-sub-column low|mid|high sums of the un-scaled cloud tau
    R *cell;                             // [15][band-major plane]: parked values per cell, see sw_band_body
    R *part;                             // [4][14][nlay+1][ncol]: cu, cd, fu, fd per band
    R *bsfc;                             // [3][14][ncol]: surface direct, total down, up (of the sky that counts as total)
    R *cot;                              // [8][14][ncol]: PAR cloud optical thickness partial sums per band
    uint32_t *err;
    int32_t *clearCounts;
    R *dbg_taug, *dbg_taur, *dbg_ssi;    // stage dump of the DBG instantiation: (ncol,112,nlay) x2, (ncol,112)
};

template <typename R> struct SwOut {
    R *swuflx, *swdflx, *swuflxc, *swdflxc, *nirr, *nirf, *parr, *parf, *uvrr, *uvrf, *fswband, *cot[8], *drband, *dfband;
};

enum SwErr { SWERR_PLEV = 12, SWERR_ALB = 13, SWERR_AER = 14 };

// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sw_validate(SwArgs<R> A)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const int ld = A.ld, nlay = A.nlay;
    uint32_t err = 0;
    const R *chk[12] = {A.play, A.tlay, A.h2o, A.o3, A.co2, A.ch4, A.o2, A.cld, A.ciwp, A.clwp, A.rei, A.rel};
    int cftop = 0;
    for (int lay = 0; lay < nlay; lay++) {
        const size_t i = (size_t)lay * ld + col;
#pragma unroll
        for (int k = 0; k < 12; k++)
            if (chk[k][i] < 0) err |= 1u << k;
        if (A.plev[i] < 0) err |= 1u << SWERR_PLEV;
        if (A.cld[i] > 0) cftop = lay + 1;
    }
    if (A.plev[(size_t)nlay * ld + col] < 0) err |= 1u << SWERR_PLEV;
    if (A.asdir[col] < 0 || A.aldir[col] < 0 || A.asdif[col] < 0 || A.aldif[col] < 0) err |= 1u << SWERR_ALB;
    const bool cloudy = cftop > 0;
    A.colcloudy[col] = (uint8_t)cftop;       // 1 + the highest layer with cloud fraction (lw_kernels.hpp k_validate_pwv)
    for (int k = 0; k < 4; k++) A.clearCounts[(size_t)k * ld + col] = cloudy ? 0 : NG_SW;   // rrtmg_sw_rad.F90:1520-1523
    if (err) atomicOr(A.err, err);
}

// aerosol assertions (SW/rrtmg_sw_rad.F90:380-383), one thread per (layer, column) so that the 2 x 14 band planes are read coalesced
template <typename R>
__global__ void __launch_bounds__(256) k_sw_validate_aer(SwArgs<R> A)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lay = blockIdx.y;
    if (col >= A.ncol) return;
    bool bad = false;
    for (int ib = 0; ib < NB_SW; ib++) {
        const size_t j = ((size_t)ib * A.nlay + lay) * A.ld + col;
        bad |= A.tauaer[j] < 0 || A.ssaaer[j] < 0;
    }
    if (bad) atomicOr(A.err, 1u << SWERR_AER);
}

// SW/rrtmg_sw_rad.F90:1370-1387 (column amounts) + SW/rrtmg_sw_setcoef.F90:89-241
template <typename R>
__global__ void __launch_bounds__(256) k_sw_setcoef(SwArgs<R> A, const SwDev<R> *__restrict__ T)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int lay = blockIdx.y;
    if (col >= A.ncol) return;
    const int ld = A.ld, n = A.ncol, nlay = A.nlay;
    const size_t i = (size_t)lay * ld + A.perm[col];          // API arrays: original column; workspace: compacted position
    const R amd = (R)28.9660, amw = (R)18.0160, stpfac = (R)296. / (R)1013.;
    const R pavel = A.play[i], tavel = A.tlay[i], h = A.h2o[i];
    const R coldry = (A.plev[i] - A.plev[i + ld]) * (R)1.e3 * T->avogad /
                     ((R)1.e2 * T->grav * (((R)1. - h) * amd + h * amw) * ((R)1. + h));
    R colh2o = coldry * h, colco2 = coldry * A.co2[i], colo3 = coldry * A.o3[i], colch4 = coldry * A.ch4[i], colo2 = coldry * A.o2[i];
    const R plog = gr_log<R>(pavel);
    const int jp = clampi((int)((R)36. - (R)5 * (plog + (R)0.04)), 1, 58);
    const R fp = (R)5. * (T->preflog[jp - 1] - plog);
    const R d0 = (tavel - T->tref[jp - 1]) / (R)15.;
    const int jt = clampi((int)((R)3. + d0), 1, 4);
    const R ft = d0 - (R)(jt - 3);
    const R d1 = (tavel - T->tref[jp]) / (R)15.;
    const int jt1 = clampi((int)((R)3. + d1), 1, 4);
    const R ft1 = d1 - (R)(jt1 - 3);
    const R water = colh2o / coldry;
    const R scalefac = pavel * stpfac / tavel;
    const bool lower = !(plog <= (R)4.56);
    R forfac = scalefac / ((R)1. + water), forfrac, selffac = 0, selffrac = 0;
    int indfor, indself = 1;
    if (!lower) {
        indfor = 3;
        forfrac = (tavel - (R)188.) / (R)36. - (R)1.;
    } else {
        R factor = ((R)332. - tavel) / (R)36.;
        indfor = clampi((int)factor, 1, 2);
        forfrac = factor - (R)indfor;
        selffac = water * forfac;
        factor = (tavel - (R)188.) / (R)7.2;
        indself = clampi((int)factor - 7, 1, 9);
        selffrac = factor - (R)(indself + 7);
    }
    colh2o = (R)1.e-20 * colh2o; colco2 = (R)1.e-20 * colco2; colo3 = (R)1.e-20 * colo3; colch4 = (R)1.e-20 * colch4;
    colo2 = (R)1.e-20 * colo2;
    const R colmol = (R)1.e-20 * coldry + colh2o;
    if (colco2 == 0) colco2 = (R)1.e-32 * coldry;
    if (colch4 == 0) colch4 = (R)1.e-32 * coldry;
    if (colo2 == 0) colo2 = (R)1.e-32 * coldry;
    const R compfp = (R)1. - fp;
    R *sc = A.sc + (size_t)lay * n + col;
    const size_t fs = (size_t)nlay * n;
    sc[SW_FAC10 * fs] = compfp * ft; sc[SW_FAC00 * fs] = compfp * ((R)1. - ft);
    sc[SW_FAC11 * fs] = fp * ft1; sc[SW_FAC01 * fs] = fp * ((R)1. - ft1);
    sc[SW_COLH2O * fs] = colh2o; sc[SW_COLCO2 * fs] = colco2; sc[SW_COLO3 * fs] = colo3; sc[SW_COLCH4 * fs] = colch4;
    sc[SW_COLO2 * fs] = colo2; sc[SW_COLMOL * fs] = colmol;
    sc[SW_SELFFAC * fs] = selffac; sc[SW_SELFFRAC * fs] = selffrac; sc[SW_FORFAC * fs] = forfac; sc[SW_FORFRAC * fs] = forfrac;
    A.scidx[(size_t)lay * n + col] = pack_idx(jp, jt, jt1, indfor, indself, 0, lower ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------
// band traits (SW/rrtmg_sw_taumol.F90, one taumolNN each)
//   KIND 0: no key species; 1: single key species (65 / 235 rows); 2: binary (585 / 1175 rows), strrat
//   SRC  0: source independent of the column; 1: at the lower-atmosphere reference layer (layreffr);
//        2: at the upper-atmosphere reference layer
// ---------------------------------------------------------------------------------------------------
struct SwB16 { static constexpr int JB = 16, NG = 6,  G0 = 0,   LOK = 2, LOA = G_H2O, LOB = G_CH4, UPK = 1, UPA = G_CH4, UPB = G_NONE, NFOR = 3, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 252.131;    static constexpr bool LOCONT = true,  UPFOR = false; };
struct SwB17 { static constexpr int JB = 17, NG = 12, G0 = 6,   LOK = 2, LOA = G_H2O, LOB = G_CO2, UPK = 2, UPA = G_H2O, UPB = G_CO2,  NFOR = 4, SRC = 2, LREF = 30, NSRC = 5; static constexpr double STR = 0.364641;   static constexpr bool LOCONT = true,  UPFOR = true;  };
struct SwB18 { static constexpr int JB = 18, NG = 8,  G0 = 18,  LOK = 2, LOA = G_H2O, LOB = G_CH4, UPK = 1, UPA = G_CH4, UPB = G_NONE, NFOR = 3, SRC = 1, LREF = 6,  NSRC = 9; static constexpr double STR = 38.9589;    static constexpr bool LOCONT = true,  UPFOR = false; };
struct SwB19 { static constexpr int JB = 19, NG = 8,  G0 = 26,  LOK = 2, LOA = G_H2O, LOB = G_CO2, UPK = 1, UPA = G_CO2, UPB = G_NONE, NFOR = 3, SRC = 1, LREF = 3,  NSRC = 9; static constexpr double STR = 5.49281;    static constexpr bool LOCONT = true,  UPFOR = false; };
struct SwB20 { static constexpr int JB = 20, NG = 10, G0 = 34,  LOK = 1, LOA = G_H2O, LOB = G_NONE, UPK = 1, UPA = G_H2O, UPB = G_NONE, NFOR = 4, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;          static constexpr bool LOCONT = true,  UPFOR = true;  };
struct SwB21 { static constexpr int JB = 21, NG = 10, G0 = 44,  LOK = 2, LOA = G_H2O, LOB = G_CO2, UPK = 2, UPA = G_H2O, UPB = G_CO2,  NFOR = 4, SRC = 1, LREF = 8,  NSRC = 9; static constexpr double STR = 0.0045321;  static constexpr bool LOCONT = true,  UPFOR = true;  };
struct SwB22 { static constexpr int JB = 22, NG = 2,  G0 = 54,  LOK = 2, LOA = G_H2O, LOB = G_O2,  UPK = 1, UPA = G_O2,  UPB = G_NONE, NFOR = 3, SRC = 1, LREF = 2,  NSRC = 9; static constexpr double STR = 1.6 * 0.022708; static constexpr bool LOCONT = true, UPFOR = false; };
struct SwB23 { static constexpr int JB = 23, NG = 10, G0 = 56,  LOK = 1, LOA = G_H2O, LOB = G_NONE, UPK = 0, UPA = G_NONE, UPB = G_NONE, NFOR = 3, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;          static constexpr bool LOCONT = true,  UPFOR = false; };
struct SwB24 { static constexpr int JB = 24, NG = 8,  G0 = 66,  LOK = 2, LOA = G_H2O, LOB = G_O2,  UPK = 1, UPA = G_O2,  UPB = G_NONE, NFOR = 3, SRC = 1, LREF = 1,  NSRC = 9; static constexpr double STR = 0.124692;   static constexpr bool LOCONT = true,  UPFOR = false; };
struct SwB25 { static constexpr int JB = 25, NG = 6,  G0 = 74,  LOK = 1, LOA = G_H2O, LOB = G_NONE, UPK = 0, UPA = G_NONE, UPB = G_NONE, NFOR = 0, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;          static constexpr bool LOCONT = false, UPFOR = false; };
struct SwB26 { static constexpr int JB = 26, NG = 6,  G0 = 80,  LOK = 0, LOA = G_NONE, LOB = G_NONE, UPK = 0, UPA = G_NONE, UPB = G_NONE, NFOR = 0, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;         static constexpr bool LOCONT = false, UPFOR = false; };
struct SwB27 { static constexpr int JB = 27, NG = 8,  G0 = 86,  LOK = 1, LOA = G_O3,  LOB = G_NONE, UPK = 1, UPA = G_O3,  UPB = G_NONE, NFOR = 0, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;          static constexpr bool LOCONT = false, UPFOR = false; };
struct SwB28 { static constexpr int JB = 28, NG = 6,  G0 = 94,  LOK = 2, LOA = G_O3,  LOB = G_O2,  UPK = 2, UPA = G_O3,  UPB = G_O2,   NFOR = 0, SRC = 2, LREF = 42, NSRC = 5; static constexpr double STR = 6.67029e-07; static constexpr bool LOCONT = false, UPFOR = false; };
struct SwB29 { static constexpr int JB = 29, NG = 12, G0 = 100, LOK = 1, LOA = G_H2O, LOB = G_NONE, UPK = 1, UPA = G_CO2, UPB = G_NONE, NFOR = 4, SRC = 0, LREF = 0,  NSRC = 1; static constexpr double STR = 0;          static constexpr bool LOCONT = true,  UPFOR = false; };

template <typename R> struct SwLayer {
    R fac00, fac01, fac10, fac11, col[5], colmol, selffac, selffrac, forfac, forfrac;
    int jp, jt, jt1, indfor, indself;
    bool lower;
};
template <typename R> GR_DEV void sw_load_layer(const SwArgs<R> &A, int lay, int col, SwLayer<R> &L)
{
    const uint32_t cell = (uint32_t)lay * (uint32_t)A.ncol + (uint32_t)col;
    const uint32_t wb = cell * (uint32_t)sizeof(R);
    const size_t fs = (size_t)A.nlay * A.ncol;
#define SCF(f) ldg(A.sc + (size_t)(f) * fs, wb)
    L.fac00 = SCF(SW_FAC00); L.fac01 = SCF(SW_FAC01); L.fac10 = SCF(SW_FAC10); L.fac11 = SCF(SW_FAC11);
    L.col[G_H2O] = SCF(SW_COLH2O); L.col[G_CO2] = SCF(SW_COLCO2); L.col[G_O3] = SCF(SW_COLO3); L.col[G_CH4] = SCF(SW_COLCH4);
    L.col[G_O2] = SCF(SW_COLO2); L.colmol = SCF(SW_COLMOL);
    L.selffac = SCF(SW_SELFFAC); L.selffrac = SCF(SW_SELFFRAC); L.forfac = SCF(SW_FORFAC); L.forfrac = SCF(SW_FORFRAC);
#undef SCF
    const uint32_t p = ldg(A.scidx, cell * 4u);
    L.jp = p & 63; L.jt = (p >> 6) & 7; L.jt1 = (p >> 9) & 7; L.indfor = (p >> 12) & 3; L.indself = (p >> 14) & 15;
    L.lower = (p >> 23) & 1;
}

// binary-species parameter of SW bands: same (js, fs) for both reference pressures (e.g. :383-388)
template <typename R> struct SwSpec { R speccomb, fs; int js; };
template <typename R> GR_DEV SwSpec<R> sw_spec(R cola, R strrat, R colb, R mult, R oneminus)
{
    SwSpec<R> s;
    s.speccomb = cola + strrat * colb;
    R specparm = cola / s.speccomb;
    if (specparm >= oneminus) specparm = oneminus;
    const R sm = mult * specparm;
    const int j = (int)sm;
    s.js = 1 + j;
    s.fs = sm - (R)j;
    return s;
}

template <typename R> struct SwPrep { SwSpec<R> sp; int ind0, ind1; R cmaj, extra; };

template <typename R, typename B> GR_DEV void sw_prep(const SwDev<R> &T, const SwLayer<R> &L, SwPrep<R> &P)
{
    P.extra = 0; P.cmaj = 0; P.ind0 = 1; P.ind1 = 1; P.sp.fs = 0; P.sp.js = 1; P.sp.speccomb = 0;
    if (L.lower) {
        if constexpr (B::LOK == 2) {
            P.sp = sw_spec<R>(L.col[B::LOA], (R)B::STR, L.col[B::LOB], 8, T.oneminus);
            P.ind0 = ((L.jp - 1) * 5 + (L.jt - 1)) * 9 + P.sp.js;
            P.ind1 = (L.jp * 5 + (L.jt1 - 1)) * 9 + P.sp.js;
            P.cmaj = P.sp.speccomb;
        } else if constexpr (B::LOK == 1) {
            P.ind0 = ((L.jp - 1) * 5 + (L.jt - 1)) + 1;
            P.ind1 = (L.jp * 5 + (L.jt1 - 1)) + 1;
            P.cmaj = L.col[B::LOA];
            if constexpr (B::JB == 23) P.cmaj = P.cmaj * (R)1.029;      // givfac (:1326)
        }
    } else {
        if constexpr (B::UPK == 2) {
            P.sp = sw_spec<R>(L.col[B::UPA], (R)B::STR, L.col[B::UPB], 4, T.oneminus);
            P.ind0 = ((L.jp - 13) * 5 + (L.jt - 1)) * 5 + P.sp.js;
            P.ind1 = ((L.jp - 12) * 5 + (L.jt1 - 1)) * 5 + P.sp.js;
            P.cmaj = P.sp.speccomb;
        } else if constexpr (B::UPK == 1) {
            P.ind0 = ((L.jp - 13) * 5 + (L.jt - 1)) + 1;
            P.ind1 = ((L.jp - 12) * 5 + (L.jt1 - 1)) + 1;
            P.cmaj = L.col[B::UPA];
            if constexpr (B::JB == 22) P.cmaj = P.cmaj * (R)1.6;        // o2adj (:1277)
        }
    }
    if constexpr (B::JB == 22) P.extra = (R)4.35e-4 * L.col[G_O2] / ((R)350.0 * (R)2.0);   // o2cont (:1228,1273)
}

// Rayleigh optical depth of W g-points starting at `go`: one value per g-point, except band 24 below the tropopause, where rayla is
// interpolated in the binary-species parameter (js, fs) (:1467)
template <typename R, typename B, int W>
GR_DEV void sw_rayl(const SwDev<R> &T, bool lower, R colmol, int js, R fs, int go, R (&ray)[W])
{
    constexpr int S = pad4(B::NG);
    const SwBandTab<R> &Bt = T.b[B::JB - 15];
    R t[W];
    if constexpr (B::JB == 24) {
        if (lower) linw<R, W, S>(t, fs, Bt.rayl, js - 1, go);      // rayla(ig, js..js+1)
        else ldw<R, W>(Bt.raylb, (uint32_t)go * (uint32_t)sizeof(R), t);
    } else {
        ldw<R, W>(Bt.rayl, (uint32_t)go * (uint32_t)sizeof(R), t);
    }
#pragma unroll
    for (int j = 0; j < W; j++) ray[j] = colmol * t[j];
}

// gas optical depth tau[] and Rayleigh optical depth ray[] of W g-points starting at `go`
template <typename R, typename B, int W>
GR_DEV void sw_eval(const SwDev<R> &T, const SwLayer<R> &L, const SwPrep<R> &P, int go, R (&tau)[W], R (&ray)[W])
{
    constexpr int S = pad4(B::NG);
    const SwBandTab<R> &Bt = T.b[B::JB - 15];
    const int kind = L.lower ? B::LOK : B::UPK;
    R m[W];
#pragma unroll
    for (int j = 0; j < W; j++) m[j] = 0;
    // lower / upper atmosphere differ by lane: ONE wave-uniform base (absa) + the lane's byte offset, which carries the distance to absb for
    // the upper lanes (both tables sit in the context's one table allocation, absa first: set_tables_sw).  A per-lane choice between the two
    // POINTERS made every row a flat load with a 64-bit lane address behind a wait for everything in flight.
    const R *tab = Bt.absa;
    const uint32_t tsel = L.lower ? 0u : (uint32_t)(reinterpret_cast<const char *>(Bt.absb) - reinterpret_cast<const char *>(Bt.absa));
    if (kind == 2) {
        const int off = L.lower ? 9 : 5;
        const R c0 = (R)1. - P.sp.fs, c1 = P.sp.fs;
        // reference order (:398-410): fac000 a(ind0) + fac100 a(ind0+1) + fac010 a(ind0+off) + fac110 a(ind0+off+1) + same for ind1
        axw<R, W, S, true>(m, c0 * L.fac00, tab, P.ind0 - 1, go, tsel);
        axw<R, W, S, false>(m, c1 * L.fac00, tab, P.ind0, go, tsel);
        axw<R, W, S, false>(m, c0 * L.fac10, tab, P.ind0 - 1 + off, go, tsel);
        axw<R, W, S, false>(m, c1 * L.fac10, tab, P.ind0 + off, go, tsel);
        axw<R, W, S, false>(m, c0 * L.fac01, tab, P.ind1 - 1, go, tsel);
        axw<R, W, S, false>(m, c1 * L.fac01, tab, P.ind1, go, tsel);
        axw<R, W, S, false>(m, c0 * L.fac11, tab, P.ind1 - 1 + off, go, tsel);
        axw<R, W, S, false>(m, c1 * L.fac11, tab, P.ind1 + off, go, tsel);
    } else if (kind == 1) {
        axw<R, W, S, true>(m, L.fac00, tab, P.ind0 - 1, go, tsel);
        axw<R, W, S, false>(m, L.fac10, tab, P.ind0, go, tsel);
        axw<R, W, S, false>(m, L.fac01, tab, P.ind1 - 1, go, tsel);
        axw<R, W, S, false>(m, L.fac11, tab, P.ind1, go, tsel);
    }
    R cont[W];
#pragma unroll
    for (int j = 0; j < W; j++) cont[j] = 0;
    if (L.lower) {
        if constexpr (B::LOCONT) {
            R t[W];
            linw<R, W, S>(t, L.selffrac, Bt.selfref, L.indself - 1, go);
#pragma unroll
            for (int j = 0; j < W; j++) cont[j] = L.selffac * t[j];
            linw<R, W, S>(t, L.forfrac, Bt.forref, L.indfor - 1, go);
#pragma unroll
            for (int j = 0; j < W; j++) cont[j] = cont[j] + L.forfac * t[j];
        }
    } else {
        if constexpr (B::UPFOR) {
            R t[W];
            linw<R, W, S>(t, L.forfrac, Bt.forref, L.indfor - 1, go);
#pragma unroll
            for (int j = 0; j < W; j++) cont[j] = L.forfac * t[j];
        }
    }
#pragma unroll
    for (int j = 0; j < W; j++) tau[j] = P.cmaj * m[j] + L.col[G_H2O] * cont[j] + P.extra;
    // band-specific extra absorbers
    if constexpr (B::JB == 20) {           // + colch4 * absch4 (:913,941)
        R x[W]; ldw<R, W>(Bt.x0, (uint32_t)go * (uint32_t)sizeof(R), x);
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = tau[j] + L.col[G_CH4] * x[j];
    } else if constexpr (B::JB == 24 || B::JB == 25) {   // + colo3 * abso3a | abso3b (:1478,1513 / :1591,1602)
        R x[W]; ldw<R, W>(Bt.x0, (uint32_t)go * (uint32_t)sizeof(R) + (L.lower ? 0u : (uint32_t)(reinterpret_cast<const char *>(Bt.x1) - reinterpret_cast<const char *>(Bt.x0))), x);
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = tau[j] + L.col[G_O3] * x[j];
    } else if constexpr (B::JB == 29) {   // lower + colco2 * absco2, upper + colh2o * absh2o (:2031,2050)
        R x[W]; ldw<R, W>(Bt.x0, (uint32_t)go * (uint32_t)sizeof(R) + (L.lower ? 0u : (uint32_t)(reinterpret_cast<const char *>(Bt.x1) - reinterpret_cast<const char *>(Bt.x0))), x);
        const R c = L.lower ? L.col[G_CO2] : L.col[G_H2O];
#pragma unroll
        for (int j = 0; j < W; j++) tau[j] = tau[j] + c * x[j];
    }
    sw_rayl<R, B, W>(T, L.lower, L.colmol, P.sp.js, P.sp.fs, go, ray);
}
// Fast-path arithmetic of the fp32 instantiation: 1-ulp hardware reciprocal / sqrt / exp2 instead of the correctly rounded
// (10-instruction) IEEE sequences.  The per-cell two-stream needs ~16 divisions, which made half of the instruction count;
// the differences (<= 2 ulp per operation) are of the size of fp32 rounding itself and far inside the parity tolerance of
// the fp32 path (tests/test_gpu_sw.py).  The fp64 instantiation: exp from the library, division / sqrt by gr_div64 / gr_sqrt64 (<= 1 ulp; parity <= 1e-6 W m-2).
template <typename R> GR_DEV R f_rcp(R x);
template <> GR_DEV float f_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> GR_DEV double f_rcp<double>(double x) { return gr_rcp64(x); }
template <typename R> GR_DEV R f_div(R a, R b);
template <> GR_DEV float f_div<float>(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
template <> GR_DEV double f_div<double>(double a, double b) { return gr_div64(a, b); }
template <typename R> GR_DEV R f_sqrt(R x);
template <> GR_DEV float f_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> GR_DEV double f_sqrt<double>(double x) { return gr_sqrt64(x); }
template <typename R> GR_DEV R f_exp(R x);
template <> GR_DEV float f_exp<float>(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
template <> GR_DEV double f_exp<double>(double x) { return exp(x); }

// `zwo >= zwcrit` of reftra_sw (SW/rrtmg_sw_spcvmc.F90:1207,1260-1262).  The reference evaluates the un-scaled single
// scattering albedo zwo = w / (1 - (1-w) (g/(1-g))^2) in real(8) even in the default-real build, rounds it to default real
// and compares.  fp64: exactly that.  fp32: the same predicate cross-multiplied (no fp64 division), comparing against the
// rounding boundary of the float constant.
template <typename R> GR_DEV bool sw_conservative(R zw, R zg);
template <> GR_DEV bool sw_conservative<double>(double zw, double zg)
{
    const double q = zg / (1.0 - zg);
    const double zwo = zw / (1.0 - (1.0 - zw) * (q * q));
    return zwo >= 0.9999995;
}
template <> GR_DEV bool sw_conservative<float>(float zw, float zg)
{
    // zwo = w / (1 - (1 - w) q^2), q = g / (1 - g): for q^2 <= 1 (g <= 1/2 - every delta-scaled asymmetry factor, g / (1 + g), is) zwo <= 1
    // and zwo >= c needs w (1 - c q^2) >= c (1 - q^2), i.e. w >= c; so w < 0.99 with g < 0.49 can never be conservative: decided
    // without the double-precision arithmetic below (exactly the same predicate, cheaper for most cells)
    if (zw < 0.99f && zg < 0.49f) return false;
    // (float)x >= c  <=>  x >= midpoint(prev(c), c) =: m   (c = 0.9999995f = 1 - 8 * 2^-24; ulp below 1 is 2^-24)
    const double m = (double)0.9999995f - 0.5 * 5.9604644775390625e-08;
    const double w = zw, g = zg, omg = 1.0 - g;
    const double a = omg * omg, d = a - (1.0 - w) * (g * g);       // zwo = w a / d
    if (d > 0.0) return w * a >= m * d;
    return d == 0.0 ? w > 0.0 : false;                            // +inf >= c ; negative zwo never is
}

// layer reflectance / transmittance, PIFM two-stream (SW/rrtmg_sw_spcvmc.F90:1236-1362)
// exp(-5): the clamped exponential of reftra, evaluated by the same instruction as the unclamped ones (not folded by the compiler)
template <typename R> GR_DEV R sw_exp_m5()
{
    R five = (R)5.;
    asm volatile("" : "+v"(five));
    return f_exp<R>(-five);
}

// dbt = exp(-zto1 / prmuz), the direct-beam transmittance of the layer (spcvmc :449-456, :547-559), comes out of the same call: where
// reftra's own exp(-min(zto1 / prmuz, 5)) is not replaced by its series (<= od_lo) or clamped, it IS that exponential
template <typename R> GR_DEV void sw_reftra(R zto1, R zw, R zg, R prmuz, R rmuz, R &ref, R &refd, R &tra, R &trad, R &dbt)
{
    // rmuz = 1 / prmuz (hoisted by the caller)
    const R eps = (R)1.e-08, od_lo = (R)0.06;
    const R zg3 = (R)3. * zg;
    const R zgamma1 = ((R)8. - zw * ((R)5. + zg3)) * (R)0.25;
    const R zgamma2 = (R)3. * (zw * ((R)1. - zg)) * (R)0.25;
    const R zgamma3 = ((R)2. - zg3 * prmuz) * (R)0.25;
    const R zgamma4 = (R)1. - zgamma3;
    if (sw_conservative<R>(zw, zg)) {
        const R za = zgamma1 * prmuz, za1 = za - zgamma3, zgt = zgamma1 * zto1;
        const R zx = zto1 * rmuz;
        dbt = f_exp<R>(-zx);
        const R ze2 = zx > (R)500. ? f_exp<R>(-(R)500.) : dbt;
        const R r1 = f_rcp<R>((R)1. + zgt);
        ref = (zgt - za1 * ((R)1. - ze2)) * r1;
        tra = (R)1. - ref;
        refd = zgt * r1;
        trad = (R)1. - refd;
        if (ze2 == (R)1.) { ref = 0; tra = 1; refd = 0; trad = 1; }
    } else {
        const R za1 = zgamma1 * zgamma4 + zgamma2 * zgamma3, za2 = zgamma1 * zgamma3 + zgamma2 * zgamma4;
        const R zrk = f_sqrt<R>(zgamma1 * zgamma1 - zgamma2 * zgamma2);
        const R zrp = zrk * prmuz, zrp1 = (R)1. + zrp, zrm1 = (R)1. - zrp, zrk2 = (R)2. * zrk;
        const R zrpp = (R)1. - zrp * zrp, zrkg = zrk + zgamma1;
        const R zr1 = zrm1 * (za2 + zrk * zgamma3), zr2 = zrp1 * (za2 - zrk * zgamma3), zr3 = zrk2 * (zgamma3 - za2 * prmuz);
        const R zr4 = zrpp * zrkg, zr5 = zrpp * (zrk - zgamma1);
        const R zt1 = zrp1 * (za1 + zrk * zgamma4), zt2 = zrm1 * (za1 - zrk * zgamma4), zt3 = zrk2 * (zgamma4 + za1 * prmuz);
        const R rzrkg = f_rcp<R>(zrkg);
        const R zbeta = (zgamma1 - zrk) * rzrkg;
        R ze1 = zrk * zto1; ze1 = ze1 > (R)5. ? (R)5. : ze1;
        const R zx = zto1 * rmuz;
        const R ze2 = zx > (R)5. ? (R)5. : zx;
        dbt = f_exp<R>(-zx);
        const R zem1 = ze1 <= od_lo ? (R)1. - ze1 + (R)0.5 * ze1 * ze1 : f_exp<R>(-ze1);
        const R zep1 = f_rcp<R>(zem1);
        const R zem2 = ze2 <= od_lo ? (R)1. - ze2 + (R)0.5 * ze2 * ze2 : (zx > (R)5. ? sw_exp_m5<R>() : dbt);
        const R zep2 = f_rcp<R>(zem2);
        const R zdenr = zr4 * zep1 + zr5 * zem1;           // zdent == zdenr (zt4 = zr4, zt5 = zr5)
        if (zdenr >= -eps && zdenr <= eps) { ref = eps; tra = zem2; }
        else {
            const R rden = f_rcp<R>(zdenr);
            ref = zw * (zr1 * zep1 - zr2 * zem1 - zr3 * zem2) * rden;
            tra = zem2 - zem2 * zw * (zt1 * zep1 - zt2 * zem1 - zt3 * zep2) * rden;
        }
        const R zemm = zem1 * zem1;
        const R zdend = f_rcp<R>((R)1. - zbeta * zemm) * rzrkg;
        refd = zgamma2 * ((R)1. - zemm) * zdend;
        trad = zrk2 * zem1 * zdend;
    }
}

// One cell's optics.  Both sweeps of the band body form them (the second from the parked gas optical depth).
template <typename R> struct SwCell { R tau, om, g, ref, refd, tra, trad, dbt; };
// clear sky incl. aerosol, delta-scaled with f = g^2 (SW/rrtmg_sw_spcvmc.F90:413-437); layer properties; direct-beam transmittance
template <typename R> GR_DEV void sw_cell_clear(R tg, R tr, R ta, R om, R as, R prmu0, R rmu0, SwCell<R> &c)
{
    R ztauo = tr + tg + ta;
    R zomco = tr + ta * om;
    R zgco = f_div<R>(as * om * ta, zomco);
    zomco = f_div<R>(zomco, ztauo);
    const R zf = zgco * zgco, zwf = zomco * zf;
    ztauo = ((R)1. - zwf) * ztauo;
    zomco = f_div<R>(zomco - zwf, (R)1. - zwf);
    zgco = f_div<R>(zgco - zf, (R)1. - zf);
    c.tau = ztauo; c.om = zomco; c.g = zgco;
    sw_reftra<R>(ztauo, zomco, zgco, prmu0, rmu0, c.ref, c.refd, c.tra, c.trad, c.dbt);
}
// total sky of a cloudy cell: the (already delta-scaled) cloud optics added to the clear-sky ones (:512-536, 541, 547-559)
template <typename R> GR_DEV void sw_cell_cloud(const SwCell<R> &c, R tc, R oc, R gc, R prmu0, R rmu0, SwCell<R> &t)
{
    R g2 = c.tau * c.om * c.g + tc * oc * gc;
    R o2 = c.tau * c.om + tc * oc;
    const R t2 = c.tau + tc;
    g2 = f_div<R>(g2, o2); o2 = f_div<R>(o2, t2);
    t.tau = t2; t.om = o2; t.g = g2;
    sw_reftra<R>(t2, o2, g2, prmu0, rmu0, t.ref, t.refd, t.tra, t.trad, t.dbt);
}

// ---------------------------------------------------------------------------------------------------
// fused band body.  Vertical index jk = 0 (TOA layer) .. nlay-1 (surface layer) of spcvmc == API layer lay = nlay-1-jk.
// Two sweeps over the layers of a column, one lane per column, all g-points of the band in registers:
//   A (surface -> TOA): k-distribution, cell optics, two-stream layer properties (reftra) and the UPWARD adding recurrences
//     (vrtqdr's prup / prupd, :1453-1505).  Parked per cell (plane q): 0 ref, 1 refd, 2 tra, 3 trad, 4 dbt of the layer (sign bit:
//     the cell is cloudy); 5 prup, 6 prupd at the layer's upper boundary.  Cloudy columns: 7..11 the layer properties of the total
//     sky in cloudy cells; 12, 13 prup / prupd of the total sky from the sub-column's lowest cloudy cell upwards (below it they
//     equal the clear sky's).
//   B (TOA -> surface): the DOWNWARD recurrences (ptdbt / ztdn / prdnd, :1530-1572) run in registers, and the fluxes of every
//     level come from the two sides (:1576-1586), band-integrated (:467-502).
// The recurrences are the reference's, evaluated in the other order: vrtqdr runs down first and parks three values per level, up
// first parks two - 28 instead of 32 bytes per clear-sky fp32 cell of a kernel that is bound by exactly this traffic
// (profiles/r02_sw_parked_cells.md, which also has the variants that re-form the layer properties in sweep B instead of parking
// them: 12 bytes per cell, but the second reftra per cell costs more than the HBM time it saves).
// ---------------------------------------------------------------------------------------------------
template <typename R, typename B, bool CLD, bool DBG>
GR_DEV void sw_band_body(const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV, int col, int nclear)
{
    constexpr int NG = B::NG, JB = B::JB, IBM = B::JB - 15, G0 = B::G0;
    constexpr int W = NG >= 4 ? 4 : 2;
    constexpr int NQ = (NG + W - 1) / W;
    constexpr int S = pad4(NG);
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const uint32_t ucol = (uint32_t)col;
    const uint32_t cb = ucol * (uint32_t)sizeof(R);          // workspace rows: compacted position
    const int pc = ldg(A.perm, ucol * 4u);                   // API arrays: original column
    const uint32_t cba = (uint32_t)pc * (uint32_t)sizeof(R);
    const SwBandTab<R> &Bt = T.b[IBM];
    // CLD kernels only see cloudy columns (DBG: all columns), so this is true there; it is deliberately left a run-time
    // value: as a compile-time constant the if-converted code needs 290 registers and halves the occupancy
    int ncl_opaque = nclear;
    asm volatile("" : "+s"(ncl_opaque));        // hides the fact from the optimiser (the caller already tested col >= nclear)
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
            // reference layer search (e.g. :585-640 lower, :439-474 upper); laytrop = number of lower layers
            int laytrop = 0;
            for (int lay = 0; lay < nlay; lay++) laytrop += (int)((ldg(A.scidx, ((uint32_t)lay * (uint32_t)n + ucol) * 4u) >> 23) & 1u);
            int lsol;   // 0-based API layer
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
            const int go = q * W;
            if constexpr (B::NSRC == 1) {
                ldw<R, W>(Bt.sflux, (uint32_t)go * (uint32_t)sizeof(R), sf); ldw<R, W>(Bt.facb, (uint32_t)go * (uint32_t)sizeof(R), fb);
                ldw<R, W>(Bt.snsp, (uint32_t)go * (uint32_t)sizeof(R), sd); ldw<R, W>(Bt.irrad, (uint32_t)go * (uint32_t)sizeof(R), ir);
            } else {
                linw<R, W, S>(sf, fs, Bt.sflux, js - 1, go); linw<R, W, S>(fb, fs, Bt.facb, js - 1, go);
                linw<R, W, S>(sd, fs, Bt.snsp, js - 1, go); linw<R, W, S>(ir, fs, Bt.irrad, js - 1, go);
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = go + j;
                if (g >= NG) continue;
                R src;
                if (SV.isolvar < 0) src = sf[j];
                else if (SV.isolvar <= 2) src = SV.svar_f * fb[j] + SV.svar_s * sd[j] + SV.svar_i * ir[j];
                else src = SV.svar_bnd[IBM] * fb[j] + SV.svar_bnd[IBM] * sd[j] + SV.svar_bnd[IBM] * ir[j];
                if (DBG) A.dbg_ssi[(size_t)pc * NG_SW + G0 + g] = src;
                zinc[g] = SV.adjflux[IBM] * src;
            }
        }
    }

    // uniform bases of the parked-cell planes of this band
    const size_t bandoff = (size_t)G0 * nlay * n;
    // parked cell planes are tiled by 256-column block: [block][layer][g][256], so that a block's scratch is one contiguous
    // run per plane instead of 1 KB pieces 400 KB apart (TLB reach, DRAM page locality)
    const uint32_t npad = ((uint32_t)n + 255u) & ~255u;
    const size_t plane = (size_t)NG_SW * nlay * npad;
    R *const cellb = A.cell + (size_t)G0 * nlay * npad;
    const uint32_t tbase = (ucol >> 8) * (uint32_t)nlay * (uint32_t)NG * 256u + (ucol & 255u);
    const R *const tcb = A.taucmc + bandoff, *const ocb = A.ssacmc + bandoff, *const gcb = A.asmcmc + bandoff;
#define CELL(q) (cellb + (size_t)(q) * plane)
#define PST(q, off, v) stg_nt(CELL(q), off, v)
#define PLD(q, off) ldg_nt(CELL(q), off)
#define CT4(lay_, g_) ((tbase + ((uint32_t)(lay_) * (uint32_t)NG + (uint32_t)(g_)) * 256u) * (uint32_t)sizeof(R))

    // ---- sweep A: surface -> TOA -----------------------------------------------------------------------
    R prup[NG], prupd[NG], prupT[NG], prupdT[NG];
    int lowc[NG];            // lowest cloudy layer of sub-column g: the parked total-sky planes hold values from there upwards
#pragma unroll
    for (int g = 0; g < NG; g++) { prup[g] = albp; prupd[g] = albd; prupT[g] = albp; prupdT[g] = albd; lowc[g] = 0x7fffffff; }
    for (int lay = 0; lay < nlay; lay++) {
        SwLayer<R> L;
        sw_load_layer<R>(A, lay, col, L);
        SwPrep<R> P;
        sw_prep<R, B>(T, L, P);
        R ta = 0, om = 1, as = 0;
        if (A.iaer == 10) {
            const uint32_t ab = ((uint32_t)lay * (uint32_t)ld + (uint32_t)pc) * (uint32_t)sizeof(R);
            const size_t bo = (size_t)(IBM - 1) * nlay * ld;
            ta = ldg(A.tauaer + bo, ab); om = ldg(A.ssaaer + bo, ab); as = ldg(A.asmaer + bo, ab);
        }
        const uint32_t cell0 = ((uint32_t)lay * (uint32_t)NG) * (uint32_t)n + ucol;
        const bool laycld = CLD && ccol && ldg(A.laycloudy, (uint32_t)lay * (uint32_t)n + ucol) != 0;
        const bool wlc = CLD && __ballot(laycld) != 0;      // some column of the wave has cloud in this layer (wave-uniform)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            // cloud optics of the group's cells: requested together behind wave-uniform tests and selected afterwards (a per-lane
            // branch around each load made the loads wait for one another); lanes without cloud here read cells nobody wrote
            R tcv[W], ocv[W], gcv[W];
#pragma unroll
            for (int j = 0; j < W; j++) { tcv[j] = 0; ocv[j] = 0; gcv[j] = 0; }
            if (wlc) {
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (q * W + j < NG) tcv[j] = ldg(tcb, (cell0 + (uint32_t)(q * W + j) * (uint32_t)n) * (uint32_t)sizeof(R));
            }
            R tg[W], tr[W];
            sw_eval<R, B, W>(T, L, P, q * W, tg, tr);
            if (wlc) {
                bool anyc = false;
#pragma unroll
                for (int j = 0; j < W; j++) { tcv[j] = laycld ? tcv[j] : (R)0; anyc = anyc || tcv[j] > 0; }
                if (__ballot(anyc) != 0) {
#pragma unroll
                    for (int j = 0; j < W; j++)
                        if (q * W + j < NG) {
                            const uint32_t o4 = (cell0 + (uint32_t)(q * W + j) * (uint32_t)n) * (uint32_t)sizeof(R);
                            ocv[j] = ldg(ocb, o4); gcv[j] = ldg(gcb, o4);
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const uint32_t ct4 = CT4(lay, g);
                if (DBG) {
                    const size_t o = ((size_t)pc * NG_SW + (G0 + g)) * nlay + lay;       // Fortran (nlay,112,ncol)
                    A.dbg_taug[o] = tg[j]; A.dbg_taur[o] = tr[j];
                }
                SwCell<R> c;
                sw_cell_clear<R>(tg[j], tr[j], ta, om, as, prmu0, rmu0, c);
                const bool cellcld = CLD && tcv[j] > 0;
                // the sign bit of the parked direct-beam transmittance (>= 0) carries "this cell is cloudy" to sweep B
                PST(0, ct4, c.ref); PST(1, ct4, c.refd); PST(2, ct4, c.tra); PST(3, ct4, c.trad); PST(4, ct4, cellcld && ccol ? -c.dbt : c.dbt);
                // upward adding (:1453-1505): reflectances of everything below the cell's upper boundary
                {
                    const R zrj = f_rcp<R>((R)1. - prupd[g] * c.refd);
                    const R pu = c.ref + (c.trad * ((c.tra - c.dbt) * prupd[g] + c.dbt * prup[g])) * zrj;
                    const R pd = c.refd + c.trad * c.trad * prupd[g] * zrj;
                    prup[g] = pu; prupd[g] = pd;
                }
                PST(5, ct4, prup[g]); PST(6, ct4, prupd[g]);
                if constexpr (CLD) {
                    const R tc = tcv[j];
                    // Below the lowest cloudy cell of a sub-column the total-sky upward state IS the clear-sky one (same
                    // recurrences, same inputs): nothing is computed or parked for it there
                    if (cellcld && lowc[g] > lay) lowc[g] = lay;
                    const bool divg = ccol && lowc[g] <= lay;
                    if (ccol && !divg) { prupT[g] = prup[g]; prupdT[g] = prupd[g]; }
                    if (divg) {
                        SwCell<R> t = c;
                        if (cellcld) {
                            sw_cell_cloud<R>(c, tc, ocv[j], gcv[j], prmu0, rmu0, t);
                            PST(7, ct4, t.ref); PST(8, ct4, t.refd); PST(9, ct4, t.tra); PST(10, ct4, t.trad); PST(11, ct4, t.dbt);
                        }
                        const R zrj = f_rcp<R>((R)1. - prupdT[g] * t.refd);
                        const R pu = t.ref + (t.trad * ((t.tra - t.dbt) * prupdT[g] + t.dbt * prupT[g])) * zrj;
                        const R pd = t.refd + t.trad * t.trad * prupdT[g] * zrj;
                        prupT[g] = pu; prupdT[g] = pd;
                        PST(12, ct4, pu); PST(13, ct4, pd);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- sweep B: TOA -> surface (:1530-1572 downward recurrences, :1576-1586 fluxes, band integration :467-502) ----
    const size_t qs = (size_t)NB_SW * (nlay + 1) * n;
    R *const part = A.part + (size_t)(IBM - 1) * (nlay + 1) * n;
#define PART(kind, lev, val) stg(part + (size_t)(kind) * qs + (size_t)(lev) * n, cb, (R)(val))
    R tdbt[NG], ztdn[NG], prdnd[NG], tdbtT[NG], ztdnT[NG], prdndT[NG];
    uint32_t dmask = 0;      // bit g: a cloudy cell has been met in sub-column g (total-sky downward state diverged from clear sky)
    // level nlay+1 of the API = TOA (spcvmc level 1): ptdbt = ztdn = 1, prdnd = 0
    {
        R cu = 0, cd = 0, fu = 0, fd = 0;
        R pu0[NG], pd0[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) { pu0[g] = PLD(5, CT4(nlay - 1, g)); pd0[g] = PLD(6, CT4(nlay - 1, g)); }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            tdbt[g] = 1; ztdn[g] = 1; prdnd[g] = 0; tdbtT[g] = 1; ztdnT[g] = 1; prdndT[g] = 0;
            const uint32_t u4 = CT4(nlay - 1, g);
            const R zi = zinc[g] * prmu0;
            const R pu = pu0[g], pd = pd0[g];
            {
                const R zr = f_rcp<R>((R)1. - prdnd[g] * pd);
                cu = cu + zi * ((tdbt[g] * pu + (ztdn[g] - tdbt[g]) * pd) * zr);
                cd = cd + zi * (tdbt[g] + (ztdn[g] - tdbt[g] + tdbt[g] * pu * prdnd[g]) * zr);
            }
            if (CLD && ccol) {
                R puT = pu, pdT = pd;
                if (lowc[g] <= nlay - 1) { puT = PLD(12, u4); pdT = PLD(13, u4); }
                const R zr = f_rcp<R>((R)1. - prdndT[g] * pdT);
                fu = fu + zi * ((tdbtT[g] * puT + (ztdnT[g] - tdbtT[g]) * pdT) * zr);
                fd = fd + zi * (tdbtT[g] + (ztdnT[g] - tdbtT[g] + tdbtT[g] * puT * prdndT[g]) * zr);
            }
        }
        PART(0, nlay, cu); PART(1, nlay, cd);
        if (CLD && ccol) { PART(2, nlay, fu); PART(3, nlay, fd); }
    }
    // The clear-sky values of a group of W cells (five layer properties, the two upward reflectances at the lower boundary) are
    // requested one group AHEAD of their use, so that a wave always has a group's loads in flight while it does another's arithmetic
    struct Park { R ref[W], refd[W], tra[W], trad[W], dbt[W], pu[W], pd[W]; };
    // (straight-line code: at the surface layer the reflectances of the layer below are requested from the layer itself and
    // replaced by the albedo on use; a branch around the loads made the compiler wait for all outstanding loads at the join)
    auto request = [&](int lay, int q, Park &b) {
        const int lu = lay > 0 ? lay - 1 : 0;
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int g = q * W + j;
            b.ref[j] = 0; b.refd[j] = 0; b.tra[j] = 0; b.trad[j] = 0; b.dbt[j] = 0; b.pu[j] = 0; b.pd[j] = 0;
            if (g >= NG) continue;
            const uint32_t c4 = CT4(lay, g), u4 = CT4(lu, g);
            b.ref[j] = PLD(0, c4); b.refd[j] = PLD(1, c4); b.tra[j] = PLD(2, c4); b.trad[j] = PLD(3, c4); b.dbt[j] = PLD(4, c4);
            b.pu[j] = PLD(5, u4); b.pd[j] = PLD(6, u4);
        }
    };
    Park nx;
    request(nlay - 1, 0, nx);
    R sdir = 0, sfd = 0, sfu = 0;
    for (int lay = nlay - 1; lay >= 0; lay--) {      // cross layer `lay`; its lower boundary is API level `lay`
        const int jk = nlay - 1 - lay;
        R cu = 0, cd = 0, fu = 0, fd = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            Park cur = nx;
            if (q + 1 < NQ) request(lay, q + 1, nx);
            else request(lay > 0 ? lay - 1 : 0, 0, nx);      // (after the surface layer: a harmless repeat)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < W; j++) { cur.pu[j] = lay > 0 ? cur.pu[j] : albp; cur.pd[j] = lay > 0 ? cur.pd[j] : albd; }
            // total sky: two wave-uniform tests (ballots) decide whether the group's cloudy-cell properties / diverged upward
            // reflectances are requested at all; lanes a test does not concern load along and select the clear-sky value afterwards
            R refT[W], refdT[W], traT[W], tradT[W], dbtT[W], puT[W], pdT[W], dbt[W];
            bool cm[W], dv[W];
            bool anycm = false, anydv = false;
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                cm[j] = false; dv[j] = false;
                dbt[j] = cur.dbt[j];
                if constexpr (CLD) {
                    if (g < NG) {
                        // the sign bit of the parked direct-beam transmittance (>= 0) carries "this cell is cloudy" from sweep A
                        cm[j] = ccol && __builtin_signbit(cur.dbt[j]);
                        dbt[j] = __builtin_signbit(cur.dbt[j]) ? -cur.dbt[j] : cur.dbt[j];
                        dv[j] = ccol && lay > 0 && lowc[g] <= lay - 1;
                        anycm = anycm || cm[j]; anydv = anydv || dv[j];
                    }
                }
                refT[j] = cur.ref[j]; refdT[j] = cur.refd[j]; traT[j] = cur.tra[j]; tradT[j] = cur.trad[j]; dbtT[j] = dbt[j];
                puT[j] = cur.pu[j]; pdT[j] = cur.pd[j];
            }
            if constexpr (CLD) {
                const bool wcm = __ballot(anycm) != 0, wdv = __ballot(anydv) != 0;
                if (wcm || wdv) {
#pragma unroll
                    for (int j = 0; j < W; j++) {
                        const int g = q * W + j;
                        if (g >= NG) continue;
                        const uint32_t c4 = CT4(lay, g);
                        if (wcm) {
                            const R a0 = PLD(7, c4), a1 = PLD(8, c4), a2 = PLD(9, c4), a3 = PLD(10, c4), a4 = PLD(11, c4);
                            if (cm[j]) { refT[j] = a0; refdT[j] = a1; traT[j] = a2; tradT[j] = a3; dbtT[j] = a4; }
                        }
                        if (wdv) {
                            const uint32_t u4 = CT4(lay > 0 ? lay - 1 : 0, g);
                            const R b0 = PLD(12, u4), b1 = PLD(13, u4);
                            if (dv[j]) { puT[j] = b0; pdT[j] = b1; }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
                const int g = q * W + j;
                if (g >= NG) continue;
                const R zi = zinc[g] * prmu0;
                // downward adding recurrences (:1530-1572): values at the lower boundary of this layer
                {
                    R zt, pr;
                    if (jk == 0) { zt = cur.tra[j]; pr = cur.refd[j]; }
                    else {
                        const R zreflect = f_rcp<R>((R)1. - cur.refd[j] * prdnd[g]);
                        zt = tdbt[g] * cur.tra[j] + (cur.trad[j] * ((ztdn[g] - tdbt[g]) + tdbt[g] * cur.ref[j] * prdnd[g])) * zreflect;
                        pr = cur.refd[j] + cur.trad[j] * cur.trad[j] * prdnd[g] * zreflect;
                    }
                    tdbt[g] = dbt[j] * tdbt[g]; ztdn[g] = zt; prdnd[g] = pr;
                }
                R u, d;
                {
                    const R zr = f_rcp<R>((R)1. - prdnd[g] * cur.pd[j]);
                    u = (tdbt[g] * cur.pu[j] + (ztdn[g] - tdbt[g]) * cur.pd[j]) * zr;
                    d = tdbt[g] + (ztdn[g] - tdbt[g] + tdbt[g] * cur.pu[j] * prdnd[g]) * zr;
                    cu = cu + zi * u; cd = cd + zi * d;
                }
                if constexpr (CLD) {
                    // Above the highest cloudy cell of a sub-column the total-sky downward state IS the clear-sky one
                    const bool divg = ccol && (((dmask >> g) & 1u) || cm[j]);
                    if (ccol && !divg) { tdbtT[g] = tdbt[g]; ztdnT[g] = ztdn[g]; prdndT[g] = prdnd[g]; }
                    if (divg) {
                        dmask |= 1u << g;
                        R zt, pr;
                        if (jk == 0) { zt = traT[j]; pr = refdT[j]; }
                        else {
                            const R zreflect = f_rcp<R>((R)1. - refdT[j] * prdndT[g]);
                            zt = tdbtT[g] * traT[j] + (tradT[j] * ((ztdnT[g] - tdbtT[g]) + tdbtT[g] * refT[j] * prdndT[g])) * zreflect;
                            pr = refdT[j] + tradT[j] * tradT[j] * prdndT[g] * zreflect;
                        }
                        tdbtT[g] = dbtT[j] * tdbtT[g]; ztdnT[g] = zt; prdndT[g] = pr;
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
            __builtin_amdgcn_sched_barrier(0);      // one group's arithmetic at a time
        }
        PART(0, lay, cu); PART(1, lay, cd);
        if (CLD && ccol) { PART(2, lay, fu); PART(3, lay, fd); }
    }
    stg(A.bsfc + (size_t)(0 * NB_SW + IBM - 1) * n, cb, sdir);
    stg(A.bsfc + (size_t)(1 * NB_SW + IBM - 1) * n, cb, sfd);
    stg(A.bsfc + (size_t)(2 * NB_SW + IBM - 1) * n, cb, sfu);
#undef PART
#undef PST
#undef PLD
#undef CT4
#undef CELL

    // ---- PAR in-cloud optical thickness diagnostics (SW/rrtmg_sw_spcvmc.F90:749-1109), bands 24-26 ---------------
    if constexpr (IBM >= 9 && IBM <= 11) {
        R d[4] = {0, 0, 0, 0}, nn[4] = {0, 0, 0, 0};
        if (CLD && ccol) {
            const R w0 = IBM == 9 ? (R)0.5 : (R)1.0;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const R wgt = w0 * zinc[g];
                const R sl = ldg(A.cotsum + (size_t)(0 * NG_SW + G0 + g) * n, cb), sm = ldg(A.cotsum + (size_t)(1 * NG_SW + G0 + g) * n, cb),
                        sh = ldg(A.cotsum + (size_t)(2 * NG_SW + G0 + g) * n, cb);
                if (sl > 0) { d[3] += wgt; nn[3] += wgt * sl; }
                if (sm > 0) { d[2] += wgt; nn[2] += wgt * sm; }
                if (sh > 0) { d[1] += wgt; nn[1] += wgt * sh; }
                const R st = sl + sm + sh;
                if (st > 0) { d[0] += wgt; nn[0] += wgt * st; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            stg(A.cot + (size_t)(k * 3 + (IBM - 9)) * n, cb, d[k]);
            stg(A.cot + (size_t)((4 + k) * 3 + (IBM - 9)) * n, cb, nn[k]);
        }
    }
}

__constant__ const int SW_BAND_ORDER[NB_SW] = {17, 29, 20, 21, 23, 18, 19, 24, 27, 16, 25, 26, 28, 22};
__host__ __device__ constexpr int sw_band_g0(int jb)
{
    constexpr int g0[14] = {0, 6, 18, 26, 34, 44, 54, 56, 66, 74, 80, 86, 94, 100};
    return g0[jb - 16];
}
__host__ __device__ constexpr int sw_band_ng(int jb)
{
    constexpr int ng[14] = {6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};
    return ng[jb - 16];
}

// Stage dump for the tests: the McICA cloud optics (delta-scaled tau, omega, g) of every (layer, g-point, column) cell in the
// reference's layout, Fortran (nlay,112,ncol), as cldprmc_sw leaves them (SW/rrtmg_sw_cldprmc.F90:311-392: 0, 1, 0 in clear cells).
// k_mcica<2> only writes the planes of layers with cloud fraction in cloudy columns; everything else is the default.
template <typename R>
__global__ void __launch_bounds__(256) k_sw_dump_cldprmc(SwArgs<R> A, R *__restrict__ taucmc, R *__restrict__ ssacmc, R *__restrict__ asmcmc)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x, lay = blockIdx.y;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, pc = A.perm[col];
    const bool written = col >= *A.nclear && A.cld[(size_t)lay * A.ld + pc] > 0;
    for (int jb = 16; jb <= 29; jb++) {
        const int g0 = sw_band_g0(jb), ng = sw_band_ng(jb);
        for (int g = 0; g < ng; g++) {
            const size_t w = (size_t)g0 * nlay * n + ((size_t)lay * ng + g) * n + col;
            const size_t o = ((size_t)pc * NG_SW + (g0 + g)) * nlay + lay;
            taucmc[o] = written ? A.taucmc[w] : (R)0;
            ssacmc[o] = written ? A.ssacmc[w] : (R)1;
            asmcmc[o] = written ? A.asmcmc[w] : (R)0;
        }
    }
}

template <typename R, bool CLD, bool DBG>
__global__ void __launch_bounds__(256, (sizeof(R) == 4 ? 2 : 1)) k_sw_bands(SwArgs<R> A, SwDev<R> T, SwSolar<R> SV)
{
    int bstart, bslot;       // one-dimensional grid: the bands of a column block run together on one XCD (lw_kernels.hpp band_block)
    if (!band_block(A.ncol, NB_SW, bstart, bslot)) return;
    const int nclear = *A.nclear;
    // every column runs the instantiation of its own class (the one mixed block is visited by both kernels, each
    // masking the other class's lanes): a column's arithmetic never depends on its neighbours -> bitwise column independence
    const int bend = bstart + (int)blockDim.x < A.ncol ? bstart + (int)blockDim.x : A.ncol;
    if (!DBG && (CLD ? bend <= nclear : bstart >= nclear)) return;
    const int col = bstart + threadIdx.x;
    if (col >= A.ncol) return;
    if (!DBG && (CLD ? col < nclear : col >= nclear)) return;
    switch (SW_BAND_ORDER[bslot]) {
        case 16: sw_band_body<R, SwB16, CLD, DBG>(A, T, SV, col, nclear); break;
        case 17: sw_band_body<R, SwB17, CLD, DBG>(A, T, SV, col, nclear); break;
        case 18: sw_band_body<R, SwB18, CLD, DBG>(A, T, SV, col, nclear); break;
        case 19: sw_band_body<R, SwB19, CLD, DBG>(A, T, SV, col, nclear); break;
        case 20: sw_band_body<R, SwB20, CLD, DBG>(A, T, SV, col, nclear); break;
        case 21: sw_band_body<R, SwB21, CLD, DBG>(A, T, SV, col, nclear); break;
        case 22: sw_band_body<R, SwB22, CLD, DBG>(A, T, SV, col, nclear); break;
        case 23: sw_band_body<R, SwB23, CLD, DBG>(A, T, SV, col, nclear); break;
        case 24: sw_band_body<R, SwB24, CLD, DBG>(A, T, SV, col, nclear); break;
        case 25: sw_band_body<R, SwB25, CLD, DBG>(A, T, SV, col, nclear); break;
        case 26: sw_band_body<R, SwB26, CLD, DBG>(A, T, SV, col, nclear); break;
        case 27: sw_band_body<R, SwB27, CLD, DBG>(A, T, SV, col, nclear); break;
        case 28: sw_band_body<R, SwB28, CLD, DBG>(A, T, SV, col, nclear); break;
        default: sw_band_body<R, SwB29, CLD, DBG>(A, T, SV, col, nclear); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_sw_reduce: one thread per (column, level) for the flux profiles, blockIdx.y = nlay + 1 for the per-column part
// (SW/rrtmg_sw_rad.F90:1515-1798): band sums in fixed order, surface broadband / band diagnostics (spcvmc :624-671),
// clear == total for cloud-free columns, normFlx.
// ---------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256) k_sw_reduce(SwArgs<R> A, SwOut<R> O)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ncol) return;
    const int n = A.ncol, nlay = A.nlay, ld = A.ld;
    const bool ccol = col >= *A.nclear;
    const int pc = A.perm[col];                 // outputs go to the original column
    const size_t qs = (size_t)NB_SW * (nlay + 1) * n;
    // TOA downward total-sky flux for the optional normalisation (:1769-1771)
    R top = 0;
    for (int b = 0; b < NB_SW; b++) top += A.part[(size_t)(ccol ? 3 : 1) * qs + ((size_t)b * (nlay + 1) + nlay) * n + col];
    R scale = 1;
    if (A.normFlx == 1) scale = top > (R)1e-7 ? top : (R)1e-7;
    if ((int)blockIdx.y <= nlay) {
        const int lev = blockIdx.y;
        R s[4] = {0, 0, 0, 0};
        for (int b = 0; b < NB_SW; b++) {
            const size_t o = ((size_t)b * (nlay + 1) + lev) * n + col;
            s[0] += A.part[o]; s[1] += A.part[qs + o];
            if (ccol) { s[2] += A.part[2 * qs + o]; s[3] += A.part[3 * qs + o]; }
        }
        if (!ccol) { s[2] = s[0]; s[3] = s[1]; }
        const size_t i = (size_t)lev * ld + pc;
        if (A.normFlx == 1) { O.swuflxc[i] = s[0] / scale; O.swdflxc[i] = s[1] / scale; O.swuflx[i] = s[2] / scale; O.swdflx[i] = s[3] / scale; }
        else { O.swuflxc[i] = s[0]; O.swdflxc[i] = s[1]; O.swuflx[i] = s[2]; O.swdflx[i] = s[3]; }
        return;
    }
    R znirr = 0, znirf = 0, zparr = 0, zparf = 0, zuvrr = 0, zuvrf = 0;
    for (int ibm = 1; ibm <= NB_SW; ibm++) {
        const R dir = A.bsfc[(size_t)(0 * NB_SW + ibm - 1) * n + col], fd = A.bsfc[(size_t)(1 * NB_SW + ibm - 1) * n + col],
                fu = A.bsfc[(size_t)(2 * NB_SW + ibm - 1) * n + col];
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
        R s = 0;
        if (ccol) for (int b = 0; b < 3; b++) s += A.cot[(size_t)(k * 3 + b) * n + col];
        O.cot[k][pc] = s;
    }
}

}  // namespace geosrad
