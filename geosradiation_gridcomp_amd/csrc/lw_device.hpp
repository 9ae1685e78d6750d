// lw_device.hpp -- device-side data structures of the RRTMG_LW / McICA hot path (gfx950).
//
// Mapping: lane = atmospheric column.  Every HBM-resident array keeps the
// reference's solver-API layout, Fortran (ncol,nlay) => [lay][col], so a wavefront reads/writes 64
// consecutive columns (256 B fp32 / 512 B fp64 per instruction).  k-distribution tables are stored
// row = (reference-pressure, temperature, species) index, columns = the band's g-points padded to a
// 16-byte multiple, so one lane fetches all g-points of a row with 16-byte loads and lanes of a wave
// (neighbouring columns, same layer) hit the same few rows in L1/L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <initializer_list>

namespace geosrad {

constexpr int NB_LW = 16;
constexpr int NG_LW = 140;
constexpr int NTBL = 10000;

template <typename R> struct Vec2;
template <> struct Vec2<float> { using T = float2; };
template <> struct Vec2<double> { using T = double2; };

// One spectral band's coefficient tables (device pointers).  Rows are NGP = pad4(ng) reals long.
template <typename R> struct BandTab {
    const R *absa, *absb;          // [65*nspa][NGP], [235*nspb][NGP]
    const R *fracrefa, *fracrefb;  // [9|1][NGP], [5|1][NGP]
    const R *selfref, *forref;     // [10][NGP], [4][NGP]
    const R *m[7];                 // band-specific minor-gas tables / per-g vectors (see lw_tables.cpp)
};

template <typename R> struct LwDev {
    BandTab<R> b[NB_LW + 1];                  // 1-based band index
    const R *totplnk, *totplnkderiv;          // Fortran (181,16)
    const R *preflog, *tref;                  // [59]
    const R *chi_mls;                         // Fortran (7,59)
    const R *rat;                             // [6 pairs][60]: chi(a,j)/chi(b,j), j = 1..59, pairs below
    const typename Vec2<R>::T *lut;           // [10001] (exp_tbl, tfn_tbl) interleaved
    const R *tau_tbl;                         // [10001]
    const R *absice0, *absice1, *absice2, *absice3, *absice4, *absliq1;  // Fortran layouts
    const R *xcw;                             // Fortran (1000,140) or nullptr (homogeneous condensate)
    int ice1b[16];
    R bpade, fluxfac, oneminus, grav, avogad;
    R delwave[NB_LW + 1];
    R aam[4], ram[4];                         // correlation-length parameters (cloud_subcol_gen.F90:100-107)
};

// chi_mls ratio pairs precomputed on the host in R precision (same IEEE division the reference does
// per layer in setcoef, rrtmg_lw_setcoef.F90:484-541)
enum RatPair { RAT_H2OCO2 = 0, RAT_H2OO3, RAT_H2ON2O, RAT_H2OCH4, RAT_N2OCO2, RAT_O3CO2, RAT_NPAIR };

// per-(layer,column) setcoef record, SoA: field f at sc + (f*nlay + lay)*ncol + col
enum ScField {
    SC_FAC00 = 0, SC_FAC01, SC_FAC10, SC_FAC11, SC_COLDRY, SC_FORFAC, SC_FORFRAC, SC_SELFFAC, SC_SELFFRAC,
    SC_MINORFRAC, SC_SCALEMINOR, SC_SCALEMINORN2, SC_COLBRD, SC_NFIELD
};

// packed integer indices of a layer: jp(6) jt(3) jt1(3) indfor(2) indself(4) indminor(5) lower(1)
__host__ __device__ inline uint32_t pack_idx(int jp, int jt, int jt1, int indfor, int indself, int indminor, int lower)
{
    return (uint32_t)jp | ((uint32_t)jt << 6) | ((uint32_t)jt1 << 9) | ((uint32_t)indfor << 12) |
           ((uint32_t)indself << 14) | ((uint32_t)indminor << 18) | ((uint32_t)lower << 23);
}

// error bits raised by the device-side input checks (reference: rrtmg_lw_rad.F90:209-318 etc.)
enum ErrBit {
    ERR_NEG_FIRST = 0,  // bits 0..22: negative values in input #k (order of LW_NEG_NAMES in geosrad.hip)
    ERR_PRESSURE_ORDER = 24,
    ERR_ICE_RADIUS_HI = 25, ERR_ICE_RADIUS_LO = 26, ERR_LIQ_RADIUS_HI = 27, ERR_LIQ_RADIUS_LO = 28
};

// Arguments of one RRTMG_LW batch (device pointers; API layouts, see include/geosrad.h)
template <typename R> struct LwArgs {
    int ncol;                    // columns in this batch (workspace leading dimension)
    int ld;                      // leading dimension (total ncol) of the API arrays; pointers pre-offset to the batch
    int nlay, dudTs, iceflg, liqflg, doy, cloudLM, cloudMH;
    const R *play, *plev, *tlay, *tlev, *tsfc, *emis;
    const R *h2o, *o3, *co2, *ch4, *n2o, *o2, *cfc11, *cfc12, *cfc22, *ccl4;
    const R *cldf, *ciwp, *clwp, *rei, *rel, *tauaer, *zm, *alat;
    // workspace
    R *sc;                       // [SC_NFIELD][nlay][ncol]
    uint32_t *scidx;             // [nlay][ncol]
    R *pwvcm;                    // [ncol]
    uint8_t *colcloudy;          // [ncol]  1 + the highest layer with cldf > 0 (0: none)
    int32_t *perm;               // [ncol]  compacted position -> column of the batch: clear columns first, then cloudy (stable)
    int32_t *nclear;             // [1]     number of clear columns of the batch
    uint8_t *laycloudy;          // [nlay][ncol]  optically cloudy for ANY g-point (cldprmc's `cloudy`)
    R *taucmc;                   // [140][nlay][ncol]
    R *alpha, *rcorr;            // [nlay][ncol] inter-layer overlap correlations (cloud_subcol_gen.F90:314-321)
    uint16_t *s1;                // [140][nlay][ncol] parked cells of the total-sky stream: the Pade index of the cell's discretised optical depth
    uint16_t *s2;                //                   s2: the clear-sky stream of cloudy columns
    R *part;                     // [6][16][nlay+1][ncol] per-band partial fluxes: dflx,dflxc,uflx,uflxc,duflx,duflxc
    uint32_t *err;               // error bit mask
    R *dbg_taug, *dbg_pfracs;    // optional Fortran (nlay,140,ncol) dumps (nullptr in production)
    int32_t *clearCounts;        // (ncol,4)
    uint32_t band_mask;          // bit ib set: k_lw_bands runs band ib (all 16 except in a RATS pass, see lw_rat_bands)
    uint32_t *pfcode;            // [16][nlay][ncol] split path (lw_split_kernels.hpp): Planck-fraction selector of the (band, layer, column):
    R *pffs;                     //                  kind | js << 8, and the interpolation weight fs
};

// RATS passes (GEOS_IrradGridComp.F90:3405-3468) remove one gas: only the bands whose optical depths read that gas's column amount
// - or, for the members of setcoef's `summol` (co2, o3, n2o, ch4: rrtmg_lw_setcoef.F90:416-418), the broadening-gas amount of the
// N2 continuum in bands 1 and 15 (rrtmg_lw_taumol.F90 taugb1, taugb15) - change; every other band's partial fluxes are those of
// the main call.  Bit ib = band ib (1..16).  Water vapour enters coldry, the continua and the diffusivity angle: all bands.
__host__ __device__ constexpr uint32_t lw_bandbits(std::initializer_list<int> l)
{
    uint32_t m = 0;
    for (int b : l) m |= 1u << b;
    return m;
}
constexpr uint32_t LW_ALL_BANDS = 0x1FFFEu;
constexpr uint32_t LW_RAT_BANDS[8] = {
    LW_ALL_BANDS,                                         // H2O
    lw_bandbits({4, 5, 7, 8, 13, 1, 15}),                 // O3
    lw_bandbits({3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 1}),   // CO2
    lw_bandbits({9, 16, 1, 15}),                          // CH4
    lw_bandbits({3, 8, 9, 13, 15, 1}),                    // N2O
    lw_bandbits({6}),                                     // CFC11
    lw_bandbits({6, 8}),                                  // CFC12
    lw_bandbits({8}),                                     // HCFC22
};

}  // namespace geosrad
