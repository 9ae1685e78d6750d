// chou_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for the Chou-Suarez longwave scheme `irrad`.
//
// Reference behaviour: GEOSirrad_GridComp/irrad.F90:27-1338 (driver + band loop), helpers :1341-2780 (planck, *exps, tablup,
// *kdis, cldovlp, sfcflux, mkicx/SORTIT); cloud optics GEOS_RadiationShared/getirtau.code:1-102.  Non-OVERCAST build.
//
// Unlike RRTMG (independent g-points, first-order vertical recurrences -> lane = column), irrad couples every pair of levels
// (k1, k2): O(np^2) transmittance products per band.  Mapping here: ONE WAVEFRONT PER (column, band).
//   * the band's per-layer exponentials, Planck terms, cloud / aerosol transmittances live in LDS (~14 KB fp32 per wave);
//   * lane = k1: each lane carries the running k-distribution products from its level k1 downwards in registers and walks
//     k2 = k1+1 .. np+1; all lanes advance in lock-step (uniform trip count, barrier per step), so at any step the lanes touch
//     DISTINCT k2 and the k2-indexed (downward) flux accumulators can be plain LDS read-modify-writes in a fixed order
//     (bitwise reproducible); the k1-indexed (upward) accumulators stay in registers;
//   * per-band partial fluxes go to HBM [column][band][kind][level] (coalesced over lanes = levels) and a second kernel sums
//     the 10 bands in band order.
// A lane = column preparation kernel turns the API arrays ([level][column], coalesced over columns) into per-column records
// [column][field][level] so that the wave-per-column kernel reads them coalesced over levels.
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {

constexpr int CH_NB = 10;
constexpr int CH_NX = 26, CH_NO = 21, CH_NC = 30, CH_NH = 31;
enum ChField { CF_PA = 0, CF_DT, CF_DH2O, CF_DCONT, CF_DCO2, CF_DO3, CF_DN2O, CF_DCH4, CF_DF11, CF_DF12, CF_DF22, CF_TA, CF_DPPA, CF_FCLD,
               CF_REFF1, CF_REFF2, CF_REFF3, CF_REFF4, CF_CWC1, CF_CWC2, CF_CWC3, CF_CWC4, CF_NFIELD };
constexpr int CH_NKIND = 10;     // flxu flcu flau flxau flxd flcd flad flxad dfdts + 1 slot (sfcem in [0])

template <typename R> struct ChouDev {
    R xkw[9], xke[9], aw[9], bw[9], pm[9];
    int mw[9];
    R fkw[54], gkw[18], cb[60], dcb[50];
    R w11, w12, w13, p11, p12, p13, dwe, dpe;
    const R *c1, *c2, *c3, *oo1, *oo2, *oo3, *h11, *h12, *h13, *h21, *h22, *h23, *h81, *h82, *h83;   // Fortran (26, n)
    R aib[30], awb[40], aiw[40], aww[40], aig[40], awg[40];
};

template <typename R> struct ChouArgs {
    int m, ld, np, trace, ict, icb, ns, na, nb;
    R co2;
    const R *ple, *ta, *wa, *oa, *tb, *n2o, *ch4, *cfc11, *cfc12, *cfc22, *cwc, *fcld, *reff, *fs, *tg, *eg, *tv, *ev, *rv;
    R *taua, *ssaa, *asya;                 // INOUT, rescaled in place like the reference (irrad.F90:655-678)
    R *taudiag;                            // (m, np, 10)
    // workspace
    R *rec;                                // [m][CF_NFIELD][K1]  per-column layer records, K1 = np + 1 (layer 0 = above the model top)
    R *part;                               // [m][10 bands][CH_NKIND][K2]  K2 = np + 2
    uint32_t *err;
};
template <typename R> struct ChouOut { R *flxu, *flcu, *flau, *flxau, *flxd, *flcd, *flad, *flxad, *dfdts, *sfcem; };

template <typename R> GR_DEV R gr_log10(R x);
template <> GR_DEV float gr_log10<float>(float x) { return log10f(x); }
template <> GR_DEV double gr_log10<double>(double x) { return log10(x); }

// ---------------------------------------------------------------------------------------------------
// k_chou_prep: lane = column; absorber amounts and scaled quantities of every layer (irrad.F90:381-453), written as the
// column's record.  (Like the reference, irrad has no input assertions.)
// ---------------------------------------------------------------------------------------------------
// The records are column-major ([column][field][level]: k_chou_bands reads a column's fields with lanes = levels), the inputs
// are level-major with the column index fastest: the transposition goes through an LDS tile of 64 columns x 8 (fp64: 4) levels, so that
// the inputs are read coalesced (lanes = columns) and the records are written in 32-byte runs (lanes = levels x fields).
template <typename R>
__global__ void __launch_bounds__(256) k_chou_prep(ChouArgs<R> A)
{
    constexpr int CHP_KC = 32 / (int)sizeof(R);      // levels per tile: one 32-byte run per (column, field)
    __shared__ R tile[CF_NFIELD * CHP_KC * 65];
    // 256 threads: the four wavefronts share the tile (46 KB: three blocks = twelve wavefronts per CU instead of three), wavefront w forms
    // the levels w, w + 4, .. of the tile for the block's 64 columns
    const int lane = threadIdx.x % 64, wv = threadIdx.x / 64, col0 = blockIdx.x * 64, i = col0 + lane;
    const bool act = i < A.m;
    const int np = A.np, ld = A.ld, K1 = np + 1;
    const int ncolb = (A.m - col0) < 64 ? (A.m - col0) : 64;
#define AP(a, k) a[(size_t)((k) - 1) * ld + i]
#define TL(f, kk) tile[((f) * CHP_KC + (kk)) * 65 + lane]
    {   // blockIdx.y = the tile of CHP_KC levels (chou_prep_tiles(np) of them): a wavefront per (64 columns, tile) instead of one per 64
        // columns walking the tiles - 10 x the wavefronts of what is a streaming kernel
        const int k0 = (int)blockIdx.y * CHP_KC;
        const int nk = (np + 1 - k0) < CHP_KC ? (np + 1 - k0) : CHP_KC;
        if (act) {
            for (int kk = wv; kk < nk; kk += 4) {
                const int k = k0 + kk;
                const int ks = k == 0 ? 1 : k;      // layer 0 copies the top layer's state (:432-453)
                R dp, pa;
                if (k == 0) { dp = AP(A.ple, 1) * (R)0.01; dp = dp > (R)0.005 ? dp : (R)0.005; pa = (R)0.5 * dp; }
                else { pa = (R)0.5 * (AP(A.ple, k + 1) + AP(A.ple, k)) * (R)0.01; dp = (AP(A.ple, k + 1) - AP(A.ple, k)) * (R)0.01; }
                const R ta = AP(A.ta, ks), wa = AP(A.wa, ks);
                R dh2o = (R)1.02 * wa * dp, do3 = (R)476. * AP(A.oa, ks) * dp, dco2 = (R)789. * A.co2 * dp;
                dh2o = dh2o > (R)1.e-10 ? dh2o : (R)1.e-10;
                do3 = do3 > (R)1.e-6 ? do3 : (R)1.e-6;
                dco2 = dco2 > (R)1.e-4 ? dco2 : (R)1.e-4;
                const R xx = pa * (R)0.001618 * wa * wa * dp;
                TL(CF_PA, kk) = pa; TL(CF_DT, kk) = ta - (R)250.0; TL(CF_DH2O, kk) = dh2o;
                TL(CF_DCONT, kk) = xx * gr_exp<R>((R)1800. / ta - (R)6.081);
                TL(CF_DCO2, kk) = dco2; TL(CF_DO3, kk) = do3;
                TL(CF_DN2O, kk) = (R)789. * AP(A.n2o, ks) * dp; TL(CF_DCH4, kk) = (R)789. * AP(A.ch4, ks) * dp;
                TL(CF_DF11, kk) = (R)789. * AP(A.cfc11, ks) * dp; TL(CF_DF12, kk) = (R)789. * AP(A.cfc12, ks) * dp;
                TL(CF_DF22, kk) = (R)789. * AP(A.cfc22, ks) * dp;
                TL(CF_TA, kk) = ta;
                TL(CF_DPPA, kk) = k == 0 ? (R)0 : AP(A.ple, k + 1) - AP(A.ple, k);
                TL(CF_FCLD, kk) = k == 0 ? (R)0 : AP(A.fcld, k);
                for (int l = 0; l < 4; l++) {
                    TL(CF_REFF1 + l, kk) = k == 0 ? (R)0 : A.reff[((size_t)l * np + (k - 1)) * ld + i];
                    TL(CF_CWC1 + l, kk) = k == 0 ? (R)0 : A.cwc[((size_t)l * np + (k - 1)) * ld + i];
                }
            }
        }
        __syncthreads();
        // write-out: lane -> (field group, level in tile); 8 consecutive lanes write 8 consecutive levels of one field
        const int kk = (int)threadIdx.x % CHP_KC, fg = (int)threadIdx.x / CHP_KC;
        if (kk < nk) {
            for (int c = 0; c < ncolb; c++) {
                R *rec = A.rec + (size_t)(col0 + c) * CF_NFIELD * K1 + k0 + kk;
                for (int f = fg; f < CF_NFIELD; f += 256 / CHP_KC) rec[(size_t)f * K1] = tile[(f * CHP_KC + kk) * 65 + c];
            }
        }
    }
#undef AP
#undef TL
}
template <typename R> constexpr int chou_prep_tiles(int np) { return (np + 1 + 32 / (int)sizeof(R) - 1) / (32 / (int)sizeof(R)); }

// planck / plancd (:1341-1376)
template <typename R> GR_DEV R ch_planck(const ChouDev<R> &T, int ibn, R t)
{
    const R *c = T.cb + 6 * (ibn - 1);
    return t * (t * (t * (t * (t * c[5] + c[4]) + c[3]) + c[2]) + c[1]) + c[0];
}
template <typename R> GR_DEV R ch_plancd(const ChouDev<R> &T, int ibn, R t)
{
    const R *d = T.dcb + 5 * (ibn - 1);
    return t * (t * (t * (t * d[4] + d[3]) + d[2]) + d[1]) + d[0];
}

// the table coordinates of tablup in the fp32 instantiation: hardware reciprocal and log2 (1 ulp) instead of the correctly rounded division and
// the library log10 (~30 instructions per step of a table band); the interpolation is continuous across the cell boundaries they may move
template <typename R> GR_DEV R ch_rcp(R x) { return (R)1.0 / x; }
template <typename R> GR_DEV R ch_log10(R x) { return gr_log10<R>(x); }
#ifndef CH_EXACT_COORD
template <> GR_DEV float ch_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> GR_DEV float ch_log10<float>(float x) { return __builtin_amdgcn_logf(x) * 0.30102999566398120f; }
#endif

// a load through a pointer that was itself read from memory (the table pointers of *Tp): told to be global, or it becomes a flat load
// with a 64-bit per-lane address
template <typename T> GR_DEV T ch_ldg(const T *base, uint32_t byteoff)
{
    typedef const char __attribute__((address_space(1))) *gchar;
    typedef const T __attribute__((address_space(1))) *gT;
    return *(gT)((gchar)base + byteoff);
}

// exponentials / powers of the per-layer absorber terms (P0, P1 of the band body) in the fp32 instantiation: v_exp_f32 / v_log_f32 (1 ulp on
// the base-2 functions; the argument scaling adds |x| 2^-24 relative) instead of the library's expf / powf (15-60 instructions each)
template <typename R> GR_DEV R ch_exp(R x) { return gr_exp<R>(x); }
template <typename R> GR_DEV R ch_pow(R x, R y) { return gr_pow<R>(x, y); }
#ifndef CH_EXACT_EXP
template <> GR_DEV float ch_exp<float>(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
template <> GR_DEV float ch_pow<float>(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#endif

// tablup (:1887-2011).  The 14 table entries around the cell (ip, iw) stay in registers (`tv`) and are gathered again only when the lane's
// cell changes: along a row the accumulated amount s1 and its mean pressure move slowly on the tables' logarithmic axes, and 14 gathers of
// 64 different addresses per step were what made the five table bands 1.7 x as slow as the k-distribution ones (profiles/r04_README.md)
template <typename R>
GR_DEV void ch_tablup(int nh, R dw, R p, R dt, R &s1, R &s2, R &s3, R w1, R p1, R dwe, R dpe, const R *__restrict__ c1,
                      const R *__restrict__ c2, const R *__restrict__ c3, R &tran, int &cell, R (&tv)[14])
{
    constexpr int nx = CH_NX;
    s1 = s1 + dw; s2 = s2 + p * dw; s3 = s3 + dt * dw;
    const R x1 = s1, x1c = ch_rcp<R>(s1), x2 = s2 * x1c, x3 = s3 * x1c;
    R we = (ch_log10<R>(x1) - w1) * dwe, pe = (ch_log10<R>(x2) - p1) * dpe;
    we = we < (R)(nh - 1) ? we : (R)(nh - 1);
    pe = pe < (R)(nx - 1) ? pe : (R)(nx - 1);
    int iw = (int)(we + (R)1.0); iw = iw < nh - 1 ? iw : nh - 1; iw = iw > 2 ? iw : 2;
    const R fw = we - (R)(iw - 1);
    int ip = (int)(pe + (R)1.0); ip = ip < nx - 1 ? ip : nx - 1; ip = ip > 1 ? ip : 1;
    const R fp = pe - (R)(ip - 1);
    const int cnow = iw * nx + ip;
    if (cnow != cell) {
        cell = cnow;
        // entry (a, b) of a Fortran (nx, n) table sits at (b - 1) nx + (a - 1): wave-uniform base + one per-lane byte offset + immediates
        const uint32_t o = (uint32_t)(((iw - 2) * nx + (ip - 1)) * (int)sizeof(R)), r1 = nx * sizeof(R), r2 = 2 * nx * sizeof(R), e = sizeof(R);
        tv[0] = ch_ldg(c1, o); tv[1] = ch_ldg(c1, o + e); tv[2] = ch_ldg(c1, o + r1); tv[3] = ch_ldg(c1, o + r1 + e);
        tv[4] = ch_ldg(c1, o + r2); tv[5] = ch_ldg(c1, o + r2 + e);
        tv[6] = ch_ldg(c2, o + r1); tv[7] = ch_ldg(c2, o + r1 + e); tv[8] = ch_ldg(c2, o + r2); tv[9] = ch_ldg(c2, o + r2 + e);
        tv[10] = ch_ldg(c3, o + r1); tv[11] = ch_ldg(c3, o + r1 + e); tv[12] = ch_ldg(c3, o + r2); tv[13] = ch_ldg(c3, o + r2 + e);
    }
    const R pa = tv[0] + (tv[1] - tv[0]) * fp;
    const R pb = tv[2] + (tv[3] - tv[2]) * fp;
    const R pc = tv[4] + (tv[5] - tv[4]) * fp;
    const R ax = ((pc + pa) * fw + (pc - pa)) * fw * (R)0.5 + pb * ((R)1. - fw * fw);
    const R ba = tv[6] + (tv[7] - tv[6]) * fp;
    const R bb = tv[8] + (tv[9] - tv[8]) * fp;
    const R t1 = ba + (bb - ba) * fw;
    const R ca = tv[10] + (tv[11] - tv[10]) * fp;
    const R cb = tv[12] + (tv[13] - tv[12]) * fp;
    const R t2 = ca + (cb - ca) * fw;
    R xx = ax + (t1 + t2 * x3) * x3;
    xx = xx < (R)0.9999999 ? xx : (R)0.9999999;
    xx = xx > (R)0.0000001 ? xx : (R)0.0000001;
    tran = tran * xx;
}

// effective Planck functions of a layer with transmittance tr (:898-905 pattern)
template <typename R> GR_DEV void ch_emis(R tr, R bl0, R bl1, R &dn, R &up)
{
    R yy = tr < (R)0.9999 ? tr : (R)0.9999;
    yy = yy > (R)0.00001 ? yy : (R)0.00001;
    const R xx = (bl0 - bl1) / gr_log<R>(yy);
    dn = (bl1 - bl0 * yy) / ((R)1.0 - yy) - xx;
    up = (bl0 + bl1) - dn;
}

// per-band packing of the exponential tables (:501-566) and band switches
struct ChBand {
    int h2o_s, con_s, co2_s, n2o_s, ch4_s, com_s, f11_s, f12_s, f22_s, ne;
    bool h2otable, conbnd, co2bnd, oznbnd, n2obnd, combnd, f11bnd, f12bnd, b10bnd;
};
__host__ __device__ constexpr ChBand ch_band(int ibn)
{
    ChBand b{};
    b.h2otable = ibn == 1 || ibn == 2 || ibn == 8; b.conbnd = ibn >= 2 && ibn <= 7; b.co2bnd = ibn == 3; b.oznbnd = ibn == 5;
    b.n2obnd = ibn == 6 || ibn == 7; b.combnd = ibn == 4 || ibn == 5; b.f11bnd = b.combnd; b.f12bnd = ibn == 4 || ibn == 6;
    b.b10bnd = ibn == 10;
    switch (ibn) {
        case 2: b.con_s = 1; break;
        case 3: b.h2o_s = 1; b.con_s = 7; break;
        case 4: b.h2o_s = 1; b.con_s = 7; b.com_s = 8; b.f11_s = 14; b.f12_s = 15; b.f22_s = 16; break;
        case 5: b.h2o_s = 1; b.con_s = 7; b.com_s = 8; b.f11_s = 14; break;
        case 6: b.h2o_s = 1; b.con_s = 7; b.n2o_s = 8; b.ch4_s = 12; b.f12_s = 16; b.f22_s = 17; break;
        case 7: b.h2o_s = 1; b.con_s = 7; b.n2o_s = 8; b.ch4_s = 12; break;
        case 9: b.h2o_s = 1; break;
        case 10: b.h2o_s = 1; b.con_s = 6; b.co2_s = 7; b.n2o_s = 13; break;
        default: break;
    }
    b.ne = b.conbnd ? (ibn == 3 ? 3 : 1) : 0;
    return b;
}

// LDS planes of K1 reals a band needs: its exponentials (highest slot used) + the absorber paths of its table look-ups
__host__ __device__ constexpr int ch_nex(int ibn)
{
    constexpr int nex[11] = {0, 0, 1, 9, 16, 14, 17, 15, 0, 6, 14};
    return nex[ibn];
}
__host__ __device__ constexpr int ch_planes(int ibn)
{
    const ChBand b = ch_band(ibn);
    const int ntab = (b.h2otable ? 1 : 0) + (b.co2bnd ? 1 : 0) + (b.oznbnd ? 1 : 0);
    return ch_nex(ibn) + (ntab ? 2 + ntab : 0);
}
__host__ __device__ constexpr int ch_planes_max()
{
    int m = 0;
    for (int ibn = 1; ibn <= 10; ibn++) m = ch_planes(ibn) > m ? ch_planes(ibn) : m;
    return m;
}
// columns per wavefront of k_chou_bands and lanes per column (see chou_band_body).  1: the whole wavefront walks one column's rows.
// 2 (two half-wavefronts of 32 lanes, each with LDS arrays of its own) cuts loop 2000 from 82 to 61.5 lock-step walks per column but
// doubles the LDS of a wavefront, i.e. halves the wavefronts per CU - and the kernel lives on those (latency of the LDS read-modify-
// writes and table gathers of a step): 42.7 against 32.1 ms per 100 000 columns (profiles/r03_chou_rows.md), so 1 stays
#ifndef CH_CPW
#define CH_CPW 1
#endif
constexpr int CH_LPC = 64 / CH_CPW;
// LDS bytes of one column of k_chou_bands for np layers (16-byte multiple), and of a block (CH_CPW columns)
// LDS of one column: ONE array of np + 2 level records, the band's fields interleaved (bu bd cu cd au ad du dd | flxd flcd flad flxad |
// enn tcld taer | icx | the band's exponentials and table paths).  A lane's accesses of a step are then one address register (its
// level) + immediate offsets, instead of one address addition per array; the record length is odd, so the 64 lanes of a step (consecutive
// levels) fall on different banks.
constexpr int CH_F_FLUX = 8, CH_F_ENN = 12, CH_F_TCLD = 13, CH_F_TAER = 14, CH_F_ICX = 15, CH_F_EX = 16;
__host__ __device__ constexpr int ch_nlf(int ibn) { return (CH_F_EX + ch_planes(ibn)) | 1; }
__host__ __device__ constexpr int ch_nlf_max() { return (CH_F_EX + ch_planes_max()) | 1; }
template <typename R> constexpr size_t ch_lds_bytes_col(int np)
{
    return ((size_t)(np + 2) * ch_nlf_max() * sizeof(R) + 15) & ~(size_t)15;
}
template <typename R> constexpr size_t chou_bands_lds_bytes(int np) { return CH_CPW * ch_lds_bytes_col<R>(np); }

// field F of the level records (see ch_lds_bytes_col): a[k] is record k's field
template <typename T, int NLF, int F> struct ChFld {
    T *b;
    GR_DEV T &operator[](int k) const { return b[k * NLF + F]; }
};

// running transmittance state of one lane (one k1)
template <typename R> struct ChState {
    R th2o[6], tcon[3], tco2[6], tn2o[4], tch4[4], tcom[6], tf11, tf12, tf22, x1, x2, x3;
    R tv[14]; int cell;          // ch_tablup: the table entries of the lane's current cell (a band has at most one table absorber)
};

// ---------------------------------------------------------------------------------------------------
// k_chou_bands: one wavefront per (CH_CPW columns, band); blockIdx.x = column group, blockIdx.y = band - 1.  Dynamic LDS.
// The band is a template parameter of the body: which absorbers a band has, where their exponentials sit and how many running
// products a lane carries are then compile-time facts (dead branches and their registers disappear).
// ---------------------------------------------------------------------------------------------------
template <typename R, int IBN>
GR_DEV void chou_band_body(const ChouArgs<R> &A, const ChouDev<R> &T, unsigned char *ch_smem)
{
    // CH_CPW columns per wavefront, CH_LPC = 64 / CH_CPW lanes each: loop 2000 walks one row k1 per lane in lock-step, rows of 73 .. 1
    // steps at 72 layers; with 64 lanes per column the second pass (rows 64 .. 72) and the idle tails leave 49 % of the lane-steps
    // unused (82 steps per column), with 32 lanes per column the passes of two columns are 73 + 41 + 9 = 123 steps (61.5 per column).
    // Each half-wavefront owns its own LDS arrays; the lanes of a half never touch the other half's.
    const int half = (int)threadIdx.x / CH_LPC, lane = (int)threadIdx.x % CH_LPC;
    const int icol = (int)blockIdx.x * CH_CPW + half;
    const bool valid = icol < A.m;                            // (odd column count: the last wavefront's second half computes along on
    const int i = valid ? icol : A.m - 1;                     //  the last column and writes nothing)
    const unsigned long long hmask = CH_LPC == 64 ? ~0ull : (((1ull << (CH_LPC & 63)) - 1ull) << (half * CH_LPC));
    constexpr int ibn = IBN;
    const int np = A.np, K1 = np + 1, K2 = np + 2, ld = A.ld;
    constexpr ChBand B = ch_band(IBN);
    const bool trace = A.trace != 0, do_aer = A.na > 0;

    // ---- LDS carve-up (ch_lds_bytes_col) ------------------------------------------------------------------
    ch_smem += (size_t)half * ch_lds_bytes_col<R>(np);
    R *const lv = reinterpret_cast<R *>(ch_smem);
    constexpr int NLF = ch_nlf(IBN);
    constexpr bool TAB = B.h2otable || B.co2bnd || B.oznbnd;
    constexpr int F_PA = CH_F_EX + ch_nex(IBN), F_DT = F_PA + 1, F_DW = F_PA + 2, F_DCO2 = F_DW + (B.h2otable ? 1 : 0), F_DO3 = F_DCO2 + (B.co2bnd ? 1 : 0);
    const ChFld<R, NLF, 0> bu{lv}; const ChFld<R, NLF, 1> bd{lv}; const ChFld<R, NLF, 2> cu{lv}; const ChFld<R, NLF, 3> cd{lv};
    const ChFld<R, NLF, 4> au{lv}; const ChFld<R, NLF, 5> ad{lv}; const ChFld<R, NLF, 6> du{lv}; const ChFld<R, NLF, 7> dd{lv};
    const ChFld<R, NLF, CH_F_FLUX> flxd{lv}; const ChFld<R, NLF, CH_F_FLUX + 1> flcd{lv}; const ChFld<R, NLF, CH_F_FLUX + 2> flad{lv};
    const ChFld<R, NLF, CH_F_FLUX + 3> flxad{lv};
    const ChFld<R, NLF, CH_F_ENN> enn{lv}; const ChFld<R, NLF, CH_F_TCLD> tcld{lv}; const ChFld<R, NLF, CH_F_TAER> taer{lv};
    const ChFld<int, NLF * (int)(sizeof(R) / sizeof(int)), CH_F_ICX * (int)(sizeof(R) / sizeof(int))> icx{reinterpret_cast<int *>(lv)};
    const ChFld<R, NLF, F_PA> s_pa{lv}; const ChFld<R, NLF, F_DT> s_dt{lv}; const ChFld<R, NLF, F_DW> s_dw{lv};
    const ChFld<R, NLF, F_DCO2> s_dco2{lv}; const ChFld<R, NLF, F_DO3> s_do3{lv};
    // the Planck fluxes of layers and levels are dead once loop 1500 has formed the layer emissions: they share the downward-flux
    // accumulators' fields (which are zeroed after it).  The per-k1 results of loop 2000 (4 upward fluxes, 4 transmittances,
    // dfdts) are parked in the band's `part` slots in HBM by the lane that reads them back in P5.
    const ChFld<R, NLF, CH_F_FLUX> blayer{lv}; const ChFld<R, NLF, CH_F_FLUX + 1> blevel{lv};
#define EX(k, j) lv[(k) * NLF + (CH_F_EX - 1) + (j)]

    const R *rec = A.rec + (size_t)i * CF_NFIELD * K1;

    // ---- P0: per-layer quantities of this band (lanes = layers) -------------------------------------------------
    for (int k = lane; k <= np; k += CH_LPC) {
        const R pa = rec[CF_PA * K1 + k], dt = rec[CF_DT * K1 + k], dh2o = rec[CF_DH2O * K1 + k], dcont = rec[CF_DCONT * K1 + k],
                dco2 = rec[CF_DCO2 * K1 + k], do3 = rec[CF_DO3 * K1 + k];
        if constexpr (TAB) { s_pa[k] = pa; s_dt[k] = dt; }
        if constexpr (B.h2otable) s_dw[k] = dh2o;
        if constexpr (B.co2bnd) s_dco2[k] = dco2;
        if constexpr (B.oznbnd) s_do3[k] = do3;
        for (int j = 1; j <= ch_nex(IBN); j++) EX(k, j) = 0;
        // water vapour line exponentials (h2oexps :1379-1458)
        if (!B.h2otable && !B.b10bnd) {
            R xh = dh2o * ch_pow<R>(pa / (R)500., T.pm[ibn - 1]) * ((R)1. + (T.aw[ibn - 1] + T.bw[ibn - 1] * dt) * dt);
            R e = ch_exp<R>(-xh * T.xkw[ibn - 1]);
            EX(k, B.h2o_s) = e;
            const int mwv = T.mw[ibn - 1];
            for (int ik = 2; ik <= 6; ik++) {
                if (mwv == 6) { xh = e * e; e = xh * xh * xh; }
                else if (mwv == 8) { xh = e * e; xh = xh * xh; e = xh * xh; }
                else if (mwv == 9) { xh = e * e * e; e = xh * xh * xh; }
                else { xh = e * e; xh = xh * xh; xh = xh * xh; e = xh * xh; }
                EX(k, B.h2o_s + ik - 1) = e;
            }
        }
        if (B.conbnd) {       // conexps :1464-1510
            const R e = ch_exp<R>(-dcont * T.xke[ibn - 1]);
            EX(k, B.con_s) = e;
            if (ibn == 3) { const R e2 = e * e; EX(k, B.con_s + 1) = e2; EX(k, B.con_s + 2) = e2 * e2; }
        }
        if (trace) {
            if (B.n2obnd) {   // n2oexps :1515-1582
                const R dn2o = rec[CF_DN2O * K1 + k];
                if (ibn == 6) {
                    R xc = dn2o * ((R)1. + ((R)1.9297e-3 + (R)4.3750e-6 * dt) * dt);
                    const R e = ch_exp<R>(-xc * (R)6.31582e-2);
                    EX(k, B.n2o_s) = e;
                    xc = e * e * e;
                    const R xc1 = xc * xc, xc2 = xc1 * xc1;
                    EX(k, B.n2o_s + 1) = xc * xc1 * xc2;
                } else {
                    R xc = dn2o * ch_pow<R>(pa / (R)500.0, (R)0.48) * ((R)1. + ((R)1.3804e-3 + (R)7.4838e-6 * dt) * dt);
                    R e = ch_exp<R>(-xc * (R)5.35779e-2);
                    EX(k, B.n2o_s) = e;
                    for (int q = 1; q <= 3; q++) { xc = e * e; xc = xc * xc; e = xc * xc; EX(k, B.n2o_s + q) = e; }
                }
                // ch4exps :1587-1651
                const R dch4 = rec[CF_DCH4 * K1 + k];
                if (ibn == 6) {
                    const R xc = dch4 * ((R)1. + ((R)1.7007e-2 + (R)1.5826e-4 * dt) * dt);
                    EX(k, B.ch4_s) = ch_exp<R>(-xc * (R)5.80708e-3);
                } else {
                    R xc = dch4 * ch_pow<R>(pa / (R)500.0, (R)0.65) * ((R)1. + ((R)5.9590e-4 - (R)2.2931e-6 * dt) * dt);
                    R e = ch_exp<R>(-xc * (R)6.29247e-2);
                    EX(k, B.ch4_s) = e;
                    for (int q = 1; q <= 3; q++) { xc = e * e * e; xc = xc * xc; e = xc * xc; EX(k, B.ch4_s + q) = e; }
                }
            }
            if (B.combnd) {   // comexps :1656-1707
                R xc = ibn == 4 ? dco2 * ((R)1. + ((R)3.5775e-2 + (R)4.0447e-4 * dt) * dt)
                                : dco2 * ((R)1. + ((R)3.4268e-2 + (R)3.7401e-4 * dt) * dt);
                R e = ch_exp<R>(-xc * (R)1.922e-7);
                EX(k, B.com_s) = e;
                for (int ik = 2; ik <= 6; ik++) { xc = e * e; xc = xc * xc; e = xc * e; EX(k, B.com_s + ik - 1) = e; }
            }
            // CFCs, Table 7 (:723-766, cfcexps :1712-1764): band 4 uses the first coefficient set, the other band the second
            if (B.f11bnd) {
                const R d = rec[CF_DF11 * K1 + k];
                const R xf = ibn == 4 ? d * ((R)1. + ((R)1.26610e-3 + (R)3.55940e-6 * dt) * dt) : d * ((R)1. + ((R)8.19370e-4 + (R)4.67810e-6 * dt) * dt);
                EX(k, B.f11_s) = ch_exp<R>(-xf * (ibn == 4 ? (R)1.89736e+1 : (R)1.01487e+1));
            }
            if (B.f12bnd) {
                const R d = rec[CF_DF12 * K1 + k];
                const R xf = ibn == 4 ? d * ((R)1. + ((R)8.77370e-4 + (R)-5.88440e-6 * dt) * dt) : d * ((R)1. + ((R)8.62000e-4 + (R)-4.22500e-6 * dt) * dt);
                EX(k, B.f12_s) = ch_exp<R>(-xf * (ibn == 4 ? (R)1.58104e+1 : (R)3.70107e+1));
                const R d2 = rec[CF_DF22 * K1 + k];
                const R xg = ibn == 4 ? d2 * ((R)1. + ((R)9.65130e-4 + (R)1.31280e-5 * dt) * dt) : d2 * ((R)1. + ((R)-3.00010e-5 + (R)5.25010e-7 * dt) * dt);
                EX(k, B.f22_s) = ch_exp<R>(-xg * (ibn == 4 ? (R)6.18536e+0 : (R)3.27912e+1));
            }
            if (B.b10bnd) {   // b10exps :1769-1884
                R xx = dh2o * (pa / (R)500.0) * ((R)1. + ((R)0.0149 + (R)6.20e-5 * dt) * dt);
                R e = ch_exp<R>(-xx * (R)0.10624);
                EX(k, B.h2o_s) = e;
                for (int q = 1; q <= 4; q++) { xx = e * e; xx = xx * xx; e = xx * xx; EX(k, B.h2o_s + q) = e; }
                EX(k, B.con_s) = ch_exp<R>(-dcont * (R)109.0);
                xx = dco2 * ch_pow<R>(pa / (R)300.0, (R)0.5) * ((R)1. + ((R)0.0179 + (R)1.02e-4 * dt) * dt);
                e = ch_exp<R>(-xx * (R)2.656e-5);
                EX(k, B.co2_s) = e;
                for (int q = 1; q <= 5; q++) { xx = e * e; xx = xx * xx; e = xx * xx; EX(k, B.co2_s + q) = e; }
                xx = rec[CF_DN2O * K1 + k] * ((R)1. + ((R)1.4476e-3 + (R)3.6656e-6 * dt) * dt);
                e = ch_exp<R>(-xx * (R)0.25238);
                EX(k, B.n2o_s) = e;
                xx = e * e;
                R xx1 = xx * xx; xx1 = xx1 * xx1;
                const R xx2 = xx1 * xx1, xx3 = xx2 * xx2;
                EX(k, B.n2o_s + 1) = xx * xx1 * xx2 * xx3;
            }
        }
        // Planck flux of the layer, cloud optics (getirtau.code), aerosol transmittance
        if (k >= 1) {
            blayer[k] = ch_planck<R>(T, ibn, rec[CF_TA * K1 + k]);
            const R *aib = T.aib + 3 * (ibn - 1), *awb = T.awb + 4 * (ibn - 1), *aiw = T.aiw + 4 * (ibn - 1), *aww = T.aww + 4 * (ibn - 1),
                    *aig = T.aig + 4 * (ibn - 1), *awg = T.awg + 4 * (ibn - 1);
            const R wp = (rec[CF_DPPA * K1 + k] * (R)1.0e3) / (R)9.80665;        // MAPL_GRAV
            const R r1 = rec[CF_REFF1 * K1 + k], r2 = rec[CF_REFF2 * K1 + k], r4 = rec[CF_REFF4 * K1 + k];
            const R rs = r4 < (R)112.0 ? r4 : (R)112.0;
            const R tau1 = r1 <= 0 ? (R)0 : (wp * rec[CF_CWC1 * K1 + k]) * (aib[0] + aib[1] / ch_pow<R>(r1, aib[2]));
            const R tau2 = (wp * rec[CF_CWC2 * K1 + k]) * (awb[0] + (awb[1] + (awb[2] + awb[3] * r2) * r2) * r2);
            const R tau3 = (R)0.00307 * (wp * rec[CF_CWC3 * K1 + k]);
            const R tau4 = rs <= 0 ? (R)0 : (wp * rec[CF_CWC4 * K1 + k]) * (aib[0] + aib[1] / ch_pow<R>(rs, aib[2]));
            if (valid) A.taudiag[((size_t)(ibn - 1) * np + (k - 1)) * ld + i] = tau1 + tau2 + tau3 + tau4;
            R tauc = tau1 + tau2 + tau3 + tau4;
            const R fc = rec[CF_FCLD * K1 + k];
            if (tauc > (R)0.02 && fc > (R)0.01) {
                const R w1 = tau1 * (aiw[0] + (aiw[1] + (aiw[2] + aiw[3] * r1) * r1) * r1);
                const R w2 = tau2 * (aww[0] + (aww[1] + (aww[2] + aww[3] * r2) * r2) * r2);
                const R w3 = tau3 * (R)0.54;
                const R w4 = tau4 * (aiw[0] + (aiw[1] + (aiw[2] + aiw[3] * rs) * rs) * rs);
                const R ww = (w1 + w2 + w3 + w4) / tauc;
                const R g1 = w1 * (aig[0] + (aig[1] + (aig[2] + aig[3] * r1) * r1) * r1);
                const R g2 = w2 * (awg[0] + (awg[1] + (awg[2] + awg[3] * r2) * r2) * r2);
                const R g3 = w3 * (R)0.95;
                const R g4 = w4 * (aig[0] + (aig[1] + (aig[2] + aig[3] * rs) * rs) * rs);
                const R gg = (w1 + w2 + w3 + w4 != 0) ? (g1 + g2 + g3 + g4) / (w1 + w2 + w3 + w4) : (R)0.5;
                const R ff = (R)0.5 + ((R)0.3739 + ((R)0.0076 + (R)0.1185 * gg) * gg) * gg;
                R sc = (R)1. - ww * ff; sc = sc > 0 ? sc : (R)0;
                tauc = sc * tauc;
                const R tcl = ch_exp<R>((R)-1.66 * tauc);
                tcld[k] = tcl; enn[k] = fc * ((R)1.0 - tcl);
            } else { tcld[k] = 1; enn[k] = 0; }
            R tae = 1;
            if (do_aer) {      // aerosol scaling, in place as in the reference (:655-678)
                const size_t j = ((size_t)(ibn - 1) * np + (k - 1)) * ld + i;
                R ta_ = A.taua[j];
                if (ta_ > (R)0.001) {
                    R ss = A.ssaa[j];
                    if (ss > (R)0.001) {
                        const R as = A.asya[j] / ss;
                        ss = ss / ta_;
                        const R ff = (R).5 + ((R).3739 + ((R)0.0076 + (R)0.1185 * as) * as) * as;
                        ta_ = ta_ * ((R)1. - ss * ff);
                        if (valid) { A.asya[j] = as; A.ssaa[j] = ss; A.taua[j] = ta_; }
                    }
                    tae = ch_exp<R>((R)-1.66 * ta_);
                }
            }
            taer[k] = tae;
        } else { tcld[0] = 1; enn[0] = 0; taer[0] = 1; }
    }
    __syncthreads();

    // ---- P1: surface (sfcflux :2608-2720), Planck at the levels -------------------------------------------------
    R bs = 0, dbs = 0, rflxs = 0;
    {
        const int ns = A.ns;
#define S2(a, j) a[(size_t)((j) - 1) * ld + i]
#define S3(a, j) a[((size_t)(ibn - 1) * ns + ((j) - 1)) * ld + i]
        const bool bare = S3(A.ev, 1) < (R)0.0001 && S3(A.rv, 1) < (R)0.0001;
        if (S2(A.fs, 1) > (R)0.9999) {
            const R bg = ch_planck<R>(T, ibn, S2(A.tg, 1)), dbg = ch_plancd<R>(T, ibn, S2(A.tg, 1));
            if (bare) { bs = S3(A.eg, 1) * bg; dbs = S3(A.eg, 1) * dbg; rflxs = (R)1.0 - S3(A.eg, 1); }
            else {
                const R bv = ch_planck<R>(T, ibn, S2(A.tv, 1)), dbv = ch_plancd<R>(T, ibn, S2(A.tv, 1));
                R xx = S3(A.ev, 1) * bv;
                const R yy = (R)1.0 - S3(A.ev, 1) - S3(A.rv, 1), zz = (R)1.0 - S3(A.eg, 1);
                bs = yy * (S3(A.eg, 1) * bg + zz * xx) + xx;
                xx = S3(A.ev, 1) * dbv;
                dbs = yy * (S3(A.eg, 1) * dbg + zz * xx) + xx;
                rflxs = S3(A.rv, 1) + zz * yy * yy / ((R)1.0 - S3(A.rv, 1) * zz);
            }
        } else {
            for (int j = 1; j <= ns; j++) {
                const R bg = ch_planck<R>(T, ibn, S2(A.tg, j)), dbg = ch_plancd<R>(T, ibn, S2(A.tg, j));
                if (bare) {
                    bs = bs + S2(A.fs, j) * S3(A.eg, j) * bg; dbs = dbs + S2(A.fs, j) * S3(A.eg, j) * dbg;
                    rflxs = rflxs + S2(A.fs, j) * ((R)1.0 - S3(A.eg, j));
                } else {
                    const R bv = ch_planck<R>(T, ibn, S2(A.tv, j)), dbv = ch_plancd<R>(T, ibn, S2(A.tv, j));
                    R xx = S3(A.ev, j) * bv;
                    const R yy = (R)1.0 - S3(A.ev, j) - S3(A.rv, j), zz = (R)1.0 - S3(A.eg, j);
                    bs = bs + S2(A.fs, j) * (yy * (S3(A.eg, j) * bg + zz * xx) + xx);
                    xx = S3(A.ev, j) * dbv;
                    dbs = dbs + S2(A.fs, j) * (yy * (S3(A.eg, j) * dbg + zz * xx) + xx);
                    rflxs = rflxs + S2(A.fs, j) * (S3(A.rv, j) + zz * yy * yy / ((R)1.0 - S3(A.rv, j) * zz));
                }
            }
        }
#undef S2
#undef S3
    }
    if (lane == 0) { blayer[0] = blayer[1]; blayer[np + 1] = bs; }
    __syncthreads();
    for (int k = lane; k <= np + 1; k += CH_LPC) {      // (:594-606)
        R v;
        if (k >= 2 && k <= np) {
            const R dpk = rec[CF_DPPA * K1 + k] * (R)0.01, dpm = rec[CF_DPPA * K1 + k - 1] * (R)0.01;      // hPa like the reference
            v = (blayer[k - 1] * dpk + blayer[k] * dpm) / (dpm + dpk);
        } else if (k <= 1) {
            const R dp1 = rec[CF_DPPA * K1 + 1] * (R)0.01, dp2 = rec[CF_DPPA * K1 + 2] * (R)0.01;
            v = blayer[1] + (blayer[1] - blayer[2]) * dp1 / (dp1 + dp2);
        } else v = ch_planck<R>(T, ibn, A.tb[i]);
        blevel[k] = v;
    }
    // ---- P2: clouds sorted by increasing N within each super-layer (mkicx / SORTIT :2729-2781), as a rank sort ----------
    int ncld0 = 0, ncld1 = 0, ncld2 = 0;
    {
        bool anyc = false;
        for (int k = lane; k <= np; k += CH_LPC) anyc |= enn[k] > 0;
        if ((__ballot(anyc) & hmask) != 0) {
            const int ict = A.ict, icb = A.icb;
            for (int k = lane; k <= np; k += CH_LPC) {
                const int g0 = k < ict ? 0 : (k < icb ? ict : icb), g1 = k < ict ? ict - 1 : (k < icb ? icb - 1 : np);
                const R e = enn[k];
                int rank = 0;
                for (int j = g0; j <= g1; j++) { const R ej = enn[j]; rank += (ej < e || (ej == e && j < k)) ? 1 : 0; }
                icx[g0 + rank] = k;
            }
            for (int k0 = 0; k0 <= np; k0 += CH_LPC) {
                const int k = k0 + lane;
                const bool pos = k <= np && enn[k] > 0;
                ncld0 += __popcll(__ballot(pos && k < ict) & hmask);
                ncld1 += __popcll(__ballot(pos && k >= ict && k < icb) & hmask);
                ncld2 += __popcll(__ballot(pos && k >= icb) & hmask);
            }
        }
    }
    __syncthreads();

    // ---- transmittance of layer km added to a lane's running state (shared by loops 1500 and 3000) -------------------
    // the band's table constants, read once (inside the lambda they were scalar loads from *Tp in every step of loop 2000)
    R fkw[6], gkw[18];
#pragma unroll
    for (int q = 0; q < 6; q++) fkw[q] = (!B.h2otable && !B.b10bnd) ? T.fkw[(ibn - 1) * 6 + q] : (R)0;
#pragma unroll
    for (int q = 0; q < 18; q++) gkw[q] = B.ne > 1 ? T.gkw[q] : (R)0;
    const R tw1 = B.h2otable ? T.w11 : (B.co2bnd ? T.w12 : T.w13), tp1 = B.h2otable ? T.p11 : (B.co2bnd ? T.p12 : T.p13), tdwe = T.dwe, tdpe = T.dpe;
    const R *const ha = B.h2otable ? (ibn == 1 ? T.h11 : (ibn == 2 ? T.h21 : T.h81)) : (B.co2bnd ? T.c1 : T.oo1);
    const R *const hb = B.h2otable ? (ibn == 1 ? T.h12 : (ibn == 2 ? T.h22 : T.h82)) : (B.co2bnd ? T.c2 : T.oo2);
    const R *const hc = B.h2otable ? (ibn == 1 ? T.h13 : (ibn == 2 ? T.h23 : T.h83)) : (B.co2bnd ? T.c3 : T.oo3);
    auto layer_tran = [&](int km, bool full, ChState<R> &S, R &trant) {
        if (B.h2otable) {
            ch_tablup<R>(CH_NH, s_dw[km], s_pa[km], s_dt[km], S.x1, S.x2, S.x3, tw1, tp1, tdwe, tdpe, ha, hb, hc, trant, S.cell, S.tv);
            if (B.conbnd) { S.tcon[0] = S.tcon[0] * EX(km, B.con_s); trant = trant * S.tcon[0]; }
        } else if (!B.b10bnd) {        // h2okdis :2017-2143
#pragma unroll
            for (int q = 0; q < 6; q++) S.th2o[q] = S.th2o[q] * EX(km, B.h2o_s + q);
            R trn = 0;
            if (B.ne <= 1) {
#pragma unroll
                for (int q = 0; q < 6; q++) trn = trn + fkw[q] * S.th2o[q];
                if (B.ne == 1) { S.tcon[0] = S.tcon[0] * EX(km, B.con_s); trn = trn * S.tcon[0]; }
            } else {
#pragma unroll
                for (int q = 0; q < 3; q++) S.tcon[q] = S.tcon[q] * EX(km, B.con_s + q);
#pragma unroll
                for (int sb = 0; sb < 3; sb++) {
                    R s = 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) s = s + gkw[sb * 6 + q] * S.th2o[q];
                    trn = trn + s * S.tcon[sb];
                }
            }
            trant = trant * trn;
        }
        if (B.co2bnd) ch_tablup<R>(CH_NC, s_dco2[km], s_pa[km], s_dt[km], S.x1, S.x2, S.x3, tw1, tp1, tdwe, tdpe, ha, hb, hc, trant, S.cell, S.tv);
        if (B.oznbnd) ch_tablup<R>(CH_NO, s_do3[km], s_pa[km], s_dt[km], S.x1, S.x2, S.x3, tw1, tp1, tdwe, tdpe, ha, hb, hc, trant, S.cell, S.tv);
        if (full && trace) {
            if (B.n2obnd) {            // n2okdis :2148-2214, ch4kdis :2219-2282
                R xc;
                if (ibn == 6) {
                    S.tn2o[0] *= EX(km, B.n2o_s); xc = (R)0.940414 * S.tn2o[0]; S.tn2o[1] *= EX(km, B.n2o_s + 1); xc = xc + (R)0.059586 * S.tn2o[1];
                } else {
                    const R w[4] = {(R)0.561961, (R)0.138707, (R)0.240670, (R)0.058662};
                    xc = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) { S.tn2o[q] *= EX(km, B.n2o_s + q); xc = xc + w[q] * S.tn2o[q]; }
                }
                trant = trant * xc;
                if (ibn == 6) { S.tch4[0] *= EX(km, B.ch4_s); xc = S.tch4[0]; }
                else {
                    const R w[4] = {(R)0.610650, (R)0.280212, (R)0.107349, (R)0.001789};
                    xc = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) { S.tch4[q] *= EX(km, B.ch4_s + q); xc = xc + w[q] * S.tch4[q]; }
                }
                trant = trant * xc;
            }
            if (B.combnd) {            // comkdis :2287-2361
                const R w4[6] = {(R)0.12159, (R)0.24359, (R)0.24981, (R)0.26427, (R)0.07807, (R)0.04267};
                const R w5[6] = {(R)0.06869, (R)0.14795, (R)0.19512, (R)0.33446, (R)0.17199, (R)0.08179};
                R xc = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) { S.tcom[q] *= EX(km, B.com_s + q); xc = xc + (ibn == 4 ? w4[q] : w5[q]) * S.tcom[q]; }
                trant = trant * xc;
            }
            if (B.f11bnd) { S.tf11 = S.tf11 * EX(km, B.f11_s); trant = trant * S.tf11; }
            if (B.f12bnd) { S.tf12 = S.tf12 * EX(km, B.f12_s); trant = trant * S.tf12; S.tf22 = S.tf22 * EX(km, B.f22_s); trant = trant * S.tf22; }
            if (B.b10bnd) {            // b10kdis :2403-2508
                const R wh[5] = {(R)0.3153, (R)0.4604, (R)0.1326, (R)0.0798, (R)0.0119};
                const R wc[6] = {(R)0.2673, (R)0.2201, (R)0.2106, (R)0.2409, (R)0.0196, (R)0.0415};
                R xx = 0;
#pragma unroll
                for (int q = 0; q < 5; q++) { S.th2o[q] *= EX(km, B.h2o_s + q); xx = xx + wh[q] * S.th2o[q]; }
                trant = xx;
                S.tcon[0] = S.tcon[0] * EX(km, B.con_s); trant = trant * S.tcon[0];
                xx = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) { S.tco2[q] *= EX(km, B.co2_s + q); xx = xx + wc[q] * S.tco2[q]; }
                trant = trant * xx;
                S.tn2o[0] *= EX(km, B.n2o_s); xx = (R)0.970831 * S.tn2o[0]; S.tn2o[1] *= EX(km, B.n2o_s + 1); xx = xx + (R)0.029169 * S.tn2o[1];
                trant = trant * (xx - (R)1.0);
            }
        }
    };
    auto reset_state = [&](ChState<R> &S) {
#pragma unroll
        for (int q = 0; q < 6; q++) { S.th2o[q] = 1; S.tco2[q] = 1; S.tcom[q] = 1; }
#pragma unroll
        for (int q = 0; q < 4; q++) { S.tn2o[q] = 1; S.tch4[q] = 1; }
        S.tcon[0] = S.tcon[1] = S.tcon[2] = 1; S.tf11 = S.tf12 = S.tf22 = 1; S.x1 = S.x2 = S.x3 = 0; S.cell = -1;
    };

    // ---- P3: loop 1500 (:802-935): upward / downward emission of every single layer (lanes = layers) ------------------
    for (int km = lane; km <= np; km += CH_LPC) {
        ChState<R> S; reset_state(S);
        R trant = 1;
        layer_tran(km, false, S, trant);
        const R taant = trant;
        if (do_aer) trant = trant * taer[km];
        const R b0 = blevel[km], b1 = blevel[km + 1], en = enn[km];
        R dn, up;
        ch_emis<R>(((R)1. - en) * trant, b0, b1, dn, up); bd[km] = dn; bu[km] = up;
        if (do_aer) { ch_emis<R>(((R)1. - en) * taant, b0, b1, dn, up); }
        dd[km] = dn; du[km] = up;
        ch_emis<R>(trant, b0, b1, dn, up); cd[km] = dn; cu[km] = up;
        if (do_aer) { ch_emis<R>(taant, b0, b1, dn, up); }
        ad[km] = dn; au[km] = up;
    }
    if (lane == 0) { bu[np + 1] = bs; au[np + 1] = bs; cu[np + 1] = bs; du[np + 1] = bs; }
    for (int k = lane; k < K2; k += CH_LPC) { flxd[k] = 0; flcd[k] = 0; flad[k] = 0; flxad[k] = 0; }
    __syncthreads();
    R *part = A.part + ((size_t)i * CH_NB + (ibn - 1)) * CH_NKIND * K2;

    // ---- P4: loop 2000 (:948-1290): lanes = k1, lock-step walk over k2 --------------------------------------------------
    const int ict = A.ict, icb = A.icb;
    for (int k1b = 0; k1b <= np; k1b += CH_LPC) {
        const int k1 = k1b + lane;
        const bool act = k1 <= np;
        ChState<R> S; reset_state(S);
        R cldlw = 0, cldmd = 0, cldhi = 0, tranal = 1, taant = 1, trant = 1, fclr = 1;
        R axu = 0, acu = 0, aau = 0, axau = 0;                 // this lane's flxu(k1), flcu(k1), flau(k1), flxau(k1)
        R bd0 = 0, bd1 = 0, cd0 = 0, cd1 = 0, ad0 = 0, ad1 = 0, dd0 = 0, dd1 = 0;
        if (act) {
            bd1 = bd[k1]; cd1 = cd[k1]; ad1 = ad[k1]; dd1 = dd[k1];
            if (k1 > 0) { bd0 = bd[k1 - 1]; cd0 = cd[k1 - 1]; ad0 = ad[k1 - 1]; dd0 = dd[k1 - 1]; }
        }
        // the first terms of Eqs. (8.15), (8.16) (the reference adds them at k2 == k1 + 1, in front of that level pair's own term)
        if (act && ibn != 10) {
            aau = -au[k1]; acu = -cu[k1]; axu = -bu[k1]; axau = -du[k1];
            const int k2 = k1 + 1;
            const R f0 = flad[k2], f1 = flcd[k2], f2 = flxd[k2], f3 = flxad[k2];
            flad[k2] = f0 + ad1; flcd[k2] = f1 + cd1; flxd[k2] = f2 + bd1; flxad[k2] = f3 + dd1;
        }
        __builtin_amdgcn_wave_barrier();
        const int tmax = np + 1 - k1b;                        // uniform: trip count of the lane with the smallest k1
        for (int tq = 0; tq < tmax; tq++) {
            const int k2 = k1 + 1 + tq;
            if (act && k2 <= np + 1) {
                const int km = k2 - 1;
                // every LDS operand of the step is requested here, in front of the arithmetic (the aerosol-free set too: without aerosols
                // du == bu, au == cu, dd == bd, ad == cd and taant == trant, so the same expressions serve both cases and the step has no
                // uniform branches that would cut the requests into round trips of their own)
                const R ekm = enn[km], tae = taer[km];
                const R bu0 = bu[k2 - 1], bu1 = bu[k2], du0 = du[k2 - 1], du1 = du[k2], cu0 = cu[k2 - 1], cu1 = cu[k2], au0 = au[k2 - 1], au1 = au[k2];
                const R fxd = flxd[k2], fxad = flxad[k2], fcd = flcd[k2], fad = flad[k2];
                taant = 1; trant = 1; fclr = 1;
                layer_tran(km, true, S, trant);
                taant = trant;
                tranal = tranal * tae; trant = trant * tranal;          // taer == 1 without aerosols
                if (ekm >= (R)0.001) {                        // cldovlp :2513-2601
                    // the group's value is picked and put back with selects: a pointer to one of the three locals would move them to scratch
                    // memory (a load / store round trip per step, which was 30 % of a cloudy column's time)
                    const int g = km < ict ? 0 : (km < icb ? 1 : 2);
                    const int kx = g == 0 ? ncld0 : (g == 1 ? ncld1 : ncld2), ke = g == 0 ? ict - 1 : (g == 1 ? icb - 1 : np), kb = ke + 1 - kx;
                    const R cur = g == 0 ? cldhi : (g == 1 ? cldmd : cldlw);
                    R v = ekm;
                    if (!(kx == 1 || cur == 0)) {
                        v = 0;
                        for (int k = kb; k <= ke; k++) { const int j = icx[k]; if (j >= k1 && j <= km) v = enn[j] + tcld[j] * v; }
                    }
                    cldhi = g == 0 ? v : cldhi; cldmd = g == 1 ? v : cldmd; cldlw = g == 2 ? v : cldlw;
                }
                fclr = ((R)1.0 - cldhi) * ((R)1.0 - cldmd) * ((R)1.0 - cldlw);
                {
                    // products rounded before they are added, as the reference's `xx = ...; flux = flux + xx` does: a clear column's all-sky and
                    // clear-sky fluxes stay bit-identical (fclr == 1), which a product fused into one of the two sums would break
#pragma clang fp contract(off)
                    const R xu = trant * (bu0 - bu1), xau = taant * (du0 - du1), xcu = trant * (cu0 - cu1), xaa = taant * (au0 - au1);
                    axu = axu + xu * fclr;
                    axau = axau + xau * fclr;
                    acu = acu + xcu;
                    aau = aau + xaa;
                    const R xd = k1 == 0 ? -trant * bd1 : trant * (bd0 - bd1);
                    const R xad = k1 == 0 ? -taant * dd1 : taant * (dd0 - dd1);
                    const R xc = k1 == 0 ? -trant * cd1 : trant * (cd0 - cd1);
                    const R xa = k1 == 0 ? -taant * ad1 : taant * (ad0 - ad1);
                    flxd[k2] = fxd + xd * fclr;
                    flxad[k2] = fxad + xad * fclr;
                    flcd[k2] = fcd + xc;
                    flad[k2] = fad + xa;
                }
            }
            // lanes touch distinct k2 within a step; between steps the block's ONE wavefront issues its LDS instructions in
            // program order and the LDS executes a wave's instructions in order, so a compiler-level barrier is all that is needed
            // (no s_waitcnt / s_barrier: the next step's loads overlap this step's stores)
            __builtin_amdgcn_wave_barrier();
        }
        if (act && valid) {      // parked where P5's loop index k == k1 of this same lane picks them up again
            part[0 * K2 + k1] = axu; part[1 * K2 + k1] = acu; part[2 * K2 + k1] = aau; part[3 * K2 + k1] = axau;
            part[4 * K2 + k1] = trant * fclr; part[5 * K2 + k1] = taant * fclr; part[6 * K2 + k1] = trant; part[7 * K2 + k1] = taant;
            part[8 * K2 + k1] = k1 > 0 ? -dbs * (trant * fclr) : (R)0;
        }
    }
    __syncthreads();

    // ---- P5: surface emission and reflection (:1292-1315), band partials to HBM ------------------------------------------
    const bool sfc = !B.b10bnd;
    const R fxd_s = flxd[np + 1], fcd_s = flcd[np + 1], fad_s = flad[np + 1], fxad_s = flxad[np + 1];
    for (int k = lane; k <= np + 1; k += CH_LPC) {
        R xu = 0, cu_ = 0, au_ = 0, xau = 0, df = 0, t0 = 1, t1 = 1, t2 = 1, t3 = 1;      // level np+1: nothing below it
        if (k <= np) {
            xu = part[0 * K2 + k]; cu_ = part[1 * K2 + k]; au_ = part[2 * K2 + k]; xau = part[3 * K2 + k];
            t0 = part[4 * K2 + k]; t1 = part[5 * K2 + k]; t2 = part[6 * K2 + k]; t3 = part[7 * K2 + k]; df = part[8 * K2 + k];
        }
        if (sfc) {
            if (k == np + 1) { xu = -bs; cu_ = -bs; au_ = -bs; xau = -bs; df = -dbs; }
            if (k >= 1) {
                au_ = au_ - fad_s * t3 * rflxs;
                cu_ = cu_ - fcd_s * t2 * rflxs;
                xu = xu - fxd_s * t0 * rflxs;
                xau = xau - fxad_s * t1 * rflxs;
            }
        }
        if (!valid) continue;
        part[0 * K2 + k] = xu; part[1 * K2 + k] = cu_; part[2 * K2 + k] = au_; part[3 * K2 + k] = xau;
        part[4 * K2 + k] = flxd[k]; part[5 * K2 + k] = flcd[k]; part[6 * K2 + k] = flad[k]; part[7 * K2 + k] = flxad[k];
        part[8 * K2 + k] = df;
        if (k == 0) part[9 * K2] = sfc ? -bs : (R)0;     // sfcem contribution of the band
    }
#undef EX
}

template <typename R>
__global__ void __launch_bounds__(64) k_chou_bands(ChouArgs<R> A, const ChouDev<R> *__restrict__ Tp)
{
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const ChouDev<R> &T = *Tp;
#ifdef CH_ONLY_BAND
    switch (CH_ONLY_BAND - 1) {          // timing experiment: every block of the grid runs this band's body
#else
    switch (blockIdx.y) {
#endif
        case 0: chou_band_body<R, 1>(A, T, ch_smem); break;
        case 1: chou_band_body<R, 2>(A, T, ch_smem); break;
        case 2: chou_band_body<R, 3>(A, T, ch_smem); break;
        case 3: chou_band_body<R, 4>(A, T, ch_smem); break;
        case 4: chou_band_body<R, 5>(A, T, ch_smem); break;
        case 5: chou_band_body<R, 6>(A, T, ch_smem); break;
        case 6: chou_band_body<R, 7>(A, T, ch_smem); break;
        case 7: chou_band_body<R, 8>(A, T, ch_smem); break;
        case 8: chou_band_body<R, 9>(A, T, ch_smem); break;
        default: chou_band_body<R, 10>(A, T, ch_smem); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_chou_reduce: sum the band partials in band order (:1317-1328), write the API outputs.  The partials are column-major (a column's
// levels contiguous, as k_chou_bands' lanes = levels leave them), the outputs level-major with the column fastest.  Block = (64 columns,
// one of the 9 kinds): a wavefront walks the (column, level) items of 16 columns as one list - runs of np + 1 consecutive values per
// column and band - the sums go through an LDS tile [level][column], lanes = columns write 256-byte rows.  (One wavefront per column with
// lanes = levels wrote 4 bytes per lane at a stride of ld: 65 M write transactions per 100 000 columns, 1.40 ms.)
// ---------------------------------------------------------------------------------------------------
constexpr int CHR_KMAX = 96;            // levels per LDS tile (24 / 49 KB): a 72-layer column in one piece, deeper columns in chunks
template <typename R>
__global__ void __launch_bounds__(256) k_chou_reduce(ChouArgs<R> A, ChouOut<R> O, int nband)
{
    __shared__ R tile[CHR_KMAX * 65];
    const int col0 = (int)blockIdx.x * 64, q = (int)blockIdx.y, w = (int)threadIdx.x / 64, lane = (int)threadIdx.x % 64;
    const int np = A.np, K2 = np + 2, ld = A.ld, nk = np + 1;
    const int ncolb = (A.m - col0) < 64 ? (A.m - col0) : 64;
    // wavefront w: columns 16 w .. 16 w + 15 of the block
    const int c_lo = 16 * w, c_n = ncolb - c_lo < 16 ? (ncolb - c_lo < 0 ? 0 : ncolb - c_lo) : 16;
    R *o = q == 0 ? O.flxu : (q == 1 ? O.flcu : (q == 2 ? O.flau : (q == 3 ? O.flxau : (q == 4 ? O.flxd : (q == 5 ? O.flcd : (q == 6 ? O.flad :
           (q == 7 ? O.flxad : O.dfdts)))))));
    for (int k0 = 0; k0 < nk; k0 += CHR_KMAX) {          // levels k0 + 1 .. k0 + nc of 1 .. np + 1
        const int nc = (nk - k0) < CHR_KMAX ? (nk - k0) : CHR_KMAX;
        for (int t = lane; t < c_n * nc; t += 64) {
            const int c = c_lo + t / nc, kk = t % nc;
            const R *p = A.part + (size_t)(col0 + c) * CH_NB * CH_NKIND * K2 + (size_t)q * K2 + k0 + 1 + kk;
            R s = 0;
            for (int b = 0; b < nband; b++) s = s + p[(size_t)b * CH_NKIND * K2];
            tile[kk * 65 + c] = s;
        }
        __syncthreads();
        if (lane < ncolb)
            for (int kk = w; kk < nc; kk += 4) o[(size_t)(k0 + kk) * ld + col0 + lane] = tile[kk * 65 + lane];
        __syncthreads();
    }
    if (q == 0 && (int)threadIdx.x < ncolb) {
        const R *p0 = A.part + (size_t)(col0 + (int)threadIdx.x) * CH_NB * CH_NKIND * K2;
        R s = 0;
        for (int b = 0; b < nband; b++) s = s + p0[(size_t)b * CH_NKIND * K2 + 9 * K2];
        O.sfcem[col0 + threadIdx.x] = s;
    }
}

}  // namespace geosrad
