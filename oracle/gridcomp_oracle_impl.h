/* oracle/gridcomp_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (included by lw_oracle.c once per precision).
 *
 * Plain-C restatement of the GridComp data path either side of the RRTMG solvers:
 *   GEOSirrad_GridComp/GEOS_IrradGridComp.F90  (IRR)  LW_Driver RRTMG branch :3188-3372, :3487-3533, :3560-3565, :3601-3615;
 *                                                     Update_Flx :3796-3999
 *   GEOSsolar_GridComp/GEOS_SolarGridComp.F90  (SOL)  SORADCORE RRTMG branch :6113-6219, :6395-6450; Chou-Suarez branch :4484-4528;
 *                                                     UPDATE_EXPORT :7540-7579
 *   GEOS_RadiationGridComp.F90                 (RAD)  :798-819
 * PARITY UNPINNED: these routines live in the ESMF/MAPL GridComps, which cannot be built in this image (ESMF, MAPL absent; writing
 * stand-ins for them is not allowed), and the reference holds no test vectors for them.  The statements below follow the
 * reference line by line (same operand order, same loop order); they are checked by property tests only.
 * Arrays: GEOS layout, column index fastest: a(ij, k) = a[k*ncol + ij].  Pointer arrays use the enum order of include/geosrad.h.
 */

#define G2(a, k) (a)[(size_t)(k) * ncol + ij]

/* IRR:3248-3256: TLEV(1..LM+1) of one column; PLE index 0..LM, T index 1..LM */
static void SFX(lwd_tlev_col)(int ncol, int lm, int ij, const REAL *ple, const REAL *t, REAL t2m, REAL *tlev /* [lm+2], 1-based */)
{
    REAL *dp = (REAL *)malloc(sizeof(REAL) * (size_t)(lm + 1));
    dp[1] = G2(ple, 1) - G2(ple, 0);
    for (int k = 2; k <= lm; k++) {
        dp[k] = G2(ple, k) - G2(ple, k - 1);
        tlev[k] = (G2(t, k - 2) * dp[k] + G2(t, k - 1) * dp[k - 1]) / (dp[k - 1] + dp[k]);
    }
    tlev[lm + 1] = t2m;
    tlev[1] = tlev[2];
    free(dp);
}

static REAL SFX(clampr)(REAL x, REAL lo, REAL hi) { x = x > lo ? x : lo; return x < hi ? x : hi; }

/* in[] = GEOSRAD_LWD_* inputs; rr[] = the 24 RRTMG-side arrays in the order of the rrtmg_lw argument list:
 * play plev tlay tlev tsfc emis h2o o3 co2 ch4 n2o o2 cfc11 cfc12 cfc22 ccl4 cldf ciwp clwp rei rel tauaer zm alat */
void SFX(oracle_lwd_prep)(int ncol, int lm, int nb, const REAL *const *in, const double *consts, int iceflg, int liqflg, REAL *const *rr)
{
    const REAL *ple = in[0], *pl = in[1], *t = in[2], *q = in[3], *o3 = in[4], *ch4 = in[5], *n2o = in[6], *co2_3d = in[7], *cfc11 = in[8],
               *cfc12 = in[9], *hcfc22 = in[10], *fcld = in[11], *cwc_liq = in[12], *cwc_ice = in[13], *reff_liq = in[14],
               *reff_ice = in[15], *taua = in[16], *ssaa = in[17], *ts = in[18], *emis = in[19], *lats = in[20], *t2m = in[21];
    REAL *PL_R = rr[0], *PLE_R = rr[1], *T_R = rr[2], *TLEV_R = rr[3], *TSFC = rr[4], *EMISS = rr[5], *Q_R = rr[6], *O3_R = rr[7],
         *CO2_R = rr[8], *CH4_R = rr[9], *N2O_R = rr[10], *O2_R = rr[11], *CFC11_R = rr[12], *CFC12_R = rr[13], *CFC22_R = rr[14],
         *CCL4_R = rr[15], *FCLD_R = rr[16], *CICEWP = rr[17], *CLIQWP = rr[18], *REICE = rr[19], *RELIQ = rr[20], *TAUAER = rr[21],
         *ZM_R = rr[22], *ALAT = rr[23];
    const REAL co2_fixed = (REAL)consts[0], o2 = (REAL)consts[1], ccl4 = (REAL)consts[2];
    const REAL airmw = (REAL)consts[3], h2omw = (REAL)consts[4], o3mw = (REAL)consts[5], rgas = (REAL)consts[6], grav = (REAL)consts[7];
    const REAL r_h2o = airmw / h2omw, r_o3 = airmw / o3mw;      /* (MAPL_AIRMW/MAPL_H2OMW), (MAPL_AIRMW/MAPL_O3MW) */
    REAL *tlev = (REAL *)malloc(sizeof(REAL) * (size_t)(lm + 2));
    for (int ij = 0; ij < ncol; ij++) {
        TSFC[ij] = ts[ij];
        for (int b = 0; b < 16; b++) EMISS[(size_t)b * ncol + ij] = emis[ij];      /* IRR:3244 */
        ALAT[ij] = lats[ij];
        SFX(lwd_tlev_col)(ncol, lm, ij, ple, t, t2m[ij], tlev);
        for (int K = 1; K <= lm; K++) {
            const int LV = lm - K + 1;
            const REAL dp = G2(ple, LV) - G2(ple, LV - 1);
            const REAL xx = (REAL)1.02 * (REAL)100 * dp;                             /* IRR:3266 */
            G2(CLIQWP, K - 1) = xx * G2(cwc_liq, LV - 1);
            G2(CICEWP, K - 1) = xx * G2(cwc_ice, LV - 1);
            REAL reliq = G2(reff_liq, LV - 1), reice = G2(reff_ice, LV - 1);
            if (liqflg == 0) reliq = SFX(clampr)(reliq, (REAL)5.0, (REAL)10.0);      /* IRR:3272-3278 */
            else if (liqflg == 1) reliq = SFX(clampr)(reliq, (REAL)2.5, (REAL)60.0);
            if (iceflg == 0) reice = SFX(clampr)(reice, (REAL)10.0, (REAL)30.0);     /* IRR:3280-3291 */
            else if (iceflg == 1) reice = SFX(clampr)(reice, (REAL)13.0, (REAL)130.0);
            else if (iceflg == 2) reice = SFX(clampr)(reice, (REAL)5.0, (REAL)131.0);
            else if (iceflg == 3) reice = SFX(clampr)(reice, (REAL)5.0, (REAL)140.0);
            else if (iceflg == 4) reice = SFX(clampr)(reice * (REAL)2., (REAL)1.0, (REAL)200.0);
            G2(RELIQ, K - 1) = reliq; G2(REICE, K - 1) = reice;
            G2(PLE_R, K - 1) = G2(ple, LV) / (REAL)100.;                             /* IRR:3297-3298 */
            G2(TLEV_R, K - 1) = tlev[LV + 1];
            G2(PL_R, K - 1) = G2(pl, LV - 1) / (REAL)100.;                           /* IRR:3303-3321 */
            G2(T_R, K - 1) = G2(t, LV - 1);
            G2(Q_R, K - 1) = G2(q, LV - 1) / ((REAL)1. - G2(q, LV - 1)) * r_h2o;
            G2(O3_R, K - 1) = G2(o3, LV - 1) * r_o3;
            G2(CH4_R, K - 1) = G2(ch4, LV - 1);
            G2(N2O_R, K - 1) = G2(n2o, LV - 1);
            G2(CO2_R, K - 1) = co2_3d ? G2(co2_3d, LV - 1) : co2_fixed;
            G2(O2_R, K - 1) = o2;
            G2(CCL4_R, K - 1) = ccl4;
            G2(CFC11_R, K - 1) = G2(cfc11, LV - 1);
            G2(CFC12_R, K - 1) = G2(cfc12, LV - 1);
            G2(CFC22_R, K - 1) = G2(hcfc22, LV - 1);
            G2(FCLD_R, K - 1) = G2(fcld, LV - 1);
            for (int b = 0; b < 16; b++) {                                            /* IRR:3335 */
                REAL v = 0;
                if (taua && b < nb) {
                    v = taua[((size_t)b * lm + (LV - 1)) * ncol + ij] - ssaa[((size_t)b * lm + (LV - 1)) * ncol + ij];
                    v = v > 0 ? v : (REAL)0;
                }
                TAUAER[((size_t)b * lm + (K - 1)) * ncol + ij] = v;
            }
        }
        G2(PLE_R, lm) = G2(ple, 0) / (REAL)100.;                                      /* IRR:3340-3342 */
        G2(TLEV_R, lm) = tlev[1];
        G2(ZM_R, 0) = 0;                                                              /* IRR:3350-3356 */
        for (int K = 2; K <= lm; K++)
            G2(ZM_R, K - 1) = G2(ZM_R, K - 2) + rgas * G2(TLEV_R, K - 1) / grav * (G2(PL_R, K - 2) - G2(PL_R, K - 1)) / G2(PLE_R, K - 1);
    }
    /* clean up negatives (IRR:3361-3371) */
    REAL *neg[11] = {Q_R, O3_R, CH4_R, N2O_R, CO2_R, O2_R, CCL4_R, CFC11_R, CFC12_R, CFC22_R, FCLD_R};
    for (int a = 0; a < 11; a++)
        for (size_t i = 0; i < (size_t)ncol * lm; i++)
            if (neg[a][i] < 0) neg[a][i] = 0;
    free(tlev);
}

/* flux[] = uflx dflx uflxc dflxc duflx_dts duflxc_dts (ncol, LM+1), 1 = surface; out[] = GEOSRAD_LWD_* outputs (first 16) */
void SFX(oracle_lwd_post)(int ncol, int lm, int ngpt, const REAL *const *flux, const int32_t *clearCounts /* (ncol,4) */, const REAL *emis,
                          const REAL *ts, REAL *const *out)
{
    const REAL *UFLX = flux[0], *DFLX = flux[1], *UFLXC = flux[2], *DFLXC = flux[3], *DU = flux[4], *DUC = flux[5];
    for (int ij = 0; ij < ncol; ij++) {
        for (int k = 0; k < 4; k++)                                                   /* IRR:3495-3506 */
            if (out[12 + k]) out[12 + k][ij] = (REAL)1.0 - (REAL)clearCounts[(size_t)k * ncol + ij] / (REAL)ngpt;
        for (int K = 0; K <= lm; K++) {                                               /* IRR:3509-3517 */
            const int LV = lm - K + 1;
            const REAL fu = -G2(UFLX, LV - 1), fd = G2(DFLX, LV - 1), cu = -G2(UFLXC, LV - 1), cd = G2(DFLXC, LV - 1);
            const REAL d = -G2(DU, LV - 1), dc = -G2(DUC, LV - 1);
            if (out[0]) G2(out[0], K) = fu;
            if (out[1]) G2(out[1], K) = fd;
            if (out[2]) G2(out[2], K) = cu;
            if (out[3]) G2(out[3], K) = cd;
            if (out[4]) G2(out[4], K) = d;
            if (out[5]) G2(out[5], K) = dc;
            if (out[6]) G2(out[6], K) = d;                                            /* DFDTSNA = DFDTS (IRR:3564) */
            if (out[7]) G2(out[7], K) = dc;
            if (out[8]) G2(out[8], K) = fd + fu;                                      /* FLX_INT = FLXD_INT + FLXU_INT (IRR:3601) */
            if (out[9]) G2(out[9], K) = cd + cu;
        }
        REAL sf = -(G2(UFLX, 0) - G2(DFLX, 0) * ((REAL)1. - emis[ij]));                /* IRR:3522 */
        sf = -sf;                                                                     /* IRR:3608 */
        if (out[10]) out[10][ij] = sf;
        if (out[11]) out[11][ij] = ts[ij];                                            /* TS_INT = TS (IRR:3616) */
    }
}

/* Update_Flx (IRR:3796-3999); in[] / out[] in the GEOSRAD_LWU_* order */
void SFX(oracle_lw_update_flx)(int ncol, int lm, int rrtmg, int lev_mid_high, int lev_low_mid, double undef_, const REAL *const *in,
                               REAL *const *out)
{
    const REAL undef = (REAL)undef_;
    const REAL *TSINST = in[0], *TS_INT = in[1], *SFCEM_INT = in[2], *FCLD = in[3], *FLX_INT = in[4], *FLXA_INT = in[5], *FLC_INT = in[6],
               *FLA_INT = in[7], *FLXU_INT = in[8], *FLXAU_INT = in[9], *FLCU_INT = in[10], *FLAU_INT = in[11], *FLXD_INT = in[12],
               *FLXAD_INT = in[13], *FLCD_INT = in[14], *FLAD_INT = in[15], *DFDTS = in[16], *DFDTSNA = in[17], *DFDTSC = in[18],
               *DFDTSCNA = in[19];
    for (int ij = 0; ij < ncol; ij++) {
        /* CLDTT (IRR:3831-3846) */
        REAL d1 = 0, d2 = 0, d3 = 0;
        for (int k = 1; k <= lev_mid_high - 1; k++) d1 = d1 > G2(FCLD, k - 1) ? d1 : G2(FCLD, k - 1);
        REAL cldtt = ((REAL)1 - d1);
        for (int k = lev_mid_high; k <= lev_low_mid - 1; k++) d2 = d2 > G2(FCLD, k - 1) ? d2 : G2(FCLD, k - 1);
        cldtt = cldtt * ((REAL)1 - d2);
        for (int k = lev_low_mid; k <= lm; k++) d3 = d3 > G2(FCLD, k - 1) ? d3 : G2(FCLD, k - 1);
        cldtt = (REAL)1.0 - cldtt * ((REAL)1 - d3);
        if (out[31]) out[31][ij] = cldtt;
        const REAL DELT = TSINST[ij] - TS_INT[ij];                                    /* IRR:3861 */
#define SET3(o, expr) if (out[o]) G2(out[o], K) = (expr)
        for (int K = 0; K <= lm; K++) {
            SET3(0, G2(FLX_INT, K) + G2(DFDTS, K) * DELT);
            SET3(1, rrtmg ? undef : G2(FLXA_INT, K) + G2(DFDTSNA, K) * DELT);
            SET3(2, G2(FLC_INT, K) + G2(DFDTSC, K) * DELT);
            SET3(3, rrtmg ? undef : G2(FLA_INT, K) + G2(DFDTSCNA, K) * DELT);
            SET3(4, G2(FLXU_INT, K) + G2(DFDTS, K) * DELT);
            SET3(5, rrtmg ? undef : G2(FLXAU_INT, K) + G2(DFDTSNA, K) * DELT);
            SET3(6, G2(FLCU_INT, K) + G2(DFDTSC, K) * DELT);
            SET3(7, rrtmg ? undef : G2(FLAU_INT, K) + G2(DFDTSCNA, K) * DELT);
            SET3(8, G2(FLXD_INT, K));
            SET3(9, rrtmg ? undef : G2(FLXAD_INT, K));
            SET3(10, G2(FLCD_INT, K));
            SET3(11, rrtmg ? undef : G2(FLAD_INT, K));
        }
#undef SET3
#define SET2(o, expr) if (out[o]) out[o][ij] = (expr)
        SET2(12, -(G2(FLX_INT, 0) + G2(DFDTS, 0) * DELT));                             /* OLR */
        SET2(13, rrtmg ? undef : -(G2(FLXA_INT, 0) + G2(DFDTSNA, 0) * DELT));
        SET2(14, -(G2(FLC_INT, 0) + G2(DFDTSC, 0) * DELT));
        SET2(15, rrtmg ? undef : -(G2(FLA_INT, 0) + G2(DFDTSCNA, 0) * DELT));
        SET2(16, cldtt <= (REAL)0.05 ? -(G2(FLC_INT, 0) + G2(DFDTSC, 0) * DELT) : undef);
        SET2(17, -G2(DFDTS, lm));                                                      /* DSFDTS */
        SET2(18, SFCEM_INT[ij] - G2(DFDTS, lm) * DELT);                                /* SFCEM */
        SET2(19, G2(FLX_INT, lm) + SFCEM_INT[ij]);                                     /* LWS */
        SET2(20, rrtmg ? undef : G2(FLXA_INT, lm) + SFCEM_INT[ij]);
        SET2(21, G2(FLC_INT, lm) + SFCEM_INT[ij]);
        SET2(22, rrtmg ? undef : G2(FLA_INT, lm) + SFCEM_INT[ij]);
        SET2(23, cldtt <= (REAL)0.05 ? G2(FLC_INT, lm) + SFCEM_INT[ij] : undef);
        SET2(24, G2(FLX_INT, lm) + G2(DFDTS, lm) * DELT);                              /* FLNS */
        SET2(25, rrtmg ? undef : G2(FLXA_INT, lm) + G2(DFDTSNA, lm) * DELT);
        SET2(26, G2(FLC_INT, lm) + G2(DFDTSC, lm) * DELT);
        SET2(27, rrtmg ? undef : G2(FLA_INT, lm) + G2(DFDTSCNA, lm) * DELT);
        SET2(28, -G2(DFDTS, lm));                                                      /* DSFDTS0 */
        SET2(29, SFCEM_INT[ij] - G2(DFDTS, lm) * DELT);                                /* SFCEM0 */
        SET2(30, TSINST[ij]);                                                          /* TSREFF */
#undef SET2
    }
}

/* SORADCORE RRTMG branch prep (SOL:6113-6212).  in[] = GEOSRAD_SWD_* (taua/ssaa/asya normalised in place);
 * rr[] = play plev tlay tlev h2o o3 co2 ch4 o2 cldf ciwp clwp rei rel zl tauaer ssaaer asmaer */
void SFX(oracle_swd_prep)(int ncol, int lm, int nb, REAL *const *in, const double *consts, int iceflg, int liqflg, REAL *const *rr)
{
    const REAL *ple = in[0], *pl = in[1], *t = in[2], *q = in[3], *o3 = in[4], *ch4 = in[5], *cl = in[6], *ts = in[7], *qq_ice = in[8],
               *qq_liq = in[9], *rr_ice = in[10], *rr_liq = in[11];
    REAL *taua = in[12], *ssaa = in[13], *asya = in[14];
    REAL *PL_R = rr[0], *PLE_R = rr[1], *T_R = rr[2], *TLEV_R = rr[3], *Q_R = rr[4], *O3_R = rr[5], *CO2_R = rr[6], *CH4_R = rr[7],
         *O2_R = rr[8], *FCLD_R = rr[9], *CICEWP = rr[10], *CLIQWP = rr[11], *REICE = rr[12], *RELIQ = rr[13], *ZL_R = rr[14],
         *TAUAER = rr[15], *SSAAER = rr[16], *ASMAER = rr[17];
    const REAL co2 = (REAL)consts[0], o2 = (REAL)consts[1], airmw = (REAL)consts[2], h2omw = (REAL)consts[3], o3mw = (REAL)consts[4],
               rgas = (REAL)consts[5], grav = (REAL)consts[6];
    const REAL r_h2o = airmw / h2omw, r_o3 = airmw / o3mw;
    const size_t cl3 = (size_t)ncol * lm * nb;
    if (taua)                                                                         /* SOL:6116-6125 */
        for (size_t i = 0; i < cl3; i++) {
            if (taua[i] > 0 && ssaa[i] > 0) { asya[i] = asya[i] / ssaa[i]; ssaa[i] = ssaa[i] / taua[i]; }
            else { taua[i] = 0; ssaa[i] = 0; asya[i] = 0; }
        }
    REAL *dpr = (REAL *)malloc(sizeof(REAL) * (size_t)(lm + 1)), *tlev = (REAL *)malloc(sizeof(REAL) * (size_t)(lm + 2));
    for (int ij = 0; ij < ncol; ij++) {
        for (int k = 1; k <= lm; k++) dpr[k] = G2(ple, k) - G2(ple, k - 1);            /* DPR(k) = PLE(k+1) - PLE(k) (SOL:6133) */
        for (int k = 2; k <= lm; k++)                                                  /* SOL:6172-6176 */
            tlev[k] = (G2(t, k - 2) * dpr[k] + G2(t, k - 1) * dpr[k - 1]) / (dpr[k - 1] + dpr[k]);
        tlev[lm + 1] = ts[ij];
        tlev[1] = tlev[2];
        for (int K = 1; K <= lm; K++) {
            const int LV = lm - K + 1;
            G2(CICEWP, K - 1) = ((REAL)1.02 * (REAL)100 * dpr[LV]) * G2(qq_ice, LV - 1);   /* SOL:6136-6137 */
            G2(CLIQWP, K - 1) = ((REAL)1.02 * (REAL)100 * dpr[LV]) * G2(qq_liq, LV - 1);
            REAL reice = G2(rr_ice, LV - 1), reliq = G2(rr_liq, LV - 1);
            if (iceflg == 0) { if (reice < 10.) reice = 10.; if (reice > 30.) reice = 30.; }          /* SOL:6144-6161 */
            else if (iceflg == 1) { if (reice < 13.) reice = 13.; if (reice > 130.) reice = 130.; }
            else if (iceflg == 2) { if (reice < 5.) reice = 5.; if (reice > 131.) reice = 131.; }
            else if (iceflg == 3) { if (reice < 5.) reice = 5.; if (reice > 140.) reice = 140.; }
            else if (iceflg == 4) { reice = reice * (REAL)2.; if (reice < 1.) reice = 1.; if (reice > 200.) reice = 200.; }
            if (liqflg == 0) { if (reliq < 10.) reliq = 10.; if (reliq > 30.) reliq = 30.; }          /* SOL:6163-6169 */
            else if (liqflg == 1) { if (reliq < 2.5) reliq = 2.5; if (reliq > 60.) reliq = 60.; }
            G2(REICE, K - 1) = reice; G2(RELIQ, K - 1) = reliq;
            G2(PLE_R, K - 1) = G2(ple, LV) / (REAL)100.;                                /* PLE_R(K) = PLE(LM+2-K) (SOL:6180) */
            G2(TLEV_R, K - 1) = tlev[LV + 1];
            G2(PL_R, K - 1) = G2(pl, LV - 1) / (REAL)100.;
            G2(T_R, K - 1) = G2(t, LV - 1);
            G2(Q_R, K - 1) = G2(q, LV - 1) / ((REAL)1. - G2(q, LV - 1)) * r_h2o;        /* SOL:6187 */
            G2(O3_R, K - 1) = G2(o3, LV - 1) * r_o3;
            G2(CH4_R, K - 1) = G2(ch4, LV - 1);
            G2(CO2_R, K - 1) = co2;
            G2(O2_R, K - 1) = o2;
            G2(FCLD_R, K - 1) = G2(cl, LV - 1);
            for (int b = 0; b < nb; b++) {                                              /* SOL:6210-6212 */
                const size_t ga = ((size_t)b * lm + (LV - 1)) * ncol + ij, oa = ((size_t)b * lm + (K - 1)) * ncol + ij;
                TAUAER[oa] = taua ? taua[ga] : 0; SSAAER[oa] = taua ? ssaa[ga] : 0; ASMAER[oa] = taua ? asya[ga] : 0;
            }
        }
        G2(PLE_R, lm) = G2(ple, 0) / (REAL)100.;
        G2(TLEV_R, lm) = tlev[1];
        G2(ZL_R, 0) = 0;                                                                /* SOL:6200-6207 */
        for (int k = 2; k <= lm; k++)
            G2(ZL_R, k - 1) = G2(ZL_R, k - 2) + rgas * G2(TLEV_R, k - 1) / grav * (G2(PL_R, k - 2) - G2(PL_R, k - 1)) / G2(PLE_R, k - 1);
    }
    REAL *neg[6] = {Q_R, O3_R, CH4_R, CO2_R, O2_R, FCLD_R};                              /* SOL:6193-6198 */
    for (int a = 0; a < 6; a++)
        for (size_t i = 0; i < (size_t)ncol * lm; i++)
            if (neg[a][i] < 0) neg[a][i] = 0;
    free(dpr); free(tlev);
}

/* SORADCORE Chou-Suarez branch prep (SOL:4484-4528).  in[] = PLE OX QI QL QR QS RI RL RR RS (GEOS layout, model ordering);
 * consts = {O3MW, AIRMW, UNDEF}; rr[] = PLhPa (ncol,LM+1), O3 (ncol,LM), QQ3 (ncol,LM,4), RR3 (ncol,LM,4).  The imports are not modified
 * (the reference overwrites undefined radii in its packed buffer, which it then discards). */
void SFX(oracle_swc_prep)(int ncol, int lm, const REAL *const *in, const double *consts, REAL *const *rr)
{
    const REAL *ple = in[0], *ox = in[1];
    const REAL o3fac = (REAL)consts[0] / (REAL)consts[1], undef = (REAL)consts[2];
    const REAL dflt[4] = {(REAL)36.e-6, (REAL)14.e-6, (REAL)50.e-6, (REAL)50.e-6};      /* SOL:4508-4511 */
    REAL *plhpa = rr[0], *o3 = rr[1], *qq3 = rr[2], *rr3 = rr[3];
    const size_t sp = (size_t)lm * ncol;
    for (int ij = 0; ij < ncol; ij++) {
        for (int k = 0; k <= lm; k++) G2(plhpa, k) = G2(ple, k) * (REAL)0.01;            /* SOL:4490 */
        for (int k = 0; k < lm; k++) {
            const REAL pl = (REAL)0.5 * (G2(ple, k) + G2(ple, k + 1));                    /* SOL:4488 */
            REAL o = G2(ox, k);                                                           /* SOL:4523-4533 */
            if (pl < (REAL)100.) {
                const REAL x = LOG10(pl) - (REAL)2.;
                o = o * EXP((REAL)-1.5 * (x * x));
            }
            o = o * o3fac;
            G2(o3, k) = o > 0 ? o : 0;
            for (int s = 0; s < 4; s++) {
                G2(qq3 + s * sp, k) = G2(in[2 + s], k);                                  /* SOL:4502-4505 */
                REAL r = G2(in[6 + s], k);
                if (r == undef) r = dflt[s];
                G2(rr3 + s * sp, k) = r * (REAL)1.e6;                                    /* SOL:4512-4515 */
            }
        }
    }
}

/* SOL:6395-6450.  flux[] = swuflx swdflx swuflxc swdflxc; cot8[] = cotd t h m l, cotn t h m l; out[] = fsw fsc fswu fscu, cldts..cldls,
 * cottp..cotlp */
void SFX(oracle_swd_post)(int ncol, int lm, int ngpt, int aerosols, double undef_, const REAL *const *flux, const int32_t *clearCounts,
                          const REAL *const *cot8, REAL *const *out)
{
    for (int ij = 0; ij < ncol; ij++) {
        for (int L = 0; L <= lm; L++) {
            const REAL u = G2(flux[0], lm - L), d = G2(flux[1], lm - L), uc = G2(flux[2], lm - L), dc = G2(flux[3], lm - L);
            if (out[0]) G2(out[0], L) = d - u;
            if (out[1]) G2(out[1], L) = dc - uc;
            if (out[2]) G2(out[2], L) = u;
            if (out[3]) G2(out[3], L) = uc;
        }
        if (aerosols)
            for (int k = 0; k < 4; k++)
                if (out[4 + k]) out[4 + k][ij] = (REAL)1. - (REAL)clearCounts[(size_t)k * ncol + ij] / (REAL)ngpt;
        for (int k = 0; k < 4; k++)
            if (out[8 + k]) out[8 + k][ij] = (cot8[4 + k][ij] > 0 && cot8[k][ij] > 0) ? cot8[4 + k][ij] / cot8[k][ij] : (REAL)undef_;
    }
}

/* UPDATE_EXPORT flux part (SOL:7540-7579); GEOSRAD_SWU_* order */
void SFX(oracle_sw_update_export)(int ncol, int lm, int nbands, const REAL *const *in, REAL *const *out)
{
    const REAL *SLR = in[0], *FSWN = in[1], *FSCN = in[2], *FSWNAN = in[3], *FSCNAN = in[4], *FSWUN = in[5], *FSCUN = in[6],
               *FSWUNAN = in[7], *FSCUNAN = in[8], *FSWBANDN = in[9], *FSWBANDNAN = in[10];
    for (int ij = 0; ij < ncol; ij++) {
        const REAL slr = SLR[ij];
        for (int L = 0; L <= lm; L++) {
            if (out[0]) G2(out[0], L) = G2(FSWN, L) * slr;
            if (out[1]) G2(out[1], L) = G2(FSCN, L) * slr;
            if (out[2]) G2(out[2], L) = G2(FSWNAN, L) * slr;
            if (out[3]) G2(out[3], L) = G2(FSCNAN, L) * slr;
            if (out[4]) G2(out[4], L) = G2(FSWUN, L) * slr;
            if (out[5]) G2(out[5], L) = G2(FSCUN, L) * slr;
            if (out[6]) G2(out[6], L) = G2(FSWUNAN, L) * slr;
            if (out[7]) G2(out[7], L) = G2(FSCUNAN, L) * slr;
            if (out[8]) G2(out[8], L) = (G2(FSWN, L) + G2(FSWUN, L)) * slr;
            if (out[9]) G2(out[9], L) = (G2(FSCN, L) + G2(FSCUN, L)) * slr;
            if (out[10]) G2(out[10], L) = (G2(FSWNAN, L) + G2(FSWUNAN, L)) * slr;
            if (out[11]) G2(out[11], L) = (G2(FSCNAN, L) + G2(FSCUNAN, L)) * slr;
        }
        for (int b = 0; b < nbands; b++) {
            if (out[12]) G2(out[12], b) = G2(FSWBANDN, b) * slr;
            if (out[13]) G2(out[13], b) = G2(FSWBANDNAN, b) * slr;
        }
        if (out[14]) out[14][ij] = G2(FSWN, 0) * slr;
        if (out[15]) out[15][ij] = G2(FSCN, 0) * slr;
        if (out[16]) out[16][ij] = G2(FSWNAN, 0) * slr;
        if (out[17]) out[17][ij] = G2(FSCNAN, 0) * slr;
        if (out[18]) out[18][ij] = G2(FSWN, lm) * slr;
        if (out[19]) out[19][ij] = G2(FSCN, lm) * slr;
        if (out[20]) out[20][ij] = G2(FSWNAN, lm) * slr;
        if (out[21]) out[21][ij] = G2(FSCNAN, lm) * slr;
        if (out[22]) out[22][ij] = ((REAL)1. - G2(FSWN, 0)) * slr;
        if (out[23]) out[23][ij] = ((REAL)1. - G2(FSCN, 0)) * slr;
        if (out[24]) out[24][ij] = ((REAL)1. - G2(FSWNAN, 0)) * slr;
        if (out[25]) out[25][ij] = ((REAL)1. - G2(FSCNAN, 0)) * slr;
    }
}

/* RAD:798-819; GEOSRAD_RT_* order */
void SFX(oracle_rad_tendencies)(int ncol, int lm, double grav_, double cp_, const REAL *const *in, REAL *const *out)
{
    const REAL grav = (REAL)grav_, cp = (REAL)cp_;
    const REAL *PLE = in[0], *FLW = in[1], *FSW = in[2], *FLWCLR = in[3], *FSWCLR = in[4], *FSWNA = in[5], *FLA = in[6], *FSCNA = in[7],
               *DSFDTS = in[8], *SFCEM = in[9], *TRD = in[10];
    for (int ij = 0; ij < ncol; ij++) {
        if (out[8]) out[8][ij] = DSFDTS[ij];
        if (out[9]) out[9][ij] = SFCEM[ij] - DSFDTS[ij] * TRD[ij];
        if (out[10]) out[10][ij] = (G2(FSW, lm) + G2(FLW, lm));
        for (int k = 0; k < lm; k++) {
            if (out[0]) G2(out[0], k) = ((G2(FLW, k) - G2(FLW, k + 1)) + (G2(FSW, k) - G2(FSW, k + 1))) * (grav / cp);
            if (!PLE) continue;
            const REAL dmi = grav / (cp * (G2(PLE, k + 1) - G2(PLE, k)));
            if (out[1]) G2(out[1], k) = (G2(FLW, k) - G2(FLW, k + 1)) * dmi;
            if (out[2]) G2(out[2], k) = (G2(FSW, k) - G2(FSW, k + 1)) * dmi;
            if (out[3]) G2(out[3], k) = (G2(FLWCLR, k) - G2(FLWCLR, k + 1)) * dmi;
            if (out[4]) G2(out[4], k) = (G2(FSWCLR, k) - G2(FSWCLR, k + 1)) * dmi;
            if (out[5]) G2(out[5], k) = (G2(FSWNA, k) - G2(FSWNA, k + 1)) * dmi;
            if (out[6]) G2(out[6], k) = (G2(FLA, k) - G2(FLA, k + 1)) * dmi;
            if (out[7]) G2(out[7], k) = (G2(FSCNA, k) - G2(FSCNA, k + 1)) * dmi;
        }
    }
}

#undef G2
