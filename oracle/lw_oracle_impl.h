/* oracle/lw_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (the product never links or calls this).
 *
 * Plain-C restatement of the reference's RRTMG_LW column solver and of the McICA sub-column
 * generator, included twice by lw_oracle.c (REAL = float -> *_f32, REAL = double -> *_f64).
 * Each function cites the reference file:line it follows (paths relative to /root/reference):
 *   LW  = GEOSirrad_GridComp/RRTMG/rrtmg_lw/gcm_model/src
 *   SH  = GEOS_RadiationShared
 * Parity is PINNED: tests/test_oracle_vs_ref.py checks every stage (setcoef/taumol intermediates,
 * McICA masks and water paths, cldprmc, fluxes) against oracle/_ref (the reference's own Fortran
 * compiled unmodified) and the committed golden fixtures under tests/golden/.
 *
 * Array layouts: "API" arrays are exactly the reference's solver API layout, Fortran (ncol,nlay) =
 * C index [lay*ncol + col].  Tables come from the GRTB blob in Fortran column-major order and are
 * indexed with the 1-based F2/F3 macros below.
 */

#define F2(t, n1, i, j) ((t)[(size_t)((j) - 1) * (n1) + ((i) - 1)])
#define F3(t, n1, n2, i, j, k) ((t)[((size_t)((k) - 1) * (n2) + ((j) - 1)) * (n1) + ((i) - 1)])

typedef struct {
    /* common */
    const REAL *tau_tbl, *exp_tbl, *tfn_tbl, *bpade, *totplnk, *totplnkderiv, *pref, *preflog, *tref, *chi_mls;
    const REAL *delwave, *fluxfac, *oneminus, *grav, *avogad;
    const REAL *absice0, *absice1, *absice2, *absice3, *absice4, *absliq1;
    const int *ngb, *ice1b;
    /* per band (index 1..16; unused entries NULL) */
    const REAL *absa[17], *absb[17], *fracrefa[17], *fracrefb[17], *selfref[17], *forref[17];
    const REAL *ka_mn2[17], *kb_mn2[17], *ka_mn2o[17], *kb_mn2o[17], *ka_mo3[17], *kb_mo3[17];
    const REAL *ka_mco2[17], *kb_mco2[17], *ka_mo2[17], *kb_mo2[17], *ka_mco[17];
    const REAL *ccl4, *cfc11adj, *cfc12_6, *cfc12_8, *cfc22adj;
    /* condensate inhomogeneity */
    const REAL *xcw; /* (1000,140) or NULL when homogeneous */
    REAL aam[4], ram[4];
} SFX(lw_tables_t);

static SFX(lw_tables_t) SFX(T);

/* name -> slot registry used by the Python loader */
int SFX(oracle_lw_set_table)(const char *name, const void *p)
{
    SFX(lw_tables_t) *t = &SFX(T);
#define SET(nm, field) if (!strcmp(name, nm)) { t->field = p; return 0; }
    SET("tau_tbl", tau_tbl) SET("exp_tbl", exp_tbl) SET("tfn_tbl", tfn_tbl) SET("bpade", bpade)
    SET("totplnk", totplnk) SET("totplnkderiv", totplnkderiv) SET("pref", pref) SET("preflog", preflog)
    SET("tref", tref) SET("chi_mls", chi_mls) SET("delwave", delwave) SET("fluxfac", fluxfac)
    SET("oneminus", oneminus) SET("grav", grav) SET("avogad", avogad)
    SET("absice0", absice0) SET("absice1", absice1) SET("absice2", absice2) SET("absice3", absice3)
    SET("absice4", absice4) SET("absliq1", absliq1) SET("ngb", ngb) SET("ice1b", ice1b)
    SET("b05_ccl4", ccl4) SET("b06_cfc11adj", cfc11adj) SET("b06_cfc12", cfc12_6) SET("b08_cfc12", cfc12_8)
    SET("b08_cfc22adj", cfc22adj) SET("xcw", xcw)
#undef SET
    if (name[0] == 'b' && name[3] == '_') {
        int b = (name[1] - '0') * 10 + (name[2] - '0');
        const char *s = name + 4;
        if (b < 1 || b > 16) return -1;
#define SETB(nm, field) if (!strcmp(s, nm)) { t->field[b] = p; return 0; }
        SETB("absa", absa) SETB("absb", absb) SETB("fracrefa", fracrefa) SETB("fracrefb", fracrefb)
        SETB("selfref", selfref) SETB("forref", forref) SETB("ka_mn2", ka_mn2) SETB("kb_mn2", kb_mn2)
        SETB("ka_mn2o", ka_mn2o) SETB("kb_mn2o", kb_mn2o) SETB("ka_mo3", ka_mo3) SETB("kb_mo3", kb_mo3)
        SETB("ka_mco2", ka_mco2) SETB("kb_mco2", kb_mco2) SETB("ka_mo2", ka_mo2) SETB("kb_mo2", kb_mo2)
        SETB("ka_mco", ka_mco)
#undef SETB
    }
    return -1; /* unknown names are ignored by the loader */
}

void SFX(oracle_set_corr_lengths)(const REAL *adl, const REAL *rdl)
{
    for (int i = 0; i < 4; i++) { SFX(T).aam[i] = adl[i]; SFX(T).ram[i] = rdl[i]; }
}

/* ------------------------------------------------------------------------------------------------
 * per-column state produced by setcoef (LW/rrtmg_lw_setcoef.F90:23-47), 1-based layer index
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nlay, laytrop;
    REAL pwvcm;
    REAL *colh2o, *colco2, *colo3, *coln2o, *colch4, *colo2, *colco, *colbrd, *colcfc11, *colcfc12, *colcfc22,
        *colccl4, *coldry;
    REAL *forfac, *forfrac, *selffac, *selffrac, *scaleminor, *scaleminorn2, *minorfrac;
    int *jp, *jt, *jt1, *indfor, *indself, *indminor;
    REAL *rat_h2oco2, *rat_h2oco2_1, *rat_h2oo3, *rat_h2oo3_1, *rat_h2on2o, *rat_h2on2o_1, *rat_h2och4,
        *rat_h2och4_1, *rat_n2oco2, *rat_n2oco2_1, *rat_o3co2, *rat_o3co2_1;
    REAL *fac00, *fac01, *fac10, *fac11;
    REAL *planklay; /* (16, nlay)   F2(planklay,16,ib,lay)   */
    REAL *planklev; /* (16, 0:nlay) planklev[lev*16 + ib-1] */
    REAL plankbnd[17], dplankbnd[17];
    REAL *pavel;
} SFX(colstate_t);

#define NREALARR 39
static void SFX(cs_alloc)(SFX(colstate_t) * s, int nlay)
{
    size_t n = (size_t)nlay + 2;
    REAL **ra[] = {&s->colh2o, &s->colco2, &s->colo3, &s->coln2o, &s->colch4, &s->colo2, &s->colco, &s->colbrd,
                   &s->colcfc11, &s->colcfc12, &s->colcfc22, &s->colccl4, &s->coldry, &s->forfac, &s->forfrac,
                   &s->selffac, &s->selffrac, &s->scaleminor, &s->scaleminorn2, &s->minorfrac, &s->rat_h2oco2,
                   &s->rat_h2oco2_1, &s->rat_h2oo3, &s->rat_h2oo3_1, &s->rat_h2on2o, &s->rat_h2on2o_1,
                   &s->rat_h2och4, &s->rat_h2och4_1, &s->rat_n2oco2, &s->rat_n2oco2_1, &s->rat_o3co2,
                   &s->rat_o3co2_1, &s->fac00, &s->fac01, &s->fac10, &s->fac11, &s->pavel};
    int **ia[] = {&s->jp, &s->jt, &s->jt1, &s->indfor, &s->indself, &s->indminor};
    for (size_t i = 0; i < sizeof(ra) / sizeof(ra[0]); i++) *ra[i] = (REAL *)calloc(n, sizeof(REAL));
    for (size_t i = 0; i < sizeof(ia) / sizeof(ia[0]); i++) *ia[i] = (int *)calloc(n, sizeof(int));
    s->planklay = (REAL *)calloc(16 * n, sizeof(REAL));
    s->planklev = (REAL *)calloc(16 * n, sizeof(REAL));
    s->nlay = nlay;
}
static void SFX(cs_free)(SFX(colstate_t) * s)
{
    REAL *ra[] = {s->colh2o, s->colco2, s->colo3, s->coln2o, s->colch4, s->colo2, s->colco, s->colbrd,
                  s->colcfc11, s->colcfc12, s->colcfc22, s->colccl4, s->coldry, s->forfac, s->forfrac,
                  s->selffac, s->selffrac, s->scaleminor, s->scaleminorn2, s->minorfrac, s->rat_h2oco2,
                  s->rat_h2oco2_1, s->rat_h2oo3, s->rat_h2oo3_1, s->rat_h2on2o, s->rat_h2on2o_1, s->rat_h2och4,
                  s->rat_h2och4_1, s->rat_n2oco2, s->rat_n2oco2_1, s->rat_o3co2, s->rat_o3co2_1, s->fac00,
                  s->fac01, s->fac10, s->fac11, s->pavel, s->planklay, s->planklev};
    int *ia[] = {s->jp, s->jt, s->jt1, s->indfor, s->indself, s->indminor};
    for (size_t i = 0; i < sizeof(ra) / sizeof(ra[0]); i++) free(ra[i]);
    for (size_t i = 0; i < sizeof(ia) / sizeof(ia[0]); i++) free(ia[i]);
}

static inline int SFX(clampi)(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------------
 * setcoef for ONE column (LW/rrtmg_lw_setcoef.F90:52-584), istart = 1.
 * Inputs are 1-based per-layer arrays for this column (index 0 of the level arrays = surface).
 * Returns 0, or 1 on "RRTMG LW pressure misordering" (:445-453).
 * ---------------------------------------------------------------------------------------------- */
static int SFX(setcoef_col)(SFX(colstate_t) * s, int dudTs, const REAL *pavel, const REAL *tavel, const REAL *pz,
                            const REAL *tz, REAL tbound, const REAL *semiss /*1..16*/, const REAL *h2ovmr,
                            const REAL *o3vmr, const REAL *co2vmr, const REAL *ch4vmr, const REAL *n2ovmr,
                            const REAL *o2vmr, const REAL *covmr, const REAL *cfc11vmr, const REAL *cfc12vmr,
                            const REAL *cfc22vmr, const REAL *ccl4vmr)
{
    const SFX(lw_tables_t) *t = &SFX(T);
    const int nlay = s->nlay;
    const REAL amd = (REAL)28.9660, amw = (REAL)18.0160;
    const REAL stpfac = (REAL)296. / (REAL)1013.;
    const REAL grav = *t->grav, avogad = *t->avogad;
    REAL *wbroad = (REAL *)calloc((size_t)nlay + 2, sizeof(REAL));

    /* coldry (:206-235) */
    for (int lay = 1; lay <= nlay; lay++) {
        REAL amm = ((REAL)1. - h2ovmr[lay]) * amd + h2ovmr[lay] * amw;
        s->coldry[lay] = (pz[lay - 1] - pz[lay]) * (REAL)1.e3 * avogad /
                         ((REAL)1.e2 * grav * amm * ((REAL)1. + h2ovmr[lay]));
        s->pavel[lay] = pavel[lay];
    }
    /* pwvcm, wbroad (:240-272) */
    REAL amttl = 0, wvttl = 0;
    for (int lay = 1; lay <= nlay; lay++) {
        REAL summol = co2vmr[lay] + o3vmr[lay] + n2ovmr[lay] + ch4vmr[lay] + o2vmr[lay];
        wbroad[lay] = s->coldry[lay] * ((REAL)1. - summol);
        REAL btemp = h2ovmr[lay] * s->coldry[lay];
        amttl = amttl + s->coldry[lay] + btemp;
        wvttl = wvttl + btemp;
    }
    REAL wvsh = (amw * wvttl) / (amd * amttl);
    s->pwvcm = wvsh * ((REAL)1.e3 * pz[0]) / ((REAL)1.e2 * grav);

    /* Planck indices for boundary and level 0 (:277-296) */
    int indbound = SFX(clampi)((int)(tbound - (REAL)159.), 1, 180);
    REAL tbndfrac = tbound - (REAL)159. - (REAL)indbound;
    int indlev0 = SFX(clampi)((int)(tz[0] - (REAL)159.), 1, 180);
    REAL t0frac = tz[0] - (REAL)159. - (REAL)indlev0;

    int upper_found = 0, rc = 0;
    s->laytrop = 0;
    for (int lay = 1; lay <= nlay; lay++) {
        REAL lcoldry = s->coldry[lay];
        REAL wv = h2ovmr[lay] * lcoldry;
        int indlay = SFX(clampi)((int)(tavel[lay] - (REAL)159.), 1, 180);
        REAL tlayfrac = tavel[lay] - (REAL)159. - (REAL)indlay;
        int indlev = SFX(clampi)((int)(tz[lay] - (REAL)159.), 1, 180);
        REAL tlevfrac = tz[lay] - (REAL)159. - (REAL)indlev;
        /* Planck functions, all 16 bands (istart /= 16 -> band 16 uses totplnk too; :325-394) */
        for (int ib = 1; ib <= 16; ib++) {
            REAL dbdtlev;
            if (lay == 1) {
                dbdtlev = F2(t->totplnk, 181, indbound + 1, ib) - F2(t->totplnk, 181, indbound, ib);
                s->plankbnd[ib] = semiss[ib] * (F2(t->totplnk, 181, indbound, ib) + tbndfrac * dbdtlev);
                dbdtlev = F2(t->totplnk, 181, indlev0 + 1, ib) - F2(t->totplnk, 181, indlev0, ib);
                s->planklev[0 * 16 + ib - 1] = F2(t->totplnk, 181, indlev0, ib) + t0frac * dbdtlev;
                if (dudTs) {
                    dbdtlev = F2(t->totplnkderiv, 181, indbound + 1, ib) - F2(t->totplnkderiv, 181, indbound, ib);
                    s->dplankbnd[ib] = semiss[ib] * (F2(t->totplnkderiv, 181, indbound, ib) + tbndfrac * dbdtlev);
                } else
                    s->dplankbnd[ib] = 0;
            }
            dbdtlev = F2(t->totplnk, 181, indlev + 1, ib) - F2(t->totplnk, 181, indlev, ib);
            s->planklev[lay * 16 + ib - 1] = F2(t->totplnk, 181, indlev, ib) + tlevfrac * dbdtlev;
            REAL dbdtlay = F2(t->totplnk, 181, indlay + 1, ib) - F2(t->totplnk, 181, indlay, ib);
            F2(s->planklay, 16, ib, lay) = F2(t->totplnk, 181, indlay, ib) + tlayfrac * dbdtlay;
        }
        /* pressure / temperature interpolation (:401-433) */
        REAL plog = LOG(pavel[lay]);
        int jp = SFX(clampi)((int)((REAL)36. - (REAL)5 * (plog + (REAL)0.04)), 1, 58);
        s->jp[lay] = jp;
        int jp1 = jp + 1;
        REAL fp = (REAL)5. * (t->preflog[jp - 1] - plog);
        int jt = SFX(clampi)((int)((REAL)3. + (tavel[lay] - t->tref[jp - 1]) / (REAL)15.), 1, 4);
        s->jt[lay] = jt;
        REAL ft = ((tavel[lay] - t->tref[jp - 1]) / (REAL)15.) - (REAL)(jt - 3);
        int jt1 = SFX(clampi)((int)((REAL)3. + (tavel[lay] - t->tref[jp1 - 1]) / (REAL)15.), 1, 4);
        s->jt1[lay] = jt1;
        REAL ft1 = ((tavel[lay] - t->tref[jp1 - 1]) / (REAL)15.) - (REAL)(jt1 - 3);
        REAL water = wv / lcoldry;
        REAL scalefac = pavel[lay] * stpfac / tavel[lay];
        REAL factor;
#define CHI(m, j) F2(t->chi_mls, 7, m, j)
        if (plog > (REAL)4.56) { /* lower atmosphere (:443-497) */
            if (upper_found) rc = 1;
            s->laytrop += 1;
            s->forfac[lay] = scalefac / ((REAL)1. + water);
            factor = ((REAL)332. - tavel[lay]) / (REAL)36.;
            s->indfor[lay] = SFX(clampi)((int)factor, 1, 2); /* min(2,max(1,int(factor))) */
            s->forfrac[lay] = factor - (REAL)s->indfor[lay];
            s->selffac[lay] = water * s->forfac[lay];
            factor = (tavel[lay] - (REAL)188.) / (REAL)7.2;
            s->indself[lay] = SFX(clampi)((int)factor - 7, 1, 9);
            s->selffrac[lay] = factor - (REAL)(s->indself[lay] + 7);
            s->scaleminor[lay] = pavel[lay] / tavel[lay];
            s->scaleminorn2[lay] = (pavel[lay] / tavel[lay]) * (wbroad[lay] / (lcoldry + wv));
            factor = (tavel[lay] - (REAL)180.8) / (REAL)7.2;
            s->indminor[lay] = SFX(clampi)((int)factor, 1, 18);
            s->minorfrac[lay] = factor - (REAL)s->indminor[lay];
            s->rat_h2oco2[lay] = CHI(1, jp) / CHI(2, jp);
            s->rat_h2oco2_1[lay] = CHI(1, jp + 1) / CHI(2, jp + 1);
            s->rat_h2oo3[lay] = CHI(1, jp) / CHI(3, jp);
            s->rat_h2oo3_1[lay] = CHI(1, jp + 1) / CHI(3, jp + 1);
            s->rat_h2on2o[lay] = CHI(1, jp) / CHI(4, jp);
            s->rat_h2on2o_1[lay] = CHI(1, jp + 1) / CHI(4, jp + 1);
            s->rat_h2och4[lay] = CHI(1, jp) / CHI(6, jp);
            s->rat_h2och4_1[lay] = CHI(1, jp + 1) / CHI(6, jp + 1);
            s->rat_n2oco2[lay] = CHI(4, jp) / CHI(2, jp);
            s->rat_n2oco2_1[lay] = CHI(4, jp + 1) / CHI(2, jp + 1);
        } else { /* upper atmosphere (:499-541) */
            upper_found = 1;
            s->forfac[lay] = scalefac / ((REAL)1. + water);
            factor = (tavel[lay] - (REAL)188.) / (REAL)36.;
            s->indfor[lay] = 3;
            s->forfrac[lay] = factor - (REAL)1.;
            s->selffac[lay] = 0;
            s->scaleminor[lay] = pavel[lay] / tavel[lay];
            s->scaleminorn2[lay] = (pavel[lay] / tavel[lay]) * (wbroad[lay] / (lcoldry + wv));
            factor = (tavel[lay] - (REAL)180.8) / (REAL)7.2;
            s->indminor[lay] = SFX(clampi)((int)factor, 1, 18);
            s->minorfrac[lay] = factor - (REAL)s->indminor[lay];
            s->rat_h2oco2[lay] = CHI(1, jp) / CHI(2, jp);
            s->rat_h2oco2_1[lay] = CHI(1, jp + 1) / CHI(2, jp + 1);
            s->rat_o3co2[lay] = CHI(3, jp) / CHI(2, jp);
            s->rat_o3co2_1[lay] = CHI(3, jp + 1) / CHI(2, jp + 1);
        }
        /* column amounts (:545-563) */
        s->colh2o[lay] = (REAL)1.e-20 * h2ovmr[lay] * lcoldry;
        s->colco2[lay] = (REAL)1.e-20 * co2vmr[lay] * lcoldry;
        s->colo3[lay] = (REAL)1.e-20 * o3vmr[lay] * lcoldry;
        s->coln2o[lay] = (REAL)1.e-20 * n2ovmr[lay] * lcoldry;
        s->colch4[lay] = (REAL)1.e-20 * ch4vmr[lay] * lcoldry;
        s->colo2[lay] = (REAL)1.e-20 * o2vmr[lay] * lcoldry;
        s->colco[lay] = (REAL)1.e-20 * covmr[lay] * lcoldry;
        s->colcfc11[lay] = (REAL)1.e-20 * cfc11vmr[lay] * lcoldry;
        s->colcfc12[lay] = (REAL)1.e-20 * cfc12vmr[lay] * lcoldry;
        s->colcfc22[lay] = (REAL)1.e-20 * cfc22vmr[lay] * lcoldry;
        s->colccl4[lay] = (REAL)1.e-20 * ccl4vmr[lay] * lcoldry;
        s->colbrd[lay] = (REAL)1.e-20 * wbroad[lay];
        if (s->colco2[lay] == 0) s->colco2[lay] = (REAL)1.e-32 * lcoldry;
        if (s->colo3[lay] == 0) s->colo3[lay] = (REAL)1.e-32 * lcoldry;
        if (s->coln2o[lay] == 0) s->coln2o[lay] = (REAL)1.e-32 * lcoldry;
        if (s->colch4[lay] == 0) s->colch4[lay] = (REAL)1.e-32 * lcoldry;
        if (s->colco[lay] == 0) s->colco[lay] = (REAL)1.e-32 * lcoldry;
        /* interpolation factors (:571-579) */
        REAL compfp = (REAL)1. - fp;
        s->fac10[lay] = compfp * ft;
        s->fac00[lay] = compfp * ((REAL)1. - ft);
        s->fac11[lay] = fp * ft1;
        s->fac01[lay] = fp * ((REAL)1. - ft1);
        s->selffac[lay] = s->colh2o[lay] * s->selffac[lay];
        s->forfac[lay] = s->colh2o[lay] * s->forfac[lay];
    }
    free(wbroad);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * taumol helpers
 * ---------------------------------------------------------------------------------------------- */
typedef struct { REAL speccomb, specparm, fs; int js; } SFX(spec_t);

/* binary species parameter (e.g. LW/rrtmg_lw_taumol.F90:435-441) */
static inline SFX(spec_t) SFX(spec)(REAL cola, REAL rat, REAL colb, REAL mult, REAL oneminus)
{
    SFX(spec_t) r;
    r.speccomb = cola + rat * colb;
    r.specparm = cola / r.speccomb;
    if (r.specparm >= oneminus) r.specparm = oneminus;
    REAL specmult = mult * r.specparm;
    r.js = 1 + (int)specmult;
    r.fs = FMOD(specmult, (REAL)1.0);
    return r;
}

/* key-species contribution of one (p-level) side in the lower atmosphere of a binary band, incl. the
 * cubic edge treatment (LW/rrtmg_lw_taumol.F90:482-541 prep, :553-606 use).  facA multiplies the lower
 * reference temperature row, facB the higher (ind+9*..): rows ind, ind+1, (ind+2 | ind-1), +9 ... */
static inline REAL SFX(major_a)(const REAL *absa, int n1, int ig, int ind, SFX(spec_t) sp, REAL facA, REAL facB)
{
    if (sp.specparm < (REAL)0.125) {
        REAL p = sp.fs - (REAL)1;
        REAL p4 = ((p * p) * p) * p;
        REAL fk0 = p4, fk1 = (REAL)1 - p - (REAL)2.0 * p4, fk2 = p + p4;
        REAL f0A = fk0 * facA, f1A = fk1 * facA, f2A = fk2 * facA;
        REAL f0B = fk0 * facB, f1B = fk1 * facB, f2B = fk2 * facB;
        return sp.speccomb * (f0A * F2(absa, n1, ind, ig) + f1A * F2(absa, n1, ind + 1, ig) +
                              f2A * F2(absa, n1, ind + 2, ig) + f0B * F2(absa, n1, ind + 9, ig) +
                              f1B * F2(absa, n1, ind + 10, ig) + f2B * F2(absa, n1, ind + 11, ig));
    } else if (sp.specparm > (REAL)0.875) {
        REAL p = -sp.fs;
        REAL p4 = ((p * p) * p) * p;
        REAL fk0 = p4, fk1 = (REAL)1 - p - (REAL)2.0 * p4, fk2 = p + p4;
        REAL f0A = fk0 * facA, f1A = fk1 * facA, f2A = fk2 * facA;
        REAL f0B = fk0 * facB, f1B = fk1 * facB, f2B = fk2 * facB;
        return sp.speccomb * (f2A * F2(absa, n1, ind - 1, ig) + f1A * F2(absa, n1, ind, ig) +
                              f0A * F2(absa, n1, ind + 1, ig) + f2B * F2(absa, n1, ind + 8, ig) +
                              f1B * F2(absa, n1, ind + 9, ig) + f0B * F2(absa, n1, ind + 10, ig));
    } else {
        REAL f0A = ((REAL)1. - sp.fs) * facA, f0B = ((REAL)1. - sp.fs) * facB;
        REAL f1A = sp.fs * facA, f1B = sp.fs * facB;
        return sp.speccomb * (f0A * F2(absa, n1, ind, ig) + f1A * F2(absa, n1, ind + 1, ig) +
                              f0B * F2(absa, n1, ind + 9, ig) + f1B * F2(absa, n1, ind + 10, ig));
    }
}

/* upper-atmosphere binary (nspb = 5) side: rows ind, ind+1, ind+5, ind+6 (e.g. :706-716) */
static inline REAL SFX(major_b5)(const REAL *absb, int ig, int ind, SFX(spec_t) sp, REAL facA, REAL facB)
{
    REAL f0A = ((REAL)1. - sp.fs) * facA, f0B = ((REAL)1. - sp.fs) * facB;
    REAL f1A = sp.fs * facA, f1B = sp.fs * facB;
    return sp.speccomb * (f0A * F2(absb, 1175, ind, ig) + f1A * F2(absb, 1175, ind + 1, ig) +
                          f0B * F2(absb, 1175, ind + 5, ig) + f1B * F2(absb, 1175, ind + 6, ig));
}

/* single key species: 4-point (p,T) interpolation (e.g. :240-244) */
static inline REAL SFX(major1)(const REAL *tab, int n1, int ig, int ind0, int ind1, REAL f00, REAL f10, REAL f01,
                               REAL f11)
{
    return f00 * F2(tab, n1, ind0, ig) + f10 * F2(tab, n1, ind0 + 1, ig) + f01 * F2(tab, n1, ind1, ig) +
           f11 * F2(tab, n1, ind1 + 1, ig);
}

static inline REAL SFX(lin2)(const REAL *tab, int n1, int i, int ig, REAL frac)
{
    return F2(tab, n1, i, ig) + frac * (F2(tab, n1, i + 1, ig) - F2(tab, n1, i, ig));
}

/* minor gas on a (species-parameter, T) grid: ka_mX(n1,19,ng) (e.g. :546-551) */
static inline REAL SFX(minor2)(const REAL *tab, int n1, int jm, int indm, int ig, REAL fm, REAL minorfrac)
{
    REAL m1 = F3(tab, n1, 19, jm, indm, ig) + fm * (F3(tab, n1, 19, jm + 1, indm, ig) - F3(tab, n1, 19, jm, indm, ig));
    REAL m2 = F3(tab, n1, 19, jm, indm + 1, ig) +
              fm * (F3(tab, n1, 19, jm + 1, indm + 1, ig) - F3(tab, n1, 19, jm, indm + 1, ig));
    return m1 + minorfrac * (m2 - m1);
}

/* "too much of a minor gas" column adjustment (e.g. :461-468) */
static inline REAL SFX(adjcol)(REAL colx, REAL coldry, REAL chiref, REAL thresh, REAL a, REAL pw)
{
    REAL chi = colx / coldry;
    REAL rat = (REAL)1.e20 * chi / chiref;
    if (rat > thresh) {
        REAL adjfac = a + POW(rat - a, pw);
        return adjfac * chiref * coldry * (REAL)1.e-20;
    }
    return colx;
}

/* ------------------------------------------------------------------------------------------------
 * taumol for ONE column (LW/rrtmg_lw_taumol.F90:155-3146).  taug/pfracs: F2(x, nlay, lay, ig).
 * taua: F2(taua, nlay, lay, ibnd) aerosol optical depth of this column.
 * ---------------------------------------------------------------------------------------------- */
static void SFX(taumol_col)(const SFX(colstate_t) * s, const REAL *taua, REAL *taug, REAL *pfracs)
{
    const SFX(lw_tables_t) *t = &SFX(T);
    const int nlay = s->nlay, laytrop = s->laytrop;
    const REAL oneminus = *t->oneminus;
    static const int ngs[17] = {0, 10, 22, 38, 52, 68, 76, 88, 96, 108, 114, 122, 130, 134, 136, 138, 140};
    static const int ngv[17] = {0, 10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
#define TAUG(lay, g) F2(taug, nlay, lay, g)
#define PFR(lay, g) F2(pfracs, nlay, lay, g)
#define IND0A(n) (((s->jp[lay] - 1) * 5 + (s->jt[lay] - 1)) * (n))
#define IND1A(n) ((s->jp[lay] * 5 + (s->jt1[lay] - 1)) * (n))
#define IND0B(n) (((s->jp[lay] - 13) * 5 + (s->jt[lay] - 1)) * (n))
#define IND1B(n) (((s->jp[lay] - 12) * 5 + (s->jt1[lay] - 1)) * (n))
#define TAUSELF(b) (s->selffac[lay] * SFX(lin2)(t->selfref[b], 10, s->indself[lay], ig, s->selffrac[lay]))
#define TAUFOR(b) (s->forfac[lay] * SFX(lin2)(t->forref[b], 4, s->indfor[lay], ig, s->forfrac[lay]))
#define PLANCKA(b, n, jpl, fpl) (F2(t->fracrefa[b], n, ig, jpl) + (fpl) * (F2(t->fracrefa[b], n, ig, (jpl) + 1) - F2(t->fracrefa[b], n, ig, jpl)))
#define PLANCKB(b, n, jpl, fpl) (F2(t->fracrefb[b], n, ig, jpl) + (fpl) * (F2(t->fracrefb[b], n, ig, (jpl) + 1) - F2(t->fracrefb[b], n, ig, jpl)))

    for (int lay = 1; lay <= nlay; lay++) {
        const int lower = (lay <= laytrop);
        const REAL f00 = s->fac00[lay], f10 = s->fac10[lay], f01 = s->fac01[lay], f11 = s->fac11[lay];
        const int jp = s->jp[lay], indm = s->indminor[lay];
        const REAL mf = s->minorfrac[lay];

        /* ---- band 1 (:214-291): h2o; minor n2 ---------------------------------------------------- */
        {
            const int b = 1, n = ngv[b], o = 0;
            REAL pp = s->pavel[lay];
            REAL scalen2 = s->colbrd[lay] * s->scaleminorn2[lay];
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                REAL corradj = 1;
                if (pp < (REAL)250.) corradj = (REAL)1. - (REAL)0.15 * ((REAL)250. - pp) / (REAL)154.4;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL taun2 = scalen2 * SFX(lin2)(t->ka_mn2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = corradj * (s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                                   tauself + taufor + taun2);
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                REAL corradj = (REAL)1. - (REAL)0.15 * (pp / (REAL)95.6);
                for (int ig = 1; ig <= n; ig++) {
                    REAL taufor = TAUFOR(b);
                    REAL taun2 = scalen2 * SFX(lin2)(t->kb_mn2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = corradj * (s->colh2o[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) +
                                                   taufor + taun2);
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 2 (:296-363): h2o ----------------------------------------------------------------- */
        {
            const int b = 2, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                REAL pp = s->pavel[lay];
                REAL corradj = (REAL)1. - (REAL).05 * (pp - (REAL)100.) / (REAL)900.;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    TAUG(lay, o + ig) = corradj * (s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                                   tauself + taufor);
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL taufor = TAUFOR(b);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) + taufor;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 3 (:368-727): h2o,co2; minor n2o ---------------------------------------------------- */
        {
            const int b = 3, n = ngv[b], o = ngs[b - 1];
            REAL adjcoln2o = SFX(adjcol)(s->coln2o[lay], s->coldry[lay], CHI(4, jp + 1), (REAL)1.5, (REAL)0.5, (REAL)0.65);
            if (lower) {
                REAL refrat_planck_a = CHI(1, 9) / CHI(2, 9), refrat_m_a = CHI(1, 3) / CHI(2, 3);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oco2[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oco2_1[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->colh2o[lay], refrat_m_a, s->colco2[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colco2[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absn2o = SFX(minor2)(t->ka_mn2o[b], 9, sm.js, indm, ig, sm.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + adjcoln2o * absn2o;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                REAL refrat_planck_b = CHI(1, 13) / CHI(2, 13), refrat_m_b = refrat_planck_b;
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oco2[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oco2_1[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->colh2o[lay], refrat_m_b, s->colco2[lay], 4, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_b, s->colco2[lay], 4, oneminus);
                int ind0 = IND0B(5) + sp.js, ind1 = IND1B(5) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL taufor = TAUFOR(b);
                    REAL absn2o = SFX(minor2)(t->kb_mn2o[b], 5, sm.js, indm, ig, sm.fs, mf);
                    TAUG(lay, o + ig) = SFX(major_b5)(t->absb[b], ig, ind0, sp, f00, f10) +
                                        SFX(major_b5)(t->absb[b], ig, ind1, sp1, f01, f11) + taufor + adjcoln2o * absn2o;
                    PFR(lay, o + ig) = PLANCKB(b, n, spl.js, spl.fs);
                }
            }
        }
        /* ---- band 4 (:732-962): h2o,co2 | o3,co2 --------------------------------------------------------- */
        {
            const int b = 4, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 11) / CHI(2, 11);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oco2[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oco2_1[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colco2[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                REAL refrat_planck_b = CHI(3, 13) / CHI(2, 13);
                SFX(spec_t) sp = SFX(spec)(s->colo3[lay], s->rat_o3co2[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colo3[lay], s->rat_o3co2_1[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colo3[lay], refrat_planck_b, s->colco2[lay], 4, oneminus);
                int ind0 = IND0B(5) + sp.js, ind1 = IND1B(5) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    TAUG(lay, o + ig) = SFX(major_b5)(t->absb[b], ig, ind0, sp, f00, f10) +
                                        SFX(major_b5)(t->absb[b], ig, ind1, sp1, f01, f11);
                    PFR(lay, o + ig) = PLANCKB(b, n, spl.js, spl.fs);
                }
                /* empirical stratospheric-cooling tweak (:951-957) */
                TAUG(lay, o + 8) = TAUG(lay, o + 8) * (REAL)0.92;
                TAUG(lay, o + 9) = TAUG(lay, o + 9) * (REAL)0.88;
                TAUG(lay, o + 10) = TAUG(lay, o + 10) * (REAL)1.07;
                TAUG(lay, o + 11) = TAUG(lay, o + 11) * (REAL)1.1;
                TAUG(lay, o + 12) = TAUG(lay, o + 12) * (REAL)0.99;
                TAUG(lay, o + 13) = TAUG(lay, o + 13) * (REAL)0.88;
                TAUG(lay, o + 14) = TAUG(lay, o + 14) * (REAL)0.943;
            }
        }
        /* ---- band 5 (:967-1229): h2o,co2 | o3,co2; minor o3, ccl4 ------------------------------------------ */
        {
            const int b = 5, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 5) / CHI(2, 5), refrat_m_a = CHI(1, 7) / CHI(2, 7);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oco2[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oco2_1[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->colh2o[lay], refrat_m_a, s->colco2[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colco2[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL abso3 = SFX(minor2)(t->ka_mo3[b], 9, sm.js, indm, ig, sm.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + abso3 * s->colo3[lay] +
                                        s->colccl4[lay] * t->ccl4[ig - 1];
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                REAL refrat_planck_b = CHI(3, 43) / CHI(2, 43);
                SFX(spec_t) sp = SFX(spec)(s->colo3[lay], s->rat_o3co2[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colo3[lay], s->rat_o3co2_1[lay], s->colco2[lay], 4, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colo3[lay], refrat_planck_b, s->colco2[lay], 4, oneminus);
                int ind0 = IND0B(5) + sp.js, ind1 = IND1B(5) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    TAUG(lay, o + ig) = SFX(major_b5)(t->absb[b], ig, ind0, sp, f00, f10) +
                                        SFX(major_b5)(t->absb[b], ig, ind1, sp1, f01, f11) +
                                        s->colccl4[lay] * t->ccl4[ig - 1];
                    PFR(lay, o + ig) = PLANCKB(b, n, spl.js, spl.fs);
                }
            }
        }
        /* ---- band 6 (:1234-1322): h2o; minor co2, cfc11, cfc12 ----------------------------------------------- */
        {
            const int b = 6, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL adjcolco2 = SFX(adjcol)(s->colco2[lay], s->coldry[lay], CHI(2, jp + 1), (REAL)3.0, (REAL)2.0, (REAL)0.77);
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absco2 = SFX(lin2)(t->ka_mco2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                        tauself + taufor + adjcolco2 * absco2 + s->colcfc11[lay] * t->cfc11adj[ig - 1] +
                                        s->colcfc12[lay] * t->cfc12_6[ig - 1];
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                for (int ig = 1; ig <= n; ig++) {
                    TAUG(lay, o + ig) = (REAL)0.0 + s->colcfc11[lay] * t->cfc11adj[ig - 1] + s->colcfc12[lay] * t->cfc12_6[ig - 1];
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            }
        }
        /* ---- band 7 (:1327-1601): h2o,o3 | o3; minor co2 ---------------------------------------------------------- */
        {
            const int b = 7, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 3) / CHI(3, 3), refrat_m_a = refrat_planck_a;
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oo3[lay], s->colo3[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oo3_1[lay], s->colo3[lay], 8, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->colh2o[lay], refrat_m_a, s->colo3[lay], 8, oneminus);
                REAL adjcolco2 = SFX(adjcol)(s->colco2[lay], s->coldry[lay], CHI(2, jp + 1), (REAL)3.0, (REAL)3.0, (REAL)0.79);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colo3[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absco2 = SFX(minor2)(t->ka_mco2[b], 9, sm.js, indm, ig, sm.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + adjcolco2 * absco2;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                REAL adjcolco2 = SFX(adjcol)(s->colco2[lay], s->coldry[lay], CHI(2, jp + 1), (REAL)3.0, (REAL)2.0, (REAL)0.79);
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL absco2 = SFX(lin2)(t->kb_mco2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colo3[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) +
                                        adjcolco2 * absco2;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
                TAUG(lay, o + 6) = TAUG(lay, o + 6) * (REAL)0.92;
                TAUG(lay, o + 7) = TAUG(lay, o + 7) * (REAL)0.88;
                TAUG(lay, o + 8) = TAUG(lay, o + 8) * (REAL)1.07;
                TAUG(lay, o + 9) = TAUG(lay, o + 9) * (REAL)1.1;
                TAUG(lay, o + 10) = TAUG(lay, o + 10) * (REAL)0.99;
                TAUG(lay, o + 11) = TAUG(lay, o + 11) * (REAL)0.855;
            }
        }
        /* ---- band 8 (:1606-1733): h2o | o3; minor co2, o3, n2o, cfc12, cfc22 ------------------------------------------ */
        {
            const int b = 8, n = ngv[b], o = ngs[b - 1];
            REAL adjcolco2 = SFX(adjcol)(s->colco2[lay], s->coldry[lay], CHI(2, jp + 1), (REAL)3.0, (REAL)2.0, (REAL)0.65);
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absco2 = SFX(lin2)(t->ka_mco2[b], 19, indm, ig, mf);
                    REAL abso3 = SFX(lin2)(t->ka_mo3[b], 19, indm, ig, mf);
                    REAL absn2o = SFX(lin2)(t->ka_mn2o[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                        tauself + taufor + adjcolco2 * absco2 + s->colo3[lay] * abso3 +
                                        s->coln2o[lay] * absn2o + s->colcfc12[lay] * t->cfc12_8[ig - 1] +
                                        s->colcfc22[lay] * t->cfc22adj[ig - 1];
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL absco2 = SFX(lin2)(t->kb_mco2[b], 19, indm, ig, mf);
                    REAL absn2o = SFX(lin2)(t->kb_mn2o[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colo3[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) +
                                        adjcolco2 * absco2 + s->coln2o[lay] * absn2o +
                                        s->colcfc12[lay] * t->cfc12_8[ig - 1] + s->colcfc22[lay] * t->cfc22adj[ig - 1];
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 9 (:1738-2001): h2o,ch4 | ch4; minor n2o -------------------------------------------------------------- */
        {
            const int b = 9, n = ngv[b], o = ngs[b - 1];
            REAL adjcoln2o = SFX(adjcol)(s->coln2o[lay], s->coldry[lay], CHI(4, jp + 1), (REAL)1.5, (REAL)0.5, (REAL)0.65);
            if (lower) {
                REAL refrat_planck_a = CHI(1, 9) / CHI(6, 9), refrat_m_a = CHI(1, 3) / CHI(6, 3);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2och4[lay], s->colch4[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2och4_1[lay], s->colch4[lay], 8, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->colh2o[lay], refrat_m_a, s->colch4[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colch4[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absn2o = SFX(minor2)(t->ka_mn2o[b], 9, sm.js, indm, ig, sm.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + adjcoln2o * absn2o;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL absn2o = SFX(lin2)(t->kb_mn2o[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colch4[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) +
                                        adjcoln2o * absn2o;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 10 (:2006-2072): h2o ------------------------------------------------------------------------------------ */
        {
            const int b = 10, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                        tauself + taufor;
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL taufor = TAUFOR(b);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) + taufor;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 11 (:2077-2160): h2o; minor o2 ---------------------------------------------------------------------------- */
        {
            const int b = 11, n = ngv[b], o = ngs[b - 1];
            REAL scaleo2 = s->colo2[lay] * s->scaleminor[lay];
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL tauo2 = scaleo2 * SFX(lin2)(t->ka_mo2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                        tauself + taufor + tauo2;
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL taufor = TAUFOR(b);
                    REAL tauo2 = scaleo2 * SFX(lin2)(t->kb_mo2[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colh2o[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11) +
                                        taufor + tauo2;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 12 (:2165-2345): h2o,co2 | nothing --------------------------------------------------------------------------- */
        {
            const int b = 12, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 10) / CHI(2, 10);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2oco2[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2oco2_1[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colco2[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                for (int ig = 1; ig <= n; ig++) { TAUG(lay, o + ig) = 0; PFR(lay, o + ig) = 0; }
            }
        }
        /* ---- band 13 (:2350-2585): h2o,n2o | o3 minor; minor co2, co ------------------------------------------------------------- */
        {
            const int b = 13, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 5) / CHI(4, 5), refrat_m_a = CHI(1, 1) / CHI(4, 1), refrat_m_a3 = CHI(1, 3) / CHI(4, 3);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2on2o[lay], s->coln2o[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2on2o_1[lay], s->coln2o[lay], 8, oneminus);
                SFX(spec_t) smco2 = SFX(spec)(s->colh2o[lay], refrat_m_a, s->coln2o[lay], 8, oneminus);
                REAL adjcolco2 = SFX(adjcol)(s->colco2[lay], s->coldry[lay], (REAL)3.55e-4, (REAL)3.0, (REAL)2.0, (REAL)0.68);
                SFX(spec_t) smco = SFX(spec)(s->colh2o[lay], refrat_m_a3, s->coln2o[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->coln2o[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL absco2 = SFX(minor2)(t->ka_mco2[b], 9, smco2.js, indm, ig, smco2.fs, mf);
                    REAL absco = SFX(minor2)(t->ka_mco[b], 9, smco.js, indm, ig, smco.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + adjcolco2 * absco2 + s->colco[lay] * absco;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                for (int ig = 1; ig <= n; ig++) {
                    REAL abso3 = SFX(lin2)(t->kb_mo3[b], 19, indm, ig, mf);
                    TAUG(lay, o + ig) = s->colo3[lay] * abso3;
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 14 (:2590-2653): co2 ----------------------------------------------------------------------------------------------- */
        {
            const int b = 14, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    TAUG(lay, o + ig) = s->colco2[lay] * SFX(major1)(t->absa[b], 65, ig, ind0, ind1, f00, f10, f01, f11) +
                                        tauself + taufor;
                    PFR(lay, o + ig) = t->fracrefa[b][ig - 1];
                }
            } else {
                int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    TAUG(lay, o + ig) = s->colco2[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11);
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
        /* ---- band 15 (:2658-2866): n2o,co2 | nothing; minor n2 ------------------------------------------------------------------------- */
        {
            const int b = 15, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(4, 1) / CHI(2, 1), refrat_m_a = refrat_planck_a;
                SFX(spec_t) sp = SFX(spec)(s->coln2o[lay], s->rat_n2oco2[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->coln2o[lay], s->rat_n2oco2_1[lay], s->colco2[lay], 8, oneminus);
                SFX(spec_t) sm = SFX(spec)(s->coln2o[lay], refrat_m_a, s->colco2[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->coln2o[lay], refrat_planck_a, s->colco2[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                REAL scalen2 = s->colbrd[lay] * s->scaleminor[lay];
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL taun2 = scalen2 * SFX(minor2)(t->ka_mn2[b], 9, sm.js, indm, ig, sm.fs, mf);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor + taun2;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                for (int ig = 1; ig <= n; ig++) { TAUG(lay, o + ig) = 0; PFR(lay, o + ig) = 0; }
            }
        }
        /* ---- band 16 (:2871-3126): h2o,ch4 | ch4 ------------------------------------------------------------------------------------------ */
        {
            const int b = 16, n = ngv[b], o = ngs[b - 1];
            if (lower) {
                REAL refrat_planck_a = CHI(1, 6) / CHI(6, 6);
                SFX(spec_t) sp = SFX(spec)(s->colh2o[lay], s->rat_h2och4[lay], s->colch4[lay], 8, oneminus);
                SFX(spec_t) sp1 = SFX(spec)(s->colh2o[lay], s->rat_h2och4_1[lay], s->colch4[lay], 8, oneminus);
                SFX(spec_t) spl = SFX(spec)(s->colh2o[lay], refrat_planck_a, s->colch4[lay], 8, oneminus);
                int ind0 = IND0A(9) + sp.js, ind1 = IND1A(9) + sp1.js;
                for (int ig = 1; ig <= n; ig++) {
                    REAL tauself = TAUSELF(b), taufor = TAUFOR(b);
                    REAL tau_major = SFX(major_a)(t->absa[b], 585, ig, ind0, sp, f00, f10);
                    REAL tau_major1 = SFX(major_a)(t->absa[b], 585, ig, ind1, sp1, f01, f11);
                    TAUG(lay, o + ig) = tau_major + tau_major1 + tauself + taufor;
                    PFR(lay, o + ig) = PLANCKA(b, n, spl.js, spl.fs);
                }
            } else {
                /* reference quirk kept: nspb(16) = 0 (rrtmg_lw_init.F90:195), so ind0 = ind1 = 1 (:3110-3111) */
                int ind0 = IND0B(0) + 1, ind1 = IND1B(0) + 1;
                for (int ig = 1; ig <= n; ig++) {
                    TAUG(lay, o + ig) = s->colch4[lay] * SFX(major1)(t->absb[b], 235, ig, ind0, ind1, f00, f10, f01, f11);
                    PFR(lay, o + ig) = t->fracrefb[b][ig - 1];
                }
            }
        }
    }
    /* addAerosols (:3130-3146) */
    for (int ig = 1; ig <= 140; ig++) {
        int ib = t->ngb[ig - 1];
        for (int lay = 1; lay <= nlay; lay++) TAUG(lay, ig) = TAUG(lay, ig) + F2(taua, nlay, lay, ib);
    }
}

/* ------------------------------------------------------------------------------------------------
 * rtrnmc for ONE column (LW/rrtmg_lw_rtrnmc.F90:27-390).
 * cloudy[lay] (1-based), taucmc F2(taucmc,nlay,lay,ig); flux outputs 0..nlay.
 * ---------------------------------------------------------------------------------------------- */
static void SFX(rtrnmc_col)(const SFX(colstate_t) * s, int dudTs, const REAL *semiss /*1..16*/, const REAL *taug,
                            const REAL *pfracs, const int *cloudy, const REAL *taucmc, REAL *totuflux,
                            REAL *totdflux, REAL *totuclfl, REAL *totdclfl, REAL *dtotuflux_dTs,
                            REAL *dtotuclfl_dTs, const int *band_output, REAL *olrb, REAL *dolrb_dTs)
{
    const SFX(lw_tables_t) *t = &SFX(T);
    const int nlay = s->nlay;
    const REAL wtdiff = (REAL)0.5, bpade = *t->bpade, tblint = (REAL)10000.0, fluxfac = *t->fluxfac;
    static const double a0[17] = {0, 1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
    static const double a1[17] = {0, 0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    static const double a2[17] = {0, 0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    REAL *agas = (REAL *)calloc((size_t)nlay + 2, sizeof(REAL)), *atot = (REAL *)calloc((size_t)nlay + 2, sizeof(REAL));
    REAL *bbugas = (REAL *)calloc((size_t)nlay + 2, sizeof(REAL)), *bbutot = (REAL *)calloc((size_t)nlay + 2, sizeof(REAL));
    int any_bo = 0;
    for (int ib = 0; ib < 16; ib++) any_bo |= band_output[ib];
    for (int l = 0; l <= nlay; l++) {
        totuflux[l] = totdflux[l] = totuclfl[l] = totdclfl[l] = 0;
        if (dudTs) dtotuflux_dTs[l] = dtotuclfl_dTs[l] = 0;
    }
    if (any_bo)
        for (int ib = 0; ib < 16; ib++) { olrb[ib] = 0; if (dudTs) dolrb_dTs[ib] = 0; }

    for (int ig = 1; ig <= 140; ig++) {
        int ibnd = t->ngb[ig - 1];
        REAL sumfac = wtdiff * t->delwave[ibnd - 1] * fluxfac;
        REAL secdiff;
        if (ibnd == 1 || ibnd == 4 || ibnd >= 10)
            secdiff = (REAL)1.66;
        else {
            secdiff = (REAL)a0[ibnd] + (REAL)a1[ibnd] * EXP((REAL)a2[ibnd] * s->pwvcm);
            if (secdiff > (REAL)1.80) secdiff = (REAL)1.80;
            else if (secdiff < (REAL)1.50) secdiff = (REAL)1.50;
        }
        REAL radld = 0, radclrd = 0;
        int diverge = 0;
        REAL deluflux = 0, deluderiv = 0;
        for (int lev = nlay; lev >= 1; lev--) {
            REAL plfrac = F2(pfracs, nlay, lev, ig);
            REAL blay = F2(s->planklay, 16, ibnd, lev);
            REAL dplankup = s->planklev[lev * 16 + ibnd - 1] - blay;
            REAL dplankdn = s->planklev[(lev - 1) * 16 + ibnd - 1] - blay;
            REAL odepth = secdiff * F2(taug, nlay, lev, ig);
            if (odepth < 0) odepth = 0;
            REAL tblind = odepth / (bpade + odepth);
            int itgas = (int)(tblint * tblind + (REAL)0.5);
            agas[lev] = (REAL)1. - t->exp_tbl[itgas];
            REAL tfacgas = t->tfn_tbl[itgas];
            REAL bbdgas = plfrac * (blay + tfacgas * dplankdn);
            bbugas[lev] = plfrac * (blay + tfacgas * dplankup);
            REAL tc = F2(taucmc, nlay, lev, ig);
            if (tc <= 0) {
                radld = radld + (bbdgas - radld) * agas[lev];
            } else {
                REAL odcld = secdiff * tc;
                odepth = t->tau_tbl[itgas];
                REAL odtot = odepth + odcld;
                tblind = odtot / (bpade + odtot);
                int ittot = (int)(tblint * tblind + (REAL)0.5);
                atot[lev] = (REAL)1. - t->exp_tbl[ittot];
                REAL tfactot = t->tfn_tbl[ittot];
                REAL bbdtot = plfrac * (blay + tfactot * dplankdn);
                bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                radld = radld + (bbdtot - radld) * atot[lev];
            }
            totdflux[lev - 1] = totdflux[lev - 1] + sumfac * radld;
            if (!diverge && cloudy[lev]) diverge = 1;
            if (diverge) radclrd = radclrd + (bbdgas - radclrd) * agas[lev];
            else radclrd = radld;
            totdclfl[lev - 1] = totdclfl[lev - 1] + sumfac * radclrd;
        }
        REAL rad0 = F2(pfracs, nlay, 1, ig) * s->plankbnd[ibnd];
        REAL d_rad0_dTs = 0;
        if (dudTs) d_rad0_dTs = F2(pfracs, nlay, 1, ig) * s->dplankbnd[ibnd];
        REAL reflect = (REAL)1. - semiss[ibnd];
        REAL radlu = rad0 + reflect * radld;
        REAL radclru = rad0 + reflect * radclrd;
        totuflux[0] = totuflux[0] + sumfac * radlu;
        totuclfl[0] = totuclfl[0] + sumfac * radclru;
        REAL d_radlu_dTs = 0, d_radclru_dTs = 0;
        if (dudTs) {
            d_radlu_dTs = d_rad0_dTs;
            d_radclru_dTs = d_rad0_dTs;
            dtotuflux_dTs[0] = dtotuflux_dTs[0] + sumfac * d_radlu_dTs;
            dtotuclfl_dTs[0] = dtotuclfl_dTs[0] + sumfac * d_radclru_dTs;
        }
        for (int lev = 1; lev <= nlay; lev++) {
            if (F2(taucmc, nlay, lev, ig) <= 0) {
                radlu = radlu + (bbugas[lev] - radlu) * agas[lev];
                if (dudTs) d_radlu_dTs = d_radlu_dTs - d_radlu_dTs * agas[lev];
            } else {
                radlu = radlu + (bbutot[lev] - radlu) * atot[lev];
                if (dudTs) d_radlu_dTs = d_radlu_dTs - d_radlu_dTs * atot[lev];
            }
            deluflux = sumfac * radlu;
            totuflux[lev] = totuflux[lev] + deluflux;
            if (diverge) radclru = radclru + (bbugas[lev] - radclru) * agas[lev];
            else radclru = radlu;
            totuclfl[lev] = totuclfl[lev] + sumfac * radclru;
            if (dudTs) {
                if (diverge) d_radclru_dTs = d_radclru_dTs - d_radclru_dTs * agas[lev];
                else d_radclru_dTs = d_radlu_dTs;
                deluderiv = sumfac * d_radlu_dTs;
                dtotuflux_dTs[lev] = dtotuflux_dTs[lev] + deluderiv;
                dtotuclfl_dTs[lev] = dtotuclfl_dTs[lev] + sumfac * d_radclru_dTs;
            }
        }
        if (band_output[ibnd - 1]) {
            olrb[ibnd - 1] = olrb[ibnd - 1] + deluflux;
            if (dudTs) dolrb_dTs[ibnd - 1] = dolrb_dTs[ibnd - 1] + deluderiv;
        }
    }
    free(agas); free(atot); free(bbugas); free(bbutot);
}

/* ------------------------------------------------------------------------------------------------
 * McICA generator (SH/cloud_subcol_gen.F90)
 * ---------------------------------------------------------------------------------------------- */
/* rng_kiss (:546-576): 32-bit wrap-around integer arithmetic, logical shifts */
static inline REAL SFX(rng_kiss)(uint32_t *s1, uint32_t *s2, uint32_t *s3, uint32_t *s4)
{
    *s1 = 69069u * *s1 + 1327217885u;
    uint32_t k = *s2;
    k ^= k << 13; k ^= k >> 17; k ^= k << 5;
    *s2 = k;
    *s3 = 18000u * (*s3 & 65535u) + (*s3 >> 16);
    *s4 = 30903u * (*s4 & 65535u) + (*s4 >> 16);
    int32_t kiss = (int32_t)(*s1 + *s2 + (*s3 << 16) + *s4);
    return (REAL)kiss * (REAL)2.328306e-10 + (REAL)0.5;
}
REAL SFX(oracle_kiss_to_real)(int32_t kiss) { return (REAL)kiss * (REAL)2.328306e-10 + (REAL)0.5; }
void SFX(oracle_kiss_stream)(const int32_t *seed, int n, REAL *out)
{
    uint32_t s1 = (uint32_t)seed[0], s2 = (uint32_t)seed[1], s3 = (uint32_t)seed[2], s4 = (uint32_t)seed[3];
    for (int i = 0; i < n; i++) out[i] = SFX(rng_kiss)(&s1, &s2, &s3, &s4);
}

/* correlation_length (:491-514) */
static REAL SFX(corr_length)(const REAL *am, int doy, REAL alat)
{
    const REAL r2d = (REAL)(180.0 / 3.14159265358979323846);
    REAL am3;
    if (doy > 181) am3 = (REAL)-4. * am[2] / (REAL)365. * (REAL)(doy - 272);
    else am3 = (REAL)4. * am[2] / (REAL)365. * (REAL)(doy - 91);
    REAL x = alat * r2d - am3;
    return (am[0] + am[1] * EXP(-(x * x) / (am[3] * am[3]))) * (REAL)1.e3;
}

/* zcw_lookup (SH/cloud_condensate_inhomogeneity.F90:86-124) */
static REAL SFX(zcw_lookup)(REAL cdf, REAL sigma_qcw)
{
    const REAL *xcw = SFX(T).xcw;
    if (!xcw) return 1;
    const int n1 = 1000, n2 = 140;
    REAL rind1 = cdf * (REAL)(n1 - 1) + (REAL)1.;
    int ind1 = (int)rind1; if (ind1 > n1 - 1) ind1 = n1 - 1; if (ind1 < 1) ind1 = 1;
    rind1 = rind1 - (REAL)ind1;
    REAL rind2 = (REAL)40. * sigma_qcw - (REAL)3.;
    int ind2 = (int)rind2; if (ind2 > n2 - 1) ind2 = n2 - 1; if (ind2 < 1) ind2 = 1;
    rind2 = rind2 - (REAL)ind2;
    return ((REAL)1.0 - rind1) * ((REAL)1.0 - rind2) * F2(xcw, n1, ind1, ind2) +
           ((REAL)1.0 - rind1) * rind2 * F2(xcw, n1, ind1, ind2 + 1) +
           rind1 * ((REAL)1.0 - rind2) * F2(xcw, n1, ind1 + 1, ind2) + rind1 * rind2 * F2(xcw, n1, ind1 + 1, ind2 + 1);
}
void SFX(oracle_zcw_lookup)(int n, const REAL *cdf, const REAL *sig, REAL *z)
{
    for (int i = 0; i < n; i++) z[i] = SFX(zcw_lookup)(cdf[i], sig[i]);
}

/* generate_stochastic_clouds for ONE column (:132-487).  Inputs 1-based per layer; outputs
 * F2(x, nlay, lay, isub).  surface_at_one as detected by the caller from play(1,1) > play(nlay,1). */
static void SFX(mcica_col)(int nsubcol, int nlay, int surface_at_one, const REAL *zmid, REAL alat, int doy,
                           const REAL *play, const REAL *cldfrac, const REAL *ciwp, const REAL *clwp, REAL cwp_tiny,
                           const int *so, int *cldy, REAL *ciwp_s, REAL *clwp_s)
{
    const int inhomo = SFX(T).xcw != NULL;
    REAL adl = SFX(corr_length)(SFX(T).aam, doy, alat), rdl = 0;
    if (inhomo) rdl = SFX(corr_length)(SFX(T).ram, doy, alat);
    size_t n = (size_t)nlay + 2;
    REAL *alpha = (REAL *)calloc(n, sizeof(REAL)), *rcorr = (REAL *)calloc(n, sizeof(REAL));
    REAL *sigma = (REAL *)calloc(n, sizeof(REAL)), *cdf1 = (REAL *)calloc(n, sizeof(REAL));
    REAL *cdf2 = (REAL *)calloc(n, sizeof(REAL)), *cdf3 = (REAL *)calloc(n, sizeof(REAL));
    for (int il = 2; il <= nlay; il++) alpha[il] = EXP(-FABS(zmid[il] - zmid[il - 1]) / adl);
    if (inhomo) {
        for (int il = 2; il <= nlay; il++) rcorr[il] = EXP(-FABS(zmid[il] - zmid[il - 1]) / rdl);
        for (int il = 1; il <= nlay; il++)
            sigma[il] = cldfrac[il] > (REAL)0.99 ? (REAL)0.5 : (cldfrac[il] > (REAL)0.9 ? (REAL)0.71 : (REAL)1.0);
    }
    /* seeds (:375-400) */
    REAL pseed[5];
    for (int k = 1; k <= 4; k++) pseed[k] = (surface_at_one ? play[k] : play[nlay + 1 - k]) * (REAL)100.;
    const int32_t maximo = 2147483647 - 1;
    uint32_t sd[5];
    for (int k = 1; k <= 4; k++) {
        REAL ps = pseed[so[k - 1]];
        sd[k] = (uint32_t)(int32_t)((ps - (REAL)(int32_t)ps) * (REAL)maximo + (REAL)1);
    }
    uint32_t s1 = sd[1], s2 = sd[2], s3 = sd[3], s4 = sd[4];
    for (int is = 1; is <= nsubcol; is++) {
        for (int il = 1; il <= nlay; il++) {
            cdf1[il] = SFX(rng_kiss)(&s1, &s2, &s3, &s4);
            cdf2[il] = SFX(rng_kiss)(&s1, &s2, &s3, &s4);
        }
        for (int il = 2; il <= nlay; il++)
            if (cdf2[il] < alpha[il]) cdf1[il] = cdf1[il - 1];
        if (inhomo) {
            for (int il = 1; il <= nlay; il++) {
                cdf2[il] = SFX(rng_kiss)(&s1, &s2, &s3, &s4);
                cdf3[il] = SFX(rng_kiss)(&s1, &s2, &s3, &s4);
            }
            for (int il = 2; il <= nlay; il++)
                if (cdf2[il] < rcorr[il]) cdf3[il] = cdf3[il - 1];
        }
        for (int il = 1; il <= nlay; il++) {
            if (cdf1[il] >= (REAL)1. - cldfrac[il]) {
                int c = 1;
                REAL ci, cl;
                if (inhomo) {
                    REAL zcw = SFX(zcw_lookup)(cdf3[il], sigma[il]);
                    ci = ciwp[il] * zcw; cl = clwp[il] * zcw;
                } else { ci = ciwp[il]; cl = clwp[il]; }
                int cin = (ci <= cwp_tiny), cln = (cl <= cwp_tiny);
                if (cin) ci = 0;
                if (cln) cl = 0;
                if (cin && cln) c = 0;
                F2(cldy, nlay, il, is) = c; F2(ciwp_s, nlay, il, is) = ci; F2(clwp_s, nlay, il, is) = cl;
            } else {
                F2(cldy, nlay, il, is) = 0; F2(ciwp_s, nlay, il, is) = 0; F2(clwp_s, nlay, il, is) = 0;
            }
        }
    }
    free(alpha); free(rcorr); free(sigma); free(cdf1); free(cdf2); free(cdf3);
}

/* clearCounts_threeBand for ONE column (:611-769); cnt[0..3] = whole, high, mid, low */
static int SFX(clearcounts_col)(int nsubcol, int nlay, int cloudLM, int cloudMH, const int *cldy, int *cnt)
{
    int lo0, lo1, mi0, mi1, hi0, hi1;
    if (cloudLM < cloudMH) { lo0 = 1; lo1 = cloudLM; mi0 = cloudLM + 1; mi1 = cloudMH; hi0 = cloudMH + 1; hi1 = nlay; }
    else if (cloudLM > cloudMH) { hi0 = 1; hi1 = cloudMH - 1; mi0 = cloudMH; mi1 = cloudLM - 1; lo0 = cloudLM; lo1 = nlay; }
    else return 1;
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    for (int is = 1; is <= nsubcol; is++) {
        int f = 0;
        for (int il = 1; il <= nlay; il++) if (F2(cldy, nlay, il, is)) { f = 1; break; }
        if (!f) cnt[0]++;
        f = 0; for (int il = hi0; il <= hi1; il++) if (F2(cldy, nlay, il, is)) { f = 1; break; }
        if (!f) cnt[1]++;
        f = 0; for (int il = mi0; il <= mi1; il++) if (F2(cldy, nlay, il, is)) { f = 1; break; }
        if (!f) cnt[2]++;
        f = 0; for (int il = lo0; il <= lo1; il++) if (F2(cldy, nlay, il, is)) { f = 1; break; }
        if (!f) cnt[3]++;
    }
    return 0;
}

/* cldprmc for ONE column (LW/rrtmg_lw_cldprmc.F90:24-385).  Returns 0 or an error code for the
 * radius-extrapolation / invalid flag `error stop`s. */
static int SFX(cldprmc_col)(int nlay, const int *cldymc, const REAL *ciwpmc, const REAL *clwpmc, const REAL *reice,
                            const REAL *reliq, int iceflag, int liqflag, REAL *taucmc, int *cloudy)
{
    const SFX(lw_tables_t) *t = &SFX(T);
    if (iceflag < 0 || iceflag > 4) return 10;
    if (liqflag != 1) return 11;
    for (int il = 1; il <= nlay; il++) {
        cloudy[il] = 0;
        for (int ig = 1; ig <= 140; ig++) { F2(taucmc, nlay, il, ig) = 0; if (F2(cldymc, nlay, il, ig)) cloudy[il] = 1; }
    }
    for (int il = 1; il <= nlay; il++) {
        if (!cloudy[il]) continue;
        REAL factor = 0, fint = 0, abscoice0 = 0;
        int index = 0, n1 = 0;
        const REAL *tab = NULL;
        if (iceflag == 0) abscoice0 = t->absice0[0] + t->absice0[1] / reice[il];
        else if (iceflag >= 2) {
            int nmax;
            if (iceflag == 2) { factor = (reice[il] - (REAL)2.) / (REAL)3.; nmax = 43; tab = t->absice2; }
            else if (iceflag == 3) { factor = (reice[il] - (REAL)2.) / (REAL)3.; nmax = 46; tab = t->absice3; }
            else { factor = reice[il]; nmax = 200; tab = t->absice4; }
            n1 = nmax;
            index = (int)factor;
            if (index >= nmax) { if (index == nmax) index = nmax - 1; else return 20 + iceflag; }
            else if (index <= 0) { if (index == 0) index = 1; else return 30 + iceflag; }
            fint = factor - (REAL)index;
        }
        for (int ig = 1; ig <= 140; ig++) {
            if (F2(cldymc, nlay, il, ig) && F2(ciwpmc, nlay, il, ig) > 0) {
                REAL abscoice;
                int ib = t->ngb[ig - 1];
                if (iceflag == 0) abscoice = abscoice0;
                else if (iceflag == 1) { int i1 = t->ice1b[ib - 1]; abscoice = F2(t->absice1, 2, 1, i1) + F2(t->absice1, 2, 2, i1) / reice[il]; }
                else abscoice = F2(tab, n1, index, ib) + fint * (F2(tab, n1, index + 1, ib) - F2(tab, n1, index, ib));
                F2(taucmc, nlay, il, ig) = F2(ciwpmc, nlay, il, ig) * abscoice;
            }
        }
        /* liquid, liqflag 1 (:318-360) */
        factor = reliq[il] - (REAL)1.5;
        index = (int)factor;
        if (index >= 58) { if (index == 58) index = 57; else return 41; }
        else if (index <= 0) { if (index == 0) index = 1; else return 42; }
        fint = factor - (REAL)index;
        for (int ig = 1; ig <= 140; ig++) {
            if (F2(cldymc, nlay, il, ig) && F2(clwpmc, nlay, il, ig) > 0) {
                int ib = t->ngb[ig - 1];
                REAL abscoliq = F2(t->absliq1, 58, index, ib) + fint * (F2(t->absliq1, 58, index + 1, ib) - F2(t->absliq1, 58, index, ib));
                F2(taucmc, nlay, il, ig) = F2(taucmc, nlay, il, ig) + F2(clwpmc, nlay, il, ig) * abscoliq;
            }
        }
    }
    /* refine to optically cloudy (:371-383) */
    for (int il = 1; il <= nlay; il++) {
        if (cloudy[il]) {
            int any = 0;
            for (int ig = 1; ig <= 140; ig++) if (F2(taucmc, nlay, il, ig) > 0) { any = 1; break; }
            if (!any) cloudy[il] = 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * public entry points
 * ---------------------------------------------------------------------------------------------- */
#define API2(a, lay, col) ((a)[(size_t)((lay) - 1) * ncol + (col)])      /* Fortran (ncol,nlay), lay 1-based, col 0-based */
#define APIL(a, lev, col) ((a)[(size_t)(lev) * ncol + (col)])             /* Fortran (ncol,0:nlay) */

static void SFX(gather1)(const REAL *a, int ncol, int nlay, int col, REAL *out)
{
    out[0] = 0;
    for (int l = 1; l <= nlay; l++) out[l] = API2(a, l, col);
}

/* McICA stand-alone: API-layout profile inputs; outputs Fortran (nlay,nsubcol,ncol) */
int SFX(oracle_mcica)(int ncol, int nsubcol, int nlay, const REAL *zmid, const REAL *alat, int doy, const REAL *play,
                      const REAL *cldfrac, const REAL *ciwp, const REAL *clwp, REAL cwp_tiny, const int *seed_order,
                      int *cldy, REAL *ciwp_s, REAL *clwp_s)
{
    size_t n = (size_t)nlay + 2;
    REAL *z = malloc(n * sizeof(REAL)), *p = malloc(n * sizeof(REAL)), *f = malloc(n * sizeof(REAL));
    REAL *ci = malloc(n * sizeof(REAL)), *cl = malloc(n * sizeof(REAL));
    int hit[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 4; k++) {
        if (seed_order[k] < 1 || seed_order[k] > 4) return 2;
        if (hit[k + 1]) return 3; /* the reference tests hit(n), n = position (cloud_subcol_gen.F90:286) */
        hit[k + 1] = 1;
    }
    int surface_at_one = API2(play, 1, 0) > API2(play, nlay, 0);
    for (int c = 0; c < ncol; c++) {
        SFX(gather1)(zmid, ncol, nlay, c, z); SFX(gather1)(play, ncol, nlay, c, p); SFX(gather1)(cldfrac, ncol, nlay, c, f);
        SFX(gather1)(ciwp, ncol, nlay, c, ci); SFX(gather1)(clwp, ncol, nlay, c, cl);
        size_t o = (size_t)c * nsubcol * nlay;
        SFX(mcica_col)(nsubcol, nlay, surface_at_one, z, alat[c], doy, p, f, ci, cl, cwp_tiny, seed_order, cldy + o,
                       ciwp_s + o, clwp_s + o);
    }
    free(z); free(p); free(f); free(ci); free(cl);
    return 0;
}

int SFX(oracle_clearcounts)(int ncol, int nsubcol, int nlay, int cloudLM, int cloudMH, const int *cldy, int *cnt /*(4,ncol)*/)
{
    for (int c = 0; c < ncol; c++)
        if (SFX(clearcounts_col)(nsubcol, nlay, cloudLM, cloudMH, cldy + (size_t)c * nsubcol * nlay, cnt + 4 * c)) return 1;
    return 0;
}

/* rrtmg_lw (LW/rrtmg_lw_rad.F90:15-344 + rrtmg_lw_part :348-610), API layouts throughout.
 * Optional intermediates (may be NULL): o_taug/o_pfracs/o_taucmc Fortran (nlay,140,ncol).
 * Returns 0 ok, 100+k negative input #k (:209-318), 1 pressure misordering, 2x cldprmc errors. */
int SFX(oracle_rrtmg_lw)(int ncol, int nlay, int dudTs, const REAL *play, const REAL *plev, const REAL *tlay,
                         const REAL *tlev, const REAL *tsfc, const REAL *emis, const REAL *h2ovmr, const REAL *o3vmr,
                         const REAL *co2vmr, const REAL *ch4vmr, const REAL *n2ovmr, const REAL *o2vmr,
                         const REAL *cfc11vmr, const REAL *cfc12vmr, const REAL *cfc22vmr, const REAL *ccl4vmr,
                         const REAL *cldf, const REAL *ciwp, const REAL *clwp, const REAL *rei, const REAL *rel,
                         int iceflglw, int liqflglw, const REAL *tauaer, const REAL *zm, const REAL *alat, int dyofyr,
                         int cloudLM, int cloudMH, int *clearCounts, REAL *uflx, REAL *dflx, REAL *uflxc, REAL *dflxc,
                         REAL *duflx_dTs, REAL *duflxc_dTs, const int *band_output, REAL *olrb, REAL *dolrb_dTs,
                         REAL *o_taug, REAL *o_pfracs, REAL *o_taucmc)
{
    /* input asserts */
    const REAL *chk2[] = {play, tlay, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr,
                          cldf, ciwp, clwp, rei, rel};
    for (size_t k = 0; k < sizeof(chk2) / sizeof(chk2[0]); k++)
        for (size_t i = 0; i < (size_t)ncol * nlay; i++) if (chk2[k][i] < 0) return 100 + (int)k;
    for (size_t i = 0; i < (size_t)ncol * (nlay + 1); i++) if (plev[i] < 0 || tlev[i] < 0) return 120;
    for (int i = 0; i < ncol; i++) if (tsfc[i] < 0) return 121;
    for (int i = 0; i < ncol * 16; i++) if (emis[i] < 0) return 122;
    for (size_t i = 0; i < (size_t)ncol * nlay * 16; i++) if (tauaer[i] < 0) return 123;

    size_t n = (size_t)nlay + 2;
    SFX(colstate_t) s;
    SFX(cs_alloc)(&s, nlay);
    REAL *v[22];
    for (int k = 0; k < 22; k++) v[k] = (REAL *)calloc(n, sizeof(REAL));
    REAL *pz = v[0], *tz = v[1], *pav = v[2], *tav = v[3], *covmr = v[14];
    REAL *taug = (REAL *)calloc((size_t)nlay * 140, sizeof(REAL)), *pfr = (REAL *)calloc((size_t)nlay * 140, sizeof(REAL));
    REAL *taucmc = (REAL *)calloc((size_t)nlay * 140, sizeof(REAL));
    REAL *ciwpmc = (REAL *)calloc((size_t)nlay * 140, sizeof(REAL)), *clwpmc = (REAL *)calloc((size_t)nlay * 140, sizeof(REAL));
    int *cldymc = (int *)calloc((size_t)nlay * 140, sizeof(int)), *cloudy = (int *)calloc(n, sizeof(int));
    REAL *taua = (REAL *)calloc((size_t)nlay * 16, sizeof(REAL));
    REAL *fl[6];
    for (int k = 0; k < 6; k++) fl[k] = (REAL *)calloc(n, sizeof(REAL));
    static const int so[4] = {1, 2, 3, 4};
    int rc = 0;
    int surface_at_one = API2(play, 1, 0) > API2(play, nlay, 0);   /* per-partition in the reference; RRTMG is always bottom-up */
    for (int c = 0; c < ncol && !rc; c++) {
        REAL semiss[17], olr[16], dolr[16];
        for (int ib = 1; ib <= 16; ib++) semiss[ib] = emis[(size_t)(ib - 1) * ncol + c];
        for (int l = 0; l <= nlay; l++) { pz[l] = APIL(plev, l, c); tz[l] = APIL(tlev, l, c); }
        SFX(gather1)(play, ncol, nlay, c, pav); SFX(gather1)(tlay, ncol, nlay, c, tav);
        const REAL *gin[] = {h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, cldf, ciwp, clwp, rei, rel, zm};
        REAL *gout[] = {v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[15], v[16], v[17], v[18], v[19], v[20]};
        for (int k = 0; k < 16; k++) SFX(gather1)(gin[k], ncol, nlay, c, gout[k]);
        for (int ib = 1; ib <= 16; ib++)
            for (int l = 1; l <= nlay; l++) F2(taua, nlay, l, ib) = tauaer[((size_t)(ib - 1) * nlay + (l - 1)) * ncol + c];
        /* McICA (:541-555) */
        SFX(mcica_col)(140, nlay, surface_at_one, v[20], alat[c], dyofyr, pav, v[15], v[16], v[17], (REAL)1.e-20, so, cldymc, ciwpmc, clwpmc);
        int cnt[4];
        if (SFX(clearcounts_col)(140, nlay, cloudLM, cloudMH, cldymc, cnt)) { rc = 5; break; }
        for (int k = 0; k < 4; k++) clearCounts[(size_t)k * ncol + c] = cnt[k];
        rc = SFX(cldprmc_col)(nlay, cldymc, ciwpmc, clwpmc, v[18], v[19], iceflglw, liqflglw, taucmc, cloudy);
        if (rc) break;
        rc = SFX(setcoef_col)(&s, dudTs, pav, tav, pz, tz, tsfc[c], semiss, v[4], v[5], v[6], v[7], v[8], v[9], covmr, v[10], v[11], v[12], v[13]);
        if (rc) break;
        SFX(taumol_col)(&s, taua, taug, pfr);
        SFX(rtrnmc_col)(&s, dudTs, semiss, taug, pfr, cloudy, taucmc, fl[0], fl[1], fl[2], fl[3], fl[4], fl[5], band_output, olr, dolr);
        for (int l = 0; l <= nlay; l++) {
            APIL(uflx, l, c) = fl[0][l]; APIL(dflx, l, c) = fl[1][l]; APIL(uflxc, l, c) = fl[2][l]; APIL(dflxc, l, c) = fl[3][l];
            if (dudTs) { APIL(duflx_dTs, l, c) = fl[4][l]; APIL(duflxc_dTs, l, c) = fl[5][l]; }
        }
        for (int ib = 0; ib < 16; ib++)
            if (band_output[ib]) { olrb[(size_t)c * 16 + ib] = olr[ib]; if (dudTs) dolrb_dTs[(size_t)c * 16 + ib] = dolr[ib]; }
        if (o_taug) memcpy(o_taug + (size_t)c * 140 * nlay, taug, sizeof(REAL) * 140 * nlay);
        if (o_pfracs) memcpy(o_pfracs + (size_t)c * 140 * nlay, pfr, sizeof(REAL) * 140 * nlay);
        if (o_taucmc) memcpy(o_taucmc + (size_t)c * 140 * nlay, taucmc, sizeof(REAL) * 140 * nlay);
    }
    for (int k = 0; k < 22; k++) free(v[k]);
    for (int k = 0; k < 6; k++) free(fl[k]);
    free(taug); free(pfr); free(taucmc); free(ciwpmc); free(clwpmc); free(cldymc); free(cloudy); free(taua);
    SFX(cs_free)(&s);
    return rc;
}

#undef TAUG
#undef PFR
#undef IND0A
#undef IND1A
#undef IND0B
#undef IND1B
#undef TAUSELF
#undef TAUFOR
#undef PLANCKA
#undef PLANCKB
#undef CHI
#undef API2
#undef APIL
