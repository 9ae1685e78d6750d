/* oracle/sw_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (the product never links or calls this).
 *
 * Plain-C restatement of the reference's RRTMG_SW column solver, included twice by sw_oracle.c
 * (REAL = float -> *_f32, REAL = double -> *_f64).  Citations: SW = /root/reference/GEOSsolar_GridComp/
 * RRTMG/rrtmg_sw/gcm_model/src.
 *
 * PINNING STATUS
 *   setcoef_sw, taumol_sw (14 bands, solar source), cldprmc_sw : PINNED -- bit-identical to the reference's
 *       own Fortran (oracle/_ref builds those files; tests/test_oracle_sw.py) and to tests/golden/sw_*.npz.
 *   NRLSSI2 adjust_solcyc_amplitudes, interpolate_indices, the isolvar = 1 cycle means : PINNED -- bit-identical to the
 *       reference's NRLSSI2.F90 (no ESMF/MAPL dependency; oracle/_ref builds it) over solcycfrac x indsolvar grids.
 *   reftra_sw, vrtqdr_sw, spcvmc_sw band integration, rrtmg_sw driver (albedo->band map, solar variability,
 *       normFlx): "parity unpinned" -- rrtmg_sw_spcvmc.F90 and rrtmg_sw_rad.F90 `use ESMF`/`use MAPL`, which
 *       this image lacks, so the reference cannot be built for them here and the reference has no golden
 *       vectors.  They are restated line by line and checked only by invariants (energy conservation,
 *       clear == total for cloud-free columns, direct-beam Beer law, flux positivity).
 */

typedef struct {
    const REAL *preflog, *tref, *oneminus, *grav, *avogad, *rrsw_scon;
    const REAL *extliq1, *ssaliq1, *asyliq1, *extice2, *ssaice2, *asyice2, *extice3, *ssaice3, *asyice3, *fdlice3,
        *extice4, *ssaice4, *asyice4, *abari, *bbari, *cbari, *dbari, *ebari, *fbari;
    const REAL *Iint, *Fint, *Sint, *Mg_avg, *Mg_0, *SB_avg, *SB_0, *mgavgcyc, *sbavgcyc;
    const int *ngb, *icxa;
    /* per band, index 16..29 */
    const REAL *absa[30], *absb[30], *selfref[30], *forref[30], *sfluxref[30], *irradnce[30], *facbrght[30], *snsptdrk[30],
        *rayl[30];
    const REAL *absch4, *abso3a24, *abso3b24, *rayla24, *raylb24, *abso3a25, *abso3b25, *absh2o, *absco2;
} SFX(sw_tables_t);
static SFX(sw_tables_t) SFX(S);

int SFX(oracle_sw_set_table)(const char *name, const void *p)
{
    SFX(sw_tables_t) *t = &SFX(S);
#define SET(nm, field) if (!strcmp(name, nm)) { t->field = p; return 0; }
    SET("preflog", preflog) SET("tref", tref) SET("oneminus", oneminus) SET("grav", grav) SET("avogad", avogad)
    SET("rrsw_scon", rrsw_scon)
    SET("extliq1", extliq1) SET("ssaliq1", ssaliq1) SET("asyliq1", asyliq1) SET("extice2", extice2) SET("ssaice2", ssaice2)
    SET("asyice2", asyice2) SET("extice3", extice3) SET("ssaice3", ssaice3) SET("asyice3", asyice3) SET("fdlice3", fdlice3)
    SET("extice4", extice4) SET("ssaice4", ssaice4) SET("asyice4", asyice4) SET("abari", abari) SET("bbari", bbari)
    SET("cbari", cbari) SET("dbari", dbari) SET("ebari", ebari) SET("fbari", fbari)
    SET("Iint", Iint) SET("Fint", Fint) SET("Sint", Sint) SET("Mg_avg", Mg_avg) SET("Mg_0", Mg_0) SET("SB_avg", SB_avg)
    SET("SB_0", SB_0) SET("mgavgcyc", mgavgcyc) SET("sbavgcyc", sbavgcyc) SET("ngb", ngb) SET("icxa", icxa)
    SET("b20_absch4", absch4) SET("b24_abso3a", abso3a24) SET("b24_abso3b", abso3b24) SET("b24_rayla", rayla24)
    SET("b24_raylb", raylb24) SET("b25_abso3a", abso3a25) SET("b25_abso3b", abso3b25) SET("b29_absh2o", absh2o)
    SET("b29_absco2", absco2)
#undef SET
    if (name[0] == 'b' && name[3] == '_') {
        int b = (name[1] - '0') * 10 + (name[2] - '0');
        const char *s = name + 4;
        if (b < 16 || b > 29) return -1;
#define SETB(nm, field) if (!strcmp(s, nm)) { t->field[b] = p; return 0; }
        SETB("absa", absa) SETB("absb", absb) SETB("selfref", selfref) SETB("forref", forref) SETB("sfluxref", sfluxref)
        SETB("irradnce", irradnce) SETB("facbrght", facbrght) SETB("snsptdrk", snsptdrk) SETB("rayl", rayl)
#undef SETB
    }
    return -1;
}

#define NGSW 112
static const int SFX(sw_ng)[30] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};
static const int SFX(sw_ngs)[30] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 0, 6, 18, 26, 34, 44, 54, 56, 66, 74, 80, 86, 94, 100};   /* first g-point (0-based) */

/* per-column state of setcoef_sw (SW/rrtmg_sw_setcoef.F90:23-241), 1-based layers */
typedef struct {
    int nlay, laytrop;
    REAL *colh2o, *colco2, *colo3, *colch4, *colo2, *colmol, *coldry;
    REAL *fac00, *fac01, *fac10, *fac11, *selffac, *selffrac, *forfac, *forfrac;
    int *jp, *jt, *jt1, *indself, *indfor;
} SFX(swcol_t);

static void SFX(swcol_alloc)(SFX(swcol_t) * s, int nlay)
{
    size_t n = (size_t)nlay + 3;
    REAL **ra[] = {&s->colh2o, &s->colco2, &s->colo3, &s->colch4, &s->colo2, &s->colmol, &s->coldry, &s->fac00, &s->fac01,
                   &s->fac10, &s->fac11, &s->selffac, &s->selffrac, &s->forfac, &s->forfrac};
    int **ia[] = {&s->jp, &s->jt, &s->jt1, &s->indself, &s->indfor};
    for (size_t i = 0; i < sizeof(ra) / sizeof(ra[0]); i++) *ra[i] = (REAL *)calloc(n, sizeof(REAL));
    for (size_t i = 0; i < sizeof(ia) / sizeof(ia[0]); i++) *ia[i] = (int *)calloc(n, sizeof(int));
    s->nlay = nlay;
}
static void SFX(swcol_free)(SFX(swcol_t) * s)
{
    REAL *ra[] = {s->colh2o, s->colco2, s->colo3, s->colch4, s->colo2, s->colmol, s->coldry, s->fac00, s->fac01, s->fac10,
                  s->fac11, s->selffac, s->selffrac, s->forfac, s->forfrac};
    int *ia[] = {s->jp, s->jt, s->jt1, s->indself, s->indfor};
    for (size_t i = 0; i < sizeof(ra) / sizeof(ra[0]); i++) free(ra[i]);
    for (size_t i = 0; i < sizeof(ia) / sizeof(ia[0]); i++) free(ia[i]);
}

/* dry-air and gas column amounts (SW/rrtmg_sw_rad.F90:1370-1387) followed by setcoef_sw (:89-241).
 * pavel/tavel/vmr 1-based per layer, plev[1..nlay+1] (1 = surface). */
static void SFX(sw_setcoef_col)(SFX(swcol_t) * s, const REAL *pavel, const REAL *tavel, const REAL *plev, const REAL *h2ovmr,
                                const REAL *co2vmr, const REAL *o3vmr, const REAL *ch4vmr, const REAL *o2vmr)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const int nlay = s->nlay;
    const REAL amd = (REAL)28.9660, amw = (REAL)18.0160, stpfac = (REAL)296. / (REAL)1013.;
    const REAL grav = *t->grav, avogad = *t->avogad;
    s->laytrop = 0;
    for (int lay = 1; lay <= nlay; lay++) {
        REAL h = h2ovmr[lay];
        s->coldry[lay] = (plev[lay] - plev[lay + 1]) * (REAL)1.e3 * avogad /
                         ((REAL)1.e2 * grav * (((REAL)1. - h) * amd + h * amw) * ((REAL)1. + h));
        s->colh2o[lay] = s->coldry[lay] * h2ovmr[lay];
        s->colco2[lay] = s->coldry[lay] * co2vmr[lay];
        s->colo3[lay] = s->coldry[lay] * o3vmr[lay];
        s->colch4[lay] = s->coldry[lay] * ch4vmr[lay];
        s->colo2[lay] = s->coldry[lay] * o2vmr[lay];
        if (LOG(pavel[lay]) >= (REAL)4.56) s->laytrop += 1;
    }
    for (int lay = 1; lay <= nlay; lay++) {
        REAL plog = LOG(pavel[lay]);
        int jp = (int)((REAL)36. - (REAL)5 * (plog + (REAL)0.04));
        if (jp < 1) jp = 1; else if (jp > 58) jp = 58;
        s->jp[lay] = jp;
        int jp1 = jp + 1;
        REAL fp = (REAL)5. * (t->preflog[jp - 1] - plog);
        int jt = (int)((REAL)3. + (tavel[lay] - t->tref[jp - 1]) / (REAL)15.);
        if (jt < 1) jt = 1; else if (jt > 4) jt = 4;
        s->jt[lay] = jt;
        REAL ft = ((tavel[lay] - t->tref[jp - 1]) / (REAL)15.) - (REAL)(jt - 3);
        int jt1 = (int)((REAL)3. + (tavel[lay] - t->tref[jp1 - 1]) / (REAL)15.);
        if (jt1 < 1) jt1 = 1; else if (jt1 > 4) jt1 = 4;
        s->jt1[lay] = jt1;
        REAL ft1 = ((tavel[lay] - t->tref[jp1 - 1]) / (REAL)15.) - (REAL)(jt1 - 3);
        REAL water = s->colh2o[lay] / s->coldry[lay];
        REAL scalefac = pavel[lay] * stpfac / tavel[lay];
        REAL factor;
        if (plog <= (REAL)4.56) {
            s->forfac[lay] = scalefac / ((REAL)1. + water);
            factor = (tavel[lay] - (REAL)188.) / (REAL)36.;
            s->indfor[lay] = 3;
            s->forfrac[lay] = factor - (REAL)1.;
            s->selffac[lay] = 0; s->selffrac[lay] = 0; s->indself[lay] = 0;
        } else {
            s->forfac[lay] = scalefac / ((REAL)1. + water);
            factor = ((REAL)332. - tavel[lay]) / (REAL)36.;
            int i = (int)factor; if (i < 1) i = 1; if (i > 2) i = 2;
            s->indfor[lay] = i;
            s->forfrac[lay] = factor - (REAL)i;
            s->selffac[lay] = water * s->forfac[lay];
            factor = (tavel[lay] - (REAL)188.) / (REAL)7.2;
            i = (int)factor - 7; if (i < 1) i = 1; if (i > 9) i = 9;
            s->indself[lay] = i;
            s->selffrac[lay] = factor - (REAL)(i + 7);
        }
        s->colh2o[lay] = (REAL)1.e-20 * s->colh2o[lay];
        s->colco2[lay] = (REAL)1.e-20 * s->colco2[lay];
        s->colo3[lay] = (REAL)1.e-20 * s->colo3[lay];
        s->colch4[lay] = (REAL)1.e-20 * s->colch4[lay];
        s->colo2[lay] = (REAL)1.e-20 * s->colo2[lay];
        s->colmol[lay] = (REAL)1.e-20 * s->coldry[lay] + s->colh2o[lay];
        if (s->colco2[lay] == 0) s->colco2[lay] = (REAL)1.e-32 * s->coldry[lay];
        if (s->colch4[lay] == 0) s->colch4[lay] = (REAL)1.e-32 * s->coldry[lay];
        if (s->colo2[lay] == 0) s->colo2[lay] = (REAL)1.e-32 * s->coldry[lay];
        REAL compfp = (REAL)1. - fp;
        s->fac10[lay] = compfp * ft;
        s->fac00[lay] = compfp * ((REAL)1. - ft);
        s->fac11[lay] = fp * ft1;
        s->fac01[lay] = fp * ((REAL)1. - ft1);
    }
}

#define LIN1(tab, n1, i, ig, f) (F2(tab, n1, i, ig) + (f) * (F2(tab, n1, (i) + 1, ig) - F2(tab, n1, i, ig)))          /* LIN2_ARG1 */
#define LIN2(tab, n1, ig, j, f) (F2(tab, n1, ig, j) + (f) * (F2(tab, n1, ig, (j) + 1) - F2(tab, n1, ig, j)))          /* LIN2_ARG2 */

typedef struct { REAL speccomb, fs; int js; } SFX(swspec_t);
static inline SFX(swspec_t) SFX(swspec)(REAL cola, REAL strrat, REAL colb, REAL mult, REAL oneminus)
{
    SFX(swspec_t) r;
    r.speccomb = cola + strrat * colb;
    REAL specparm = cola / r.speccomb;
    if (specparm >= oneminus) specparm = oneminus;
    REAL specmult = mult * specparm;
    r.js = 1 + (int)specmult;
    r.fs = FMOD(specmult, (REAL)1.);
    return r;
}
/* 8-point (species, T, p) interpolation of a binary band: off = 9 (lower, nspa = 9) or 5 (upper, nspb = 5) */
static inline REAL SFX(sw_major2)(const REAL *tab, int n1, int ig, int ind0, int ind1, int off, REAL fs, REAL f00, REAL f10,
                                  REAL f01, REAL f11)
{
    REAL fac000 = ((REAL)1. - fs) * f00, fac010 = ((REAL)1. - fs) * f10, fac100 = fs * f00, fac110 = fs * f10;
    REAL fac001 = ((REAL)1. - fs) * f01, fac011 = ((REAL)1. - fs) * f11, fac101 = fs * f01, fac111 = fs * f11;
    return fac000 * F2(tab, n1, ind0, ig) + fac100 * F2(tab, n1, ind0 + 1, ig) + fac010 * F2(tab, n1, ind0 + off, ig) +
           fac110 * F2(tab, n1, ind0 + off + 1, ig) + fac001 * F2(tab, n1, ind1, ig) + fac101 * F2(tab, n1, ind1 + 1, ig) +
           fac011 * F2(tab, n1, ind1 + off, ig) + fac111 * F2(tab, n1, ind1 + off + 1, ig);
}
static inline REAL SFX(sw_major1)(const REAL *tab, int n1, int ig, int ind0, int ind1, REAL f00, REAL f10, REAL f01, REAL f11)
{
    return f00 * F2(tab, n1, ind0, ig) + f10 * F2(tab, n1, ind0 + 1, ig) + f01 * F2(tab, n1, ind1, ig) + f11 * F2(tab, n1, ind1 + 1, ig);
}

/* solar source of one band's g-points (e.g. SW/rrtmg_sw_taumol.F90:325-347 simple, :475-526 at laysolfr).
 * nsrc = 1: tables (ng); nsrc > 1: tables (ng,nsrc) interpolated at (js,fs). */
static void SFX(sw_source)(int b, int nsrc, int js, REAL fs, int isolvar, const REAL *svar, const REAL *svar_bnd /*(29,3) F*/,
                           REAL *ssi, REAL *sfluxzen)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const int n = SFX(sw_ng)[b], o = SFX(sw_ngs)[b];
    for (int ig = 1; ig <= n; ig++) {
        REAL sf, fb, sd, ir;
        if (nsrc == 1) {
            sf = t->sfluxref[b][ig - 1]; fb = t->facbrght[b][ig - 1]; sd = t->snsptdrk[b][ig - 1]; ir = t->irradnce[b][ig - 1];
        } else {
            sf = LIN2(t->sfluxref[b], n, ig, js, fs); fb = LIN2(t->facbrght[b], n, ig, js, fs);
            sd = LIN2(t->snsptdrk[b], n, ig, js, fs); ir = LIN2(t->irradnce[b], n, ig, js, fs);
        }
        if (isolvar < 0) sfluxzen[o + ig - 1] = sf;
        else if (isolvar <= 2) ssi[o + ig - 1] = svar[0] * fb + svar[1] * sd + svar[2] * ir;
        else if (isolvar == 3)
            ssi[o + ig - 1] = F2(svar_bnd, 29, b, 1) * fb + F2(svar_bnd, 29, b, 2) * sd + F2(svar_bnd, 29, b, 3) * ir;
    }
}

/* taumol_sw for ONE column (SW/rrtmg_sw_taumol.F90:27-2084).  taug/taur: F2(x,nlay,lay,ig); ssi/sfluxzen[112]. */
static void SFX(sw_taumol_col)(const SFX(swcol_t) * s, int isolvar, const REAL *svar, const REAL *svar_bnd, REAL *taug,
                               REAL *taur, REAL *ssi, REAL *sfluxzen)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const int nlay = s->nlay, laytrop = s->laytrop;
    const REAL oneminus = *t->oneminus;
#define TG(lay, g) F2(taug, nlay, lay, g)
#define TR(lay, g) F2(taur, nlay, lay, g)
#define I0A(n) (((s->jp[lay] - 1) * 5 + (s->jt[lay] - 1)) * (n))
#define I1A(n) ((s->jp[lay] * 5 + (s->jt1[lay] - 1)) * (n))
#define I0B(n) (((s->jp[lay] - 13) * 5 + (s->jt[lay] - 1)) * (n))
#define I1B(n) (((s->jp[lay] - 12) * 5 + (s->jt1[lay] - 1)) * (n))
#define SELF(b) (s->selffac[lay] * LIN1(t->selfref[b], 10, s->indself[lay], ig, s->selffrac[lay]))
#define FORN(b, nf) (s->forfac[lay] * LIN1(t->forref[b], nf, s->indfor[lay], ig, s->forfrac[lay]))
#define FAC s->fac00[lay], s->fac10[lay], s->fac01[lay], s->fac11[lay]
    /* reference layer for the solar source: lower-atmosphere search (e.g. :585-640) */
#define LAYSOL_LOWER(layreffr, out)                                                                  \
    { int laysolfr = laytrop; out = laytrop;                                                         \
      for (int lay = 1; lay <= laytrop; lay++) {                                                     \
          if (s->jp[lay] < (layreffr) && s->jp[lay + 1] >= (layreffr)) laysolfr = (lay + 1 < laytrop) ? lay + 1 : laytrop; \
          if (lay == laysolfr) { out = lay; break; } } }
    /* upper-atmosphere search (:439-474) */
#define LAYSOL_UPPER(layreffr, out)                                                                  \
    { int laysolfr = nlay; out = 0;                                                                  \
      for (int lay = laytrop + 1; lay <= nlay; lay++) {                                              \
          if (s->jp[lay - 1] < (layreffr) && s->jp[lay] >= (layreffr)) laysolfr = lay;              \
          if (lay == laysolfr) { out = lay; break; } } }

    for (int lay = 1; lay <= nlay; lay++) {
        const int lower = lay <= laytrop;
        /* band 16 (:205-348): h2o,ch4 | ch4 */
        { const int b = 16, n = 6, o = 0; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)252.131, s->colch4[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 3)); TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colch4[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC); TR(lay, o + ig) = tauray; } } }
        /* band 17 (:353-540): h2o,co2 both */
        { const int b = 17, n = 12, o = 6; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)0.364641, s->colco2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 4)); TR(lay, o + ig) = tauray; }
          } else { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)0.364641, s->colco2[lay], 4, oneminus);
              int ind0 = I0B(5) + sp.js, ind1 = I1B(5) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absb[b], 1175, ig, ind0, ind1, 5, sp.fs, FAC) +
                    s->colh2o[lay] * s->forfac[lay] * LIN1(t->forref[b], 4, s->indfor[lay], ig, s->forfrac[lay]); TR(lay, o + ig) = tauray; } } }
        /* band 18 (:545-698): h2o,ch4 | ch4 */
        { const int b = 18, n = 8, o = 18; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)38.9589, s->colch4[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 3)); TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colch4[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC); TR(lay, o + ig) = tauray; } } }
        /* band 19 (:703-850): h2o,co2 | co2 */
        { const int b = 19, n = 8, o = 26; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)5.49281, s->colco2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 3)); TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colco2[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC); TR(lay, o + ig) = tauray; } } }
        /* band 20 (:855-968): h2o; + ch4 */
        { const int b = 20, n = 10, o = 34; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { int ind0 = I0A(1) + 1, ind1 = I1A(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colh2o[lay] * ((SFX(sw_major1)(t->absa[b], 65, ig, ind0, ind1, FAC)) + SELF(b) + FORN(b, 4)) +
                    s->colch4[lay] * t->absch4[ig - 1]; TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colh2o[lay] * (SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC) + FORN(b, 4)) +
                    s->colch4[lay] * t->absch4[ig - 1]; TR(lay, o + ig) = tauray; } } }
        /* band 21 (:973-1145): h2o,co2 both */
        { const int b = 21, n = 10, o = 44; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)0.0045321, s->colco2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 4)); TR(lay, o + ig) = tauray; }
          } else { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)0.0045321, s->colco2[lay], 4, oneminus);
              int ind0 = I0B(5) + sp.js, ind1 = I1B(5) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absb[b], 1175, ig, ind0, ind1, 5, sp.fs, FAC) +
                    s->colh2o[lay] * s->forfac[lay] * LIN1(t->forref[b], 4, s->indfor[lay], ig, s->forfrac[lay]); TR(lay, o + ig) = tauray; } } }
        /* band 22 (:1150-1310): h2o,o2 | o2; o2 continuum */
        { const int b = 22, n = 2, o = 54; REAL tauray = s->colmol[lay] * *t->rayl[b];
          const REAL o2adj = (REAL)1.6, strrat = (REAL)0.022708;
          if (lower) { REAL o2cont = (REAL)4.35e-4 * s->colo2[lay] / ((REAL)350.0 * (REAL)2.0);
              SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], o2adj * strrat, s->colo2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colh2o[lay] * (SELF(b) + FORN(b, 3)) + o2cont; TR(lay, o + ig) = tauray; }
          } else { REAL o2cont = (REAL)4.35e-4 * s->colo2[lay] / ((REAL)350. * (REAL)2.);
              int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colo2[lay] * o2adj * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC) + o2cont;
                    TR(lay, o + ig) = tauray; } } }
        /* band 23 (:1315-1405): h2o | nothing; rayl(ig) */
        { const int b = 23, n = 10, o = 56; const REAL givfac = (REAL)1.029;
          if (lower) { int ind0 = I0A(1) + 1, ind1 = I1A(1) + 1;
              for (int ig = 1; ig <= n; ig++) { REAL tauray = s->colmol[lay] * t->rayl[b][ig - 1];
                  TG(lay, o + ig) = s->colh2o[lay] * (givfac * SFX(sw_major1)(t->absa[b], 65, ig, ind0, ind1, FAC) + SELF(b) + FORN(b, 3));
                  TR(lay, o + ig) = tauray; }
          } else for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = 0; TR(lay, o + ig) = s->colmol[lay] * t->rayl[b][ig - 1]; } }
        /* band 24 (:1410-1540): h2o,o2 | o2; + o3; rayla(ig,js) / raylb(ig) */
        { const int b = 24, n = 8, o = 66;
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[lay], (REAL)0.124692, s->colo2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { REAL tauray = s->colmol[lay] * LIN2(t->rayla24, 8, ig, sp.js, sp.fs);
                  TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC) +
                    s->colo3[lay] * t->abso3a24[ig - 1] + s->colh2o[lay] * (SELF(b) + FORN(b, 3)); TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { REAL tauray = s->colmol[lay] * t->raylb24[ig - 1];
                  TG(lay, o + ig) = s->colo2[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC) + s->colo3[lay] * t->abso3b24[ig - 1];
                  TR(lay, o + ig) = tauray; } } }
        /* band 25 (:1545-1640): h2o | nothing; + o3 */
        { const int b = 25, n = 6, o = 74;
          if (lower) { int ind0 = I0A(1) + 1, ind1 = I1A(1) + 1;
              for (int ig = 1; ig <= n; ig++) { REAL tauray = s->colmol[lay] * t->rayl[b][ig - 1];
                  TG(lay, o + ig) = s->colh2o[lay] * SFX(sw_major1)(t->absa[b], 65, ig, ind0, ind1, FAC) + s->colo3[lay] * t->abso3a25[ig - 1];
                  TR(lay, o + ig) = tauray; }
          } else for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colo3[lay] * t->abso3b25[ig - 1]; TR(lay, o + ig) = s->colmol[lay] * t->rayl[b][ig - 1]; } }
        /* band 26 (:1645-1710): Rayleigh only */
        { const int b = 26, n = 6, o = 80;
          for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = 0; TR(lay, o + ig) = s->colmol[lay] * t->rayl[b][ig - 1]; } }
        /* band 27 (:1715-1810): o3 both */
        { const int b = 27, n = 8, o = 86;
          if (lower) { int ind0 = I0A(1) + 1, ind1 = I1A(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colo3[lay] * SFX(sw_major1)(t->absa[b], 65, ig, ind0, ind1, FAC);
                  TR(lay, o + ig) = s->colmol[lay] * t->rayl[b][ig - 1]; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colo3[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC);
                  TR(lay, o + ig) = s->colmol[lay] * t->rayl[b][ig - 1]; } } }
        /* band 28 (:1815-1975): o3,o2 both */
        { const int b = 28, n = 6, o = 94; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { SFX(swspec_t) sp = SFX(swspec)(s->colo3[lay], (REAL)6.67029e-07, s->colo2[lay], 8, oneminus);
              int ind0 = I0A(9) + sp.js, ind1 = I1A(9) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absa[b], 585, ig, ind0, ind1, 9, sp.fs, FAC); TR(lay, o + ig) = tauray; }
          } else { SFX(swspec_t) sp = SFX(swspec)(s->colo3[lay], (REAL)6.67029e-07, s->colo2[lay], 4, oneminus);
              int ind0 = I0B(5) + sp.js, ind1 = I1B(5) + sp.js;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = sp.speccomb * SFX(sw_major2)(t->absb[b], 1175, ig, ind0, ind1, 5, sp.fs, FAC); TR(lay, o + ig) = tauray; } } }
        /* band 29 (:1980-2082): h2o + co2 | co2 + h2o */
        { const int b = 29, n = 12, o = 100; REAL tauray = s->colmol[lay] * *t->rayl[b];
          if (lower) { int ind0 = I0A(1) + 1, ind1 = I1A(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colh2o[lay] * ((SFX(sw_major1)(t->absa[b], 65, ig, ind0, ind1, FAC)) + SELF(b) + FORN(b, 4)) +
                    s->colco2[lay] * t->absco2[ig - 1]; TR(lay, o + ig) = tauray; }
          } else { int ind0 = I0B(1) + 1, ind1 = I1B(1) + 1;
              for (int ig = 1; ig <= n; ig++) { TG(lay, o + ig) = s->colco2[lay] * SFX(sw_major1)(t->absb[b], 235, ig, ind0, ind1, FAC) +
                    s->colh2o[lay] * t->absh2o[ig - 1]; TR(lay, o + ig) = tauray; } } }
    }
    /* ---- solar source terms --------------------------------------------------------------------- */
    SFX(sw_source)(16, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    { int l; LAYSOL_UPPER(30, l);   /* band 17 (:439-526) */
      if (l) { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)0.364641, s->colco2[l], 4, oneminus); SFX(sw_source)(17, 5, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    { int l; LAYSOL_LOWER(6, l);
      { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)38.9589, s->colch4[l], 8, oneminus); SFX(sw_source)(18, 9, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    { int l; LAYSOL_LOWER(3, l);
      { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)5.49281, s->colco2[l], 8, oneminus); SFX(sw_source)(19, 9, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    SFX(sw_source)(20, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    { int l; LAYSOL_LOWER(8, l);
      { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)0.0045321, s->colco2[l], 8, oneminus); SFX(sw_source)(21, 9, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    { int l; LAYSOL_LOWER(2, l);
      { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)1.6 * (REAL)0.022708, s->colo2[l], 8, oneminus); SFX(sw_source)(22, 9, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    SFX(sw_source)(23, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    { int l; LAYSOL_LOWER(1, l);
      { SFX(swspec_t) sp = SFX(swspec)(s->colh2o[l], (REAL)0.124692, s->colo2[l], 8, oneminus); SFX(sw_source)(24, 9, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    SFX(sw_source)(25, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    SFX(sw_source)(26, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    SFX(sw_source)(27, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
    { int l; LAYSOL_UPPER(42, l);
      if (l) { SFX(swspec_t) sp = SFX(swspec)(s->colo3[l], (REAL)6.67029e-07, s->colo2[l], 4, oneminus); SFX(sw_source)(28, 5, sp.js, sp.fs, isolvar, svar, svar_bnd, ssi, sfluxzen); } }
    SFX(sw_source)(29, 1, 0, 0, isolvar, svar, svar_bnd, ssi, sfluxzen);
#undef TG
#undef TR
#undef I0A
#undef I1A
#undef I0B
#undef I1B
#undef SELF
#undef FORN
#undef FAC
#undef LAYSOL_LOWER
#undef LAYSOL_UPPER
}

/* cldprmc_sw for ONE column (SW/rrtmg_sw_cldprmc.F90:36-418), non-SOLAR_RADVAL.  Arrays F2(x,nlay,lay,ig);
 * reicmc/relqmc 1-based per layer.  Returns 0, 10 invalid iceflag, 11 invalid liqflag. */
static int SFX(sw_cldprmc_col)(int nlay, int iceflag, int liqflag, const int *cldymc, const REAL *ciwpmc, const REAL *clwpmc,
                               const REAL *reicmc, const REAL *relqmc, REAL *taormc, REAL *taucmc, REAL *ssacmc, REAL *asmcmc)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const REAL epsg = (REAL)1.e-06, cldmin = (REAL)1.e-20;
    if (iceflag < 1 || iceflag > 4) return 10;
    if (liqflag != 1) return 11;
    for (int ig = 1; ig <= NGSW; ig++) {
        const int ibf = t->ngb[ig - 1];         /* 16..29 */
        const int ib = ibf - 15;                /* second index of the (n,16:29) tables */
        for (int lay = 1; lay <= nlay; lay++) {
            if (!F2(cldymc, nlay, lay, ig)) {
                F2(taormc, nlay, lay, ig) = 0; F2(taucmc, nlay, lay, ig) = 0; F2(ssacmc, nlay, lay, ig) = 1; F2(asmcmc, nlay, lay, ig) = 0;
                continue;
            }
            REAL extcoice = 0, ssacoice = 0, gice = 0, forwice = 0, extcoliq = 0, ssacoliq = 0, gliq = 0, forwliq = 0;
            const REAL ciwp = F2(ciwpmc, nlay, lay, ig), clwp = F2(clwpmc, nlay, lay, ig);
            if (ciwp != 0) {
                REAL radice = reicmc[lay];
                if (iceflag == 1) {
                    int ic = t->icxa[ibf - 16];
                    extcoice = t->abari[ic - 1] + t->bbari[ic - 1] / radice;
                    ssacoice = (REAL)1. - t->cbari[ic - 1] - t->dbari[ic - 1] * radice;
                    gice = t->ebari[ic - 1] + t->fbari[ic - 1] * radice;
                    if (gice > (REAL)1. - epsg) gice = (REAL)1. - epsg;
                    forwice = gice * gice;
                } else if (iceflag == 2) {
                    REAL factor = (radice - (REAL)2.) / (REAL)3.; int index = (int)factor; if (index == 43) index = 42;
                    REAL fint = factor - (REAL)index;
                    extcoice = LIN1(t->extice2, 43, index, ib, fint); ssacoice = LIN1(t->ssaice2, 43, index, ib, fint);
                    gice = LIN1(t->asyice2, 43, index, ib, fint); forwice = gice * gice;
                } else if (iceflag == 3) {
                    REAL factor = (radice - (REAL)2.) / (REAL)3.; int index = (int)factor; if (index == 46) index = 45;
                    REAL fint = factor - (REAL)index;
                    extcoice = LIN1(t->extice3, 46, index, ib, fint); ssacoice = LIN1(t->ssaice3, 46, index, ib, fint);
                    gice = LIN1(t->asyice3, 46, index, ib, fint);
                    REAL fdelta = LIN1(t->fdlice3, 46, index, ib, fint);
                    forwice = fdelta + (REAL)0.5 / ssacoice;
                    if (forwice > gice) forwice = gice;
                } else {
                    REAL factor = radice; int index = (int)factor; REAL fint = factor - (REAL)index;
                    extcoice = LIN1(t->extice4, 200, index, ib, fint); ssacoice = LIN1(t->ssaice4, 200, index, ib, fint);
                    gice = LIN1(t->asyice4, 200, index, ib, fint); forwice = gice * gice;
                }
            }
            if (clwp != 0) {
                REAL radliq = relqmc[lay];
                int index = (int)(radliq - (REAL)1.5);
                if (index == 0) index = 1;
                if (index == 58) index = 57;
                REAL fint = radliq - (REAL)1.5 - (REAL)index;
                extcoliq = LIN1(t->extliq1, 58, index, ib, fint);
                ssacoliq = LIN1(t->ssaliq1, 58, index, ib, fint);
                if (fint < 0 && ssacoliq > (REAL)1.) ssacoliq = F2(t->ssaliq1, 58, index, ib);
                gliq = LIN1(t->asyliq1, 58, index, ib, fint);
                forwliq = gliq * gliq;
            }
            REAL tauliqorig = clwp * extcoliq, tauiceorig = ciwp * extcoice;
            F2(taormc, nlay, lay, ig) = tauliqorig + tauiceorig;
            REAL ssaliq = ssacoliq * ((REAL)1. - forwliq) / ((REAL)1. - forwliq * ssacoliq);
            REAL ssaice = ssacoice * ((REAL)1. - forwice) / ((REAL)1. - forwice * ssacoice);
            REAL tauliq = ((REAL)1. - forwliq * ssacoliq) * tauliqorig;
            REAL tauice = ((REAL)1. - forwice * ssacoice) * tauiceorig;
            REAL scatliq = ssaliq * tauliq, scatice = ssaice * tauice;
            REAL tc = tauliq + tauice;
            if (tc == 0) tc = cldmin;
            if (scatice == 0) scatice = cldmin;
            F2(taucmc, nlay, lay, ig) = tc;
            F2(ssacmc, nlay, lay, ig) = (scatliq + scatice) / tc;
            if (iceflag == 3)
                F2(asmcmc, nlay, lay, ig) = ((REAL)1. / (scatliq + scatice)) *
                    (scatliq * (gliq - forwliq) / ((REAL)1. - forwliq) + scatice * ((gice - forwice) / ((REAL)1. - forwice)));
            else
                F2(asmcmc, nlay, lay, ig) = (scatliq * (gliq - forwliq) / ((REAL)1. - forwliq) +
                                             scatice * (gice - forwice) / ((REAL)1. - forwice)) / (scatliq + scatice);
        }
    }
    return 0;
}

/* reftra_sw for ONE (column, g-point) (SW/rrtmg_sw_spcvmc.F90:1115-1370); arrays 1-based over jk (1 = TOA layer).
 * cloudy: 1-based over ikl = nlay+1-jk (bottom-up) or NULL when !update_cloudy_only. */
static void SFX(sw_reftra)(int nlay, const int *cloudy_bu, const REAL *pgg, REAL prmuz, const REAL *ptau, const REAL *pw,
                           REAL *pref, REAL *prefd, REAL *ptra, REAL *ptrad, int update_cloudy_only)
{
    const REAL eps = (REAL)1.e-08, zwcrit = (REAL)0.9999995, od_lo = (REAL)0.06;
    for (int jk = 1; jk <= nlay; jk++) {
        if (update_cloudy_only && !cloudy_bu[nlay + 1 - jk]) continue;
        REAL zto1 = ptau[jk], zw = pw[jk], zg = pgg[jk];
        double zw8 = zw, zg8 = zg;
        REAL zg3 = (REAL)3. * zg;
        REAL zgamma1 = ((REAL)8. - zw * ((REAL)5. + zg3)) * (REAL)0.25;
        REAL zgamma2 = (REAL)3. * (zw * ((REAL)1. - zg)) * (REAL)0.25;
        REAL zgamma3 = ((REAL)2. - zg3 * prmuz) * (REAL)0.25;
        REAL zgamma4 = (REAL)1. - zgamma3;
        double q = zg8 / (1.0 - zg8);
        double zwo8 = zw8 / (1.0 - (1.0 - zw8) * (q * q));
        REAL zwo = (REAL)zwo8;
        if (zwo >= zwcrit) {
            REAL za = zgamma1 * prmuz, za1 = za - zgamma3, zgt = zgamma1 * zto1;
            REAL ze1 = zto1 / prmuz; if (ze1 > (REAL)500.) ze1 = (REAL)500.;
            REAL ze2 = EXP(-ze1);
            pref[jk] = (zgt - za1 * ((REAL)1. - ze2)) / ((REAL)1. + zgt);
            ptra[jk] = (REAL)1. - pref[jk];
            prefd[jk] = zgt / ((REAL)1. + zgt);
            ptrad[jk] = (REAL)1. - prefd[jk];
            if (ze2 == (REAL)1.) { pref[jk] = 0; ptra[jk] = 1; prefd[jk] = 0; ptrad[jk] = 1; }
        } else {
            REAL za1 = zgamma1 * zgamma4 + zgamma2 * zgamma3, za2 = zgamma1 * zgamma3 + zgamma2 * zgamma4;
            REAL zrk = SQRT(zgamma1 * zgamma1 - zgamma2 * zgamma2);
            REAL zrp = zrk * prmuz, zrp1 = (REAL)1. + zrp, zrm1 = (REAL)1. - zrp, zrk2 = (REAL)2. * zrk;
            REAL zrpp = (REAL)1. - zrp * zrp, zrkg = zrk + zgamma1;
            REAL zr1 = zrm1 * (za2 + zrk * zgamma3), zr2 = zrp1 * (za2 - zrk * zgamma3), zr3 = zrk2 * (zgamma3 - za2 * prmuz);
            REAL zr4 = zrpp * zrkg, zr5 = zrpp * (zrk - zgamma1);
            REAL zt1 = zrp1 * (za1 + zrk * zgamma4), zt2 = zrm1 * (za1 - zrk * zgamma4), zt3 = zrk2 * (zgamma4 + za1 * prmuz);
            REAL zt4 = zr4, zt5 = zr5;
            REAL zbeta = (zgamma1 - zrk) / zrkg;
            REAL ze1 = zrk * zto1; if (ze1 > (REAL)5.) ze1 = (REAL)5.;
            REAL ze2 = zto1 / prmuz; if (ze2 > (REAL)5.) ze2 = (REAL)5.;
            REAL zem1 = ze1 <= od_lo ? (REAL)1. - ze1 + (REAL)0.5 * ze1 * ze1 : EXP(-ze1);
            REAL zep1 = (REAL)1. / zem1;
            REAL zem2 = ze2 <= od_lo ? (REAL)1. - ze2 + (REAL)0.5 * ze2 * ze2 : EXP(-ze2);
            REAL zep2 = (REAL)1. / zem2;
            REAL zdenr = zr4 * zep1 + zr5 * zem1, zdent = zt4 * zep1 + zt5 * zem1;
            if (zdenr >= -eps && zdenr <= eps) { pref[jk] = eps; ptra[jk] = zem2; }
            else {
                pref[jk] = zw * (zr1 * zep1 - zr2 * zem1 - zr3 * zem2) / zdenr;
                ptra[jk] = zem2 - zem2 * zw * (zt1 * zep1 - zt2 * zem1 - zt3 * zep2) / zdent;
            }
            REAL zemm = zem1 * zem1;
            REAL zdend = (REAL)1. / (((REAL)1. - zbeta * zemm) * zrkg);
            prefd[jk] = zgamma2 * ((REAL)1. - zemm) * zdend;
            ptrad[jk] = zrk2 * zem1 * zdend;
        }
    }
}

/* vrtqdr_sw for ONE (column, g-point) (:1374-1588); arrays 1-based, level nlay+1 = surface */
static void SFX(sw_vrtqdr)(int nlay, const REAL *pref, const REAL *prefd, const REAL *ptra, const REAL *ptrad, const REAL *pdbt,
                           const REAL *ptdbt, REAL *pfd, REAL *pfu, REAL *ztdn, REAL *prup, REAL *prupd, REAL *prdnd)
{
    prup[nlay + 1] = pref[nlay + 1];
    prupd[nlay + 1] = prefd[nlay + 1];
    REAL zreflect = (REAL)1. / ((REAL)1. - prefd[nlay + 1] * prefd[nlay]);
    prup[nlay] = pref[nlay] + (ptrad[nlay] * ((ptra[nlay] - pdbt[nlay]) * prefd[nlay + 1] + pdbt[nlay] * pref[nlay + 1])) * zreflect;
    prupd[nlay] = prefd[nlay] + ptrad[nlay] * ptrad[nlay] * prefd[nlay + 1] * zreflect;
    for (int jk = 1; jk <= nlay - 1; jk++) {
        int ikp = nlay + 1 - jk, ikx = ikp - 1;
        REAL zreflectj = (REAL)1. / ((REAL)1. - prupd[ikp] * prefd[ikx]);
        prup[ikx] = pref[ikx] + (ptrad[ikx] * ((ptra[ikx] - pdbt[ikx]) * prupd[ikp] + pdbt[ikx] * prup[ikp])) * zreflectj;
        prupd[ikx] = prefd[ikx] + ptrad[ikx] * ptrad[ikx] * prupd[ikp] * zreflectj;
    }
    ztdn[1] = 1; prdnd[1] = 0; ztdn[2] = ptra[1]; prdnd[2] = prefd[1];
    for (int jk = 2; jk <= nlay; jk++) {
        int ikp = jk + 1;
        zreflect = (REAL)1. / ((REAL)1. - prefd[jk] * prdnd[jk]);
        ztdn[ikp] = ptdbt[jk] * ptra[jk] + (ptrad[jk] * ((ztdn[jk] - ptdbt[jk]) + ptdbt[jk] * pref[jk] * prdnd[jk])) * zreflect;
        prdnd[ikp] = prefd[jk] + ptrad[jk] * ptrad[jk] * prdnd[jk] * zreflect;
    }
    for (int jk = 1; jk <= nlay + 1; jk++) {
        zreflect = (REAL)1. / ((REAL)1. - prdnd[jk] * prupd[jk]);
        pfu[jk] = (ptdbt[jk] * prup[jk] + (ztdn[jk] - ptdbt[jk]) * prupd[jk]) * zreflect;
        pfd[jk] = ptdbt[jk] + (ztdn[jk] - ptdbt[jk] + ptdbt[jk] * prup[jk] * prdnd[jk]) * zreflect;
    }
}

/* ---- public: pinned stages -------------------------------------------------------------------------- */
/* setcoef_sw + taumol_sw; inputs numpy (ncol,nlay)/(ncol,nlay+1) C-order = Fortran (nlay,ncol) partition layout */
void SFX(oracle_sw_setcoef_taumol)(int ncol, int nlay, const REAL *play, const REAL *tlay, const REAL *plev, const REAL *h2ovmr,
                                   const REAL *co2vmr, const REAL *o3vmr, const REAL *ch4vmr, const REAL *o2vmr, int isolvar,
                                   const REAL *svar, const REAL *svar_bnd, REAL *taug, REAL *taur, REAL *ssi, REAL *sfluxzen,
                                   REAL *colmol, int *laytrop)
{
    SFX(swcol_t) s;
    SFX(swcol_alloc)(&s, nlay);
    for (int c = 0; c < ncol; c++) {
        const size_t o = (size_t)c * nlay, ov = (size_t)c * (nlay + 1);
        SFX(sw_setcoef_col)(&s, play + o - 1, tlay + o - 1, plev + ov - 1, h2ovmr + o - 1, co2vmr + o - 1, o3vmr + o - 1, ch4vmr + o - 1,
                            o2vmr + o - 1);
        for (int g = 0; g < NGSW; g++) { ssi[(size_t)c * NGSW + g] = 0; sfluxzen[(size_t)c * NGSW + g] = 0; }
        SFX(sw_taumol_col)(&s, isolvar, svar, svar_bnd, taug + (size_t)c * NGSW * nlay, taur + (size_t)c * NGSW * nlay,
                           ssi + (size_t)c * NGSW, sfluxzen + (size_t)c * NGSW);
        for (int l = 1; l <= nlay; l++) colmol[o + l - 1] = s.colmol[l];
        laytrop[c] = s.laytrop;
    }
    SFX(swcol_free)(&s);
}

int SFX(oracle_sw_cldprmc)(int ncol, int nlay, int iceflag, int liqflag, const int *cldy, const REAL *ciwpmc, const REAL *clwpmc,
                           const REAL *rei, const REAL *rel, REAL *taormc, REAL *taucmc, REAL *ssacmc, REAL *asmcmc)
{
    for (int c = 0; c < ncol; c++) {
        const size_t o = (size_t)c * NGSW * nlay;
        int rc = SFX(sw_cldprmc_col)(nlay, iceflag, liqflag, cldy + o, ciwpmc + o, clwpmc + o, rei + (size_t)c * nlay - 1,
                                     rel + (size_t)c * nlay - 1, taormc + o, taucmc + o, ssacmc + o, asmcmc + o);
        if (rc) return rc;
    }
    return 0;
}

/* ---- NRLSSI2 host routines of the isolvar = 1 branch (SW/NRLSSI2.F90) ------------------------------------------
 * nsolfrac = 134 (:57); intrvl_len = 1 / (nsolfrac - 2), intrvl_len_hf = half of it (:120-121). */
#define NSOLFRAC 134
/* adjust_solcyc_amplitudes (:236-271): amplitude scaling 1 at the solar minimum, indsolvar at the maximum, linear in between.
 * Returns 1 where the reference error-stops (solcycfr outside [0, 1]). */
static int SFX(nrlssi2_adjust)(REAL solcycfr, const REAL *indsolvar, REAL *scl)
{
    const REAL solcycfrac_min = (REAL)0.0189, solcycfrac_max = (REAL)0.3750;
    const REAL fracdiff_min2max = solcycfrac_max - solcycfrac_min, fracdiff_max2min = (REAL)1. - fracdiff_min2max;
    REAL wgt;
    if (solcycfr >= 0 && solcycfr < solcycfrac_min) {
        wgt = (solcycfr + (REAL)1. - solcycfrac_max) / fracdiff_max2min;
        scl[0] = indsolvar[0] + wgt * ((REAL)1. - indsolvar[0]);
        scl[1] = indsolvar[1] + wgt * ((REAL)1. - indsolvar[1]);
    } else if (solcycfr >= solcycfrac_min && solcycfr <= solcycfrac_max) {
        wgt = (solcycfr - solcycfrac_min) / fracdiff_min2max;
        scl[0] = (REAL)1. + wgt * (indsolvar[0] - (REAL)1.);
        scl[1] = (REAL)1. + wgt * (indsolvar[1] - (REAL)1.);
    } else if (solcycfr > solcycfrac_max && solcycfr <= 1) {
        wgt = (solcycfr - solcycfrac_max) / fracdiff_max2min;
        scl[0] = indsolvar[0] + wgt * ((REAL)1. - indsolvar[0]);
        scl[1] = indsolvar[1] + wgt * ((REAL)1. - indsolvar[1]);
    } else return 1;
    return 0;
}

/* interpolate_indices (:277-332): Mg and SB of the mean cycle AvgCyc11 at solcycfr in [0, 1] */
static int SFX(nrlssi2_interp)(REAL solcycfr, REAL *Mg, REAL *SB)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const REAL intrvl_len = (REAL)1.0 / (REAL)(NSOLFRAC - 2), intrvl_len_hf = (REAL)0.5 * intrvl_len;
    const REAL *mg = t->mgavgcyc - 1, *sb = t->sbavgcyc - 1;      /* 1-based like the reference */
    if (solcycfr > 0 && solcycfr < 1) {
        int sfid; REAL fraclo, frachi;
        if (solcycfr <= intrvl_len_hf) { sfid = 1; fraclo = 0; frachi = intrvl_len_hf; }
        else if (solcycfr > intrvl_len_hf && solcycfr < (REAL)1. - intrvl_len_hf) {
            sfid = (int)FLOOR((solcycfr - intrvl_len_hf) * (REAL)(NSOLFRAC - 2)) + 2;
            fraclo = (REAL)(sfid - 2) * intrvl_len + intrvl_len_hf;
            frachi = fraclo + intrvl_len;
        } else { sfid = (NSOLFRAC - 2) + 1; fraclo = (REAL)1. - intrvl_len_hf; frachi = 1; }
        const REAL intfrac = (solcycfr - fraclo) / (frachi - fraclo);
        *Mg = mg[sfid] + intfrac * (mg[sfid + 1] - mg[sfid]);
        *SB = sb[sfid] + intfrac * (sb[sfid + 1] - sb[sfid]);
    } else if (solcycfr == 0) { *Mg = mg[1]; *SB = sb[1]; }
    else if (solcycfr == 1) { *Mg = mg[NSOLFRAC]; *SB = sb[NSOLFRAC]; }
    else return 1;
    return 0;
}

/* initialize_NRLSSI2, isolvar == 1 part (:160-232): <svar_f>, <svar_s> over the cycle with the amplitude scaling applied.
 * indsolvar == NULL: absent optional argument (= 1, 1). */
static void SFX(nrlssi2_means)(const REAL *indsolvar_opt, REAL *mean_f, REAL *mean_s)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const REAL intrvl_len = (REAL)1.0 / (REAL)(NSOLFRAC - 2), intrvl_len_hf = (REAL)0.5 * intrvl_len;
    const REAL *mg = t->mgavgcyc - 1, *sb = t->sbavgcyc - 1;
    REAL indsolvar[2] = {1, 1};
    if (indsolvar_opt) { indsolvar[0] = indsolvar_opt[0]; indsolvar[1] = indsolvar_opt[1]; }
    *mean_f = 1; *mean_s = 1;
    const int scl1 = indsolvar[0] != 1, scl2 = indsolvar[1] != 1;
    if (scl1 || scl2) {
        REAL iscl1_mean = 0, iscl2_mean = 0, iscl1_Mg_mean = 0, iscl2_SB_mean = 0, scl[2];
        if (scl1) iscl1_mean = ((REAL)1. + indsolvar[0]) / (REAL)2.;
        if (scl2) iscl2_mean = ((REAL)1. + indsolvar[1]) / (REAL)2.;
        REAL solcycfr = intrvl_len_hf;
        for (int n = 2; n <= NSOLFRAC - 1; n++) {
            SFX(nrlssi2_adjust)(solcycfr, indsolvar, scl);
            if (scl1) iscl1_Mg_mean = iscl1_Mg_mean + scl[0] * mg[n];
            if (scl2) iscl2_SB_mean = iscl2_SB_mean + scl[1] * sb[n];
            solcycfr = solcycfr + intrvl_len;
        }
        if (scl1) iscl1_Mg_mean = iscl1_Mg_mean / (REAL)(NSOLFRAC - 2);
        if (scl2) iscl2_SB_mean = iscl2_SB_mean / (REAL)(NSOLFRAC - 2);
        if (scl1) *mean_f = (iscl1_Mg_mean - iscl1_mean * *t->Mg_0) / (*t->Mg_avg - *t->Mg_0);
        if (scl2) *mean_s = (iscl2_SB_mean - iscl2_mean * *t->SB_0) / (*t->SB_avg - *t->SB_0);
    }
}

int SFX(oracle_nrlssi2_adjust)(REAL solcycfr, const REAL *indsolvar, REAL *scl) { return SFX(nrlssi2_adjust)(solcycfr, indsolvar, scl); }
int SFX(oracle_nrlssi2_interp)(REAL solcycfr, REAL *MgSB) { return SFX(nrlssi2_interp)(solcycfr, MgSB, MgSB + 1); }
void SFX(oracle_nrlssi2_means)(const REAL *indsolvar, REAL *means) { SFX(nrlssi2_means)(indsolvar, means, means + 1); }

/* the scalars of the solar-variability block (SW/rrtmg_sw_rad.F90:893-1127) alone: svar[3] = svar_f, svar_s, svar_i.  isolvar = 1 only. */
int SFX(oracle_sw_solar_isolvar1)(REAL scon, const REAL *indsolvar, const REAL *solcycfrac, REAL *svar);

/* ---- public: full rrtmg_sw (SW/rrtmg_sw_rad.F90:68-1801 + spcvmc_sw :34-1112), API layouts --------------------
 * isolvar in {-1, 0, 1, 2, 3} (1 needs solcycfrac; GEOS itself never passes 1: GEOS_SolarGridComp.F90:6286-6292).
 * indsolvar / bndscl / solcycfrac may be NULL (absent optional arguments).
 * cot[8] = cotdtp,cotdhp,cotdmp,cotdlp,cotntp,cotnhp,cotnmp,cotnlp (ncol each); drband/dfband (ncol,14) or NULL.
 * Returns 0, 100+k negative input #k, 10/11 invalid cloud flags, 20 invalid isolvar / negative scon, 21 isolvar 1 without
 * solcycfrac, 22 solcycfr outside [0, 1], 5 bad super-layers. */
int SFX(oracle_rrtmg_sw)(int ncol, int nlay, REAL scon, REAL adjes, const REAL *coszen, int isolvar, const REAL *play,
                         const REAL *plev, const REAL *tlay, const REAL *h2ovmr, const REAL *o3vmr, const REAL *co2vmr,
                         const REAL *ch4vmr, const REAL *o2vmr, int iceflgsw, int liqflgsw, const REAL *cld, const REAL *ciwp,
                         const REAL *clwp, const REAL *rei, const REAL *rel, int dyofyr, const REAL *zm, const REAL *alat,
                         int iaer, const REAL *tauaer, const REAL *ssaaer, const REAL *asmaer, const REAL *asdir,
                         const REAL *asdif, const REAL *aldir, const REAL *aldif, int cloudLM, int cloudMH, int normFlx,
                         int *clearCounts, REAL *swuflx, REAL *swdflx, REAL *swuflxc, REAL *swdflxc, REAL *nirr, REAL *nirf,
                         REAL *parr, REAL *parf, REAL *uvrr, REAL *uvrf, REAL *fswband, REAL *cot, int do_drfband,
                         REAL *drband, REAL *dfband, const REAL *bndscl, const REAL *indsolvar, const REAL *solcycfrac)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const size_t cl = (size_t)ncol * nlay;
    /* input assertions (:365-383) */
    const REAL *chk[] = {play, tlay, h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, cld, ciwp, clwp, rei, rel};
    for (size_t k = 0; k < sizeof(chk) / sizeof(chk[0]); k++)
        for (size_t i = 0; i < cl; i++) if (chk[k][i] < 0) return 100 + (int)k;
    for (size_t i = 0; i < (size_t)ncol * (nlay + 1); i++) if (plev[i] < 0) return 120;
    for (int i = 0; i < ncol; i++) if (asdir[i] < 0 || aldir[i] < 0 || asdif[i] < 0 || aldif[i] < 0) return 121;
    if (iaer == 10) for (size_t i = 0; i < cl * 14; i++) if (tauaer[i] < 0 || ssaaer[i] < 0) return 122;
    if (cloudLM == cloudMH) return 5;

    /* solar variability (:893-1127) */
    REAL solvar[30], adjflux[30], svar[3] = {1, 1, 1}, svar_bnd[29 * 3];
    for (int b = 0; b < 30; b++) { solvar[b] = 1; adjflux[b] = 1; }
    for (int i = 0; i < 29 * 3; i++) svar_bnd[i] = 1;
    REAL ndx[2] = {*t->Mg_avg, *t->SB_avg};
    if (isolvar == 2 && indsolvar) { ndx[0] = indsolvar[0]; ndx[1] = indsolvar[1]; }
    const REAL Iint = *t->Iint, Fint = *t->Fint, Sint = *t->Sint;
    if (isolvar != -1 && isolvar != 0 && isolvar != 1 && isolvar != 2 && isolvar != 3) return 20;
    /* isolvar == 1 (:906-930): position in AvgCyc11 from solcycfrac, amplitude scaling from indsolvar; (:994-1008, :1060-1079) */
    if (isolvar == 1) {
        REAL sv1[3];
        if (!solcycfrac) return 21;
        int e = SFX(oracle_sw_solar_isolvar1)(scon, indsolvar, solcycfrac, sv1);
        if (e) return e;
        svar[0] = sv1[0]; svar[1] = sv1[1]; svar[2] = sv1[2];
    }
    if (scon == 0) {
        if (isolvar == -1) { if (bndscl) for (int b = 16; b <= 29; b++) solvar[b] = bndscl[b - 16]; }
        else if (isolvar == 2) { svar[0] = (ndx[0] - *t->Mg_0) / (*t->Mg_avg - *t->Mg_0); svar[1] = (ndx[1] - *t->SB_0) / (*t->SB_avg - *t->SB_0); svar[2] = 1; }
        else if (isolvar == 3) { if (bndscl) for (int b = 16; b <= 29; b++) solvar[b] = bndscl[b - 16];
            for (int b = 16; b <= 29; b++) { F2(svar_bnd, 29, b, 1) = solvar[b]; F2(svar_bnd, 29, b, 2) = solvar[b]; F2(svar_bnd, 29, b, 3) = solvar[b]; } }
    } else if (scon > 0) {
        if (isolvar == -1) { for (int b = 16; b <= 29; b++) solvar[b] = scon / *t->rrsw_scon;
            if (bndscl) for (int b = 16; b <= 29; b++) solvar[b] = solvar[b] * bndscl[b - 16]; }
        else if (isolvar == 0) { REAL scon_int = Fint + Sint + Iint, r = scon / scon_int; svar[0] = r; svar[1] = r; svar[2] = r; }
        else if (isolvar == 2) { svar[0] = (ndx[0] - *t->Mg_0) / (*t->Mg_avg - *t->Mg_0); svar[1] = (ndx[1] - *t->SB_0) / (*t->SB_avg - *t->SB_0);
            svar[2] = (scon - (svar[0] * Fint + svar[1] * Sint)) / Iint; }
        else { REAL scon_int = Fint + Sint + Iint; for (int b = 16; b <= 29; b++) solvar[b] = scon / scon_int;
            if (bndscl) for (int b = 16; b <= 29; b++) solvar[b] = solvar[b] * bndscl[b - 16];
            for (int b = 16; b <= 29; b++) { F2(svar_bnd, 29, b, 1) = solvar[b]; F2(svar_bnd, 29, b, 2) = solvar[b]; F2(svar_bnd, 29, b, 3) = solvar[b]; } }
    } else return 20;
    for (int b = 16; b <= 29; b++) adjflux[b] = adjes;
    if (isolvar < 0) for (int b = 16; b <= 29; b++) adjflux[b] = adjflux[b] * solvar[b];

    const size_t n1 = (size_t)nlay + 3, ng = (size_t)NGSW * nlay;
    SFX(swcol_t) s;
    SFX(swcol_alloc)(&s, nlay);
    REAL *w[16];
    for (int k = 0; k < 16; k++) w[k] = (REAL *)calloc(n1, sizeof(REAL));
    REAL *pav = w[0], *tav = w[1], *plv = w[2], *vh = w[3], *vc = w[4], *vo = w[5], *vm = w[6], *vx = w[7], *vcld = w[8], *vci = w[9],
         *vcl = w[10], *vrei = w[11], *vrel = w[12], *vzm = w[13];
    REAL *taug = calloc(ng, sizeof(REAL)), *taur = calloc(ng, sizeof(REAL)), *taorm = calloc(ng, sizeof(REAL)), *taucm = calloc(ng, sizeof(REAL)),
         *ssacm = calloc(ng, sizeof(REAL)), *asmcm = calloc(ng, sizeof(REAL)), *ciwpm = calloc(ng, sizeof(REAL)), *clwpm = calloc(ng, sizeof(REAL));
    int *cldym = calloc(ng, sizeof(int)), *cldg = calloc(n1, sizeof(int));
    REAL *v[20];
    for (int k = 0; k < 20; k++) v[k] = (REAL *)calloc(n1, sizeof(REAL));
    REAL *zgco = v[0], *zomco = v[1], *ztauo = v[2], *zdbt = v[3], *ztdbt = v[4], *zfd = v[5], *zfu = v[6], *zref = v[7], *zrefd = v[8],
         *ztra = v[9], *ztrad = v[10], *ztdn = v[11], *prup = v[12], *prupd = v[13], *prdnd = v[14], *zfdc = v[15], *zfuc = v[16], *ztdbtc = v[17];
    REAL *acc = calloc(6 * n1, sizeof(REAL));   /* pbbcu pbbcd pbbfu pbbfd (+2 spare) over ikl = 1..nlay+1 */
    static const int so[4] = {4, 3, 2, 1};       /* seed_order of the SW call (:1401) */
    const int surface_at_one = play[0] > play[(size_t)(nlay - 1) * ncol];
    int rc = 0;
    for (int c = 0; c < ncol && !rc; c++) {
        for (int l = 1; l <= nlay; l++) {
            size_t i = (size_t)(l - 1) * ncol + c;
            pav[l] = play[i]; tav[l] = tlay[i]; vh[l] = h2ovmr[i]; vc[l] = co2vmr[i]; vo[l] = o3vmr[i]; vm[l] = ch4vmr[i]; vx[l] = o2vmr[i];
            vcld[l] = cld[i]; vci[l] = ciwp[i]; vcl[l] = clwp[i]; vrei[l] = rei[i]; vrel[l] = rel[i]; vzm[l] = zm[i];
        }
        for (int l = 1; l <= nlay + 1; l++) plv[l] = plev[(size_t)(l - 1) * ncol + c];
        int cloudy_col = 0;
        for (int l = 1; l <= nlay; l++) if (vcld[l] > 0) cloudy_col = 1;
        /* albedo per band (:1230-1248) */
        REAL albdir[15], albdif[15];
        for (int b = 1; b <= 8; b++) { albdir[b] = aldir[c]; albdif[b] = aldif[c]; }
        albdir[14] = aldir[c]; albdif[14] = aldif[c];
        for (int b = 10; b <= 13; b++) { albdir[b] = asdir[c]; albdif[b] = asdif[c]; }
        albdir[9] = (asdir[c] + aldir[c]) / (REAL)2.; albdif[9] = (asdif[c] + aldif[c]) / (REAL)2.;
        REAL cossza = coszen[c] > (REAL)1.e-10 ? coszen[c] : (REAL)1.e-10;
        int cnt[4] = {NGSW, NGSW, NGSW, NGSW};
        if (cloudy_col) {
            SFX(mcica_col)(NGSW, nlay, surface_at_one, vzm, alat[c], dyofyr, pav, vcld, vci, vcl, (REAL)1.e-20, so, cldym, ciwpm, clwpm);
            SFX(clearcounts_col)(NGSW, nlay, cloudLM, cloudMH, cldym, cnt);
            rc = SFX(sw_cldprmc_col)(nlay, iceflgsw, liqflgsw, cldym, ciwpm, clwpm, vrei, vrel, taorm, taucm, ssacm, asmcm);
            if (rc) break;
        }
        for (int k = 0; k < 4; k++) clearCounts[(size_t)k * ncol + c] = cnt[k];
        SFX(sw_setcoef_col)(&s, pav, tav, plv, vh, vc, vo, vm, vx);
        REAL ssi[NGSW], sfz[NGSW];
        for (int g = 0; g < NGSW; g++) { ssi[g] = 0; sfz[g] = 0; }
        SFX(sw_taumol_col)(&s, isolvar, svar, svar_bnd, taug, taur, ssi, sfz);
        memset(acc, 0, 6 * n1 * sizeof(REAL));
        REAL *pbbcu = acc, *pbbcd = acc + n1, *pbbfu = acc + 2 * n1, *pbbfd = acc + 3 * n1;
        REAL znirr = 0, znirf = 0, zparr = 0, zparf = 0, zuvrr = 0, zuvrf = 0, fnds[15], zdr[15], zdf[15];
        for (int b = 0; b < 15; b++) { fnds[b] = 0; zdr[b] = 0; zdf[b] = 0; }
        REAL cotd[4] = {0, 0, 0, 0}, cotn[4] = {0, 0, 0, 0};
        for (int iw = 1; iw <= NGSW; iw++) {
            const int jb = t->ngb[iw - 1], ibm = jb - 15;
            zref[nlay + 1] = albdir[ibm]; zrefd[nlay + 1] = albdif[ibm]; ztra[nlay + 1] = 0; ztrad[nlay + 1] = 0; ztdbt[1] = 1;
            /* clear-sky delta-scaled optical properties (:413-437) */
            for (int jk = 1; jk <= nlay; jk++) {
                int ikl = nlay + 1 - jk;
                REAL ta = 0, om = 1, as = 0;
                if (iaer == 10) { size_t i = ((size_t)(ibm - 1) * nlay + (ikl - 1)) * ncol + c; ta = tauaer[i]; om = ssaaer[i]; as = asmaer[i]; }
                ztauo[jk] = F2(taur, nlay, ikl, iw) + F2(taug, nlay, ikl, iw) + ta;
                zomco[jk] = F2(taur, nlay, ikl, iw) + ta * om;
                zgco[jk] = (as * om * ta) / zomco[jk];
                zomco[jk] = zomco[jk] / ztauo[jk];
                REAL zf = zgco[jk] * zgco[jk], zwf = zomco[jk] * zf;
                ztauo[jk] = ((REAL)1. - zwf) * ztauo[jk];
                zomco[jk] = (zomco[jk] - zwf) / ((REAL)1. - zwf);
                zgco[jk] = (zgco[jk] - zf) / ((REAL)1. - zf);
            }
            SFX(sw_reftra)(nlay, NULL, zgco, cossza, ztauo, zomco, zref, zrefd, ztra, ztrad, 0);
            for (int jk = 1; jk <= nlay; jk++) { zdbt[jk] = EXP(-ztauo[jk] / cossza); ztdbt[jk + 1] = zdbt[jk] * ztdbt[jk]; }
            SFX(sw_vrtqdr)(nlay, zref, zrefd, ztra, ztrad, zdbt, ztdbt, zfdc, zfuc, ztdn, prup, prupd, prdnd);
            REAL zincflx = isolvar < 0 ? adjflux[jb] * sfz[iw - 1] * cossza : adjflux[jb] * ssi[iw - 1] * cossza;
            for (int ikl = 1; ikl <= nlay + 1; ikl++) {
                int jk = nlay + 2 - ikl;
                pbbcu[ikl] = pbbcu[ikl] + zincflx * zfuc[jk];
                pbbcd[ikl] = pbbcd[ikl] + zincflx * zfdc[jk];
            }
            for (int jk = 1; jk <= nlay + 1; jk++) ztdbtc[jk] = ztdbt[jk];
            const REAL *fdS = zfdc, *fuS = zfuc, *tdbtS = ztdbtc;   /* surface values of the sky that counts as "total" */
            if (cloudy_col) {
                for (int jk = 1; jk <= nlay; jk++) {
                    int ikl = nlay + 1 - jk;
                    cldg[ikl] = F2(cldym, nlay, ikl, iw);
                    if (cldg[ikl]) {
                        REAL tc = F2(taucm, nlay, ikl, iw), oc = F2(ssacm, nlay, ikl, iw), gc = F2(asmcm, nlay, ikl, iw);
                        zgco[jk] = ztauo[jk] * zomco[jk] * zgco[jk] + tc * oc * gc;
                        zomco[jk] = ztauo[jk] * zomco[jk] + tc * oc;
                        ztauo[jk] = ztauo[jk] + tc;
                        zgco[jk] = zgco[jk] / zomco[jk];
                        zomco[jk] = zomco[jk] / ztauo[jk];
                    }
                }
                SFX(sw_reftra)(nlay, cldg, zgco, cossza, ztauo, zomco, zref, zrefd, ztra, ztrad, 1);
                for (int jk = 1; jk <= nlay; jk++) {
                    int ikl = nlay + 1 - jk;
                    if (cldg[ikl]) zdbt[jk] = EXP(-ztauo[jk] / cossza);
                    ztdbt[jk + 1] = zdbt[jk] * ztdbt[jk];
                }
                SFX(sw_vrtqdr)(nlay, zref, zrefd, ztra, ztrad, zdbt, ztdbt, zfd, zfu, ztdn, prup, prupd, prdnd);
                for (int ikl = 1; ikl <= nlay + 1; ikl++) {
                    int jk = nlay + 2 - ikl;
                    pbbfu[ikl] = pbbfu[ikl] + zincflx * zfu[jk];
                    pbbfd[ikl] = pbbfd[ikl] + zincflx * zfd[jk];
                }
                fdS = zfd; fuS = zfu; tdbtS = ztdbt;
            }
            /* surface band fluxes (:624-671) use the total-sky arrays (== clear for cloud-free columns) */
            {
                const REAL dirs = zincflx * tdbtS[nlay + 1], tots = zincflx * fdS[nlay + 1];
                if (ibm == 14 || ibm <= 8) { znirr = znirr + dirs; znirf = znirf + tots; }
                else if (ibm >= 10 && ibm <= 11) { zparr = zparr + dirs; zparf = zparf + tots; }
                else if (ibm >= 12 && ibm <= 13) { zuvrr = zuvrr + dirs; zuvrf = zuvrf + tots; }
                else if (ibm == 9) {
                    zparr = zparr + (REAL)0.5 * zincflx * tdbtS[nlay + 1]; zparf = zparf + (REAL)0.5 * zincflx * fdS[nlay + 1];
                    znirr = znirr + (REAL)0.5 * zincflx * tdbtS[nlay + 1]; znirf = znirf + (REAL)0.5 * zincflx * fdS[nlay + 1];
                }
                fnds[ibm] = fnds[ibm] + zincflx * (fdS[nlay + 1] - fuS[nlay + 1]);
                if (do_drfband) { zdr[ibm] = zdr[ibm] + dirs; zdf[ibm] = zdf[ibm] + tots; }
            }
            /* PAR in-cloud optical thickness diagnostics (:749-1109) */
            if (cloudy_col) {
                REAL wgt;
                if (ibm >= 10 && ibm <= 11) wgt = (REAL)1.0; else if (ibm == 9) wgt = (REAL)0.5; else wgt = -1;
                if (wgt > 0) {
                    REAL zi = isolvar < 0 ? adjflux[jb] * sfz[iw - 1] : adjflux[jb] * ssi[iw - 1];
                    wgt = wgt * zi;
                    REAL sl = 0, sm = 0, sh = 0;
                    for (int l = 1; l <= cloudLM; l++) sl = sl + F2(taorm, nlay, l, iw);
                    for (int l = cloudLM + 1; l <= cloudMH; l++) sm = sm + F2(taorm, nlay, l, iw);
                    for (int l = cloudMH + 1; l <= nlay; l++) sh = sh + F2(taorm, nlay, l, iw);
                    if (sl > 0) { cotd[3] = cotd[3] + wgt; cotn[3] = cotn[3] + wgt * sl; }
                    if (sm > 0) { cotd[2] = cotd[2] + wgt; cotn[2] = cotn[2] + wgt * sm; }
                    if (sh > 0) { cotd[1] = cotd[1] + wgt; cotn[1] = cotn[1] + wgt * sh; }
                    REAL st = sl + sm + sh;
                    if (st > 0) { cotd[0] = cotd[0] + wgt; cotn[0] = cotn[0] + wgt * st; }
                }
            }
        }
        if (!cloudy_col) for (int ikl = 1; ikl <= nlay + 1; ikl++) { pbbfu[ikl] = pbbcu[ikl]; pbbfd[ikl] = pbbcd[ikl]; }
        /* scatter back (:1515-1755) */
        for (int lev = 1; lev <= nlay + 1; lev++) {
            size_t i = (size_t)(lev - 1) * ncol + c;
            swuflxc[i] = pbbcu[lev]; swdflxc[i] = pbbcd[lev]; swuflx[i] = pbbfu[lev]; swdflx[i] = pbbfd[lev];
        }
        nirr[c] = znirr; nirf[c] = znirf - znirr; parr[c] = zparr; parf[c] = zparf - zparr; uvrr[c] = zuvrr; uvrf[c] = zuvrf - zuvrr;
        for (int b = 1; b <= 14; b++) {
            fswband[(size_t)(b - 1) * ncol + c] = fnds[b];
            if (do_drfband) { drband[(size_t)(b - 1) * ncol + c] = zdr[b]; dfband[(size_t)(b - 1) * ncol + c] = zdf[b] - zdr[b]; }
        }
        for (int k = 0; k < 4; k++) { cot[(size_t)k * ncol + c] = cotd[k]; cot[(size_t)(4 + k) * ncol + c] = cotn[k]; }
        if (normFlx == 1) {
            REAL top = swdflx[(size_t)nlay * ncol + c]; if (top < (REAL)1e-7) top = (REAL)1e-7;
            for (int lev = 1; lev <= nlay + 1; lev++) {
                size_t i = (size_t)(lev - 1) * ncol + c;
                swuflxc[i] = swuflxc[i] / top; swdflxc[i] = swdflxc[i] / top; swuflx[i] = swuflx[i] / top; swdflx[i] = swdflx[i] / top;
            }
            nirr[c] /= top; nirf[c] /= top; parr[c] /= top; parf[c] /= top; uvrr[c] /= top; uvrf[c] /= top;
            for (int b = 0; b < 14; b++) {
                fswband[(size_t)b * ncol + c] /= top;
                if (do_drfband) { drband[(size_t)b * ncol + c] /= top; dfband[(size_t)b * ncol + c] /= top; }
            }
        }
    }
    for (int k = 0; k < 16; k++) free(w[k]);
    for (int k = 0; k < 20; k++) free(v[k]);
    free(taug); free(taur); free(taorm); free(taucm); free(ssacm); free(asmcm); free(ciwpm); free(clwpm); free(cldym); free(cldg); free(acc);
    SFX(swcol_free)(&s);
    return rc;
}

#undef LIN1
#undef LIN2
#undef NGSW


int SFX(oracle_sw_solar_isolvar1)(REAL scon, const REAL *indsolvar, const REAL *solcycfrac, REAL *svar)
{
    const SFX(sw_tables_t) *t = &SFX(S);
    const REAL Iint = *t->Iint, Fint = *t->Fint, Sint = *t->Sint;
    REAL scl[2] = {1, 1}, Mg_now, SB_now, mean_f, mean_s;
    if (!solcycfrac) return 21;
    const REAL solcycfr = *solcycfrac;
    if (scon < 0) return 20;
    if (indsolvar && (indsolvar[0] != 1 || indsolvar[1] != 1))
        if (SFX(nrlssi2_adjust)(solcycfr, indsolvar, scl)) return 22;
    SFX(nrlssi2_means)(indsolvar, &mean_f, &mean_s);                  /* call initialize_NRLSSI2 (isolvar, indsolvar) (:946) */
    if (SFX(nrlssi2_interp)(solcycfr, &Mg_now, &SB_now)) return 22;
    svar[0] = scl[0] * (Mg_now - *t->Mg_0) / (*t->Mg_avg - *t->Mg_0);
    svar[1] = scl[1] * (SB_now - *t->SB_0) / (*t->SB_avg - *t->SB_0);
    if (scon == 0) svar[2] = 1;
    else svar[2] = (scon - (mean_f * Fint + mean_s * Sint)) / Iint;
    return 0;
}
