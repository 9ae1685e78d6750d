/* chou_oracle_impl.h -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * Plain-C restatement of the Chou-Suarez longwave scheme `irrad` (reference: GEOSirrad_GridComp/irrad.F90:27-1338 and its
 * helpers :1341-2780; cloud optics GEOS_RadiationShared/getirtau.code:1-102), included once per precision by lw_oracle.c
 * (REAL, SFX() as in lw_oracle_impl.h).  Non-OVERCAST build (maximum-random overlap by sorted super-layers).
 *
 * PARITY UNPINNED: irrad.F90 cannot be compiled in this image -- it `use`s module gettau, whose source needs
 * MAPL_ConstantsMod (MAPL is absent and no stand-in is written), and the reference ships no fixtures for it.  Only the
 * coefficient tables are reference data (irrad_constants / rad_constants compiled from the reference and dumped by
 * oracle/ref_glue.F90:ref_chou_lw_dump_tables -> geosradiation_gridcomp_amd/data/chou_lw_*.grtb).  MAPL_GRAV = 9.80665 is
 * MAPL's value (not defined in the reference repository).  tests/test_oracle_chou.py holds this
 * restatement to physical invariants and to consistency with the pinned RRTMG_LW oracle instead.
 *
 * Index conventions follow the reference: layers 1..np from the TOP down, extra layer 0 above the model top, level np+1 =
 * surface; API arrays Fortran (m, np[+1][, x]) = column index fastest; upward fluxes are negative.
 */

typedef struct {
    const REAL *xkw, *xke, *aw, *bw, *pm, *fkw, *gkw, *cb, *dcb;
    const int *mw;
    const REAL *w11, *w12, *w13, *p11, *p12, *p13, *dwe, *dpe;
    const REAL *c1, *c2, *c3, *oo1, *oo2, *oo3, *h11, *h12, *h13, *h21, *h22, *h23, *h81, *h82, *h83;
    const REAL *aib_ir, *awb_ir, *aiw_ir, *aww_ir, *aig_ir, *awg_ir;
} SFX(chou_tables_t);
static SFX(chou_tables_t) SFX(CH);

int SFX(oracle_chou_set_table)(const char *name, const void *p)
{
    SFX(chou_tables_t) *t = &SFX(CH);
#define SETC(nm) if (!strcmp(name, #nm)) { t->nm = p; return 0; }
    SETC(xkw) SETC(xke) SETC(aw) SETC(bw) SETC(pm) SETC(fkw) SETC(gkw) SETC(cb) SETC(dcb) SETC(mw)
    SETC(w11) SETC(w12) SETC(w13) SETC(p11) SETC(p12) SETC(p13) SETC(dwe) SETC(dpe)
    SETC(c1) SETC(c2) SETC(c3) SETC(oo1) SETC(oo2) SETC(oo3) SETC(h11) SETC(h12) SETC(h13) SETC(h21) SETC(h22) SETC(h23)
    SETC(h81) SETC(h82) SETC(h83) SETC(aib_ir) SETC(awb_ir) SETC(aiw_ir) SETC(aww_ir) SETC(aig_ir) SETC(awg_ir)
#undef SETC
    return 1;
}

#define CH_NX 26
#define CH_NO 21
#define CH_NC 30
#define CH_NH 31
#define CH_GRAV ((REAL)9.80665)

/* planck / plancd (irrad.F90:1341-1376) */
static REAL SFX(ch_planck)(int ibn, REAL t)
{
    const REAL *cb = SFX(CH).cb + 6 * (ibn - 1);
    return t * (t * (t * (t * (t * cb[5] + cb[4]) + cb[3]) + cb[2]) + cb[1]) + cb[0];
}
static REAL SFX(ch_plancd)(int ibn, REAL t)
{
    const REAL *d = SFX(CH).dcb + 5 * (ibn - 1);
    return t * (t * (t * (t * d[4] + d[3]) + d[2]) + d[1]) + d[0];
}

/* tablup (irrad.F90:1887-2011): running absorber-weighted p, T and quadratic/linear table interpolation */
static void SFX(ch_tablup)(int nx, int nh, REAL dw, REAL p, REAL dt, REAL *s1, REAL *s2, REAL *s3, REAL w1, REAL p1, REAL dwe,
                           REAL dpe, const REAL *coef1, const REAL *coef2, const REAL *coef3, REAL *tran)
{
    *s1 = *s1 + dw; *s2 = *s2 + p * dw; *s3 = *s3 + dt * dw;
    const REAL x1 = *s1, x1c = (REAL)1.0 / *s1, x2 = *s2 * x1c, x3 = *s3 * x1c;
    REAL we = (LOG10(x1) - w1) * dwe, pe = (LOG10(x2) - p1) * dpe;
    if (we > (REAL)(nh - 1)) we = (REAL)(nh - 1);
    if (pe > (REAL)(nx - 1)) pe = (REAL)(nx - 1);
    int iw = (int)(we + (REAL)1.0); if (iw > nh - 1) iw = nh - 1; if (iw < 2) iw = 2;
    const REAL fw = we - (REAL)(iw - 1);
    int ip = (int)(pe + (REAL)1.0); if (ip > nx - 1) ip = nx - 1; if (ip < 1) ip = 1;
    const REAL fp = pe - (REAL)(ip - 1);
#define C(t, i, j) F2(t, nx, i, j)
    const REAL pa = C(coef1, ip, iw - 1) + (C(coef1, ip + 1, iw - 1) - C(coef1, ip, iw - 1)) * fp;
    const REAL pb = C(coef1, ip, iw) + (C(coef1, ip + 1, iw) - C(coef1, ip, iw)) * fp;
    const REAL pc = C(coef1, ip, iw + 1) + (C(coef1, ip + 1, iw + 1) - C(coef1, ip, iw + 1)) * fp;
    const REAL ax = ((pc + pa) * fw + (pc - pa)) * fw * (REAL)0.5 + pb * ((REAL)1. - fw * fw);
    const REAL ba = C(coef2, ip, iw) + (C(coef2, ip + 1, iw) - C(coef2, ip, iw)) * fp;
    const REAL bb = C(coef2, ip, iw + 1) + (C(coef2, ip + 1, iw + 1) - C(coef2, ip, iw + 1)) * fp;
    const REAL t1 = ba + (bb - ba) * fw;
    const REAL ca = C(coef3, ip, iw) + (C(coef3, ip + 1, iw) - C(coef3, ip, iw)) * fp;
    const REAL cb = C(coef3, ip, iw + 1) + (C(coef3, ip + 1, iw + 1) - C(coef3, ip, iw + 1)) * fp;
    const REAL t2 = ca + (cb - ca) * fw;
#undef C
    REAL xx = ax + (t1 + t2 * x3) * x3;
    if (xx > (REAL)0.9999999) xx = (REAL)0.9999999;
    if (xx < (REAL)0.0000001) xx = (REAL)0.0000001;
    *tran = *tran * xx;
}

/* SORTIT / mkicx (irrad.F90:2729-2781): insertion sort of the layer list of one super-layer by increasing enn */
static void SFX(ch_sortit)(const REAL *enn, int *lst, int *nc, int ibg, int iend)
{
    *nc = enn[ibg] > 0 ? 1 : 0;
    for (int l = ibg + 1; l <= iend; l++) {
        const REAL eno = enn[lst[l]];
        if (eno > 0) *nc = *nc + 1;
        const int ll = lst[l];
        int i = l - 1;
        while (i > ibg - 1) {
            if (enn[lst[i]] <= eno) break;
            lst[i + 1] = lst[i];
            i--;
        }
        lst[i + 1] = ll;
    }
}

/* cldovlp (irrad.F90:2513-2601) */
static void SFX(ch_cldovlp)(int np, int k1, int k2, int ict, int icb, const int *icx, const int *ncld, const REAL *enn, const REAL *ett,
                            REAL *cldhi, REAL *cldmd, REAL *cldlw)
{
    const int km = k2 - 1;
    REAL *c; int kx, kb, ke;
    if (km < ict) { c = cldhi; kx = ncld[0]; kb = ict - kx; ke = ict - 1; }
    else if (km >= ict && km < icb) { c = cldmd; kx = ncld[1]; kb = icb - kx; ke = icb - 1; }
    else { c = cldlw; kx = ncld[2]; kb = np + 1 - kx; ke = np; }
    if (kx == 1 || *c == 0) { *c = enn[km]; return; }
    *c = 0;
    if (kx != 0)
        for (int k = kb; k <= ke; k++) {
            const int j = icx[k];
            if (j >= k1 && j <= km) *c = enn[j] + ett[j] * *c;
        }
}

/* getirtau (getirtau.code:1-102): cloud optical thickness, scattering scaling, diffuse transmittance, N */
static void SFX(ch_getirtau)(int ib, int np, const REAL *dp_pa /*1..np*/, const REAL *fcld, const REAL *reff /*[l][k]*/,
                             const REAL *hyd /*[l][k]*/, REAL *taudiag /*[l][k]*/, REAL *tcldlyr /*0..np*/, REAL *enn)
{
    const SFX(chou_tables_t) *t = &SFX(CH);
    const REAL *aib = t->aib_ir + 3 * (ib - 1), *awb = t->awb_ir + 4 * (ib - 1), *aiw = t->aiw_ir + 4 * (ib - 1),
               *aww = t->aww_ir + 4 * (ib - 1), *aig = t->aig_ir + 4 * (ib - 1), *awg = t->awg_ir + 4 * (ib - 1);
    const int n1 = np + 1;
#define RE(l) reff[(l - 1) * n1 + k]
#define HY(l) hyd[(l - 1) * n1 + k]
    tcldlyr[0] = 1; enn[0] = 0;
    for (int k = 1; k <= np; k++) {
        const REAL wp = (dp_pa[k] * (REAL)1.0e3) / CH_GRAV;
        REAL tau1, tau2, tau3, tau4;
        if (RE(1) <= 0) tau1 = 0;
        else tau1 = (wp * HY(1)) * (aib[0] + aib[1] / POW(RE(1), aib[2]));
        tau2 = (wp * HY(2)) * (awb[0] + (awb[1] + (awb[2] + awb[3] * RE(2)) * RE(2)) * RE(2));
        tau3 = (REAL)0.00307 * (wp * HY(3));
        const REAL rs = RE(4) < (REAL)112.0 ? RE(4) : (REAL)112.0;
        if (rs <= 0) tau4 = 0;
        else tau4 = (wp * HY(4)) * (aib[0] + aib[1] / POW(rs, aib[2]));
        taudiag[0 * n1 + k] = tau1; taudiag[1 * n1 + k] = tau2; taudiag[2 * n1 + k] = tau3; taudiag[3 * n1 + k] = tau4;
        REAL tauc = tau1 + tau2 + tau3 + tau4;
        if (tauc > (REAL)0.02 && fcld[k] > (REAL)0.01) {
            const REAL w1 = tau1 * (aiw[0] + (aiw[1] + (aiw[2] + aiw[3] * RE(1)) * RE(1)) * RE(1));
            const REAL w2 = tau2 * (aww[0] + (aww[1] + (aww[2] + aww[3] * RE(2)) * RE(2)) * RE(2));
            const REAL w3 = tau3 * (REAL)0.54;
            const REAL w4 = tau4 * (aiw[0] + (aiw[1] + (aiw[2] + aiw[3] * rs) * rs) * rs);
            const REAL ww = (w1 + w2 + w3 + w4) / tauc;
            const REAL g1 = w1 * (aig[0] + (aig[1] + (aig[2] + aig[3] * RE(1)) * RE(1)) * RE(1));
            const REAL g2 = w2 * (awg[0] + (awg[1] + (awg[2] + awg[3] * RE(2)) * RE(2)) * RE(2));
            const REAL g3 = w3 * (REAL)0.95;
            const REAL g4 = w4 * (aig[0] + (aig[1] + (aig[2] + aig[3] * rs) * rs) * rs);
            REAL gg;
            if (w1 + w2 + w3 + w4 != 0) gg = (g1 + g2 + g3 + g4) / (w1 + w2 + w3 + w4); else gg = (REAL)0.5;
            const REAL ff = (REAL)0.5 + ((REAL)0.3739 + ((REAL)0.0076 + (REAL)0.1185 * gg) * gg) * gg;
            REAL sc = (REAL)1. - ww * ff; if (sc < 0) sc = 0;
            tauc = sc * tauc;
            tcldlyr[k] = EXP((REAL)-1.66 * tauc);
            enn[k] = fcld[k] * ((REAL)1.0 - tcldlyr[k]);
        } else { tcldlyr[k] = 1; enn[k] = 0; }
    }
#undef RE
#undef HY
}

/* sfcflux (irrad.F90:2608-2720) */
static void SFX(ch_sfcflux)(int ibn, int m, int i, int ns, const REAL *fs, const REAL *tg, const REAL *eg, const REAL *tv, const REAL *ev,
                            const REAL *rv, REAL *bs, REAL *dbs, REAL *rflxs)
{
#define S2(a, j) a[(size_t)(j - 1) * m + i]
#define S3(a, j) a[((size_t)(ibn - 1) * ns + (j - 1)) * m + i]
    REAL bg[16], bv[16], dbg[16], dbv[16];
    for (int j = 1; j <= ns; j++) {
        bg[j] = SFX(ch_planck)(ibn, S2(tg, j)); bv[j] = SFX(ch_planck)(ibn, S2(tv, j));
        dbg[j] = SFX(ch_plancd)(ibn, S2(tg, j)); dbv[j] = SFX(ch_plancd)(ibn, S2(tv, j));
    }
    if (S2(fs, 1) > (REAL)0.9999) {
        if (S3(ev, 1) < (REAL)0.0001 && S3(rv, 1) < (REAL)0.0001) {
            *bs = S3(eg, 1) * bg[1]; *dbs = S3(eg, 1) * dbg[1]; *rflxs = (REAL)1.0 - S3(eg, 1);
        } else {
            REAL xx = S3(ev, 1) * bv[1];
            const REAL yy = (REAL)1.0 - S3(ev, 1) - S3(rv, 1), zz = (REAL)1.0 - S3(eg, 1);
            *bs = yy * (S3(eg, 1) * bg[1] + zz * xx) + xx;
            xx = S3(ev, 1) * dbv[1];
            *dbs = yy * (S3(eg, 1) * dbg[1] + zz * xx) + xx;
            *rflxs = S3(rv, 1) + zz * yy * yy / ((REAL)1.0 - S3(rv, 1) * zz);
        }
    } else {
        *bs = 0; *dbs = 0; *rflxs = 0;
        if (S3(ev, 1) < (REAL)0.0001 && S3(rv, 1) < (REAL)0.0001) {
            for (int j = 1; j <= ns; j++) {
                *bs = *bs + S2(fs, j) * S3(eg, j) * bg[j];
                *dbs = *dbs + S2(fs, j) * S3(eg, j) * dbg[j];
                *rflxs = *rflxs + S2(fs, j) * ((REAL)1.0 - S3(eg, j));
            }
        } else {
            for (int j = 1; j <= ns; j++) {
                REAL xx = S3(ev, j) * bv[j];
                const REAL yy = (REAL)1.0 - S3(ev, j) - S3(rv, j), zz = (REAL)1.0 - S3(eg, j);
                *bs = *bs + S2(fs, j) * (yy * (S3(eg, j) * bg[j] + zz * xx) + xx);
                xx = S3(ev, j) * dbv[j];
                *dbs = *dbs + S2(fs, j) * (yy * (S3(eg, j) * dbg[j] + zz * xx) + xx);
                *rflxs = *rflxs + S2(fs, j) * (S3(rv, j) + zz * yy * yy / ((REAL)1.0 - S3(rv, j) * zz));
            }
        }
    }
#undef S2
#undef S3
}

/* layer emission pair (irrad.F90:898-905 pattern): effective Planck functions of a layer with transmittance `tr` */
static void SFX(ch_emis)(REAL tr, REAL bl0, REAL bl1, REAL *dn, REAL *up)
{
    REAL yy = tr < (REAL)0.9999 ? tr : (REAL)0.9999;
    if (yy < (REAL)0.00001) yy = (REAL)0.00001;
    const REAL xx = (bl0 - bl1) / LOG(yy);
    *dn = (bl1 - bl0 * yy) / ((REAL)1.0 - yy) - xx;
    *up = (bl0 + bl1) - *dn;
}

/* irrad (irrad.F90:27-1338).  taua/ssaa/asya are INOUT as in the reference (rescaled in place, :660-675).
 * Returns 0; 2 if ns > 15. */
int SFX(oracle_irrad)(int m, int np, const REAL *ple, const REAL *ta, const REAL *wa, const REAL *oa, const REAL *tb, REAL co2,
                      int trace, const REAL *n2o, const REAL *ch4, const REAL *cfc11, const REAL *cfc12, const REAL *cfc22,
                      const REAL *cwc, const REAL *fcld, int ict, int icb, const REAL *reff, int ns, const REAL *fs, const REAL *tg,
                      const REAL *eg, const REAL *tv, const REAL *ev, const REAL *rv, int na, int nb, REAL *taua, REAL *ssaa,
                      REAL *asya, REAL *flxu, REAL *flcu, REAL *flau, REAL *flxau, REAL *flxd, REAL *flcd, REAL *flad, REAL *flxad,
                      REAL *dfdts, REAL *sfcem, REAL *taudiag)
{
    const SFX(chou_tables_t) *t = &SFX(CH);
    if (ns > 15) return 2;
    const int n1 = np + 1, n2 = np + 2;
    const size_t cl = (size_t)m * np;
    /* work arrays, 0-based index = the reference's index */
    REAL *W = (REAL *)calloc((size_t)64 * n2 + (size_t)17 * n1 + 16 * n1, sizeof(REAL));
    REAL *p = W;
#define TAKE(n) (p += (n), p - (n))
    REAL *pa = TAKE(n2), *dt = TAKE(n2), *dp = TAKE(n2), *dp_pa = TAKE(n2), *dh2o = TAKE(n2), *dcont = TAKE(n2), *dco2 = TAKE(n2),
         *do3 = TAKE(n2), *dn2o = TAKE(n2), *dch4 = TAKE(n2), *df11 = TAKE(n2), *df12 = TAKE(n2), *df22 = TAKE(n2);
    REAL *blayer = TAKE(n2), *blevel = TAKE(n2), *dd = TAKE(n2), *du = TAKE(n2), *cd = TAKE(n2), *cu = TAKE(n2), *bd = TAKE(n2),
         *bu = TAKE(n2), *ad = TAKE(n2), *au = TAKE(n2);
    REAL *transfc = TAKE(n2), *transfca = TAKE(n2), *trantcr = TAKE(n2), *trantca = TAKE(n2);
    REAL *flau_c = TAKE(n2), *flad_c = TAKE(n2), *flcu_c = TAKE(n2), *flcd_c = TAKE(n2), *flxu_c = TAKE(n2), *flxd_c = TAKE(n2),
         *flxau_c = TAKE(n2), *flxad_c = TAKE(n2);
    REAL *taerlyr = TAKE(n2), *enn = TAKE(n2), *tcldlyr = TAKE(n2), *fcld_c = TAKE(n2);
    REAL *exptbl = TAKE((size_t)17 * n1);              /* exptbl[(j-1)*n1 + k], k = 0..np, j = 1..17 */
    REAL *reff_c = TAKE((size_t)4 * n1), *cwc_c = TAKE((size_t)4 * n1), *taud = TAKE((size_t)4 * n1);
#undef TAKE
    int *icx = (int *)calloc(n2, sizeof(int));
#define EX(k, j) exptbl[(size_t)((j) - 1) * n1 + (k)]
#define A2(a, k) a[(size_t)((k) - 1) * m + i]               /* Fortran a(i,k) */

    for (int i = 0; i < m; i++) {
        for (int k = 1; k <= np; k++) {
            pa[k] = (REAL)0.5 * (A2(ple, k + 1) + A2(ple, k)) * (REAL)0.01;
            dp[k] = (A2(ple, k + 1) - A2(ple, k)) * (REAL)0.01;
            dp_pa[k] = A2(ple, k + 1) - A2(ple, k);
            dt[k] = A2(ta, k) - (REAL)250.0;
            dh2o[k] = (REAL)1.02 * A2(wa, k) * dp[k];
            do3[k] = (REAL)476. * A2(oa, k) * dp[k];
            dco2[k] = (REAL)789. * co2 * dp[k];
            dch4[k] = (REAL)789. * A2(ch4, k) * dp[k];
            dn2o[k] = (REAL)789. * A2(n2o, k) * dp[k];
            df11[k] = (REAL)789. * A2(cfc11, k) * dp[k];
            df12[k] = (REAL)789. * A2(cfc12, k) * dp[k];
            df22[k] = (REAL)789. * A2(cfc22, k) * dp[k];
            if (dh2o[k] < (REAL)1.e-10) dh2o[k] = (REAL)1.e-10;
            if (do3[k] < (REAL)1.e-6) do3[k] = (REAL)1.e-6;
            if (dco2[k] < (REAL)1.e-4) dco2[k] = (REAL)1.e-4;
            const REAL xx = pa[k] * (REAL)0.001618 * A2(wa, k) * A2(wa, k) * dp[k];
            dcont[k] = xx * EXP((REAL)1800. / A2(ta, k) - (REAL)6.081);
            fcld_c[k] = A2(fcld, k);
            for (int l = 0; l < 4; l++) {
                reff_c[l * n1 + k] = reff[((size_t)l * np + (k - 1)) * m + i];
                cwc_c[l * n1 + k] = cwc[((size_t)l * np + (k - 1)) * m + i];
            }
        }
        /* layer 0 above the model top (:432-453) */
        dp[0] = A2(ple, 1) * (REAL)0.01 > (REAL)0.005 ? A2(ple, 1) * (REAL)0.01 : (REAL)0.005;
        pa[0] = (REAL)0.5 * dp[0];
        dt[0] = A2(ta, 1) - (REAL)250.0;
        dh2o[0] = (REAL)1.02 * A2(wa, 1) * dp[0];
        do3[0] = (REAL)476. * A2(oa, 1) * dp[0];
        dco2[0] = (REAL)789. * co2 * dp[0];
        dch4[0] = (REAL)789. * A2(ch4, 1) * dp[0];
        dn2o[0] = (REAL)789. * A2(n2o, 1) * dp[0];
        df11[0] = (REAL)789. * A2(cfc11, 1) * dp[0];
        df12[0] = (REAL)789. * A2(cfc12, 1) * dp[0];
        df22[0] = (REAL)789. * A2(cfc22, 1) * dp[0];
        if (dh2o[0] < (REAL)1.e-10) dh2o[0] = (REAL)1.e-10;
        if (do3[0] < (REAL)1.e-6) do3[0] = (REAL)1.e-6;
        if (dco2[0] < (REAL)1.e-4) dco2[0] = (REAL)1.e-4;
        {
            const REAL xx = pa[0] * (REAL)0.001618 * A2(wa, 1) * A2(wa, 1) * dp[0];
            dcont[0] = xx * EXP((REAL)1800. / A2(ta, 1) - (REAL)6.081);
        }
        sfcem[i] = 0;
        transfc[np + 1] = 1; transfca[np + 1] = 1; trantcr[np + 1] = 1; trantca[np + 1] = 1;
        for (int k = 1; k <= np + 1; k++) {
            A2(flxu, k) = 0; A2(flxau, k) = 0; A2(flcu, k) = 0; A2(flau, k) = 0; A2(flxd, k) = 0; A2(flxad, k) = 0; A2(flcd, k) = 0;
            A2(flad, k) = 0; A2(dfdts, k) = 0;
        }
        for (int l = 0; l < 10; l++) for (int k = 1; k <= np; k++) taudiag[((size_t)l * np + (k - 1)) * m + i] = 0;

        for (int ibn = 1; ibn <= 10; ibn++) {
            /* the reference `return`s here (irrad.F90:478), i.e. stops after the first column when trace is false; GEOS always
             * passes trace = .true. (GEOS_IrradGridComp.F90:1487) -- band 10 is simply skipped here */
            if (ibn == 10 && !trace) break;
            {   /* test hook: ORACLE_CHOU_BAND=n keeps band n only (per-band comparison with the tables of the technical memoranda) */
                const char *only = getenv("ORACLE_CHOU_BAND");
                if (only && atoi(only) != ibn) continue;
            }
            const int h2otable = ibn == 1 || ibn == 2 || ibn == 8, conbnd = ibn >= 2 && ibn <= 7, co2bnd = ibn == 3, oznbnd = ibn == 5,
                      n2obnd = ibn == 6 || ibn == 7, ch4bnd = n2obnd, combnd = ibn == 4 || ibn == 5, f11bnd = combnd,
                      f12bnd = ibn == 4 || ibn == 6, f22bnd = f12bnd, b10bnd = ibn == 10, do_aerosol = na > 0;
            memset(exptbl, 0, (size_t)17 * n1 * sizeof(REAL));
            /* packing of the exponential tables by band (:501-566) */
            int h2o_s = 0, con_s = 0, co2_s = 0, n2o_s = 0, ch4_s = 0, com_s = 0, f11_s = 0, f12_s = 0, f22_s = 0;
            switch (ibn) {
                case 2: con_s = 1; break;
                case 3: h2o_s = 1; con_s = 7; break;
                case 4: h2o_s = 1; con_s = 7; com_s = 8; f11_s = 14; f12_s = 15; f22_s = 16; break;
                case 5: h2o_s = 1; con_s = 7; com_s = 8; f11_s = 14; break;
                case 6: h2o_s = 1; con_s = 7; n2o_s = 8; ch4_s = 12; f12_s = 16; f22_s = 17; break;
                case 7: h2o_s = 1; con_s = 7; n2o_s = 8; ch4_s = 12; break;
                case 9: h2o_s = 1; break;
                case 10: h2o_s = 1; con_s = 6; co2_s = 7; n2o_s = 13; break;
                default: break;
            }
            for (int k = 1; k <= np; k++) blayer[k] = SFX(ch_planck)(ibn, A2(ta, k));
            blayer[0] = blayer[1]; blevel[0] = blayer[1];
            REAL bs, dbs, rflxs;
            SFX(ch_sfcflux)(ibn, m, i, ns, fs, tg, eg, tv, ev, rv, &bs, &dbs, &rflxs);
            blayer[np + 1] = bs;
            for (int k = 2; k <= np; k++) blevel[k] = (blayer[k - 1] * dp[k] + blayer[k] * dp[k - 1]) / (dp[k - 1] + dp[k]);
            blevel[1] = blayer[1] + (blayer[1] - blayer[2]) * dp[1] / (dp[1] + dp[2]);
            blevel[0] = blevel[1];
            blevel[np + 1] = SFX(ch_planck)(ibn, tb[i]);
            SFX(ch_getirtau)(ibn, np, dp_pa, fcld_c, reff_c, cwc_c, taud, tcldlyr, enn);
            for (int k = 1; k <= np; k++)
                taudiag[((size_t)(ibn - 1) * np + (k - 1)) * m + i] += taud[0 * n1 + k] + taud[1 * n1 + k] + taud[2 * n1 + k] + taud[3 * n1 + k];
            int ncld[3];
            for (int k = 0; k <= np; k++) icx[k] = k;
            SFX(ch_sortit)(enn, icx, &ncld[0], 0, ict - 1);
            SFX(ch_sortit)(enn, icx, &ncld[1], ict, icb - 1);
            SFX(ch_sortit)(enn, icx, &ncld[2], icb, np);
            /* aerosol scaling, in place as in the reference (:655-678) */
            if (do_aerosol) {
                taerlyr[0] = 1;
                for (int k = 1; k <= np; k++) {
                    const size_t j = ((size_t)(ibn - 1) * np + (k - 1)) * m + i;
                    taerlyr[k] = 1;
                    if (taua[j] > (REAL)0.001) {
                        if (ssaa[j] > (REAL)0.001) {
                            asya[j] = asya[j] / ssaa[j];
                            ssaa[j] = ssaa[j] / taua[j];
                            const REAL ff = (REAL).5 + ((REAL).3739 + ((REAL)0.0076 + (REAL)0.1185 * asya[j]) * asya[j]) * asya[j];
                            taua[j] = taua[j] * ((REAL)1. - ssaa[j] * ff);
                        }
                        taerlyr[k] = EXP((REAL)-1.66 * taua[j]);
                    }
                }
            }
            /* exponentials of the k-distribution terms per layer (:684-780; helpers :1379-1884) */
            if (!h2otable && !b10bnd) {
                for (int k = 0; k <= np; k++) {
                    REAL xh = dh2o[k] * POW(pa[k] / (REAL)500., t->pm[ibn - 1]) * ((REAL)1. + (t->aw[ibn - 1] + t->bw[ibn - 1] * dt[k]) * dt[k]);
                    EX(k, h2o_s) = EXP(-xh * t->xkw[ibn - 1]);
                    for (int ik = 2; ik <= 6; ik++) {
                        const REAL e = EX(k, h2o_s + ik - 2);
                        const int mwv = t->mw[ibn - 1];
                        if (mwv == 6) { xh = e * e; EX(k, h2o_s + ik - 1) = xh * xh * xh; }
                        else if (mwv == 8) { xh = e * e; xh = xh * xh; EX(k, h2o_s + ik - 1) = xh * xh; }
                        else if (mwv == 9) { xh = e * e * e; EX(k, h2o_s + ik - 1) = xh * xh * xh; }
                        else { xh = e * e; xh = xh * xh; xh = xh * xh; EX(k, h2o_s + ik - 1) = xh * xh; }
                    }
                }
            }
            int ne = 0;
            if (conbnd) {
                ne = 1; if (ibn == 3) ne = 3;
                for (int k = 0; k <= np; k++) {
                    EX(k, con_s) = EXP(-dcont[k] * t->xke[ibn - 1]);
                    if (ibn == 3) { EX(k, con_s + 1) = EX(k, con_s) * EX(k, con_s); EX(k, con_s + 2) = EX(k, con_s + 1) * EX(k, con_s + 1); }
                }
            }
            if (trace) {
                if (n2obnd)
                    for (int k = 0; k <= np; k++) {
                        if (ibn == 6) {
                            REAL xc = dn2o[k] * ((REAL)1. + ((REAL)1.9297e-3 + (REAL)4.3750e-6 * dt[k]) * dt[k]);
                            EX(k, n2o_s) = EXP(-xc * (REAL)6.31582e-2);
                            xc = EX(k, n2o_s) * EX(k, n2o_s) * EX(k, n2o_s);
                            const REAL xc1 = xc * xc, xc2 = xc1 * xc1;
                            EX(k, n2o_s + 1) = xc * xc1 * xc2;
                        } else {
                            REAL xc = dn2o[k] * POW(pa[k] / (REAL)500.0, (REAL)0.48) * ((REAL)1. + ((REAL)1.3804e-3 + (REAL)7.4838e-6 * dt[k]) * dt[k]);
                            EX(k, n2o_s) = EXP(-xc * (REAL)5.35779e-2);
                            for (int q = 1; q <= 3; q++) { xc = EX(k, n2o_s + q - 1) * EX(k, n2o_s + q - 1); xc = xc * xc; EX(k, n2o_s + q) = xc * xc; }
                        }
                    }
                if (ch4bnd)
                    for (int k = 0; k <= np; k++) {
                        if (ibn == 6) {
                            const REAL xc = dch4[k] * ((REAL)1. + ((REAL)1.7007e-2 + (REAL)1.5826e-4 * dt[k]) * dt[k]);
                            EX(k, ch4_s) = EXP(-xc * (REAL)5.80708e-3);
                        } else {
                            REAL xc = dch4[k] * POW(pa[k] / (REAL)500.0, (REAL)0.65) * ((REAL)1. + ((REAL)5.9590e-4 - (REAL)2.2931e-6 * dt[k]) * dt[k]);
                            EX(k, ch4_s) = EXP(-xc * (REAL)6.29247e-2);
                            for (int q = 1; q <= 3; q++) {
                                xc = EX(k, ch4_s + q - 1) * EX(k, ch4_s + q - 1) * EX(k, ch4_s + q - 1); xc = xc * xc; EX(k, ch4_s + q) = xc * xc;
                            }
                        }
                    }
                if (combnd)
                    for (int k = 0; k <= np; k++) {
                        REAL xc;
                        if (ibn == 4) xc = dco2[k] * ((REAL)1. + ((REAL)3.5775e-2 + (REAL)4.0447e-4 * dt[k]) * dt[k]);
                        else xc = dco2[k] * ((REAL)1. + ((REAL)3.4268e-2 + (REAL)3.7401e-4 * dt[k]) * dt[k]);
                        EX(k, com_s) = EXP(-xc * (REAL)1.922e-7);
                        for (int ik = 2; ik <= 6; ik++) { xc = EX(k, com_s + ik - 2) * EX(k, com_s + ik - 2); xc = xc * xc; EX(k, com_s + ik - 1) = xc * EX(k, com_s + ik - 2); }
                    }
                /* CFCs, Table 7 (:723-766): band 4 uses (a1,b1,fk1), the other band (a2,b2,fk2) */
                static const double cf11[6] = {1.26610e-3, 3.55940e-6, 1.89736e+1, 8.19370e-4, 4.67810e-6, 1.01487e+1};
                static const double cf12[6] = {8.77370e-4, -5.88440e-6, 1.58104e+1, 8.62000e-4, -4.22500e-6, 3.70107e+1};
                static const double cf22[6] = {9.65130e-4, 1.31280e-5, 6.18536e+0, -3.00010e-5, 5.25010e-7, 3.27912e+1};
                for (int q = 0; q < 3; q++) {
                    const int on = q == 0 ? f11bnd : (q == 1 ? f12bnd : f22bnd), s = q == 0 ? f11_s : (q == 1 ? f12_s : f22_s);
                    const double *c = q == 0 ? cf11 : (q == 1 ? cf12 : cf22);
                    const REAL *dcfc = q == 0 ? df11 : (q == 1 ? df12 : df22);
                    if (!on) continue;
                    const int o = ibn == 4 ? 0 : 3;
                    for (int k = 0; k <= np; k++) {
                        const REAL xf = dcfc[k] * ((REAL)1. + ((REAL)c[o] + (REAL)c[o + 1] * dt[k]) * dt[k]);
                        EX(k, s) = EXP(-xf * (REAL)c[o + 2]);
                    }
                }
                if (b10bnd)
                    for (int k = 0; k <= np; k++) {
                        REAL xx = dh2o[k] * (pa[k] / (REAL)500.0) * ((REAL)1. + ((REAL)0.0149 + (REAL)6.20e-5 * dt[k]) * dt[k]);
                        EX(k, h2o_s) = EXP(-xx * (REAL)0.10624);
                        for (int q = 1; q <= 4; q++) { xx = EX(k, h2o_s + q - 1) * EX(k, h2o_s + q - 1); xx = xx * xx; EX(k, h2o_s + q) = xx * xx; }
                        EX(k, con_s) = EXP(-dcont[k] * (REAL)109.0);
                        xx = dco2[k] * POW(pa[k] / (REAL)300.0, (REAL)0.5) * ((REAL)1. + ((REAL)0.0179 + (REAL)1.02e-4 * dt[k]) * dt[k]);
                        EX(k, co2_s) = EXP(-xx * (REAL)2.656e-5);
                        for (int q = 1; q <= 5; q++) { xx = EX(k, co2_s + q - 1) * EX(k, co2_s + q - 1); xx = xx * xx; EX(k, co2_s + q) = xx * xx; }
                        xx = dn2o[k] * ((REAL)1. + ((REAL)1.4476e-3 + (REAL)3.6656e-6 * dt[k]) * dt[k]);
                        EX(k, n2o_s) = EXP(-xx * (REAL)0.25238);
                        xx = EX(k, n2o_s) * EX(k, n2o_s);
                        REAL xx1 = xx * xx; xx1 = xx1 * xx1;
                        const REAL xx2 = xx1 * xx1, xx3 = xx2 * xx2;
                        EX(k, n2o_s + 1) = xx * xx1 * xx2 * xx3;
                    }
            }
            bu[0] = 0; bd[0] = blayer[1]; bu[np + 1] = blayer[np + 1];
            au[0] = 0; ad[0] = blayer[1]; au[np + 1] = blayer[np + 1];
            cu[0] = 0; cd[0] = blayer[1]; cu[np + 1] = blayer[np + 1];
            du[0] = 0; dd[0] = blayer[1]; du[np + 1] = blayer[np + 1];

            /* transmittance of one layer km added to the running state (the shared body of loops 1500 and 3000) */
            REAL th2o[6], tcon[3], tco2[6], tn2o[4], tch4[4], tcom[6], tf11 = 1, tf12 = 1, tf22 = 1, x1, x2, x3;
#define LAYER_TRAN(km, full, trant)                                                                                          \
    do {                                                                                                                      \
        if (h2otable) {                                                                                                       \
            const REAL *ha = ibn == 1 ? t->h11 : (ibn == 2 ? t->h21 : t->h81), *hb = ibn == 1 ? t->h12 : (ibn == 2 ? t->h22 : t->h82),   \
                       *hc = ibn == 1 ? t->h13 : (ibn == 2 ? t->h23 : t->h83);                                                \
            SFX(ch_tablup)(CH_NX, CH_NH, dh2o[km], pa[km], dt[km], &x1, &x2, &x3, *t->w11, *t->p11, *t->dwe, *t->dpe, ha, hb, hc, &trant); \
            if (conbnd) { tcon[0] = tcon[0] * EX(km, con_s); trant = trant * tcon[0]; }                                     \
        } else if (!b10bnd) {                                                                                                 \
            for (int q = 0; q < 6; q++) th2o[q] = th2o[q] * EX(km, h2o_s + q);                                               \
            REAL trn;                                                                                                         \
            if (ne == 0) {                                                                                                    \
                trn = 0; for (int q = 0; q < 6; q++) trn = trn + F2(t->fkw, 6, q + 1, ibn) * th2o[q];                       \
            } else if (ne == 1) {                                                                                             \
                tcon[0] = tcon[0] * EX(km, con_s);                                                                          \
                trn = 0; for (int q = 0; q < 6; q++) trn = trn + F2(t->fkw, 6, q + 1, ibn) * th2o[q];                       \
                trn = trn * tcon[0];                                                                                          \
            } else {                                                                                                          \
                for (int q = 0; q < 3; q++) tcon[q] = tcon[q] * EX(km, con_s + q);                                          \
                trn = 0;                                                                                                      \
                for (int sb = 1; sb <= 3; sb++) {                                                                             \
                    REAL s = 0; for (int q = 0; q < 6; q++) s = s + F2(t->gkw, 6, q + 1, sb) * th2o[q];                     \
                    trn = trn + s * tcon[sb - 1];                                                                             \
                }                                                                                                             \
            }                                                                                                                 \
            trant = trant * trn;                                                                                              \
        }                                                                                                                     \
        if (co2bnd) SFX(ch_tablup)(CH_NX, CH_NC, dco2[km], pa[km], dt[km], &x1, &x2, &x3, *t->w12, *t->p12, *t->dwe, *t->dpe, t->c1, t->c2, t->c3, &trant); \
        if (oznbnd) SFX(ch_tablup)(CH_NX, CH_NO, do3[km], pa[km], dt[km], &x1, &x2, &x3, *t->w13, *t->p13, *t->dwe, *t->dpe, t->oo1, t->oo2, t->oo3, &trant); \
        if ((full) && trace) {                                                                                                \
            if (n2obnd) {                                                                                                     \
                REAL xc;                                                                                                      \
                if (ibn == 6) { tn2o[0] *= EX(km, n2o_s); xc = (REAL)0.940414 * tn2o[0]; tn2o[1] *= EX(km, n2o_s + 1); xc = xc + (REAL)0.059586 * tn2o[1]; } \
                else { static const double w[4] = {0.561961, 0.138707, 0.240670, 0.058662}; xc = 0;                          \
                       for (int q = 0; q < 4; q++) { tn2o[q] *= EX(km, n2o_s + q); xc = xc + (REAL)w[q] * tn2o[q]; } }       \
                trant = trant * xc;                                                                                           \
            }                                                                                                                 \
            if (ch4bnd) {                                                                                                     \
                REAL xc;                                                                                                      \
                if (ibn == 6) { tch4[0] *= EX(km, ch4_s); xc = tch4[0]; }                                                    \
                else { static const double w[4] = {0.610650, 0.280212, 0.107349, 0.001789}; xc = 0;                          \
                       for (int q = 0; q < 4; q++) { tch4[q] *= EX(km, ch4_s + q); xc = xc + (REAL)w[q] * tch4[q]; } }       \
                trant = trant * xc;                                                                                           \
            }                                                                                                                 \
            if (combnd) {                                                                                                     \
                static const double w4[6] = {0.12159, 0.24359, 0.24981, 0.26427, 0.07807, 0.04267};                           \
                static const double w5[6] = {0.06869, 0.14795, 0.19512, 0.33446, 0.17199, 0.08179};                           \
                const double *w = ibn == 4 ? w4 : w5; REAL xc = 0;                                                            \
                for (int q = 0; q < 6; q++) { tcom[q] *= EX(km, com_s + q); xc = xc + (REAL)w[q] * tcom[q]; }                \
                trant = trant * xc;                                                                                           \
            }                                                                                                                 \
            if (f11bnd) { tf11 = tf11 * EX(km, f11_s); trant = trant * tf11; }                                              \
            if (f12bnd) { tf12 = tf12 * EX(km, f12_s); trant = trant * tf12; }                                              \
            if (f22bnd) { tf22 = tf22 * EX(km, f22_s); trant = trant * tf22; }                                              \
            if (b10bnd) {                                                                                                     \
                static const double wh[5] = {0.3153, 0.4604, 0.1326, 0.0798, 0.0119};                                         \
                static const double wc[6] = {0.2673, 0.2201, 0.2106, 0.2409, 0.0196, 0.0415};                                 \
                REAL xx = 0; for (int q = 0; q < 5; q++) { th2o[q] *= EX(km, h2o_s + q); xx = xx + (REAL)wh[q] * th2o[q]; }  \
                trant = xx;                                                                                                   \
                tcon[0] = tcon[0] * EX(km, con_s); trant = trant * tcon[0];                                                 \
                xx = 0; for (int q = 0; q < 6; q++) { tco2[q] *= EX(km, co2_s + q); xx = xx + (REAL)wc[q] * tco2[q]; }       \
                trant = trant * xx;                                                                                           \
                tn2o[0] *= EX(km, n2o_s); xx = (REAL)0.970831 * tn2o[0]; tn2o[1] *= EX(km, n2o_s + 1); xx = xx + (REAL)0.029169 * tn2o[1]; \
                trant = trant * (xx - (REAL)1.0);                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
    } while (0)

            /* loop 1500 (:802-935): emission of the single layer k2-1 */
            for (int k2 = 1; k2 <= np + 1; k2++) {
                if (!h2otable) for (int q = 0; q < 6; q++) th2o[q] = 1;
                tcon[0] = tcon[1] = tcon[2] = 1;
                x1 = 0; x2 = 0; x3 = 0;
                REAL trant = 1;
                const int km = k2 - 1;
                LAYER_TRAN(km, 0, trant);
                const REAL taant = trant;
                if (do_aerosol) trant = trant * taerlyr[km];
                SFX(ch_emis)(((REAL)1. - enn[km]) * trant, blevel[km], blevel[k2], &bd[km], &bu[km]);
                if (do_aerosol) SFX(ch_emis)(((REAL)1. - enn[km]) * taant, blevel[km], blevel[k2], &dd[km], &du[km]);
                else { dd[km] = bd[km]; du[km] = bu[km]; }
                SFX(ch_emis)(trant, blevel[km], blevel[k2], &cd[km], &cu[km]);
                if (do_aerosol) SFX(ch_emis)(taant, blevel[km], blevel[k2], &ad[km], &au[km]);
                else { ad[km] = cd[km]; au[km] = cu[km]; }
            }
            for (int k = 0; k <= np + 1; k++) { flxu_c[k] = 0; flxd_c[k] = 0; flxau_c[k] = 0; flxad_c[k] = 0; flcu_c[k] = 0; flcd_c[k] = 0; flau_c[k] = 0; flad_c[k] = 0; }

            /* loop 2000 (:948-1290): transmittance between levels k1 and k2, fluxes */
            for (int k1 = 0; k1 <= np; k1++) {
                REAL cldlw = 0, cldmd = 0, cldhi = 0, tranal = 1;
                if (!h2otable) for (int q = 0; q < 6; q++) th2o[q] = 1;
                tcon[0] = tcon[1] = tcon[2] = 1;
                if (trace) {
                    if (n2obnd) for (int q = 0; q < 4; q++) tn2o[q] = 1;
                    if (ch4bnd) for (int q = 0; q < 4; q++) tch4[q] = 1;
                    if (combnd) for (int q = 0; q < 6; q++) tcom[q] = 1;
                    if (f11bnd) tf11 = 1;
                    if (f12bnd) tf12 = 1;
                    if (f22bnd) tf22 = 1;
                    if (b10bnd) { for (int q = 0; q < 6; q++) { th2o[q] = 1; tco2[q] = 1; } tcon[0] = 1; for (int q = 0; q < 4; q++) tn2o[q] = 1; }
                }
                x1 = 0; x2 = 0; x3 = 0;
                REAL taant = 1, trant = 1, fclr = 1;
                for (int k2 = k1 + 1; k2 <= np + 1; k2++) {
                    taant = 1; trant = 1; fclr = 1;
                    const int km = k2 - 1;
                    LAYER_TRAN(km, 1, trant);
                    taant = trant;
                    if (do_aerosol) { tranal = tranal * taerlyr[km]; trant = trant * tranal; }
                    if (enn[km] >= (REAL)0.001) SFX(ch_cldovlp)(np, k1, k2, ict, icb, icx, ncld, enn, tcldlyr, &cldhi, &cldmd, &cldlw);
                    fclr = ((REAL)1.0 - cldhi) * ((REAL)1.0 - cldmd) * ((REAL)1.0 - cldlw);
                    if (k2 == k1 + 1 && ibn != 10) {
                        flau_c[k1] -= au[k1]; flad_c[k2] += ad[k1]; flcu_c[k1] -= cu[k1]; flcd_c[k2] += cd[k1];
                        flxu_c[k1] -= bu[k1]; flxd_c[k2] += bd[k1]; flxau_c[k1] -= du[k1]; flxad_c[k2] += dd[k1];
                    }
                    REAL xx = trant * (bu[k2 - 1] - bu[k2]);
                    flxu_c[k1] = flxu_c[k1] + xx * fclr;
                    if (do_aerosol) xx = taant * (du[k2 - 1] - du[k2]);
                    flxau_c[k1] = flxau_c[k1] + xx * fclr;
                    xx = trant * (cu[k2 - 1] - cu[k2]);
                    flcu_c[k1] = flcu_c[k1] + xx;
                    if (do_aerosol) xx = taant * (au[k2 - 1] - au[k2]);
                    flau_c[k1] = flau_c[k1] + xx;
                    if (k1 == 0) xx = -trant * bd[k1]; else xx = trant * (bd[k1 - 1] - bd[k1]);
                    flxd_c[k2] = flxd_c[k2] + xx * fclr;
                    if (do_aerosol) { if (k1 == 0) xx = -taant * dd[k1]; else xx = taant * (dd[k1 - 1] - dd[k1]); }
                    flxad_c[k2] = flxad_c[k2] + xx * fclr;
                    if (k1 == 0) xx = -trant * cd[k1]; else xx = trant * (cd[k1 - 1] - cd[k1]);
                    flcd_c[k2] = flcd_c[k2] + xx;
                    if (do_aerosol) { if (k1 == 0) xx = -taant * ad[k1]; else xx = taant * (ad[k1 - 1] - ad[k1]); }
                    flad_c[k2] = flad_c[k2] + xx;
                }
                trantca[k1] = taant; trantcr[k1] = trant; transfc[k1] = trant * fclr; transfca[k1] = taant * fclr;
                if (k1 > 0) A2(dfdts, k1) = A2(dfdts, k1) - dbs * transfc[k1];
            }
#undef LAYER_TRAN
            if (!b10bnd) {
                flau_c[np + 1] = -blayer[np + 1]; flcu_c[np + 1] = -blayer[np + 1]; flxu_c[np + 1] = -blayer[np + 1]; flxau_c[np + 1] = -blayer[np + 1];
                sfcem[i] = sfcem[i] - blayer[np + 1];
                A2(dfdts, np + 1) = A2(dfdts, np + 1) - dbs;
                for (int k = 1; k <= np + 1; k++) {
                    flau_c[k] = flau_c[k] - flad_c[np + 1] * trantca[k] * rflxs;
                    flcu_c[k] = flcu_c[k] - flcd_c[np + 1] * trantcr[k] * rflxs;
                    flxu_c[k] = flxu_c[k] - flxd_c[np + 1] * transfc[k] * rflxs;
                    flxau_c[k] = flxau_c[k] - flxad_c[np + 1] * transfca[k] * rflxs;
                }
            }
            for (int k = 1; k <= np + 1; k++) {
                A2(flau, k) += flau_c[k]; A2(flcu, k) += flcu_c[k]; A2(flxu, k) += flxu_c[k]; A2(flxau, k) += flxau_c[k];
                A2(flad, k) += flad_c[k]; A2(flcd, k) += flcd_c[k]; A2(flxd, k) += flxd_c[k]; A2(flxad, k) += flxad_c[k];
            }
        }
    }
    (void)cl; (void)nb;
    free(W); free(icx);
    return 0;
#undef EX
#undef A2
}

#undef CH_NX
#undef CH_NO
#undef CH_NC
#undef CH_NH
#undef CH_GRAV
