/* chou_sw_oracle_impl.h -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * Plain-C restatement of the Chou-Suarez shortwave scheme `sorad` (reference: GEOSsolar_GridComp/sorad.F90:43-1588, deledd
 * :1592-1706; cloud optics GEOS_RadiationShared/getvistau.code, getnirtau.code), included once per precision by lw_oracle.c.
 * Non-OVERCAST build (fractional cloud cover, 8 sky situations).
 *
 * PARITY UNPINNED: sorad.F90 `use`s MAPL_ConstantsMod (MAPL is absent from this image and no stand-in is written), and the
 * reference ships no fixtures for it.  Only the coefficient tables are reference data (sorad_constants / rad_constants compiled
 * from the reference and dumped by oracle/ref_glue.F90:ref_chou_sw_dump_tables -> data/chou_sw_*.grtb).  MAPL_GRAV = 9.80665.
 * tests/test_oracle_chou.py holds this restatement to invariants and to consistency with the RRTMG_SW oracle.
 *
 * Conventions of the reference: layers 1..np from the TOP down, pl in hPa, level np+1 = surface; fluxes are fractions of the
 * TOA insolation; API arrays Fortran (m, np[+1][, x]) = column index fastest.  deledd computes in fp64 whatever the default
 * real kind (real(MAPL_R8) temporaries, :1614-1625).
 */

typedef struct {
    const REAL *zk_uv, *wk_uv, *ry_uv, *xk_ir, *ry_ir, *coa, *cah;
    const REAL *aig_uv, *awg_uv, *arg_uv, *aib_uv, *awb_uv, *arb_uv, *aib_nir, *awb_nir, *arb_nir, *aia_nir, *awa_nir, *ara_nir,
        *aig_nir, *awg_nir, *arg_nir, *caib, *caif;
} SFX(chsw_tables_t);
static SFX(chsw_tables_t) SFX(CS);

int SFX(oracle_chou_sw_set_table)(const char *name, const void *p)
{
    SFX(chsw_tables_t) *t = &SFX(CS);
#define SETC(nm) if (!strcmp(name, #nm)) { t->nm = p; return 0; }
    SETC(zk_uv) SETC(wk_uv) SETC(ry_uv) SETC(xk_ir) SETC(ry_ir) SETC(coa) SETC(cah) SETC(aig_uv) SETC(awg_uv) SETC(arg_uv) SETC(aib_uv)
    SETC(awb_uv) SETC(arb_uv) SETC(aib_nir) SETC(awb_nir) SETC(arb_nir) SETC(aia_nir) SETC(awa_nir) SETC(ara_nir) SETC(aig_nir)
    SETC(awg_nir) SETC(arg_nir) SETC(caib) SETC(caif)
#undef SETC
    return 1;
}

#define CS_GRAV ((REAL)9.80665)
#define CS_DSM ((REAL)0.602)

/* deledd (sorad.F90:1592-1706): delta-Eddington layer reflectance / total transmittance / direct transmittance, in fp64 */
static void SFX(cs_deledd)(REAL tau1, REAL ssc1, REAL g01, REAL cza1, REAL *rr1, REAL *tt1, REAL *td1)
{
    double zth = (double)cza1;
    const double g0 = (double)g01, tau = (double)tau1, ssc = (double)ssc1;
    const double ff = g0 * g0;
    double xx = 1.0 - ff * ssc;
    const double taup = tau * xx, sscp = ssc * (1.0 - ff) / xx, gp = g0 / (1.0 + g0);
    xx = 3.0 * gp;
    const double gm1 = (7.0 - sscp * (4.0 + xx)) * 0.25, gm2 = -(1.0 - sscp * (4.0 - xx)) * 0.25;
    const double akk = sqrt((gm1 + gm2) * (gm1 - gm2));
    xx = akk * zth;
    double st7 = 1.0 - xx, st8 = 1.0 + xx, st3 = st7 * st8;
    if (fabs(st3) < 1.e-8) {
        zth = zth + 0.0010;
        if (zth > 1.0) zth = zth - 0.0020;
        xx = akk * zth; st7 = 1.0 - xx; st8 = 1.0 + xx; st3 = st7 * st8;
    }
    const double td = exp(-taup / zth);
    const double gm3 = (2.0 - zth * 3.0 * gp) * 0.25;
    xx = gm1 - gm2;
    const double alf1 = gm1 - gm3 * xx, alf2 = gm2 + gm3 * xx;
    xx = akk * 2.0;
    const double all = (gm3 - alf2 * zth) * xx * td, bll = (1.0 - gm3 + alf1 * zth) * xx;
    xx = akk * gm3;
    const double cll = (alf2 + xx) * st7, dll = (alf2 - xx) * st8;
    xx = akk * (1.0 - gm3);
    const double fll = (alf1 + xx) * st8, ell = (alf1 - xx) * st7;
    const double st2 = exp(-akk * taup), st4 = st2 * st2;
    const double st1 = sscp / ((akk + gm1 + (akk - gm1) * st4) * st3);
    double rr = (cll - dll * st4 - all * st2) * st1;
    double tt = -((fll - ell * st4) * td - bll * st2) * st1;
    if (rr < 0) rr = 0;
    if (tt < 0) tt = 0;
    tt = tt + td;
    *td1 = (REAL)td; *rr1 = (REAL)rr; *tt1 = (REAL)tt;
}

/* getvistau / getnirtau (getvistau.code, getnirtau.code): cloud optical thickness of the 4 hydrometeor species scaled for the
 * maximum-random overlap (beam and diffuse), asymmetry factor [and single-scattering albedo, NIR].  ib = 0: UV/PAR;
 * ib = 1..3: NIR band.  Arrays 1-based over k. */
static void SFX(cs_gettau)(int ib, int np, REAL cosz, const REAL *dp_pa, const REAL *fcld, const REAL *reff /*[l][k]*/,
                           const REAL *hyd, int ict, int icb, REAL *tauclb, REAL *tauclf, REAL *asycl, REAL *ssacl)
{
    const SFX(chsw_tables_t) *t = &SFX(CS);
    const int n1 = np + 1;
    const REAL dm = (REAL)0.1, dt = (REAL)0.30103, da = (REAL)0.1, t1 = (REAL)-0.9031;
#define RE(l) reff[(l - 1) * n1 + k]
#define HY(l) hyd[(l - 1) * n1 + k]
#define CAIB(a, b, c) F3(t->caib, 11, 9, a, b, c)
#define CAIF(a, b) F2(t->caif, 9, a, b)
#define N2(tab, j) F2(tab, 3, ib, j)
    REAL cc[4] = {0, 0, 0, 0};
    if (ict != 0) {
        for (int k = 1; k <= ict - 1; k++) if (fcld[k] > cc[1]) cc[1] = fcld[k];
        for (int k = ict; k <= icb - 1; k++) if (fcld[k] > cc[2]) cc[2] = fcld[k];
        for (int k = icb; k <= np; k++) if (fcld[k] > cc[3]) cc[3] = fcld[k];
    }
    for (int k = 1; k <= np; k++) {
        const REAL wp = (dp_pa[k] * (REAL)1.0e3) / CS_GRAV;
        REAL tc1, tc2, tc3, tc4;
        const REAL rs = RE(4) < (REAL)112.0 ? RE(4) : (REAL)112.0;
        if (ib == 0) {
            tc1 = RE(1) <= 0 ? (REAL)0 : (wp * HY(1)) * *t->aib_uv / RE(1);
            tc2 = RE(2) <= 0 ? (REAL)0 : (wp * HY(2)) * (t->awb_uv[0] + t->awb_uv[1] / RE(2));
            tc3 = (wp * HY(3)) * t->arb_uv[0];
            tc4 = rs <= 0 ? (REAL)0 : (wp * HY(4)) * *t->aib_uv / rs;
        } else {
            tc1 = RE(1) <= 0 ? (REAL)0 : (wp * HY(1)) * *t->aib_nir / RE(1);
            tc2 = RE(2) <= 0 ? (REAL)0 : (wp * HY(2)) * (N2(t->awb_nir, 1) + N2(t->awb_nir, 2) / RE(2));
            tc3 = (wp * HY(3)) * N2(t->arb_nir, 1);
            tc4 = rs <= 0 ? (REAL)0 : (wp * HY(4)) * *t->aib_nir / rs;
        }
        REAL tb[4] = {0, 0, 0, 0}, tf[4] = {0, 0, 0, 0};
        if (ict != 0) {
            const int kk = k < ict ? 1 : (k < icb ? 2 : 3);
            REAL tauc = tc1 + tc2 + tc3 + tc4;
            if (tauc > (REAL)0.02 && fcld[k] > (REAL)0.01) {
                REAL fa;
                if (ib == 0) fa = fcld[k] / cc[kk];
                else fa = cc[kk] != 0 ? fcld[k] / cc[kk] : (REAL)0;
                if (tauc > (REAL)32.) tauc = (REAL)32.;
                REAL fm = cosz / dm, ft = (LOG10(tauc) - t1) / dt;
                fa = fa / da;
                int im = (int)(fm + (REAL)1.5), it = (int)(ft + (REAL)1.5), ia = (int)(fa + (REAL)1.5);
                if (im < 2) im = 2; if (it < 2) it = 2; if (ia < 2) ia = 2;
                if (im > 10) im = 10; if (it > 8) it = 8; if (ia > 10) ia = 10;
                fm = fm - (REAL)(im - 1); ft = ft - (REAL)(it - 1); fa = fa - (REAL)(ia - 1);
                REAL xai = (-CAIB(im - 1, it, ia) * ((REAL)1. - fm) + CAIB(im + 1, it, ia) * ((REAL)1. + fm)) * fm * (REAL).5 +
                           CAIB(im, it, ia) * ((REAL)1. - fm * fm);
                xai = xai + (-CAIB(im, it - 1, ia) * ((REAL)1. - ft) + CAIB(im, it + 1, ia) * ((REAL)1. + ft)) * ft * (REAL).5 +
                      CAIB(im, it, ia) * ((REAL)1. - ft * ft);
                xai = xai + (-CAIB(im, it, ia - 1) * ((REAL)1. - fa) + CAIB(im, it, ia + 1) * ((REAL)1. + fa)) * fa * (REAL).5 +
                      CAIB(im, it, ia) * ((REAL)1. - fa * fa);
                xai = xai - (REAL)2. * CAIB(im, it, ia);
                if (xai < 0) xai = 0;
                if (xai > 1) xai = 1;
                tb[0] = tc1 * xai; tb[1] = tc2 * xai; tb[2] = tc3 * xai; tb[3] = tc4 * xai;
                xai = (-CAIF(it - 1, ia) * ((REAL)1. - ft) + CAIF(it + 1, ia) * ((REAL)1. + ft)) * ft * (REAL).5 + CAIF(it, ia) * ((REAL)1. - ft * ft);
                xai = xai + (-CAIF(it, ia - 1) * ((REAL)1. - fa) + CAIF(it, ia + 1) * ((REAL)1. + fa)) * fa * (REAL).5 + CAIF(it, ia) * ((REAL)1. - fa * fa);
                xai = xai - CAIF(it, ia);
                if (xai < 0) xai = 0;
                if (xai > 1) xai = 1;
                tf[0] = tc1 * xai; tf[1] = tc2 * xai; tf[2] = tc3 * xai; tf[3] = tc4 * xai;
            }
        } else { tb[0] = tf[0] = tc1; tb[1] = tf[1] = tc2; tb[2] = tf[2] = tc3; tb[3] = tf[3] = tc4; }
        tauclb[k] = tb[0] + tb[1] + tb[2] + tb[3];
        tauclf[k] = tf[0] + tf[1] + tf[2] + tf[3];
        REAL asy = 1, ssa = (REAL)0.99999;
        const REAL tauc = tc1 + tc2 + tc3 + tc4;
        if (tauc > (REAL)0.02 && fcld[k] > (REAL)0.01) {
            if (ib == 0) {
                const REAL g1 = (t->aig_uv[0] + (t->aig_uv[1] + t->aig_uv[2] * RE(1)) * RE(1)) * tc1;
                const REAL g2 = (t->awg_uv[0] + (t->awg_uv[1] + t->awg_uv[2] * RE(2)) * RE(2)) * tc2;
                const REAL g3 = t->arg_uv[0] * tc3;
                const REAL g4 = (t->aig_uv[0] + (t->aig_uv[1] + t->aig_uv[2] * rs) * rs) * tc4;
                asy = (g1 + g2 + g3 + g4) / tauc;
            } else {
                const REAL w1 = ((REAL)1. - (N2(t->aia_nir, 1) + (N2(t->aia_nir, 2) + N2(t->aia_nir, 3) * RE(1)) * RE(1))) * tc1;
                const REAL w2 = ((REAL)1. - (N2(t->awa_nir, 1) + (N2(t->awa_nir, 2) + N2(t->awa_nir, 3) * RE(2)) * RE(2))) * tc2;
                const REAL w3 = ((REAL)1. - N2(t->ara_nir, 1)) * tc3;
                const REAL w4 = ((REAL)1. - (N2(t->aia_nir, 1) + (N2(t->aia_nir, 2) + N2(t->aia_nir, 3) * rs) * rs)) * tc4;
                ssa = (w1 + w2 + w3 + w4) / tauc;
                const REAL g1 = (N2(t->aig_nir, 1) + (N2(t->aig_nir, 2) + N2(t->aig_nir, 3) * RE(1)) * RE(1)) * w1;
                const REAL g2 = (N2(t->awg_nir, 1) + (N2(t->awg_nir, 2) + N2(t->awg_nir, 3) * RE(2)) * RE(2)) * w2;
                const REAL g3 = N2(t->arg_nir, 1) * w3;
                /* the reference uses reff(k,4) here, not the capped reff_snow (getnirtau.code) */
                const REAL g4 = (N2(t->aig_nir, 1) + (N2(t->aig_nir, 2) + N2(t->aig_nir, 3) * RE(4)) * RE(4)) * w4;
                if (w1 + w2 + w3 + w4 != 0) asy = (g1 + g2 + g3 + g4) / (w1 + w2 + w3 + w4);
            }
        }
        asycl[k] = asy; ssacl[k] = ssa;
    }
#undef RE
#undef HY
#undef CAIB
#undef CAIF
#undef N2
}

/* CLDFLX (sorad.F90:689-872 = :1217-1397): fluxes of the 8 sky situations of the three cloud groups.
 * rr, tt, td, rs, ts: [ (k)*2 + (i-1) ], k = 0..np+1, i = 1 clear / 2 cloudy portion of the layer. */
static void SFX(cs_cldflx)(int np, int ict, int icb, REAL cc1, REAL cc2, REAL cc3, const REAL *rr, const REAL *tt, const REAL *td,
                           const REAL *rs, const REAL *ts, REAL *W, REAL *fclr, REAL *fall, REAL *fupc, REAL *fupa, REAL *fsdir, REAL *fsdif)
{
    const int n2 = np + 2;
    REAL *tda = W, *tta = W + 4 * n2, *rsa = W + 8 * n2, *rra = W + 12 * n2, *rxa = W + 16 * n2;
#define L2(a, k, i) a[(k) * 2 + (i) - 1]
#define A3(a, k, i, j) a[((k) * 2 + (i) - 1) * 2 + (j) - 1]
    for (int ih = 1; ih <= 2; ih++) {
        A3(tda, 0, ih, 1) = L2(td, 0, ih); A3(tta, 0, ih, 1) = L2(tt, 0, ih); A3(rsa, 0, ih, 1) = L2(rs, 0, ih);
        A3(tda, 0, ih, 2) = L2(td, 0, ih); A3(tta, 0, ih, 2) = L2(tt, 0, ih); A3(rsa, 0, ih, 2) = L2(rs, 0, ih);
        for (int k = 1; k <= ict - 1; k++) {
            const REAL denm = L2(ts, k, ih) / ((REAL)1. - A3(rsa, k - 1, ih, 1) * L2(rs, k, ih));
            A3(tda, k, ih, 1) = A3(tda, k - 1, ih, 1) * L2(td, k, ih);
            A3(tta, k, ih, 1) = A3(tda, k - 1, ih, 1) * L2(tt, k, ih) +
                                (A3(tda, k - 1, ih, 1) * A3(rsa, k - 1, ih, 1) * L2(rr, k, ih) + A3(tta, k - 1, ih, 1) - A3(tda, k - 1, ih, 1)) * denm;
            A3(rsa, k, ih, 1) = L2(rs, k, ih) + L2(ts, k, ih) * A3(rsa, k - 1, ih, 1) * denm;
            A3(tda, k, ih, 2) = A3(tda, k, ih, 1); A3(tta, k, ih, 2) = A3(tta, k, ih, 1); A3(rsa, k, ih, 2) = A3(rsa, k, ih, 1);
        }
        for (int k = ict; k <= icb - 1; k++)
            for (int im = 1; im <= 2; im++) {
                const REAL denm = L2(ts, k, im) / ((REAL)1. - A3(rsa, k - 1, ih, im) * L2(rs, k, im));
                A3(tda, k, ih, im) = A3(tda, k - 1, ih, im) * L2(td, k, im);
                A3(tta, k, ih, im) = A3(tda, k - 1, ih, im) * L2(tt, k, im) +
                                     (A3(tda, k - 1, ih, im) * A3(rsa, k - 1, ih, im) * L2(rr, k, im) + A3(tta, k - 1, ih, im) - A3(tda, k - 1, ih, im)) * denm;
                A3(rsa, k, ih, im) = L2(rs, k, im) + L2(ts, k, im) * A3(rsa, k - 1, ih, im) * denm;
            }
    }
    for (int is = 1; is <= 2; is++) {
        A3(rra, np + 1, 1, is) = L2(rr, np + 1, is); A3(rxa, np + 1, 1, is) = L2(rs, np + 1, is);
        A3(rra, np + 1, 2, is) = L2(rr, np + 1, is); A3(rxa, np + 1, 2, is) = L2(rs, np + 1, is);
        for (int k = np; k >= icb; k--) {
            const REAL denm = L2(ts, k, is) / ((REAL)1. - L2(rs, k, is) * A3(rxa, k + 1, 1, is));
            A3(rra, k, 1, is) = L2(rr, k, is) + (L2(td, k, is) * A3(rra, k + 1, 1, is) + (L2(tt, k, is) - L2(td, k, is)) * A3(rxa, k + 1, 1, is)) * denm;
            A3(rxa, k, 1, is) = L2(rs, k, is) + L2(ts, k, is) * A3(rxa, k + 1, 1, is) * denm;
            A3(rra, k, 2, is) = A3(rra, k, 1, is); A3(rxa, k, 2, is) = A3(rxa, k, 1, is);
        }
        for (int k = icb - 1; k >= ict; k--)
            for (int im = 1; im <= 2; im++) {
                const REAL denm = L2(ts, k, im) / ((REAL)1. - L2(rs, k, im) * A3(rxa, k + 1, im, is));
                A3(rra, k, im, is) = L2(rr, k, im) + (L2(td, k, im) * A3(rra, k + 1, im, is) + (L2(tt, k, im) - L2(td, k, im)) * A3(rxa, k + 1, im, is)) * denm;
                A3(rxa, k, im, is) = L2(rs, k, im) + L2(ts, k, im) * A3(rxa, k + 1, im, is) * denm;
            }
    }
    for (int ih = 1; ih <= 2; ih++) {
        const REAL ch = ih == 1 ? (REAL)1.0 - cc1 : cc1;
        for (int im = 1; im <= 2; im++) {
            const REAL cm = im == 1 ? ch * ((REAL)1.0 - cc2) : ch * cc2;
            for (int is = 1; is <= 2; is++) {
                const REAL ct = is == 1 ? cm * ((REAL)1.0 - cc3) : cm * cc3;
                for (int k = icb; k <= np; k++) {
                    const REAL denm = L2(ts, k, is) / ((REAL)1. - A3(rsa, k - 1, ih, im) * L2(rs, k, is));
                    A3(tda, k, ih, im) = A3(tda, k - 1, ih, im) * L2(td, k, is);
                    A3(tta, k, ih, im) = A3(tda, k - 1, ih, im) * L2(tt, k, is) +
                                         (A3(tda, k - 1, ih, im) * L2(rr, k, is) * A3(rsa, k - 1, ih, im) + A3(tta, k - 1, ih, im) - A3(tda, k - 1, ih, im)) * denm;
                    A3(rsa, k, ih, im) = L2(rs, k, is) + L2(ts, k, is) * A3(rsa, k - 1, ih, im) * denm;
                }
                for (int k = ict - 1; k >= 0; k--) {
                    const REAL denm = L2(ts, k, ih) / ((REAL)1. - L2(rs, k, ih) * A3(rxa, k + 1, im, is));
                    A3(rra, k, im, is) = L2(rr, k, ih) + (L2(td, k, ih) * A3(rra, k + 1, im, is) + (L2(tt, k, ih) - L2(td, k, ih)) * A3(rxa, k + 1, im, is)) * denm;
                    A3(rxa, k, im, is) = L2(rs, k, ih) + L2(ts, k, ih) * A3(rxa, k + 1, im, is) * denm;
                }
                REAL fdndir = 0, fdndif = 0;
                for (int k = 1; k <= np + 1; k++) {
                    const REAL denm = (REAL)1. / ((REAL)1. - A3(rsa, k - 1, ih, im) * A3(rxa, k, im, is));
                    fdndir = A3(tda, k - 1, ih, im);
                    const REAL xx4 = A3(tda, k - 1, ih, im) * A3(rra, k, im, is);
                    const REAL yy = A3(tta, k - 1, ih, im) - A3(tda, k - 1, ih, im);
                    fdndif = (xx4 * A3(rsa, k - 1, ih, im) + yy) * denm;
                    const REAL fupdif = (xx4 + yy * A3(rxa, k, im, is)) * denm;
                    const REAL flxdn = fdndir + fdndif - fupdif;
                    if (ih == 1 && im == 1 && is == 1) { fupc[k] = fupdif; fclr[k] = flxdn; }
                    fupa[k] = fupa[k] + fupdif * ct;
                    fall[k] = fall[k] + flxdn * ct;
                }
                *fsdir = *fsdir + fdndir * ct;
                *fsdif = *fsdif + fdndif * ct;
            }
        }
    }
#undef A3
#undef L2
}

/* sorad (sorad.F90:43-1588).  hk_uv(5), hk_ir(3,10) are inputs as in the reference.  drband/dfband (m,8) may be NULL. */
int SFX(oracle_sorad)(int m, int np, int nb, const REAL *cosz, const REAL *pl, const REAL *ta, const REAL *wa, const REAL *oa, REAL co2,
                      const REAL *cwc, const REAL *fcld, int ict, int icb, const REAL *reff, const REAL *hk_uv, const REAL *hk_ir,
                      const REAL *taua, const REAL *ssaa, const REAL *asya, const REAL *rsuvbm, const REAL *rsuvdf, const REAL *rsirbm,
                      const REAL *rsirdf, REAL *flx, REAL *flc, REAL *fdiruv, REAL *fdifuv, REAL *fdirpar, REAL *fdifpar, REAL *fdirir,
                      REAL *fdifir, REAL *flxu, REAL *flcu, REAL *flx_sfc_band, int do_drfband, REAL *drband, REAL *dfband)
{
    const SFX(chsw_tables_t) *t = &SFX(CS);
    const int n1 = np + 1, n2 = np + 2;
    REAL *W = (REAL *)calloc((size_t)60 * n2, sizeof(REAL));
    REAL *p = W;
#define TAKE(n) (p += (n), p - (n))
    REAL *dp = TAKE(n2), *dp_pa = TAKE(n2), *wh = TAKE(n2), *oh = TAKE(n2), *scal = TAKE(n2), *swh = TAKE(n2), *so2 = TAKE(n2), *df = TAKE(n2);
    REAL *tauclb = TAKE(n2), *tauclf = TAKE(n2), *asycl = TAKE(n2), *ssacl = TAKE(n2), *fcld_c = TAKE(n2);
    REAL *reff_c = TAKE(4 * n1), *cwc_c = TAKE(4 * n1);
    REAL *rr = TAKE(2 * n2), *tt = TAKE(2 * n2), *td = TAKE(2 * n2), *rs = TAKE(2 * n2), *ts = TAKE(2 * n2);
    REAL *fall = TAKE(n2), *fclr = TAKE(n2), *fupa = TAKE(n2), *fupc = TAKE(n2);
    REAL *CW = TAKE(20 * n2);
#undef TAKE
#define A2(a, k) a[(size_t)((k) - 1) * m + i]
#define A3B(a, k, ib) a[((size_t)((ib) - 1) * np + ((k) - 1)) * m + i]
    (void)nb;
    for (int i = 0; i < m; i++) {
        int ntop = 0;
        const REAL cz = cosz[i];
        const REAL snt = (REAL)1.0 / cz;
        const REAL xtoa = A2(pl, 1) > (REAL)1.e-3 ? A2(pl, 1) : (REAL)1.e-3;
        const REAL scal0 = xtoa * POW((REAL)0.5 * xtoa / (REAL)300., (REAL).8);
        const REAL o3toa = (REAL)1.02 * A2(oa, 1) * xtoa * (REAL)466.7 + (REAL)1.0e-8;
        const REAL wvtoa = (REAL)1.02 * A2(wa, 1) * scal0 * ((REAL)1.0 + (REAL)0.00135 * (A2(ta, 1) - (REAL)240.)) + (REAL)1.0e-9;
        swh[1] = wvtoa;
        for (int k = 1; k <= np; k++) {
            dp[k] = A2(pl, k + 1) - A2(pl, k);
            dp_pa[k] = dp[k] * (REAL)100.;
            const REAL pa = (REAL)0.5 * (A2(pl, k) + A2(pl, k + 1));
            scal[k] = dp[k] * POW(pa / (REAL)300., (REAL).8);
            wh[k] = (REAL)1.02 * A2(wa, k) * scal[k] * ((REAL)1. + (REAL)0.00135 * (A2(ta, k) - (REAL)240.)) + (REAL)1.e-9;
            swh[k + 1] = swh[k] + wh[k];
            oh[k] = (REAL)1.02 * A2(oa, k) * dp[k] * (REAL)466.7 + (REAL)1.e-8;
            fcld_c[k] = A2(fcld, k);
            for (int l = 0; l < 4; l++) {
                reff_c[l * n1 + k] = reff[((size_t)l * np + (k - 1)) * m + i];
                cwc_c[l * n1 + k] = cwc[((size_t)l * np + (k - 1)) * m + i];
            }
        }
        memset(rr, 0, 2 * n2 * sizeof(REAL)); memset(tt, 0, 2 * n2 * sizeof(REAL)); memset(td, 0, 2 * n2 * sizeof(REAL));
        memset(rs, 0, 2 * n2 * sizeof(REAL)); memset(ts, 0, 2 * n2 * sizeof(REAL)); memset(CW, 0, 20 * n2 * sizeof(REAL));
        for (int k = 1; k <= np + 1; k++) { A2(flx, k) = 0; A2(flc, k) = 0; A2(flxu, k) = 0; A2(flcu, k) = 0; }
        for (int ib = 1; ib <= 8; ib++) {
            flx_sfc_band[(size_t)(ib - 1) * m + i] = 0;
            if (do_drfband) { drband[(size_t)(ib - 1) * m + i] = 0; dfband[(size_t)(ib - 1) * m + i] = 0; }
        }
        REAL cc1 = 0, cc2 = 0, cc3 = 0;
        for (int k = 1; k <= np; k++) {
            if (k < ict) { if (fcld_c[k] > cc1) cc1 = fcld_c[k]; }
            else if (k < icb) { if (fcld_c[k] > cc2) cc2 = fcld_c[k]; }
            else if (fcld_c[k] > cc3) cc3 = fcld_c[k];
        }
        /* ---- UV + PAR (SOLUV inline, :359-905) ---- */
        fdiruv[i] = 0; fdifuv[i] = 0;
#define L2(a, k, j) a[(k) * 2 + (j) - 1]
        for (int j = 1; j <= 2; j++) {
            L2(rr, np + 1, j) = rsuvbm[i]; L2(rs, np + 1, j) = rsuvdf[i]; L2(td, np + 1, j) = 0; L2(tt, np + 1, j) = 0; L2(ts, np + 1, j) = 0;
            L2(rr, 0, j) = 0; L2(rs, 0, j) = 0; L2(tt, 0, j) = 1; L2(ts, 0, j) = 1;
        }
        SFX(cs_gettau)(0, np, cz, dp_pa, fcld_c, reff_c, cwc_c, ict, icb, tauclb, tauclf, asycl, ssacl);
        for (int ib = 1; ib <= 5; ib++) {
            L2(td, 0, 1) = EXP(-(wvtoa * t->wk_uv[ib - 1] + o3toa * t->zk_uv[ib - 1]) / cz);
            L2(td, 0, 2) = L2(td, 0, 1);
            for (int k = 1; k <= np; k++) {
                const REAL taurs = t->ry_uv[ib - 1] * dp[k], tauoz = t->zk_uv[ib - 1] * oh[k], tauwv = t->wk_uv[ib - 1] * wh[k];
                const REAL tausto = taurs + tauoz + tauwv + A3B(taua, k, ib) + (REAL)1.0e-7;
                const REAL ssatau = A3B(ssaa, k, ib) + taurs;
                const REAL asysto = A3B(asya, k, ib);
                REAL tautob = tausto, asytob = asysto / ssatau, ssatob = ssatau / tautob + (REAL)1.0e-8;
                if (ssatob > (REAL)0.999999) ssatob = (REAL)0.999999;
                REAL rrt, ttt, tdt, rst, tst, dum;
                SFX(cs_deledd)(tautob, ssatob, asytob, cz, &rrt, &ttt, &tdt);
                SFX(cs_deledd)(tautob, ssatob, asytob, CS_DSM, &rst, &tst, &dum);
                L2(rr, k, 1) = rrt; L2(tt, k, 1) = ttt; L2(td, k, 1) = tdt; L2(rs, k, 1) = rst; L2(ts, k, 1) = tst;
                tautob = tausto + tauclb[k];
                ssatob = (ssatau + tauclb[k]) / tautob + (REAL)1.0e-8;
                if (ssatob > (REAL)0.999999) ssatob = (REAL)0.999999;
                asytob = (asysto + asycl[k] * tauclb[k]) / (ssatob * tautob);
                const REAL tautof = tausto + tauclf[k];
                REAL ssatof = (ssatau + tauclf[k]) / tautof + (REAL)1.0e-8;
                if (ssatof > (REAL)0.999999) ssatof = (REAL)0.999999;
                const REAL asytof = (asysto + asycl[k] * tauclf[k]) / (ssatof * tautof);
                SFX(cs_deledd)(tautob, ssatob, asytob, cz, &rrt, &ttt, &tdt);
                SFX(cs_deledd)(tautof, ssatof, asytof, CS_DSM, &rst, &tst, &dum);
                L2(rr, k, 2) = rrt; L2(tt, k, 2) = ttt; L2(td, k, 2) = tdt; L2(rs, k, 2) = rst; L2(ts, k, 2) = tst;
            }
            for (int k = 1; k <= np + 1; k++) { fclr[k] = 0; fall[k] = 0; fupa[k] = 0; fupc[k] = 0; }
            REAL fsdir = 0, fsdif = 0;
            SFX(cs_cldflx)(np, ict, icb, cc1, cc2, cc3, rr, tt, td, rs, ts, CW, fclr, fall, fupc, fupa, &fsdir, &fsdif);
            const REAL hk = hk_uv[ib - 1];
            for (int k = 1; k <= np + 1; k++) {
                A2(flx, k) += fall[k] * hk; A2(flc, k) += fclr[k] * hk; A2(flxu, k) += fupa[k] * hk; A2(flcu, k) += fupc[k] * hk;
            }
            flx_sfc_band[(size_t)(ib - 1) * m + i] += fall[np + 1] * hk;
            if (do_drfband) { drband[(size_t)(ib - 1) * m + i] += fsdir * hk; dfband[(size_t)(ib - 1) * m + i] += fsdif * hk; }
            if (ib < 5) { fdiruv[i] += fsdir * hk; fdifuv[i] += fsdif * hk; }
            else { fdirpar[i] = fsdir * hk; fdifpar[i] = fsdif * hk; }
        }
        /* ---- near IR (SOLIR inline, :907-1423) ---- */
        fdirir[i] = 0; fdifir[i] = 0;
        for (int j = 1; j <= 2; j++) {
            L2(rr, np + 1, j) = rsirbm[i]; L2(rs, np + 1, j) = rsirdf[i]; L2(td, np + 1, j) = 0; L2(tt, np + 1, j) = 0; L2(ts, np + 1, j) = 0;
            L2(rr, 0, j) = 0; L2(rs, 0, j) = 0; L2(tt, 0, j) = 1; L2(ts, 0, j) = 1;
        }
        for (int ib = 1; ib <= 3; ib++) {
            const int iv = ib + 5;
            SFX(cs_gettau)(ib, np, cz, dp_pa, fcld_c, reff_c, cwc_c, ict, icb, tauclb, tauclf, asycl, ssacl);
            for (int ik = 1; ik <= 10; ik++) {
                L2(td, 0, 1) = EXP(-wvtoa * t->xk_ir[ik - 1] / cz);
                L2(td, 0, 2) = L2(td, 0, 1);
                for (int k = 1; k <= np; k++) {
                    const REAL taurs = t->ry_ir[ib - 1] * dp[k], tauwv = t->xk_ir[ik - 1] * wh[k];
                    const REAL tausto = taurs + tauwv + A3B(taua, k, iv) + (REAL)1.0e-7;
                    const REAL ssatau = A3B(ssaa, k, iv) + taurs + (REAL)1.0e-8;
                    const REAL asysto = A3B(asya, k, iv);
                    REAL tautob = tausto, asytob = asysto / ssatau, ssatob = ssatau / tautob + (REAL)1.0e-8;
                    if (ssatob > (REAL)0.999999) ssatob = (REAL)0.999999;
                    REAL rrt, ttt, tdt, rst, tst, dum;
                    SFX(cs_deledd)(tautob, ssatob, asytob, cz, &rrt, &ttt, &tdt);
                    SFX(cs_deledd)(tautob, ssatob, asytob, CS_DSM, &rst, &tst, &dum);
                    L2(rr, k, 1) = rrt; L2(tt, k, 1) = ttt; L2(td, k, 1) = tdt; L2(rs, k, 1) = rst; L2(ts, k, 1) = tst;
                    tautob = tausto + tauclb[k];
                    ssatob = (ssatau + ssacl[k] * tauclb[k]) / tautob + (REAL)1.0e-8;
                    if (ssatob > (REAL)0.999999) ssatob = (REAL)0.999999;
                    asytob = (asysto + asycl[k] * ssacl[k] * tauclb[k]) / (ssatob * tautob);
                    const REAL tautof = tausto + tauclf[k];
                    REAL ssatof = (ssatau + ssacl[k] * tauclf[k]) / tautof + (REAL)1.0e-8;
                    if (ssatof > (REAL)0.999999) ssatof = (REAL)0.999999;
                    const REAL asytof = (asysto + asycl[k] * ssacl[k] * tauclf[k]) / (ssatof * tautof);
                    SFX(cs_deledd)(tautob, ssatob, asytob, cz, &rrt, &ttt, &tdt);
                    SFX(cs_deledd)(tautof, ssatof, asytof, CS_DSM, &rst, &tst, &dum);
                    L2(rr, k, 2) = rrt; L2(tt, k, 2) = ttt; L2(td, k, 2) = tdt; L2(rs, k, 2) = rst; L2(ts, k, 2) = tst;
                }
                for (int k = 1; k <= np + 1; k++) { fclr[k] = 0; fall[k] = 0; fupa[k] = 0; fupc[k] = 0; }
                REAL fsdir = 0, fsdif = 0;
                SFX(cs_cldflx)(np, ict, icb, cc1, cc2, cc3, rr, tt, td, rs, ts, CW, fclr, fall, fupc, fupa, &fsdir, &fsdif);
                const REAL hk = F2(hk_ir, 3, ib, ik);
                for (int k = 1; k <= np + 1; k++) {
                    A2(flx, k) += fall[k] * hk; A2(flc, k) += fclr[k] * hk; A2(flxu, k) += fupa[k] * hk; A2(flcu, k) += fupc[k] * hk;
                }
                fdirir[i] += fsdir * hk; fdifir[i] += fsdif * hk;
                flx_sfc_band[(size_t)(iv - 1) * m + i] += fall[np + 1] * hk;
                if (do_drfband) { drband[(size_t)(iv - 1) * m + i] += fsdir * hk; dfband[(size_t)(iv - 1) * m + i] += fsdif * hk; }
            }
        }
#undef L2
        /* ---- O2 and CO2 flux reductions (:1425-1552) ---- */
        df[0] = 0;
        const REAL cnt = (REAL)165.22 * snt;
        so2[1] = scal0 * cnt;
        df[1] = (REAL)0.0633 * ((REAL)1. - EXP((REAL)-0.000155 * SQRT(so2[1])));
        for (int k = 1; k <= np; k++) {
            so2[k + 1] = so2[k] + scal[k] * cnt;
            df[k + 1] = (REAL)0.0633 * ((REAL)1.0 - EXP((REAL)-0.000155 * SQRT(so2[k + 1])));
        }
        so2[1] = ((REAL)789. * co2) * scal0;
        for (int k = 1; k <= np; k++) so2[k + 1] = so2[k] + ((REAL)789. * co2) * scal[k];
        {   /* band 7: table cah(43,37) in (log10 co2 amount, log10 h2o amount) */
            const REAL u1 = (REAL)-3.0, du = (REAL)0.15, w1 = (REAL)-4.0, dw = (REAL)0.15;
            const int nu = 43, nw = 37;
            const REAL x0 = u1 + (REAL)nu * du, y0 = w1 + (REAL)nw * dw, x1 = u1 - (REAL)0.5 * du, y1 = w1 - (REAL)0.5 * dw;
            for (int k = 1; k <= np + 1; k++) {
                REAL ulog = LOG10(so2[k] * snt); if (ulog > x0) ulog = x0;
                REAL wlog = LOG10(swh[k] * snt); if (wlog > y0) wlog = y0;
                int ic = (int)((ulog - x1) / du + (REAL)1.), iw = (int)((wlog - y1) / dw + (REAL)1.);
                if (ic < 2) ic = 2; if (iw < 2) iw = 2; if (ic > nu) ic = nu; if (iw > nw) iw = nw;
                const REAL dc = ulog - (REAL)(ic - 2) * du - u1, dd = wlog - (REAL)(iw - 2) * dw - w1;
                const REAL x2 = F2(t->cah, 43, ic - 1, iw - 1) + (F2(t->cah, 43, ic - 1, iw) - F2(t->cah, 43, ic - 1, iw - 1)) / dw * dd;
                REAL y2 = x2 + (F2(t->cah, 43, ic, iw - 1) - F2(t->cah, 43, ic - 1, iw - 1)) / du * dc;
                if (y2 < 0) y2 = 0;
                df[k] = df[k] + (REAL)1.5 * y2;
            }
        }
        {   /* band 8: table coa(62,101) in (co2 * sec, log10 p) */
            const REAL u1 = (REAL)0.000250, du = (REAL)0.000050, w1 = (REAL)-2.0, dw = (REAL)0.05;
            const int nx = 62, ny = 101;
            const REAL x0 = u1 + (REAL)nx * du, y0 = w1 + (REAL)ny * dw, x1 = u1 - (REAL)0.5 * du, y1 = w1 - (REAL)0.5 * dw;
            for (int k = 1; k <= np + 1; k++) {
                REAL ulog = co2 * snt; if (ulog > x0) ulog = x0;
                REAL wlog = LOG10(A2(pl, k)); if (wlog > y0) wlog = y0;
                int ic = (int)((ulog - x1) / du + (REAL)1.), iw = (int)((wlog - y1) / dw + (REAL)1.);
                if (ic < 2) ic = 2; if (iw < 2) iw = 2; if (ic > nx) ic = nx; if (iw > ny) iw = ny;
                const REAL dc = ulog - (REAL)(ic - 2) * du - u1, dd = wlog - (REAL)(iw - 2) * dw - w1;
                const REAL x2 = F2(t->coa, 62, ic - 1, iw - 1) + (F2(t->coa, 62, ic - 1, iw) - F2(t->coa, 62, ic - 1, iw - 1)) / dw * dd;
                REAL y2 = x2 + (F2(t->coa, 62, ic, iw - 1) - F2(t->coa, 62, ic - 1, iw - 1)) / du * dc;
                if (y2 < 0) y2 = 0;
                df[k] = df[k] + (REAL)1.5 * y2;
            }
        }
        int foundtop = 0;
        for (int k = 1; k <= np; k++) if (fcld_c[k] > (REAL)0.02 && !foundtop) { foundtop = 1; ntop = k; }
        if (!foundtop) ntop = np + 1;
        const REAL dftop = df[ntop];
        for (int k = 1; k <= np + 1; k++)
            if (k > ntop) { const REAL xx4 = A2(flx, k) / A2(flx, ntop); df[k] = dftop + xx4 * (df[k] - dftop); }
        for (int k = 1; k <= np + 1; k++) {
            if (df[k] > A2(flx, k) - (REAL)1.0e-8) df[k] = A2(flx, k) - (REAL)1.0e-8;
            A2(flx, k) = A2(flx, k) - df[k];
            A2(flc, k) = A2(flc, k) - df[k];
        }
        REAL xx4 = A2(flx, np + 1) + df[np + 1];
        const REAL eps = sizeof(REAL) == 4 ? (REAL)1.1920929e-07 : (REAL)2.220446049250313e-16;
        if (FABS(xx4) > eps) {
            xx4 = (REAL)1.0 - df[np + 1] / xx4;
            if (xx4 > 1) xx4 = 1;
            if (xx4 < 0) xx4 = 0;
        } else xx4 = 0;
        fdirir[i] *= xx4; fdifir[i] *= xx4; fdiruv[i] *= xx4; fdifuv[i] *= xx4; fdirpar[i] *= xx4; fdifpar[i] *= xx4;
        for (int ib = 1; ib <= 8; ib++) {
            flx_sfc_band[(size_t)(ib - 1) * m + i] *= xx4;
            if (do_drfband) { drband[(size_t)(ib - 1) * m + i] *= xx4; dfband[(size_t)(ib - 1) * m + i] *= xx4; }
        }
    }
    free(W);
    return 0;
#undef A2
#undef A3B
}

#undef CS_GRAV
#undef CS_DSM
