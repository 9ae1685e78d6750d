/* geosrad.h -- C ABI of the MI355X-native radiation hot path (libgeosrad.so).
 *
 * Drop-in boundary: these entry points are what a Fortran ISO_C_BINDING shim binds in place of the
 * reference's solver entry points (paths relative to the GEOS-ESM/GEOSradiation_GridComp checkout):
 *
 *   geosrad_rrtmg_lw[_dev]        <- rrtmg_lw_rad::rrtmg_lw
 *                                    GEOSirrad_GridComp/RRTMG/rrtmg_lw/gcm_model/src/rrtmg_lw_rad.F90:15-23,110-201
 *   geosrad_set_tables_lw         <- rrtmg_lw_init::rrtmg_lw_ini           .../src/rrtmg_lw_init.F90:22
 *   geosrad_rrtmg_sw[_dev]        <- rrtmg_sw_rad::rrtmg_sw
 *                                    GEOSsolar_GridComp/RRTMG/rrtmg_sw/gcm_model/src/rrtmg_sw_rad.F90:68-124
 *   geosrad_set_tables_sw         <- rrtmg_sw_init::rrtmg_sw_ini           .../src/rrtmg_sw_init.F90:23
 *   geosrad_irrad[_dev]           <- irradmod::irrad                       GEOSirrad_GridComp/irrad.F90:27-35
 *   geosrad_sorad[_dev]           <- soradmod::sorad                       GEOSsolar_GridComp/sorad.F90:43-51
 *   geosrad_set_tables_chou_sw    <- sorad_constants / rad_constants (data modules)  GEOSsolar_GridComp/soradconstants.F90
 *   geosrad_set_tables_chou_lw    <- irrad_constants / rad_constants (data modules)  GEOSirrad_GridComp/irradconstants.F90
 *   geosrad_mcica[_dev]           <- cloud_subcol_gen::generate_stochastic_clouds
 *                                    GEOS_RadiationShared/cloud_subcol_gen.F90:132-137
 *   geosrad_clearcounts           <- cloud_subcol_gen::clearCounts_threeBand   .../cloud_subcol_gen.F90:611-614
 *   geosrad_set_corr_lengths      <- cloud_subcol_gen::initialize_cloud_subcol_gen .../cloud_subcol_gen.F90:109-111
 *   geosrad_set_inhomogeneity     <- cloud_condensate_inhomogeneity::{set,unset}_inhomogeneity
 *                                    GEOS_RadiationShared/cloud_condensate_inhomogeneity.F90:45,75
 *
 * Conventions
 *   - One context per (process, GPU, precision).  `real_kind` = 4 or 8 fixes the element type of every
 *     `const void*`/`void*` real array below (float or double), like the reference's default `real`.
 *   - Array layouts are exactly the reference's Fortran layouts, e.g. play(ncol,nlay): column index
 *     fastest; RRTMG ordering (layer 1 = lowest layer).  Nothing is transposed at the boundary.
 *   - Plain entry points take caller-owned HOST pointers (copied to/from HBM internally); `_dev` entry
 *     points take DEVICE pointers and a hipStream_t (as void*) and are asynchronous on that stream.
 *   - Every function returns 0 on success or a GEOSRAD_E* code; geosrad_last_error() gives the text.
 *     Where the reference would `error stop`, the code is GEOSRAD_EINPUT and the text is the reference's
 *     message (e.g. "negative values in input: play", rrtmg_lw_rad.F90:209-318).
 *   - Thread safety: a context may be used by one thread at a time; distinct contexts are independent
 *     (the reference's module-level state is replaced by the context).
 */
#ifndef GEOSRAD_H
#define GEOSRAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct geosrad_ctx geosrad_ctx;

enum {
    GEOSRAD_OK = 0,
    GEOSRAD_EINVAL = 1,   /* bad argument / call order                                   */
    GEOSRAD_ENODEV = 2,   /* no usable HIP device: the product never falls back to CPU */
    GEOSRAD_EHIP = 3,     /* HIP runtime error                                          */
    GEOSRAD_ETABLE = 4,   /* malformed / missing coefficient table                      */
    GEOSRAD_EINPUT = 5,   /* the reference would `error stop` on these inputs           */
    GEOSRAD_ENOMEM = 6
};

/* NB_LW = nbndlw, NG_LW = ngptlw (parrrtm.F90:33,39) */
#define GEOSRAD_NB_LW 16
#define GEOSRAD_NG_LW 140

/* ---- context ------------------------------------------------------------------------------------
 * The reference keeps its state in module variables of the process; here it lives in a context.  One context = one HIP
 * device (geosrad_create) or several (geosrad_create_multi, SURVEY 8b: the host-pointer solver entry points then cut [0, ncol) into
 * contiguous shards, one per device, processed concurrently; results bitwise those of one device).
 * device_id = GEOSRAD_DEVICE_AUTO: the device geosrad_pick_device chooses - GEOSRAD_DEVICE if set (taken as it is: an id outside
 * [0, ndev) is GEOSRAD_ENODEV, not another GPU), else the launcher's node-local MPI rank (OMPI_COMM_WORLD_LOCAL_RANK, SLURM_LOCALID,
 * MV2_COMM_WORLD_LOCAL_RANK, MPI_LOCALRANKID, PMI_LOCAL_RANK) modulo the number of visible devices - what the Fortran drop-in uses, so the ranks of a GEOS job (GEOS_SolarGridComp.F90:3701-3709 balances their
 * work) spread over a node's GPUs without configuration. */
#define GEOSRAD_DEVICE_AUTO (-1)
int geosrad_create(geosrad_ctx **ctx, int device_id, int real_kind /* 4 | 8 */);
int geosrad_create_multi(geosrad_ctx **ctx, const int *device_ids, int ndev, int real_kind);
int geosrad_pick_device(int ndev);      /* pure function of the environment; no device needed; -1 = GEOSRAD_DEVICE out of range */
int geosrad_destroy(geosrad_ctx *ctx);
const char *geosrad_last_error(const geosrad_ctx *ctx);
int geosrad_real_kind(const geosrad_ctx *ctx);
/* max columns processed per internal batch (bounds the HBM workspace); default 131072 */
int geosrad_set_chunk(geosrad_ctx *ctx, int max_columns);
/* bytes of HBM workspace currently held */
size_t geosrad_workspace_bytes(const geosrad_ctx *ctx);

/* ---- tables ("GRTB" blobs, geosradiation_gridcomp_amd/data/) ------------------------------------- */
/* what rrtmg_lw_ini leaves in rrlw_kgNN/rrlw_tbl/rrlw_wvn/rrlw_ref/rrlw_cld (rrtmg_lw_init.F90:22-165) */
int geosrad_set_tables_lw(geosrad_ctx *ctx, const void *blob, size_t nbytes);
int geosrad_load_tables_lw(geosrad_ctx *ctx, const char *path);
/* what rrtmg_sw_ini leaves in rrsw_kgNN/rrsw_ref/rrsw_cld/rrsw_wvn/NRLSSI2 (SW/rrtmg_sw_init.F90:23-190) */
int geosrad_set_tables_sw(geosrad_ctx *ctx, const void *blob, size_t nbytes);
int geosrad_load_tables_sw(geosrad_ctx *ctx, const char *path);
/* the coefficient tables of the Chou-Suarez LW scheme: irrad_constants + the IR part of rad_constants */
int geosrad_set_tables_chou_lw(geosrad_ctx *ctx, const void *blob, size_t nbytes);
int geosrad_load_tables_chou_lw(geosrad_ctx *ctx, const char *path);
/* the coefficient tables of the Chou-Suarez SW scheme: sorad_constants + the UV / NIR part of rad_constants */
int geosrad_set_tables_chou_sw(geosrad_ctx *ctx, const void *blob, size_t nbytes);
int geosrad_load_tables_chou_sw(geosrad_ctx *ctx, const char *path);
/* ih = 0 homogeneous (blob ignored), 1 beta, 2 gamma; blob = xcw(1000,140) table of that kind */
int geosrad_set_inhomogeneity(geosrad_ctx *ctx, int ih, const void *xcw_blob, size_t nbytes);
int geosrad_load_inhomogeneity(geosrad_ctx *ctx, int ih, const char *path);
/* adl = (am1, am2, am30, am4) cloud presence, rdl = same for condensate; NULL keeps the current value */
int geosrad_set_corr_lengths(geosrad_ctx *ctx, const double *adl, const double *rdl);
/* host copy of one named real array of a *.grtb coefficient file; needs neither a context nor a device.  The Fortran shim uses
 * it for the xcw table behind the reference's host function zcw_lookup (cloud_condensate_inhomogeneity.F90:86-124), which
 * GEOS_IrradGridComp.F90:1472 / GEOS_SolarGridComp.F90:3346 import.  count = number of reals expected (checked). */
int geosrad_read_table(const char *path, const char *name, int real_kind, void *dst, size_t count);

/* ---- RRTMG_LW --------------------------------------------------------------------------------------
 * Argument names, order, units and meaning are those of rrtmg_lw (rrtmg_lw_rad.F90:15-201).
 *   in : play,tlay,h2ovmr..ccl4vmr,cldf,ciwp,clwp,rei,rel,zm (ncol,nlay); plev,tlev (ncol,0:nlay);
 *        tsfc,alat (ncol); emis (ncol,16); tauaer (ncol,nlay,16); band_output[16] (0/1)
 *   out: uflx,dflx,uflxc,dflxc,duflx_dTs,duflxc_dTs (ncol,nlay+1); clearCounts int32 (ncol,4);
 *        olrb,dolrb_dTs (16,ncol) [touched only for bands with band_output != 0]
 * psize (cache-blocking partition size) is accepted for signature compatibility and ignored.
 * duflx*_dTs / dolrb_dTs may be NULL when dudTs == 0.  tauaer == NULL means "no aerosol" (all zero). */
int geosrad_rrtmg_lw(geosrad_ctx *ctx, int ncol, int nlay, int psize, int dudTs,
                     const void *play, const void *plev, const void *tlay, const void *tlev,
                     const void *tsfc, const void *emis,
                     const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr,
                     const void *n2ovmr, const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr,
                     const void *cfc22vmr, const void *ccl4vmr,
                     const void *cldf, const void *ciwp, const void *clwp, const void *rei, const void *rel,
                     int iceflglw, int liqflglw,
                     const void *tauaer, const void *zm, const void *alat, int dyofyr, int cloudLM, int cloudMH,
                     int32_t *clearCounts,
                     void *uflx, void *dflx, void *uflxc, void *dflxc, void *duflx_dTs, void *duflxc_dTs,
                     const int32_t *band_output, void *olrb, void *dolrb_dTs);

/* same, all array pointers in device memory; asynchronous on `stream` (hipStream_t).  Input validation
 * results (the reference's error stops) are reported by geosrad_check(). */
int geosrad_rrtmg_lw_dev(geosrad_ctx *ctx, void *stream, int ncol, int nlay, int psize, int dudTs,
                         const void *play, const void *plev, const void *tlay, const void *tlev,
                         const void *tsfc, const void *emis,
                         const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr,
                         const void *n2ovmr, const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr,
                         const void *cfc22vmr, const void *ccl4vmr,
                         const void *cldf, const void *ciwp, const void *clwp, const void *rei, const void *rel,
                         int iceflglw, int liqflglw,
                         const void *tauaer, const void *zm, const void *alat, int dyofyr, int cloudLM, int cloudMH,
                         int32_t *clearCounts,
                         void *uflx, void *dflx, void *uflxc, void *dflxc, void *duflx_dTs, void *duflxc_dTs,
                         const int32_t *band_output /* host */, void *olrb, void *dolrb_dTs);

/* Synchronise `stream` and translate the device-side input checks of the preceding *_dev calls into a
 * return code (GEOSRAD_OK / GEOSRAD_EINPUT + message).  The host-pointer entry points call this. */
int geosrad_check(geosrad_ctx *ctx, void *stream);

/* Per-kernel timing with HIP events recorded on the launch stream, around every kernel the *_dev entry points
 * enqueue (kernel ids 0..5 = k_validate_pwv, k_setcoef, k_overlap, k_mcica, k_lw_bands, k_lw_reduce; 6..9 =
 * k_sw_validate, k_sw_setcoef, k_sw_bands, k_sw_reduce; 10..11 = k_chou_prep, k_chou_bands; 12..13 = k_sorad_prep (+ cloud), k_sorad_pass; k_overlap / k_mcica are shared).
 * geosrad_profile(ctx, 1) resets and enables, geosrad_profile_read() waits for the recorded events and
 * returns the accumulated milliseconds and launch count of one kernel. */
int geosrad_profile(geosrad_ctx *ctx, int enable);
int geosrad_profile_read(geosrad_ctx *ctx, int kernel_id, double *total_ms, long *launches);
const char *geosrad_kernel_name(int kernel_id);      /* the slot's generic name (the default path's first mapping) */
/* the kernel that runs in that slot under the context's kernel paths, as it is named in a rocprofv3 kernel trace
 * (k_sw_reform | k_sw_bands, k_lw_bands | k_lw_cols, k_sorad_pass | k_sorad_col ...) */
const char *geosrad_kernel_label(geosrad_ctx *ctx, int kernel_id);

/* Test hook (host pointers, n doubles each): a / b, 1 / b and sqrt(b) as the fp64 RRTMG_SW instantiation evaluates them on the device
 * (hardware reciprocal / reciprocal square root + Newton steps + a residual correction: <= 1 ulp for operands in the normal range;
 * tests/test_gpu_fastmath.py). */
int geosrad_dbg_fast64(int n, const double *a, const double *b, double *quot, double *rcp, double *root);

/* Debug / test hooks: gas optical depth and Planck fraction as the reference's taumol leaves them,
 * Fortran (nlay,140,ncol), host pointers; ncol*nlay*140 reals each.  Uses the same kernels as
 * geosrad_rrtmg_lw with an extra store. */
int geosrad_rrtmg_lw_taumol(geosrad_ctx *ctx, int ncol, int nlay,
                            const void *play, const void *plev, const void *tlay, const void *tlev,
                            const void *tsfc, const void *emis,
                            const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr,
                            const void *n2ovmr, const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr,
                            const void *cfc22vmr, const void *ccl4vmr, const void *tauaer,
                            void *taug, void *pfracs);

/* ---- RRTMG_SW ------------------------------------------------------------------------------------------
 * rrtmg_sw (SW/rrtmg_sw_rad.F90:68-124), same argument order without the MAPL handle / RC (the return code is
 * the RC).  Real arrays, element type = real_kind, Fortran layouts:
 *   coszen, alat, asdir, asdif, aldir, aldif (ncol); play, tlay, gases, cld, ciwp, clwp, rei, rel, zm (ncol,nlay);
 *   plev (ncol,nlay+1); tauaer, ssaaer, asmaer (ncol,nlay,14) (read only when iaer == 10, else may be NULL);
 *   swuflx, swdflx, swuflxc, swdflxc (ncol,nlay+1); nirr..uvrf, cot* (ncol); fswband, drband, dfband (ncol,14)
 *   (drband/dfband written only when do_drfband != 0); clearCounts int32 (ncol,4).
 *   scon, adjes: double here, rounded to real_kind like the reference's default-real dummy arguments.
 *   bndscl (14 reals), indsolvar (2 reals) and solcycfrac (1 real): HOST pointers in both variants, NULL = absent optional argument.
 *   isolvar in {-1, 0, 1, 2, 3} (rrtmg_sw_rad.F90:893-1127).  1 = position solcycfrac in [0, 1] of the mean solar cycle AvgCyc11 with
 *   NRLSSI2's adjust_solcyc_amplitudes / interpolate_indices (NRLSSI2.F90:236-332): needs solcycfrac (GEOSRAD_EINPUT otherwise, as the
 *   reference's _FAIL).  The GridComp-level entry point geosrad_sw_driver_rrtmg_dev refuses it like SORADCORE does
 *   (GEOS_SolarGridComp.F90:6286-6292).
 *   rpart (the reference's column-partition size) is accepted and ignored: the GPU batches internally. */
int geosrad_rrtmg_sw(geosrad_ctx *ctx, int rpart, int ncol, int nlay, double scon, double adjes, const void *coszen, int isolvar,
                     const void *play, const void *plev, const void *tlay,
                     const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *o2vmr,
                     int iceflgsw, int liqflgsw,
                     const void *cld, const void *ciwp, const void *clwp, const void *rei, const void *rel,
                     int dyofyr, const void *zm, const void *alat,
                     int iaer, const void *tauaer, const void *ssaaer, const void *asmaer,
                     const void *asdir, const void *asdif, const void *aldir, const void *aldif,
                     int cloudLM, int cloudMH, int normFlx,
                     int32_t *clearCounts, void *swuflx, void *swdflx, void *swuflxc, void *swdflxc,
                     void *nirr, void *nirf, void *parr, void *parf, void *uvrr, void *uvrf, void *fswband,
                     void *cotdtp, void *cotdhp, void *cotdmp, void *cotdlp,
                     void *cotntp, void *cotnhp, void *cotnmp, void *cotnlp,
                     int do_drfband, void *drband, void *dfband, const void *bndscl, const void *indsolvar,
                     const void *solcycfrac);
/* same, DEVICE pointers, asynchronous on `stream`; input assertions are reported by geosrad_check() */
int geosrad_rrtmg_sw_dev(geosrad_ctx *ctx, void *stream, int rpart, int ncol, int nlay, double scon, double adjes,
                         const void *coszen, int isolvar,
                         const void *play, const void *plev, const void *tlay,
                         const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *o2vmr,
                         int iceflgsw, int liqflgsw,
                         const void *cld, const void *ciwp, const void *clwp, const void *rei, const void *rel,
                         int dyofyr, const void *zm, const void *alat,
                         int iaer, const void *tauaer, const void *ssaaer, const void *asmaer,
                         const void *asdir, const void *asdif, const void *aldir, const void *aldif,
                         int cloudLM, int cloudMH, int normFlx,
                         int32_t *clearCounts, void *swuflx, void *swdflx, void *swuflxc, void *swdflxc,
                         void *nirr, void *nirf, void *parr, void *parf, void *uvrr, void *uvrf, void *fswband,
                         void *cotdtp, void *cotdhp, void *cotdmp, void *cotdlp,
                         void *cotntp, void *cotnhp, void *cotnmp, void *cotnlp,
                         int do_drfband, void *drband, void *dfband, const void *bndscl, const void *indsolvar,
                         const void *solcycfrac);
/* Debug / test hook: taug, taur Fortran (nlay,112,ncol) and the solar source ssi (112,ncol) [sfluxzen when
 * isolvar < 0] as the reference's taumol_sw leaves them (SW/rrtmg_sw_taumol.F90:27); host pointers. */
int geosrad_rrtmg_sw_taumol(geosrad_ctx *ctx, int ncol, int nlay, double scon, int isolvar,
                            const void *play, const void *plev, const void *tlay,
                            const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *o2vmr,
                            const void *bndscl, const void *indsolvar, const void *solcycfrac,
                            void *taug, void *taur, void *ssi);
/* Debug / test hook: the McICA cloud optics of the solver's own sub-columns - delta-scaled optical depth, single-scattering
 * albedo and asymmetry parameter, Fortran (nlay,112,ncol), as the reference's cldprmc_sw leaves taucmc, ssacmc, asmcmc
 * (SW/rrtmg_sw_cldprmc.F90:36-418; 0, 1, 0 in clear cells); host pointers. */
int geosrad_rrtmg_sw_cldprmc(geosrad_ctx *ctx, int ncol, int nlay, const void *play, const void *plev, const void *tlay,
                             const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *o2vmr,
                             int iceflgsw, int liqflgsw, const void *cld, const void *ciwp, const void *clwp, const void *rei,
                             const void *rel, int dyofyr, const void *zm, const void *alat, int cloudLM, int cloudMH,
                             void *taucmc, void *ssacmc, void *asmcmc);

/* ---- Chou-Suarez longwave ------------------------------------------------------------------------------
 * irrad (GEOSirrad_GridComp/irrad.F90:27-35), same argument order (the logical `trace` as int).  Layers from the TOP down
 * (k = 1 top layer), ple in Pa; Fortran layouts: ple (m,np+1); ta, wa, oa, n2o, ch4, cfc11, cfc12, cfc22, fcld (m,np); tb (m);
 * cwc, reff (m,np,4); fs, tg, tv (m,ns); eg, ev, rv (m,ns,10); taua, ssaa, asya (m,np,nb) are IN-OUT and rescaled in place
 * exactly like the reference does (irrad.F90:655-678; may be NULL when na == 0); outputs flxu ... flxad, dfdts (m,np+1),
 * sfcem (m), taudiag (m,np,10).  Upward fluxes are negative.  Non-OVERCAST behaviour (maximum-random overlap).
 * trace == 0: band 10 is skipped for every column (the reference `return`s out of its column loop there, irrad.F90:478;
 * GEOS always passes trace = .true.). */
int geosrad_irrad(geosrad_ctx *ctx, int m, int np, const void *ple, const void *ta, const void *wa, const void *oa, const void *tb,
                  double co2, int trace, const void *n2o, const void *ch4, const void *cfc11, const void *cfc12, const void *cfc22,
                  const void *cwc, const void *fcld, int ict, int icb, const void *reff,
                  int ns, const void *fs, const void *tg, const void *eg, const void *tv, const void *ev, const void *rv,
                  int na, int nb, void *taua, void *ssaa, void *asya,
                  void *flxu, void *flcu, void *flau, void *flxau, void *flxd, void *flcd, void *flad, void *flxad,
                  void *dfdts, void *sfcem, void *taudiag);
int geosrad_irrad_dev(geosrad_ctx *ctx, void *stream, int m, int np, const void *ple, const void *ta, const void *wa, const void *oa,
                      const void *tb, double co2, int trace, const void *n2o, const void *ch4, const void *cfc11, const void *cfc12,
                      const void *cfc22, const void *cwc, const void *fcld, int ict, int icb, const void *reff,
                      int ns, const void *fs, const void *tg, const void *eg, const void *tv, const void *ev, const void *rv,
                      int na, int nb, void *taua, void *ssaa, void *asya,
                      void *flxu, void *flcu, void *flau, void *flxau, void *flxd, void *flcd, void *flad, void *flxad,
                      void *dfdts, void *sfcem, void *taudiag);

/* ---- Chou-Suarez shortwave -----------------------------------------------------------------------------
 * sorad (GEOSsolar_GridComp/sorad.F90:43-51), same argument order.  Layers from the TOP down, pl in hPa; Fortran layouts:
 * cosz, rsuvbm, rsuvdf, rsirbm, rsirdf (m); pl (m,np+1); ta, wa, oa, fcld (m,np); cwc, reff (m,np,4); taua, ssaa, asya
 * (m,np,nb = 8) in the (tau, tau*ssa, tau*ssa*g) form the reference expects; hk_uv (5), hk_ir (3,10): HOST pointers in both
 * variants; outputs flx, flc, flxu, flcu (m,np+1), fdir/fdif uv/par/ir (m), flx_sfc_band (m,8), drband, dfband (m,8; written
 * only when do_drfband != 0, else may be NULL).  Fluxes are fractions of the TOA insolation.  Non-OVERCAST behaviour. */
int geosrad_sorad(geosrad_ctx *ctx, int m, int np, int nb, const void *cosz, const void *pl, const void *ta, const void *wa,
                  const void *oa, double co2, const void *cwc, const void *fcld, int ict, int icb, const void *reff,
                  const void *hk_uv, const void *hk_ir, const void *taua, const void *ssaa, const void *asya,
                  const void *rsuvbm, const void *rsuvdf, const void *rsirbm, const void *rsirdf,
                  void *flx, void *flc, void *fdiruv, void *fdifuv, void *fdirpar, void *fdifpar, void *fdirir, void *fdifir,
                  void *flxu, void *flcu, void *flx_sfc_band, int do_drfband, void *drband, void *dfband);
int geosrad_sorad_dev(geosrad_ctx *ctx, void *stream, int m, int np, int nb, const void *cosz, const void *pl, const void *ta,
                      const void *wa, const void *oa, double co2, const void *cwc, const void *fcld, int ict, int icb,
                      const void *reff, const void *hk_uv, const void *hk_ir, const void *taua, const void *ssaa, const void *asya,
                      const void *rsuvbm, const void *rsuvdf, const void *rsirbm, const void *rsirdf,
                      void *flx, void *flc, void *fdiruv, void *fdifuv, void *fdirpar, void *fdifpar, void *fdirir, void *fdifir,
                      void *flxu, void *flcu, void *flx_sfc_band, int do_drfband, void *drband, void *dfband);

/* ---- McICA ------------------------------------------------------------------------------------------
 * generate_stochastic_clouds (cloud_subcol_gen.F90:132): profile inputs Fortran (nlay,dncol) there; here
 * the solver-API layout (ncol,nlay) is used for the inputs (what rrtmg_lw receives), outputs are
 * Fortran (nlay,nsubcol,ncol): cldy int32 (0/1), ciwp_stoch, clwp_stoch reals. */
int geosrad_mcica(geosrad_ctx *ctx, int ncol, int nsubcol, int nlay,
                  const void *zmid, const void *alat, int doy, const void *play, const void *cldfrac,
                  const void *ciwp, const void *clwp, double cwp_tiny, const int32_t seed_order[4],
                  int32_t *cldy_stoch, void *ciwp_stoch, void *clwp_stoch);
/* same, DEVICE pointers, asynchronous on `stream` (seed_order stays a host pointer) */
int geosrad_mcica_dev(geosrad_ctx *ctx, void *stream, int ncol, int nsubcol, int nlay,
                      const void *zmid, const void *alat, int doy, const void *play, const void *cldfrac,
                      const void *ciwp, const void *clwp, double cwp_tiny, const int32_t seed_order[4],
                      int32_t *cldy_stoch, void *ciwp_stoch, void *clwp_stoch);
/* clearCounts_threeBand (cloud_subcol_gen.F90:611): cldy (nlay,nsubcol,ncol) int32 -> clearCnts (4,ncol) */
int geosrad_clearcounts(geosrad_ctx *ctx, int ncol, int nsubcol, int nlay, int cloudLM, int cloudMH,
                        const int32_t *cldy_stoch, int32_t *clearCnts);

/* ---- RATS diagnostics (GEOS_IrradGridComp.F90:3389-3469) ------------------------------------------
 * LW_Driver calls rrtmg_lw once more per gas listed under RATS_DIAGNOSTICS: with that gas's mixing ratio set to zero and keeps
 * the total-sky UFLX, DFLX, DUFLX_DTS of each call.  geosrad_rrtmg_lw_rats_dev = geosrad_rrtmg_lw_dev (same arguments, same
 * results) + those profiles for `nrats` gases from the SAME call: uflx_rat, dflx_rat, duflx_dTs_rat are (ncol,nlay+1,nrats)
 * device arrays (gas slowest, like UFLXRAT(IM*JM,LM+1,nRATS), :3390-3392).  The input checks, the clear | cloudy partition, the
 * overlap correlations, the McICA sub-columns with their cloud optical depths and clearCounts do not depend on the gases and are
 * computed once; a gas costs setcoef + the band sweeps + the flux reduction.  Results are bitwise those of a separate call with the
 * gas's array zeroed.  NB the reference zeroes HCFC22 under the name 'HCFC22_R' but restores it under 'HCFC22' (:3434, :3464), so
 * that toggle never matches there; GEOSRAD_RAT_HCFC22 implements the evident intent. */
enum { GEOSRAD_RAT_H2O, GEOSRAD_RAT_O3, GEOSRAD_RAT_CO2, GEOSRAD_RAT_CH4, GEOSRAD_RAT_N2O, GEOSRAD_RAT_CFC11, GEOSRAD_RAT_CFC12,
       GEOSRAD_RAT_HCFC22, GEOSRAD_RAT_NGAS };
int geosrad_rrtmg_lw_rats_dev(geosrad_ctx *ctx, void *stream, int ncol, int nlay, int psize, int dudTs,
                              const void *play, const void *plev, const void *tlay, const void *tlev, const void *tsfc, const void *emis,
                              const void *h2ovmr, const void *o3vmr, const void *co2vmr, const void *ch4vmr, const void *n2ovmr,
                              const void *o2vmr, const void *cfc11vmr, const void *cfc12vmr, const void *cfc22vmr, const void *ccl4vmr,
                              const void *cldf, const void *ciwp, const void *clwp, const void *rei, const void *rel,
                              int iceflglw, int liqflglw, const void *tauaer, const void *zm, const void *alat, int dyofyr,
                              int cloudLM, int cloudMH, int32_t *clearCounts,
                              void *uflx, void *dflx, void *uflxc, void *dflxc, void *duflx_dTs, void *duflxc_dTs,
                              const int32_t *band_output, void *olrb, void *dolrb_dTs,
                              int nrats, const int32_t *rat_gas /*host, GEOSRAD_RAT_**/, void *uflx_rat, void *dflx_rat,
                              void *duflx_dTs_rat);

/* ---- GridComp data path either side of the solvers -------------------------------------
 * All arrays are DEVICE pointers of the context's real kind in the GEOS layout (IM*JM columns fastest, then the level /
 * layer index in MODEL ordering, 1 = top), asynchronous on `stream`.  A NULL output = Fortran "not associated" (export
 * not requested).  MAPL's physical constants are not defined in the reference repository: the caller passes them.
 *
 * geosrad_lw_driver_rrtmg_dev: the RRTMG branch of LW_Driver (GEOS_IrradGridComp.F90:3188-3372 prep / flip, :3470-3477 the
 * rrtmg_lw call with Ts_derivs = .true., :3487-3533 + :3560-3565 + :3601-3615 un-flip, sign conventions, SFCEM, net fluxes).
 * lcldlm / lcldmh are the MODEL-ordering super-layer interface indices (the routine flips them like IRR:3237-3239). */
enum { GEOSRAD_LWD_PLE /*(ncol,0:LM) Pa*/, GEOSRAD_LWD_PL /*(ncol,LM) Pa*/, GEOSRAD_LWD_T, GEOSRAD_LWD_Q, GEOSRAD_LWD_O3,
       GEOSRAD_LWD_CH4, GEOSRAD_LWD_N2O, GEOSRAD_LWD_CO2_3D /*nullable: CO2_FIXED is used*/, GEOSRAD_LWD_CFC11, GEOSRAD_LWD_CFC12,
       GEOSRAD_LWD_HCFC22, GEOSRAD_LWD_FCLD, GEOSRAD_LWD_CWC_LIQ, GEOSRAD_LWD_CWC_ICE, GEOSRAD_LWD_REFF_LIQ, GEOSRAD_LWD_REFF_ICE,
       GEOSRAD_LWD_TAUA /*(ncol,LM,nb) nullable*/, GEOSRAD_LWD_SSAA, GEOSRAD_LWD_TS /*(ncol)*/, GEOSRAD_LWD_EMIS, GEOSRAD_LWD_LATS,
       GEOSRAD_LWD_T2M, GEOSRAD_LWD_NIN };
enum { GEOSRAD_LWD_C_CO2_FIXED, GEOSRAD_LWD_C_O2, GEOSRAD_LWD_C_CCL4, GEOSRAD_C_AIRMW, GEOSRAD_C_H2OMW, GEOSRAD_C_O3MW,
       GEOSRAD_C_RGAS, GEOSRAD_C_GRAV, GEOSRAD_LWD_NCONST };
enum { GEOSRAD_LWD_FLXU_INT /*(ncol,0:LM)*/, GEOSRAD_LWD_FLXD_INT, GEOSRAD_LWD_FLCU_INT, GEOSRAD_LWD_FLCD_INT, GEOSRAD_LWD_DFDTS,
       GEOSRAD_LWD_DFDTSC, GEOSRAD_LWD_DFDTSNA, GEOSRAD_LWD_DFDTSCNA, GEOSRAD_LWD_FLX_INT, GEOSRAD_LWD_FLC_INT,
       GEOSRAD_LWD_SFCEM_INT /*(ncol), positive*/, GEOSRAD_LWD_TS_INT, GEOSRAD_LWD_CLDTTLW, GEOSRAD_LWD_CLDHILW, GEOSRAD_LWD_CLDMDLW,
       GEOSRAD_LWD_CLDLOLW, GEOSRAD_LWD_OLRB /*(16,ncol)*/, GEOSRAD_LWD_DOLRB, GEOSRAD_LWD_NOUT };
int geosrad_lw_driver_rrtmg_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int nb_aer, const void *const *in,
                                const double *consts, int iceflglw, int liqflglw, int doy, int lcldlm, int lcldmh,
                                const int32_t *band_output, void *const *out);
/* the same with the RATS loop (GEOS_IrradGridComp.F90:3389-3469) and its share of the un-flip (:3522-3530, :3614): rat_out =
 * INTERNAL FLXU_RAT, FLXD_RAT, FLX_RAT, DFDTS_RAT (ncol,0:LM,nrats) and SFCEM_RAT (ncol,nrats); any may be NULL. */
enum { GEOSRAD_LWD_FLXU_RAT, GEOSRAD_LWD_FLXD_RAT, GEOSRAD_LWD_FLX_RAT, GEOSRAD_LWD_DFDTS_RAT, GEOSRAD_LWD_SFCEM_RAT, GEOSRAD_LWD_NRATOUT };
int geosrad_lw_driver_rrtmg_rats_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int nb_aer, const void *const *in,
                                     const double *consts, int iceflglw, int liqflglw, int doy, int lcldlm, int lcldmh,
                                     const int32_t *band_output, void *const *out, int nrats, const int32_t *rat_gas,
                                     void *const *rat_out);

/* geosrad_lw_chou_post_dev: what the Chou-Suarez branch of LW_Driver adds after `call IRRAD` (which takes the GEOS fields as they are):
 * DFDTSC = 0, DFDTSNA = DFDTS, DFDTSCNA = 0 (GEOS_IrradGridComp.F90:2101-2108), the four net fluxes FL*_INT = FL*D_INT + FL*U_INT,
 * SFCEM_INT = -SFCEM_INT (in place), TS_INT = TS (:3601-3616).  Outputs may be NULL. */
enum { GEOSRAD_LWC_FLXU_INT, GEOSRAD_LWC_FLCU_INT, GEOSRAD_LWC_FLAU_INT, GEOSRAD_LWC_FLXAU_INT, GEOSRAD_LWC_FLXD_INT, GEOSRAD_LWC_FLCD_INT,
       GEOSRAD_LWC_FLAD_INT, GEOSRAD_LWC_FLXAD_INT, GEOSRAD_LWC_DFDTS, GEOSRAD_LWC_TS, GEOSRAD_LWC_NIN };
enum { GEOSRAD_LWC_SFCEM_INT /*in-out*/, GEOSRAD_LWC_FLX_INT, GEOSRAD_LWC_FLXA_INT, GEOSRAD_LWC_FLC_INT, GEOSRAD_LWC_FLA_INT, GEOSRAD_LWC_DFDTSC,
       GEOSRAD_LWC_DFDTSNA, GEOSRAD_LWC_DFDTSCNA, GEOSRAD_LWC_TS_INT, GEOSRAD_LWC_NOUT };
int geosrad_lw_chou_post_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, const void *const *in, void *const *out);

/* geosrad_sw_driver_rrtmg_dev: the RRTMG branch of SORADCORE on the packed daytime columns (GEOS_SolarGridComp.F90:6113-6219
 * prep / flip incl. the in-place aerosol normalisation, :6330-6388 the rrtmg_sw call, :6395-6450 un-flip, cloud fractions,
 * in-cloud optical thickness, net fluxes).  ple is (ncol,LM+1).  solvar = {bndscl[14], indsolvar[2]} host pointers or NULL. */
enum { GEOSRAD_SWD_PLE, GEOSRAD_SWD_PL, GEOSRAD_SWD_T, GEOSRAD_SWD_Q, GEOSRAD_SWD_O3, GEOSRAD_SWD_CH4, GEOSRAD_SWD_CL,
       GEOSRAD_SWD_TS, GEOSRAD_SWD_QQ_ICE, GEOSRAD_SWD_QQ_LIQ, GEOSRAD_SWD_RR_ICE, GEOSRAD_SWD_RR_LIQ,
       GEOSRAD_SWD_TAUA /*(ncol,LM,nb) in/out, nullable*/, GEOSRAD_SWD_SSAA, GEOSRAD_SWD_ASYA, GEOSRAD_SWD_ZT /*cos zenith*/,
       GEOSRAD_SWD_ALAT, GEOSRAD_SWD_ALBVR, GEOSRAD_SWD_ALBVF, GEOSRAD_SWD_ALBNR, GEOSRAD_SWD_ALBNF, GEOSRAD_SWD_NIN };
enum { GEOSRAD_SWD_C_CO2, GEOSRAD_SWD_C_O2, GEOSRAD_SWD_C_AIRMW, GEOSRAD_SWD_C_H2OMW, GEOSRAD_SWD_C_O3MW, GEOSRAD_SWD_C_RGAS,
       GEOSRAD_SWD_C_GRAV, GEOSRAD_SWD_C_UNDEF, GEOSRAD_SWD_NCONST };
enum { GEOSRAD_SWD_FSW /*(ncol,LM+1) model ordering*/, GEOSRAD_SWD_FSC, GEOSRAD_SWD_FSWU, GEOSRAD_SWD_FSCU, GEOSRAD_SWD_NIRR,
       GEOSRAD_SWD_NIRF, GEOSRAD_SWD_PARR, GEOSRAD_SWD_PARF, GEOSRAD_SWD_UVRR, GEOSRAD_SWD_UVRF, GEOSRAD_SWD_FSWBAND /*(ncol,14)*/,
       GEOSRAD_SWD_CLDTS, GEOSRAD_SWD_CLDHS, GEOSRAD_SWD_CLDMS, GEOSRAD_SWD_CLDLS, GEOSRAD_SWD_COTTP, GEOSRAD_SWD_COTHP,
       GEOSRAD_SWD_COTMP, GEOSRAD_SWD_COTLP,
       /* the no-aerosol flavour (FSWNAN ... of the GridComp, which calls the whole of SORADCORE a second time for them,
        * GEOS_SolarGridComp.F90:3249-3259): requested by a non-NULL pointer; validation, setcoef and McICA are shared with the
        * with-aerosol pass, only the band sweeps are repeated */
       GEOSRAD_SWD_FSWNA, GEOSRAD_SWD_FSCNA, GEOSRAD_SWD_FSWUNA, GEOSRAD_SWD_FSCUNA, GEOSRAD_SWD_FSWBANDNA /*(ncol,14)*/,
       GEOSRAD_SWD_NOUT };
int geosrad_sw_driver_rrtmg_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int nb_aer, const void *const *in,
                                const double *consts, int iceflgsw, int liqflgsw, double sc, double dist, int isolvar, int dyofyr,
                                int include_aerosols, int lcldlm, int lcldmh, int normflx, const void *bndsolvar,
                                const void *indsolvar, void *const *out);

/* geosrad_sw_driver_chou_dev: the Chou-Suarez branch of SORADCORE on the packed daytime columns (GEOS_SolarGridComp.F90:4484-4553
 * prep: interface pressures in hPa, odd oxygen -> non-negative ozone mass fraction, the species arrays QQ3 / RR3 with MAPL_UNDEF radii
 * replaced, zero aerosol arrays when there are none; :4558-4572 `call shrtwave` = SHRTWAVE :6597-6672 = `call SORAD`).  Fields in the
 * GEOS layout and MODEL ordering (what sorad expects).  The RADSW_BINARY_CLOUDS option (:4480-4481) is the caller's `where`.
 * taua / ssaa / asya (ncol,LM,8) in sorad's (tau, tau*ssa, tau*ssa*g) form, or all three NULL (num_aero_vars == 0).  hk_uv (5),
 * hk_ir (3,10): HOST pointers, the GridComp's HK_UV_TEMP / HK_IR_TEMP (:2997-3028).  lcldmh / lcldlm: sorad's ict / icb.
 * Outputs as SHRTWAVE's: FSW, FSC, FSWU, FSCU (ncol,LM+1), the six surface components (ncol), FSWBAND (ncol,8); DRBAND / DFBAND
 * (ncol,8) are written when do_drfband != 0 (SOLAR_TO_OBIO .and. include_aerosols), else may be NULL.  Fractions of the TOA insolation. */
enum { GEOSRAD_SWC_PLE /*(ncol,LM+1) Pa*/, GEOSRAD_SWC_T, GEOSRAD_SWC_Q, GEOSRAD_SWC_OX, GEOSRAD_SWC_CL, GEOSRAD_SWC_QI, GEOSRAD_SWC_QL,
       GEOSRAD_SWC_QR, GEOSRAD_SWC_QS, GEOSRAD_SWC_RI /*m, MAPL_UNDEF allowed*/, GEOSRAD_SWC_RL, GEOSRAD_SWC_RR, GEOSRAD_SWC_RS,
       GEOSRAD_SWC_TAUA /*(ncol,LM,8) nullable*/, GEOSRAD_SWC_SSAA, GEOSRAD_SWC_ASYA, GEOSRAD_SWC_ZT /*cos zenith*/, GEOSRAD_SWC_ALBVR,
       GEOSRAD_SWC_ALBVF, GEOSRAD_SWC_ALBNR, GEOSRAD_SWC_ALBNF, GEOSRAD_SWC_NIN };
enum { GEOSRAD_SWC_C_CO2, GEOSRAD_SWC_C_O3MW, GEOSRAD_SWC_C_AIRMW, GEOSRAD_SWC_C_UNDEF, GEOSRAD_SWC_NCONST };
enum { GEOSRAD_SWC_FSW, GEOSRAD_SWC_FSC, GEOSRAD_SWC_FSWU, GEOSRAD_SWC_FSCU, GEOSRAD_SWC_NIRR, GEOSRAD_SWC_NIRF, GEOSRAD_SWC_PARR,
       GEOSRAD_SWC_PARF, GEOSRAD_SWC_UVRR, GEOSRAD_SWC_UVRF, GEOSRAD_SWC_FSWBAND, GEOSRAD_SWC_DRBAND, GEOSRAD_SWC_DFBAND, GEOSRAD_SWC_NOUT };
int geosrad_sw_driver_chou_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, const void *const *in, const double *consts,
                               int lcldmh, int lcldlm, const void *hk_uv, const void *hk_ir, int do_drfband, void *const *out);

/* geosrad_lw_update_flx_dev: Update_Flx (GEOS_IrradGridComp.F90:3796-3999), the per-model-step linearisation of the LW fluxes in
 * the surface temperature.  rrtmg != 0: the no-aerosol flavours are `undef` and their internals may be NULL (IRR:3927-3990).
 * lev_mid_high / lev_low_mid: the model levels found from PREF (IRR:3811-3829). */
enum { GEOSRAD_LWU_TSINST, GEOSRAD_LWU_TS_INT, GEOSRAD_LWU_SFCEM_INT, GEOSRAD_LWU_FCLD, GEOSRAD_LWU_FLX_INT, GEOSRAD_LWU_FLXA_INT,
       GEOSRAD_LWU_FLC_INT, GEOSRAD_LWU_FLA_INT, GEOSRAD_LWU_FLXU_INT, GEOSRAD_LWU_FLXAU_INT, GEOSRAD_LWU_FLCU_INT,
       GEOSRAD_LWU_FLAU_INT, GEOSRAD_LWU_FLXD_INT, GEOSRAD_LWU_FLXAD_INT, GEOSRAD_LWU_FLCD_INT, GEOSRAD_LWU_FLAD_INT,
       GEOSRAD_LWU_DFDTS, GEOSRAD_LWU_DFDTSNA, GEOSRAD_LWU_DFDTSC, GEOSRAD_LWU_DFDTSCNA, GEOSRAD_LWU_NIN };
enum { GEOSRAD_LWU_FLX, GEOSRAD_LWU_FLXA, GEOSRAD_LWU_FLC, GEOSRAD_LWU_FLA, GEOSRAD_LWU_FLXU, GEOSRAD_LWU_FLXAU, GEOSRAD_LWU_FLCU,
       GEOSRAD_LWU_FLAU, GEOSRAD_LWU_FLXD, GEOSRAD_LWU_FLXAD, GEOSRAD_LWU_FLCD, GEOSRAD_LWU_FLAD, GEOSRAD_LWU_OLR, GEOSRAD_LWU_OLRA,
       GEOSRAD_LWU_OLC, GEOSRAD_LWU_OLA, GEOSRAD_LWU_OLCC5, GEOSRAD_LWU_DSFDTS, GEOSRAD_LWU_SFCEM, GEOSRAD_LWU_LWS, GEOSRAD_LWU_LWSA,
       GEOSRAD_LWU_LCS, GEOSRAD_LWU_LAS, GEOSRAD_LWU_LCSC5, GEOSRAD_LWU_FLNS, GEOSRAD_LWU_FLNSNA, GEOSRAD_LWU_FLNSC, GEOSRAD_LWU_FLNSA,
       GEOSRAD_LWU_DSFDTS0, GEOSRAD_LWU_SFCEM0, GEOSRAD_LWU_TSREFF, GEOSRAD_LWU_CLDTT, GEOSRAD_LWU_NOUT };
int geosrad_lw_update_flx_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int rrtmg, int lev_mid_high, int lev_low_mid,
                              double undef, const void *const *in, void *const *out);

/* geosrad_lw_update_rats_dev: the RATS exports of Update_Flx (GEOS_IrradGridComp.F90:4036-4120): for every listed gas n, from the
 * INTERNAL state with and without the gas: dOLR_n, dLWS_n, dFLNS_n, dSFCEM_n, NETTRAP_n (ncol,nrats); COLTRAP_n (ncol,LM,nrats);
 * FLX_n, DFDTS_n (ncol,0:LM,nrats).  Gas slowest, like the (IM,JM,0:LM,nRATS) internals; any output may be NULL. */
enum { GEOSRAD_LWR_FLX_INT /*(ncol,0:LM)*/, GEOSRAD_LWR_SFCEM_INT /*(ncol)*/, GEOSRAD_LWR_DFDTS, GEOSRAD_LWR_FLX_RAT /*(ncol,0:LM,nrats)*/,
       GEOSRAD_LWR_SFCEM_RAT /*(ncol,nrats)*/, GEOSRAD_LWR_DFDTS_RAT, GEOSRAD_LWR_NIN };
enum { GEOSRAD_LWR_DOLR, GEOSRAD_LWR_DLWS, GEOSRAD_LWR_DFLNS, GEOSRAD_LWR_DSFCEM, GEOSRAD_LWR_NETTRAP, GEOSRAD_LWR_COLTRAP, GEOSRAD_LWR_FLX,
       GEOSRAD_LWR_DFDTS_OUT, GEOSRAD_LWR_NOUT };
int geosrad_lw_update_rats_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int nrats, const void *const *in, void *const *out);

/* geosrad_lw_update_bands_dev: the band OLR / brightness-temperature exports of Update_Flx (GEOS_IrradGridComp.F90:3993-4021 with
 * Tbr_from_band_flux / invert_Planck_for_T, :4132-4208) for the bands selected by band_output[16]: OLRBbbRG = OLRB_int + DOLRB_int *
 * (TSINST - TS_INT); TBRBbbRG = narrow-band inversion of the Planck function in double precision, MAPL_UNDEF for a band whose updated
 * flux is zero everywhere (before the first full calculation).  olrb_int / dolrb_int: (16,ncol) as the driver returns them
 * (GEOSRAD_LWD_OLRB / _DOLRB); olrb_exp / tbrb_exp: (ncol,16), band bb's export is the contiguous slice bb, either may be NULL;
 * wavenum1 / wavenum2: rrlw_wvn's band limits in cm-1 (host). */
int geosrad_lw_update_bands_dev(geosrad_ctx *ctx, void *stream, int ncol, const int32_t *band_output, const double *wavenum1,
                                const double *wavenum2, double undef, const void *tsinst, const void *ts_int, const void *olrb_int,
                                const void *dolrb_int, void *olrb_exp, void *tbrb_exp);

/* geosrad_sw_update_export_dev: the flux part of UPDATE_EXPORT (GEOS_SolarGridComp.F90:7540-7579): exports = normalised
 * internals x SLR (3-D net / up / down, per-band, TOA and surface). */
enum { GEOSRAD_SWU_SLR, GEOSRAD_SWU_FSWN, GEOSRAD_SWU_FSCN, GEOSRAD_SWU_FSWNAN, GEOSRAD_SWU_FSCNAN, GEOSRAD_SWU_FSWUN,
       GEOSRAD_SWU_FSCUN, GEOSRAD_SWU_FSWUNAN, GEOSRAD_SWU_FSCUNAN, GEOSRAD_SWU_FSWBANDN, GEOSRAD_SWU_FSWBANDNAN, GEOSRAD_SWU_NIN };
enum { GEOSRAD_SWU_FSW, GEOSRAD_SWU_FSC, GEOSRAD_SWU_FSWNA, GEOSRAD_SWU_FSCNA, GEOSRAD_SWU_FSWU, GEOSRAD_SWU_FSCU, GEOSRAD_SWU_FSWUNA,
       GEOSRAD_SWU_FSCUNA, GEOSRAD_SWU_FSWD, GEOSRAD_SWU_FSCD, GEOSRAD_SWU_FSWDNA, GEOSRAD_SWU_FSCDNA, GEOSRAD_SWU_FSWBAND,
       GEOSRAD_SWU_FSWBANDNA, GEOSRAD_SWU_RSR, GEOSRAD_SWU_RSC, GEOSRAD_SWU_RSRNA, GEOSRAD_SWU_RSCNA, GEOSRAD_SWU_RSRS, GEOSRAD_SWU_RSCS,
       GEOSRAD_SWU_RSRSNA, GEOSRAD_SWU_RSCSNA, GEOSRAD_SWU_OSR, GEOSRAD_SWU_OSRCLR, GEOSRAD_SWU_OSRNA, GEOSRAD_SWU_OSRCNA,
       GEOSRAD_SWU_NOUT };
int geosrad_sw_update_export_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, int nbands, const void *const *in,
                                 void *const *out);

/* geosrad_sw_update_surface_dev: the 2-D block of UPDATE_EXPORT above the flux part (GEOS_SolarGridComp.F90:7403-7533): the four
 * albedo exports (import where SLR > 0, else MAPL_UNDEF), the total surface albedo ALB / ALBEDO (:7458-7465), the incident and
 * surface fluxes SLRTP, DR/DF UVR PAR NIR, the normal-incidence DRN*, SLRSF*, SLRSUF* (with SLN = SLR / ZTH where ZTH > 0, :6877-6881).
 * in: (ncol) fields, FSWN / FSCN / FSWNAN / FSCNAN (ncol,0:LM) (the last three only for the exports that read them). */
enum { GEOSRAD_SWS_SLR, GEOSRAD_SWS_ZTH, GEOSRAD_SWS_ALBVF, GEOSRAD_SWS_ALBVR, GEOSRAD_SWS_ALBNF, GEOSRAD_SWS_ALBNR, GEOSRAD_SWS_DRUVRN,
       GEOSRAD_SWS_DFUVRN, GEOSRAD_SWS_DRPARN, GEOSRAD_SWS_DFPARN, GEOSRAD_SWS_DRNIRN, GEOSRAD_SWS_DFNIRN, GEOSRAD_SWS_FSWN, GEOSRAD_SWS_FSCN,
       GEOSRAD_SWS_FSWNAN, GEOSRAD_SWS_FSCNAN, GEOSRAD_SWS_NIN };
enum { GEOSRAD_SWS_ALBVF_X, GEOSRAD_SWS_ALBVR_X, GEOSRAD_SWS_ALBNF_X, GEOSRAD_SWS_ALBNR_X, GEOSRAD_SWS_ALBEDO, GEOSRAD_SWS_SLRTP,
       GEOSRAD_SWS_DRUVR, GEOSRAD_SWS_DFUVR, GEOSRAD_SWS_DRPAR, GEOSRAD_SWS_DFPAR, GEOSRAD_SWS_DRNIR, GEOSRAD_SWS_DFNIR, GEOSRAD_SWS_DRNUVR,
       GEOSRAD_SWS_DRNPAR, GEOSRAD_SWS_DRNNIR, GEOSRAD_SWS_SLRSF, GEOSRAD_SWS_SLRSFC, GEOSRAD_SWS_SLRSFNA, GEOSRAD_SWS_SLRSFCNA,
       GEOSRAD_SWS_SLRSUF, GEOSRAD_SWS_SLRSUFC, GEOSRAD_SWS_SLRSUFNA, GEOSRAD_SWS_SLRSUFCNA, GEOSRAD_SWS_NOUT };
int geosrad_sw_update_surface_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, double undef, const void *const *in, void *const *out);

/* geosrad_rad_tendencies_dev: the parent's heating rates (GEOS_RadiationGridComp.F90:798-819). */
enum { GEOSRAD_RT_PLE, GEOSRAD_RT_FLW, GEOSRAD_RT_FSW, GEOSRAD_RT_FLWCLR, GEOSRAD_RT_FSWCLR, GEOSRAD_RT_FSWNA, GEOSRAD_RT_FLA,
       GEOSRAD_RT_FSCNA, GEOSRAD_RT_DSFDTS, GEOSRAD_RT_SFCEM, GEOSRAD_RT_TRD, GEOSRAD_RT_NIN };
enum { GEOSRAD_RT_DTDT, GEOSRAD_RT_RADLW, GEOSRAD_RT_RADSW, GEOSRAD_RT_RADLWC, GEOSRAD_RT_RADSWC, GEOSRAD_RT_RADSWNA,
       GEOSRAD_RT_RADLWCNA, GEOSRAD_RT_RADSWCNA, GEOSRAD_RT_BLW, GEOSRAD_RT_ALW, GEOSRAD_RT_RADSRF, GEOSRAD_RT_NOUT };
int geosrad_rad_tendencies_dev(geosrad_ctx *ctx, void *stream, int ncol, int lm, double grav, double cp, const void *const *in,
                               void *const *out);

/* ---- lit-column compaction of the solar component ----------------------------------------------------------------------
 * GEOS_SolarGridComp.F90:3686 (`daytime = ZTH > 0.`, NumLit = count(daytime)) and PackIt / UnPackIt (:7753-7799): SORADCORE works on
 * the daytime columns only, packed to the front of every field in (i, j) order; on a node each GPU therefore takes its share of the
 * LIT columns.  All pointers are device pointers of the context's real kind (int32 for the index arrays).
 *   geosrad_lit_index_dev : lit_index[m] = column of packed position m (stable order), lit_pos[column] = packed position or -1,
 *                           *nlit_dev = NumLit; when nlit_host is not NULL the stream is synchronised and NumLit returned in it
 *   geosrad_lit_pack_dev  : Packed(m, l) = UnPacked(lit_index[m], l), l = 1..nlev; Packed is (pdim, nlev), UnPacked (udim, nlev)
 *   geosrad_lit_unpack_dev: UnPacked(lit_index[m], l) = Packed(m, l); dark columns receive `dflt` when use_default != 0
 *                           (UnPackIt's optional DEFAULT), otherwise they keep their values */
int geosrad_lit_index_dev(geosrad_ctx *ctx, void *stream, int ncol, const void *zth, int32_t *lit_index, int32_t *lit_pos,
                          int32_t *nlit_dev, int *nlit_host);
int geosrad_lit_pack_dev(geosrad_ctx *ctx, void *stream, int pdim, int udim, int nlev, const int32_t *lit_index, const int32_t *nlit_dev,
                         const void *unpacked, void *packed);
int geosrad_lit_unpack_dev(geosrad_ctx *ctx, void *stream, int pdim, int udim, int nlev, const int32_t *lit_pos, const void *packed,
                           void *unpacked, int use_default, double dflt);

#ifdef __cplusplus
}
#endif
#endif /* GEOSRAD_H */
