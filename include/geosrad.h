/* geosrad.h -- C ABI of the MI355X-native radiation hot path (libgeosrad.so).
 *
 * Drop-in boundary: these entry points are what a Fortran ISO_C_BINDING shim binds in place of the
 * reference's solver entry points (paths relative to the GEOS-ESM/GEOSradiation_GridComp checkout):
 *
 *   geosrad_rrtmg_lw[_dev]        <- rrtmg_lw_rad::rrtmg_lw
 *                                    GEOSirrad_GridComp/RRTMG/rrtmg_lw/gcm_model/src/rrtmg_lw_rad.F90:15-23,110-201
 *   geosrad_set_tables_lw         <- rrtmg_lw_init::rrtmg_lw_ini           .../src/rrtmg_lw_init.F90:22
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

/* ---- context ------------------------------------------------------------------------------------ */
int geosrad_create(geosrad_ctx **ctx, int device_id, int real_kind /* 4 | 8 */);
int geosrad_destroy(geosrad_ctx *ctx);
const char *geosrad_last_error(const geosrad_ctx *ctx);
int geosrad_real_kind(const geosrad_ctx *ctx);
/* max columns processed per internal batch (bounds the HBM workspace); default 65536 */
int geosrad_set_chunk(geosrad_ctx *ctx, int max_columns);
/* bytes of HBM workspace currently held */
size_t geosrad_workspace_bytes(const geosrad_ctx *ctx);

/* ---- tables ("GRTB" blobs, geosradiation_gridcomp_amd/data/) ------------------------------------- */
/* what rrtmg_lw_ini leaves in rrlw_kgNN/rrlw_tbl/rrlw_wvn/rrlw_ref/rrlw_cld (rrtmg_lw_init.F90:22-165) */
int geosrad_set_tables_lw(geosrad_ctx *ctx, const void *blob, size_t nbytes);
int geosrad_load_tables_lw(geosrad_ctx *ctx, const char *path);
/* ih = 0 homogeneous (blob ignored), 1 beta, 2 gamma; blob = xcw(1000,140) table of that kind */
int geosrad_set_inhomogeneity(geosrad_ctx *ctx, int ih, const void *xcw_blob, size_t nbytes);
int geosrad_load_inhomogeneity(geosrad_ctx *ctx, int ih, const char *path);
/* adl = (am1, am2, am30, am4) cloud presence, rdl = same for condensate; NULL keeps the current value */
int geosrad_set_corr_lengths(geosrad_ctx *ctx, const double *adl, const double *rdl);

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
 * enqueue (kernel ids 0..5 = k_validate_pwv, k_setcoef, k_overlap, k_mcica, k_lw_bands, k_lw_reduce).
 * geosrad_profile(ctx, 1) resets and enables, geosrad_profile_read() waits for the recorded events and
 * returns the accumulated milliseconds and launch count of one kernel. */
int geosrad_profile(geosrad_ctx *ctx, int enable);
int geosrad_profile_read(geosrad_ctx *ctx, int kernel_id, double *total_ms, long *launches);
const char *geosrad_kernel_name(int kernel_id);

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

/* ---- McICA ------------------------------------------------------------------------------------------
 * generate_stochastic_clouds (cloud_subcol_gen.F90:132): profile inputs Fortran (nlay,dncol) there; here
 * the solver-API layout (ncol,nlay) is used for the inputs (what rrtmg_lw receives), outputs are
 * Fortran (nlay,nsubcol,ncol): cldy int32 (0/1), ciwp_stoch, clwp_stoch reals. */
int geosrad_mcica(geosrad_ctx *ctx, int ncol, int nsubcol, int nlay,
                  const void *zmid, const void *alat, int doy, const void *play, const void *cldfrac,
                  const void *ciwp, const void *clwp, double cwp_tiny, const int32_t seed_order[4],
                  int32_t *cldy_stoch, void *ciwp_stoch, void *clwp_stoch);
/* clearCounts_threeBand (cloud_subcol_gen.F90:611): cldy (nlay,nsubcol,ncol) int32 -> clearCnts (4,ncol) */
int geosrad_clearcounts(geosrad_ctx *ctx, int ncol, int nsubcol, int nlay, int cloudLM, int cloudMH,
                        const int32_t *cldy_stoch, int32_t *clearCnts);

#ifdef __cplusplus
}
#endif
#endif /* GEOSRAD_H */
