! gridcomp_shims.F90 -- Fortran view of the GridComp data-path entry points of include/geosrad.h (SURVEY section 8f rows 1-2): what
! GEOS_IrradGridComp's LW_Driver / Update_Flx, GEOS_SolarGridComp's SORADCORE / UPDATE_EXPORT and the parent's RUN call once their
! fields live in device memory.  Field lists are passed as arrays of C pointers indexed by the LWD_* / SWD_* / LWU_* / SWU_* / RT_*
! parameters below (1-based mirrors of the GEOSRAD_* enums); c_null_ptr = "not associated".  The small hip* interface block is all a
! GridComp needs to keep its INTERNAL state on the device between the full calculation and the heartbeat updates.
module geosrad_gridcomp
   use iso_c_binding
   use geosrad_c, only : geosrad_ctx_handle, geosrad_fail, geosrad_data_path, geosrad_load_tables_chou_sw
   implicit none
   private
   public :: lw_driver_rrtmg, lw_driver_rrtmg_rats, lw_update_rats, lw_update_bands, sw_update_surface, sw_driver_rrtmg, sw_driver_chou, lw_chou_post, lw_update_flx, sw_update_export, rad_tendencies
   public :: lit_index, lit_pack, lit_unpack
   public :: dev_alloc, dev_free, dev_put, dev_get, dev_sync

   ! ---- GEOSRAD_LWD_* ----
   integer, parameter, public :: LWD_PLE = 1, LWD_PL = 2, LWD_T = 3, LWD_Q = 4, LWD_O3 = 5, LWD_CH4 = 6, LWD_N2O = 7, LWD_CO2_3D = 8, &
      LWD_CFC11 = 9, LWD_CFC12 = 10, LWD_HCFC22 = 11, LWD_FCLD = 12, LWD_CWC_LIQ = 13, LWD_CWC_ICE = 14, LWD_REFF_LIQ = 15, &
      LWD_REFF_ICE = 16, LWD_TAUA = 17, LWD_SSAA = 18, LWD_TS = 19, LWD_EMIS = 20, LWD_LATS = 21, LWD_T2M = 22, LWD_NIN = 22
   integer, parameter, public :: LWD_C_CO2_FIXED = 1, LWD_C_O2 = 2, LWD_C_CCL4 = 3, LWD_C_AIRMW = 4, LWD_C_H2OMW = 5, LWD_C_O3MW = 6, &
      LWD_C_RGAS = 7, LWD_C_GRAV = 8, LWD_NCONST = 8
   integer, parameter, public :: LWD_FLXU_INT = 1, LWD_FLXD_INT = 2, LWD_FLCU_INT = 3, LWD_FLCD_INT = 4, LWD_DFDTS = 5, LWD_DFDTSC = 6, &
      LWD_DFDTSNA = 7, LWD_DFDTSCNA = 8, LWD_FLX_INT = 9, LWD_FLC_INT = 10, LWD_SFCEM_INT = 11, LWD_TS_INT = 12, LWD_CLDTTLW = 13, &
      LWD_CLDHILW = 14, LWD_CLDMDLW = 15, LWD_CLDLOLW = 16, LWD_OLRB = 17, LWD_DOLRB = 18, LWD_NOUT = 18
   ! ---- GEOSRAD_SWS_* (2-D block of UPDATE_EXPORT) ----
   integer, parameter, public :: SWS_SLR = 1, SWS_ZTH = 2, SWS_ALBVF = 3, SWS_ALBVR = 4, SWS_ALBNF = 5, SWS_ALBNR = 6, SWS_DRUVRN = 7, SWS_DFUVRN = 8, &
      SWS_DRPARN = 9, SWS_DFPARN = 10, SWS_DRNIRN = 11, SWS_DFNIRN = 12, SWS_FSWN = 13, SWS_FSCN = 14, SWS_FSWNAN = 15, SWS_FSCNAN = 16, SWS_NIN = 16
   integer, parameter, public :: SWS_ALBVF_X = 1, SWS_ALBVR_X = 2, SWS_ALBNF_X = 3, SWS_ALBNR_X = 4, SWS_ALBEDO = 5, SWS_SLRTP = 6, SWS_DRUVR = 7, &
      SWS_DFUVR = 8, SWS_DRPAR = 9, SWS_DFPAR = 10, SWS_DRNIR = 11, SWS_DFNIR = 12, SWS_DRNUVR = 13, SWS_DRNPAR = 14, SWS_DRNNIR = 15, SWS_SLRSF = 16, &
      SWS_SLRSFC = 17, SWS_SLRSFNA = 18, SWS_SLRSFCNA = 19, SWS_SLRSUF = 20, SWS_SLRSUFC = 21, SWS_SLRSUFNA = 22, SWS_SLRSUFCNA = 23, SWS_NOUT = 23
   ! ---- GEOSRAD_LWR_* (RATS exports of Update_Flx) ----
   integer, parameter, public :: LWR_FLX_INT = 1, LWR_SFCEM_INT = 2, LWR_DFDTS = 3, LWR_FLX_RAT = 4, LWR_SFCEM_RAT = 5, LWR_DFDTS_RAT = 6, LWR_NIN = 6
   integer, parameter, public :: LWR_DOLR = 1, LWR_DLWS = 2, LWR_DFLNS = 3, LWR_DSFCEM = 4, LWR_NETTRAP = 5, LWR_COLTRAP = 6, LWR_FLX = 7, &
      LWR_DFDTS_OUT = 8, LWR_NOUT = 8
   integer, parameter, public :: LWD_FLXU_RAT = 1, LWD_FLXD_RAT = 2, LWD_FLX_RAT = 3, LWD_DFDTS_RAT = 4, LWD_SFCEM_RAT = 5, LWD_NRATOUT = 5
   ! ---- GEOSRAD_LWC_* ----
   integer, parameter, public :: LWC_FLXU_INT = 1, LWC_FLCU_INT = 2, LWC_FLAU_INT = 3, LWC_FLXAU_INT = 4, LWC_FLXD_INT = 5, LWC_FLCD_INT = 6, &
      LWC_FLAD_INT = 7, LWC_FLXAD_INT = 8, LWC_DFDTS = 9, LWC_TS = 10, LWC_NIN = 10
   integer, parameter, public :: LWC_SFCEM_INT = 1, LWC_FLX_INT = 2, LWC_FLXA_INT = 3, LWC_FLC_INT = 4, LWC_FLA_INT = 5, LWC_DFDTSC = 6, &
      LWC_DFDTSNA = 7, LWC_DFDTSCNA = 8, LWC_TS_INT = 9, LWC_NOUT = 9
   ! ---- GEOSRAD_SWC_* ----
   integer, parameter, public :: SWC_PLE = 1, SWC_T = 2, SWC_Q = 3, SWC_OX = 4, SWC_CL = 5, SWC_QI = 6, SWC_QL = 7, SWC_QR = 8, SWC_QS = 9, &
      SWC_RI = 10, SWC_RL = 11, SWC_RR = 12, SWC_RS = 13, SWC_TAUA = 14, SWC_SSAA = 15, SWC_ASYA = 16, SWC_ZT = 17, SWC_ALBVR = 18, &
      SWC_ALBVF = 19, SWC_ALBNR = 20, SWC_ALBNF = 21, SWC_NIN = 21
   integer, parameter, public :: SWC_C_CO2 = 1, SWC_C_O3MW = 2, SWC_C_AIRMW = 3, SWC_C_UNDEF = 4, SWC_NCONST = 4
   integer, parameter, public :: SWC_FSW = 1, SWC_FSC = 2, SWC_FSWU = 3, SWC_FSCU = 4, SWC_NIRR = 5, SWC_NIRF = 6, SWC_PARR = 7, &
      SWC_PARF = 8, SWC_UVRR = 9, SWC_UVRF = 10, SWC_FSWBAND = 11, SWC_DRBAND = 12, SWC_DFBAND = 13, SWC_NOUT = 13
   ! ---- GEOSRAD_SWD_* ----
   integer, parameter, public :: SWD_PLE = 1, SWD_PL = 2, SWD_T = 3, SWD_Q = 4, SWD_O3 = 5, SWD_CH4 = 6, SWD_CL = 7, SWD_TS = 8, &
      SWD_QQ_ICE = 9, SWD_QQ_LIQ = 10, SWD_RR_ICE = 11, SWD_RR_LIQ = 12, SWD_TAUA = 13, SWD_SSAA = 14, SWD_ASYA = 15, SWD_ZT = 16, &
      SWD_ALAT = 17, SWD_ALBVR = 18, SWD_ALBVF = 19, SWD_ALBNR = 20, SWD_ALBNF = 21, SWD_NIN = 21
   integer, parameter, public :: SWD_C_CO2 = 1, SWD_C_O2 = 2, SWD_C_AIRMW = 3, SWD_C_H2OMW = 4, SWD_C_O3MW = 5, SWD_C_RGAS = 6, &
      SWD_C_GRAV = 7, SWD_C_UNDEF = 8, SWD_NCONST = 8
   integer, parameter, public :: SWD_FSW = 1, SWD_FSC = 2, SWD_FSWU = 3, SWD_FSCU = 4, SWD_NIRR = 5, SWD_NIRF = 6, SWD_PARR = 7, &
      SWD_PARF = 8, SWD_UVRR = 9, SWD_UVRF = 10, SWD_FSWBAND = 11, SWD_CLDTS = 12, SWD_CLDHS = 13, SWD_CLDMS = 14, SWD_CLDLS = 15, &
      SWD_COTTP = 16, SWD_COTHP = 17, SWD_COTMP = 18, SWD_COTLP = 19, SWD_FSWNA = 20, SWD_FSCNA = 21, SWD_FSWUNA = 22, SWD_FSCUNA = 23, &
      SWD_FSWBANDNA = 24, SWD_NOUT = 24
   ! ---- GEOSRAD_LWU_* ----
   integer, parameter, public :: LWU_TSINST = 1, LWU_TS_INT = 2, LWU_SFCEM_INT = 3, LWU_FCLD = 4, LWU_FLX_INT = 5, LWU_FLXA_INT = 6, &
      LWU_FLC_INT = 7, LWU_FLA_INT = 8, LWU_FLXU_INT = 9, LWU_FLXAU_INT = 10, LWU_FLCU_INT = 11, LWU_FLAU_INT = 12, LWU_FLXD_INT = 13, &
      LWU_FLXAD_INT = 14, LWU_FLCD_INT = 15, LWU_FLAD_INT = 16, LWU_DFDTS = 17, LWU_DFDTSNA = 18, LWU_DFDTSC = 19, LWU_DFDTSCNA = 20, &
      LWU_NIN = 20
   integer, parameter, public :: LWU_FLX = 1, LWU_FLXA = 2, LWU_FLC = 3, LWU_FLA = 4, LWU_FLXU = 5, LWU_FLXAU = 6, LWU_FLCU = 7, &
      LWU_FLAU = 8, LWU_FLXD = 9, LWU_FLXAD = 10, LWU_FLCD = 11, LWU_FLAD = 12, LWU_OLR = 13, LWU_OLRA = 14, LWU_OLC = 15, LWU_OLA = 16, &
      LWU_OLCC5 = 17, LWU_DSFDTS = 18, LWU_SFCEM = 19, LWU_LWS = 20, LWU_LWSA = 21, LWU_LCS = 22, LWU_LAS = 23, LWU_LCSC5 = 24, &
      LWU_FLNS = 25, LWU_FLNSNA = 26, LWU_FLNSC = 27, LWU_FLNSA = 28, LWU_DSFDTS0 = 29, LWU_SFCEM0 = 30, LWU_TSREFF = 31, &
      LWU_CLDTT = 32, LWU_NOUT = 32
   ! ---- GEOSRAD_SWU_* ----
   integer, parameter, public :: SWU_SLR = 1, SWU_FSWN = 2, SWU_FSCN = 3, SWU_FSWNAN = 4, SWU_FSCNAN = 5, SWU_FSWUN = 6, SWU_FSCUN = 7, &
      SWU_FSWUNAN = 8, SWU_FSCUNAN = 9, SWU_FSWBANDN = 10, SWU_FSWBANDNAN = 11, SWU_NIN = 11
   integer, parameter, public :: SWU_FSW = 1, SWU_FSC = 2, SWU_FSWNA = 3, SWU_FSCNA = 4, SWU_FSWU = 5, SWU_FSCU = 6, SWU_FSWUNA = 7, &
      SWU_FSCUNA = 8, SWU_FSWD = 9, SWU_FSCD = 10, SWU_FSWDNA = 11, SWU_FSCDNA = 12, SWU_FSWBAND = 13, SWU_FSWBANDNA = 14, SWU_RSR = 15, &
      SWU_RSC = 16, SWU_RSRNA = 17, SWU_RSCNA = 18, SWU_RSRS = 19, SWU_RSCS = 20, SWU_RSRSNA = 21, SWU_RSCSNA = 22, SWU_OSR = 23, &
      SWU_OSRCLR = 24, SWU_OSRNA = 25, SWU_OSRCNA = 26, SWU_NOUT = 26
   ! ---- GEOSRAD_RT_* ----
   integer, parameter, public :: RT_PLE = 1, RT_FLW = 2, RT_FSW = 3, RT_FLWCLR = 4, RT_FSWCLR = 5, RT_FSWNA = 6, RT_FLA = 7, RT_FSCNA = 8, &
      RT_DSFDTS = 9, RT_SFCEM = 10, RT_TRD = 11, RT_NIN = 11
   integer, parameter, public :: RT_DTDT = 1, RT_RADLW = 2, RT_RADSW = 3, RT_RADLWC = 4, RT_RADSWC = 5, RT_RADSWNA = 6, RT_RADLWCNA = 7, &
      RT_RADSWCNA = 8, RT_BLW = 9, RT_ALW = 10, RT_RADSRF = 11, RT_NOUT = 11

   interface
      integer(c_int) function geosrad_lw_driver_rrtmg_dev(ctx, stream, ncol, lm, nb_aer, fin, consts, iceflglw, liqflglw, doy, lcldlm, &
            lcldmh, band_output, fout) bind(C, name='geosrad_lw_driver_rrtmg_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, nb_aer, iceflglw, liqflglw, doy, lcldlm, lcldmh
         type(c_ptr), intent(in) :: fin(*), fout(*)
         real(c_double), intent(in) :: consts(*)
         integer(c_int), intent(in) :: band_output(*)
      end function
      integer(c_int) function geosrad_lw_driver_rrtmg_rats_dev(ctx, stream, ncol, lm, nb_aer, fin, consts, iceflglw, liqflglw, doy, lcldlm, &
            lcldmh, band_output, fout, nrats, rat_gas, rat_out) bind(C, name='geosrad_lw_driver_rrtmg_rats_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, nb_aer, iceflglw, liqflglw, doy, lcldlm, lcldmh, nrats
         type(c_ptr), intent(in) :: fin(*), fout(*), rat_out(*)
         real(c_double), intent(in) :: consts(*)
         integer(c_int), intent(in) :: band_output(*), rat_gas(*)
      end function
      integer(c_int) function geosrad_lw_update_rats_dev(ctx, stream, ncol, lm, nrats, fin, fout) bind(C, name='geosrad_lw_update_rats_dev')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, nrats
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_lw_update_bands_dev(ctx, stream, ncol, band_output, wavenum1, wavenum2, undef, tsinst, ts_int, olrb_int, &
            dolrb_int, olrb_exp, tbrb_exp) bind(C, name='geosrad_lw_update_bands_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream, tsinst, ts_int, olrb_int, dolrb_int, olrb_exp, tbrb_exp
         integer(c_int), value :: ncol
         integer(c_int), intent(in) :: band_output(*)
         real(c_double), intent(in) :: wavenum1(*), wavenum2(*)
         real(c_double), value :: undef
      end function
      integer(c_int) function geosrad_sw_update_surface_dev(ctx, stream, ncol, lm, undef, fin, fout) bind(C, name='geosrad_sw_update_surface_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm
         real(c_double), value :: undef
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_sw_driver_rrtmg_dev(ctx, stream, ncol, lm, nb_aer, fin, consts, iceflgsw, liqflgsw, sc, dist, isolvar, &
            dyofyr, include_aerosols, lcldlm, lcldmh, normflx, bndsolvar, indsolvar, fout) bind(C, name='geosrad_sw_driver_rrtmg_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream, bndsolvar, indsolvar
         integer(c_int), value :: ncol, lm, nb_aer, iceflgsw, liqflgsw, isolvar, dyofyr, include_aerosols, lcldlm, lcldmh, normflx
         real(c_double), value :: sc, dist
         type(c_ptr), intent(in) :: fin(*), fout(*)
         real(c_double), intent(in) :: consts(*)
      end function
      integer(c_int) function geosrad_sw_driver_chou_dev(ctx, stream, ncol, lm, fin, consts, lcldmh, lcldlm, hk_uv, hk_ir, do_drfband, fout) &
            bind(C, name='geosrad_sw_driver_chou_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, lcldmh, lcldlm, do_drfband
         type(c_ptr), intent(in) :: fin(*), fout(*)
         real(c_double), intent(in) :: consts(*)
         real, intent(in) :: hk_uv(*), hk_ir(*)
      end function
      integer(c_int) function geosrad_lw_chou_post_dev(ctx, stream, ncol, lm, fin, fout) bind(C, name='geosrad_lw_chou_post_dev')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_lw_update_flx_dev(ctx, stream, ncol, lm, rrtmg, lev_mid_high, lev_low_mid, undef, fin, fout) &
            bind(C, name='geosrad_lw_update_flx_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, rrtmg, lev_mid_high, lev_low_mid
         real(c_double), value :: undef
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_sw_update_export_dev(ctx, stream, ncol, lm, nbands, fin, fout) bind(C, name='geosrad_sw_update_export_dev')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm, nbands
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_rad_tendencies_dev(ctx, stream, ncol, lm, grav, cp, fin, fout) bind(C, name='geosrad_rad_tendencies_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream
         integer(c_int), value :: ncol, lm
         real(c_double), value :: grav, cp
         type(c_ptr), intent(in) :: fin(*), fout(*)
      end function
      integer(c_int) function geosrad_lit_index_dev(ctx, stream, ncol, zth, lit_index, lit_pos, nlit_dev, nlit_host) bind(C, name='geosrad_lit_index_dev')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream, zth, lit_index, lit_pos, nlit_dev
         integer(c_int), value :: ncol
         integer(c_int), intent(out) :: nlit_host
      end function
      integer(c_int) function geosrad_lit_pack_dev(ctx, stream, pdim, udim, nlev, lit_index, nlit_dev, unpacked, packed) bind(C, name='geosrad_lit_pack_dev')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream, lit_index, nlit_dev, unpacked, packed
         integer(c_int), value :: pdim, udim, nlev
      end function
      integer(c_int) function geosrad_lit_unpack_dev(ctx, stream, pdim, udim, nlev, lit_pos, packed, unpacked, use_default, dflt) &
            bind(C, name='geosrad_lit_unpack_dev')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx, stream, lit_pos, packed, unpacked
         integer(c_int), value :: pdim, udim, nlev, use_default
         real(c_double), value :: dflt
      end function
      integer(c_int) function geosrad_check(ctx, stream) bind(C, name='geosrad_check')
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx, stream
      end function
      ! HIP runtime (libamdhip64): device memory for the GridComp's fields
      integer(c_int) function hipMalloc(p, bytes) bind(C, name='hipMalloc')
         import :: c_int, c_ptr, c_size_t
         type(c_ptr), intent(out) :: p
         integer(c_size_t), value :: bytes
      end function
      integer(c_int) function hipFree(p) bind(C, name='hipFree')
         import :: c_int, c_ptr
         type(c_ptr), value :: p
      end function
      integer(c_int) function hipMemcpy(dst, src, bytes, kind) bind(C, name='hipMemcpy')
         import :: c_int, c_ptr, c_size_t
         type(c_ptr), value :: dst, src
         integer(c_size_t), value :: bytes
         integer(c_int), value :: kind
      end function
      integer(c_int) function hipDeviceSynchronize() bind(C, name='hipDeviceSynchronize')
         import :: c_int
      end function
   end interface

contains

   function dev_alloc(nreal) result(p)
      integer, intent(in) :: nreal
      type(c_ptr) :: p
      real :: x
      type(c_ptr) :: h
      h = geosrad_ctx_handle()          ! selects the device
      if (hipMalloc(p, int(nreal, c_size_t) * int(storage_size(x) / 8, c_size_t)) /= 0) error stop 'geosrad_gridcomp: hipMalloc failed'
   end function
   subroutine dev_free(p)
      type(c_ptr), intent(inout) :: p
      if (c_associated(p)) then
         if (hipFree(p) /= 0) error stop 'geosrad_gridcomp: hipFree failed'
      end if
      p = c_null_ptr
   end subroutine
   subroutine dev_put(p, a, n)
      type(c_ptr), intent(in) :: p
      integer, intent(in) :: n
      real, intent(in), target :: a(n)
      if (hipMemcpy(p, c_loc(a), int(n, c_size_t) * int(storage_size(a) / 8, c_size_t), 1_c_int) /= 0) error stop 'geosrad_gridcomp: H2D copy failed'
   end subroutine
   subroutine dev_get(a, p, n)
      integer, intent(in) :: n
      real, intent(out), target :: a(n)
      type(c_ptr), intent(in) :: p
      if (hipMemcpy(c_loc(a), p, int(n, c_size_t) * int(storage_size(a) / 8, c_size_t), 2_c_int) /= 0) error stop 'geosrad_gridcomp: D2H copy failed'
   end subroutine
   subroutine dev_sync()
      if (geosrad_check(geosrad_ctx_handle(), c_null_ptr) /= 0) call geosrad_fail('geosrad_gridcomp')
      if (hipDeviceSynchronize() /= 0) error stop 'geosrad_gridcomp: device error'
   end subroutine

   ! RRTMG branch of LW_Driver (GEOS_IrradGridComp.F90:3188-3615); lcldlm / lcldmh in model ordering
   subroutine lw_driver_rrtmg(ncol, lm, nb_aer, fin, consts, iceflglw, liqflglw, doy, lcldlm, lcldmh, band_output, fout)
      integer, intent(in) :: ncol, lm, nb_aer, iceflglw, liqflglw, doy, lcldlm, lcldmh
      type(c_ptr), intent(in) :: fin(LWD_NIN), fout(LWD_NOUT)
      real(c_double), intent(in) :: consts(LWD_NCONST)
      logical, intent(in) :: band_output(16)
      integer(c_int) :: bo(16)
      bo = merge(1_c_int, 0_c_int, band_output)
      if (geosrad_lw_driver_rrtmg_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), int(nb_aer,c_int), fin, consts, &
            int(iceflglw,c_int), int(liqflglw,c_int), int(doy,c_int), int(lcldlm,c_int), int(lcldmh,c_int), bo, fout) /= 0) &
         call geosrad_fail('LW_Driver (RRTMG)')
   end subroutine

   ! the same with the RATS loop (GEOS_IrradGridComp.F90:3389-3469, :3522-3530, :3614): nameRATS as in AGCM.rc's RATS_DIAGNOSTICS;
   ! rat_out = INTERNAL FLXU_RAT, FLXD_RAT, FLX_RAT, DFDTS_RAT (IM,JM,0:LM,nRATS), SFCEM_RAT (IM,JM,nRATS) on the device
   subroutine lw_driver_rrtmg_rats(ncol, lm, nb_aer, fin, consts, iceflglw, liqflglw, doy, lcldlm, lcldmh, band_output, fout, nrats, &
         nameRATS, rat_out)
      integer, intent(in) :: ncol, lm, nb_aer, iceflglw, liqflglw, doy, lcldlm, lcldmh, nrats
      type(c_ptr), intent(in) :: fin(LWD_NIN), fout(LWD_NOUT), rat_out(LWD_NRATOUT)
      real(c_double), intent(in) :: consts(LWD_NCONST)
      logical, intent(in) :: band_output(16)
      character(len=*), intent(in) :: nameRATS(nrats)
      integer(c_int) :: bo(16), gas(8)
      integer :: n
      if (nrats > 8) error stop 'LW_Driver (RRTMG): more than 8 RATS'
      do n = 1, nrats
         select case (trim(nameRATS(n)))
         case ('H2O');    gas(n) = 0
         case ('O3');     gas(n) = 1
         case ('CO2');    gas(n) = 2
         case ('CH4');    gas(n) = 3
         case ('N2O');    gas(n) = 4
         case ('CFC11');  gas(n) = 5
         case ('CFC12');  gas(n) = 6
         case ('HCFC22'); gas(n) = 7
         case default;    error stop 'LW_Driver (RRTMG): unknown RAT name'
         end select
      end do
      bo = merge(1_c_int, 0_c_int, band_output)
      if (geosrad_lw_driver_rrtmg_rats_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), int(nb_aer,c_int), fin, &
            consts, int(iceflglw,c_int), int(liqflglw,c_int), int(doy,c_int), int(lcldlm,c_int), int(lcldmh,c_int), bo, fout, &
            int(nrats,c_int), gas, rat_out) /= 0) call geosrad_fail('LW_Driver (RRTMG, RATS)')
   end subroutine

   ! RATS exports of Update_Flx (GEOS_IrradGridComp.F90:4036-4120): fout(LWR_DOLR) = device array (IM,JM,nRATS) whose slice n is the export
   ! 'dOLR_'//nameRATS(n), ... ; c_null_ptr = not associated
   subroutine lw_update_rats(ncol, lm, nrats, fin, fout)
      integer, intent(in) :: ncol, lm, nrats
      type(c_ptr), intent(in) :: fin(LWR_NIN), fout(LWR_NOUT)
      if (geosrad_lw_update_rats_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), int(nrats,c_int), fin, fout) /= 0) &
         call geosrad_fail('Update_Flx (RATS)')
   end subroutine

   ! band OLR / brightness temperature exports of Update_Flx (GEOS_IrradGridComp.F90:3993-4021): wavenum1 / wavenum2 = rrlw_wvn's [cm-1];
   ! olrb_int / dolrb_int = the driver's OLRB / DOLRB (16,IM*JM); olrb_exp / tbrb_exp (IM,JM,16) device arrays or c_null_ptr
   subroutine lw_update_bands(ncol, band_output, wavenum1, wavenum2, undef, tsinst, ts_int, olrb_int, dolrb_int, olrb_exp, tbrb_exp)
      integer, intent(in) :: ncol
      logical, intent(in) :: band_output(16)
      real, intent(in) :: wavenum1(16), wavenum2(16), undef
      type(c_ptr), intent(in) :: tsinst, ts_int, olrb_int, dolrb_int, olrb_exp, tbrb_exp
      integer(c_int) :: bo(16)
      bo = merge(1_c_int, 0_c_int, band_output)
      if (geosrad_lw_update_bands_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), bo, real(wavenum1,c_double), real(wavenum2,c_double), &
            real(undef,c_double), tsinst, ts_int, olrb_int, dolrb_int, olrb_exp, tbrb_exp) /= 0) call geosrad_fail('Update_Flx (band OLR)')
   end subroutine

   ! 2-D block of UPDATE_EXPORT (GEOS_SolarGridComp.F90:7403-7533): albedo exports, ALBEDO, incident and surface fluxes
   subroutine sw_update_surface(ncol, lm, undef, fin, fout)
      integer, intent(in) :: ncol, lm
      real, intent(in) :: undef
      type(c_ptr), intent(in) :: fin(SWS_NIN), fout(SWS_NOUT)
      if (geosrad_sw_update_surface_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), real(undef,c_double), fin, fout) /= 0) &
         call geosrad_fail('UPDATE_EXPORT (surface)')
   end subroutine

   ! RRTMG branch of SORADCORE (GEOS_SolarGridComp.F90:6113-6450) on the packed daytime columns
   subroutine sw_driver_rrtmg(ncol, lm, nb_aer, fin, consts, iceflgsw, liqflgsw, sc, dist, isolvar, dyofyr, include_aerosols, lcldlm, &
         lcldmh, fout, rc)
      integer, intent(in) :: ncol, lm, nb_aer, iceflgsw, liqflgsw, isolvar, dyofyr, lcldlm, lcldmh
      logical, intent(in) :: include_aerosols
      real, intent(in) :: sc, dist
      type(c_ptr), intent(in) :: fin(SWD_NIN), fout(SWD_NOUT)
      real(c_double), intent(in) :: consts(SWD_NCONST)
      integer, intent(out) :: rc
      rc = geosrad_sw_driver_rrtmg_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), int(nb_aer,c_int), fin, consts, &
            int(iceflgsw,c_int), int(liqflgsw,c_int), real(sc,c_double), real(dist,c_double), int(isolvar,c_int), int(dyofyr,c_int), &
            merge(1_c_int, 0_c_int, include_aerosols), int(lcldlm,c_int), int(lcldmh,c_int), 1_c_int, c_null_ptr, c_null_ptr, fout)
   end subroutine

   ! Chou-Suarez branch of SORADCORE (GEOS_SolarGridComp.F90:4484-4572 and the SHRTWAVE cover :6597-6672) on the packed daytime columns;
   ! hk_uv_temp / hk_ir_temp: the GridComp's band weights (:2997-3028)
   subroutine sw_driver_chou(ncol, lm, fin, consts, lcldmh, lcldlm, hk_uv_temp, hk_ir_temp, do_drfband, fout, rc)
      integer, intent(in) :: ncol, lm, lcldmh, lcldlm
      type(c_ptr), intent(in) :: fin(SWC_NIN), fout(SWC_NOUT)
      real(c_double), intent(in) :: consts(SWC_NCONST)
      real, intent(in) :: hk_uv_temp(5), hk_ir_temp(3,10)
      logical, intent(in) :: do_drfband
      integer, intent(out) :: rc
      real :: x
      logical, save :: loaded = .false.
      if (.not. loaded) then       ! sorad's coefficient tables (the reference keeps them as module data in sorad_constants)
         if (kind(x) == 4) then
            rc = geosrad_load_tables_chou_sw(geosrad_ctx_handle(), geosrad_data_path('chou_sw_r4.grtb'))
         else
            rc = geosrad_load_tables_chou_sw(geosrad_ctx_handle(), geosrad_data_path('chou_sw_r8.grtb'))
         end if
         if (rc /= 0) return
         loaded = .true.
      end if
      rc = geosrad_sw_driver_chou_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), fin, consts, int(lcldmh,c_int), &
            int(lcldlm,c_int), hk_uv_temp, hk_ir_temp, merge(1_c_int, 0_c_int, do_drfband), fout)
   end subroutine

   ! after `call IRRAD` in the Chou-Suarez branch of LW_Driver (GEOS_IrradGridComp.F90:2101-2108, :3601-3616)
   subroutine lw_chou_post(ncol, lm, fin, fout)
      integer, intent(in) :: ncol, lm
      type(c_ptr), intent(in) :: fin(LWC_NIN), fout(LWC_NOUT)
      if (geosrad_lw_chou_post_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), fin, fout) /= 0) &
         call geosrad_fail('LW_Driver (Chou-Suarez)')
   end subroutine

   ! Update_Flx (GEOS_IrradGridComp.F90:3796-3999)
   subroutine lw_update_flx(ncol, lm, use_rrtmg, lev_mid_high, lev_low_mid, undef, fin, fout)
      integer, intent(in) :: ncol, lm, lev_mid_high, lev_low_mid
      logical, intent(in) :: use_rrtmg
      real, intent(in) :: undef
      type(c_ptr), intent(in) :: fin(LWU_NIN), fout(LWU_NOUT)
      if (geosrad_lw_update_flx_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), merge(1_c_int, 0_c_int, use_rrtmg), &
            int(lev_mid_high,c_int), int(lev_low_mid,c_int), real(undef,c_double), fin, fout) /= 0) call geosrad_fail('Update_Flx')
   end subroutine

   ! flux part of UPDATE_EXPORT (GEOS_SolarGridComp.F90:7540-7579)
   subroutine sw_update_export(ncol, lm, nbands, fin, fout)
      integer, intent(in) :: ncol, lm, nbands
      type(c_ptr), intent(in) :: fin(SWU_NIN), fout(SWU_NOUT)
      if (geosrad_sw_update_export_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), int(nbands,c_int), fin, fout) /= 0) &
         call geosrad_fail('UPDATE_EXPORT')
   end subroutine

   ! heating rates of the parent (GEOS_RadiationGridComp.F90:798-819)
   subroutine rad_tendencies(ncol, lm, grav, cp, fin, fout)
      integer, intent(in) :: ncol, lm
      real, intent(in) :: grav, cp
      type(c_ptr), intent(in) :: fin(RT_NIN), fout(RT_NOUT)
      if (geosrad_rad_tendencies_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), int(lm,c_int), real(grav,c_double), real(cp,c_double), &
            fin, fout) /= 0) call geosrad_fail('GEOS_RadiationGridComp RUN')
   end subroutine

   ! lit-column compaction on device fields (GEOS_SolarGridComp.F90:3686 `daytime = ZTH > 0.`; PackIt / UnPackIt :7753-7799).
   ! zth (ncol) reals; lit_idx, lit_pos (ncol) and nlit_dev (1) default integers, all device addresses.
   subroutine lit_index(ncol, zth, lit_idx, lit_pos, nlit_dev, NumLit)
      integer, intent(in) :: ncol
      type(c_ptr), intent(in) :: zth, lit_idx, lit_pos, nlit_dev
      integer, intent(out) :: NumLit
      integer(c_int) :: n
      if (geosrad_lit_index_dev(geosrad_ctx_handle(), c_null_ptr, int(ncol,c_int), zth, lit_idx, lit_pos, nlit_dev, n) /= 0) &
         call geosrad_fail('daytime compaction')
      NumLit = n
   end subroutine
   subroutine lit_pack(Packed, UnPacked, lit_idx, nlit_dev, Pdim, Udim, LM)      ! PackIt's argument order, device addresses
      type(c_ptr), intent(in) :: Packed, UnPacked, lit_idx, nlit_dev
      integer, intent(in) :: Pdim, Udim, LM
      if (geosrad_lit_pack_dev(geosrad_ctx_handle(), c_null_ptr, int(Pdim,c_int), int(Udim,c_int), int(LM,c_int), lit_idx, nlit_dev, &
            UnPacked, Packed) /= 0) call geosrad_fail('PackIt')
   end subroutine
   subroutine lit_unpack(Packed, UnPacked, lit_pos, Pdim, Udim, LM, DEFAULT)    ! UnPackIt's argument order, device addresses
      type(c_ptr), intent(in) :: Packed, UnPacked, lit_pos
      integer, intent(in) :: Pdim, Udim, LM
      real, intent(in), optional :: DEFAULT
      real(c_double) :: d
      d = 0
      if (present(DEFAULT)) d = DEFAULT
      if (geosrad_lit_unpack_dev(geosrad_ctx_handle(), c_null_ptr, int(Pdim,c_int), int(Udim,c_int), int(LM,c_int), lit_pos, Packed, &
            UnPacked, merge(1_c_int, 0_c_int, present(DEFAULT)), d) /= 0) call geosrad_fail('UnPackIt')
   end subroutine
end module geosrad_gridcomp
