! geosrad_c.F90 -- ISO_C_BINDING view of include/geosrad.h plus the one per-process context that replaces the
! reference's module-level state.  Everything here is host-side plumbing; the arithmetic lives in libgeosrad.so.
module geosrad_c
   use iso_c_binding
   use iso_fortran_env, only : error_unit
   implicit none
   private
   public :: geosrad_ctx_handle, geosrad_fail, geosrad_warn, geosrad_data_path, geosrad_load_tables_sw, geosrad_rrtmg_sw, geosrad_load_tables_chou_lw, geosrad_load_tables_chou_sw, geosrad_irrad, geosrad_sorad
   public :: geosrad_create, geosrad_destroy, geosrad_last_error, geosrad_load_tables_lw, geosrad_load_inhomogeneity
   public :: geosrad_set_corr_lengths, geosrad_rrtmg_lw, geosrad_mcica, geosrad_clearcounts, geosrad_read_table

   type(c_ptr), save :: ctx = c_null_ptr

   interface
      integer(c_int) function geosrad_create(ctx, device_id, real_kind) bind(C, name='geosrad_create')
         import; type(c_ptr), intent(out) :: ctx; integer(c_int), value :: device_id, real_kind
      end function
      integer(c_int) function geosrad_destroy(ctx) bind(C, name='geosrad_destroy')
         import; type(c_ptr), value :: ctx
      end function
      type(c_ptr) function geosrad_last_error(ctx) bind(C, name='geosrad_last_error')
         import; type(c_ptr), value :: ctx
      end function
      integer(c_int) function geosrad_load_tables_lw(ctx, path) bind(C, name='geosrad_load_tables_lw')
         import; type(c_ptr), value :: ctx; character(kind=c_char), intent(in) :: path(*)
      end function
      integer(c_int) function geosrad_load_tables_sw(ctx, path) bind(C, name='geosrad_load_tables_sw')
         import :: c_int, c_ptr, c_char
         type(c_ptr), value :: ctx
         character(kind=c_char) :: path(*)
      end function
      integer(c_int) function geosrad_rrtmg_sw(ctx, rpart, ncol, nlay, scon, adjes, coszen, isolvar, play, plev, tlay, &
            h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, iceflgsw, liqflgsw, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, &
            iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, cloudLM, cloudMH, normFlx, clearCounts, &
            swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
            cotdtp, cotdhp, cotdmp, cotdlp, cotntp, cotnhp, cotnmp, cotnlp, do_drfband, drband, dfband, bndscl, indsolvar, &
            solcycfrac) bind(C, name='geosrad_rrtmg_sw')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: rpart, ncol, nlay, isolvar, iceflgsw, liqflgsw, dyofyr, iaer, cloudLM, cloudMH, normFlx, do_drfband
         real(c_double), value :: scon, adjes
         type(c_ptr), value :: coszen, play, plev, tlay, h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, cld, ciwp, clwp, rei, rel, zm, alat, &
            tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, clearCounts, swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, &
            parf, uvrr, uvrf, fswband, cotdtp, cotdhp, cotdmp, cotdlp, cotntp, cotnhp, cotnmp, cotnlp, drband, dfband, bndscl, indsolvar, &
            solcycfrac
      end function
      integer(c_int) function geosrad_load_tables_chou_lw(ctx, path) bind(C, name='geosrad_load_tables_chou_lw')
         import :: c_int, c_ptr, c_char
         type(c_ptr), value :: ctx
         character(kind=c_char) :: path(*)
      end function
      integer(c_int) function geosrad_load_tables_chou_sw(ctx, path) bind(C, name='geosrad_load_tables_chou_sw')
         import :: c_int, c_ptr, c_char
         type(c_ptr), value :: ctx
         character(kind=c_char) :: path(*)
      end function
      integer(c_int) function geosrad_irrad(ctx, m, np, ple, ta, wa, oa, tb, co2, trace, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, ict, icb, &
            reff, ns, fs, tg, eg, tv, ev, rv, na, nb, taua, ssaa, asya, flxu, flcu, flau, flxau, flxd, flcd, flad, flxad, dfdts, sfcem, &
            taudiag) bind(C, name='geosrad_irrad')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: m, np, trace, ict, icb, ns, na, nb
         real(c_double), value :: co2
         type(c_ptr), value :: ple, ta, wa, oa, tb, n2o, ch4, cfc11, cfc12, cfc22, cwc, fcld, reff, fs, tg, eg, tv, ev, rv, taua, ssaa, asya, &
            flxu, flcu, flau, flxau, flxd, flcd, flad, flxad, dfdts, sfcem, taudiag
      end function
      integer(c_int) function geosrad_sorad(ctx, m, np, nb, cosz, pl, ta, wa, oa, co2, cwc, fcld, ict, icb, reff, hk_uv, hk_ir, taua, ssaa, &
            asya, rsuvbm, rsuvdf, rsirbm, rsirdf, flx, flc, fdiruv, fdifuv, fdirpar, fdifpar, fdirir, fdifir, flxu, flcu, flx_sfc_band, &
            do_drfband, drband, dfband) bind(C, name='geosrad_sorad')
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: ctx
         integer(c_int), value :: m, np, nb, ict, icb, do_drfband
         real(c_double), value :: co2
         type(c_ptr), value :: cosz, pl, ta, wa, oa, cwc, fcld, reff, hk_uv, hk_ir, taua, ssaa, asya, rsuvbm, rsuvdf, rsirbm, rsirdf, flx, flc, &
            fdiruv, fdifuv, fdirpar, fdifpar, fdirir, fdifir, flxu, flcu, flx_sfc_band, drband, dfband
      end function
      integer(c_int) function geosrad_load_inhomogeneity(ctx, ih, path) bind(C, name='geosrad_load_inhomogeneity')
         import; type(c_ptr), value :: ctx; integer(c_int), value :: ih; character(kind=c_char), intent(in) :: path(*)
      end function
      integer(c_int) function geosrad_read_table(path, name, real_kind, dst, count) bind(C, name='geosrad_read_table')
         import; character(kind=c_char), intent(in) :: path(*), name(*); integer(c_int), value :: real_kind
         type(c_ptr), value :: dst; integer(c_size_t), value :: count
      end function
      integer(c_int) function geosrad_set_corr_lengths(ctx, adl, rdl) bind(C, name='geosrad_set_corr_lengths')
         import; type(c_ptr), value :: ctx; real(c_double), intent(in) :: adl(4), rdl(4)
      end function
      integer(c_int) function geosrad_rrtmg_lw(ctx, ncol, nlay, psize, dudTs, play, plev, tlay, tlev, tsfc, emis, &
            h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, &
            cldf, ciwp, clwp, rei, rel, iceflglw, liqflglw, tauaer, zm, alat, dyofyr, cloudLM, cloudMH, clearCounts, &
            uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, band_output, olrb, dolrb_dTs) bind(C, name='geosrad_rrtmg_lw')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: ncol, nlay, psize, dudTs, iceflglw, liqflglw, dyofyr, cloudLM, cloudMH
         type(c_ptr), value :: play, plev, tlay, tlev, tsfc, emis, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
            cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, cldf, ciwp, clwp, rei, rel, tauaer, zm, alat, clearCounts, &
            uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, band_output, olrb, dolrb_dTs
      end function
      integer(c_int) function geosrad_mcica(ctx, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny, &
            seed_order, cldy, ciwp_s, clwp_s) bind(C, name='geosrad_mcica')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: ncol, nsubcol, nlay, doy
         real(c_double), value :: cwp_tiny
         type(c_ptr), value :: zmid, alat, play, cldfrac, ciwp, clwp, seed_order, cldy, ciwp_s, clwp_s
      end function
      integer(c_int) function geosrad_clearcounts(ctx, ncol, nsubcol, nlay, cloudLM, cloudMH, cldy, cnt) bind(C, name='geosrad_clearcounts')
         import
         type(c_ptr), value :: ctx
         integer(c_int), value :: ncol, nsubcol, nlay, cloudLM, cloudMH
         type(c_ptr), value :: cldy, cnt
      end function
   end interface

contains

   ! the process-wide context, created on first use in the default real kind on the device the library picks for this process
   ! (GEOSRAD_DEVICE_AUTO = -1: GEOSRAD_DEVICE if set, else the launcher's node-local MPI rank - OMPI_COMM_WORLD_LOCAL_RANK, SLURM_LOCALID,
   ! MV2_COMM_WORLD_LOCAL_RANK, ... - modulo the number of visible devices, so the ranks of a node spread over its GPUs unconfigured)
   function geosrad_ctx_handle() result(h)
      type(c_ptr) :: h
      integer(c_int) :: rc
      real :: x
      if (.not. c_associated(ctx)) then
         rc = geosrad_create(ctx, -1_c_int, int(kind(x), c_int))
         if (rc /= 0) then
            write(error_unit,*) 'geosrad_create failed, rc =', rc, ' (no usable HIP device? there is no CPU fallback)'
            error stop 'geosrad: cannot create context'
         end if
      end if
      h = ctx
   end function

   ! directory holding the GRTB table blobs: $GEOSRAD_DATA
   function geosrad_data_path(name) result(p)
      character(*), intent(in) :: name
      character(len=:), allocatable :: p
      character(len=1024) :: dir
      integer :: stat
      call get_environment_variable('GEOSRAD_DATA', dir, status=stat)
      if (stat /= 0) error stop 'geosrad: set GEOSRAD_DATA to the directory holding the *.grtb coefficient tables'
      p = trim(dir) // '/' // name // c_null_char
   end function

   ! message of the last failure on stderr, without stopping (callers that follow MAPL's RC convention)
   subroutine geosrad_warn(where)
      character(*), intent(in) :: where
      character(kind=c_char), pointer :: s(:)
      type(c_ptr) :: cp
      integer :: n
      cp = geosrad_last_error(ctx)
      n = 0
      if (c_associated(cp)) then
         call c_f_pointer(cp, s, [1024])
         do while (n < 1024)
            if (s(n+1) == c_null_char) exit
            n = n + 1
         end do
         write(error_unit,'(3a,1024a1)') ' ', where, ': ', s(1:n)
      end if
   end subroutine

   ! mirror of the reference's `error stop <message>`
   subroutine geosrad_fail(where)
      character(*), intent(in) :: where
      character(kind=c_char), pointer :: s(:)
      type(c_ptr) :: cp
      integer :: n
      cp = geosrad_last_error(ctx)
      n = 0
      if (c_associated(cp)) then
         call c_f_pointer(cp, s, [1024])
         do while (n < 1024)
            if (s(n+1) == c_null_char) exit
            n = n + 1
         end do
         write(error_unit,'(3a,1024a1)') ' ', where, ': ', s(1:n)
      end if
      error stop 'geosrad: see message above'
   end subroutine

end module geosrad_c
