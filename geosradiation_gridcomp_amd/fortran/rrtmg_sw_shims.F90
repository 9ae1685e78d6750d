! rrtmg_sw_shims.F90 -- drop-in modules with the REFERENCE's module and procedure names for the shortwave path;
! bodies call the C ABI.  GEOS_SolarGridComp.F90 `use`s exactly these names (GEOS_SolarGridComp.F90:179-181,
! 6678-6679), so linking this file instead of the reference's rrtmg_sw sources switches the SW hot path to the
! MI355X without touching the driver.
!
!   parrrsw          : nbndsw, ngptsw, jpband, jpb1, jpb2       (SW/modules/parrrsw.F90)
!   rrsw_wvn         : wavenum1, wavenum2                       (SW/modules/rrsw_wvn.F90, set in rrtmg_sw_init.F90:187-190)
!   rrtmg_sw_init    : rrtmg_sw_ini                             (SW/src/rrtmg_sw_init.F90:23)
!   rrtmg_sw_rad     : rrtmg_sw                                 (SW/src/rrtmg_sw_rad.F90:68-124)
!
! The reference's rrtmg_sw takes the GridComp's MAPL handle only to drive timers; inside GEOS build this file with
! -DGEOSRAD_WITH_MAPL so the dummy has the reference's type, elsewhere it is an unlimited polymorphic placeholder.
module parrrsw
   implicit none
   integer, parameter :: nbndsw = 14, ngptsw = 112, jpband = 29, jpb1 = 16, jpb2 = 29, mxlay = 203
end module parrrsw

module rrsw_wvn
   use parrrsw, only : jpb1, jpb2
   implicit none
   real, parameter :: wavenum1(jpb1:jpb2) = [2600., 3250., 4000., 4650., 5150., 6150., 7700., 8050., 12850., 16000., 22650., &
                                             29000., 38000., 820.]
   real, parameter :: wavenum2(jpb1:jpb2) = [3250., 4000., 4650., 5150., 6150., 7700., 8050., 12850., 16000., 22650., 29000., &
                                             38000., 50000., 2600.]
   real, parameter :: delwave(jpb1:jpb2) = wavenum2 - wavenum1
end module rrsw_wvn

module rrtmg_sw_init
   use iso_c_binding
   use geosrad_c
   implicit none
   logical, save, private :: loaded = .false.
contains
   ! tables are uploaded to HBM once; later calls (GEOS_SolarGridComp.F90:6225 calls it every SW refresh) are no-ops
   subroutine rrtmg_sw_ini
      integer(c_int) :: rc
      real :: x
      if (loaded) return
      if (kind(x) == 4) then
         rc = geosrad_load_tables_sw(geosrad_ctx_handle(), geosrad_data_path('rrtmg_sw_r4.grtb'))
      else
         rc = geosrad_load_tables_sw(geosrad_ctx_handle(), geosrad_data_path('rrtmg_sw_r8.grtb'))
      end if
      if (rc /= 0) call geosrad_fail('rrtmg_sw_ini')
      loaded = .true.
   end subroutine rrtmg_sw_ini
end module rrtmg_sw_init

module rrtmg_sw_rad
   use iso_c_binding
   use geosrad_c
#ifdef GEOSRAD_WITH_MAPL
   use MAPL, only : MAPL_MetaComp, MAPL_TimerOn, MAPL_TimerOff
#endif
   implicit none
contains
   subroutine rrtmg_sw (MAPL, &
      rpart, ncol, nlay, &
      scon, adjes, coszen, isolvar, &
      play, plev, tlay, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, &
      iceflgsw, liqflgsw, &
      cld, ciwp, clwp, rei, rel, &
      dyofyr, zm, alat, &
      iaer, tauaer, ssaaer, asmaer, &
      asdir, asdif, aldir, aldif, &
      cloudLM, cloudMH, normFlx, &
      clearCounts, swuflx, swdflx, swuflxc, swdflxc, &
      nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
      cotdtp, cotdhp, cotdmp, cotdlp, &
      cotntp, cotnhp, cotnmp, cotnlp, &
      do_drfband, drband, dfband, &
      bndscl, indsolvar, solcycfrac, &
      RC)
      use parrrsw, only : nbndsw
#ifdef GEOSRAD_WITH_MAPL
      type(MAPL_MetaComp), pointer, intent(inout) :: MAPL
#else
      class(*), intent(inout) :: MAPL
#endif
      integer, intent(in) :: rpart, ncol, nlay
      real, intent(in) :: scon, adjes
      real, intent(in), target :: coszen(ncol)
      integer, intent(in) :: isolvar
      real, intent(in), target :: play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay)
      real, intent(in), target, dimension(ncol,nlay) :: h2ovmr, o3vmr, co2vmr, ch4vmr, o2vmr, cld, ciwp, clwp, rei, rel, zm
      integer, intent(in) :: iceflgsw, liqflgsw, dyofyr, iaer, cloudLM, cloudMH, normFlx
      real, intent(in), target :: alat(ncol)
      real, intent(in), target, dimension(ncol,nlay,nbndsw) :: tauaer, ssaaer, asmaer
      real, intent(in), target, dimension(ncol) :: asdir, asdif, aldir, aldif
      integer, intent(out), target :: clearCounts(ncol,4)
      real, intent(out), target, dimension(ncol,nlay+1) :: swuflx, swdflx, swuflxc, swdflxc
      real, intent(out), target, dimension(ncol) :: nirr, nirf, parr, parf, uvrr, uvrf
      real, intent(out), target :: fswband(ncol,nbndsw)
      real, intent(out), target, dimension(ncol) :: cotdtp, cotdhp, cotdmp, cotdlp, cotntp, cotnhp, cotnmp, cotnlp
      logical, intent(in) :: do_drfband
      ! exactly the reference's dummies (rrtmg_sw_rad.F90:355): GEOS passes DRBAND / DFBAND disassociated unless SOLAR_TO_OBIO is
      ! set, and then as the non-contiguous section ptr2(1:Num2do,:) (GEOS_SolarGridComp.F90:778,4148-4151,6385)
      real, intent(inout), dimension(:,:), pointer :: drband, dfband
      real, intent(in), optional, target :: bndscl(nbndsw), indsolvar(2)
      real, intent(in), optional, target :: solcycfrac
      integer, intent(out), optional :: RC
      integer(c_int), target :: cc(ncol,4)
      integer(c_int) :: st
      type(c_ptr) :: pb, pi, pdr, pdf, pf
      real, allocatable, target :: zdr(:,:), zdf(:,:)       ! contiguous (ncol,nbndsw) images of the pointer targets
      pb = c_null_ptr; pi = c_null_ptr; pdr = c_null_ptr; pdf = c_null_ptr; pf = c_null_ptr
      if (do_drfband) then                                   ! the pointers are touched only in this case, like the reference
         allocate(zdr(ncol,nbndsw), zdf(ncol,nbndsw))
         pdr = c_loc(zdr); pdf = c_loc(zdf)
      end if
      if (present(bndscl)) pb = c_loc(bndscl)
      if (present(indsolvar)) pi = c_loc(indsolvar)
      if (present(solcycfrac)) pf = c_loc(solcycfrac)        ! isolvar = 1 (rrtmg_sw_rad.F90:906-930)
      ! The reference's timers (rrtmg_sw_rad.F90:1181-1200 registers ---RRTMG_PART, _CLDSGEN, _CLDPRMC, _SETCOEF, _TAUMOL, _REFTRA, _VRTQDR):
      ! the whole call is one asynchronous pipeline here, so on the host it is charged to ---RRTMG_PART (the timer that brackets the
      ! reference's partition loop); the stages carry the reference's names as roctx ranges on the GPU timeline (GEOSRAD_ROCTX=1,
      ! rocprofv3 --marker-trace)
#ifdef GEOSRAD_WITH_MAPL
      call MAPL_TimerOn(MAPL, "---RRTMG_PART")
#endif
      st = geosrad_rrtmg_sw(geosrad_ctx_handle(), int(rpart,c_int), int(ncol,c_int), int(nlay,c_int), real(scon,c_double), &
         real(adjes,c_double), c_loc(coszen), int(isolvar,c_int), c_loc(play), c_loc(plev), c_loc(tlay), &
         c_loc(h2ovmr), c_loc(o3vmr), c_loc(co2vmr), c_loc(ch4vmr), c_loc(o2vmr), int(iceflgsw,c_int), int(liqflgsw,c_int), &
         c_loc(cld), c_loc(ciwp), c_loc(clwp), c_loc(rei), c_loc(rel), int(dyofyr,c_int), c_loc(zm), c_loc(alat), &
         int(iaer,c_int), c_loc(tauaer), c_loc(ssaaer), c_loc(asmaer), c_loc(asdir), c_loc(asdif), c_loc(aldir), c_loc(aldif), &
         int(cloudLM,c_int), int(cloudMH,c_int), int(normFlx,c_int), c_loc(cc), &
         c_loc(swuflx), c_loc(swdflx), c_loc(swuflxc), c_loc(swdflxc), &
         c_loc(nirr), c_loc(nirf), c_loc(parr), c_loc(parf), c_loc(uvrr), c_loc(uvrf), c_loc(fswband), &
         c_loc(cotdtp), c_loc(cotdhp), c_loc(cotdmp), c_loc(cotdlp), c_loc(cotntp), c_loc(cotnhp), c_loc(cotnmp), c_loc(cotnlp), &
         merge(1_c_int, 0_c_int, do_drfband), pdr, pdf, pb, pi, pf)
#ifdef GEOSRAD_WITH_MAPL
      call MAPL_TimerOff(MAPL, "---RRTMG_PART")
#endif
      clearCounts = cc
      if (do_drfband .and. st == 0) then
         drband(1:ncol,1:nbndsw) = zdr; dfband(1:ncol,1:nbndsw) = zdf
      end if
      ! the reference reports failures through MAPL's RC convention (_FAIL / _RETURN(_SUCCESS), rrtmg_sw_rad.F90:365-383)
      if (present(RC)) then
         RC = st
         if (st /= 0) call geosrad_warn('rrtmg_sw')
      else if (st /= 0) then
         call geosrad_fail('rrtmg_sw')
      end if
   end subroutine rrtmg_sw
end module rrtmg_sw_rad
