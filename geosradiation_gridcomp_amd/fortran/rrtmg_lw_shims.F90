! rrtmg_lw_shims.F90 -- drop-in modules with the REFERENCE's module and procedure names, whose bodies call the
! C ABI.  GEOS_IrradGridComp.F90 `use`s exactly these names (GEOS_IrradGridComp.F90:67-75), so linking this file
! instead of the reference's rrtmg_lw sources switches the LW hot path to the MI355X without touching the driver.
!
!   parrrtm          : nbndlw, ngptlw                          (LW/modules/parrrtm.F90:33,39)
!   rrlw_wvn         : wavenum1, wavenum2                      (LW/modules/rrlw_wvn.F90)
!   rrtmg_lw_init    : rrtmg_lw_ini                            (LW/src/rrtmg_lw_init.F90:22)
!   rrtmg_lw_rad     : rrtmg_lw                                (LW/src/rrtmg_lw_rad.F90:15-23)
module parrrtm
   implicit none
   integer, parameter :: nbndlw = 16, ngptlw = 140, mxlay = 203
end module parrrtm

module rrlw_wvn
   use parrrtm, only : nbndlw
   implicit none
   real, parameter :: wavenum1(nbndlw) = [10., 350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600.]
   real, parameter :: wavenum2(nbndlw) = [350., 500., 630., 700., 820., 980., 1080., 1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600., 3250.]
   real, parameter :: delwave(nbndlw) = wavenum2 - wavenum1
end module rrlw_wvn

module rrtmg_lw_init
   use iso_c_binding
   use geosrad_c
   implicit none
   logical, save, private :: loaded = .false.
contains
   ! The reference rebuilds its tables at every refresh (GEOS_IrradGridComp.F90:3381); here the reduced tables
   ! are uploaded to HBM once and later calls are no-ops.
   subroutine rrtmg_lw_ini
      integer(c_int) :: rc
      real :: x
      if (loaded) return
      if (kind(x) == 4) then
         rc = geosrad_load_tables_lw(geosrad_ctx_handle(), geosrad_data_path('rrtmg_lw_r4.grtb'))
      else
         rc = geosrad_load_tables_lw(geosrad_ctx_handle(), geosrad_data_path('rrtmg_lw_r8.grtb'))
      end if
      if (rc /= 0) call geosrad_fail('rrtmg_lw_ini')
      loaded = .true.
   end subroutine rrtmg_lw_ini
end module rrtmg_lw_init

module rrtmg_lw_rad
   use iso_c_binding
   use geosrad_c
   implicit none
contains
   subroutine rrtmg_lw( &
      ncol, nlay, psize, dudTs, &
      play, plev, tlay, tlev, tsfc, emis, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
      cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, &
      cldf, ciwp, clwp, rei, rel, iceflglw, liqflglw, &
      tauaer, zm, alat, dyofyr, cloudLM, cloudMH, clearCounts, &
      uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, &
      band_output, olrb, dolrb_dTs)
      use parrrtm, only: nbndlw
      integer, intent(in) :: ncol, nlay, psize
      logical, intent(in) :: dudTs
      real, intent(in), target :: play(ncol,nlay), plev(ncol,0:nlay), tlay(ncol,nlay), tlev(ncol,0:nlay)
      real, intent(in), target :: tsfc(ncol), emis(ncol,nbndlw)
      real, intent(in), target, dimension(ncol,nlay) :: h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
         cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, cldf, ciwp, clwp, rei, rel
      integer, intent(in) :: iceflglw, liqflglw
      real, intent(in), target :: tauaer(ncol,nlay,nbndlw), zm(ncol,nlay), alat(ncol)
      integer, intent(in) :: dyofyr, cloudLM, cloudMH
      integer, intent(out), target :: clearCounts(ncol,4)
      real, intent(out), target, dimension(ncol,nlay+1) :: uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs
      logical, intent(in) :: band_output(nbndlw)
      real, intent(out), target :: olrb(nbndlw,ncol), dolrb_dTs(nbndlw,ncol)
      integer(c_int), target :: bo(nbndlw), cc(ncol,4)
      integer(c_int) :: rc
      bo = merge(1_c_int, 0_c_int, band_output)
      rc = geosrad_rrtmg_lw(geosrad_ctx_handle(), int(ncol,c_int), int(nlay,c_int), int(psize,c_int), &
         merge(1_c_int, 0_c_int, dudTs), c_loc(play), c_loc(plev), c_loc(tlay), c_loc(tlev), c_loc(tsfc), c_loc(emis), &
         c_loc(h2ovmr), c_loc(o3vmr), c_loc(co2vmr), c_loc(ch4vmr), c_loc(n2ovmr), c_loc(o2vmr), &
         c_loc(cfc11vmr), c_loc(cfc12vmr), c_loc(cfc22vmr), c_loc(ccl4vmr), &
         c_loc(cldf), c_loc(ciwp), c_loc(clwp), c_loc(rei), c_loc(rel), int(iceflglw,c_int), int(liqflglw,c_int), &
         c_loc(tauaer), c_loc(zm), c_loc(alat), int(dyofyr,c_int), int(cloudLM,c_int), int(cloudMH,c_int), c_loc(cc), &
         c_loc(uflx), c_loc(dflx), c_loc(uflxc), c_loc(dflxc), c_loc(duflx_dTs), c_loc(duflxc_dTs), &
         c_loc(bo), c_loc(olrb), c_loc(dolrb_dTs))
      if (rc /= 0) call geosrad_fail('rrtmg_lw')      ! the reference `error stop`s on the same conditions
      clearCounts = cc
   end subroutine rrtmg_lw
end module rrtmg_lw_rad
