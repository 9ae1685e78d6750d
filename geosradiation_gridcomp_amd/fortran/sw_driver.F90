! sw_driver.F90 -- a caller written as GEOS_SolarGridComp's SORADCORE calls the solver (GEOS_SolarGridComp.F90:6225
! `call RRTMG_SW_INI`, :6331 `call RRTMG_SW (MAPL, RPART, NCOL, LM, SC, ADJES, ZT, ISOLVAR, ...)`), linked against the
! shim modules instead of the reference's sources.  Reads a column batch written by tests/test_fortran_shim.py and
! writes the fluxes back.
program sw_driver
   use rrtmg_sw_init, only : rrtmg_sw_ini
   use rrtmg_sw_rad, only : rrtmg_sw
   use parrrsw, only : nbndsw
   use cloud_condensate_inhomogeneity, only : set_inhomogeneity
   implicit none
   integer :: ncol, nlay, ih, dyofyr, cloudLM, cloudMH, iaer, normFlx, isolvar, u, i, rc, mapl_placeholder
   real(4) :: scon4
   real(4), allocatable :: buf(:)
   real, allocatable, dimension(:,:) :: play, plev, tlay, h2o, o3, co2, ch4, o2, cld, ciwp, clwp, rei, rel, zm, &
      swuflx, swdflx, swuflxc, swdflxc, fswband
   real, pointer, dimension(:,:) :: drband, dfband            ! pointers, as in SORADCORE (GEOS_SolarGridComp.F90:6385)
   real, allocatable, dimension(:) :: coszen, alat, asdir, asdif, aldir, aldif, nirr, nirf, parr, parf, uvrr, uvrf, &
      c1, c2, c3, c4, c5, c6, c7, c8
   real, allocatable, dimension(:,:,:) :: tauaer, ssaaer, asmaer
   integer, allocatable :: cc(:,:)
   character(len=512) :: fin, fout
   character(len=32) :: frep, ffrac
   integer :: nrep
   real :: solcycfrac, indsolvar(2)
   integer(8) :: t0, t1, trate
   call get_command_argument(1, fin); call get_command_argument(2, fout)
   open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
   read(u) ncol, nlay, ih, dyofyr, cloudLM, cloudMH, iaer, normFlx, isolvar, scon4
   allocate(play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), h2o(ncol,nlay), o3(ncol,nlay), co2(ncol,nlay), ch4(ncol,nlay), &
      o2(ncol,nlay), cld(ncol,nlay), ciwp(ncol,nlay), clwp(ncol,nlay), rei(ncol,nlay), rel(ncol,nlay), zm(ncol,nlay), &
      coszen(ncol), alat(ncol), asdir(ncol), asdif(ncol), aldir(ncol), aldif(ncol), tauaer(ncol,nlay,nbndsw), &
      ssaaer(ncol,nlay,nbndsw), asmaer(ncol,nlay,nbndsw), cc(ncol,4), swuflx(ncol,nlay+1), swdflx(ncol,nlay+1), &
      swuflxc(ncol,nlay+1), swdflxc(ncol,nlay+1), nirr(ncol), nirf(ncol), parr(ncol), parf(ncol), uvrr(ncol), uvrf(ncol), &
      fswband(ncol,nbndsw), drband(ncol,nbndsw), dfband(ncol,nbndsw), c1(ncol), c2(ncol), c3(ncol), c4(ncol), c5(ncol), &
      c6(ncol), c7(ncol), c8(ncol))
   call rd1(coszen); call rd2(play); call rd2(plev); call rd2(tlay); call rd2(h2o); call rd2(o3); call rd2(co2); call rd2(ch4)
   call rd2(o2); call rd2(cld); call rd2(ciwp); call rd2(clwp); call rd2(rei); call rd2(rel); call rd2(zm); call rd1(alat)
   do i = 1, nbndsw
      call rd2(tauaer(:,:,i))
   end do
   do i = 1, nbndsw
      call rd2(ssaaer(:,:,i))
   end do
   do i = 1, nbndsw
      call rd2(asmaer(:,:,i))
   end do
   call rd1(asdir); call rd1(asdif); call rd1(aldir); call rd1(aldif)
   close(u)
   if (ih /= 0) call set_inhomogeneity(ih)        ! RAD:Initialize (GEOS_RadiationGridComp.F90:564-565)
   call rrtmg_sw_ini
   mapl_placeholder = 0
   ! optional fourth argument: the position in the mean solar cycle for isolvar = 1 (the optional arguments INDSOLVAR, SOLCYCFRAC of
   ! rrtmg_sw_rad.F90:122-124 by keyword)
   call get_command_argument(4, ffrac)
   if (len_trim(ffrac) > 0) then
      read(ffrac, *) solcycfrac
      indsolvar = (/ 1.15, 0.9 /)
      call rrtmg_sw(mapl_placeholder, 4, ncol, nlay, real(scon4), 1.0, coszen, isolvar, play, plev, tlay, h2o, o3, co2, ch4, o2, &
         3, 1, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, &
         cloudLM, cloudMH, normFlx, cc, swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
         c1, c2, c3, c4, c5, c6, c7, c8, .true., drband, dfband, INDSOLVAR=indsolvar, SOLCYCFRAC=solcycfrac, RC=rc)
   else
   call rrtmg_sw(mapl_placeholder, 4, ncol, nlay, real(scon4), 1.0, coszen, isolvar, play, plev, tlay, h2o, o3, co2, ch4, o2, &
      3, 1, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, &
      cloudLM, cloudMH, normFlx, cc, swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
      c1, c2, c3, c4, c5, c6, c7, c8, .true., drband, dfband, RC=rc)
   end if
   ! optional third argument: repeat the call that many times and report the caller-side time of one call
   call get_command_argument(3, frep)
   if (len_trim(frep) > 0) then
      read(frep, *) nrep
      call system_clock(t0, trate)
      do i = 1, nrep
      call rrtmg_sw(mapl_placeholder, 4, ncol, nlay, real(scon4), 1.0, coszen, isolvar, play, plev, tlay, h2o, o3, co2, ch4, o2, &
      3, 1, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, &
      cloudLM, cloudMH, normFlx, cc, swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
      c1, c2, c3, c4, c5, c6, c7, c8, .true., drband, dfband, RC=rc)
      end do
      call system_clock(t1)
      write(*,'(a,i0,a,f10.3)') 'rrtmg_sw from Fortran: ncol ', ncol, ' ms per call ', 1.0d3 * dble(t1 - t0) / dble(trate) / dble(nrep)
   end if
   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) rc, real(swuflx,8), real(swdflx,8), real(swuflxc,8), real(swdflxc,8), real(nirr,8), real(parf,8), real(fswband,8), &
      real(drband,8), real(dfband,8), real(c1,8), cc
   close(u)
contains
   subroutine rd2(a)
      real, intent(out) :: a(:,:)
      if (allocated(buf)) deallocate(buf)
      allocate(buf(size(a))); read(u) buf; a = reshape(real(buf, kind(a)), shape(a))
   end subroutine
   subroutine rd1(a)
      real, intent(out) :: a(:)
      if (allocated(buf)) deallocate(buf)
      allocate(buf(size(a))); read(u) buf; a = real(buf, kind(a))
   end subroutine
end program sw_driver
