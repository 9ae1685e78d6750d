! lw_driver.F90 -- a caller written exactly as GEOS_IrradGridComp's LW_Driver calls the solver
! (GEOS_IrradGridComp.F90:3381 `call RRTMG_LW_INI`, :3471 `call rrtmg_lw(IM*JM, LM, PARTITION_SIZE, ...)`), linked
! against the shim modules instead of the reference's sources.  Reads a column batch written by
! tests/test_fortran_shim.py, writes the fluxes back; the test compares them with the golden vectors.
program lw_driver
   use rrtmg_lw_init, only : rrtmg_lw_ini
   use rrtmg_lw_rad, only : rrtmg_lw
   use parrrtm, only : nbndlw
   use cloud_condensate_inhomogeneity, only : set_inhomogeneity
   implicit none
   integer :: ncol, nlay, ih, dyofyr, cloudLM, cloudMH, u, i
   real(4), allocatable :: buf(:)
   real, allocatable, dimension(:,:) :: play, plev, tlay, tlev, emis, h2o, o3, co2, ch4, n2o, o2, c11, c12, c22, ccl4, &
      cldf, ciwp, clwp, rei, rel, zm, uflx, dflx, uflxc, dflxc, du, duc, olrb, dolrb
   real, allocatable :: tsfc(:), alat(:), tauaer(:,:,:)
   integer, allocatable :: cc(:,:)
   logical :: bo(nbndlw)
   character(len=512) :: fin, fout
   character(len=32) :: frep
   integer :: nrep
   integer(8) :: t0, t1, trate
   call get_command_argument(1, fin); call get_command_argument(2, fout)
   open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
   read(u) ncol, nlay, ih, dyofyr, cloudLM, cloudMH
   allocate(play(ncol,nlay), plev(ncol,0:nlay), tlay(ncol,nlay), tlev(ncol,0:nlay), tsfc(ncol), emis(ncol,nbndlw), &
      h2o(ncol,nlay), o3(ncol,nlay), co2(ncol,nlay), ch4(ncol,nlay), n2o(ncol,nlay), o2(ncol,nlay), c11(ncol,nlay), &
      c12(ncol,nlay), c22(ncol,nlay), ccl4(ncol,nlay), cldf(ncol,nlay), ciwp(ncol,nlay), clwp(ncol,nlay), rei(ncol,nlay), &
      rel(ncol,nlay), tauaer(ncol,nlay,nbndlw), zm(ncol,nlay), alat(ncol), cc(ncol,4), uflx(ncol,nlay+1), dflx(ncol,nlay+1), &
      uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), du(ncol,nlay+1), duc(ncol,nlay+1), olrb(nbndlw,ncol), dolrb(nbndlw,ncol))
   call rd2(play); call rd2(plev); call rd2(tlay); call rd2(tlev); call rd1(tsfc); call rd2(emis)
   call rd2(h2o); call rd2(o3); call rd2(co2); call rd2(ch4); call rd2(n2o); call rd2(o2); call rd2(c11); call rd2(c12)
   call rd2(c22); call rd2(ccl4); call rd2(cldf); call rd2(ciwp); call rd2(clwp); call rd2(rei); call rd2(rel)
   do i = 1, nbndlw
      call rd2(tauaer(:,:,i))
   end do
   call rd2(zm); call rd1(alat)
   close(u)
   if (ih /= 0) call set_inhomogeneity(ih)        ! RAD:Initialize (GEOS_RadiationGridComp.F90:564-565)
   call rrtmg_lw_ini
   bo = .false.
   call rrtmg_lw(ncol, nlay, 4, .true., play, plev, tlay, tlev, tsfc, emis, h2o, o3, co2, ch4, n2o, o2, c11, c12, c22, ccl4, &
      cldf, ciwp, clwp, rei, rel, 3, 1, tauaer, zm, alat, dyofyr, cloudLM, cloudMH, cc, uflx, dflx, uflxc, dflxc, du, duc, bo, olrb, dolrb)
   ! optional third argument: repeat the call that many times and report the caller-side time of one call (host arrays in, host arrays out)
   call get_command_argument(3, frep)
   if (len_trim(frep) > 0) then
      read(frep, *) nrep
      call system_clock(t0, trate)
      do i = 1, nrep
         call rrtmg_lw(ncol, nlay, 4, .true., play, plev, tlay, tlev, tsfc, emis, h2o, o3, co2, ch4, n2o, o2, c11, c12, c22, ccl4, &
            cldf, ciwp, clwp, rei, rel, 3, 1, tauaer, zm, alat, dyofyr, cloudLM, cloudMH, cc, uflx, dflx, uflxc, dflxc, du, duc, bo, olrb, dolrb)
      end do
      call system_clock(t1)
      write(*,'(a,i0,a,f10.3)') 'rrtmg_lw from Fortran: ncol ', ncol, ' ms per call ', 1.0d3 * dble(t1 - t0) / dble(trate) / dble(nrep)
   end if
   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) real(uflx,8), real(dflx,8), real(uflxc,8), real(dflxc,8), real(du,8), real(duc,8), cc
   close(u)
contains
   ! inputs are stored as float32 (what GEOS feeds the solver) and widened when the driver is built with r8
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
end program lw_driver
