! uselist_driver.F90 -- does the shim really stand in for the reference's modules?  Each subroutine below carries the `use ... only:`
! lines of one GridComp exactly as the reference has them (the ESMF / MAPL / RRTMGP / MKL lines left out), so this file only
! compiles if the shim modules export every name the unchanged GridComps import:
!    irr_uses  : GEOS_IrradGridComp.F90:67-75, 1472-1474
!    sol_uses  : GEOS_SolarGridComp.F90:175-183, 3346-3348, 6678-6679
!    rad_uses  : GEOS_RadiationGridComp.F90:481-484
! `sorad_constants` is the reference's own data module (no MAPL dependency): it stays in a GEOS build and is compiled here from the
! reference tree when that is present (-DHAVE_SORAD_CONSTANTS); `gettau::getvistau` (GEOS_SolarGridComp.F90:177) stays in a GEOS
! build as well but needs MAPL_ConstantsMod, which this image lacks, so its line is the one import not exercised here.
!
! sol_uses then calls rrtmg_sw the two ways SORADCORE does (GEOS_SolarGridComp.F90:6331-6387): with DRBAND / DFBAND disassociated
! and do_drfband false (every run without SOLAR_TO_OBIO), and with them pointing at the non-contiguous section ptr2(1:Num2do,:)
! of a larger array (:4148-4151).  Reads the batch tests/test_fortran_shim.py writes for sw_driver.F90.
module uselist_io
   implicit none
   integer :: ncol, nlay, ih, dyofyr, cloudLM, cloudMH, iaer, normFlx, isolvar
   real :: scon
   real, allocatable, dimension(:,:) :: play, plev, tlay, h2o, o3, co2, ch4, o2, cld, ciwp, clwp, rei, rel, zm
   real, allocatable, dimension(:) :: coszen, alat, asdir, asdif, aldir, aldif
   real, allocatable, dimension(:,:,:) :: tauaer, ssaaer, asmaer
   real(8), allocatable :: out(:)
   integer :: nout = 0
contains
   subroutine put(a)
      real, intent(in) :: a(:)
      out(nout+1:nout+size(a)) = real(a, 8); nout = nout + size(a)
   end subroutine
   subroutine read_batch(fin, nb)
      character(*), intent(in) :: fin
      integer, intent(in) :: nb
      integer :: u, i
      real(4) :: scon4
      open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
      read(u) ncol, nlay, ih, dyofyr, cloudLM, cloudMH, iaer, normFlx, isolvar, scon4
      scon = scon4
      allocate(play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), h2o(ncol,nlay), o3(ncol,nlay), co2(ncol,nlay), ch4(ncol,nlay), &
         o2(ncol,nlay), cld(ncol,nlay), ciwp(ncol,nlay), clwp(ncol,nlay), rei(ncol,nlay), rel(ncol,nlay), zm(ncol,nlay), &
         coszen(ncol), alat(ncol), asdir(ncol), asdif(ncol), aldir(ncol), aldif(ncol), tauaer(ncol,nlay,nb), &
         ssaaer(ncol,nlay,nb), asmaer(ncol,nlay,nb))
      call rd1(coszen); call rd2(play); call rd2(plev); call rd2(tlay); call rd2(h2o); call rd2(o3); call rd2(co2); call rd2(ch4)
      call rd2(o2); call rd2(cld); call rd2(ciwp); call rd2(clwp); call rd2(rei); call rd2(rel); call rd2(zm); call rd1(alat)
      do i = 1, nb
         call rd2(tauaer(:,:,i))
      end do
      do i = 1, nb
         call rd2(ssaaer(:,:,i))
      end do
      do i = 1, nb
         call rd2(asmaer(:,:,i))
      end do
      call rd1(asdir); call rd1(asdif); call rd1(aldir); call rd1(aldif)
      close(u)
   contains
      subroutine rd2(a)
         real, intent(out) :: a(:,:)
         real(4), allocatable :: buf(:)
         allocate(buf(size(a))); read(u) buf; a = reshape(real(buf, kind(a)), shape(a))
      end subroutine
      subroutine rd1(a)
         real, intent(out) :: a(:)
         real(4), allocatable :: buf(:)
         allocate(buf(size(a))); read(u) buf; a = real(buf, kind(a))
      end subroutine
   end subroutine
end module uselist_io

subroutine rad_uses(ih)
   use cloud_condensate_inhomogeneity, only: set_inhomogeneity
   use cloud_subcol_gen, only : initialize_cloud_subcol_gen, &
     def_aam1, def_aam2, def_aam30, def_aam4, &
     def_ram1, def_ram2, def_ram30, def_ram4
   implicit none
   integer, intent(in) :: ih
   ! RAD:Initialize (GEOS_RadiationGridComp.F90:560-590): resources default to the def_ parameters
   call initialize_cloud_subcol_gen(adl_am1=def_aam1, adl_am2=def_aam2, adl_am30=def_aam30, adl_am4=def_aam4, &
      rdl_am1=def_ram1, rdl_am2=def_ram2, rdl_am30=def_ram30, rdl_am4=def_ram4)
   if (ih /= 0) call set_inhomogeneity(ih)
end subroutine

subroutine irr_uses()
   use rrtmg_lw_rad, only: rrtmg_lw
   use rrtmg_lw_init, only: rrtmg_lw_ini
   use parrrtm, only: ngptlw, nbndlw
   use rrlw_wvn, only: wavenum1, wavenum2
   use irradmod, only: IRRAD
   use cloud_condensate_inhomogeneity, only: condensate_inhomogeneous, zcw_lookup
   use cloud_subcol_gen, only : &
     correlation_length_cloud_fraction, correlation_length_condensate
   use uselist_io
   implicit none
   real :: adl(ncol), rdl(ncol), z(3)
   ! the host calls of the RRTMGP branch (GEOS_IrradGridComp.F90:2859-2973): decorrelation lengths per column, zcw per cell
   call correlation_length_cloud_fraction(ncol, ncol, dyofyr, alat, adl)
   call correlation_length_condensate(ncol, ncol, dyofyr, alat, rdl)
   z = [zcw_lookup(0.25, 0.5), zcw_lookup(0.9, 1.0), merge(1., 0., condensate_inhomogeneous())]
   call put(adl); call put(rdl); call put(z)
   call put([real(ngptlw), real(nbndlw), wavenum1(1), wavenum2(nbndlw)])
end subroutine

subroutine sol_uses()
   use soradmod, only: SORAD
#ifdef HAVE_SORAD_CONSTANTS
   use sorad_constants, only : HK_IR_OLD, HK_UV_OLD
#endif
   use rrtmg_sw_rad, only: rrtmg_sw
   use rrtmg_sw_init, only: rrtmg_sw_ini
   use parrrsw, only: ngptsw
   use cloud_subcol_gen, only: &
      generate_stochastic_clouds, clearCounts_threeBand
   use cloud_condensate_inhomogeneity, only: condensate_inhomogeneous, zcw_lookup
   use cloud_subcol_gen, only : &
     correlation_length_cloud_fraction, correlation_length_condensate
   use parrrsw, only: nbndsw, jpb1, jpb2
   use rrsw_wvn, only: wavenum1, wavenum2
   use uselist_io
   implicit none
   real, dimension(ncol,nlay+1) :: swuflx, swdflx, swuflxc, swdflxc, u2, d2, uc2, dc2
   real, dimension(ncol) :: nirr, nirf, parr, parf, uvrr, uvrf, c1, c2, c3, c4, c5, c6, c7, c8
   real :: fswband(ncol,nbndsw)
   integer :: cc(ncol,4), rc, mapl_placeholder, npad
   real, pointer, dimension(:,:) :: DRBAND, DFBAND
   real, pointer :: ptr2(:,:), ptr3(:,:)
   mapl_placeholder = 0
   call rrtmg_sw_ini
   ! 1. default GEOS run: SOLAR_TO_OBIO false -> DRBAND / DFBAND never associated
   nullify(DRBAND, DFBAND)
   call rrtmg_sw(mapl_placeholder, 4, ncol, nlay, scon, 1.0, coszen, isolvar, play, plev, tlay, h2o, o3, co2, ch4, o2, &
      3, 1, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, &
      cloudLM, cloudMH, normFlx, cc, swuflx, swdflx, swuflxc, swdflxc, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
      c1, c2, c3, c4, c5, c6, c7, c8, .false., DRBAND, DFBAND, RC=rc)
   call put([real(rc)]); call put(reshape(swuflx, [size(swuflx)])); call put(reshape(swdflx, [size(swdflx)]))
   ! 2. SOLAR_TO_OBIO: the pointers are sections of the packed work arrays, whose first dimension is larger than Num2do
   npad = ncol + 7
   allocate(ptr2(npad,nbndsw), ptr3(npad,nbndsw))
   ptr2 = -777.; ptr3 = -777.
   DRBAND => ptr2(1:ncol,:); DFBAND => ptr3(1:ncol,:)
   call rrtmg_sw(mapl_placeholder, 4, ncol, nlay, scon, 1.0, coszen, isolvar, play, plev, tlay, h2o, o3, co2, ch4, o2, &
      3, 1, cld, ciwp, clwp, rei, rel, dyofyr, zm, alat, iaer, tauaer, ssaaer, asmaer, asdir, asdif, aldir, aldif, &
      cloudLM, cloudMH, normFlx, cc, u2, d2, uc2, dc2, nirr, nirf, parr, parf, uvrr, uvrf, fswband, &
      c1, c2, c3, c4, c5, c6, c7, c8, .true., DRBAND, DFBAND, RC=rc)
   call put([real(rc)]); call put(reshape(u2, [size(u2)])); call put(reshape(d2, [size(d2)]))
   call put(reshape(ptr2, [size(ptr2)])); call put(reshape(ptr3, [size(ptr3)]))
   call put([real(ngptsw), real(jpb1), real(jpb2), wavenum1(jpb1), wavenum2(jpb2)])
#ifdef HAVE_SORAD_CONSTANTS
   call put([HK_UV_OLD(1), HK_IR_OLD(1,1)])
#else
   call put([0., 0.])
#endif
end subroutine

program uselist_driver
   use uselist_io
   implicit none
   character(len=512) :: fin, fout
   integer :: u
   call get_command_argument(1, fin); call get_command_argument(2, fout)
   call read_batch(fin, 14)
   allocate(out(8 * (nlay + 2) * (ncol + 8) + 64 * (ncol + 8)))
   call rad_uses(ih)
   call irr_uses()
   call sol_uses()
   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) nout, out(1:nout)
   close(u)
end program uselist_driver
