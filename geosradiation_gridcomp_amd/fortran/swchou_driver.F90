! swchou_driver.F90 -- the Chou-Suarez branch of SORADCORE as GEOS_SolarGridComp would run it with its packed daytime fields on the device
! (GEOS_SolarGridComp.F90:4484-4572 -> SHRTWAVE :6597-6672): GEOS-native fields in (model ordering, Pa, odd oxygen, radii in metres with
! MAPL_UNDEF cells), `call sw_driver_chou`, the INTERNAL fluxes out.  Reads a batch written by tests/test_fortran_shim.py (fields in SWC_*
! order), writes FSW, FSWU, NIRR, FSWBAND, DRBAND.
program swchou_driver
   use iso_c_binding
   use geosrad_gridcomp
   implicit none
   integer :: ncol, lm, lcldmh, lcldlm, u, k, rc, n3, n3p
   integer :: sz(SWC_NIN)
   real(8) :: consts(SWC_NCONST)
   real(4), allocatable :: buf(:)
   real(4) :: hk4(35)
   real :: hk_uv(5), hk_ir(3,10)
   real, allocatable :: a(:), fsw(:), fswu(:), nirr(:), fswband(:), drband(:)
   type(c_ptr) :: fin(SWC_NIN), fout(SWC_NOUT)
   character(len=512) :: fi, fo
   call get_command_argument(1, fi); call get_command_argument(2, fo)
   open(newunit=u, file=trim(fi), access='stream', form='unformatted', status='old')
   read(u) ncol, lm, lcldmh, lcldlm
   read(u) consts
   n3 = ncol * lm; n3p = ncol * (lm + 1)
   sz = n3
   sz(SWC_PLE) = n3p
   sz([SWC_TAUA, SWC_SSAA, SWC_ASYA]) = n3 * 8
   sz([SWC_ZT, SWC_ALBVR, SWC_ALBVF, SWC_ALBNR, SWC_ALBNF]) = ncol
   do k = 1, SWC_NIN
      allocate(buf(sz(k)), a(sz(k))); read(u) buf; a = real(buf, kind(a))
      fin(k) = dev_alloc(sz(k)); call dev_put(fin(k), a, sz(k))
      deallocate(buf, a)
   end do
   read(u) hk4
   close(u)
   hk_uv = real(hk4(1:5), kind(hk_uv)); hk_ir = reshape(real(hk4(6:35), kind(hk_ir)), [3,10])
   do k = SWC_FSW, SWC_FSCU
      fout(k) = dev_alloc(n3p)
   end do
   do k = SWC_NIRR, SWC_UVRF
      fout(k) = dev_alloc(ncol)
   end do
   do k = SWC_FSWBAND, SWC_DFBAND
      fout(k) = dev_alloc(ncol * 8)
   end do
   call sw_driver_chou(ncol, lm, fin, consts, lcldmh, lcldlm, hk_uv, hk_ir, .true., fout, rc)
   if (rc /= 0) error stop 'sw_driver_chou failed'
   call dev_sync()
   allocate(fsw(n3p), fswu(n3p), nirr(ncol), fswband(ncol * 8), drband(ncol * 8))
   call dev_get(fsw, fout(SWC_FSW), n3p); call dev_get(fswu, fout(SWC_FSWU), n3p); call dev_get(nirr, fout(SWC_NIRR), ncol)
   call dev_get(fswband, fout(SWC_FSWBAND), ncol * 8); call dev_get(drband, fout(SWC_DRBAND), ncol * 8)
   open(newunit=u, file=trim(fo), access='stream', form='unformatted', status='replace')
   write(u) real(fsw,8), real(fswu,8), real(nirr,8), real(fswband,8), real(drband,8)
   close(u)
   do k = 1, SWC_NIN
      call dev_free(fin(k))
   end do
   do k = 1, SWC_NOUT
      call dev_free(fout(k))
   end do
end program swchou_driver
