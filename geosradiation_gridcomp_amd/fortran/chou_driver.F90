! chou_driver.F90 -- a caller written as GEOS_IrradGridComp / GEOS_SolarGridComp call the Chou-Suarez schemes
! (GEOS_IrradGridComp.F90:2093 `call IRRAD(IM*JM, LM, PLE, T, Q, O3, T2M, CO2, TRACE, ...)`; GEOS_SolarGridComp `call SORAD(...)`),
! linked against the shim modules.  Reads a batch written by tests/test_fortran_shim.py, writes the fluxes back.
program chou_driver
   use irradmod, only : irrad
   use soradmod, only : sorad
   implicit none
   integer :: m, np, ict, icb, na, u, l
   real(4) :: co2_4
   real(4), allocatable :: buf(:)
   real, allocatable, dimension(:,:) :: ple, ta, wa, oa, n2o, ch4, c11, c12, c22, fcld, fs, tg, tv, pl, flxu, flcu, flau, flxau, flxd, flcd, &
      flad, flxad, dfdts, flx, flc, sflxu, sflcu, sfcband
   real, allocatable, dimension(:,:,:) :: cwc, reff, eg, ev, rv, taua, ssaa, asya, taudiag, staua, sssaa, sasya
   real, allocatable, dimension(:) :: tb, sfcem, cosz, rsuvbm, rsuvdf, rsirbm, rsirdf, f1, f2, f3, f4, f5, f6
   real :: hk_uv(5), hk_ir(3,10)
   real, pointer :: drband(:,:), dfband(:,:)
   character(len=512) :: fin, fout
   call get_command_argument(1, fin); call get_command_argument(2, fout)
   open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
   read(u) m, np, ict, icb, na, co2_4
   allocate(ple(m,np+1), ta(m,np), wa(m,np), oa(m,np), tb(m), n2o(m,np), ch4(m,np), c11(m,np), c12(m,np), c22(m,np), cwc(m,np,4), &
      fcld(m,np), reff(m,np,4), fs(m,1), tg(m,1), eg(m,1,10), tv(m,1), ev(m,1,10), rv(m,1,10), taua(m,np,10), ssaa(m,np,10), asya(m,np,10), &
      flxu(m,np+1), flcu(m,np+1), flau(m,np+1), flxau(m,np+1), flxd(m,np+1), flcd(m,np+1), flad(m,np+1), flxad(m,np+1), dfdts(m,np+1), &
      sfcem(m), taudiag(m,np,10), cosz(m), pl(m,np+1), staua(m,np,8), sssaa(m,np,8), sasya(m,np,8), rsuvbm(m), rsuvdf(m), rsirbm(m), &
      rsirdf(m), flx(m,np+1), flc(m,np+1), sflxu(m,np+1), sflcu(m,np+1), sfcband(m,8), f1(m), f2(m), f3(m), f4(m), f5(m), f6(m), &
      drband(m,8), dfband(m,8))
   call rd2(ple); call rd2(ta); call rd2(wa); call rd2(oa); call rd1(tb); call rd2(n2o); call rd2(ch4); call rd2(c11); call rd2(c12)
   call rd2(c22); call rd3(cwc); call rd2(fcld); call rd3(reff); call rd2(fs); call rd2(tg); call rd3(eg); call rd2(tv); call rd3(ev)
   call rd3(rv); call rd3(taua); call rd3(ssaa); call rd3(asya)
   call rd1(cosz); call rd2(pl); call rd3(staua); call rd3(sssaa); call rd3(sasya); call rd1(rsuvbm); call rd1(rsuvdf); call rd1(rsirbm)
   call rd1(rsirdf)
   allocate(buf(35)); read(u) buf; hk_uv = real(buf(1:5), kind(hk_uv)); hk_ir = reshape(real(buf(6:35), kind(hk_ir)), [3,10])
   close(u)
   call irrad(m, np, ple, ta, wa, oa, tb, real(co2_4), .true., n2o, ch4, c11, c12, c22, cwc, fcld, ict, icb, reff, 1, fs, tg, eg, tv, ev, rv, &
      na, 10, taua, ssaa, asya, flxu, flcu, flau, flxau, flxd, flcd, flad, flxad, dfdts, sfcem, taudiag)
   call sorad(m, np, 8, cosz, pl, ta, wa, oa, real(co2_4), cwc, fcld, ict, icb, reff, hk_uv, hk_ir, staua, sssaa, sasya, rsuvbm, rsuvdf, &
      rsirbm, rsirdf, flx, flc, f1, f2, f3, f4, f5, f6, sflxu, sflcu, sfcband, .true., drband, dfband)
   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) real(flxu,8), real(flxd,8), real(flcu,8), real(dfdts,8), real(sfcem,8), real(flx,8), real(flc,8), real(sflxu,8), real(f3,8), &
      real(sfcband,8), real(drband,8)
   close(u)
contains
   subroutine rd3(a)
      real, intent(out) :: a(:,:,:)
      if (allocated(buf)) deallocate(buf)
      allocate(buf(size(a))); read(u) buf; a = reshape(real(buf, kind(a)), shape(a))
   end subroutine
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
end program chou_driver
