! gridcomp_driver.F90 -- the RRTMG branch of LW_Driver followed by one heartbeat Update_Flx, as GEOS_IrradGridComp would run them with
! its fields on the device: GEOS-native fields (model ordering, SI units) in, INTERNAL state and exports out.
! Reads a batch written by tests/test_fortran_shim.py (fields in LWD_* order), writes FLX_INT, DFDTS, SFCEM_INT, CLDTTLW and the
! exports FLX, OLR, FLNS, SFCEM of a step with TSINST = TS + 1 K, then the RATS internals FLX_RAT, SFCEM_RAT of
! RATS_DIAGNOSTICS: CO2 H2O (GEOS_IrradGridComp.F90:3389-3469) from the same driver call.
program gridcomp_driver
   use iso_c_binding
   use rrtmg_lw_init, only : rrtmg_lw_ini
   use cloud_condensate_inhomogeneity, only : set_inhomogeneity
   use geosrad_gridcomp
   implicit none
   integer :: ncol, lm, nb, doy, lcldlm, lcldmh, ih, u, k, n3, n3p
   integer :: sz(LWD_NIN)
   real(8) :: consts(LWD_NCONST)
   real(4), allocatable :: buf(:)
   real, allocatable :: a(:), ts(:), flx_int(:), dfdts(:), sfcem_int(:), cldtt(:), flx(:), olr(:), flns(:), sfcem(:)
   type(c_ptr) :: fin(LWD_NIN), fout(LWD_NOUT), uin(LWU_NIN), uout(LWU_NOUT), d_tsinst, rout(LWD_NRATOUT)
   integer, parameter :: nrats = 2
   character(len=6) :: nameRATS(nrats) = ['CO2   ', 'H2O   ']
   real, allocatable :: flx_rat(:), sfcem_rat(:)
   logical :: bo(16)
   character(len=512) :: fi, fo
   call get_command_argument(1, fi); call get_command_argument(2, fo)
   open(newunit=u, file=trim(fi), access='stream', form='unformatted', status='old')
   read(u) ncol, lm, nb, ih, doy, lcldlm, lcldmh
   read(u) consts
   n3 = ncol * lm; n3p = ncol * (lm + 1)
   sz = n3
   sz(LWD_PLE) = n3p; sz(LWD_TAUA) = n3 * nb; sz(LWD_SSAA) = n3 * nb
   sz([LWD_TS, LWD_EMIS, LWD_LATS, LWD_T2M]) = ncol
   sz(LWD_CO2_3D) = 0
   fin = c_null_ptr
   allocate(ts(ncol))
   do k = 1, LWD_NIN
      if (sz(k) == 0) cycle
      allocate(buf(sz(k)), a(sz(k))); read(u) buf; a = real(buf, kind(a))
      fin(k) = dev_alloc(sz(k)); call dev_put(fin(k), a, sz(k))
      if (k == LWD_TS) ts = a
      deallocate(buf, a)
   end do
   close(u)
   fout = c_null_ptr
   do k = LWD_FLXU_INT, LWD_FLC_INT
      fout(k) = dev_alloc(n3p)
   end do
   do k = LWD_SFCEM_INT, LWD_CLDLOLW
      fout(k) = dev_alloc(ncol)
   end do
   call set_inhomogeneity(ih)
   call rrtmg_lw_ini
   bo = .false.
   rout = c_null_ptr                     ! FLXU_RAT, FLXD_RAT, DFDTS_RAT: "not associated"
   rout(LWD_FLX_RAT) = dev_alloc(n3p * nrats); rout(LWD_SFCEM_RAT) = dev_alloc(ncol * nrats)
   call lw_driver_rrtmg_rats(ncol, lm, nb, fin, consts, 3, 1, doy, lcldlm, lcldmh, bo, fout, nrats, nameRATS, rout)
   ! heartbeat: the surface has warmed by 1 K since the full calculation
   d_tsinst = dev_alloc(ncol)
   ts = ts + 1.0
   call dev_put(d_tsinst, ts, ncol)
   uin = c_null_ptr; uout = c_null_ptr
   uin(LWU_TSINST) = d_tsinst; uin(LWU_TS_INT) = fout(LWD_TS_INT); uin(LWU_SFCEM_INT) = fout(LWD_SFCEM_INT); uin(LWU_FCLD) = fin(LWD_FCLD)
   uin(LWU_FLX_INT) = fout(LWD_FLX_INT); uin(LWU_FLC_INT) = fout(LWD_FLC_INT); uin(LWU_FLXU_INT) = fout(LWD_FLXU_INT)
   uin(LWU_FLCU_INT) = fout(LWD_FLCU_INT); uin(LWU_FLXD_INT) = fout(LWD_FLXD_INT); uin(LWU_FLCD_INT) = fout(LWD_FLCD_INT)
   uin(LWU_DFDTS) = fout(LWD_DFDTS); uin(LWU_DFDTSC) = fout(LWD_DFDTSC)
   uout(LWU_FLX) = dev_alloc(n3p); uout(LWU_OLR) = dev_alloc(ncol); uout(LWU_FLNS) = dev_alloc(ncol); uout(LWU_SFCEM) = dev_alloc(ncol)
   call lw_update_flx(ncol, lm, .true., lcldmh, lcldlm, 1.0e15, uin, uout)
   call dev_sync()
   allocate(flx_int(n3p), dfdts(n3p), sfcem_int(ncol), cldtt(ncol), flx(n3p), olr(ncol), flns(ncol), sfcem(ncol))
   call dev_get(flx_int, fout(LWD_FLX_INT), n3p); call dev_get(dfdts, fout(LWD_DFDTS), n3p)
   call dev_get(sfcem_int, fout(LWD_SFCEM_INT), ncol); call dev_get(cldtt, fout(LWD_CLDTTLW), ncol)
   call dev_get(flx, uout(LWU_FLX), n3p); call dev_get(olr, uout(LWU_OLR), ncol); call dev_get(flns, uout(LWU_FLNS), ncol)
   call dev_get(sfcem, uout(LWU_SFCEM), ncol)
   open(newunit=u, file=trim(fo), access='stream', form='unformatted', status='replace')
   write(u) real(flx_int,8), real(dfdts,8), real(sfcem_int,8), real(cldtt,8), real(flx,8), real(olr,8), real(flns,8), real(sfcem,8)
   allocate(flx_rat(n3p * nrats), sfcem_rat(ncol * nrats))
   call dev_get(flx_rat, rout(LWD_FLX_RAT), n3p * nrats); call dev_get(sfcem_rat, rout(LWD_SFCEM_RAT), ncol * nrats)
   write(u) real(flx_rat,8), real(sfcem_rat,8)
   close(u)
   do k = 1, LWD_NIN
      call dev_free(fin(k))
   end do
   do k = 1, LWD_NOUT
      call dev_free(fout(k))
   end do
   do k = 1, LWD_NRATOUT
      call dev_free(rout(k))
   end do
end program gridcomp_driver
