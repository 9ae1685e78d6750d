! oracle/ref_glue.F90 -- TEST INFRASTRUCTURE (our own code; not part of the product, not a stand-in
! for any missing library).  C-callable wrappers around the reference's *unmodified* Fortran
! entry points, compiled together with the reference sources by oracle/Makefile into
! oracle/_ref/libref_{r4,r8}.so.  `real` below is the default kind, so the same file yields the fp32
! (r4) and the -fdefault-real-8 (r8) glue.
!
! Wrapped reference entry points (file:line in /root/reference):
!   rrtmg_lw_ini            GEOSirrad_GridComp/RRTMG/rrtmg_lw/gcm_model/src/rrtmg_lw_init.F90:22
!   rrtmg_lw                .../src/rrtmg_lw_rad.F90:15
!   setcoef / taumol        .../src/rrtmg_lw_setcoef.F90:52, .../src/rrtmg_lw_taumol.F90:155
!   generate_stochastic_clouds, clearCounts_threeBand, initialize_cloud_subcol_gen
!                           GEOS_RadiationShared/cloud_subcol_gen.F90:132,611,109
!   set_inhomogeneity, unset_inhomogeneity, zcw_lookup
!                           GEOS_RadiationShared/cloud_condensate_inhomogeneity.F90:45,75,86
!   rrtmg_sw_ini, setcoef_sw, taumol_sw, cldprmc_sw   (SW parts free of ESMF/MAPL)
!
! Table dump format ("GRTB" blob, little-endian, Fortran column-major data):
!   magic 'GRTB' | int32 version=1 | int32 realbytes (4|8)
!   repeated: char name[32] | int32 kind (4|8 = real bytes, -4 = int32) | int32 ndim | int32 dims[4]
!             | data | zero pad to 8-byte multiple
!   terminated by an entry named 'END' with ndim=0.

module ref_glue_io
   use iso_c_binding
   implicit none
   integer :: u = 0
   integer(8) :: pos = 0
   interface put
      module procedure put_r0, put_r1, put_r2, put_r3, put_r4d, put_i0, put_i1
   end interface
contains
   subroutine hdr(name, kind, ndim, dims)
      character(*), intent(in) :: name
      integer, intent(in) :: kind, ndim, dims(4)
      character(32) :: nm
      integer :: i
      nm = name
      do i = len_trim(name)+1, 32
         nm(i:i) = char(0)
      end do
      write(u) nm, int(kind,4), int(ndim,4), int(dims,4)
   end subroutine
   subroutine pad(nbytes)
      integer(8), intent(in) :: nbytes
      integer :: i, r
      r = int(mod(8 - mod(nbytes,8_8), 8_8))
      do i = 1, r
         write(u) int(0,1)
      end do
   end subroutine
   subroutine put_r0(name, x)
      character(*), intent(in) :: name
      real, intent(in) :: x
      call hdr(name, kind(x), 0, [1,1,1,1]); write(u) x; call pad(int(kind(x),8))
   end subroutine
   subroutine put_r1(name, x)
      character(*), intent(in) :: name
      real, intent(in) :: x(:)
      call hdr(name, kind(x), 1, [size(x,1),1,1,1]); write(u) x; call pad(int(kind(x),8)*size(x))
   end subroutine
   subroutine put_r2(name, x)
      character(*), intent(in) :: name
      real, intent(in) :: x(:,:)
      call hdr(name, kind(x), 2, [size(x,1),size(x,2),1,1]); write(u) x; call pad(int(kind(x),8)*size(x))
   end subroutine
   subroutine put_r3(name, x)
      character(*), intent(in) :: name
      real, intent(in) :: x(:,:,:)
      call hdr(name, kind(x), 3, [size(x,1),size(x,2),size(x,3),1]); write(u) x; call pad(int(kind(x),8)*size(x))
   end subroutine
   subroutine put_r4d(name, x)
      character(*), intent(in) :: name
      real, intent(in) :: x(:,:,:,:)
      call hdr(name, kind(x), 4, [size(x,1),size(x,2),size(x,3),size(x,4)]); write(u) x
      call pad(int(kind(x),8)*size(x))
   end subroutine
   subroutine put_i0(name, x)
      character(*), intent(in) :: name
      integer, intent(in) :: x
      call hdr(name, -4, 0, [1,1,1,1]); write(u) int(x,4); call pad(4_8)
   end subroutine
   subroutine put_i1(name, x)
      character(*), intent(in) :: name
      integer, intent(in) :: x(:)
      call hdr(name, -4, 1, [size(x,1),1,1,1]); write(u) int(x,4); call pad(4_8*size(x))
   end subroutine
   subroutine open_blob(cpath, n)
      character(kind=c_char), intent(in) :: cpath(*)
      integer, intent(in) :: n
      character(len=n) :: path
      integer :: i
      real :: x
      do i = 1, n
         path(i:i) = cpath(i)
      end do
      open(newunit=u, file=path, access='stream', form='unformatted', status='replace')
      write(u) 'GRTB', int(1,4), int(kind(x),4)
   end subroutine
   subroutine close_blob
      call hdr('END', 0, 0, [0,0,0,0])
      close(u)
   end subroutine
end module ref_glue_io


! ---------------------------------------------------------------------------------------------
! RRTMG_LW
! ---------------------------------------------------------------------------------------------

subroutine ref_real_bytes(n) bind(C, name='ref_real_bytes')
   use iso_c_binding
   integer(c_int), intent(out) :: n
   real :: x
   n = kind(x)
end subroutine

subroutine ref_lw_ini() bind(C, name='ref_lw_ini')
   use rrtmg_lw_init, only: rrtmg_lw_ini
   call rrtmg_lw_ini
end subroutine

subroutine ref_lw_dump_tables(cpath, n) bind(C, name='ref_lw_dump_tables')
   use iso_c_binding
   use ref_glue_io
   implicit none
   character(kind=c_char), intent(in) :: cpath(*)
   integer(c_int), value :: n
   call open_blob(cpath, n)
   call dump_common
   call dump01; call dump02; call dump03; call dump04; call dump05; call dump06; call dump07; call dump08
   call dump09; call dump10; call dump11; call dump12; call dump13; call dump14; call dump15; call dump16
   call close_blob
contains
   subroutine dump_common
      use parrrtm, only: nbndlw, ngptlw
      use rrlw_tbl, only: tau_tbl, exp_tbl, tfn_tbl, bpade, tblint
      use rrlw_wvn, only: ng, nspa, nspb, ngb, ngs, ngc, delwave, wavenum1, wavenum2, &
                          totplnk, totplk16, totplnkderiv, totplk16deriv
      use rrlw_ref, only: pref, preflog, tref, chi_mls
      use rrlw_con, only: fluxfac, oneminus, grav, avogad, pi
      use rrlw_cld, only: absice0, absice1, absice2, absice3, absice4, absliq1, ice1b
      call put('nbndlw', nbndlw); call put('ngptlw', ngptlw)
      call put('ng', ng); call put('nspa', nspa); call put('nspb', nspb)
      call put('ngb', ngb); call put('ngs', ngs); call put('ngc', ngc)
      call put('delwave', delwave); call put('wavenum1', wavenum1); call put('wavenum2', wavenum2)
      call put('tau_tbl', tau_tbl); call put('exp_tbl', exp_tbl); call put('tfn_tbl', tfn_tbl)
      call put('bpade', bpade); call put('tblint', tblint)
      call put('totplnk', totplnk); call put('totplk16', totplk16)
      call put('totplnkderiv', totplnkderiv); call put('totplk16deriv', totplk16deriv)
      call put('pref', pref); call put('preflog', preflog); call put('tref', tref)
      call put('chi_mls', chi_mls)
      call put('fluxfac', fluxfac); call put('oneminus', oneminus)
      call put('grav', grav); call put('avogad', avogad); call put('pi', pi)
      call put('absice0', absice0); call put('absice1', absice1); call put('absice2', absice2)
      call put('absice3', absice3); call put('absice4', absice4); call put('absliq1', absliq1)
      call put('ice1b', ice1b)
   end subroutine
   subroutine dump01
      use rrlw_kg01
      call put('b01_absa', absa); call put('b01_absb', absb)
      call put('b01_fracrefa', fracrefa); call put('b01_fracrefb', fracrefb)
      call put('b01_selfref', selfref); call put('b01_forref', forref)
      call put('b01_ka_mn2', ka_mn2); call put('b01_kb_mn2', kb_mn2)
   end subroutine
   subroutine dump02
      use rrlw_kg02
      call put('b02_absa', absa); call put('b02_absb', absb)
      call put('b02_fracrefa', fracrefa); call put('b02_fracrefb', fracrefb)
      call put('b02_selfref', selfref); call put('b02_forref', forref)
   end subroutine
   subroutine dump03
      use rrlw_kg03
      call put('b03_absa', absa); call put('b03_absb', absb)
      call put('b03_fracrefa', fracrefa); call put('b03_fracrefb', fracrefb)
      call put('b03_selfref', selfref); call put('b03_forref', forref)
      call put('b03_ka_mn2o', ka_mn2o); call put('b03_kb_mn2o', kb_mn2o)
   end subroutine
   subroutine dump04
      use rrlw_kg04
      call put('b04_absa', absa); call put('b04_absb', absb)
      call put('b04_fracrefa', fracrefa); call put('b04_fracrefb', fracrefb)
      call put('b04_selfref', selfref); call put('b04_forref', forref)
   end subroutine
   subroutine dump05
      use rrlw_kg05
      call put('b05_absa', absa); call put('b05_absb', absb)
      call put('b05_fracrefa', fracrefa); call put('b05_fracrefb', fracrefb)
      call put('b05_selfref', selfref); call put('b05_forref', forref)
      call put('b05_ka_mo3', ka_mo3); call put('b05_ccl4', ccl4)
   end subroutine
   subroutine dump06
      use rrlw_kg06
      call put('b06_absa', absa)
      call put('b06_fracrefa', fracrefa)
      call put('b06_selfref', selfref); call put('b06_forref', forref)
      call put('b06_ka_mco2', ka_mco2); call put('b06_cfc11adj', cfc11adj); call put('b06_cfc12', cfc12)
   end subroutine
   subroutine dump07
      use rrlw_kg07
      call put('b07_absa', absa); call put('b07_absb', absb)
      call put('b07_fracrefa', fracrefa); call put('b07_fracrefb', fracrefb)
      call put('b07_selfref', selfref); call put('b07_forref', forref)
      call put('b07_ka_mco2', ka_mco2); call put('b07_kb_mco2', kb_mco2)
   end subroutine
   subroutine dump08
      use rrlw_kg08
      call put('b08_absa', absa); call put('b08_absb', absb)
      call put('b08_fracrefa', fracrefa); call put('b08_fracrefb', fracrefb)
      call put('b08_selfref', selfref); call put('b08_forref', forref)
      call put('b08_ka_mco2', ka_mco2); call put('b08_kb_mco2', kb_mco2)
      call put('b08_ka_mo3', ka_mo3)
      call put('b08_ka_mn2o', ka_mn2o); call put('b08_kb_mn2o', kb_mn2o)
      call put('b08_cfc12', cfc12); call put('b08_cfc22adj', cfc22adj)
   end subroutine
   subroutine dump09
      use rrlw_kg09
      call put('b09_absa', absa); call put('b09_absb', absb)
      call put('b09_fracrefa', fracrefa); call put('b09_fracrefb', fracrefb)
      call put('b09_selfref', selfref); call put('b09_forref', forref)
      call put('b09_ka_mn2o', ka_mn2o); call put('b09_kb_mn2o', kb_mn2o)
   end subroutine
   subroutine dump10
      use rrlw_kg10
      call put('b10_absa', absa); call put('b10_absb', absb)
      call put('b10_fracrefa', fracrefa); call put('b10_fracrefb', fracrefb)
      call put('b10_selfref', selfref); call put('b10_forref', forref)
   end subroutine
   subroutine dump11
      use rrlw_kg11
      call put('b11_absa', absa); call put('b11_absb', absb)
      call put('b11_fracrefa', fracrefa); call put('b11_fracrefb', fracrefb)
      call put('b11_selfref', selfref); call put('b11_forref', forref)
      call put('b11_ka_mo2', ka_mo2); call put('b11_kb_mo2', kb_mo2)
   end subroutine
   subroutine dump12
      use rrlw_kg12
      call put('b12_absa', absa)
      call put('b12_fracrefa', fracrefa)
      call put('b12_selfref', selfref); call put('b12_forref', forref)
   end subroutine
   subroutine dump13
      use rrlw_kg13
      call put('b13_absa', absa)
      call put('b13_fracrefa', fracrefa); call put('b13_fracrefb', fracrefb)
      call put('b13_selfref', selfref); call put('b13_forref', forref)
      call put('b13_ka_mco2', ka_mco2); call put('b13_kb_mo3', kb_mo3); call put('b13_ka_mco', ka_mco)
   end subroutine
   subroutine dump14
      use rrlw_kg14
      call put('b14_absa', absa); call put('b14_absb', absb)
      call put('b14_fracrefa', fracrefa); call put('b14_fracrefb', fracrefb)
      call put('b14_selfref', selfref); call put('b14_forref', forref)
   end subroutine
   subroutine dump15
      use rrlw_kg15
      call put('b15_absa', absa)
      call put('b15_fracrefa', fracrefa)
      call put('b15_selfref', selfref); call put('b15_forref', forref)
      call put('b15_ka_mn2', ka_mn2)
   end subroutine
   subroutine dump16
      use rrlw_kg16
      call put('b16_absa', absa); call put('b16_absb', absb)
      call put('b16_fracrefa', fracrefa); call put('b16_fracrefb', fracrefb)
      call put('b16_selfref', selfref); call put('b16_forref', forref)
   end subroutine
end subroutine ref_lw_dump_tables

subroutine ref_rrtmg_lw( &
      ncol, nlay, psize, dudTs, play, plev, tlay, tlev, tsfc, emis, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, &
      cldf, ciwp, clwp, rei, rel, iceflglw, liqflglw, tauaer, zm, alat, dyofyr, cloudLM, cloudMH, &
      clearCounts, uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, band_output, olrb, dolrb_dTs) &
      bind(C, name='ref_rrtmg_lw')
   use iso_c_binding
   use rrtmg_lw_rad, only: rrtmg_lw
   implicit none
   integer(c_int), value :: ncol, nlay, psize, dudTs, iceflglw, liqflglw, dyofyr, cloudLM, cloudMH
   real, intent(in) :: play(ncol,nlay), plev(ncol,0:nlay), tlay(ncol,nlay), tlev(ncol,0:nlay)
   real, intent(in) :: tsfc(ncol), emis(ncol,16)
   real, intent(in), dimension(ncol,nlay) :: h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
      cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, cldf, ciwp, clwp, rei, rel, zm
   real, intent(in) :: tauaer(ncol,nlay,16), alat(ncol)
   integer(c_int), intent(out) :: clearCounts(ncol,4)
   real, intent(out), dimension(ncol,nlay+1) :: uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs
   integer(c_int), intent(in) :: band_output(16)
   real, intent(out) :: olrb(16,ncol), dolrb_dTs(16,ncol)
   logical :: bo(16)
   integer :: cc(ncol,4)
   bo = (band_output /= 0)
   call rrtmg_lw(ncol, nlay, psize, dudTs /= 0, play, plev, tlay, tlev, tsfc, emis, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, &
      cldf, ciwp, clwp, rei, rel, iceflglw, liqflglw, tauaer, zm, alat, dyofyr, cloudLM, cloudMH, &
      cc, uflx, dflx, uflxc, dflxc, duflx_dTs, duflxc_dTs, bo, olrb, dolrb_dTs)
   clearCounts = cc
end subroutine ref_rrtmg_lw

! setcoef + taumol only, inputs already in (nlay,ncol) order: intermediates for pinning the restatement
subroutine ref_lw_setcoef_taumol(ncol, nlay, dudTs, pavel, tavel, pz, tz, tbound, semiss, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, taua, &
      taug, pfracs, o_planklay, o_planklev, o_plankbnd, o_dplankbnd, o_pwvcm, o_laytrop) &
      bind(C, name='ref_lw_setcoef_taumol')
   use iso_c_binding
   use rrtmg_lw_setcoef
   use rrtmg_lw_taumol, only: taumol
   implicit none
   integer(c_int), value :: ncol, nlay, dudTs
   real, intent(in) :: pavel(nlay,ncol), tavel(nlay,ncol), pz(0:nlay,ncol), tz(0:nlay,ncol)
   real, intent(in) :: tbound(ncol), semiss(16,ncol)
   real, intent(in), dimension(nlay,ncol) :: h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
      cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr
   real, intent(in) :: taua(nlay,16,ncol)
   real, intent(out) :: taug(nlay,140,ncol), pfracs(nlay,140,ncol)
   real, intent(out) :: o_planklay(16,nlay,ncol), o_planklev(16,0:nlay,ncol)
   real, intent(out) :: o_plankbnd(16,ncol), o_dplankbnd(16,ncol), o_pwvcm(ncol)
   integer(c_int), intent(out) :: o_laytrop(ncol)
   real :: covmr(nlay,ncol)
   covmr = 0.
   call setcoef(ncol, nlay, 1, dudTs /= 0, pavel, tavel, pz, tz, tbound, semiss, &
      h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, covmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr)
   call taumol(ncol, nlay, pavel, taua, taug, pfracs)
   o_planklay = planklay; o_planklev = planklev; o_plankbnd = plankbnd
   if (dudTs /= 0) then
      o_dplankbnd = dplankbnd_dTs
   else
      o_dplankbnd = 0.
   end if
   o_pwvcm = pwvcm; o_laytrop = laytrop
   call setcoef_free
end subroutine ref_lw_setcoef_taumol


! ---------------------------------------------------------------------------------------------
! McICA generator + condensate inhomogeneity
! ---------------------------------------------------------------------------------------------

subroutine ref_set_inhomogeneity(ih) bind(C, name='ref_set_inhomogeneity')
   use iso_c_binding
   use cloud_condensate_inhomogeneity, only: set_inhomogeneity, unset_inhomogeneity
   integer(c_int), value :: ih
   call unset_inhomogeneity
   if (ih /= 0) call set_inhomogeneity(ih)
end subroutine

subroutine ref_init_subcol_gen(adl, rdl) bind(C, name='ref_init_subcol_gen')
   use cloud_subcol_gen, only: initialize_cloud_subcol_gen
   real, intent(in) :: adl(4), rdl(4)
   call initialize_cloud_subcol_gen(adl(1), adl(2), adl(3), adl(4), rdl(1), rdl(2), rdl(3), rdl(4))
end subroutine

subroutine ref_subcol_defaults(adl, rdl) bind(C, name='ref_subcol_defaults')
   use cloud_subcol_gen, only: def_aam1, def_aam2, def_aam30, def_aam4, def_ram1, def_ram2, def_ram30, def_ram4
   real, intent(out) :: adl(4), rdl(4)
   adl = [def_aam1, def_aam2, def_aam30, def_aam4]
   rdl = [def_ram1, def_ram2, def_ram30, def_ram4]
end subroutine

subroutine ref_zcw_lookup(n, cdf, sigma, zcw) bind(C, name='ref_zcw_lookup')
   use iso_c_binding
   use cloud_condensate_inhomogeneity, only: zcw_lookup
   integer(c_int), value :: n
   real, intent(in) :: cdf(n), sigma(n)
   real, intent(out) :: zcw(n)
   integer :: i
   do i = 1, n
      zcw(i) = zcw_lookup(cdf(i), sigma(i))
   end do
end subroutine

! Recover the module-private table xcw(1000,140) exactly through the public zcw_lookup: find
! arguments for which the bilinear weights are exactly (1,0,0,0) [or hit the clamped last row/col].
subroutine ref_dump_xcw(cpath, n, ih) bind(C, name='ref_dump_xcw')
   use iso_c_binding
   use ref_glue_io
   use cloud_condensate_inhomogeneity, only: zcw_lookup, set_inhomogeneity, unset_inhomogeneity
   implicit none
   character(kind=c_char), intent(in) :: cpath(*)
   integer(c_int), value :: n, ih
   integer, parameter :: n1 = 1000, n2 = 140
   real :: xcw(n1,n2), cdfs(n1), sigs(n2), c, s, r
   integer :: i, j, k, nmiss
   logical :: ok
   nmiss = 0
   call unset_inhomogeneity
   call set_inhomogeneity(ih)
   do i = 1, n1
      c = real(i-1) / real(n1-1)
      ok = .false.
      do k = 0, 64
         r = c * (n1 - 1) + 1.
         if (r == real(i)) then
            ok = .true.; exit
         end if
         if (r < real(i)) then
            c = nearest(c, 1.)
         else
            c = nearest(c, -1.)
         end if
      end do
      if (.not. ok) nmiss = nmiss + 1   ! not exactly reachable in this precision: nearest candidate
      cdfs(i) = c
   end do
   do j = 1, n2
      s = real(j+3) / 40.
      ok = .false.
      do k = 0, 64
         r = 40. * s - 3.
         if (r == real(j)) then
            ok = .true.; exit
         end if
         if (r < real(j)) then
            s = nearest(s, 1.)
         else
            s = nearest(s, -1.)
         end if
      end do
      if (.not. ok) nmiss = nmiss + 1
      sigs(j) = s
   end do
   do j = 1, n2
      do i = 1, n1
         xcw(i,j) = zcw_lookup(cdfs(i), sigs(j))
      end do
   end do
   call open_blob(cpath, n)
   call put('ih', int(ih))
   call put('inexact_points', nmiss)   ! 0 => xcw recovered bit-exactly
   call put('xcw', xcw)
   call close_blob
end subroutine ref_dump_xcw

subroutine ref_mcica(dncol, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny, &
      seed_order, cldy, ciwp_s, clwp_s) bind(C, name='ref_mcica')
   use iso_c_binding
   use cloud_subcol_gen, only: generate_stochastic_clouds
   implicit none
   integer(c_int), value :: dncol, ncol, nsubcol, nlay, doy
   real, intent(in) :: zmid(nlay,dncol), alat(dncol), play(nlay,dncol), cldfrac(nlay,dncol)
   real, intent(in) :: ciwp(nlay,dncol), clwp(nlay,dncol), cwp_tiny
   integer(c_int), intent(in) :: seed_order(4)
   integer(c_int), intent(out) :: cldy(nlay,nsubcol,dncol)
   real, intent(out) :: ciwp_s(nlay,nsubcol,dncol), clwp_s(nlay,nsubcol,dncol)
   logical, allocatable :: l(:,:,:)
   integer :: so(4)
   allocate(l(nlay,nsubcol,dncol))
   so = seed_order
   call generate_stochastic_clouds(dncol, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, &
      ciwp, clwp, cwp_tiny, l, ciwp_s, clwp_s, seed_order=so)
   cldy = merge(1, 0, l)
   deallocate(l)
end subroutine ref_mcica

subroutine ref_clearcounts(dncol, ncol, nsubcol, nlay, cloudLM, cloudMH, cldy, cnts) bind(C, name='ref_clearcounts')
   use iso_c_binding
   use cloud_subcol_gen, only: clearCounts_threeBand
   implicit none
   integer(c_int), value :: dncol, ncol, nsubcol, nlay, cloudLM, cloudMH
   integer(c_int), intent(in) :: cldy(nlay,nsubcol,dncol)
   integer(c_int), intent(out) :: cnts(4,dncol)
   logical, allocatable :: l(:,:,:)
   integer :: c(4,dncol)
   allocate(l(nlay,nsubcol,dncol))
   l = (cldy /= 0)
   call clearCounts_threeBand(dncol, ncol, nsubcol, nlay, cloudLM, cloudMH, l, c)
   cnts = c
   deallocate(l)
end subroutine ref_clearcounts

subroutine ref_corr_lengths(ncol, doy, alat, adl, rdl) bind(C, name='ref_corr_lengths')
   use iso_c_binding
   use cloud_subcol_gen, only: correlation_length_cloud_fraction, correlation_length_condensate
   integer(c_int), value :: ncol, doy
   real, intent(in) :: alat(ncol)
   real, intent(out) :: adl(ncol), rdl(ncol)
   call correlation_length_cloud_fraction(ncol, ncol, doy, alat, adl)
   call correlation_length_condensate(ncol, ncol, doy, alat, rdl)
end subroutine

! cldprmc alone (LW), inputs in reference partition order
subroutine ref_lw_cldprmc(ncol, nlay, cldy, ciwpmc, clwpmc, reice, reliq, iceflag, liqflag, taucmc, cloudy) &
      bind(C, name='ref_lw_cldprmc')
   use iso_c_binding
   use rrtmg_lw_cldprmc, only: cldprmc
   implicit none
   integer(c_int), value :: ncol, nlay, iceflag, liqflag
   integer(c_int), intent(in) :: cldy(nlay,140,ncol)
   real, intent(in) :: ciwpmc(nlay,140,ncol), clwpmc(nlay,140,ncol), reice(nlay,ncol), reliq(nlay,ncol)
   real, intent(out) :: taucmc(nlay,140,ncol)
   integer(c_int), intent(out) :: cloudy(nlay,ncol)
   logical, allocatable :: l(:,:,:)
   logical :: lc(nlay,ncol)
   allocate(l(nlay,140,ncol))
   l = (cldy /= 0)
   call cldprmc(ncol, nlay, l, ciwpmc, clwpmc, reice, reliq, iceflag, liqflag, taucmc, lc)
   cloudy = merge(1, 0, lc)
   deallocate(l)
end subroutine ref_lw_cldprmc
