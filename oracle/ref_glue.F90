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


! ---------------------------------------------------------------------------------------------
! RRTMG_SW: only the parts of the reference that compile without ESMF/MAPL
!   rrtmg_sw_ini   SW/src/rrtmg_sw_init.F90:49      setcoef_sw  SW/src/rrtmg_sw_setcoef.F90:23
!   taumol_sw      SW/src/rrtmg_sw_taumol.F90:27    cldprmc_sw  SW/src/rrtmg_sw_cldprmc.F90:36
! (SW = GEOSsolar_GridComp/RRTMG/rrtmg_sw/gcm_model)
! ---------------------------------------------------------------------------------------------

subroutine ref_sw_ini() bind(C, name='ref_sw_ini')
   use rrtmg_sw_init, only: rrtmg_sw_ini
   call rrtmg_sw_ini
end subroutine

subroutine ref_sw_dump_tables(cpath, n) bind(C, name='ref_sw_dump_tables')
   use iso_c_binding
   use ref_glue_io
   implicit none
   character(kind=c_char), intent(in) :: cpath(*)
   integer(c_int), value :: n
   call open_blob(cpath, n)
   call dump_common
   call d16; call d17; call d18; call d19; call d20; call d21; call d22
   call d23; call d24; call d25; call d26; call d27; call d28; call d29
   call close_blob
contains
   subroutine dump_common
      use parrrsw, only: nbndsw, ngptsw, jpb1, jpb2, rrsw_scon
      use rrsw_wvn, only: ng, nspa, nspb, ngb, ngs, ngc, icxa, wavenum1, wavenum2
      use rrsw_ref, only: pref, preflog, tref
      use rrsw_con, only: oneminus, grav, avogad, pi
      use rrsw_tbl, only: od_lo
      use rrsw_cld, only: extliq1, ssaliq1, asyliq1, extice2, ssaice2, asyice2, extice3, ssaice3, asyice3, fdlice3, &
         extice4, ssaice4, asyice4, abari, bbari, cbari, dbari, ebari, fbari
      use NRLSSI2, only: Iint, Fint, Sint, Mg_avg, Mg_0, SB_avg, SB_0
      call put('nbndsw', nbndsw); call put('ngptsw', ngptsw); call put('jpb1', jpb1); call put('jpb2', jpb2)
      call put('ng', ng); call put('nspa', nspa); call put('nspb', nspb); call put('ngb', ngb); call put('ngs', ngs)
      call put('ngc', ngc); call put('icxa', icxa); call put('wavenum1', wavenum1); call put('wavenum2', wavenum2)
      call put('pref', pref); call put('preflog', preflog); call put('tref', tref)
      call put('oneminus', oneminus); call put('grav', grav); call put('avogad', avogad); call put('pi', pi)
      call put('od_lo', od_lo); call put('rrsw_scon', rrsw_scon)
      call put('extliq1', extliq1); call put('ssaliq1', ssaliq1); call put('asyliq1', asyliq1)
      call put('extice2', extice2); call put('ssaice2', ssaice2); call put('asyice2', asyice2)
      call put('extice3', extice3); call put('ssaice3', ssaice3); call put('asyice3', asyice3); call put('fdlice3', fdlice3)
      call put('extice4', extice4); call put('ssaice4', ssaice4); call put('asyice4', asyice4)
      call put('abari', abari); call put('bbari', bbari); call put('cbari', cbari); call put('dbari', dbari)
      call put('ebari', ebari); call put('fbari', fbari)
      call put('Iint', Iint); call put('Fint', Fint); call put('Sint', Sint)
      call put('Mg_avg', Mg_avg); call put('Mg_0', Mg_0); call put('SB_avg', SB_avg); call put('SB_0', SB_0)
   end subroutine
   subroutine d16
      use rrsw_kg16
      call put('b16_absa', absa); call put('b16_absb', absb); call put('b16_selfref', selfref); call put('b16_forref', forref)
      call put('b16_sfluxref', sfluxref); call put('b16_irradnce', irradnce); call put('b16_facbrght', facbrght)
      call put('b16_snsptdrk', snsptdrk); call put('b16_rayl', rayl)
   end subroutine
   subroutine d17
      use rrsw_kg17
      call put('b17_absa', absa); call put('b17_absb', absb); call put('b17_selfref', selfref); call put('b17_forref', forref)
      call put('b17_sfluxref', sfluxref); call put('b17_irradnce', irradnce); call put('b17_facbrght', facbrght)
      call put('b17_snsptdrk', snsptdrk); call put('b17_rayl', rayl)
   end subroutine
   subroutine d18
      use rrsw_kg18
      call put('b18_absa', absa); call put('b18_absb', absb); call put('b18_selfref', selfref); call put('b18_forref', forref)
      call put('b18_sfluxref', sfluxref); call put('b18_irradnce', irradnce); call put('b18_facbrght', facbrght)
      call put('b18_snsptdrk', snsptdrk); call put('b18_rayl', rayl)
   end subroutine
   subroutine d19
      use rrsw_kg19
      call put('b19_absa', absa); call put('b19_absb', absb); call put('b19_selfref', selfref); call put('b19_forref', forref)
      call put('b19_sfluxref', sfluxref); call put('b19_irradnce', irradnce); call put('b19_facbrght', facbrght)
      call put('b19_snsptdrk', snsptdrk); call put('b19_rayl', rayl)
   end subroutine
   subroutine d20
      use rrsw_kg20
      call put('b20_absa', absa); call put('b20_absb', absb); call put('b20_selfref', selfref); call put('b20_forref', forref)
      call put('b20_sfluxref', sfluxref); call put('b20_irradnce', irradnce); call put('b20_facbrght', facbrght)
      call put('b20_snsptdrk', snsptdrk); call put('b20_rayl', rayl); call put('b20_absch4', absch4)
   end subroutine
   subroutine d21
      use rrsw_kg21
      call put('b21_absa', absa); call put('b21_absb', absb); call put('b21_selfref', selfref); call put('b21_forref', forref)
      call put('b21_sfluxref', sfluxref); call put('b21_irradnce', irradnce); call put('b21_facbrght', facbrght)
      call put('b21_snsptdrk', snsptdrk); call put('b21_rayl', rayl)
   end subroutine
   subroutine d22
      use rrsw_kg22
      call put('b22_absa', absa); call put('b22_absb', absb); call put('b22_selfref', selfref); call put('b22_forref', forref)
      call put('b22_sfluxref', sfluxref); call put('b22_irradnce', irradnce); call put('b22_facbrght', facbrght)
      call put('b22_snsptdrk', snsptdrk); call put('b22_rayl', rayl)
   end subroutine
   subroutine d23
      use rrsw_kg23
      call put('b23_absa', absa); call put('b23_selfref', selfref); call put('b23_forref', forref)
      call put('b23_sfluxref', sfluxref); call put('b23_irradnce', irradnce); call put('b23_facbrght', facbrght)
      call put('b23_snsptdrk', snsptdrk); call put('b23_rayl', rayl)
   end subroutine
   subroutine d24
      use rrsw_kg24
      call put('b24_absa', absa); call put('b24_absb', absb); call put('b24_selfref', selfref); call put('b24_forref', forref)
      call put('b24_sfluxref', sfluxref); call put('b24_irradnce', irradnce); call put('b24_facbrght', facbrght)
      call put('b24_snsptdrk', snsptdrk); call put('b24_abso3a', abso3a); call put('b24_abso3b', abso3b)
      call put('b24_rayla', rayla); call put('b24_raylb', raylb)
   end subroutine
   subroutine d25
      use rrsw_kg25
      call put('b25_absa', absa); call put('b25_sfluxref', sfluxref); call put('b25_irradnce', irradnce)
      call put('b25_facbrght', facbrght); call put('b25_snsptdrk', snsptdrk); call put('b25_abso3a', abso3a)
      call put('b25_abso3b', abso3b); call put('b25_rayl', rayl)
   end subroutine
   subroutine d26
      use rrsw_kg26
      call put('b26_sfluxref', sfluxref); call put('b26_irradnce', irradnce); call put('b26_facbrght', facbrght)
      call put('b26_snsptdrk', snsptdrk); call put('b26_rayl', rayl)
   end subroutine
   subroutine d27
      use rrsw_kg27
      call put('b27_absa', absa); call put('b27_absb', absb); call put('b27_sfluxref', sfluxref); call put('b27_irradnce', irradnce)
      call put('b27_facbrght', facbrght); call put('b27_snsptdrk', snsptdrk); call put('b27_rayl', rayl)
   end subroutine
   subroutine d28
      use rrsw_kg28
      call put('b28_absa', absa); call put('b28_absb', absb); call put('b28_sfluxref', sfluxref); call put('b28_irradnce', irradnce)
      call put('b28_facbrght', facbrght); call put('b28_snsptdrk', snsptdrk); call put('b28_rayl', rayl)
   end subroutine
   subroutine d29
      use rrsw_kg29
      call put('b29_absa', absa); call put('b29_absb', absb); call put('b29_selfref', selfref); call put('b29_forref', forref)
      call put('b29_sfluxref', sfluxref); call put('b29_irradnce', irradnce); call put('b29_facbrght', facbrght)
      call put('b29_snsptdrk', snsptdrk); call put('b29_rayl', rayl); call put('b29_absh2o', absh2o); call put('b29_absco2', absco2)
   end subroutine
end subroutine ref_sw_dump_tables

! setcoef_sw + taumol_sw on columns given in the reference's partition layout (nlay,ncol).  The dry-air /
! gas column amounts that rrtmg_sw_sub forms inline before setcoef_sw (rrtmg_sw_rad.F90:1370-1387; that file
! itself needs MAPL) are formed here the same way.
subroutine ref_sw_setcoef_taumol(ncol, nlay, play, tlay, plev, h2ovmr, co2vmr, o3vmr, ch4vmr, o2vmr, isolvar, svar, svar_bnd, &
      taug, taur, ssi, sfluxzen, o_colmol, o_laytrop) bind(C, name='ref_sw_setcoef_taumol')
   use iso_c_binding
   use parrrsw, only: ngptsw, jpband
   use rrsw_con, only: grav, avogad
   use rrtmg_sw_setcoef, only: setcoef_sw
   use rrtmg_sw_taumol, only: taumol_sw
   implicit none
   integer(c_int), value :: ncol, nlay, isolvar
   real, intent(in) :: play(nlay,ncol), tlay(nlay,ncol), plev(nlay+1,ncol)
   real, intent(in), dimension(nlay,ncol) :: h2ovmr, co2vmr, o3vmr, ch4vmr, o2vmr
   real, intent(in) :: svar(3), svar_bnd(jpband,3)
   real, intent(out) :: taug(nlay,ngptsw,ncol), taur(nlay,ngptsw,ncol), ssi(ngptsw,ncol), sfluxzen(ngptsw,ncol)
   real, intent(out) :: o_colmol(nlay,ncol)
   integer(c_int), intent(out) :: o_laytrop(ncol)
   real, parameter :: amd = 28.9660, amw = 18.0160
   real, dimension(nlay,ncol) :: coldry, colh2o, colco2, colo3, colch4, colo2, colmol, fac00, fac01, fac10, fac11, &
      selffac, selffrac, forfac, forfrac
   integer, dimension(nlay,ncol) :: jp, jt, jt1, indself, indfor
   integer :: laytrop(ncol), icol, ilay
   colh2o = h2ovmr; colco2 = co2vmr; colo3 = o3vmr; colch4 = ch4vmr; colo2 = o2vmr
   do icol = 1,ncol
      do ilay = 1,nlay
         coldry(ilay,icol) = (plev(ilay,icol)-plev(ilay+1,icol)) * 1.e3 * avogad / &
            (1.e2 * grav * ((1.-colh2o(ilay,icol)) * amd + colh2o(ilay,icol) * amw) * &
            (1. + colh2o(ilay,icol)))
      enddo
   enddo
   colh2o = coldry * colh2o; colco2 = coldry * colco2; colo3 = coldry * colo3; colch4 = coldry * colch4; colo2 = coldry * colo2
   call setcoef_sw(ncol, ncol, nlay, play, tlay, coldry, colch4, colco2, colh2o, colmol, colo2, colo3, &
      laytrop, jp, jt, jt1, fac00, fac01, fac10, fac11, selffac, selffrac, indself, forfac, forfrac, indfor)
   ssi = 0.; sfluxzen = 0.
   call taumol_sw(ncol, ncol, nlay, colh2o, colco2, colch4, colo2, colo3, colmol, laytrop, jp, jt, jt1, &
      fac00, fac01, fac10, fac11, selffac, selffrac, indself, forfac, forfrac, indfor, &
      isolvar, svar(1), svar(2), svar(3), svar_bnd(:,1), svar_bnd(:,2), svar_bnd(:,3), ssi, sfluxzen, taug, taur)
   o_colmol = colmol; o_laytrop = laytrop
end subroutine ref_sw_setcoef_taumol

subroutine ref_sw_cldprmc(ncol, nlay, iceflag, liqflag, cldy, ciwpmc, clwpmc, rei, rel, taormc, taucmc, ssacmc, asmcmc) &
      bind(C, name='ref_sw_cldprmc')
   use iso_c_binding
   use parrrsw, only: ngptsw
   use rrtmg_sw_cldprmc, only: cldprmc_sw
   implicit none
   integer(c_int), value :: ncol, nlay, iceflag, liqflag
   integer(c_int), intent(in) :: cldy(nlay,ngptsw,ncol)
   real, intent(in) :: ciwpmc(nlay,ngptsw,ncol), clwpmc(nlay,ngptsw,ncol), rei(nlay,ncol), rel(nlay,ncol)
   real, intent(out), dimension(nlay,ngptsw,ncol) :: taormc, taucmc, ssacmc, asmcmc
   logical, allocatable :: l(:,:,:)
   allocate(l(nlay,ngptsw,ncol))
   l = (cldy /= 0)
   call cldprmc_sw(ncol, ncol, nlay, iceflag, liqflag, l, ciwpmc, clwpmc, rei, rel, taormc, taucmc, ssacmc, asmcmc)
   deallocate(l)
end subroutine ref_sw_cldprmc

! NRLSSI2 (SW/src/NRLSSI2.F90; no ESMF/MAPL dependency): the two host routines of the isolvar = 1 branch of the solar-variability
! block (rrtmg_sw_rad.F90:906-930,994-1008) and the cycle means initialize_NRLSSI2 keeps for it (NRLSSI2.F90:160-232)
subroutine ref_nrlssi2_adjust(solcycfr, indsolvar, scl) bind(C, name='ref_nrlssi2_adjust')
   use NRLSSI2, only: adjust_solcyc_amplitudes
   implicit none
   real, intent(in) :: solcycfr, indsolvar(2)
   real, intent(out) :: scl(2)
   call adjust_solcyc_amplitudes(solcycfr, indsolvar, scl)
end subroutine

subroutine ref_nrlssi2_interp(solcycfr, Mg, SB) bind(C, name='ref_nrlssi2_interp')
   use NRLSSI2, only: interpolate_indices
   implicit none
   real, intent(in) :: solcycfr
   real, intent(out) :: Mg, SB
   call interpolate_indices(solcycfr, Mg, SB)
end subroutine

subroutine ref_nrlssi2_means(has_ind, indsolvar, mean_f, mean_s) bind(C, name='ref_nrlssi2_means')
   use iso_c_binding
   use NRLSSI2, only: initialize_NRLSSI2, isolvar_1_mean_svar_f, isolvar_1_mean_svar_s
   implicit none
   integer(c_int), value :: has_ind
   real, intent(in) :: indsolvar(2)
   real, intent(out) :: mean_f, mean_s
   if (has_ind /= 0) then
      call initialize_NRLSSI2(1, indsolvar)
   else
      call initialize_NRLSSI2(1)
   end if
   mean_f = isolvar_1_mean_svar_f; mean_s = isolvar_1_mean_svar_s
end subroutine

! Chou-Suarez LW coefficient tables (irrad_constants, rad_constants): data modules only -- irrad.F90 itself is not
! buildable here (module gettau -> MAPL_ConstantsMod).
subroutine ref_chou_lw_dump_tables(cpath, n) bind(C, name='ref_chou_lw_dump_tables')
   use iso_c_binding
   use ref_glue_io
   use irrad_constants
   use rad_constants, only: aib_ir, awb_ir, aiw_ir, aww_ir, aig_ir, awg_ir
   implicit none
   character(kind=c_char), intent(in) :: cpath(*)
   integer(c_int), value :: n
   call open_blob(cpath, n)
   call put('xkw', xkw); call put('xke', xke); call put('mw', mw); call put('aw', aw); call put('bw', bw); call put('pm', pm)
   call put('fkw', fkw); call put('gkw', gkw); call put('cb', cb); call put('dcb', dcb)
   call put('w11', w11); call put('w12', w12); call put('w13', w13); call put('p11', p11); call put('p12', p12); call put('p13', p13)
   call put('dwe', dwe); call put('dpe', dpe)
   call put('c1', c1); call put('c2', c2); call put('c3', c3); call put('oo1', oo1); call put('oo2', oo2); call put('oo3', oo3)
   call put('h11', h11); call put('h12', h12); call put('h13', h13); call put('h21', h21); call put('h22', h22); call put('h23', h23)
   call put('h81', h81); call put('h82', h82); call put('h83', h83)
   call put('aib_ir', aib_ir); call put('awb_ir', awb_ir); call put('aiw_ir', aiw_ir); call put('aww_ir', aww_ir)
   call put('aig_ir', aig_ir); call put('awg_ir', awg_ir)
   call close_blob
end subroutine ref_chou_lw_dump_tables

! Chou-Suarez SW coefficient tables (sorad_constants, UV/NIR part of rad_constants): data modules only -- sorad.F90 itself is
! not buildable here (MAPL_ConstantsMod).
subroutine ref_chou_sw_dump_tables(cpath, n) bind(C, name='ref_chou_sw_dump_tables')
   use iso_c_binding
   use ref_glue_io
   use sorad_constants
   use rad_constants, only: aig_uv, awg_uv, arg_uv, aib_uv, awb_uv, arb_uv, aib_nir, awb_nir, arb_nir, aia_nir, awa_nir, ara_nir, &
      aig_nir, awg_nir, arg_nir, caib, caif
   implicit none
   character(kind=c_char), intent(in) :: cpath(*)
   integer(c_int), value :: n
   call open_blob(cpath, n)
   call put('zk_uv', zk_uv); call put('wk_uv', wk_uv); call put('ry_uv', ry_uv); call put('xk_ir', xk_ir); call put('ry_ir', ry_ir)
   call put('coa', coa); call put('cah', cah); call put('hk_uv_old', hk_uv_old); call put('hk_ir_old', hk_ir_old)
   call put('aig_uv', aig_uv); call put('awg_uv', awg_uv); call put('arg_uv', arg_uv); call put('aib_uv', aib_uv)
   call put('awb_uv', awb_uv); call put('arb_uv', arb_uv); call put('aib_nir', aib_nir); call put('awb_nir', awb_nir)
   call put('arb_nir', arb_nir); call put('aia_nir', aia_nir); call put('awa_nir', awa_nir); call put('ara_nir', ara_nir)
   call put('aig_nir', aig_nir); call put('awg_nir', awg_nir); call put('arg_nir', arg_nir); call put('caib', caib); call put('caif', caif)
   call close_blob
end subroutine ref_chou_sw_dump_tables
