! cloud_shims.F90 -- drop-in modules for GEOS_RadiationShared's McICA interface
! (cloud_subcol_gen.F90:98-103, cloud_condensate_inhomogeneity.F90:34-37), bodies on the C ABI.
!
! The public lists are the reference's: besides the calls of RAD:Initialize (set_inhomogeneity, initialize_cloud_subcol_gen,
! def_*) and of the RRTMG solvers, the two leaf GridComps import the HOST functions zcw_lookup,
! correlation_length_cloud_fraction and correlation_length_condensate unconditionally (GEOS_IrradGridComp.F90:1472-1474,
! GEOS_SolarGridComp.F90:3346-3348; their RRTMGP branches call them per column on the CPU).  Those three are therefore plain
! Fortran here, evaluated in the caller's default real kind on a host copy of the same table / parameters the device holds.
module cloud_condensate_inhomogeneity
   use iso_c_binding
   use geosrad_c
   implicit none
   private
   public :: set_inhomogeneity, unset_inhomogeneity, condensate_inhomogeneous, zcw_lookup
   public :: geosrad_host_inhomogeneity      ! (ours) host table only, no device: what set_inhomogeneity does first
   integer, parameter :: n1 = 1000, n2 = 140              ! cloud_condensate_inhomogeneity.F90:24-25
   integer, save :: inhm = 0
   real, allocatable, save, target :: xcw(:,:)
contains
   ! host side of set_inhomogeneity: the xcw(n1,n2) table of the default real kind from the shipped coefficient file
   subroutine geosrad_host_inhomogeneity(ih)
      integer, intent(in) :: ih
      integer(c_int) :: rc
      real :: x
      character(len=:), allocatable :: f
      if (allocated(xcw)) deallocate(xcw)
      inhm = 0
      if (ih == 0) return
      if (ih /= 1 .and. ih /= 2) error stop 'unknown inhomogeneity type'
      f = merge('xcw_beta_ ', 'xcw_gamma_', ih == 1)
      f = trim(f) // merge('r4', 'r8', kind(x) == 4) // '.grtb'
      allocate(xcw(n1,n2))
      rc = geosrad_read_table(geosrad_data_path(f), 'xcw' // c_null_char, int(kind(x),c_int), c_loc(xcw), int(n1*n2, c_size_t))
      if (rc /= 0) error stop 'geosrad: cannot read the xcw table (set GEOSRAD_DATA)'
      inhm = ih
   end subroutine
   subroutine set_inhomogeneity(ih)
      integer, intent(in) :: ih
      integer(c_int) :: rc
      real :: x
      character(len=:), allocatable :: f
      if (ih == inhm) return
      if (inhm /= 0) error stop 'must call unset_inhomogeneity first'
      if (ih /= 1 .and. ih /= 2) error stop 'unknown inhomogeneity type'
      call geosrad_host_inhomogeneity(ih)
      f = merge('xcw_beta_ ', 'xcw_gamma_', ih == 1)
      f = trim(f) // merge('r4', 'r8', kind(x) == 4) // '.grtb'
      rc = geosrad_load_inhomogeneity(geosrad_ctx_handle(), int(ih,c_int), geosrad_data_path(f))
      if (rc /= 0) call geosrad_fail('set_inhomogeneity')
   end subroutine
   subroutine unset_inhomogeneity
      integer(c_int) :: rc
      rc = geosrad_load_inhomogeneity(geosrad_ctx_handle(), 0_c_int, c_null_char)
      if (rc /= 0) call geosrad_fail('unset_inhomogeneity')
      call geosrad_host_inhomogeneity(0)
   end subroutine
   pure function condensate_inhomogeneous() result(inhomo)
      logical :: inhomo
      inhomo = (inhm > 0)
   end function
   ! ratio of the condensate to its in-cloud mean at cumulative probability cdf for relative standard deviation sigma_qcw:
   ! bilinear in the table, index conventions of cloud_condensate_inhomogeneity.F90:86-124 (the device path is
   ! mcica_kernels.hpp zcw_lookup; both read the same numbers)
   pure function zcw_lookup(cdf, sigma_qcw) result(zcw)
      real, intent(in) :: cdf, sigma_qcw
      real :: zcw, f1, f2
      integer :: i1, i2
      if (inhm == 0) then
         zcw = 1.
         return
      end if
      f1 = cdf * (n1 - 1) + 1.
      i1 = max(1, min(int(f1), n1-1))
      f1 = f1 - i1
      f2 = 40. * sigma_qcw - 3.
      i2 = max(1, min(int(f2), n2-1))
      f2 = f2 - i2
      zcw = (1.0-f1) * (1.0-f2) * xcw(i1,i2) + (1.0-f1) * f2 * xcw(i1,i2+1) &
          + f1 * (1.0-f2) * xcw(i1+1,i2) + f1 * f2 * xcw(i1+1,i2+1)
   end function
end module cloud_condensate_inhomogeneity

module cloud_subcol_gen
   use iso_c_binding
   use geosrad_c
   implicit none
   private
   real, public, parameter :: def_aam1 = 1.4315, def_aam2 = 2.1219, def_aam30 = 7., def_aam4 = -25.584
   real, public, parameter :: def_ram1 = 0.72192, def_ram2 = 0.78996, def_ram30 = 8.5, def_ram4 = 40.404
   real, save :: aam(4) = [def_aam1, def_aam2, def_aam30, def_aam4], ram(4) = [def_ram1, def_ram2, def_ram30, def_ram4]
   public :: initialize_cloud_subcol_gen, generate_stochastic_clouds, clearCounts_threeBand
   public :: correlation_length_cloud_fraction, correlation_length_condensate
contains
   subroutine initialize_cloud_subcol_gen(adl_am1, adl_am2, adl_am30, adl_am4, rdl_am1, rdl_am2, rdl_am30, rdl_am4)
      real, intent(in), optional :: adl_am1, adl_am2, adl_am30, adl_am4, rdl_am1, rdl_am2, rdl_am30, rdl_am4
      integer(c_int) :: rc
      if (present(adl_am1)) aam(1) = adl_am1;  if (present(adl_am2)) aam(2) = adl_am2
      if (present(adl_am30)) aam(3) = adl_am30; if (present(adl_am4)) aam(4) = adl_am4
      if (present(rdl_am1)) ram(1) = rdl_am1;  if (present(rdl_am2)) ram(2) = rdl_am2
      if (present(rdl_am30)) ram(3) = rdl_am30; if (present(rdl_am4)) ram(4) = rdl_am4
      rc = geosrad_set_corr_lengths(geosrad_ctx_handle(), real(aam, c_double), real(ram, c_double))
      if (rc /= 0) call geosrad_fail('initialize_cloud_subcol_gen')
   end subroutine

   ! decorrelation lengths [m] of Oreopoulos et al. (2012), cloud_subcol_gen.F90:491-542: host functions (the RRTMGP branches
   ! of the GridComps call them); the solvers evaluate the same expression on the device (mcica_kernels.hpp k_overlap)
   pure subroutine corr_length(dncol, ncol, am, doy, alat, clength)
      integer, intent(in) :: dncol, ncol, doy
      real, intent(in) :: am(4), alat(dncol)
      real, intent(out) :: clength(dncol)
      real, parameter :: r2d = 180.d0 / 3.14159265358979323846d0
      real :: am3
      integer :: i
      if (doy > 181) then
         am3 = -4.*am(3)/365.*(doy-272)
      else
         am3 =  4.*am(3)/365.*(doy- 91)
      end if
      do i = 1, ncol
         clength(i) = (am(1)+am(2)*exp(-(alat(i)*r2d-am3)**2/am(4)**2))*1.e3
      end do
   end subroutine
   pure subroutine correlation_length_cloud_fraction(dncol, ncol, doy, alat, clength)
      integer, intent(in) :: dncol, ncol, doy
      real, intent(in) :: alat(dncol)
      real, intent(out) :: clength(dncol)
      call corr_length(dncol, ncol, aam, doy, alat, clength)
   end subroutine
   pure subroutine correlation_length_condensate(dncol, ncol, doy, alat, clength)
      integer, intent(in) :: dncol, ncol, doy
      real, intent(in) :: alat(dncol)
      real, intent(out) :: clength(dncol)
      call corr_length(dncol, ncol, ram, doy, alat, clength)
   end subroutine

   ! reference layout is (nlay,dncol) in / (nlay,nsubcol,dncol) out; the C ABI takes the solver-API layout
   ! (ncol,nlay) for the profiles, so this compatibility path transposes on the host.  (rrtmg_lw does NOT come
   ! through here: its sub-columns are generated on the GPU inside the solver.)
   subroutine generate_stochastic_clouds(dncol, ncol, nsubcol, nlay, zmid, alat, doy, play, cldfrac, ciwp, clwp, cwp_tiny, &
         cldy_stoch, ciwp_stoch, clwp_stoch, seed_order)
      integer, intent(in) :: dncol, ncol, nsubcol, nlay, doy
      real, intent(in) :: zmid(nlay,dncol), alat(dncol), play(nlay,dncol), cldfrac(nlay,dncol), ciwp(nlay,dncol), clwp(nlay,dncol)
      real, intent(in) :: cwp_tiny
      logical, intent(out) :: cldy_stoch(nlay,nsubcol,dncol)
      real, intent(out), target :: ciwp_stoch(nlay,nsubcol,dncol), clwp_stoch(nlay,nsubcol,dncol)
      integer, intent(in), optional :: seed_order(4)
      real, target :: z(ncol,nlay), p(ncol,nlay), f(ncol,nlay), ci(ncol,nlay), cl(ncol,nlay), al(ncol)
      integer(c_int), target :: so(4), cy(nlay,nsubcol,ncol)
      real, target :: cis(nlay,nsubcol,ncol), cls(nlay,nsubcol,ncol)
      integer(c_int) :: rc
      so = [1, 2, 3, 4]
      if (present(seed_order)) so = seed_order
      z = transpose(zmid(:,1:ncol)); p = transpose(play(:,1:ncol)); f = transpose(cldfrac(:,1:ncol))
      ci = transpose(ciwp(:,1:ncol)); cl = transpose(clwp(:,1:ncol)); al = alat(1:ncol)
      rc = geosrad_mcica(geosrad_ctx_handle(), int(ncol,c_int), int(nsubcol,c_int), int(nlay,c_int), c_loc(z), c_loc(al), &
         int(doy,c_int), c_loc(p), c_loc(f), c_loc(ci), c_loc(cl), real(cwp_tiny,c_double), c_loc(so), c_loc(cy), c_loc(cis), c_loc(cls))
      if (rc /= 0) call geosrad_fail('generate_stochastic_clouds')
      cldy_stoch(:,:,1:ncol) = cy /= 0
      ciwp_stoch(:,:,1:ncol) = cis; clwp_stoch(:,:,1:ncol) = cls
   end subroutine

   subroutine clearCounts_threeBand(dncol, ncol, nsubcol, nlay, cloudLM, cloudMH, cldy_stoch, clearCnts)
      integer, intent(in) :: dncol, ncol, nsubcol, nlay, cloudLM, cloudMH
      logical, intent(in) :: cldy_stoch(nlay,nsubcol,dncol)
      integer, intent(out) :: clearCnts(4,dncol)
      integer(c_int), target :: cy(nlay,nsubcol,ncol), cnt(4,ncol)
      integer(c_int) :: rc
      cy = merge(1_c_int, 0_c_int, cldy_stoch(:,:,1:ncol))
      rc = geosrad_clearcounts(geosrad_ctx_handle(), int(ncol,c_int), int(nsubcol,c_int), int(nlay,c_int), int(cloudLM,c_int), &
         int(cloudMH,c_int), c_loc(cy), c_loc(cnt))
      if (rc /= 0) call geosrad_fail('clearCounts_threeBand')      ! 'invalid pressure super-layers!'
      clearCnts = 0
      clearCnts(:,1:ncol) = cnt
   end subroutine
end module cloud_subcol_gen
