! chou_shims.F90 -- drop-in modules with the REFERENCE's module and procedure names for the Chou-Suarez schemes; bodies call
! the C ABI.  GEOS_IrradGridComp.F90 / GEOS_SolarGridComp.F90 `use irradmod, only: irrad` / `use soradmod, only: sorad` and call
!   call IRRAD(IM*JM, LM, PLE, T, Q, O3, T2M, CO2, TRACE, N2O, CH4, CFC11, CFC12, HCFC22, CWC, FCLD, LCLDMH, LCLDLM, REFF, NS, FS, TG,
!              EG, TV, EV, RV, NA, NB_CHOU, TAUA, SSAA, ASYA, FLXU, FLCU, FLAU, FLXAU, FLXD, FLCD, FLAD, FLXAD, DFDTS, SFCEM, TAUDIAG)
!                                                                                         (GEOS_IrradGridComp.F90:2093-2101)
!   irradmod : irrad      GEOSirrad_GridComp/irrad.F90:27-35
!   soradmod : sorad      GEOSsolar_GridComp/sorad.F90:43-51
! The coefficient tables (the reference keeps them as module data in irrad_constants / sorad_constants / rad_constants) are
! uploaded to HBM on the first call.
module irradmod
   use iso_c_binding
   use geosrad_c
   implicit none
   private
   public :: irrad
   logical, save :: loaded = .false.
contains
   subroutine irrad (m,np,ple,ta,wa,oa,tb,co2,&
                     trace,n2o,ch4,cfc11,cfc12,cfc22,&
                     cwc,fcld,ict,icb,reff,&
                     ns,fs,tg,eg,tv,ev,rv,&
                     na,nb,taua,ssaa,asya,&
                     flxu,flcu,flau,flxau,&
                     flxd,flcd,flad,flxad,&
                     dfdts,sfcem,taudiag)
      integer :: m,np,na,ns,ict,icb,nb
      logical :: trace
      real    :: co2
      real, target :: ple(m,np+1),ta(m,np),wa(m,np),oa(m,np),tb(m)
      real, target :: n2o(m,np),ch4(m,np),cfc11(m,np),cfc12(m,np),cfc22(m,np)
      real, target :: fs(m,ns),tg(m,ns),eg(m,ns,10)
      real, target :: tv(m,ns),ev(m,ns,10),rv(m,ns,10)
      real, target :: cwc(m,np,4),fcld(m,np),reff(m,np,4)
      real, target :: taua(m,np,nb),ssaa(m,np,nb),asya(m,np,nb)
      real, target :: flxu(m,np+1),flxau(m,np+1),flcu(m,np+1),flau(m,np+1)
      real, target :: flxd(m,np+1),flxad(m,np+1),flcd(m,np+1),flad(m,np+1)
      real, target :: dfdts(m,np+1),sfcem(m),taudiag(m,np,10)
      integer(c_int) :: rc
      real :: x
      if (.not. loaded) then
         if (kind(x) == 4) then
            rc = geosrad_load_tables_chou_lw(geosrad_ctx_handle(), geosrad_data_path('chou_lw_r4.grtb'))
         else
            rc = geosrad_load_tables_chou_lw(geosrad_ctx_handle(), geosrad_data_path('chou_lw_r8.grtb'))
         end if
         if (rc /= 0) call geosrad_fail('irrad (tables)')
         loaded = .true.
      end if
      rc = geosrad_irrad(geosrad_ctx_handle(), int(m,c_int), int(np,c_int), c_loc(ple), c_loc(ta), c_loc(wa), c_loc(oa), c_loc(tb), &
         real(co2,c_double), merge(1_c_int, 0_c_int, trace), c_loc(n2o), c_loc(ch4), c_loc(cfc11), c_loc(cfc12), c_loc(cfc22), &
         c_loc(cwc), c_loc(fcld), int(ict,c_int), int(icb,c_int), c_loc(reff), int(ns,c_int), c_loc(fs), c_loc(tg), c_loc(eg), &
         c_loc(tv), c_loc(ev), c_loc(rv), int(na,c_int), int(nb,c_int), c_loc(taua), c_loc(ssaa), c_loc(asya), &
         c_loc(flxu), c_loc(flcu), c_loc(flau), c_loc(flxau), c_loc(flxd), c_loc(flcd), c_loc(flad), c_loc(flxad), &
         c_loc(dfdts), c_loc(sfcem), c_loc(taudiag))
      if (rc /= 0) call geosrad_fail('irrad')
   end subroutine irrad
end module irradmod

module soradmod
   use iso_c_binding
   use geosrad_c
   implicit none
   private
   public :: sorad
   integer, parameter :: nband = 8
   logical, save :: loaded = .false.
contains
   subroutine sorad (m,np,nb,cosz,pl,ta,wa,oa,co2,&
         cwc,fcld,ict,icb,reff,hk_uv,hk_ir,&
         taua,ssaa,asya,&
         rsuvbm,rsuvdf,rsirbm,rsirdf,&
         flx,flc,fdiruv,fdifuv,&
         fdirpar,fdifpar,fdirir,fdifir,&
         flxu,flcu,&
         flx_sfc_band,&
         do_drfband,drband,dfband)
      integer :: m,np,ict,icb,nb
      real, target :: cosz(m),pl(m,np+1),ta(m,np),wa(m,np),oa(m,np)
      real :: co2
      real, target :: cwc(m,np,4),fcld(m,np),reff(m,np,4),hk_uv(5),hk_ir(3,10)
      real, target :: rsuvbm(m),rsuvdf(m),rsirbm(m),rsirdf(m)
      real, target :: taua(m,np,nb),ssaa(m,np,nb),asya(m,np,nb)
      logical, intent(in) :: do_drfband
      real, target :: flx(m,np+1),flc(m,np+1),flxu(m,np+1),flcu(m,np+1)
      real, target :: fdiruv(m),fdifuv(m),fdirpar(m),fdifpar(m),fdirir(m),fdifir(m)
      real, target :: flx_sfc_band(m,nband)
      real, intent(inout), dimension(:,:), pointer :: drband, dfband      ! only touched if (do_drfband), as in the reference
      integer(c_int) :: rc
      real :: x
      type(c_ptr) :: pdr, pdf
      real, allocatable, target :: dr(:,:), df(:,:)
      if (.not. loaded) then
         if (kind(x) == 4) then
            rc = geosrad_load_tables_chou_sw(geosrad_ctx_handle(), geosrad_data_path('chou_sw_r4.grtb'))
         else
            rc = geosrad_load_tables_chou_sw(geosrad_ctx_handle(), geosrad_data_path('chou_sw_r8.grtb'))
         end if
         if (rc /= 0) call geosrad_fail('sorad (tables)')
         loaded = .true.
      end if
      pdr = c_null_ptr; pdf = c_null_ptr
      if (do_drfband) then       ! the pointer targets need not be contiguous: stage through contiguous buffers
         allocate(dr(m,nband), df(m,nband))
         pdr = c_loc(dr); pdf = c_loc(df)
      end if
      rc = geosrad_sorad(geosrad_ctx_handle(), int(m,c_int), int(np,c_int), int(nb,c_int), c_loc(cosz), c_loc(pl), c_loc(ta), c_loc(wa), &
         c_loc(oa), real(co2,c_double), c_loc(cwc), c_loc(fcld), int(ict,c_int), int(icb,c_int), c_loc(reff), c_loc(hk_uv), c_loc(hk_ir), &
         c_loc(taua), c_loc(ssaa), c_loc(asya), c_loc(rsuvbm), c_loc(rsuvdf), c_loc(rsirbm), c_loc(rsirdf), c_loc(flx), c_loc(flc), &
         c_loc(fdiruv), c_loc(fdifuv), c_loc(fdirpar), c_loc(fdifpar), c_loc(fdirir), c_loc(fdifir), c_loc(flxu), c_loc(flcu), &
         c_loc(flx_sfc_band), merge(1_c_int, 0_c_int, do_drfband), pdr, pdf)
      if (rc /= 0) call geosrad_fail('sorad')
      if (do_drfband) then
         drband(1:m,1:nband) = dr; dfband(1:m,1:nband) = df
      end if
   end subroutine sorad
end module soradmod
