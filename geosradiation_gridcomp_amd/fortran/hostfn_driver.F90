! hostfn_driver.F90 -- the HOST functions of the McICA modules that the GridComps import (zcw_lookup,
! correlation_length_cloud_fraction / _condensate; GEOS_IrradGridComp.F90:1472-1474), evaluated through the shim modules
! WITHOUT a device: reads (ih, doy, n, cdf(n), sigma(n), alat(n)) and writes zcw(n), adl(n), rdl(n).
! tests/test_host.py compares the result bit for bit with the reference's own functions (oracle/_ref).
program hostfn_driver
   use cloud_condensate_inhomogeneity, only: condensate_inhomogeneous, zcw_lookup, geosrad_host_inhomogeneity
   use cloud_subcol_gen, only: correlation_length_cloud_fraction, correlation_length_condensate
   implicit none
   integer :: ih, doy, n, u, i
   real, allocatable :: cdf(:), sigma(:), alat(:), zcw(:), adl(:), rdl(:)
   character(len=512) :: fin, fout
   call get_command_argument(1, fin); call get_command_argument(2, fout)
   open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
   read(u) ih, doy, n
   allocate(cdf(n), sigma(n), alat(n), zcw(n), adl(n), rdl(n))
   read(u) cdf, sigma, alat
   close(u)
   call geosrad_host_inhomogeneity(ih)             ! the host half of set_inhomogeneity (no device needed)
   if (condensate_inhomogeneous() .neqv. (ih > 0)) error stop 'condensate_inhomogeneous'
   do i = 1, n
      zcw(i) = zcw_lookup(cdf(i), sigma(i))
   end do
   call correlation_length_cloud_fraction(n, n, doy, alat, adl)
   call correlation_length_condensate(n, n, doy, alat, rdl)
   open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
   write(u) zcw, adl, rdl
   close(u)
end program hostfn_driver
