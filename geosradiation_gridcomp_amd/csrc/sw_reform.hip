// sw_reform.hip -- translation unit of the default RRTMG_SW band sweeps: k_sw_reform, k_swr_reduce (sw_reform_kernels.hpp) and their
// launchers.  Built twice, like geosrad.hip: -DGEOSRAD_PART=4 instantiates the fp32 kernels, -DGEOSRAD_PART=8 the fp64 ones.
#include "sw_reform_kernels.hpp"
#include "sw_reform.hpp"

namespace geosrad {

template <typename R> hipError_t sw_reform_launch(hipStream_t st, const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV)
{
    static_assert(swr_nslot<R> <= SWR_SLOTS_MAX && swr_ncot<R> <= 6, "workspace slots");
    const dim3 grid(band_grid(A.ncol, swr_nslot<R>)), blk(256);
    hipLaunchKernelGGL((k_sw_reform<R, false>), grid, blk, 0, st, A, T, SV);
    hipLaunchKernelGGL((k_sw_reform<R, true>), grid, blk, 0, st, A, T, SV);
    return hipGetLastError();
}

template <typename R> int sw_reform_nslot() { return swr_nslot<R>; }

template <typename R> hipError_t sw_reform_reduce(hipStream_t st, const SwArgs<R> &A, const SwOut<R> &O)
{
    hipLaunchKernelGGL(k_swr_reduce<R>, dim3((unsigned)((A.ncol + 255) / 256), A.nlay + 2), dim3(256), 0, st, A, O);
    return hipGetLastError();
}

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
template hipError_t sw_reform_launch<float>(hipStream_t, const SwArgs<float> &, const SwDev<float> &, const SwSolar<float> &);
template hipError_t sw_reform_reduce<float>(hipStream_t, const SwArgs<float> &, const SwOut<float> &);
template int sw_reform_nslot<float>();
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
template hipError_t sw_reform_launch<double>(hipStream_t, const SwArgs<double> &, const SwDev<double> &, const SwSolar<double> &);
template hipError_t sw_reform_reduce<double>(hipStream_t, const SwArgs<double> &, const SwOut<double> &);
template int sw_reform_nslot<double>();
#endif

}  // namespace geosrad
