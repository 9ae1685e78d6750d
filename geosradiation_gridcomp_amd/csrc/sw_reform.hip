// sw_reform.hip -- translation unit of the default RRTMG_SW band sweeps: k_sw_reform, k_swr_reduce (sw_reform_kernels.hpp) and their
// launchers.  Built twice, like geosrad.hip: -DGEOSRAD_PART=4 instantiates the fp32 kernels, -DGEOSRAD_PART=8 the fp64 ones.
#include "sw_reform_kernels.hpp"
#include "sw_reform.hpp"

namespace geosrad {

template <typename R> hipError_t sw_reform_launch(hipStream_t st, const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV)
{
    static_assert(SWR_NSLOT <= SWR_SLOTS_MAX && SWR_NCOT <= 6, "workspace slots");
    const dim3 grid(band_grid(A.ncol, SWR_NSLOT)), blk(256);
    hipLaunchKernelGGL((k_sw_reform<R, false>), grid, blk, 0, st, A, T, SV);
    hipLaunchKernelGGL((k_sw_reform<R, true>), grid, blk, 0, st, A, T, SV);
    return hipGetLastError();
}

template <typename R> hipError_t sw_reform_reduce(hipStream_t st, const SwArgs<R> &A, const SwOut<R> &O)
{
    hipLaunchKernelGGL(k_swr_reduce<R>, dim3((unsigned)((A.ncol + 255) / 256), A.nlay + 2), dim3(256), 0, st, A, O);
    return hipGetLastError();
}

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
template hipError_t sw_reform_launch<float>(hipStream_t, const SwArgs<float> &, const SwDev<float> &, const SwSolar<float> &);
template hipError_t sw_reform_reduce<float>(hipStream_t, const SwArgs<float> &, const SwOut<float> &);
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
template hipError_t sw_reform_launch<double>(hipStream_t, const SwArgs<double> &, const SwDev<double> &, const SwSolar<double> &);
template hipError_t sw_reform_reduce<double>(hipStream_t, const SwArgs<double> &, const SwOut<double> &);
#endif

}  // namespace geosrad
