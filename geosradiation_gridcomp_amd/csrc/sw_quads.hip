// sw_quads.hip -- translation unit of the quad-mapped RRTMG_SW band sweeps: k_sw_quads, k_swq_reduce (sw_quads_kernels.hpp) and their
// launchers.  Built twice, like geosrad.hip: -DGEOSRAD_PART=4 instantiates the fp32 kernels, -DGEOSRAD_PART=8 the fp64 ones.
#include "sw_quads_kernels.hpp"
#include "sw_quads.hpp"

namespace geosrad {

static_assert(SWQ_SLOTS == SWQ_NSLOT, "slot count");

template <typename R> hipError_t sw_quads_launch(hipStream_t st, const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV)
{
#ifdef SWQ_GRID2D        // experiment: (column block, slot) = (x, y) order - every resident block runs the same quad body
    const dim3 grid((unsigned)((A.ncol + 255) / 256), SWQ_NBLK), blk(256);
#else
    const dim3 grid(band_grid(A.ncol, SWQ_NBLK)), blk(256);
#endif
    hipLaunchKernelGGL((k_sw_quads<R, false>), grid, blk, 0, st, A, T, SV);
    hipLaunchKernelGGL((k_sw_quads<R, true>), grid, blk, 0, st, A, T, SV);
    return hipGetLastError();
}

template <typename R> hipError_t sw_quads_reduce(hipStream_t st, const SwArgs<R> &A, const SwOut<R> &O)
{
    hipLaunchKernelGGL(k_swq_reduce<R>, dim3((unsigned)((A.ncol + 255) / 256), A.nlay + 2), dim3(256), 0, st, A, O);
    return hipGetLastError();
}

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
template hipError_t sw_quads_launch<float>(hipStream_t, const SwArgs<float> &, const SwDev<float> &, const SwSolar<float> &);
template hipError_t sw_quads_reduce<float>(hipStream_t, const SwArgs<float> &, const SwOut<float> &);
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
template hipError_t sw_quads_launch<double>(hipStream_t, const SwArgs<double> &, const SwDev<double> &, const SwSolar<double> &);
template hipError_t sw_quads_reduce<double>(hipStream_t, const SwArgs<double> &, const SwOut<double> &);
#endif

}  // namespace geosrad
