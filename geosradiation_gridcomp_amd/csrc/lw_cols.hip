// lw_cols.hip -- translation unit of the on-chip RRTMG_LW band sweeps: k_lw_cols (lw_cols_kernels.hpp) and its launcher.
// Built twice, like geosrad.hip: -DGEOSRAD_PART=4 instantiates the fp32 kernels, -DGEOSRAD_PART=8 the fp64 ones.
#include "lw_cols_kernels.hpp"
#include "lw_cols.hpp"

namespace geosrad {

template <typename R, bool CLD, bool DBG, int C>
static hipError_t lwc_launch_one(hipStream_t st, const LwArgs<R> &A, const LwOut<R> &O, const LwDev<R> &T)
{
    const unsigned nwt = (unsigned)((C * A.nlay + 63) / 64 * 64);          // workers, whole wavefronts
    if (nwt + 64u > (unsigned)LWC_MAXT) return hipErrorInvalidValue;
    const unsigned ngroups = (unsigned)((A.ncol + C - 1) / C);
    const unsigned grid = 8u * ((ngroups + 7u) / 8u);
    const size_t lds = lwc_lds_bytes<R, CLD>(A.nlay, C);
    // the dynamic-LDS limit is a per-device attribute of the function: set on every launch (cheap), so that a second context on another
    // GPU of the same process gets it too
    hipError_t e = hipFuncSetAttribute((const void *)k_lw_cols<R, CLD, DBG, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_lw_cols<R, CLD, DBG, C>), dim3(grid), dim3(nwt + 64u), lds, st, A, O, T);
    return hipGetLastError();
}

template <typename R, int C> static hipError_t lwc_launch_c(hipStream_t st, const LwArgs<R> &A, const LwOut<R> &O, const LwDev<R> &T)
{
    hipError_t e = lwc_launch_one<R, false, false, C>(st, A, O, T);
    if (e != hipSuccess) return e;
    return lwc_launch_one<R, true, false, C>(st, A, O, T);
}

template <typename R> hipError_t lw_cols_launch(hipStream_t st, const LwArgs<R> &A, const LwOut<R> &O, const LwDev<R> &T, bool dbg)
{
    constexpr int CMAX = lwc_columns_per_block<R>(1), CMIN = lwc_columns_per_block<R>(203);
#ifndef LWC_FAST_BUILD
    if (dbg) return lwc_launch_one<R, true, true, CMIN>(st, A, O, T);          // test hook: any layer count, speed irrelevant
#endif
    const int C = lwc_columns_per_block<R>(A.nlay);
#ifdef LWC_FAST_BUILD          // kernel experiments (A/B builds): only the instantiations of <= 72 layers
    if (C != CMAX) return hipErrorInvalidValue;
    return lwc_launch_c<R, CMAX>(st, A, O, T);
#else
    if (C == CMAX) return lwc_launch_c<R, CMAX>(st, A, O, T);
    if (C == CMAX / 2) return lwc_launch_c<R, CMAX / 2>(st, A, O, T);
    if (C == CMIN) return lwc_launch_c<R, CMIN>(st, A, O, T);
    return lwc_launch_c<R, (CMAX / 4 > 0 ? CMAX / 4 : 1)>(st, A, O, T);
#endif
}

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
template hipError_t lw_cols_launch<float>(hipStream_t, const LwArgs<float> &, const LwOut<float> &, const LwDev<float> &, bool);
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
template hipError_t lw_cols_launch<double>(hipStream_t, const LwArgs<double> &, const LwOut<double> &, const LwDev<double> &, bool);
#endif

}  // namespace geosrad
