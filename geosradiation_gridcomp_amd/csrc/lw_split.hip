// lw_split.hip -- translation unit of the two-kernel RRTMG_LW band sweeps: k_lw_cells, k_lw_sweep (lw_split_kernels.hpp) and their
// launcher.  Built twice, like geosrad.hip: -DGEOSRAD_PART=4 instantiates the fp32 kernels, -DGEOSRAD_PART=8 the fp64 ones.
#include "lw_split_kernels.hpp"
#include "lw_split.hpp"

namespace geosrad {

template <typename R> hipError_t lw_split_launch(hipStream_t st, const LwArgs<R> &A, const LwDev<R> &T)
{
    const dim3 gc(band_grid(A.ncol, NB_LW * LWS_CHUNKS));
    hipLaunchKernelGGL((k_lw_cells<R, false>), gc, dim3(256), 0, st, A, T);
    hipLaunchKernelGGL((k_lw_cells<R, true>), gc, dim3(256), 0, st, A, T);
    const size_t lds = lw_bands_lds_bytes<R>();
    if (lds > 65536) {      // per-device attribute of the function: set on every launch (cheap; a second context on another GPU gets it too)
        hipError_t e = hipFuncSetAttribute((const void *)k_lw_sweep<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_lw_sweep<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    constexpr int B0 = lws_block<R, false>, B1 = lws_block<R, true>;
    hipLaunchKernelGGL((k_lw_sweep<R, false>), dim3((unsigned)((A.ncol + B0 - 1) / B0), NB_LW), dim3(B0), lds, st, A, T);
    hipLaunchKernelGGL((k_lw_sweep<R, true>), dim3((unsigned)((A.ncol + B1 - 1) / B1), NB_LW), dim3(B1), lds, st, A, T);
    return hipGetLastError();
}

#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 4
template hipError_t lw_split_launch<float>(hipStream_t, const LwArgs<float> &, const LwDev<float> &);
#endif
#if !defined(GEOSRAD_PART) || GEOSRAD_PART == 8
template hipError_t lw_split_launch<double>(hipStream_t, const LwArgs<double> &, const LwDev<double> &);
#endif

}  // namespace geosrad
