// lw_cols.hpp -- host-side entry of the on-chip RRTMG_LW band sweeps (lw_cols_kernels.hpp), compiled as a translation unit of its
// own (lw_cols.hip) so that the C-ABI layer does not have to re-instantiate the sixteen band bodies a third time.
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {
// launches the cloud-free and the cloudy instantiation (dbg: one launch over all columns that also dumps taug / pfracs) on `st`;
// A.perm / A.nclear / A.sc / A.scidx / A.pwvcm (and, for cloudy columns, A.taucmc / A.laycloudy) must have been produced on `st`
template <typename R> hipError_t lw_cols_launch(hipStream_t st, const LwArgs<R> &A, const LwOut<R> &O, const LwDev<R> &T, bool dbg);
}  // namespace geosrad
