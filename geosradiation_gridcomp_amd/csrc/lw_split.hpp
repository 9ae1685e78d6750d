// lw_split.hpp -- host-side entry of the two-kernel RRTMG_LW band sweeps (lw_split_kernels.hpp), a translation unit of its own
// (lw_split.hip, built per precision like geosrad.hip).
#pragma once
#include "lw_kernels.hpp"

namespace geosrad {
// k_lw_cells (cloud-free and cloudy instantiation) then k_lw_sweep (both) on `st`; the band partials land in A.part as k_lw_bands leaves
// them (k_lw_reduce follows).  A.s1 / A.s2 are used in the split path's own cell layout; A.pfcode / A.pffs must be provided.
template <typename R> hipError_t lw_split_launch(hipStream_t st, const LwArgs<R> &A, const LwDev<R> &T);
}  // namespace geosrad
