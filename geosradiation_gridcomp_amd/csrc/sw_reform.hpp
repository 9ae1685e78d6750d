// sw_reform.hpp -- host-side entry of the default RRTMG_SW band sweeps (sw_reform_kernels.hpp), a translation unit of its own
// (sw_reform.hip, built per precision like geosrad.hip).
#pragma once
#include "sw_kernels.hpp"

namespace geosrad {
constexpr int SWR_SLOTS_MAX = 32;        // upper bound of the partial-flux slots per column (sw_reform_kernels.hpp swr_nslot<R> <= this)
// partial-flux slots per column of the precision's unit mapping (23 for fp32: units of <= 6 g-points; 32 for fp64: units of <= 4)
template <typename R> int sw_reform_nslot();
// lane = (column, unit of g-points), second sweep re-forming the cell optics: the cloud-free and the cloudy instantiation of k_sw_reform
// on `st`; partials per unit: sw_reform_reduce (k_swr_reduce) sums them into the caller's flux arrays and surface diagnostics
template <typename R> hipError_t sw_reform_launch(hipStream_t st, const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV);
template <typename R> hipError_t sw_reform_reduce(hipStream_t st, const SwArgs<R> &A, const SwOut<R> &O);
}  // namespace geosrad
