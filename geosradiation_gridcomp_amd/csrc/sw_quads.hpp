// sw_quads.hpp -- host-side entry of the quad-mapped RRTMG_SW band sweeps (sw_quads_kernels.hpp), a translation unit of its own
// (sw_quads.hip, built per precision like geosrad.hip).
#pragma once
#include "sw_kernels.hpp"

namespace geosrad {
constexpr int SWQ_SLOTS = 32;        // partial-flux slots per column (sw_quads_kernels.hpp SWQ_NSLOT)
// the cloud-free and the cloudy instantiation of k_sw_quads on `st`, then nothing else: the caller launches the reduction
template <typename R> hipError_t sw_quads_launch(hipStream_t st, const SwArgs<R> &A, const SwDev<R> &T, const SwSolar<R> &SV);
// k_swq_reduce: slot sums -> the caller's flux arrays and surface diagnostics
template <typename R> hipError_t sw_quads_reduce(hipStream_t st, const SwArgs<R> &A, const SwOut<R> &O);
}  // namespace geosrad
