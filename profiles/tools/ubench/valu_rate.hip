// valu_rate.hip -- issue rates of the fp32 vector instructions the SW two-stream is made of, on gfx950: v_fma_f32 against
// v_pk_fma_f32 (two fp32 FMAs per lane and instruction), and the transcendental unit (v_rcp_f32, v_exp_f32, v_sqrt_f32).
// Every CU runs `waves` wavefronts per SIMD of independent chains; prints wave-instructions per cycle and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, int iters, float b, float c)
{
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (v2f){a[i], a[i] + 0.5f}; }
    const v2f pb = {b, b}, pc = {c, c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
                if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);
                if (MODE == 4) a[i] = __builtin_amdgcn_sqrtf(a[i]);
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (MODE == 7) { a[i] = __builtin_fmaf(a[i], b, c); if (i % 4 == 0) a[i] = __builtin_amdgcn_rcpf(a[i]); }      // 4 FMA : 1 rcp
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int waves, float per_iter)
{
    const int blocks = 256 * waves;          // `waves` 256-thread blocks per CU = `waves` wavefronts per SIMD
    float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * per_iter;          // wave-instructions
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double cyc = ms * 1e-3 * clk * 1e3;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at %d MHz nominal)\n", name, waves, ms, cyc * 1024 / insts, clk / 1000);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, 32);
        run<1>("v_pk_fma_f32", w, 32);
        run<5>("v_pk_mul_f32", w, 32);
        run<6>("v_pk_add_f32", w, 32);
        run<2>("v_rcp_f32", w, 32);
        run<3>("v_exp_f32", w, 32);
        run<4>("v_sqrt_f32", w, 32);
        run<7>("4 v_fma : 1 v_rcp", w, 40);
    }
    return 0;
}
