// gfx950 probe: issue rate of v_mfma_f32_4x4x1_16b_f32 against v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 (one wave per SIMD, four independent
// accumulators each, the whole chip busy).  Prints time per instruction and FLOP/clk/SIMD at the clock the 32x32x2 run implies (64 cycles each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void spin(float *out, int iters)
{
    const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    float r = 0.f;
    if constexpr (KIND == 0) {
        f32x4 c[4] = {};
        for (int i = 0; i < iters; i += 8)
#pragma unroll
            for (int k = 0; k < 32; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k & 3], 0, 0, 0);
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (KIND == 1) {
        f32x4 c[4] = {};
        for (int i = 0; i < iters; i += 8)
#pragma unroll
            for (int k = 0; k < 32; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[k & 3], 0, 0, 0);
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        f32x16 c[4] = {};
        for (int i = 0; i < iters; i += 8)
#pragma unroll
            for (int k = 0; k < 32; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k & 3], 0, 0, 0);
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
    if (r == 12345.678f) out[0] = r;
}
template <int KIND> static float run(float *d, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.f;
}
int main()
{
    float *d; (void)hipMalloc(&d, 64);
    const int iters = 20000;                                  // x 4 instructions per wave
    const float t32 = run<2>(d, iters), t16 = run<1>(d, iters), t4 = run<0>(d, iters);
    const double clk = 64.0 * 4 * iters / (t32 * 1e-3);       // Hz, from the 64-cycle 32x32x2
    printf("per instruction: 32x32x2 %.1f ns (64 cycles by definition -> %.2f GHz), 16x16x4 %.1f ns = %.1f cycles, 4x4x1_16b %.1f ns = %.1f cycles\n",
           t32 * 1e6 / (4.0 * iters), clk * 1e-9, t16 * 1e6 / (4.0 * iters), t16 * 1e-3 / (4.0 * iters) * clk, t4 * 1e6 / (4.0 * iters), t4 * 1e-3 / (4.0 * iters) * clk);
    printf("FLOP/clk/SIMD: 32x32x2 %.1f, 16x16x4 %.1f, 4x4x1_16b %.1f\n", 4096.0 / 64.0, 2048.0 / (t16 * 1e-3 / (4.0 * iters) * clk), 512.0 / (t4 * 1e-3 / (4.0 * iters) * clk));
    return 0;
}
