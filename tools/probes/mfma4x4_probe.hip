// gfx950 probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4x1 products per instruction), and of the 4-byte LDS-DMA form.
// Build here (hipcc --offload-arch=gfx950), run on the GPU box; prints the layout it finds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out)
{
    const int lane = threadIdx.x;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // d1 = A(lane id) x B(1): D[i][j] of block b = A[i] of block b
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, z, 0, 0, 0);
    // d2 = A(1) x B(lane id)
    f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)lane, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = d1[r]; out[lane * 8 + 4 + r] = d2[r]; }
}
int main()
{
    float *d; hipMalloc(&d, 64 * 8 * sizeof(float));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    bool a_ok = true, b_ok = true;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            if (h[l * 8 + r] != (float)(4 * (l / 4) + r)) a_ok = false;        // register r = row i: the A value of lane 4*block + i
            if (h[l * 8 + 4 + r] != (float)l) b_ok = false;                    // lane = 4*block + column j: its own B value
        }
    printf("mfma_f32_4x4x1_16b: A lane = 4*block + row, B lane = 4*block + col, D[reg = row][lane = 4*block + col]: %s / %s\n", a_ok ? "A yes" : "A NO", b_ok ? "B yes" : "B NO");
    if (!a_ok || !b_ok)
        for (int l = 0; l < 64; ++l) printf("lane %2d: d1 %g %g %g %g   d2 %g %g %g %g\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    return 0;
}
