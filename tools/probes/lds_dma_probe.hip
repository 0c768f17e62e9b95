// Probe: semantics of `buffer_load_dwordx4 ... lds` on gfx950 (layout of the 1-KB wave write, and
// what an out-of-range lane writes).  Build: hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* p, float* o, int nfloats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = -7.f;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nfloats * 4, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane L loads quad (63-L) of the source: reversal shows the lane->LDS mapping; lanes 5, 40 are OOB
    unsigned vo = (lane == 5 || lane == 40) ? 0x80000000u : (unsigned)(63 - lane) * 16u + wave * 1024u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, vo, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) o[i] = sm[i];
}
int main() {
    const int n = 512;
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, d, o, n);
    std::vector<float> r(n); hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 2; ++w) for (int L = 0; L < 64; ++L) for (int e = 0; e < 4; ++e) {
        float got = r[w * 256 + L * 4 + e];
        float exp = (L == 5 || L == 40) ? 0.f : (float)(w * 256 + (63 - L) * 4 + e);
        if (got != exp) { if (bad < 8) printf("wave %d lane %d e %d: got %g expected %g\n", w, L, e, got, exp); ++bad; }
    }
    printf("lds_dma_probe: %s (%d mismatches): LDS[lane*16] <- src[voffset], OOB lanes write 0\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
