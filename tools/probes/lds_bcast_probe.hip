// gfx950 probe (round 5): can a vector-ALU last conv take its WEIGHTS as broadcast ds_read_b128 from LDS "nearly free beside the arithmetic"?
//
// The form in question: lane = low-res source pixel, 12 packed accumulators (4 sub-pixel parities x 3 channels), v_pk_fma_f32 only: per pixel and 4 input channels
// 96 v_pk_fma_f32 (384 issue cycles of its SIMD) against 48 wave-uniform ds_read_b128 of weights + 9 lane-distinct ones of activations.  A CU has FOUR SIMDs but ONE
// LDS pipe, so the question is what a ds_read_b128 costs that pipe when all 64 lanes read the SAME 16 bytes: if it is the full 1 KB return (8 cycles at 128 B/clk)
// the form is LDS-bound at 4 x 57 x 8 = 1 824 cycles per 384 of arithmetic; if a broadcast returns in ~1-2 cycles it is VALU-bound at 41 us for 8 frames.
// This program measures it on a stand-in with that instruction mix: every wave loops over batches of 8 ds_read_b128 (lgkmcnt counts to 15) (uniform address = broadcast, or lane * 16 =
// distinct) double-buffered against F v_pk_fma_f32 per read (F = 0, 2, 4, 8: pixels per lane P = F / 2), 4 or 8 waves per CU, every CU busy.
// Printed: shader cycles per ds_read_b128 of one wave, and the same per CU (/ waves per CU) -- the LDS pipe's cost per wave-wide read.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bcast_probe tools/probes/lds_bcast_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OFF> __device__ __forceinline__ void lds_read16(v4f &dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int F, bool BCAST>
__global__ void probe(float *out, unsigned long long *cycles, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = 1.0f + (float)(i & 255) * 1e-6f;      // 32 KB
    __syncthreads();
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned base = (unsigned)(unsigned long long)(lds_float *)lds;
    // broadcast: every lane the same address (a weight table walked in order); distinct: lane * 16 (an activation tile, conflict-free)
    unsigned addr = base + (BCAST ? 0u : (unsigned)lane * 16u);
    v2f acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = v2f{(float)k, (float)lane};
    v2f x[4] = {v2f{1.0f + lane * 1e-7f, 1.0f}, v2f{0.999f, 1.0f}, v2f{1.001f, 1.0f}, v2f{1.0f, 0.9999f}};
    v4f w[2][8];
    auto batch = [&](v4f (&d)[8], unsigned a) {
        lds_read16<0>(d[0], a); lds_read16<16>(d[1], a); lds_read16<32>(d[2], a); lds_read16<48>(d[3], a);
        lds_read16<1024>(d[4], a); lds_read16<1040>(d[5], a); lds_read16<1056>(d[6], a); lds_read16<1072>(d[7], a);
    };
    auto consume = [&](const v4f (&d)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const v2f lo = v2f{d[r].x, d[r].y}, hi = v2f{d[r].z, d[r].w};
            if constexpr (F == 0) {
                if ((r & 3) == 0) acc[r >> 2] += lo + hi;      // one add per four reads: the reads stay live
                else asm volatile("" :: "v"(d[r]));
            } else {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int k = (r * F + f) % 12;
                    acc[k] = __builtin_elementwise_fma(x[f & 3], (f & 1) ? hi : lo, acc[k]);
                }
            }
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    batch(w[0], addr);
    for (int it = 0; it < iters; it += 2) {
        batch(w[1], addr + 4096u);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        consume(w[0]);
        batch(w[0], addr);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        consume(w[1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) s += acc[k].x + acc[k].y;
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int F, bool BCAST>
static void run(int waves_per_cu, int cus, float *out, unsigned long long *cyc)
{
    const int iters = 512, threads = 256, blocks = cus * (waves_per_cu / 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<F, BCAST>), dim3(blocks), dim3(threads), 32768, 0, out, cyc, iters);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double reads = (double)(iters + 1) * 8;               // per wave
    const double med = (double)h[h.size() / 2], per_read = med / reads;
    printf("%-9s F=%d v_pk_fma_f32 per read, %d waves/CU: %7.2f cycles per ds_read_b128 of a wave = %5.2f per CU-wide read slot; arithmetic alone would be %5.1f\n",
           BCAST ? "broadcast" : "distinct", F, waves_per_cu, per_read, per_read / waves_per_cu, 4.0 * F);
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; per-wave cycles (s_memtime, 100 MHz-independent shader clock) over 4 104 ds_read_b128 per wave, median over all waves\n", prop.gcnArchName, cus);
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 8 * 8));
    for (int wpc : {4, 8}) {
        run<0, true>(wpc, cus, out, cyc); run<0, false>(wpc, cus, out, cyc);
        run<2, true>(wpc, cus, out, cyc); run<2, false>(wpc, cus, out, cyc);
        run<4, true>(wpc, cus, out, cyc); run<8, true>(wpc, cus, out, cyc); run<16, true>(wpc, cus, out, cyc);
    }
    return 0;
}
