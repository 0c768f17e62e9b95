// gfx950 probe: what does a dependent kernel boundary cost behind B bytes of freshly written output, and do write-through (sc1) stores remove that part?
// The batch-1 forward is 79 dependent launches; a Winograd layer leaves 2.1 - 16.8 MB of output behind.  The guide prices a boundary at 1.7 - 1.9 us between
// streaming kernels "+ B / 6 TB/s when the predecessor leaves B bytes dirty" in its XCD's L2 (MI355X_MICROARCH.md, row `boundary`).  This program measures exactly
// that on a stand-in with the layers' shape: 512 workgroups x 256 threads (two per CU, one round), ~10 us of dependent arithmetic per thread, then every workgroup
// stores its 1/512 of B bytes as 16-byte stores in the LAST microseconds of the launch (as the Winograd epilogue does) --
//   plain       global_store_dwordx4                     (dirty lines stay in L2 until the end-of-kernel write-back)
//   write-thru  buffer_store_dwordx4 ... sc1             (what the tune key out_wt=1 makes wino3x3 / winoup3x3 do)
// -- chained L times in a hipGraph behind each other (layer l reads a few bytes of layer l - 1's output, so the chain is a true dependence), B = 0 .. 32 MB.
// Printed: microseconds per launch for both store kinds and their difference; B = 0 is the bare boundary + the arithmetic.
// Build here (hipcc --offload-arch=gfx950 -O3 -o tools/probes/boundary_probe tools/probes/boundary_probe.hip), run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static constexpr int kWG = 512, kThreads = 256;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool WT>
__global__ __launch_bounds__(256, 2) void layer(const float *in, float *out, unsigned bytes, int spin)
{
    const int tid = threadIdx.x;
    // a true dependence on the previous launch: one value of its output per workgroup (first bytes; always written, also at B = 0: see below)
    float a = in[blockIdx.x * 4] * 1e-9f + 1.0f;
    // stand-in arithmetic: a dependent chain, `spin` steps of 4 FMAs
    float b = (float)tid * 1e-6f;
    for (int i = 0; i < spin; ++i) { b = b * a + 1e-7f; b = b * a + 1e-7f; b = b * a + 1e-7f; b = b * a + 1e-7f; }
    // the epilogue: this workgroup's share of B bytes, coalesced 16-byte stores, 4 KB per workgroup and pass
    const unsigned per_wg = bytes / kWG;                       // multiple of 4096 (host)
    const float4 v = make_float4(b, b + 1.f, b + 2.f, b + 3.f);
    if (WT) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(bytes + 8192u), 0x00020000);
        for (unsigned o = 0; o < per_wg; o += 4096u)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, 8192u + blockIdx.x * per_wg + o + (unsigned)tid * 16u, 0, 16);
    } else {
        float4 *q = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + 8192u + (size_t)blockIdx.x * per_wg);
        for (unsigned o = 0; o < per_wg; o += 4096u) q[(o >> 4) + tid] = v;
    }
    if (tid == 0) out[blockIdx.x * 4] = b;                     // the 8 KB head the next launch reads from (plain in both arms)
}

template <bool WT>
static float chain_us(float *b0, float *b1, unsigned bytes, int spin, int L, hipStream_t s)
{
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(layer<WT>, dim3(kWG), dim3(kThreads), 0, s, (l & 1) ? b1 : b0, (l & 1) ? b0 : b1, bytes, spin);
    CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int rep = 0; rep < 14; ++rep) {
        float ms;
        CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
        CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) t.push_back(ms * 1e3f / L);
    }
    std::sort(t.begin(), t.end());
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    const int L = 24;
    const int spin = argc > 1 ? atoi(argv[1]) : 2500;          // ~10 us of dependent FMAs at ~2.3 GHz (4 cycles each)
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs; %d workgroups x %d threads per launch, %d launches per graph, spin %d\n", prop.name, prop.multiProcessorCount, kWG, kThreads, L, spin);
    const size_t cap = ((size_t)32 << 20) + 8192;
    float *b0, *b1; CHECK(hipMalloc(&b0, cap)); CHECK(hipMalloc(&b1, cap));
    CHECK(hipMemset(b0, 0, cap)); CHECK(hipMemset(b1, 0, cap));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    printf("%10s %12s %12s %10s %14s\n", "B (MB)", "plain us", "sc1 us", "delta us", "delta = B / x TB/s");
    for (unsigned mb2 : {0u, 2u, 4u, 8u, 16u, 32u, 64u}) {     // half megabytes: 0, 1, 2, 4, 8, 16, 32 MB
        const unsigned bytes = mb2 * (1u << 19);
        float p[2], w[2];
        for (int r = 0; r < 2; ++r) { p[r] = chain_us<false>(b0, b1, bytes, spin, L, s); w[r] = chain_us<true>(b0, b1, bytes, spin, L, s); }   // A-B-A-B
        const float pm = 0.5f * (p[0] + p[1]), wm = 0.5f * (w[0] + w[1]), d = pm - wm;
        printf("%10.1f %12.2f %12.2f %10.2f %14s\n", bytes / 1048576.0, pm, wm, d, bytes && d > 0.05f ? (std::to_string(bytes / d / 1e6f).substr(0, 5)).c_str() : "-");
    }
    return 0;
}
