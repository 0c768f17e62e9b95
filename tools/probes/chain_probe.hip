// gfx950 probe: what would ONE persistent launch buy on the weight-streaming tail of the generator?
// The tail's tiny-M layers (conv3x3_smallm: 512 -> 512 channels at 4x4 / 2x2, 14 in a row in `large`) are a dependent chain of launches, each one
//   256 workgroups x (read the whole previous tensor, <= 32 KB; stream its own 36 KB of the layer's 9.4 MB of weights; reduce; write 2 channels).
// This program runs exactly that traffic pattern with a stand-in for the arithmetic, two ways, and prints microseconds per layer:
//   (a) L launches in a hipGraph (one kernel boundary per layer: what the library does),
//   (b) ONE launch of 256 co-resident workgroups that walks the L layers with a grid-wide barrier between them (write-through sc1 stores, drain,
//       one relaxed agent-scope ticket per workgroup, relaxed polling, sc1 loads of the previous layer's tensor -- the guide's Guideline 16 hand-off),
//       with the NEXT layer's weight loads issued before the barrier wait (so they fly under it).
// Build here (hipcc --offload-arch=gfx950 -O3 -o tools/probes/chain_probe tools/probes/chain_probe.hip), run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static constexpr int kWG = 256, kThreads = 256, kC = 512;
static constexpr int kRowFloats = 9 * kC;                 // one output channel's weights: 4608 floats = 18 KB; a workgroup owns two rows

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the work of one layer for one workgroup; `coherent`: the input was written by other workgroups of THIS launch (read past the L1 / stale L2 lines, write through)
template <bool COHERENT>
__device__ __forceinline__ void layer_body(const float *in, float *out, const float4 (&w)[2][5], int npix, float *lds)
{
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, npix * kC * 4, 0x00020000);
    // whole previous tensor -> LDS (npix x 512 floats)
    for (int i = tid; i < npix * kC / 4; i += kThreads) {
        const u32x4 v = COHERENT ? __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)i * 16u, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)i * 16u, 0, 0);
        reinterpret_cast<u32x4 *>(lds)[i] = v;
    }
    __syncthreads();
    // stand-in arithmetic: this thread's K slice against every pixel (the real kernel: 9 taps x 512 channels / 256 threads = 5 float4 per row)
    float acc[2] = {0.f, 0.f};
    for (int p = 0; p < npix; ++p)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float4 a = reinterpret_cast<const float4 *>(lds)[(p * kC / 4 + tid + 256 * j) % (npix * kC / 4)];
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] += a.x * w[r][j].x + a.y * w[r][j].y + a.z * w[r][j].z + a.w * w[r][j].w;
        }
    __syncthreads();
    // block reduction through LDS, then npix x 2 outputs
    lds[tid] = acc[0]; lds[256 + tid] = acc[1];
    __syncthreads();
    if (tid < 2 * npix) {
        float s = 0.f;
        for (int i = 0; i < 256; ++i) s += lds[(tid & 1) * 256 + i];
        const int pix = tid >> 1, ch = blockIdx.x * 2 + (tid & 1);
        const float v = s * 1e-3f + (float)pix;
        if (COHERENT) {
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, npix * kC * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, (unsigned)(pix * kC + ch) * 4u, 0, 16);
        } else {
            out[pix * kC + ch] = v;
        }
    }
}

__device__ __forceinline__ void load_rows(const float *w, float4 (&dst)[2][5])
{
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k4 = threadIdx.x + 256 * j;
            dst[r][j] = k4 < kRowFloats / 4 ? reinterpret_cast<const float4 *>(w + (size_t)(blockIdx.x * 2 + r) * kRowFloats)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
}

__global__ __launch_bounds__(256) void one_layer(const float *in, float *out, const float *w, int npix)
{
    extern __shared__ float lds[];
    float4 wv[2][5];
    load_rows(w, wv);
    layer_body<false>(in, out, wv, npix, lds);
}

// (b): every workgroup walks all layers; buf[l & 1] -> buf[(l + 1) & 1]; counter l counts the workgroups that have finished layer l
__global__ __launch_bounds__(256) void chain(float *buf0, float *buf1, const float *w, size_t wstride, int layers, int npix, unsigned *cnt, unsigned *timeout)
{
    extern __shared__ float lds[];
    float4 wv[2][5];
    load_rows(w, wv);
    for (int l = 0; l < layers; ++l) {
        float *in = (l & 1) ? buf1 : buf0, *out = (l & 1) ? buf0 : buf1;
        if (l == 0) layer_body<false>(in, out, wv, npix, lds);
        else layer_body<true>(in, out, wv, npix, lds);
        if (l + 1 == layers) break;
        // publish: write-through stores above -> every wave drains -> barrier -> one relaxed agent-scope ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the next layer's weights do not depend on anybody: requested before the wait, they fly under it
        load_rows(w + (size_t)(l + 1) * wstride, wv);
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)gridDim.x) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *timeout = 1u; break; }
            }
        }
        __syncthreads();
    }
}

int main()
{
    const int L = 14;
    int dev = 0; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
    printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const size_t wlayer = (size_t)kC * kRowFloats;          // floats per layer: 9.4 MB
    float *w, *b0, *b1; unsigned *cnt, *tmo;
    CHECK(hipMalloc(&w, (size_t)L * wlayer * 4)); CHECK(hipMalloc(&b0, 16 * kC * 4)); CHECK(hipMalloc(&b1, 16 * kC * 4));
    CHECK(hipMalloc(&cnt, 64 * 4)); CHECK(hipMalloc(&tmo, 4));
    CHECK(hipMemset(w, 0, (size_t)L * wlayer * 4)); CHECK(hipMemset(b0, 0, 16 * kC * 4)); CHECK(hipMemset(b1, 0, 16 * kC * 4)); CHECK(hipMemset(tmo, 0, 4));
    float *flush; const size_t flush_bytes = (size_t)600 << 20;      // between replays: > the 256 MB memory-side cache, so that every layer's weights are cold (as in the forward)
    CHECK(hipMalloc(&flush, flush_bytes));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, chain, kThreads, 33 * 1024));
    printf("chain kernel: %d workgroup(s) per CU by the occupancy query -> %d resident of %d needed\n", occ, occ * prop.multiProcessorCount, kWG);
    if (occ * prop.multiProcessorCount < kWG) { printf("not co-resident: no measurement\n"); return 0; }
    for (int npix : {4, 16}) {
        const size_t smem = (size_t)std::max(npix * kC * 4, 2048);
        // (a) graph of L launches
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(one_layer, dim3(kWG), dim3(kThreads), smem, s, (l & 1) ? b1 : b0, (l & 1) ? b0 : b1, w + (size_t)l * wlayer, npix);
        CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        std::vector<float> ta, tb;
        for (int rep = 0; rep < 12; ++rep) {
            float ms;
            CHECK(hipMemsetAsync(flush, rep, flush_bytes, s));
            CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) ta.push_back(ms * 1e3f / L);
            CHECK(hipMemsetAsync(flush, rep + 1, flush_bytes, s));
            CHECK(hipMemsetAsync(cnt, 0, 64 * 4, s));
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(chain, dim3(kWG), dim3(kThreads), smem, s, b0, b1, w, wlayer, L, npix, cnt, tmo);
            CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) tb.push_back(ms * 1e3f / L);
        }
        unsigned t = 0; CHECK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("%2d pixels (%s level): graph of %d launches %.2f us per layer (min %.2f) | one launch + grid barriers %.2f us per layer (min %.2f)%s\n", npix, npix == 4 ? "2x2" : "4x4", L,
               ta[ta.size() / 2], ta[0], tb[tb.size() / 2], tb[0], t ? "  [BARRIER TIMEOUT]" : "");
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
    // the same with warm weights (no flush): the boundary / barrier cost alone
    for (int npix : {4, 16}) {
        const size_t smem = (size_t)std::max(npix * kC * 4, 2048);
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; ++l)
            hipLaunchKernelGGL(one_layer, dim3(kWG), dim3(kThreads), smem, s, (l & 1) ? b1 : b0, (l & 1) ? b0 : b1, w, npix);      // one layer's weights for all: warm
        CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        std::vector<float> ta, tb;
        for (int rep = 0; rep < 22; ++rep) {
            float ms;
            CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) ta.push_back(ms * 1e3f / L);
            CHECK(hipMemsetAsync(cnt, 0, 64 * 4, s));
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(chain, dim3(kWG), dim3(kThreads), smem, s, b0, b1, w, (size_t)0, L, npix, cnt, tmo);
            CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) tb.push_back(ms * 1e3f / L);
        }
        std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
        printf("%2d pixels, weights warm: graph %.2f us per layer (min %.2f) | one launch + grid barriers %.2f us per layer (min %.2f)\n", npix, ta[ta.size() / 2], ta[0], tb[tb.size() / 2], tb[0]);
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
    return 0;
}
