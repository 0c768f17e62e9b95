// fp32 MFMA GEMM shared by the small dense stages around the renderer (csrc/a2h.hip, csrc/lle.hip), gfx950 only.
//   C[m][n] = act( acc * scale[n] + shift[n] + rowadd[m] ),  acc = sum_k A[m][k] * W[n][k]
// A [M][K] and W [N][K] row-major (nn.Linear / 1x1-conv weight layout); M and N ragged, K % 4 == 0 (rows 16-B aligned).
#pragma once
#include <hip/hip_runtime.h>

namespace lspgemm {

// Hand-off kernels (a2h_pipe, rnn_wave, rnn_layer: the workgroups of ONE launch poll each other's mailboxes) only make progress when all
// their working blocks are resident at once.  A launch-time check against the occupancy query is what hipLaunchCooperativeKernel buys
// (MI355X_MICROARCH.md, coop-launch row: +15-19 us per launch and the same residency as a plain launch), so it is done here once per kernel
// instead: working blocks <= resident capacity of the device, with one block per CU of margin where several fit (the API answer can be one
// block per CU high at some SGPR counts).  What a concurrent stream may occupy at run time is covered by the bounded polls + status word
// and the callers' retry (livespeechportraits_amd/a2h_engine.py, rnn_engine.py).
inline hipError_t fits_resident(const void *kernel, int threads, size_t dyn_lds, int working_blocks, bool *ok)
{
    int dev = 0, per_cu = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds);
    if (e != hipSuccess) return e;
    const long cap = (long)(per_cu > 1 ? per_cu - 1 : per_cu) * cus;
    *ok = cap >= working_blocks;
    return hipSuccess;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float *A;       // [M][K]
    const float *W;       // [N][K]
    const float *scale;   // [N] or null (= alpha)
    const float *shift;   // [N] or null (= 0)
    const float *rowadd;  // [M] or null (= 0)
    float *C;             // [M][N]
    int M, N, K;
    float alpha;          // used when scale == null
    int leaky;            // LeakyReLU(0.2) on the result
};

// 64x64 tile, 4 waves (2x2) of one 32x32 v_mfma_f32_32x32x2_f32 accumulator, K step 32.
static __global__ __launch_bounds__(256) void gemm_f32(GemmParams p)   // one copy per translation unit
{
    constexpr int LD = 36;
    __shared__ float As[64 * LD];
    __shared__ float Bs[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lrow = tid >> 3, lq = (tid & 7) * 4;   // staging: 32 rows x 8 float4 per pass, 2 passes
    for (int k0 = 0; k0 < p.K; k0 += 32) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = pass * 32 + lrow;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            const bool kin = k0 + lq < p.K;                  // K tail: zeros
            if (kin && m0 + r < p.M) a = *reinterpret_cast<const float4 *>(p.A + (size_t)(m0 + r) * p.K + k0 + lq);
            if (kin && n0 + r < p.N) b = *reinterpret_cast<const float4 *>(p.W + (size_t)(n0 + r) * p.K + k0 + lq);
            *reinterpret_cast<float4 *>(As + r * LD + lq) = a;
            *reinterpret_cast<float4 *>(Bs + r * LD + lq) = b;
        }
        __syncthreads();
        const float *ap = As + (wm * 32 + (lane & 31)) * LD + (lane >> 5);
        const float *bp = Bs + (wn * 32 + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int k = 0; k < 32; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= p.N) return;
    const float sc = p.scale ? p.scale[n] : p.alpha, sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[r] * sc + (p.rowadd ? p.rowadd[m] + sh : sh);
        if (p.leaky) v = v > 0.f ? v : 0.2f * v;
        p.C[(size_t)m * p.N + n] = v;
    }
}

// returns hipSuccess / the launch error; hipErrorInvalidValue if K % 4 != 0
static inline hipError_t launch_gemm_f32(const GemmParams &p, hipStream_t s)
{
    if (p.K % 4 || p.K < 4 || p.M < 1 || p.N < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gemm_f32, dim3((p.N + 63) / 64, (p.M + 63) / 64), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace lspgemm
