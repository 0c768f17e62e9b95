// Manifold projection of the APC features (include/lsplle.h): KNN + LLE reconstruction, gfx950 only.
//   row_norms     |x|^2 per row
//   gemm_f32      (gemm_f32.h) diss[i][j] = (|f_i|^2 + |b_j|^2) - 2 f_i . b_j          [n][m]
//   topk_rows     one workgroup per frame: per-thread sorted top-K in registers, then K rounds of block-wide argmin
//   lle_rows      one wave per frame: gather the K neighbours, Gram matrix + right-hand side by wave reduction,
//                 (K-1)x(K-1) LU with partial pivoting, double-precision reconstruction
#include "../../include/lsplle.h"

#include <hip/hip_runtime.h>

#include "gemm_f32.h"

#include <string>

namespace lsplle {

constexpr int MAXK = LSPLLE_MAX_K;

__global__ __launch_bounds__(256) void row_norms(const float *x, int rows, int d, float *out)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float *r = x + (size_t)wave * d;
    float a = 0.f;
    for (int c = lane; c < d; c += 64) a = fmaf(r[c], r[c], a);
    for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[wave] = a;
}

// ascending insertion of (v, j) into a sorted K-list held in registers (static indexing only)
template <int K> __device__ __forceinline__ void insert(float (&bv)[K], int (&bj)[K], float v, int j)
{
    if (!(v < bv[K - 1])) return;
    bv[K - 1] = v; bj[K - 1] = j;
#pragma unroll
    for (int i = K - 1; i > 0; --i) {
        if (bv[i] < bv[i - 1]) {
            const float tv = bv[i]; bv[i] = bv[i - 1]; bv[i - 1] = tv;
            const int tj = bj[i]; bj[i] = bj[i - 1]; bj[i - 1] = tj;
        }
    }
}

template <int K> __global__ __launch_bounds__(256) void topk_rows(const float *diss, int m, int k, long long *ind)
{
    __shared__ float sv[4];
    __shared__ int sj[4], st[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *r = diss + (size_t)row * m;
    float bv[K];
    int bj[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bv[i] = __builtin_inff(); bj[i] = 0x7fffffff; }
    for (int j = tid; j < m; j += 256) insert<K>(bv, bj, r[j], j);
    // K rounds: the block's smallest head wins (lowest index on ties), its owner pops
    for (int round = 0; round < k; ++round) {
        float v = bv[0];
        int j = bj[0], t = tid;
        for (int o = 32; o; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oj = __shfl_xor(j, o), ot = __shfl_xor(t, o);
            if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; t = ot; }
        }
        if (lane == 0) { sv[wave] = v; sj[wave] = j; st[wave] = t; }
        __syncthreads();
        v = sv[0]; j = sj[0]; t = st[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] < v || (sv[w] == v && sj[w] < j)) { v = sv[w]; j = sj[w]; t = st[w]; }
        // a row with fewer than k finite distances (NaN input) has no k-th neighbour: -1, which lle_rows refuses
        if (tid == 0) ind[(size_t)row * k + round] = j < m ? j : -1;
        if (tid == t) {
#pragma unroll
            for (int i = 0; i + 1 < K; ++i) { bv[i] = bv[i + 1]; bj[i] = bj[i + 1]; }
            bv[K - 1] = __builtin_inff(); bj[K - 1] = 0x7fffffff;
        }
        __syncthreads();
    }
}

struct LleParams {
    const float *feats, *db;
    const long long *ind;
    double *weights;
    float *fuse, *blend;
    int n, m, d, K;
    float percent;
};

__device__ __forceinline__ float wave_sum(float a)
{
    for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
    return a;
}

// 4 waves per block, one frame per wave.  LDS per wave: G [15][15], rhs [15], w double [16].
__global__ __launch_bounds__(256) void lle_rows(LleParams p)
{
    __shared__ float Gs[4][(MAXK - 1) * (MAXK - 1)];
    __shared__ float Bs[4][MAXK];
    __shared__ double Ws[4][MAXK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.n) return;
    const int K = p.K, n1 = K - 1;
    const float *f = p.feats + (size_t)row * p.d;
    const long long *id = p.ind + (size_t)row * K;
    // neighbour indices are caller data: a row holding one outside [0, m) (or topk_rows' -1) reads nothing and yields NaN
    {
        const bool bad = lane < K && (id[lane] < 0 || id[lane] >= (long long)p.m);
        if (__ballot(bad)) {
            const float qn = __builtin_nanf("");
            if (p.weights && lane < K) p.weights[(size_t)row * K + lane] = (double)qn;
            for (int c = lane; c < p.d; c += 64) {
                if (p.fuse) p.fuse[(size_t)row * p.d + c] = qn;
                if (p.blend) p.blend[(size_t)row * p.d + c] = qn;
            }
            return;
        }
    }
    const float *b0 = p.db + (size_t)id[0] * p.d;
    float *G = Gs[wave], *R = Bs[wave];
    double *W = Ws[wave];

    if (K == 1) {          // utils.py:143-145
        if (lane == 0) W[0] = 1.0;
    } else {
        // G = A^T A, rhs = A^T B with A columns fb_k - fb_0 and B = f - fb_0   (utils.py:148-151), fp32
        for (int a = 0; a < n1; ++a) {
            const float *ba = p.db + (size_t)id[a + 1] * p.d;
            float sb = 0.f;
            for (int c = lane; c < p.d; c += 64) sb = fmaf(ba[c] - b0[c], f[c] - b0[c], sb);
            sb = wave_sum(sb);
            if (lane == 0) R[a] = sb;
            for (int b = a; b < n1; ++b) {
                const float *bb = p.db + (size_t)id[b + 1] * p.d;
                float s = 0.f;
                for (int c = lane; c < p.d; c += 64) s = fmaf(ba[c] - b0[c], bb[c] - b0[c], s);
                s = wave_sum(s);
                if (lane == 0) { G[a * n1 + b] = s; G[b * n1 + a] = s; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {   // LU with partial pivoting + substitutions (what LAPACK gesv does), fp32
            for (int c = 0; c < n1; ++c) {
                int piv = c;
                float best = fabsf(G[c * n1 + c]);
                for (int r = c + 1; r < n1; ++r) { const float v = fabsf(G[r * n1 + c]); if (v > best) { best = v; piv = r; } }
                if (piv != c) {
                    for (int k = 0; k < n1; ++k) { const float t = G[c * n1 + k]; G[c * n1 + k] = G[piv * n1 + k]; G[piv * n1 + k] = t; }
                    const float t = R[c]; R[c] = R[piv]; R[piv] = t;
                }
                const float inv = 1.0f / G[c * n1 + c];
                for (int r = c + 1; r < n1; ++r) {
                    const float l = G[r * n1 + c] * inv;
                    for (int k = c + 1; k < n1; ++k) G[r * n1 + k] -= l * G[c * n1 + k];
                    R[r] -= l * R[c];
                }
            }
            for (int r = n1 - 1; r >= 0; --r) {
                float s = R[r];
                for (int k = r + 1; k < n1; ++k) s -= G[r * n1 + k] * R[k];
                R[r] = s / G[r * n1 + r];
            }
            double sum = 0.0;
            for (int k = 0; k < n1; ++k) { W[k + 1] = (double)R[k]; sum += (double)R[k]; }
            W[0] = 1.0 - sum;                              // utils.py:152
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (p.weights && lane < K) p.weights[(size_t)row * K + lane] = W[lane];
    // feat_fuse = w . feat_base in double (utils.py:153), stored as float32 (feat_fuse = zeros_like(feats), :173)
    for (int c = lane; c < p.d; c += 64) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += W[k] * (double)p.db[(size_t)id[k] * p.d + c];
        const float fu = K == 1 ? p.db[(size_t)id[0] * p.d + c] : (float)acc;
        if (p.fuse) p.fuse[(size_t)row * p.d + c] = fu;
        if (p.blend) p.blend[(size_t)row * p.d + c] = f[c] * (1.0f - p.percent) + fu * p.percent;   // demo.py:200
    }
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hipfail(hipError_t e, const char *what) { return fail(LSPLLE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace lsplle

using namespace lsplle;

extern "C" {

const char *lsplle_last_error(void) { return g_err.c_str(); }

size_t lsplle_knn_workspace_bytes(int n, int m)
{
    if (n < 1 || m < 1) return 0;
    return align256((size_t)n * m * 4) + align256((size_t)n * 4) + align256((size_t)m * 4);
}

int lsplle_knn(const float *feats_dev, int n, const float *db_dev, int m, int d, int K, int64_t *ind_dev,
               void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (!feats_dev || !db_dev || !ind_dev || !workspace_dev) return fail(LSPLLE_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || m < 1 || d < 4 || d % 4) return fail(LSPLLE_ERR_SHAPE, "need n, m >= 1 and d a multiple of 4");
    if (K < 1 || K > MAXK || K > m) return fail(LSPLLE_ERR_UNSUPPORTED, "K must be in 1..min(m, 16)");
    if (workspace_bytes < lsplle_knn_workspace_bytes(n, m)) return fail(LSPLLE_ERR_SHAPE, "workspace smaller than lsplle_knn_workspace_bytes()");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *w = static_cast<char *>(workspace_dev);
    float *diss = reinterpret_cast<float *>(w);
    float *fn = reinterpret_cast<float *>(w + align256((size_t)n * m * 4));
    float *dn = reinterpret_cast<float *>(w + align256((size_t)n * m * 4) + align256((size_t)n * 4));
    hipLaunchKernelGGL(row_norms, dim3((n + 3) / 4), dim3(256), 0, s, feats_dev, n, d, fn);
    hipLaunchKernelGGL(row_norms, dim3((m + 3) / 4), dim3(256), 0, s, db_dev, m, d, dn);
    // (|f|^2 + |b|^2) - 2 f.b^T : acc * (-2) + (rowadd + shift), one rounding like the reference's subtraction
    lspgemm::GemmParams g{feats_dev, db_dev, nullptr, dn, fn, diss, n, m, d, -2.0f, 0};
    hipError_t e = lspgemm::launch_gemm_f32(g, s);
    if (e != hipSuccess) return hipfail(e, "distance gemm launch");
    long long *ind = reinterpret_cast<long long *>(ind_dev);
    if (K <= 4) hipLaunchKernelGGL(topk_rows<4>, dim3(n), dim3(256), 0, s, diss, m, K, ind);
    else if (K <= 10) hipLaunchKernelGGL(topk_rows<10>, dim3(n), dim3(256), 0, s, diss, m, K, ind);
    else hipLaunchKernelGGL(topk_rows<16>, dim3(n), dim3(256), 0, s, diss, m, K, ind);
    e = hipGetLastError();
    return e == hipSuccess ? LSPLLE_OK : hipfail(e, "topk launch");
}

int lsplle_solve(const float *feats_dev, int n, const float *db_dev, int m, int d, const int64_t *ind_dev, int K,
                 double *weights_dev, float *fuse_dev, float *blend_dev, float percent, void *stream)
{
    if (!feats_dev || !db_dev || !ind_dev) return fail(LSPLLE_ERR_INVALID_ARGUMENT, "null argument");
    if (!weights_dev && !fuse_dev && !blend_dev) return fail(LSPLLE_ERR_INVALID_ARGUMENT, "no output requested");
    if (n < 1 || m < 1 || d < 1) return fail(LSPLLE_ERR_SHAPE, "need n, m, d >= 1");
    if (K < 1 || K > MAXK) return fail(LSPLLE_ERR_UNSUPPORTED, "K must be in 1..16");
    LleParams p{feats_dev, db_dev, reinterpret_cast<const long long *>(ind_dev), weights_dev, fuse_dev, blend_dev, n, m, d, K, percent};
    hipLaunchKernelGGL(lle_rows, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LSPLLE_OK : hipfail(e, "lle_rows launch");
}

}  // extern "C"
