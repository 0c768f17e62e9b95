// Head-pose generator (include/lspa2h.h): the autoregressive WaveNet of the reference's
// Audio2HeadposeModel.generate_sequences (models/audio2headpose_model.py:133-187), gfx950 only.
//
// Two kernels:
//   gemm_f32     (gemm_f32.h) fp32 MFMA GEMM with a per-column affine (+LeakyReLU) epilogue: the audio_downsample MLP
//                (models/audio2headpose.py:16-21, BatchNorm1d folded) and ALL layers' cond_filter/cond_gate
//                1x1 convs (models/networks.py:277-287) for every audio frame at once -- none of it depends
//                on the sampled poses, so it leaves the sequential loop.
//   a2h_stream   ONE persistent workgroup runs every time step of the loop: start convs, the gated residual
//                layers evaluated incrementally (per-layer dilation queues in LDS), end convs, GMM sampling,
//                feedback.  The step-to-step dependence goes through the 12 sampled values, so there is no
//                parallelism across steps or layers; the kernel is a chain of 256x256 / 384x128 mat-vecs
//                whose weights (6.4 MB per step) stream from L2 in kernel-specific packed layouts, one
//                coalesced 1-KB wave load per instruction, double-buffered in registers across the barriers.
#include "../../include/lspa2h.h"

#include <hip/hip_runtime.h>

#include "gemm_f32.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace lspa2h {

constexpr int RC = 128;        // residual == dilation channels
constexpr int SC = 256;        // skip channels
constexpr int NT = 512;        // threads of the stream workgroup
constexpr int MAX_LAYERS = 32;
constexpr int MAX_OUT = 64;    // (2*ndim+1)*ncenter
constexpr float LRELU = 0.2f;  // nn.LeakyReLU(0.2), networks.py:147

// ---------------------------------------------------------------------------------------------- stream
struct StreamParams {
    // packed weights (device blob)
    const float *start1_w, *start1_b;   // [128][ndim], [128]
    const float *blob; unsigned blob_bytes;  // packed matrices below are byte offsets into it
    unsigned start2_w; const float *start2_b;   // packed 8 x 512 float4, [128]
    unsigned fg_w;                      // per layer: packed 32 x 512 float4 (65536 floats)
    unsigned rs_w;                      // per layer: packed 24 x 512 float4 (49152 floats)
    const float *rs_b;                  // per layer: [384] residual bias (128) + skip bias (256)
    unsigned end1_w; const float *end1_b;       // packed 8 x 512 float4, [64]
    const float *end2_w, *end2_b;       // [nout][nout], [nout]
    // per-call tensors
    const float *proj;                  // [n_audio][layers*256]: cond projections + conv biases, packed row order
    const float *pre, *noise, *expq;
    float *out;
    int layers, ndim, ncenter, nout, loss;
    int field;                          // receptive field R
    int nframe, frame_future;
    float sigma_scale;
    int dil[MAX_LAYERS];                // dilation of layer l (power of two)
    int qoff[MAX_LAYERS];               // first queue row of layer l
    // layer-pipelined kernel only
    unsigned long long *xbox;           // [layers+1][nsteps][128] granules {epoch, value}: input of layer e / of the head
    unsigned long long *sbox;           // [layers][nframe][256] granules: skip_conv output of layer l for frame f
    unsigned *status;                   // 0 ok, else the code of the first hand-off that timed out
    unsigned epoch;                     // tag of this call (never 0)
    int stride;                         // only blocks with blockIdx.x % stride == 0 work (8: all on one XCD, for speed only)
};

template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over aligned groups of 4 / 8 lanes; every lane of the group gets the total
__device__ __forceinline__ float sum4(float v) { return dpp_add<0x4E>(dpp_add<0xB1>(v)); }          // quad_perm xor1, xor2
__device__ __forceinline__ float sum8(float v) { return dpp_add<0x141>(sum4(v)); }                    // + row_half_mirror

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU * v; }

// float4 loads [B, E) of a packed matrix that starts `off` bytes into the blob: instruction i reads float4
// (i*512 + tid) -> 1 KB contiguous per wave.  Buffer loads: one VGPR (tid*16) addresses all of them, the
// per-instruction part sits in an SGPR (64-bit global addresses would cost two VGPRs per load in flight).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int B, int E, int N>
__device__ __forceinline__ void load_packed(float4 (&w)[N], __amdgpu_buffer_rsrc_t blob, unsigned off, int voff)
{
#pragma unroll
    for (int i = B; i < E; ++i)
        w[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(blob, voff, (int)(off + (unsigned)i * NT * 16u), 0));
}
__device__ __forceinline__ float dot4(float4 w, float4 v, float acc)
{
    acc = fmaf(w.x, v.x, acc); acc = fmaf(w.y, v.y, acc); acc = fmaf(w.z, v.z, acc); acc = fmaf(w.w, v.w, acc);
    return acc;
}

__global__ __launch_bounds__(NT) void a2h_stream(StreamParams p)
{
    extern __shared__ float smem[];
    float *xbuf = smem;                  // [128] current layer input
    float *zbuf = xbuf + RC;             // [128] gated activation
    float *tbuf = zbuf + RC;             // [128] start_conv1 output
    float *sbuf = tbuf + RC;             // [256] lrelu(skip sum)
    float *r1 = sbuf + SC;               // [64]  end_conv_1 output (lrelu applied)
    float *r2 = r1 + MAX_OUT;            // [64]  end_conv_2 output
    float *inb = r2 + MAX_OUT;           // [16]  WaveNet input of this step (head pose)
    float *queue = inb + 16;             // [sum of dilations][128]
    const int tid = threadIdx.x;

    int qrows = 0;
    for (int l = 0; l < p.layers; ++l) qrows += p.dil[l];
    for (int i = tid; i < qrows * RC; i += NT) queue[i] = 0.f;
    if (tid < 16) inb[tid] = tid < p.ndim ? p.pre[tid] : 0.f;

    const int nsteps = p.field - 1 + p.nframe;
    const int fu = tid >> 3, fpart = tid & 7;        // fg item: channels fu and fu+64, 32-column part
    const int rq = tid >> 2, rpart = tid & 3;        // rs item: rows rq, 128+rq, 256+rq, 32-column part
    const unsigned fg_stride = 32 * NT * 16, rs_stride = 24 * NT * 16;   // bytes per layer
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void *)p.blob, 0, (int)p.blob_bytes, 0x00020000);
    const int voff = tid * 16;
    const int projN = p.layers * 256;

    // Weight registers.  F + R together are 87 % of the CU's register file, so the next mat-vec's weights
    // cannot all be in flight while the current set is live: half of the next set is requested before a
    // set is consumed, the other half as soon as its registers are free.
    float4 F[32], R[24];
    load_packed<0, 32>(F, blob, p.fg_w, voff);
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        int arow = s + p.frame_future - (p.field - 1);
        arow = arow < 0 ? 0 : arow;                  // the reference prepends field-1 copies of audio row 0
        const float *projrow = p.proj + (size_t)arow * projN;
        // ---- start convs (networks.py:198-199): 1x1, bias, LeakyReLU
        if (tid < RC) {
            float a = p.start1_b[tid];
            for (int k = 0; k < p.ndim; ++k) a = fmaf(p.start1_w[tid * p.ndim + k], inb[k], a);
            tbuf[tid] = lrelu(a);
        }
        __syncthreads();
        {
            float4 w[8];
            load_packed<0, 8>(w, blob, p.start2_w, voff);
            const float4 *v = reinterpret_cast<const float4 *>(tbuf + rpart * 32);
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) a = dot4(w[q], v[q], a);
            a = sum4(a);
            if (rpart == 0) xbuf[rq] = lrelu(a + p.start2_b[rq]);
        }
        float skip0 = 0.f, skip1 = 0.f;              // skip rows rq and 128+rq (leader lanes)
        __syncthreads();

        for (int l = 0; l < p.layers; ++l) {
            const int d = p.dil[l];
            float *qslot = queue + (size_t)(p.qoff[l] + (s & (d - 1))) * RC;   // holds x[t-d]; overwritten with x[t]
            const unsigned rsw = p.rs_w + (unsigned)l * rs_stride;
            load_packed<0, 16>(R, blob, rsw, voff);                                    // in flight during the fg mat-vec
            float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fpart == 0) pb = *reinterpret_cast<const float4 *>(projrow + l * 256 + fu * 4);
            // ---- filter/gate dilated convs + cond (networks.py:303-314): 256 rows x [x[t-d] ; x[t]]
            {
                const float4 *v = reinterpret_cast<const float4 *>((fpart < 4 ? qslot + fpart * 32 : xbuf + (fpart - 4) * 32));
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 x = v[q];
                    a0 = dot4(F[q], x, a0); a1 = dot4(F[8 + q], x, a1); a2 = dot4(F[16 + q], x, a2); a3 = dot4(F[24 + q], x, a3);
                }
                asm volatile("" ::: "memory");                  // keep the late half late: its registers are not free earlier
                load_packed<16, 24>(R, blob, rsw, voff);
                a0 = sum8(a0); a1 = sum8(a1); a2 = sum8(a2); a3 = sum8(a3);
                if (fpart == 0) {   // tanh(filter) * sigmoid(gate), networks.py:317-319
                    zbuf[fu] = tanhf(a0 + pb.x) * (1.f / (1.f + expf(-(a1 + pb.y))));
                    zbuf[fu + 64] = tanhf(a2 + pb.z) * (1.f / (1.f + expf(-(a3 + pb.w))));
                }
            }
            __syncthreads();
            // next fg weights (next layer, or layer 0 of the next step) stream in during the rs mat-vec
            const unsigned fgw = p.fg_w + (unsigned)(l + 1 == p.layers ? 0 : l + 1) * fg_stride;
            load_packed<0, 16>(F, blob, fgw, voff);
            // ---- residual + skip 1x1 convs (networks.py:322-323)
            {
                const float4 *v = reinterpret_cast<const float4 *>(zbuf + rpart * 32);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 z = v[q];
                    a0 = dot4(R[q], z, a0); a1 = dot4(R[8 + q], z, a1); a2 = dot4(R[16 + q], z, a2);
                }
                asm volatile("" ::: "memory");
                load_packed<16, 32>(F, blob, fgw, voff);
                a0 = sum4(a0); a1 = sum4(a1); a2 = sum4(a2);
                if (rpart == 0) {
                    const float *b = p.rs_b + l * 384;
                    const float x = xbuf[rq];
                    qslot[rq] = x;                       // every reader of x[t-d] is past the barrier above
                    xbuf[rq] = a0 + b[rq] + x;           // residual = residual_conv(x) + input
                    skip0 += a1 + b[128 + rq];
                    skip1 += a2 + b[256 + rq];
                }
            }
            __syncthreads();
        }

        const int frame = s - (p.field - 1);
        if (frame < 0) continue;                         // still filling the first receptive field
        // ---- end convs (networks.py:207-208) on the summed skips
        if (rpart == 0) { sbuf[rq] = lrelu(skip0); sbuf[128 + rq] = lrelu(skip1); }
        __syncthreads();
        {
            float4 w[8];
            load_packed<0, 8>(w, blob, p.end1_w, voff);
            const float4 *v = reinterpret_cast<const float4 *>(sbuf + fpart * 32);
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) a = dot4(w[q], v[q], a);
            a = sum8(a);
            if (fpart == 0) r1[fu] = lrelu(a + p.end1_b[fu]);
        }
        __syncthreads();
        if (tid < p.nout) {
            float a = p.end2_b[tid];
            for (int k = 0; k < p.nout; ++k) a = fmaf(p.end2_w[tid * p.nout + k], r1[k], a);
            r2[tid] = a;
        }
        __syncthreads();
        // ---- Sample_GMM (losses.py:68-112) / L2 passthrough
        if (tid < p.ndim) {
            float v;
            if (p.loss == LSPA2H_LOSS_L2) {
                v = r2[tid];
            } else {
                int idx = 0;
                if (p.ncenter > 1) {   // softmax -> prob / Exp(1) draw -> argmax  (torch.multinomial, one sample)
                    float mx = r2[0];
                    for (int k = 1; k < p.ncenter; ++k) mx = fmaxf(mx, r2[k]);
                    float den = 0.f;
                    for (int k = 0; k < p.ncenter; ++k) den += expf(r2[k] - mx);
                    float best = -1.f;
                    for (int k = 0; k < p.ncenter; ++k) {
                        const float val = (expf(r2[k] - mx) / den) / p.expq[(size_t)frame * p.ncenter + k];
                        if (val > best) { best = val; idx = k; }
                    }
                }
                const float mu = r2[p.ncenter + idx * p.ndim + tid];
                const float sigma = expf(-r2[p.ncenter + p.ncenter * p.ndim + idx * p.ndim + tid]) * p.sigma_scale;
                const float nz = p.noise ? p.noise[(size_t)frame * p.ndim + tid] : 0.f;
                v = nz * sigma + mu;
            }
            p.out[(size_t)frame * p.ndim + tid] = v;
            inb[tid] = v;                                // history_headpose <- cat(history[1:], pred), :186
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------- pipeline
// Layer-pipelined form of the same loop: block 0 is the head (start convs, end convs, sampling), block 1+l owns
// residual layer l with ALL its weights resident on the CU (fg: 128 VGPRs, residual + first skip half: 64 VGPRs,
// second skip half: 64 KB of LDS) next to its dilation queue, so nothing is streamed per step.  The 128-float
// activation hops from block to block through global memory as 8-byte {value, epoch} granules stored
// write-through and polled past L1 (sc1): the data is its own flag, no fence, no dependence on dispatch order or
// XCD placement (cdna_hip_programming.md Guideline 16, form R2).  Skip outputs do not ride the chain: each layer
// posts its 256 values to the head after it has forwarded the activation, and the head adds them in layer order
// (the reference's order) while it waits.  Every
// (edge, step) has its own slot, so there is no flow control and a producer can run arbitrarily far ahead --
// which is what happens while the first receptive field fills: those steps do not depend on any sample, the
// head posts them all at once and the layers work on different steps concurrently.
// All polls are bounded: a lost hand-off ends the kernel with a status code instead of hanging the GPU.
// Mailboxes are addressed as buffers (SGPR resource + SGPR slot offset + one VGPR lane offset): with 192 VGPRs
// holding weights there is no room for per-thread 64-bit pointers.  aux 16 = sc1: the store is written through
// and the load is served by L2/memory, never by this CU's L1 -- what a relaxed agent-scope atomic lowers to.
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int AUX_SC1 = 16;

// granule i of the slot that starts `slot` bytes into the box: ONE aligned 8-byte store {value, epoch}
__device__ __forceinline__ void put_granule(__amdgpu_buffer_rsrc_t box, unsigned slot, int i, unsigned epoch, float v)
{
    u32x2 g; g.x = __float_as_uint(v); g.y = epoch;
    __builtin_amdgcn_raw_buffer_store_b64(g, box, i * 8, (int)slot, AUX_SC1);
}
// every thread i < n waits for granule i of the slot; returns false for the whole block on timeout
__device__ __forceinline__ bool get_granules(__amdgpu_buffer_rsrc_t box, unsigned slot, int n, unsigned epoch, float *dst,
                                             unsigned *status, unsigned code)
{
    const int tid = threadIdx.x;
    bool ok = true;
    if (tid < n) {
        unsigned spins = 0;
        for (;;) {
            const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(box, tid * 8, (int)slot, AUX_SC1);
            asm volatile("" ::: "memory");               // a fresh load every pass
            if (g.y == epoch) { dst[tid] = __uint_as_float(g.x); break; }
            if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(status, RLX_AGENT) != 0)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicCAS(status, 0u, code);
    }
    return __syncthreads_and(ok);
}

__global__ __launch_bounds__(NT) void a2h_pipe(StreamParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xbuf = smem;                  // [128]
    float *zbuf = xbuf + RC;             // [128]
    float *tbuf = zbuf + RC;             // [128]
    float *sbuf = tbuf + RC;             // [256]
    float *r1 = sbuf + SC;               // [64]
    float *r2 = r1 + MAX_OUT;            // [64]
    float *inb = r2 + MAX_OUT;           // [16]
    float *queue = inb + 16;             // [dilation of this layer][128]
    const int tid = threadIdx.x;
    if (blockIdx.x % p.stride) return;
    const int block = blockIdx.x / p.stride;
    const int nsteps = p.field - 1 + p.nframe;
    const int fu = tid >> 3, fpart = tid & 7;
    const int rq = tid >> 2, rpart = tid & 3;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void *)p.blob, 0, (int)p.blob_bytes, 0x00020000);
    const int voff = tid * 16;
    const unsigned xedge = (unsigned)nsteps * RC * 8u, sedge = (unsigned)p.nframe * SC * 8u;   // bytes per edge
    const __amdgpu_buffer_rsrc_t xbox = __builtin_amdgcn_make_buffer_rsrc((void *)p.xbox, 0, (int)(xedge * (unsigned)(p.layers + 1)), 0x00020000);
    const __amdgpu_buffer_rsrc_t sbox = __builtin_amdgcn_make_buffer_rsrc((void *)p.sbox, 0, (int)(sedge * (unsigned)p.layers), 0x00020000);

    if (block == 0) {
        // ------------------------------------------------------------------ head
        float4 w2[8], we[8];
        load_packed<0, 8>(w2, blob, p.start2_w, voff);
        load_packed<0, 8>(we, blob, p.end1_w, voff);
        if (tid < 16) inb[tid] = tid < p.ndim ? p.pre[tid] : 0.f;
        __syncthreads();
        for (int frame = -1; frame < p.nframe; ++frame) {
            // frame -1: the constant input of the first `field` steps; frame f >= 0: sample f, then the input of step field+f
            if (frame >= 0) {
                const int s = p.field - 1 + frame;
                bool ok = true;
                if (tid < SC) {   // skip = sum over layers, in layer order (networks.py:203-204); layer l's values arrive ~one hop apart
                    float acc = 0.f;
                    for (int l = 0; l < p.layers && ok; ++l) {
                        const int slot = (int)((unsigned)l * sedge + (unsigned)frame * (SC * 8u));
                        for (unsigned spins = 0;;) {
                            const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(sbox, tid * 8, slot, AUX_SC1);
                            asm volatile("" ::: "memory");
                            if (g.y == p.epoch) { acc = l ? acc + __uint_as_float(g.x) : __uint_as_float(g.x); break; }
                            if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(p.status, RLX_AGENT) != 0)) { ok = false; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    if (!ok) atomicCAS(p.status, 0u, 0x100u + frame);
                    sbuf[tid] = lrelu(acc);
                }
                if (!__syncthreads_and(ok)) return;
                {   // end convs (networks.py:207-208)
                    const float4 *v = reinterpret_cast<const float4 *>(sbuf + fpart * 32);
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a = dot4(we[q], v[q], a);
                    a = sum8(a);
                    if (fpart == 0) r1[fu] = lrelu(a + p.end1_b[fu]);
                }
                __syncthreads();
                if (tid < p.nout) {
                    float a = p.end2_b[tid];
                    for (int k = 0; k < p.nout; ++k) a = fmaf(p.end2_w[tid * p.nout + k], r1[k], a);
                    r2[tid] = a;
                }
                __syncthreads();
                if (tid < p.ndim) {   // Sample_GMM (losses.py:68-112) / L2 passthrough
                    float v;
                    if (p.loss == LSPA2H_LOSS_L2) {
                        v = r2[tid];
                    } else {
                        int idx = 0;
                        if (p.ncenter > 1) {
                            float mx = r2[0];
                            for (int k = 1; k < p.ncenter; ++k) mx = fmaxf(mx, r2[k]);
                            float den = 0.f;
                            for (int k = 0; k < p.ncenter; ++k) den += expf(r2[k] - mx);
                            float best = -1.f;
                            for (int k = 0; k < p.ncenter; ++k) {
                                const float val = (expf(r2[k] - mx) / den) / p.expq[(size_t)frame * p.ncenter + k];
                                if (val > best) { best = val; idx = k; }
                            }
                        }
                        const float mu = r2[p.ncenter + idx * p.ndim + tid];
                        const float sigma = expf(-r2[p.ncenter + p.ncenter * p.ndim + idx * p.ndim + tid]) * p.sigma_scale;
                        const float nz = p.noise ? p.noise[(size_t)frame * p.ndim + tid] : 0.f;
                        v = nz * sigma + mu;
                    }
                    p.out[(size_t)frame * p.ndim + tid] = v;
                    inb[tid] = v;
                }
                __syncthreads();
                if (frame + 1 == p.nframe) break;
                (void)s;
            }
            // start convs (networks.py:198-199) on the pose in inb
            if (tid < RC) {
                float a = p.start1_b[tid];
                for (int k = 0; k < p.ndim; ++k) a = fmaf(p.start1_w[tid * p.ndim + k], inb[k], a);
                tbuf[tid] = lrelu(a);
            }
            __syncthreads();
            {
                const float4 *v = reinterpret_cast<const float4 *>(tbuf + rpart * 32);
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) a = dot4(w2[q], v[q], a);
                a = sum4(a);
                if (rpart == 0) {
                    const float x0 = lrelu(a + p.start2_b[rq]);
                    if (frame < 0) { for (int s = 0; s < p.field; ++s) if (s < nsteps) put_granule(xbox, (unsigned)s * (RC * 8u), rq, p.epoch, x0); }
                    else put_granule(xbox, (unsigned)(p.field + frame) * (RC * 8u), rq, p.epoch, x0);
                }
            }
            __syncthreads();
        }
        return;
    }

    // ------------------------------------------------------------------ layer l
    const int l = block - 1;
    const int d = p.dil[l];
    float4 F[32], R[16];
    load_packed<0, 32>(F, blob, p.fg_w + (unsigned)l * (32 * NT * 16), voff);
    load_packed<0, 16>(R, blob, p.rs_w + (unsigned)l * (24 * NT * 16), voff);
    float4 *r2lds = reinterpret_cast<float4 *>(queue + (size_t)d * RC);      // skip rows 128..255: [8][512] float4, same thread map
    {
        float4 t[8];
        load_packed<0, 8>(t, blob, p.rs_w + (unsigned)l * (24 * NT * 16) + 16u * NT * 16u, voff);
#pragma unroll
        for (int q = 0; q < 8; ++q) r2lds[q * NT + tid] = t[q];
    }
    for (int i = tid; i < d * RC; i += NT) queue[i] = 0.f;
    const float *rsb = p.rs_b + l * 384;
    const float br = rpart == 0 ? rsb[rq] : 0.f, bs0 = rpart == 0 ? rsb[128 + rq] : 0.f, bs1 = rpart == 0 ? rsb[256 + rq] : 0.f;
    const unsigned projN = (unsigned)p.layers * 256u * 4u;                                   // bytes per proj row
    const __amdgpu_buffer_rsrc_t projb = __builtin_amdgcn_make_buffer_rsrc((void *)p.proj, 0, 0x7fffffff, 0x00020000);
    const unsigned xin = (unsigned)l * xedge, xout = xin + xedge, sout = (unsigned)l * sedge;
    const bool last = l + 1 == p.layers;
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const int frame = s - (p.field - 1);             // >= 0: this step produces a sample, skips are live
        int arow = s + p.frame_future - (p.field - 1);
        arow = arow < 0 ? 0 : arow;
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fpart == 0)
            pb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(projb, fu * 16, (int)((unsigned)arow * projN + (unsigned)l * 1024u), 0));
        if (!get_granules(xbox, xin + (unsigned)s * (RC * 8u), RC, p.epoch, xbuf, p.status, 0x10000u * (l + 1) + s)) return;
        float *qslot = queue + (size_t)(s & (d - 1)) * RC;
        {   // filter/gate (networks.py:303-319)
            const float4 *v = reinterpret_cast<const float4 *>((fpart < 4 ? qslot + fpart * 32 : xbuf + (fpart - 4) * 32));
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = v[q];
                a0 = dot4(F[q], x, a0); a1 = dot4(F[8 + q], x, a1); a2 = dot4(F[16 + q], x, a2); a3 = dot4(F[24 + q], x, a3);
            }
            a0 = sum8(a0); a1 = sum8(a1); a2 = sum8(a2); a3 = sum8(a3);
            if (fpart == 0) {
                zbuf[fu] = tanhf(a0 + pb.x) * (1.f / (1.f + expf(-(a1 + pb.y))));
                zbuf[fu + 64] = tanhf(a2 + pb.z) * (1.f / (1.f + expf(-(a3 + pb.w))));
            }
        }
        __syncthreads();
        {   // residual + skip (networks.py:322-323); the skip rows only once samples are being produced
            const float4 *v = reinterpret_cast<const float4 *>(zbuf + rpart * 32);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) a0 = dot4(R[q], v[q], a0);
            a0 = sum4(a0);
            if (rpart == 0) {   // the next layer is waiting for this: forward it before anything else
                const float x = xbuf[rq];
                qslot[rq] = x;
                if (!last) put_granule(xbox, xout + (unsigned)s * (RC * 8u), rq, p.epoch, a0 + br + x);
            }
            if (frame >= 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float4 z = v[q]; a1 = dot4(R[8 + q], z, a1); a2 = dot4(r2lds[q * NT + tid], z, a2); }
                a1 = sum4(a1); a2 = sum4(a2);
                if (rpart == 0) {
                    put_granule(sbox, sout + (unsigned)frame * (SC * 8u), rq, p.epoch, a1 + bs0);
                    put_granule(sbox, sout + (unsigned)frame * (SC * 8u), 128 + rq, p.epoch, a2 + bs1);
                }
            }
        }
        __syncthreads();   // (dropping this barrier with a double-buffered xbuf measured 1 % slower)
    }
}

// Sample_GMM for independent rows (the LSTM decoder samples the whole sequence in one call)
__global__ __launch_bounds__(256) void gmm_sample_rows(const float *params, int rows, int nc, int nd, const float *noise,
                                                       const float *expq, float sigma_scale, float *out)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * nd) return;
    const int r = gid / nd, d = gid - r * nd;
    const float *g = params + (size_t)r * (2 * nd + 1) * nc;
    int idx = 0;
    if (nc > 1) {
        float mx = g[0];
        for (int k = 1; k < nc; ++k) mx = fmaxf(mx, g[k]);
        float den = 0.f;
        for (int k = 0; k < nc; ++k) den += expf(g[k] - mx);
        float best = -1.f;
        for (int k = 0; k < nc; ++k) {
            const float val = (expf(g[k] - mx) / den) / expq[(size_t)r * nc + k];
            if (val > best) { best = val; idx = k; }
        }
    }
    const float mu = g[nc + idx * nd + d];
    const float sigma = expf(-g[nc + nc * nd + idx * nd + d]) * sigma_scale;
    out[gid] = (noise ? noise[gid] : 0.f) * sigma + mu;
}

// ------------------------------------------------------------------------------------------------ host
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hipfail(hipError_t e, const char *what) { return fail(LSPA2H_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

struct TensorSlot {
    std::string key;
    std::vector<size_t> shape;
    size_t numel = 0;
    std::vector<float> data;
    bool set = false;
};

}  // namespace lspa2h

using namespace lspa2h;

struct lspa2h_handle {
    lspa2h_config cfg{};
    int layers = 0, nout = 0, field = 0;
    std::vector<int> dil, qoff;
    int qrows = 0;
    std::vector<TensorSlot> tensors;
    std::map<std::string, int> index;
    // blob offsets (floats)
    size_t o_mlp0_w, o_mlp0_scale, o_mlp0_shift, o_mlp1_w, o_mlp1_b, o_proj_w, o_proj_b;
    size_t o_start1_w, o_start1_b, o_start2_w, o_start2_b, o_fg_w, o_rs_w, o_rs_b, o_end1_w, o_end1_b, o_end2_w, o_end2_b;
    size_t blob_floats = 0;
    const float *blob = nullptr;
    size_t blob_bytes = 0;
    float *ws = nullptr;
    size_t ws_bytes = 0;
    bool packed = false;
    bool attr_done = false, pipe_attr_done = false, pipe_fits = false, boxes_clean = false;
    unsigned epoch = 0;
    int last_rows = 0;
    size_t xbox_bytes() const { return (size_t)(layers + 1) * (size_t)(field - 1 + cfg.max_audio_frames) * RC * 8; }
    size_t sbox_bytes() const { return (size_t)layers * (size_t)cfg.max_audio_frames * SC * 8; }

    void add(const std::string &key, std::vector<size_t> shape)
    {
        TensorSlot t;
        t.key = key; t.shape = shape; t.numel = 1;
        for (size_t s : shape) t.numel *= s;
        index[key] = (int)tensors.size();
        tensors.push_back(std::move(t));
    }
    const std::vector<float> &T(const std::string &key) const { return tensors[index.at(key)].data; }
};

static size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

extern "C" {

const char *lspa2h_last_error(void) { return g_err.c_str(); }
int lspa2h_abi_version(void) { return LSPA2H_ABI_VERSION; }

int lspa2h_create(const lspa2h_config *cfg, lspa2h_handle **out)
{
    if (!cfg || !out) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != LSPA2H_ABI_VERSION) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "abi_version mismatch");
    if (cfg->residual_channels != RC || cfg->dilation_channels != RC || cfg->skip_channels != SC || cfg->kernel_size != 2)
        return fail(LSPA2H_ERR_UNSUPPORTED, "kernels are built for residual/dilation 128, skip 256, kernel_size 2 (the reference defaults)");
    if (cfg->residual_layers < 1 || cfg->residual_layers > 10 || cfg->residual_blocks < 1 ||
        cfg->residual_layers * cfg->residual_blocks > MAX_LAYERS)
        return fail(LSPA2H_ERR_UNSUPPORTED, "residual_layers in 1..10 and layers*blocks <= 32");
    if (cfg->ndim < 1 || cfg->ndim > 16 || cfg->ncenter < 1 || cfg->ncenter > 8) return fail(LSPA2H_ERR_UNSUPPORTED, "ndim <= 16, ncenter <= 8");
    if (cfg->input_channels != cfg->ndim) return fail(LSPA2H_ERR_SHAPE, "input_channels must equal ndim (samples are fed back as history)");
    if (cfg->loss != LSPA2H_LOSS_GMM && cfg->loss != LSPA2H_LOSS_L2) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "loss");
    if (cfg->cond_channels != cfg->hidden_size || cfg->hidden_size % 64 || cfg->hidden_size < 64)
        return fail(LSPA2H_ERR_SHAPE, "cond_channels must equal hidden_size, a multiple of 64");
    if (cfg->max_audio_frames < 1 || cfg->max_audio_frames > 32768)    // mailbox offsets are 32-bit buffer offsets
        return fail(LSPA2H_ERR_INVALID_ARGUMENT, "max_audio_frames must be in 1..32768");
    lspa2h_handle *h = new (std::nothrow) lspa2h_handle;
    if (!h) return fail(LSPA2H_ERR_STATE, "out of host memory");
    h->cfg = *cfg;
    h->layers = cfg->residual_layers * cfg->residual_blocks;
    h->nout = cfg->loss == LSPA2H_LOSS_GMM ? (2 * cfg->ndim + 1) * cfg->ncenter : cfg->ndim;
    if (h->nout > MAX_OUT) { delete h; return fail(LSPA2H_ERR_UNSUPPORTED, "(2*ndim+1)*ncenter must be <= 64"); }
    h->field = 1;
    for (int b = 0; b < cfg->residual_blocks; ++b)
        for (int i = 0; i < cfg->residual_layers; ++i) {
            h->dil.push_back(1 << i);
            h->qoff.push_back(h->qrows);
            h->qrows += 1 << i;
            h->field += 1 << i;     // kernel_size 2: additional_scope doubles per layer (networks.py:150-166)
        }
    const size_t lds = (size_t)(RC * 3 + SC + 2 * MAX_OUT + 16 + (size_t)h->qrows * RC) * sizeof(float);
    if (lds > 160 * 1024) { delete h; return fail(LSPA2H_ERR_UNSUPPORTED, "dilation queues exceed the 160 KB LDS of one CU"); }

    const size_t H = cfg->hidden_size, nd = cfg->ndim, no = h->nout;
    h->add("audio_downsample.0.weight", {H, 2 * H});
    h->add("audio_downsample.0.bias", {H});
    h->add("audio_downsample.1.weight", {H});
    h->add("audio_downsample.1.bias", {H});
    h->add("audio_downsample.1.running_mean", {H});
    h->add("audio_downsample.1.running_var", {H});
    h->add("audio_downsample.3.weight", {H, H});
    h->add("audio_downsample.3.bias", {H});
    h->add("WaveNet.start_conv1.weight", {RC, nd, 1});
    h->add("WaveNet.start_conv1.bias", {RC});
    h->add("WaveNet.start_conv2.weight", {RC, RC, 1});
    h->add("WaveNet.start_conv2.bias", {RC});
    for (int l = 0; l < h->layers; ++l) {
        const std::string p = "WaveNet.residual_blocks." + std::to_string(l) + ".";
        h->add(p + "filter_conv.weight", {RC, RC, 2});
        h->add(p + "filter_conv.bias", {RC});
        h->add(p + "gate_conv.weight", {RC, RC, 2});
        h->add(p + "gate_conv.bias", {RC});
        h->add(p + "residual_conv.weight", {RC, RC, 1});
        h->add(p + "residual_conv.bias", {RC});
        h->add(p + "skip_conv.weight", {SC, RC, 1});
        h->add(p + "skip_conv.bias", {SC});
        h->add(p + "cond_filter_conv.weight", {RC, H, 1});
        h->add(p + "cond_filter_conv.bias", {RC});
        h->add(p + "cond_gate_conv.weight", {RC, H, 1});
        h->add(p + "cond_gate_conv.bias", {RC});
    }
    h->add("WaveNet.end_conv_1.weight", {no, SC, 1});
    h->add("WaveNet.end_conv_1.bias", {no});
    h->add("WaveNet.end_conv_2.weight", {no, no, 1});
    h->add("WaveNet.end_conv_2.bias", {no});

    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o = align64(o + n); return at; };
    const size_t L = h->layers;
    h->o_mlp0_w = take(H * 2 * H); h->o_mlp0_scale = take(H); h->o_mlp0_shift = take(H);
    h->o_mlp1_w = take(H * H); h->o_mlp1_b = take(H);
    h->o_proj_w = take(L * 256 * H); h->o_proj_b = take(L * 256);
    h->o_start1_w = take(RC * nd); h->o_start1_b = take(RC);
    h->o_start2_w = take(8 * NT * 4); h->o_start2_b = take(RC);
    h->o_fg_w = take(L * 32 * NT * 4); h->o_rs_w = take(L * 24 * NT * 4); h->o_rs_b = take(L * 384);
    h->o_end1_w = take(8 * NT * 4); h->o_end1_b = take(MAX_OUT);
    h->o_end2_w = take(no * no); h->o_end2_b = take(no);
    h->blob_floats = o;
    *out = h;
    return LSPA2H_OK;
}

int lspa2h_destroy(lspa2h_handle *h) { delete h; return LSPA2H_OK; }
int lspa2h_receptive_field(const lspa2h_handle *h) { return h ? h->field : fail(LSPA2H_ERR_INVALID_ARGUMENT, "null handle"); }
int lspa2h_num_tensors(const lspa2h_handle *h) { return h ? (int)h->tensors.size() : fail(LSPA2H_ERR_INVALID_ARGUMENT, "null handle"); }

int lspa2h_tensor_info(const lspa2h_handle *h, int index, const char **key, size_t *numel)
{
    if (!h || index < 0 || index >= (int)h->tensors.size()) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "tensor index out of range");
    if (key) *key = h->tensors[index].key.c_str();
    if (numel) *numel = h->tensors[index].numel;
    return LSPA2H_OK;
}

int lspa2h_set_tensor(lspa2h_handle *h, const char *key, const float *host_data, size_t numel)
{
    if (!h || !key || !host_data) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    auto it = h->index.find(key);
    if (it == h->index.end()) return fail(LSPA2H_ERR_INVALID_ARGUMENT, std::string("unknown tensor key: ") + key);
    TensorSlot &t = h->tensors[it->second];
    if (numel != t.numel) return fail(LSPA2H_ERR_SHAPE, std::string("wrong element count for ") + key);
    t.data.assign(host_data, host_data + numel);
    t.set = true;
    h->packed = false;
    return LSPA2H_OK;
}

size_t lspa2h_packed_bytes(const lspa2h_handle *h) { return h ? h->blob_floats * sizeof(float) : 0; }

int lspa2h_pack_weights(lspa2h_handle *h, void *host_dst, size_t bytes)
{
    if (!h || !host_dst) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->blob_floats * sizeof(float)) return fail(LSPA2H_ERR_SHAPE, "destination smaller than lspa2h_packed_bytes()");
    for (const TensorSlot &t : h->tensors)
        if (!t.set) return fail(LSPA2H_ERR_STATE, "tensor not set: " + t.key);   // the reference's strict=False would be silent
    float *d = static_cast<float *>(host_dst);
    std::memset(d, 0, h->blob_floats * sizeof(float));
    const int H = h->cfg.hidden_size, nd = h->cfg.ndim, no = h->nout, L = h->layers;

    std::memcpy(d + h->o_mlp0_w, h->T("audio_downsample.0.weight").data(), sizeof(float) * (size_t)H * 2 * H);
    {   // Linear bias + eval BatchNorm1d (eps 1e-5) folded: y = (xW^T) * s + ((b - mean) * s + beta)
        const auto &b = h->T("audio_downsample.0.bias"), &g = h->T("audio_downsample.1.weight"), &be = h->T("audio_downsample.1.bias");
        const auto &mu = h->T("audio_downsample.1.running_mean"), &var = h->T("audio_downsample.1.running_var");
        for (int i = 0; i < H; ++i) {
            const double s = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
            d[h->o_mlp0_scale + i] = (float)s;
            d[h->o_mlp0_shift + i] = (float)(((double)b[i] - (double)mu[i]) * s + (double)be[i]);
        }
    }
    std::memcpy(d + h->o_mlp1_w, h->T("audio_downsample.3.weight").data(), sizeof(float) * (size_t)H * H);
    std::memcpy(d + h->o_mlp1_b, h->T("audio_downsample.3.bias").data(), sizeof(float) * H);
    std::memcpy(d + h->o_start1_w, h->T("WaveNet.start_conv1.weight").data(), sizeof(float) * RC * nd);
    std::memcpy(d + h->o_start1_b, h->T("WaveNet.start_conv1.bias").data(), sizeof(float) * RC);
    std::memcpy(d + h->o_start2_b, h->T("WaveNet.start_conv2.bias").data(), sizeof(float) * RC);
    {   // start2: thread t = (row t>>2, part t&3), float4 q of its 32 columns at (q*512 + t)
        const auto &w = h->T("WaveNet.start_conv2.weight");
        for (int t = 0; t < NT; ++t)
            for (int q = 0; q < 8; ++q)
                for (int e = 0; e < 4; ++e)
                    d[h->o_start2_w + ((size_t)q * NT + t) * 4 + e] = w[(size_t)(t >> 2) * RC + (t & 3) * 32 + q * 4 + e];
    }
    for (int l = 0; l < L; ++l) {
        const std::string p = "WaveNet.residual_blocks." + std::to_string(l) + ".";
        const auto &fw = h->T(p + "filter_conv.weight"), &gw = h->T(p + "gate_conv.weight");
        const auto &fb = h->T(p + "filter_conv.bias"), &gb = h->T(p + "gate_conv.bias");
        const auto &cfw = h->T(p + "cond_filter_conv.weight"), &cgw = h->T(p + "cond_gate_conv.weight");
        const auto &cfb = h->T(p + "cond_filter_conv.bias"), &cgb = h->T(p + "cond_gate_conv.bias");
        // fg: thread t = (u = t>>3, part = t&7); item j: 0 filter[u], 1 gate[u], 2 filter[u+64], 3 gate[u+64];
        // column c of [x[t-d] ; x[t]]: c < 128 -> tap 0 (the zero-padded side, networks.py:303), else tap 1
        float *fg = d + h->o_fg_w + (size_t)l * 32 * NT * 4;
        for (int t = 0; t < NT; ++t)
            for (int j = 0; j < 4; ++j) {
                const int ch = (t >> 3) + (j >> 1) * 64;
                const auto &w = (j & 1) ? gw : fw;
                for (int q = 0; q < 8; ++q)
                    for (int e = 0; e < 4; ++e) {
                        const int c = (t & 7) * 32 + q * 4 + e;
                        const int tap = c >= RC, ci = c & (RC - 1);
                        fg[((size_t)(j * 8 + q) * NT + t) * 4 + e] = w[((size_t)ch * RC + ci) * 2 + tap];
                    }
            }
        // cond projection rows in the same (u, j) order; their bias carries both conv biases
        for (int u = 0; u < 64; ++u)
            for (int j = 0; j < 4; ++j) {
                const int ch = u + (j >> 1) * 64, row = l * 256 + u * 4 + j;
                const auto &w = (j & 1) ? cgw : cfw;
                std::memcpy(d + h->o_proj_w + (size_t)row * H, w.data() + (size_t)ch * H, sizeof(float) * H);
                d[h->o_proj_b + row] = (j & 1) ? cgb[ch] + gb[ch] : cfb[ch] + fb[ch];
            }
        // rs: thread t = (rq = t>>2, part = t&3); item j: row j*128 + rq of [residual_conv ; skip_conv]
        const auto &rw = h->T(p + "residual_conv.weight"), &sw = h->T(p + "skip_conv.weight");
        float *rs = d + h->o_rs_w + (size_t)l * 24 * NT * 4;
        for (int t = 0; t < NT; ++t)
            for (int j = 0; j < 3; ++j)
                for (int q = 0; q < 8; ++q)
                    for (int e = 0; e < 4; ++e) {
                        const int c = (t & 3) * 32 + q * 4 + e, r = t >> 2;
                        rs[((size_t)(j * 8 + q) * NT + t) * 4 + e] = j == 0 ? rw[(size_t)r * RC + c] : sw[(size_t)((j - 1) * 128 + r) * RC + c];
                    }
        std::memcpy(d + h->o_rs_b + (size_t)l * 384, h->T(p + "residual_conv.bias").data(), sizeof(float) * RC);
        std::memcpy(d + h->o_rs_b + (size_t)l * 384 + RC, h->T(p + "skip_conv.bias").data(), sizeof(float) * SC);
    }
    {   // end1: thread t = (row t>>3, part t&7); rows >= nout stay zero
        const auto &w = h->T("WaveNet.end_conv_1.weight");
        for (int t = 0; t < NT; ++t)
            if ((t >> 3) < no)
                for (int q = 0; q < 8; ++q)
                    for (int e = 0; e < 4; ++e)
                        d[h->o_end1_w + ((size_t)q * NT + t) * 4 + e] = w[(size_t)(t >> 3) * SC + (t & 7) * 32 + q * 4 + e];
        std::memcpy(d + h->o_end1_b, h->T("WaveNet.end_conv_1.bias").data(), sizeof(float) * no);
    }
    std::memcpy(d + h->o_end2_w, h->T("WaveNet.end_conv_2.weight").data(), sizeof(float) * (size_t)no * no);
    std::memcpy(d + h->o_end2_b, h->T("WaveNet.end_conv_2.bias").data(), sizeof(float) * no);
    h->packed = true;
    return LSPA2H_OK;
}

int lspa2h_bind_weights(lspa2h_handle *h, const void *packed_dev, size_t bytes)
{
    if (!h || !packed_dev) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->blob_floats * sizeof(float)) return fail(LSPA2H_ERR_SHAPE, "blob smaller than lspa2h_packed_bytes()");
    if ((uintptr_t)packed_dev & 15) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "blob must be 16-byte aligned");
    h->blob = static_cast<const float *>(packed_dev);
    h->blob_bytes = bytes;
    return LSPA2H_OK;
}

size_t lspa2h_workspace_bytes(const lspa2h_handle *h)
{
    if (!h) return 0;
    const size_t rows = h->cfg.max_audio_frames, H = h->cfg.hidden_size;
    return (align64(rows * H) * 2 + align64(rows * (size_t)h->layers * 256)) * sizeof(float) + h->xbox_bytes() + h->sbox_bytes() + 256;
}

int lspa2h_bind_workspace(lspa2h_handle *h, void *workspace_dev, size_t bytes)
{
    if (!h || !workspace_dev) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < lspa2h_workspace_bytes(h)) return fail(LSPA2H_ERR_SHAPE, "workspace smaller than lspa2h_workspace_bytes()");
    if ((uintptr_t)workspace_dev & 15) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "workspace must be 16-byte aligned");
    h->ws = static_cast<float *>(workspace_dev);
    h->ws_bytes = bytes;
    h->boxes_clean = false;
    return LSPA2H_OK;
}

static int launch_gemm(const float *A, const float *W, const float *scale, const float *shift, float *C, int M, int N, int K,
                       int leaky, hipStream_t s)
{
    lspgemm::GemmParams p{A, W, scale, shift, nullptr, C, M, N, K, 1.0f, leaky};
    const hipError_t e = lspgemm::launch_gemm_f32(p, s);
    return e == hipSuccess ? LSPA2H_OK : hipfail(e, "gemm_f32 launch");
}

static int generate_impl(lspa2h_handle *h, const float *audio_dev, int n_audio, const float *pre_dev, const float *noise_dev,
                         const float *expq_dev, float sigma_scale, int frame_future, float *out_dev, int nframe, hipStream_t s,
                         hipEvent_t mid)
{
    if (!h || !audio_dev || !pre_dev || !out_dev) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->blob) return fail(LSPA2H_ERR_STATE, "weights not bound (lspa2h_bind_weights)");
    if (!h->ws) return fail(LSPA2H_ERR_STATE, "workspace not bound (lspa2h_bind_workspace)");
    if (n_audio < 1 || n_audio > h->cfg.max_audio_frames) return fail(LSPA2H_ERR_SHAPE, "n_audio out of range (max_audio_frames)");
    if (frame_future < 0 || nframe < 1 || nframe != n_audio - frame_future)
        return fail(LSPA2H_ERR_SHAPE, "nframe must equal n_audio - frame_future and be >= 1");
    if (h->cfg.loss == LSPA2H_LOSS_GMM && h->cfg.ncenter > 1 && !expq_dev)
        return fail(LSPA2H_ERR_INVALID_ARGUMENT, "expq_dev is required when ncenter > 1");
    const int H = h->cfg.hidden_size, L = h->layers;
    const size_t rows = h->cfg.max_audio_frames;
    float *hid = h->ws, *cond = hid + align64(rows * H), *proj = cond + align64(rows * H);
    const float *b = h->blob;
    int rc;
    // audio_downsample: Linear(2H->H) + BatchNorm1d(eval) + LeakyReLU(0.2) + Linear(H->H)   (audio2headpose.py:16-21)
    if ((rc = launch_gemm(audio_dev, b + h->o_mlp0_w, b + h->o_mlp0_scale, b + h->o_mlp0_shift, hid, n_audio, H, 2 * H, 1, s))) return rc;
    if ((rc = launch_gemm(hid, b + h->o_mlp1_w, nullptr, b + h->o_mlp1_b, cond, n_audio, H, H, 0, s))) return rc;
    // every layer's cond_filter_conv / cond_gate_conv on every frame (networks.py:310-311)
    if ((rc = launch_gemm(cond, b + h->o_proj_w, nullptr, b + h->o_proj_b, proj, n_audio, L * 256, H, 0, s))) return rc;
    if (mid) (void)hipEventRecord(mid, s);
    h->last_rows = n_audio;

    StreamParams p{};
    p.start1_w = b + h->o_start1_w; p.start1_b = b + h->o_start1_b;
    p.blob = b; p.blob_bytes = (unsigned)(h->blob_floats * sizeof(float));
    p.start2_w = (unsigned)(h->o_start2_w * 4); p.start2_b = b + h->o_start2_b;
    p.fg_w = (unsigned)(h->o_fg_w * 4); p.rs_w = (unsigned)(h->o_rs_w * 4); p.rs_b = b + h->o_rs_b;
    p.end1_w = (unsigned)(h->o_end1_w * 4); p.end1_b = b + h->o_end1_b;
    p.end2_w = b + h->o_end2_w; p.end2_b = b + h->o_end2_b;
    p.proj = proj; p.pre = pre_dev; p.noise = noise_dev; p.expq = expq_dev; p.out = out_dev;
    p.layers = L; p.ndim = h->cfg.ndim; p.ncenter = h->cfg.ncenter; p.nout = h->nout; p.loss = h->cfg.loss;
    p.field = h->field; p.nframe = nframe; p.frame_future = frame_future; p.sigma_scale = sigma_scale;
    for (int l = 0; l < L; ++l) { p.dil[l] = h->dil[l]; p.qoff[l] = h->qoff[l]; }
    const bool single = (h->cfg.flags & LSPA2H_FLAG_SINGLE_WORKGROUP) != 0;
    char *tail = reinterpret_cast<char *>(proj + align64(rows * (size_t)L * 256));
    p.xbox = reinterpret_cast<unsigned long long *>(tail);
    p.sbox = reinterpret_cast<unsigned long long *>(tail + h->xbox_bytes());
    p.status = reinterpret_cast<unsigned *>(tail + h->xbox_bytes() + h->sbox_bytes());
    if (hipMemsetAsync(p.status, 0, 64, s) != hipSuccess) return fail(LSPA2H_ERR_HIP, "hipMemsetAsync(status)");
    if (!single) {
        // mailbox tags: the call counter, so slots of earlier calls never match; whatever the buffer held when it
        // was bound is cleared once
        if (!h->boxes_clean) {
            const hipError_t e = hipMemsetAsync(tail, 0, h->xbox_bytes() + h->sbox_bytes(), s);
            if (e != hipSuccess) return hipfail(e, "hipMemsetAsync(mailboxes)");
            h->boxes_clean = true;
        }
        if (++h->epoch == 0) h->epoch = 1;
        p.epoch = h->epoch;
        const int maxd = 1 << (h->cfg.residual_layers - 1);
        const size_t lds_pipe = (size_t)(RC * 3 + SC + 2 * MAX_OUT + 16 + (size_t)maxd * RC) * sizeof(float) + 8 * NT * 16;
        if (!h->pipe_attr_done) {
            // not the full 160 KB: __syncthreads_and keeps a word of static LDS
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&a2h_pipe),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe);
            if (e != hipSuccess) return hipfail(e, "hipFuncSetAttribute(a2h_pipe)");
            // the L + 1 working blocks poll each other: they must all fit the device at once (checked once per handle)
            bool ok = false;
            e = lspgemm::fits_resident(reinterpret_cast<const void *>(&a2h_pipe), NT, lds_pipe, L + 1, &ok);
            if (e != hipSuccess) return hipfail(e, "occupancy query (a2h_pipe)");
            h->pipe_fits = ok;
            h->pipe_attr_done = true;
        }
        if (!h->pipe_fits) goto single_workgroup;            // a device too small for the pipeline: the one-workgroup kernel needs no co-residency
        // Blocks are dealt round-robin over the 8 XCDs (observed, not contractual): using every 8th block puts the
        // whole chain behind one L2.  Measured 45.4 vs 51.0 us per frame; results do not depend on it.
        p.stride = (h->cfg.flags & LSPA2H_FLAG_CONSECUTIVE_BLOCKS) ? 1 : 8;     // (the flag: tools only)
        hipLaunchKernelGGL(a2h_pipe, dim3((L + 1) * p.stride), dim3(NT), lds_pipe, s, p);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? LSPA2H_OK : hipfail(e, "a2h_pipe launch");
    }
single_workgroup:
    const size_t lds = (size_t)(RC * 3 + SC + 2 * MAX_OUT + 16 + (size_t)h->qrows * RC) * sizeof(float);
    if (!h->attr_done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&a2h_stream),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hipfail(e, "hipFuncSetAttribute(a2h_stream)");
        h->attr_done = true;
    }
    hipLaunchKernelGGL(a2h_stream, dim3(1), dim3(NT), lds, s, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LSPA2H_OK : hipfail(e, "a2h_stream launch");
}

int lspa2h_generate(lspa2h_handle *h, const float *audio_dev, int n_audio, const float *pre_dev, const float *noise_dev,
                    const float *expq_dev, float sigma_scale, int frame_future, float *out_dev, int nframe, void *stream)
{
    return generate_impl(h, audio_dev, n_audio, pre_dev, noise_dev, expq_dev, sigma_scale, frame_future, out_dev, nframe,
                         static_cast<hipStream_t>(stream), nullptr);
}

int lspa2h_generate_timed(lspa2h_handle *h, const float *audio_dev, int n_audio, const float *pre_dev, const float *noise_dev,
                          const float *expq_dev, float sigma_scale, int frame_future, float *out_dev, int nframe, void *stream,
                          float *precompute_ms, float *loop_ms)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1, e2;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess)
        return fail(LSPA2H_ERR_HIP, "hipEventCreate");
    (void)hipEventRecord(e0, s);
    int rc = generate_impl(h, audio_dev, n_audio, pre_dev, noise_dev, expq_dev, sigma_scale, frame_future, out_dev, nframe, s, e1);
    if (rc == LSPA2H_OK) {
        (void)hipEventRecord(e2, s);
        const hipError_t e = hipEventSynchronize(e2);
        if (e != hipSuccess) rc = hipfail(e, "hipEventSynchronize");
        else {
            if (precompute_ms) (void)hipEventElapsedTime(precompute_ms, e0, e1);
            if (loop_ms) (void)hipEventElapsedTime(loop_ms, e1, e2);
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return rc;
}

int lspa2h_sample_gmm(const float *params_dev, int rows, int ncenter, int ndim, const float *noise_dev,
                      const float *expq_dev, float sigma_scale, float *out_dev, void *stream)
{
    if (!params_dev || !out_dev) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (rows < 1 || ncenter < 1 || ncenter > 8 || ndim < 1) return fail(LSPA2H_ERR_SHAPE, "rows >= 1, 1 <= ncenter <= 8, ndim >= 1");
    if (ncenter > 1 && !expq_dev) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "expq_dev is required when ncenter > 1");
    hipLaunchKernelGGL(gmm_sample_rows, dim3((rows * ndim + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       params_dev, rows, ncenter, ndim, noise_dev, expq_dev, sigma_scale, out_dev);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? LSPA2H_OK : hipfail(e, "gmm_sample_rows launch");
}

int lspa2h_status(lspa2h_handle *h, void *stream, uint32_t *code)
{
    if (!h || !code) return fail(LSPA2H_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->ws) return fail(LSPA2H_ERR_STATE, "workspace not bound");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize");
    const size_t rows = h->cfg.max_audio_frames, H = h->cfg.hidden_size;
    const char *tail = reinterpret_cast<const char *>(h->ws + align64(rows * H) * 2 + align64(rows * (size_t)h->layers * 256));
    e = hipMemcpy(code, tail + h->xbox_bytes() + h->sbox_bytes(), sizeof(uint32_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? LSPA2H_OK : hipfail(e, "hipMemcpy(status)");
}

int lspa2h_debug_cond(const lspa2h_handle *h, const float **cond_dev, int *rows, int *cols)
{
    if (!h || !h->ws) return fail(LSPA2H_ERR_STATE, "workspace not bound");
    if (cond_dev) *cond_dev = h->ws + align64((size_t)h->cfg.max_audio_frames * h->cfg.hidden_size);
    if (rows) *rows = h->last_rows;
    if (cols) *cols = h->cfg.hidden_size;
    return LSPA2H_OK;
}

}  // extern "C"
