// Recurrent stacks of the audio front-end (include/lsprnn.h): multi-layer GRU / LSTM over one sequence, gfx950 only.
//
// Per layer: one gemm_f32 launch computes W_ih x_t + biases for every step (no recurrence in it), then one launch of
// rnn_layer runs the recurrence.  rnn_layer splits the hidden units over H*P/512 workgroups (P = H/32): a workgroup
// keeps the W_hh rows of its 512/P units in registers for the whole sequence (96 VGPRs per thread for a GRU, 128 for
// an LSTM) and, every step, all-gathers h_{t-1} from the others through 8-byte {value, epoch} granules (write-through
// stores, polls past L1, one slot per step: cdna_hip_programming.md Guideline 16 form R2 -- the same hand-off as
// csrc/a2h.hip).  The chain is latency-bound: one poll round trip + a 32-column mat-vec slice per step.
#include "../../include/lsprnn.h"

#include <hip/hip_runtime.h>

#include "gemm_f32.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace lsprnn {

constexpr int NT = 512;
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int AUX_SC1 = 16;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct LayerParams {
    const float *blob; unsigned blob_bytes;
    unsigned whh;                  // byte offset of this layer's packed W_hh: [workgroup][gate][8][512] float4
    const float *bhn;              // GRU: b_hn [H] (stays inside r * (...)); LSTM: unused
    const float *xproj;            // [T][GATES*H]: W_ih x_t + b_ih (+ b_hh for every gate except the GRU's n)
    float *hseq;                   // [T][H] output of this layer
    unsigned long long *hbox;      // [T][H] granules
    unsigned *status;
    unsigned epoch;
    int T, H, stride;
};

template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over aligned groups of P lanes (8, 16 or 32), every lane gets the total
template <int P> __device__ __forceinline__ float sum_p(float v)
{
    v = dpp_add<0x4E>(dpp_add<0xB1>(v));        // quad_perm xor 1, xor 2
    v = dpp_add<0x141>(v);                      // row_half_mirror: + the other quad of the 8
    if (P >= 16) v = dpp_add<0x140>(v);         // row_mirror: + the other half of the 16
    if (P == 32) v += __shfl_xor(v, 16);        // the neighbouring DPP row
    return v;
}
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float dot4(float4 w, float4 v, float acc)
{
    acc = fmaf(w.x, v.x, acc); acc = fmaf(w.y, v.y, acc); acc = fmaf(w.z, v.z, acc); acc = fmaf(w.w, v.w, acc);
    return acc;
}

// GATES = 3 (GRU) or 4 (LSTM); P = H / 32 column parts (8 or 16); a workgroup owns U = 512 / P hidden units
template <int GATES, int P> __global__ __launch_bounds__(NT) void rnn_layer(LayerParams p)
{
    __shared__ __attribute__((aligned(16))) float hbuf[512];
    if (blockIdx.x % p.stride) return;
    const int wg = blockIdx.x / p.stride;
    constexpr int U = NT / P;
    const int tid = threadIdx.x, part = tid % P, ul = tid / P;
    const int unit = wg * U + ul;
    const int H = p.H;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void *)p.blob, 0, (int)p.blob_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t box = __builtin_amdgcn_make_buffer_rsrc((void *)p.hbox, 0, (int)((unsigned)p.T * (unsigned)H * 8u), 0x00020000);
    float4 W[GATES * 8];
#pragma unroll
    for (int i = 0; i < GATES * 8; ++i)
        W[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            blob, tid * 16, (int)(p.whh + ((unsigned)wg * GATES * 8 + (unsigned)i) * NT * 16u), 0));
    const float bhn = (GATES == 3 && part == 0) ? p.bhn[unit] : 0.f;
    float hprev = 0.f, cprev = 0.f;                       // leader lanes: own unit's state
    for (int t = 0; t < p.T; ++t) {
        float xg[GATES];
        if (part == 0) {
#pragma unroll
            for (int g = 0; g < GATES; ++g) xg[g] = p.xproj[((size_t)t * GATES + g) * H + unit];
        }
        bool ok = true;
        if (t == 0) {
            if (tid < H) hbuf[tid] = 0.f;                   // zero initial state
        } else if (tid < H) {
            const int slot = (int)((unsigned)(t - 1) * (unsigned)H * 8u);
            for (unsigned spins = 0;;) {
                const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(box, tid * 8, slot, AUX_SC1);
                asm volatile("" ::: "memory");
                if (g.y == p.epoch) { hbuf[tid] = __uint_as_float(g.x); break; }
                if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(p.status, RLX_AGENT) != 0)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicCAS(p.status, 0u, 0x1000000u + (unsigned)t);
        }
        if (!__syncthreads_and(ok)) return;
        float a[GATES];
        {
            const float4 *v = reinterpret_cast<const float4 *>(hbuf + part * 32);
#pragma unroll
            for (int g = 0; g < GATES; ++g) a[g] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = v[q];
#pragma unroll
                for (int g = 0; g < GATES; ++g) a[g] = dot4(W[g * 8 + q], x, a[g]);
            }
#pragma unroll
            for (int g = 0; g < GATES; ++g) a[g] = sum_p<P>(a[g]);
        }
        if (part == 0) {
            float hn;
            if (GATES == 3) {
                const float r = sigmoidf(xg[0] + a[0]);
                const float z = sigmoidf(xg[1] + a[1]);
                const float n = tanhf(xg[2] + r * (a[2] + bhn));
                hn = (1.f - z) * n + z * hprev;
            } else {
                const float i = sigmoidf(xg[0] + a[0]);
                const float f = sigmoidf(xg[1] + a[1]);
                const float g = tanhf(xg[2] + a[2]);
                const float o = sigmoidf(xg[GATES - 1] + a[GATES - 1]);
                cprev = f * cprev + i * g;
                hn = o * tanhf(cprev);
            }
            hprev = hn;
            u32x2 gr; gr.x = __float_as_uint(hn); gr.y = p.epoch;
            if (t + 1 < p.T)
                __builtin_amdgcn_raw_buffer_store_b64(gr, box, unit * 8, (int)((unsigned)t * (unsigned)H * 8u), AUX_SC1);
            p.hseq[(size_t)t * H + unit] = hn;
        }
        __syncthreads();
    }
}

// Wavefront form: every layer of the stack runs in the same launch, layer l working on step t while layer l-1 is already
// on a later step.  A workgroup of layer l >= 1 also keeps the W_ih rows of its units resident (its input is the layer
// below's h_t, known only step by step) and polls two vectors per step: its own layer's h_{t-1} and the layer below's
// h_t.  Layer 0 still takes its input projection from the gemm (x is known for all steps).  Two matrices per thread,
// so the columns are cut into parts of 16: P = H / 16 lanes share a unit (48 + 48 VGPRs of weights for a GRU,
// 64 + 64 for an LSTM); 32 parts of 32 columns spilled for the GRU-512.
struct WaveParams {
    const float *blob; unsigned blob_bytes;
    unsigned whh[8], wih[8];       // byte offsets per layer: packed W_hh; packed W_ih (layers >= 1, same thread map)
    const float *bias[8];          // layers >= 1: b_ih (+ b_hh except the GRU's n gate), [GATES*H]
    const float *bhn[8];           // GRU: b_hn [H]
    const float *xproj;            // layer 0: [T][GATES*H]
    float *out;                    // [T][H] top layer
    unsigned long long *hbox;      // [layers][T][H] granules
    unsigned *status;
    unsigned epoch;
    int T, H, layers, wgs_per_layer, stride;
};

template <int GATES, int P, int CPP> __global__ __launch_bounds__(NT) void rnn_wave(WaveParams p)
{
    __shared__ __attribute__((aligned(16))) float hown[512];
    __shared__ __attribute__((aligned(16))) float hlow[512];
    if (blockIdx.x % p.stride) return;
    const int b = blockIdx.x / p.stride;
    const int l = b / p.wgs_per_layer, wg = b % p.wgs_per_layer;
    constexpr int U = NT / P, Q = CPP / 4;                  // units per workgroup; float4 per gate per thread
    const int tid = threadIdx.x, part = tid % P, ul = tid / P;
    const int unit = wg * U + ul;
    const int H = p.H;
    const __amdgpu_buffer_rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc((void *)p.blob, 0, (int)p.blob_bytes, 0x00020000);
    const unsigned plane = (unsigned)p.T * (unsigned)H * 8u;
    const __amdgpu_buffer_rsrc_t box = __builtin_amdgcn_make_buffer_rsrc((void *)p.hbox, 0, (int)(plane * (unsigned)p.layers), 0x00020000);
    float4 W[GATES * Q], V[GATES * Q];
#pragma unroll
    for (int i = 0; i < GATES * Q; ++i) {
        W[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            blob, tid * 16, (int)(p.whh[l] + ((unsigned)wg * GATES * Q + (unsigned)i) * NT * 16u), 0));
        V[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l > 0)
            V[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                blob, tid * 16, (int)(p.wih[l] + ((unsigned)wg * GATES * Q + (unsigned)i) * NT * 16u), 0));
    }
    float bg[GATES];
#pragma unroll
    for (int g = 0; g < GATES; ++g) bg[g] = (l > 0 && part == 0) ? p.bias[l][g * H + unit] : 0.f;
    const float bhn = (GATES == 3 && part == 0) ? p.bhn[l][unit] : 0.f;
    const bool top = l + 1 == p.layers;
    float hprev = 0.f, cprev = 0.f;
    for (int t = 0; t < p.T; ++t) {
        float xg[GATES];
#pragma unroll
        for (int g = 0; g < GATES; ++g) xg[g] = bg[g];
        if (l == 0 && part == 0) {
#pragma unroll
            for (int g = 0; g < GATES; ++g) xg[g] = p.xproj[((size_t)t * GATES + g) * H + unit];
        }
        bool ok = true;
        // two polls per thread at most: own layer's h_{t-1} (slot t-1), the layer below's h_t (slot t)
        for (int which = 0; which < 2 && ok; ++which) {
            float *dst = which ? hlow : hown;
            if (which == 0 && t == 0) { if (tid < H) dst[tid] = 0.f; continue; }
            if (which == 1 && l == 0) continue;
            if (tid < H) {
                const int slot = (int)((unsigned)(which ? l - 1 : l) * plane + (unsigned)(which ? t : t - 1) * (unsigned)H * 8u);
                for (unsigned spins = 0;;) {
                    const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(box, tid * 8, slot, AUX_SC1);
                    asm volatile("" ::: "memory");
                    if (g.y == p.epoch) { dst[tid] = __uint_as_float(g.x); break; }
                    if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(p.status, RLX_AGENT) != 0)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) atomicCAS(p.status, 0u, 0x2000000u + ((unsigned)l << 20) + (unsigned)t);
            }
        }
        if (!__syncthreads_and(ok)) return;
        float a[GATES], c[GATES];
        {
            const float4 *v = reinterpret_cast<const float4 *>(hown + part * CPP);
            const float4 *u = reinterpret_cast<const float4 *>(hlow + part * CPP);
#pragma unroll
            for (int g = 0; g < GATES; ++g) { a[g] = 0.f; c[g] = 0.f; }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float4 x = v[q];
#pragma unroll
                for (int g = 0; g < GATES; ++g) a[g] = dot4(W[g * Q + q], x, a[g]);
            }
            if (l > 0) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float4 x = u[q];
#pragma unroll
                    for (int g = 0; g < GATES; ++g) c[g] = dot4(V[g * Q + q], x, c[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < GATES; ++g) { a[g] = sum_p<P>(a[g]); c[g] = sum_p<P>(c[g]); }
        }
        if (part == 0) {
#pragma unroll
            for (int g = 0; g < GATES; ++g) xg[g] += c[g];       // input side: W_ih h_below + biases (layer 0: from the gemm)
            float hn;
            if (GATES == 3) {
                const float r = sigmoidf(xg[0] + a[0]);
                const float z = sigmoidf(xg[1] + a[1]);
                const float n = tanhf(xg[2] + r * (a[2] + bhn));
                hn = (1.f - z) * n + z * hprev;
            } else {
                const float i = sigmoidf(xg[0] + a[0]);
                const float f = sigmoidf(xg[1] + a[1]);
                const float g = tanhf(xg[2] + a[2]);
                const float o = sigmoidf(xg[GATES - 1] + a[GATES - 1]);
                cprev = f * cprev + i * g;
                hn = o * tanhf(cprev);
            }
            hprev = hn;
            u32x2 gr; gr.x = __float_as_uint(hn); gr.y = p.epoch;
            if (t + 1 < p.T || !top)          // consumers: this layer's next step, and the layer above at this step
                __builtin_amdgcn_raw_buffer_store_b64(gr, box, unit * 8, (int)((unsigned)l * plane + (unsigned)t * (unsigned)H * 8u), AUX_SC1);
            if (top) p.out[(size_t)t * H + unit] = hn;
        }
        __syncthreads();
    }
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hipfail(hipError_t e, const char *what) { return fail(LSPRNN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
static size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

struct Slot {
    std::string key;
    size_t numel = 0;
    std::vector<float> data;
    bool set = false;
};

}  // namespace lsprnn

using namespace lsprnn;

struct lsprnn_handle {
    lsprnn_config cfg{};
    int gates = 0, P = 0, U = 0, G = 0;
    std::vector<Slot> tensors;
    std::map<std::string, int> index;
    std::vector<size_t> o_wih, o_bias, o_whh, o_bhn;     // per layer, float offsets
    std::vector<size_t> o_whh_w, o_wih_w;                // wavefront kernel: packed W_hh / W_ih (its own thread map)
    int Pw = 0, CPPw = 0, Uw = 0, Gw = 0;               // wavefront geometry
    size_t blob_floats = 0;
    const float *blob = nullptr;
    float *ws = nullptr;
    bool boxes_clean = false, wave_fit_checked = false, wave_fits = false, layer_fit_checked = false;
    int fit_device = -1;         // the device the cached residency answers belong to
    unsigned epoch = 0;
    void add(const std::string &k, size_t n) { Slot s; s.key = k; s.numel = n; index[k] = (int)tensors.size(); tensors.push_back(std::move(s)); }
    const std::vector<float> &T(const std::string &k) const { return tensors[index.at(k)].data; }
    int in_size(int l) const { return l == 0 ? cfg.input_size : cfg.hidden_size; }
    size_t xproj_floats() const { return align64((size_t)cfg.max_steps * gates * cfg.hidden_size); }
    size_t hseq_floats() const { return align64((size_t)cfg.max_steps * cfg.hidden_size); }
    size_t box_bytes() const { return (size_t)cfg.num_layers * cfg.max_steps * cfg.hidden_size * 8; }
};

extern "C" {

const char *lsprnn_last_error(void) { return g_err.c_str(); }
int lsprnn_abi_version(void) { return LSPRNN_ABI_VERSION; }

int lsprnn_create(const lsprnn_config *cfg, lsprnn_handle **out)
{
    if (!cfg || !out) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != LSPRNN_ABI_VERSION) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "abi_version mismatch");
    if (cfg->cell != LSPRNN_CELL_GRU && cfg->cell != LSPRNN_CELL_LSTM) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "cell");
    if (cfg->hidden_size != 256 && cfg->hidden_size != 512)
        return fail(LSPRNN_ERR_UNSUPPORTED, "hidden_size must be 256 or 512 (the sizes the reference's GRU / LSTM stacks use)");
    if (cfg->num_layers < 1 || cfg->num_layers > 8) return fail(LSPRNN_ERR_UNSUPPORTED, "num_layers in 1..8");
    if (cfg->input_size < 4 || cfg->input_size % 4) return fail(LSPRNN_ERR_SHAPE, "input_size must be a multiple of 4");
    if (cfg->max_steps < 1 || cfg->max_steps > (1 << 20)) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "max_steps");
    lsprnn_handle *h = new (std::nothrow) lsprnn_handle;
    if (!h) return fail(LSPRNN_ERR_STATE, "out of host memory");
    h->cfg = *cfg;
    h->gates = cfg->cell == LSPRNN_CELL_GRU ? 3 : 4;
    h->P = cfg->hidden_size / 32;
    h->U = NT / h->P;
    h->G = cfg->hidden_size / h->U;
    const size_t H = cfg->hidden_size, gh = (size_t)h->gates * H;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o = align64(o + n); return at; };
    for (int l = 0; l < cfg->num_layers; ++l) {
        const std::string s = "_l" + std::to_string(l);
        h->add("weight_ih" + s, gh * h->in_size(l));
        h->add("weight_hh" + s, gh * H);
        h->add("bias_ih" + s, gh);
        h->add("bias_hh" + s, gh);
        h->o_wih.push_back(take(gh * h->in_size(l)));
        h->o_bias.push_back(take(gh));
        h->o_whh.push_back(take(gh * H));
        h->o_bhn.push_back(take(H));
        h->o_whh_w.push_back(take(gh * H));
        h->o_wih_w.push_back(l ? take(gh * H) : 0);
    }
    // wavefront geometry: two weight matrices per thread -> parts of 16 columns
    h->CPPw = 16; h->Pw = cfg->hidden_size / 16; h->Uw = NT / h->Pw; h->Gw = cfg->hidden_size / h->Uw;
    h->blob_floats = o;
    *out = h;
    return LSPRNN_OK;
}

int lsprnn_destroy(lsprnn_handle *h) { delete h; return LSPRNN_OK; }
int lsprnn_num_tensors(const lsprnn_handle *h) { return h ? (int)h->tensors.size() : fail(LSPRNN_ERR_INVALID_ARGUMENT, "null handle"); }

int lsprnn_tensor_info(const lsprnn_handle *h, int index, const char **key, size_t *numel)
{
    if (!h || index < 0 || index >= (int)h->tensors.size()) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "tensor index out of range");
    if (key) *key = h->tensors[index].key.c_str();
    if (numel) *numel = h->tensors[index].numel;
    return LSPRNN_OK;
}

int lsprnn_set_tensor(lsprnn_handle *h, const char *key, const float *host_data, size_t numel)
{
    if (!h || !key || !host_data) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    auto it = h->index.find(key);
    if (it == h->index.end()) return fail(LSPRNN_ERR_INVALID_ARGUMENT, std::string("unknown tensor key: ") + key);
    Slot &t = h->tensors[it->second];
    if (numel != t.numel) return fail(LSPRNN_ERR_SHAPE, std::string("wrong element count for ") + key);
    t.data.assign(host_data, host_data + numel);
    t.set = true;
    return LSPRNN_OK;
}

size_t lsprnn_packed_bytes(const lsprnn_handle *h) { return h ? h->blob_floats * sizeof(float) : 0; }

int lsprnn_pack_weights(lsprnn_handle *h, void *host_dst, size_t bytes)
{
    if (!h || !host_dst) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->blob_floats * sizeof(float)) return fail(LSPRNN_ERR_SHAPE, "destination smaller than lsprnn_packed_bytes()");
    for (const Slot &t : h->tensors)
        if (!t.set) return fail(LSPRNN_ERR_STATE, "tensor not set: " + t.key);
    float *d = static_cast<float *>(host_dst);
    std::memset(d, 0, h->blob_floats * sizeof(float));
    const int H = h->cfg.hidden_size, GT = h->gates, P = h->P, U = h->U;
    for (int l = 0; l < h->cfg.num_layers; ++l) {
        const std::string s = "_l" + std::to_string(l);
        const auto &wih = h->T("weight_ih" + s), &whh = h->T("weight_hh" + s), &bih = h->T("bias_ih" + s), &bhh = h->T("bias_hh" + s);
        std::memcpy(d + h->o_wih[l], wih.data(), wih.size() * sizeof(float));
        for (int r = 0; r < GT * H; ++r) {
            const bool gru_n = GT == 3 && r >= 2 * H;          // b_hn stays with the hidden product (inside r * (...))
            d[h->o_bias[l] + r] = gru_n ? bih[r] : bih[r] + bhh[r];
        }
        if (GT == 3) std::memcpy(d + h->o_bhn[l], bhh.data() + 2 * H, sizeof(float) * H);
        // the wavefront kernel's layout of W_hh and (layers >= 1) W_ih: Pw parts of CPPw columns, Uw units per workgroup
        {
            const int Pw = h->Pw, Q = h->CPPw / 4, Uw = h->Uw;
            for (int m = 0; m < (l ? 2 : 1); ++m) {
                const std::vector<float> &src = m ? wih : whh;
                float *o2 = d + (m ? h->o_wih_w[l] : h->o_whh_w[l]);
                for (int w = 0; w < h->Gw; ++w)
                    for (int g = 0; g < GT; ++g)
                        for (int q = 0; q < Q; ++q)
                            for (int t = 0; t < NT; ++t)
                                for (int e = 0; e < 4; ++e) {
                                    const int row = g * H + w * Uw + t / Pw, col = (t % Pw) * h->CPPw + q * 4 + e;
                                    o2[((((size_t)w * GT + g) * Q + q) * NT + t) * 4 + e] = src[(size_t)row * H + col];
                                }
            }
        }
        // W_hh: workgroup w, thread t = (unit w*U + t/P, column part t%P), gate g, float4 q of its 32 columns
        float *o = d + h->o_whh[l];
        for (int w = 0; w < h->G; ++w)
            for (int g = 0; g < GT; ++g)
                for (int q = 0; q < 8; ++q)
                    for (int t = 0; t < NT; ++t)
                        for (int e = 0; e < 4; ++e) {
                            const int row = g * H + w * U + t / P, col = (t % P) * 32 + q * 4 + e;
                            o[((((size_t)w * GT + g) * 8 + q) * NT + t) * 4 + e] = whh[(size_t)row * H + col];
                        }
    }
    return LSPRNN_OK;
}

int lsprnn_bind_weights(lsprnn_handle *h, const void *packed_dev, size_t bytes)
{
    if (!h || !packed_dev) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->blob_floats * sizeof(float)) return fail(LSPRNN_ERR_SHAPE, "blob smaller than lsprnn_packed_bytes()");
    if ((uintptr_t)packed_dev & 15) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "blob must be 16-byte aligned");
    h->blob = static_cast<const float *>(packed_dev);
    return LSPRNN_OK;
}

size_t lsprnn_workspace_bytes(const lsprnn_handle *h)
{
    if (!h) return 0;
    return (h->xproj_floats() + 2 * h->hseq_floats()) * sizeof(float) + h->box_bytes() + 256;
}

int lsprnn_bind_workspace(lsprnn_handle *h, void *workspace_dev, size_t bytes)
{
    if (!h || !workspace_dev) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < lsprnn_workspace_bytes(h)) return fail(LSPRNN_ERR_SHAPE, "workspace smaller than lsprnn_workspace_bytes()");
    if ((uintptr_t)workspace_dev & 15) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "workspace must be 16-byte aligned");
    h->ws = static_cast<float *>(workspace_dev);
    h->boxes_clean = false;
    return LSPRNN_OK;
}

int lsprnn_forward(lsprnn_handle *h, const float *x_dev, int T, float *out_dev, void *stream)
{
    if (!h || !x_dev || !out_dev) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->blob) return fail(LSPRNN_ERR_STATE, "weights not bound (lsprnn_bind_weights)");
    if (!h->ws) return fail(LSPRNN_ERR_STATE, "workspace not bound (lsprnn_bind_workspace)");
    if (T < 1 || T > h->cfg.max_steps) return fail(LSPRNN_ERR_SHAPE, "T out of range (max_steps)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = h->cfg.hidden_size, GT = h->gates, L = h->cfg.num_layers;
    float *xproj = h->ws, *hs0 = xproj + h->xproj_floats(), *hs1 = hs0 + h->hseq_floats();
    char *tail = reinterpret_cast<char *>(hs1 + h->hseq_floats());
    unsigned long long *box = reinterpret_cast<unsigned long long *>(tail);
    unsigned *status = reinterpret_cast<unsigned *>(tail + h->box_bytes());
    if (hipMemsetAsync(status, 0, 64, s) != hipSuccess) return fail(LSPRNN_ERR_HIP, "hipMemsetAsync(status)");
    if (!h->boxes_clean) {   // tags are launch counters: earlier launches never match; clear what the buffer held when bound
        const hipError_t e = hipMemsetAsync(box, 0, h->box_bytes(), s);
        if (e != hipSuccess) return hipfail(e, "hipMemsetAsync(mailboxes)");
        h->boxes_clean = true;
    }
    // Route: stacks run the wavefront kernel (all layers in one launch, L * Gw workgroups polling each other) unless LSPRNN_FLAG_PER_LAYER asks for
    // one launch per layer (G polling workgroups) -- or the device cannot hold the stack at once (a partitioned or small device): then the
    // per-layer route is taken silently.  The residency answers are cached per handle AND device.
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (dev != h->fit_device) { h->fit_device = dev; h->wave_fit_checked = h->layer_fit_checked = false; h->wave_fits = false; }
    bool wave = (h->cfg.flags & LSPRNN_FLAG_PER_LAYER) ? false : L > 1;
    if (wave && !h->wave_fit_checked) {
        void (*kern0)(WaveParams) = GT == 3 ? (h->Pw == 32 ? rnn_wave<3, 32, 16> : rnn_wave<3, 16, 16>) : (h->Pw == 32 ? rnn_wave<4, 32, 16> : rnn_wave<4, 16, 16>);
        bool ok = false;
        const hipError_t e0 = lspgemm::fits_resident(reinterpret_cast<const void *>(kern0), NT, 0, L * h->Gw, &ok);
        if (e0 != hipSuccess) return hipfail(e0, "occupancy query (rnn_wave)");
        h->wave_fits = ok;
        h->wave_fit_checked = true;
    }
    if (wave && !h->wave_fits) wave = false;
    if (wave) {
        lspgemm::GemmParams g{x_dev, h->blob + h->o_wih[0], nullptr, h->blob + h->o_bias[0], nullptr, xproj, T, GT * H, h->in_size(0), 1.0f, 0};
        hipError_t e = lspgemm::launch_gemm_f32(g, s);
        if (e != hipSuccess) return hipfail(e, "input projection gemm launch");
        if (++h->epoch == 0) h->epoch = 1;
        WaveParams p{};
        p.blob = h->blob; p.blob_bytes = (unsigned)(h->blob_floats * sizeof(float));
        for (int l = 0; l < L; ++l) {
            p.whh[l] = (unsigned)(h->o_whh_w[l] * sizeof(float)); p.wih[l] = (unsigned)(h->o_wih_w[l] * sizeof(float));
            p.bias[l] = h->blob + h->o_bias[l]; p.bhn[l] = h->blob + h->o_bhn[l];
        }
        p.xproj = xproj; p.out = out_dev; p.hbox = box; p.status = status; p.epoch = h->epoch;
        p.T = T; p.H = H; p.layers = L; p.wgs_per_layer = h->Gw;
        // workgroup b runs on XCD b % 8 (observed, speed only): keep the stack on as few XCDs as its size allows
        const int nwg = L * h->Gw;
        p.stride = nwg <= 32 ? 8 : (nwg <= 64 ? 4 : (nwg <= 128 ? 2 : 1));
        const dim3 grid(L * h->Gw * p.stride), block(NT);
        void (*kern)(WaveParams) = GT == 3 ? (h->Pw == 32 ? rnn_wave<3, 32, 16> : rnn_wave<3, 16, 16>) : (h->Pw == 32 ? rnn_wave<4, 32, 16> : rnn_wave<4, 16, 16>);
        hipLaunchKernelGGL(kern, grid, block, 0, s, p);
        e = hipGetLastError();
        return e == hipSuccess ? LSPRNN_OK : hipfail(e, "rnn_wave launch");
    }
    const float *in = x_dev;
    for (int l = 0; l < L; ++l) {
        lspgemm::GemmParams g{in, h->blob + h->o_wih[l], nullptr, h->blob + h->o_bias[l], nullptr, xproj, T, GT * H, h->in_size(l), 1.0f, 0};
        hipError_t e = lspgemm::launch_gemm_f32(g, s);
        if (e != hipSuccess) return hipfail(e, "input projection gemm launch");
        float *hseq = l + 1 == L ? out_dev : (l & 1 ? hs1 : hs0);
        if (++h->epoch == 0) h->epoch = 1;
        LayerParams p{};
        p.blob = h->blob; p.blob_bytes = (unsigned)(h->blob_floats * sizeof(float));
        p.whh = (unsigned)(h->o_whh[l] * sizeof(float)); p.bhn = h->blob + h->o_bhn[l];
        p.xproj = xproj; p.hseq = hseq; p.hbox = box; p.status = status; p.epoch = h->epoch;
        p.T = T; p.H = H;
        p.stride = 8;           // every 8th block: the whole all-gather sits behind one XCD's L2 (speed only, see csrc/a2h.hip)
        const dim3 grid(h->G * p.stride), block(NT);
        void (*kern)(LayerParams) = GT == 3 ? (h->P == 16 ? rnn_layer<3, 16> : rnn_layer<3, 8>) : (h->P == 16 ? rnn_layer<4, 16> : rnn_layer<4, 8>);
        if (!h->layer_fit_checked) {
            bool ok = false;
            e = lspgemm::fits_resident(reinterpret_cast<const void *>(kern), NT, 0, h->G, &ok);
            if (e != hipSuccess) return hipfail(e, "occupancy query (rnn_layer)");
            if (!ok) return fail(LSPRNN_ERR_UNSUPPORTED, "rnn_layer: the G = H / P polling workgroups of a layer (16..64) do not fit this device at once -- "
                                                         "the recurrent kernels need at least that many free workgroup slots (include/lsprnn.h)");
            h->layer_fit_checked = true;
        }
        hipLaunchKernelGGL(kern, grid, block, 0, s, p);
        e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "rnn_layer launch");
        in = hseq;
    }
    return LSPRNN_OK;
}

int lsprnn_status(lsprnn_handle *h, void *stream, uint32_t *code)
{
    if (!h || !code) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->ws) return fail(LSPRNN_ERR_STATE, "workspace not bound");
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize");
    const char *tail = reinterpret_cast<const char *>(h->ws + h->xproj_floats() + 2 * h->hseq_floats());
    e = hipMemcpy(code, tail + h->box_bytes(), sizeof(uint32_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? LSPRNN_OK : hipfail(e, "hipMemcpy(status)");
}

int lsprnn_linear(const float *x_dev, const float *w_dev, const float *scale_dev, const float *shift_dev, float *y_dev,
                  int M, int N, int K, int leaky, void *stream)
{
    if (!x_dev || !w_dev || !shift_dev || !y_dev) return fail(LSPRNN_ERR_INVALID_ARGUMENT, "null argument");
    if (M < 1 || N < 1 || K < 4 || K % 4) return fail(LSPRNN_ERR_SHAPE, "need M, N >= 1 and K a multiple of 4");
    lspgemm::GemmParams g{x_dev, w_dev, scale_dev, shift_dev, nullptr, y_dev, M, N, K, 1.0f, leaky};
    const hipError_t e = lspgemm::launch_gemm_f32(g, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? LSPRNN_OK : hipfail(e, "linear gemm launch");
}

}  // extern "C"
