// C ABI of liblspf2f.so (include/lspf2f.h).  Host side: owns the plan, never owns device memory.
#include "../../include/lspf2f.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "plan.h"

using namespace lspf2f;

struct GraphKey {
    const void *feat, *cand, *out, *out_u8, *ws, *blob;
    int cand_batch, batch;
    bool operator==(const GraphKey &o) const
    {
        return feat == o.feat && cand == o.cand && out == o.out && out_u8 == o.out_u8 && ws == o.ws && blob == o.blob &&
               cand_batch == o.cand_batch && batch == o.batch;
    }
};

struct CachedGraph {
    GraphKey key{};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void reset()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr;
    }
};
static const size_t kMaxCachedGraphs = 8;   // demo.py-style callers cycle through a few buffers

struct lspf2f_handle {
    Plan plan;
    lspf2f_config cfg{};
    const char *blob = nullptr;   // device, caller-owned
    size_t blob_size = 0;
    char *ws = nullptr;           // device, caller-owned
    size_t ws_size = 0;
    bool packed = false;
    bool use_graph = true;
    // (the next four: `tune` keys of lspf2f_create_tuned -- tools, tests, A-B runs)
    bool last_direct = false;     // lastconv_direct=1: 16-bit plans run the direct last-conv kernel instead of the GEMM form
    bool prefetch = true;         // prefetch=0: weight-streaming layers do not request the next launch's weights
    bool in_small_regs = true;    // in_small_regs=0: InstanceNorm plans run in_small in its three-pass form (re-reads the slab for the variance and the normalisation)
    bool smallm_dma = true;       // smallm_dma=0: conv3x3_smallm stages its input tensor through registers (the form of rounds 2-4) instead of LDS-DMA pieces
    bool fuse_splitk = true;      // fused_splitk=0: always combine split-K slabs with a separate launch
    bool counters_clean = false;  // the arrival counters at the head of the workspace were zeroed since it was bound
    int first_direct = 0;         // firstconv=1: vector-ALU first conv; 2: register-staged matrix-core kernel
    int timing_part = 3;          // lspf2f_subset_timed: 1 = main kernels only, 2 = split-K reduce only, 3 = everything (always 3 on the hot path)
    int last_route = 0;           // lastconv=<route>: forced direct last-conv kernel (LastConvParams::route; tests only)
    const void *cand_cached = nullptr;   // candidate stack whose first-conv contribution sits in the workspace cache
    hipStream_t cap_stream = nullptr;
    // tail_prefetch (round 6, off by default): a side branch of the captured forward reads the weights of the <= 16x16 levels while the levels above compute
    int tail_prefetch = 0;        // 0 off | 1 plain loads | 2 non-temporal loads
    int tail_prefetch_at = -1;    // layer index the branch forks in front of (-1: the first layer)
    int tail_prefetch_wgs = 32;   // workgroups of the touching kernel
    int tail_prefetch_mb = 0;     // 0 = the whole range, else only its first MB
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<CachedGraph> graphs;
    size_t next_victim = 0;
    void drop_graphs()
    {
        for (auto &g : graphs) g.reset();
        graphs.clear();
    }
    ~lspf2f_handle()
    {
        drop_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};

#ifdef LSPF2F_ABLATE
// tools/ablate.sh builds only: the ablation bits of the kernels (the shipped library never reads the process environment)
static int ablate_dbg() { const char *env = std::getenv("LSP_HIP_DBG"); return env ? std::atoi(env) : 0; }
#endif

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hipfail(hipError_t e, const char *what)
{
    return fail(LSPF2F_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

extern "C" {

const char *lspf2f_last_error(void) { return g_err.c_str(); }
int lspf2f_abi_version(void) { return LSPF2F_ABI_VERSION; }

// "key=value,key=value" -> (key, integer value) pairs; "" / NULL = none.  Whitespace around tokens is ignored.
static bool parse_tune(const char *tune, std::vector<std::pair<std::string, int>> *out, std::string *bad)
{
    if (!tune) return true;
    std::string s(tune);
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        std::string tok = s.substr(i, j - i);
        i = j + 1;
        const size_t a = tok.find_first_not_of(" \t"), b = tok.find_last_not_of(" \t");
        if (a == std::string::npos) continue;
        tok = tok.substr(a, b - a + 1);
        const size_t eq = tok.find('=');
        if (eq == std::string::npos || eq == 0 || eq + 1 >= tok.size()) { *bad = tok; return false; }
        char *endp = nullptr;
        const long v = std::strtol(tok.c_str() + eq + 1, &endp, 10);
        if (*endp) { *bad = tok; return false; }
        out->emplace_back(tok.substr(0, eq), (int)v);
    }
    return true;
}

int lspf2f_create(const lspf2f_config *cfg, lspf2f_handle **out) { return lspf2f_create_tuned(cfg, nullptr, out); }

int lspf2f_create_tuned(const lspf2f_config *cfg, const char *tune, lspf2f_handle **out)
{
    if (!cfg || !out) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != LSPF2F_ABI_VERSION) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    if (cfg->dtype != LSPF2F_DTYPE_F32 && cfg->dtype != LSPF2F_DTYPE_BF16 && cfg->dtype != LSPF2F_DTYPE_F16) return fail(LSPF2F_ERR_UNSUPPORTED, "unknown dtype");
    if (cfg->height != cfg->width) return fail(LSPF2F_ERR_UNSUPPORTED, "frames must be square (loadSize x loadSize)");
    if (cfg->max_batch < 1) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "max_batch must be >= 1");
    std::vector<std::pair<std::string, int>> kv;
    std::string bad;
    if (!parse_tune(tune, &kv, &bad)) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "tune: expected key=integer, got '" + bad + "'");
    lspf2f_handle *h = new (std::nothrow) lspf2f_handle();
    if (!h) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "out of host memory");
    h->cfg = *cfg;
    h->use_graph = (cfg->flags & LSPF2F_FLAG_NO_GRAPH) == 0;
    Plan &P = h->plan;
    P.use_wino4 = (cfg->flags & LSPF2F_FLAG_WINO4) != 0;
    // The switches of tools, tests and A-B runs: applied HERE, once per handle, before the plan is built (some decide what the packed blob
    // carries); nothing in this library reads the process environment.
    for (const auto &e : kv) {
        const std::string &k = e.first;
        const int v = e.second;
        if (k == "graph") h->use_graph = h->use_graph && v != 0;
        else if (k == "wino") P.use_wino = v != 0;
        else if (k == "wino4") P.use_wino4 = v != 0;
        else if (k == "wino_pre") P.wino_pre = v != 0;
        else if (k == "wino_xcd") P.wino_xcd = v;
        else if (k == "wino_il") P.wino_il = v != 0;
        else if (k == "wino_ureg") P.wino_ureg = v;
        else if (k == "in_wino_stats") P.in_wino_stats = v != 0;
        else if (k == "wino_rot") P.wino_rot = v != 0;
        else if (k == "winoup") P.use_winoup = v != 0;
        else if (k == "winoup_nb") P.winoup_nb = v;
        else if (k == "winoup_target") P.winoup_target = v;
        else if (k == "igemm_xcd") P.igemm_xcd = v;
        else if (k == "bandconv") P.use_bandconv = v != 0;
        else if (k == "bandconv_min_blocks") P.bandconv_min_blocks = v;
        else if (k == "bandconv_min_frames") P.bandconv_min_frames_small = v;
        else if (k == "patch16") P.use_patch16 = v != 0;
        else if (k == "patchup16") P.use_patchup16 = v != 0;
        else if (k == "patch16_deep") P.patch16_deep = v != 0;
        else if (k == "patch16_min_blocks") P.patch16_min_blocks = v;
        else if (k == "rowup") P.use_rowup = v != 0;
        else if (k == "rowlast") P.use_rowlast = v != 0;
        else if (k == "rowlast_fused") P.rowlast_fused = v != 0;
        else if (k == "rowconv") P.use_rowconv = v != 0;
        else if (k == "fullk_split") P.use_fullk_split = v != 0;
        else if (k == "fullk_split_tiles") P.fullk_split_max_tiles = v;
        else if (k == "fullk_s2") P.use_fullk_s2 = v;
        else if (k == "all_forms") P.keep_all_forms = v != 0;
        else if (k == "blob_pad_kb") P.blob_pad_kb = v < 0 ? 0 : v;
        else if (k == "fused_splitk") h->fuse_splitk = v != 0;
        else if (k == "fused_splitk16") P.fused_splitk16 = v != 0;
        else if (k == "out_wt") P.out_wt = v;
        else if (k == "fullk16") P.fullk16_levels = v;
        else if (k == "fullk16_min_frames") P.fullk16_min_frames = v;
        else if (k == "wino_prio") P.wino_prio = v;
        else if (k == "prefetch") h->prefetch = v != 0;
        else if (k == "tail_prefetch") h->tail_prefetch = v;
        else if (k == "tail_prefetch_at") h->tail_prefetch_at = v;
        else if (k == "tail_prefetch_wgs") h->tail_prefetch_wgs = v;
        else if (k == "tail_prefetch_mb") h->tail_prefetch_mb = v;
        else if (k == "smallm_dma") h->smallm_dma = v != 0;
        else if (k == "smallm_kb") P.smallm_kb = v;
        else if (k == "in_small_regs") h->in_small_regs = v != 0;
        else if (k == "in_smallm_fused") P.in_smallm_fused = v != 0;
        else if (k == "in_small_max_hw") P.in_small_max_hw = v;
        else if (k == "lastconv_direct") h->last_direct = v != 0;      // 16-bit plans: the direct last-conv kernel instead of the GEMM form
        else if (k == "lastconv") h->last_route = v;                   // LastConvParams::route (0 = by shape)
        else if (k == "firstconv") h->first_direct = v;                // FirstConvParams::force_direct (0 = by shape)
        else { delete h; return fail(LSPF2F_ERR_INVALID_ARGUMENT, "tune: unknown key '" + k + "'"); }
    }
    const std::string e = P.build(cfg->variant, cfg->input_nc, cfg->feat_nc, cfg->output_nc, cfg->ngf,
                                  cfg->num_downs, cfg->height,
                                  (cfg->flags & LSPF2F_FLAG_KEEP_INTERMEDIATES) != 0, cfg->dtype,
                                  (cfg->flags & LSPF2F_FLAG_INSTANCE_NORM) ? 1 : 0, cfg->max_batch);
    if (!e.empty()) { delete h; return fail(LSPF2F_ERR_UNSUPPORTED, e); }
    P.plan_batch(cfg->max_batch);
    *out = h;
    return LSPF2F_OK;
}

int lspf2f_destroy(lspf2f_handle *h)
{
    delete h;
    return LSPF2F_OK;
}

int lspf2f_num_tensors(const lspf2f_handle *h) { return h ? (int)h->plan.params.size() : fail(LSPF2F_ERR_INVALID_ARGUMENT, "null handle"); }

int lspf2f_tensor_info(const lspf2f_handle *h, int i, const char **name, int64_t dims[4], int *ndim)
{
    if (!h || i < 0 || i >= (int)h->plan.params.size()) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "bad tensor index");
    const ParamDesc &p = h->plan.params[i];
    if (name) *name = p.key.c_str();
    if (ndim) *ndim = (int)p.dims.size();
    if (dims) for (size_t d = 0; d < p.dims.size() && d < 4; ++d) dims[d] = p.dims[d];
    return LSPF2F_OK;
}

int lspf2f_set_tensor(lspf2f_handle *h, const char *key, const float *host, size_t numel)
{
    if (!h || !key || !host) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    auto it = h->plan.param_index.find(key);
    if (it == h->plan.param_index.end())
        return fail(LSPF2F_ERR_INVALID_ARGUMENT, std::string("unexpected state-dict key: ") + key);
    ParamDesc &p = h->plan.params[it->second];
    if (numel != p.numel())
        return fail(LSPF2F_ERR_SHAPE, std::string("size mismatch for ") + key + ": got " + std::to_string(numel) +
                                          ", expected " + std::to_string(p.numel()));
    p.data.assign(host, host + numel);
    p.set = true;
    h->packed = false;
    return LSPF2F_OK;
}

size_t lspf2f_packed_bytes(const lspf2f_handle *h) { return h ? h->plan.blob_bytes : 0; }

int lspf2f_pack_weights(lspf2f_handle *h, void *host_blob, size_t bytes)
{
    if (!h || !host_blob) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    const std::string e = h->plan.pack(host_blob, bytes);
    if (!e.empty())
        return fail(e.rfind("missing", 0) == 0 ? LSPF2F_ERR_MISSING_TENSOR : LSPF2F_ERR_INVALID_ARGUMENT, e);
    // the fp32 copies are no longer needed once packed; a second pack of the same handle needs the tensors set again (it answers MISSING_TENSOR, it used to crash)
    for (auto &p : h->plan.params) { std::vector<float>().swap(p.data); p.set = false; }
    h->packed = true;
    return LSPF2F_OK;
}

int lspf2f_bind_weights(lspf2f_handle *h, const void *dev_blob, size_t bytes)
{
    if (!h || !dev_blob) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes < h->plan.blob_bytes) return fail(LSPF2F_ERR_SHAPE, "packed weight arena too small");
    if ((uintptr_t)dev_blob % 256) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "weight arena must be 256-byte aligned");
    h->drop_graphs();
    h->cand_cached = nullptr;
    h->blob = static_cast<const char *>(dev_blob);
    h->blob_size = bytes;
    return LSPF2F_OK;
}

size_t lspf2f_workspace_bytes(const lspf2f_handle *h, int batch)
{
    if (!h || batch < 1) return 0;
    // enough for EVERY batch of 1 .. batch frames: the plans of different batch sizes pick different kernels (split-K slabs come and go), so the need is
    // not monotonic in the batch -- a handle planned for 5 frames may need more for 3 of them than for 5
    size_t need = 0;
    for (int b = 1; b <= batch; ++b) need = std::max(need, h->plan.workspace_bytes(b));
    return need;
}

int lspf2f_bind_workspace(lspf2f_handle *h, void *dev_workspace, size_t bytes)
{
    if (!h || !dev_workspace) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if ((uintptr_t)dev_workspace % 256) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
    h->drop_graphs();
    h->cand_cached = nullptr;
    h->counters_clean = false;
    h->ws = static_cast<char *>(dev_workspace);
    h->ws_size = bytes;
    return LSPF2F_OK;
}

int lspf2f_num_layers(const lspf2f_handle *h) { return h ? (int)h->plan.layers.size() : fail(LSPF2F_ERR_INVALID_ARGUMENT, "null handle"); }

int lspf2f_plan_batch(lspf2f_handle *h, int batch)
{
    if (!h || batch < 1 || batch > h->cfg.max_batch) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "batch out of range");
    h->plan.plan_batch(batch);
    return LSPF2F_OK;
}

static const char *kernel_name(const LayerDesc &l, const Plan &P)
{
    switch (l.kind) {
    case kFirstConv: return "first_conv";
    case kLastConv: return l.wgemm_off >= 0 ? (l.wrl_off >= 0 && P.use_rowlast ? "last_conv (rowlast128 + pixel_shuffle_tanh)" : "last_conv (igemm3x3 + pixel_shuffle_tanh)") : "last_conv";
    default:
        if (l.wino4) return l.inorm ? (l.in_route == kInSmall ? "wino4_3x3+in_small" : "wino4_3x3+in_reduce_stats+in_finalize+in_apply")
                                    : (l.splits > 1 ? "wino4_3x3 (split-K combined in the launch)" : "wino4_3x3");
        if (l.inorm && (l.wino || l.winoup)) {
            const bool sm = l.in_route == kInSmall;
            if (l.winoup && l.in_route == kInWino) return l.winoup == 2 ? "winoup3x3<2>(stats)+in_finalize+in_apply" : "winoup3x3<1>(stats)+in_finalize+in_apply";
            if (l.winoup) return l.winoup == 2 ? (sm ? "winoup3x3<2>+in_small" : "winoup3x3<2>+in_reduce_stats+in_finalize+in_apply")
                                               : (sm ? "winoup3x3<1>+in_small" : "winoup3x3<1>+in_reduce_stats+in_finalize+in_apply");
            if (l.in_route == kInWino) return l.wino == 2 ? "wino3x3<2>(stats)+in_finalize+in_apply" : "wino3x3<1>(stats)+in_finalize+in_apply";
            return l.wino == 2 ? (sm ? "wino3x3<2>+in_small" : "wino3x3<2>+in_reduce_stats+in_finalize+in_apply")
                               : (sm ? "wino3x3<1>+in_small" : "wino3x3<1>+in_reduce_stats+in_finalize+in_apply");
        }
        if (l.winoup) return l.winoup == 2 ? (l.splits > 1 ? "winoup3x3<2> (split-K combined in the launch)" : "winoup3x3<2>")
                                           : (l.splits > 1 ? "winoup3x3<1> (split-K combined in the launch)" : "winoup3x3<1>");
        if (l.wino) return l.wino == 2 ? (l.splits > 1 ? "wino3x3<2> (split-K combined in the launch)" : "wino3x3<2>")
                                       : (l.splits > 1 ? "wino3x3<1> (split-K combined in the launch)" : "wino3x3<1>");
        if (l.inorm && l.fullk) return "conv3x3_fullk+in_small";
        if (l.fullk) return P.dtype ? "conv3x3_fullk16" : "conv3x3_fullk";
        if (l.rowup) return "rowup256";
        if (l.patch16) return l.up4 ? "conv3x3_patchup16" : "conv3x3_patch16";
        if (l.bandconv) return "bandconv512";
        if (l.rowconv) return l.c0 == 64 ? "rowconv64" : "rowconv128";
        if (l.inorm) return l.smallm ? (P.in_smallm_fused ? "conv3x3_smallm(in)" : "conv3x3_smallm+in_small") : l.in_route == kInFused ? "igemm3x3(stats)+in_finalize+in_apply"
                          : l.in_route == kInSmall ? "igemm3x3+in_small" : "igemm3x3+in_reduce_stats+in_finalize+in_apply";
        return l.smallm ? "conv3x3_smallm" : (l.splits > 1 ? (l.fused_splitk ? "igemm3x3 (split-K combined in the launch)" : "igemm3x3+splitk_reduce") : "igemm3x3");
    }
}

int lspf2f_layer_info_get(const lspf2f_handle *h, int i, lspf2f_layer_info *o)
{
    if (!h || !o || i < 0 || i >= (int)h->plan.layers.size()) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "bad layer index");
    const LayerDesc &l = h->plan.layers[i];
    o->name = l.name.c_str();
    o->kernel = kernel_name(l, h->plan);
    o->cin = l.cin; o->cout = l.cout; o->h_in = l.hs; o->h_out = l.ho; o->stride = l.stride;
    o->upsample = l.up || l.up4; o->concat = l.concat; o->residual = l.residual; o->relu = l.relu; o->tanh_out = l.tanh_out;
    o->tile_m = l.bm; o->tile_n = l.bn; o->split_k = l.splits; o->k_group = l.group;
    o->flops_per_frame = h->plan.layer_flops(l);
    o->act_bytes_per_frame = h->plan.layer_act_bytes(l);
    const bool sub = l.up4 || l.kind == kLastConv;   // sub-pixel form: 16/9 weight bytes, 4/9 FLOPs
    o->weight_bytes = (int64_t)l.cout * l.cin * (sub ? 16 : 9) * (int64_t)(h->plan.layer_weights_typed(l) ? h->plan.elt() : 4);
    o->exec_flops_per_frame = sub ? o->flops_per_frame * 4 / 9 : o->flops_per_frame;
    if (l.winoup) {     // up-conv Winograd form: 9 multiplies per 2x2 outputs instead of 36 (16 in the sub-pixel form); 9 transformed taps per (co, ci)
        o->exec_flops_per_frame = o->flops_per_frame / 4;
        o->weight_bytes = (int64_t)l.cout * l.cin * 9 * 4;
    }
    if (l.wino) {       // Winograd F(2x2, 3x3): 16 multiplies per 2x2 outputs instead of 36; the weights it reads are the 4x4 transformed ones
        o->exec_flops_per_frame = o->flops_per_frame * 4 / 9;
        o->weight_bytes = (int64_t)l.cout * l.cin * 16 * 4;
    }
    if (l.wino4) {      // Winograd F(4x4, 3x3): 36 multiplies per 4x4 outputs instead of 144; 36 transformed taps per (co, ci)
        o->exec_flops_per_frame = o->flops_per_frame / 4;
        o->weight_bytes = (int64_t)l.cout * l.cin * 36 * 4;
    }
    o->w_offset = l.w_off; o->scale_offset = l.scale_off; o->shift_offset = l.shift_off;
    o->out_offset = l.out >= 0 ? (int64_t)h->plan.tensors[l.out].offset : -1;
    return LSPF2F_OK;
}

int lspf2f_unet_prepare(const float *src_dev, int src_nchw, int batch, int h, int w, int c, float slope,
                        float *s2d_out_dev, int s2d_channels, float *relu_out_dev, void *hip_stream)
{
    if (!src_dev || (!s2d_out_dev && !relu_out_dev)) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    PrepareParams p{src_dev, src_nchw ? 1 : 0, batch, h, w, c, slope, s2d_out_dev, s2d_channels, relu_out_dev};
    const hipError_t e = launch_unet_prepare(p, static_cast<hipStream_t>(hip_stream));
    if (e == hipErrorInvalidValue) return fail(LSPF2F_ERR_SHAPE, "unet_prepare: h, w must be even and s2d_channels >= 4c, a multiple of 4");
    return e == hipSuccess ? LSPF2F_OK : hipfail(e, "unet_prepare launch");
}

int lspf2f_pixel_shuffle(const float *g_dev, int batch, int hs, int ws, int cout, int apply_tanh,
                         float *out_f32_dev, unsigned char *out_u8_dev, void *hip_stream)
{
    if (!g_dev || (!out_f32_dev && !out_u8_dev)) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (batch < 1 || hs < 1 || ws < 1 || cout < 1 || cout > 4) return fail(LSPF2F_ERR_SHAPE, "pixel_shuffle: cout must be in 1..4");
    ShuffleParams sp{g_dev, out_f32_dev, out_u8_dev, batch, hs, ws, cout, apply_tanh ? 1 : 0, 0, nullptr};
    const hipError_t e = launch_pixel_shuffle(sp, static_cast<hipStream_t>(hip_stream));
    return e == hipSuccess ? LSPF2F_OK : hipfail(e, "pixel_shuffle launch");
}

int64_t lspf2f_layer_form_offset(const lspf2f_handle *h, int layer, int form)
{
    if (!h || layer < 0 || layer >= (int)h->plan.layers.size()) return -1;
    const LayerDesc &l = h->plan.layers[layer];
    switch (form) {
    case 0: return l.w_off;
    case 1: return l.wfk_off;
    case 2: return l.wfk2_off;
    case 3: return l.wwg_off;
    case 4: return l.ww4_off;
    case 5: return l.wwu_off;
    case 6: return l.wru_off;
    case 7: return l.wbc_off;
    case 8: return l.wrc_off;
    case 9: return l.wgemm_off;
    case 10: return l.wrl_off;
    default: return -1;
    }
}

int lspf2f_memcpy(void *dst, const void *src, size_t bytes, void *hip_stream)
{
    if (!dst || !src) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    const hipError_t e = launch_copy16(dst, src, bytes, static_cast<hipStream_t>(hip_stream));
    if (e == hipErrorInvalidValue) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "lspf2f_memcpy: pointers and size must be multiples of 16 bytes");
    return e == hipSuccess ? LSPF2F_OK : hipfail(e, "lspf2f_memcpy launch");
}

int lspf2f_clock_probe(unsigned long long *out_dev, unsigned duration_us, void *hip_stream)
{
    if (!out_dev || duration_us == 0 || duration_us > 2000000u) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "clock_probe: null output or duration outside 1..2 000 000 us");
    const hipError_t e = launch_clock_probe(out_dev, duration_us, static_cast<hipStream_t>(hip_stream));
    return e == hipSuccess ? LSPF2F_OK : hipfail(e, "clock_probe launch");
}

}  // extern "C"

static int run_layer(lspf2f_handle *h, const LayerDesc &l, const float *feat, const float *cand, int cand_batch,
                     float *out, unsigned char *out_u8, int batch, hipStream_t s)
{
    const Plan &P = h->plan;
    auto tptr = [&](int t) -> float * { return t < 0 ? nullptr : reinterpret_cast<float *>(h->ws + P.tensors[t].offset); };
    auto bptr = [&](int64_t off) -> const float * { return off < 0 ? nullptr : reinterpret_cast<const float *>(h->blob + off); };
    hipError_t e = hipSuccess;
    // InstanceNorm behind a kernel that has written the complete raw conv output (+ bias) itself (the Winograd kernels: their split-K slabs are
    // combined in the launch): statistics + normalisation (+ residual, ReLU) as separate passes over that tensor
    auto in_after_complete_output = [&](hipError_t prev) -> hipError_t {
        if (prev != hipSuccess) return prev;
        float *st = reinterpret_cast<float *>(h->ws + P.stats_offset);
        const size_t slab = (size_t)batch * P.stats_groups_max;
        InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
        q.x = tptr(l.out); q.residual = tptr(l.res); q.relu = l.relu; q.partial = nullptr; q.splits = 1; q.bias = nullptr;
        q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
        if (l.in_route == kInSmall) return launch_in_small(q, s);
        q.psum = st; q.psq = st + slab * l.cout; q.pshift = st + 2 * slab * l.cout;
        q.mean = st + 3 * slab * l.cout; q.rstd = q.mean + (size_t)batch * l.cout;
        q.groups = (q.hw + 63) / 64; q.rows_per_group = 64;
        hipError_t r = launch_in_reduce_stats(q, s);
        if (r == hipSuccess) r = launch_in_finalize(q, s);
        if (r == hipSuccess) r = launch_in_apply(q, s);
        return r;
    };
    if (l.kind == kFirstConv) {
        FirstConvParams p{};
        p.feat = feat; p.cand = cand; p.w = bptr(l.w_off); p.out = tptr(l.out);
        p.B = batch; p.H = l.hs; p.W = l.hs; p.feat_nc = P.feat_nc; p.cand_nc = P.input_nc - P.feat_nc;
        p.cand_batch = cand_batch; p.Cout = l.cout; p.dtype = P.dtype;
        p.ci_begin = 0; p.ci_end = P.input_nc; p.base = nullptr; p.relu = 1;
        p.bias = bptr(l.shift_off);          // InstanceNorm plans: the conv bias (scale is 1); nullptr otherwise
        // two slots at the head of the workspace: [0] lspf2f_set_candidates' per-person cache, [1] the per-forward share of a
        // broadcast stack -- separate, so a broadcast forward never overwrites what the cache holds
        float *cache = reinterpret_cast<float *>(cand != nullptr ? h->ws + P.cand_cache_bytes() : h->ws);
        p.force_direct = h->first_direct;
#ifdef LSPF2F_ABLATE
        p.dbg = ablate_dbg();
#endif
        // a candidate stack shared by a batch (cand_batch == 1, batch > 1): its 12-channel share is computed once (matrix-core kernel,
        // channel range) and every frame adds its feature-map channel in a streaming pass -- measured 77 us vs 150 us at batch 8 for
        // running the full 13-channel layer per frame
        const bool shared = p.cand_nc > 0 && P.feat_nc > 0 && (cand == nullptr || (cand_batch == 1 && batch > 1));
        if (shared) {
            // candidate stack shared by the whole batch: its contribution is computed once (or taken
            // from lspf2f_set_candidates' cache when cand == NULL), each frame then adds its own
            // feature-map channels
            if (cand != nullptr) {
                FirstConvParams c = p;
                c.B = 1; c.cand_batch = 1; c.ci_begin = P.feat_nc; c.out = cache; c.relu = 0; c.bias = nullptr;
                e = launch_first_conv(c, s);
            }
            p.ci_end = P.feat_nc; p.base = cache;
            if (e == hipSuccess) e = launch_first_conv(p, s);
        } else {
            e = launch_first_conv(p, s);
        }
    } else if (l.kind == kLastConv && P.last_as_gemm(l) && !h->last_direct) {
        // bf16: 3x3 conv on the low-res source with N = 4 parities x cout through the MFMA kernel, then shuffle + tanh
        if (l.wrl_off >= 0 && P.use_rowlast) {
            RowLastParams q{};
            q.src0 = tptr(l.src0); q.src1 = tptr(l.src1); q.w = h->blob + l.wrl_off;
            q.out = reinterpret_cast<float *>(h->ws + P.partial_offset);
            q.B = batch; q.H = l.hs; q.W = l.hs; q.R = rowlast_rows(batch, l.hs, l.hs); q.dtype = P.dtype;
            // fp32 frames only: shuffle + tanh in the kernel's epilogue (no intermediate, no second launch); uint8 rows (tensor2im) keep the two-launch form
            const bool fused = P.rowlast_fused && out && !out_u8;
            if (fused) { q.out_nchw = out; q.cout = l.cout; q.apply_tanh = l.tanh_out ? 1 : 0; }
            e = launch_rowlast(q, s);
            if (e == hipSuccess && !fused) {
                ShuffleParams sp{reinterpret_cast<const float *>(h->ws + P.partial_offset), out, out_u8, batch, l.hs, l.hs, l.cout, l.tanh_out ? 1 : 0, 1, nullptr};
                e = launch_pixel_shuffle(sp, s);
            }
            if (e != hipSuccess) return hipfail(e, ("launch " + l.name).c_str());
            return LSPF2F_OK;
        }
        IgemmParams g{};
        g.src0 = tptr(l.src0); g.src1 = tptr(l.src1); g.w = h->blob + l.wgemm_off;
        g.out = h->ws + P.partial_offset; g.out_f32 = 1;
        g.B = batch; g.Hs = l.hs; g.Ws = l.hs; g.Ho = l.hs; g.Wo = l.hs;
        g.C0 = l.c0; g.C1 = l.c1; g.Cin = l.cin; g.Cout = 4 * l.cout;
        g.stride = 1; g.Mout = batch * l.hs * l.hs; g.M = g.Mout; g.dtype = P.dtype;
        g.ktiles_total = 9 * l.cin / P.ktile_channels(); g.splits = 1; g.ktiles_per_split = g.ktiles_total;
        e = launch_igemm(g, 128, 32, 1, s);
        if (e == hipSuccess) {
            ShuffleParams sp{reinterpret_cast<const float *>(h->ws + P.partial_offset), out, out_u8, batch, l.hs, l.hs, l.cout, l.tanh_out ? 1 : 0, 0, nullptr};
            e = launch_pixel_shuffle(sp, s);
        }
    } else if (l.kind == kLastConv) {
        LastConvParams p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.w_off); p.out = out;
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.C0 = l.c0; p.C1 = l.c1; p.Cout = l.cout; p.apply_tanh = l.tanh_out; p.out_u8 = out_u8; p.dtype = P.dtype;
        p.route = h->last_route;
        p.bias = bptr(l.shift_off);
        e = launch_last_conv(p, s);
    } else if (l.smallm) {
        SmallMParams p{};
        p.src = tptr(l.src0); p.w = bptr(l.w_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        const bool in_fused = l.inorm && P.in_smallm_fused;      // InstanceNorm plans: the workgroup holds every pixel of its channels -> normalisation in the epilogue
        p.residual = (l.inorm && !in_fused) ? nullptr : tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.Ho = l.ho; p.Wo = l.ho; p.Cin = l.cin; p.Cout = l.cout;
        p.stride = l.stride; p.up = l.up; p.relu = (l.inorm && !in_fused) ? 0 : l.relu; p.M = batch * l.ho * l.ho; p.dtype = P.dtype;
        p.in_fused = in_fused ? 1 : 0;
        p.stage_regs = h->smallm_dma ? 0 : 1;
        if (h->prefetch && P.dtype == 0 && !l.inorm) {
            // the next launch, if it is another weight-streaming layer: its weights are requested from inside this one
            const size_t li = (size_t)(&l - P.layers.data());
            if (li + 1 < P.layers.size()) {
                const LayerDesc &n = P.layers[li + 1];
                const int64_t off = n.smallm ? n.w_off : n.fullk ? ((n.splits == 2 && !n.c1) ? n.wfk2_off : n.wfk_off) : -1;
                if (off >= 0 && n.kind == kIgemm) { p.pf = h->blob + off; p.pf_bytes = (unsigned)((size_t)n.cout * 9 * n.cin * sizeof(float)); }
            }
        }
        e = launch_smallm(p, s);
        if (e == hipSuccess && l.inorm && !in_fused) {          // raw conv output (+ bias) -> statistics + normalisation (+ residual, ReLU) in one launch
            InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
            q.x = tptr(l.out); q.partial = nullptr; q.splits = 1; q.bias = nullptr; q.residual = tptr(l.res); q.relu = l.relu;
            q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
            e = launch_in_small(q, s);
        }
    } else if (l.winoup) {
        WinoUpParams p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.u = bptr(l.wwu_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.out = tptr(l.out);
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.C0 = l.c0; p.C1 = l.c1; p.N = l.cout; p.relu = l.inorm ? 0 : l.relu; p.splits = l.splits;
        p.out_wt = P.out_wt; p.prio = P.wino_prio;
        if (l.splits > 1) {
            p.partial = reinterpret_cast<float *>(h->ws + P.partial_offset);
            p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.counters_offset());
        }
        if (l.inorm && l.in_route == kInWino) {
            // the kernel's epilogue (or its split-K combine) leaves the sums of every tile-block of 128 output pixels: finalize + normalise only
            float *st = reinterpret_cast<float *>(h->ws + P.stats_offset);
            const size_t slab = (size_t)batch * P.stats_groups_max;
            InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
            q.x = tptr(l.out); q.residual = tptr(l.res); q.relu = l.relu; q.partial = nullptr; q.splits = 1; q.bias = nullptr;
            q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
            q.psum = st; q.psq = st + slab * l.cout; q.pshift = st + 2 * slab * l.cout;
            q.mean = st + 3 * slab * l.cout; q.rstd = q.mean + (size_t)batch * l.cout;
            q.groups = q.hw / 128; q.rows_per_group = 128;
            p.psum = q.psum; p.psq = q.psq; p.pshift = q.pshift;
            if (h->timing_part & 1) e = launch_winoup(p, l.winoup, s);
            if (e == hipSuccess) e = launch_in_finalize(q, s);
            if (e == hipSuccess) e = launch_in_apply(q, s);
        } else {
            if (h->timing_part & 1) e = launch_winoup(p, l.winoup, s);
            if (l.inorm) e = in_after_complete_output(e);
        }
    } else if (l.wino4) {
        WinoParams p{};
        p.src = tptr(l.src0); p.u = bptr(l.ww4_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = l.inorm ? nullptr : tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.H = l.ho; p.W = l.ho; p.C = l.cin; p.N = l.cout; p.relu = l.inorm ? 0 : l.relu; p.splits = l.splits;
        if (l.splits > 1) {
            p.partial = reinterpret_cast<float *>(h->ws + P.partial_offset);
            p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.counters_offset());
        }
        if (h->timing_part & 1) e = launch_wino4(p, s);
        if (l.inorm) e = in_after_complete_output(e);
    } else if (l.wino) {
        WinoParams p{};
        p.src = tptr(l.src0); p.u = bptr(l.wwg_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = l.inorm ? nullptr : tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.H = l.ho; p.W = l.ho; p.C = l.cin; p.N = l.cout; p.relu = l.inorm ? 0 : l.relu; p.splits = l.splits;
        if (l.splits > 1) {
            p.partial = reinterpret_cast<float *>(h->ws + P.partial_offset);
            p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.counters_offset());
        }
        p.nopre = P.wino_pre ? 0 : 1; p.xcd_force = P.wino_xcd + 1; p.no_il = P.wino_il ? 0 : 1; p.no_rot = P.wino_rot ? 0 : 1; p.ureg = P.wino_ureg; p.out_wt = P.out_wt; p.prio = P.wino_prio;
        if (l.inorm && l.in_route == kInWino) {
            // the kernel's epilogue (or its split-K combine) leaves the sums of every tile-block of 128 pixels: finalize + normalise only
            float *st = reinterpret_cast<float *>(h->ws + P.stats_offset);
            const size_t slab = (size_t)batch * P.stats_groups_max;
            InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
            q.x = tptr(l.out); q.residual = tptr(l.res); q.relu = l.relu; q.partial = nullptr; q.splits = 1; q.bias = nullptr;
            q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
            q.psum = st; q.psq = st + slab * l.cout; q.pshift = st + 2 * slab * l.cout;
            q.mean = st + 3 * slab * l.cout; q.rstd = q.mean + (size_t)batch * l.cout;
            q.groups = q.hw / 128; q.rows_per_group = 128;
            p.psum = q.psum; p.psq = q.psq; p.pshift = q.pshift;
            if (h->timing_part & 1) e = launch_wino(p, l.wino, s);
            if (e == hipSuccess) e = launch_in_finalize(q, s);
            if (e == hipSuccess) e = launch_in_apply(q, s);
        } else {
            if (h->timing_part & 1) e = launch_wino(p, l.wino, s);
            if (l.inorm) e = in_after_complete_output(e);
        }
    } else if (l.rowup) {
        RowUpParams p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.wru_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.out = tptr(l.out);
        p.B = batch; p.H = l.hs; p.W = l.hs; p.R = l.rowup; p.relu = l.relu; p.dtype = P.dtype;
        e = launch_rowup(p, s);
    } else if (l.patch16 && l.up4) {
        PatchConvParams p{};
        p.src = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.w_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.H = l.hs; p.W = l.hs; p.C = l.c0; p.C1 = l.c1; p.Cout = l.cout; p.relu = l.relu; p.dtype = P.dtype;
        e = launch_patchup16(p, l.patch16, l.bn, s);
    } else if (l.patch16) {
        PatchConvParams p{};
        p.src = tptr(l.src0); p.w = bptr(l.w_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.H = l.ho; p.W = l.ho; p.C = l.c0; p.Cout = l.cout; p.relu = l.relu; p.dtype = P.dtype;
        p.deep = P.patch16_deep;
        e = launch_patch16(p, l.patch16, l.bn, s);
    } else if (l.bandconv) {
        BandConvParams p{};
        p.src = tptr(l.src0); p.w = bptr(l.wbc_off); p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.W = l.ho; p.Cout = l.cout; p.relu = l.relu; p.dtype = P.dtype;
        e = launch_bandconv(p, s);
    } else if (l.rowconv) {
        RowConvParams p{};
        p.src = tptr(l.src0); p.w = bptr(l.wrc_off); p.wfrag = 1; p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.H = l.ho; p.W = l.ho; p.C = l.c0; p.R = l.rowconv; p.relu = l.relu; p.dtype = P.dtype;
        e = launch_rowconv(p, s);
    } else if (l.fullk && P.dtype != 0) {
        FullK16Params p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.wfk_off); p.wtile = 1; p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.Ho = l.ho; p.Wo = l.ho; p.C0 = l.c0; p.C1 = l.c1; p.Cout = l.cout;
        p.up = l.up; p.relu = l.relu; p.stride = l.stride; p.dtype = P.dtype;
        e = launch_fullk16(p, l.fullk, s);
    } else if (l.fullk) {
        FullKParams p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.wfk_off); p.wtile = 1; p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = l.inorm ? nullptr : tptr(l.res); p.out = tptr(l.out);
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.Ho = l.ho; p.Wo = l.ho; p.C0 = l.c0; p.C1 = l.c1; p.Cout = l.cout;
        p.up = l.up; p.relu = l.inorm ? 0 : l.relu; p.stride = l.stride;
        if (l.splits == 2) {                       // K in two halves, combined in the launch
            p.split = 2;
            if (!l.c1) p.w = bptr(l.wfk2_off);     // a single source read as two half-sources
            p.partial = reinterpret_cast<float *>(h->ws + P.partial_offset);
            p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.counters_offset());
        }
        e = launch_fullk(p, l.fullk, s);
        if (e == hipSuccess && l.inorm) {          // InstanceNorm plans: H*W <= 256 here, the one-launch statistics route
            InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
            q.x = tptr(l.out); q.partial = nullptr; q.splits = 1; q.bias = nullptr; q.residual = tptr(l.res); q.relu = l.relu;
            q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
            e = launch_in_small(q, s);
        }
    } else {
        IgemmParams p{};
        p.src0 = tptr(l.src0); p.src1 = tptr(l.src1); p.w = bptr(l.w_off);
        p.scale = bptr(l.scale_off); p.shift = bptr(l.shift_off);
        p.residual = tptr(l.res); p.out = tptr(l.out);
        p.partial = reinterpret_cast<float *>(h->ws + P.partial_offset);
        p.B = batch; p.Hs = l.hs; p.Ws = l.hs; p.Ho = l.ho; p.Wo = l.ho;
        p.C0 = l.c0; p.C1 = l.c1; p.Cin = l.cin; p.Cout = l.cout;
        p.stride = l.stride; p.up = l.up; p.relu = l.relu;
        p.up4 = l.up4;
        p.Mout = batch * l.ho * l.ho;
        p.M = l.up4 ? batch * l.hs * l.hs : p.Mout;
        p.dtype = P.dtype;
        p.ktiles_total = (l.up4 ? 4 : 9) * l.cin / P.ktile_channels();
        p.splits = l.splits;
        p.ktiles_per_split = (p.ktiles_total + l.splits - 1) / l.splits;
        if (l.inorm) {
            // InstanceNorm follows: the conv writes its raw output (+ bias); residual add and ReLU move behind the normalisation
            float *st = reinterpret_cast<float *>(h->ws + P.stats_offset);
            const size_t slab = (size_t)batch * P.stats_groups_max;            // [B][groups][C] with C <= the plan's widest layer
            InstNormParams q{};
            q.three_pass = h->in_small_regs ? 0 : 1;
            q.x = tptr(l.out); q.residual = tptr(l.res); q.relu = l.relu;
            q.B = batch; q.hw = l.ho * l.ho; q.C = l.cout;
            q.psum = st; q.psq = st + slab * l.cout; q.pshift = st + 2 * slab * l.cout;
            q.mean = st + 3 * slab * l.cout; q.rstd = q.mean + (size_t)batch * l.cout;
            p.residual = nullptr; p.relu = 0;
            if (l.in_route == kInFused) {
                const int rhw = l.up4 ? l.hs * l.hs : l.ho * l.ho;
                q.rows_per_group = l.bm == 32 ? 32 : l.bm / 2;                  // rows of one wave
                q.groups = (l.up4 ? 4 : 1) * rhw / q.rows_per_group;
                p.psum = q.psum; p.psq = q.psq; p.pshift = q.pshift; p.in_groups = q.groups;
                e = launch_igemm(p, l.bm, l.bn, l.group, s);
                if (e == hipSuccess) e = launch_in_finalize(q, s);
                if (e == hipSuccess) e = launch_in_apply(q, s);
            } else {
                e = launch_igemm(p, l.bm, l.bn, l.group, s);
                q.splits = l.splits;
                q.partial = l.splits > 1 ? p.partial : nullptr;
                q.bias = bptr(l.shift_off);
                if (l.in_route == kInSmall) {
                    if (e == hipSuccess) e = launch_in_small(q, s);
                } else {
                    q.groups = (q.hw + 63) / 64; q.rows_per_group = 64;
                    if (e == hipSuccess) e = launch_in_reduce_stats(q, s);
                    if (e == hipSuccess) e = launch_in_finalize(q, s);
                    if (e == hipSuccess) e = launch_in_apply(q, s);
                }
            }
        } else {
            const bool fused = l.fused_splitk && h->fuse_splitk;
            if (fused) {
                p.tile_cnt = reinterpret_cast<unsigned *>(h->ws + P.counters_offset());
                p.slab_bytes = (size_t)l.splits * p.Mout * l.cout * sizeof(float);
            }
            p.xcd_force = P.igemm_xcd + 1;
            if (h->timing_part & 1) e = launch_igemm(p, l.bm, l.bn, l.group, s);
            if (e == hipSuccess && l.splits > 1 && !fused && (h->timing_part & 2)) e = launch_splitk_reduce(p, s);
        }
    }
    if (e != hipSuccess) return hipfail(e, ("launch " + l.name).c_str());
    return LSPF2F_OK;
}

// the arrival counters of the in-launch split-K combine must be zero before the first launch that uses them; every last arriver
// resets its own, so once per workspace binding is enough
static int clean_counters(lspf2f_handle *h, hipStream_t s)
{
    if (h->counters_clean) return LSPF2F_OK;
    const hipError_t e = hipMemsetAsync(h->ws + h->plan.counters_offset(), 0, Plan::kTileCounters * sizeof(unsigned), s);
    if (e != hipSuccess) return hipfail(e, "hipMemsetAsync (split-K arrival counters)");
    h->counters_clean = true;
    return LSPF2F_OK;
}

static int check_forward_args(lspf2f_handle *h, const float *feat, const float *cand, int cand_batch, float *out,
                              int batch)
{
    if (!h || !feat || !out) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "batch out of range (max_batch)");
    const int cand_nc = h->plan.input_nc - h->plan.feat_nc;
    if (cand_nc > 0 && !cand && !h->cand_cached)
        return fail(LSPF2F_ERR_INVALID_ARGUMENT, "cand_image is required (input_nc > feat_nc) unless lspf2f_set_candidates() was called");
    if (cand_nc > 0 && cand && cand_batch != 1 && cand_batch != batch)
        return fail(LSPF2F_ERR_SHAPE, "cand_batch must be 1 (broadcast) or equal to batch");
    if (!h->blob) return fail(LSPF2F_ERR_STATE, "weights not bound (lspf2f_bind_weights)");
    if (!h->ws) return fail(LSPF2F_ERR_STATE, "workspace not bound (lspf2f_bind_workspace)");
    h->plan.plan_batch(batch);
    if (h->ws_size < h->plan.act_bytes + h->plan.partial_bytes + h->plan.stats_bytes)
        return fail(LSPF2F_ERR_STATE, "workspace too small for this batch (lspf2f_workspace_bytes)");
    return LSPF2F_OK;
}

extern "C" {

int lspf2f_forward(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev,
                   int batch, void *hip_stream)
{
    return lspf2f_forward_ex(h, feat_dev, cand_dev, cand_batch, out_dev, nullptr, batch, hip_stream);
}

int lspf2f_forward_ex(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev,
                      unsigned char *out_u8_dev, int batch, void *hip_stream)
{
    if (!out_dev && !out_u8_dev) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "at least one of out_dev / out_u8_dev is required");
    int rc = check_forward_args(h, feat_dev, cand_dev, cand_batch, out_dev ? out_dev : reinterpret_cast<float *>(out_u8_dev), batch);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    rc = clean_counters(h, s);
    if (rc) return rc;

    // The ~80-150 launches of one forward are replayed from a hipGraph (captured once per
    // distinct set of pointers on a private stream), which removes the per-launch host cost that
    // dominates the <= 16x16 levels at batch 1.  If the caller is itself capturing `s`, or graphs
    // are disabled, launch eagerly into `s` instead.
    bool eager = !h->use_graph;
    if (!eager) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) eager = true;
    }
    if (eager) {
        for (const auto &l : h->plan.layers) {
            rc = run_layer(h, l, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch, s);
            if (rc) return rc;
        }
        return LSPF2F_OK;
    }

    GraphKey key{feat_dev, cand_dev, out_dev, out_u8_dev, h->ws, h->blob, cand_batch, batch};
    CachedGraph *g = nullptr;
    for (auto &c : h->graphs)
        if (c.exec && c.key == key) { g = &c; break; }
    if (!g) {
        if (!h->cap_stream) {
            const hipError_t e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
            if (e != hipSuccess) return hipfail(e, "hipStreamCreateWithFlags");
        }
        if (h->graphs.size() < kMaxCachedGraphs) h->graphs.emplace_back();
        g = &h->graphs[h->next_victim++ % h->graphs.size()];
        g->reset();
        // tail_prefetch: the byte range of the blob the <= 16x16 levels read (the forms this batch's plan takes), walked by a side branch of the graph
        const char *tp_lo = nullptr, *tp_hi = nullptr;
        if (h->tail_prefetch) {
            if (!h->side_stream && hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess) return fail(LSPF2F_ERR_HIP, "tail_prefetch: side stream");
            if (!h->ev_fork && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(LSPF2F_ERR_HIP, "tail_prefetch: event");
            if (!h->ev_join && hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) return fail(LSPF2F_ERR_HIP, "tail_prefetch: event");
            for (const auto &l : h->plan.layers) {
                if (l.kind != kIgemm || l.ho > 16) continue;
                const size_t wb = (size_t)(l.up4 ? 16 : 9) * l.cin * l.cout * h->plan.elt();
                const int64_t off = l.fullk ? ((h->plan.dtype == 0 && (l.stride == 2 || (l.splits == 2 && !l.c1))) ? l.wfk2_off : l.wfk_off) : l.bandconv ? l.wbc_off : l.w_off;
                if (off < 0) continue;
                const char *lo = h->blob + off, *hi = lo + wb;
                if (!tp_lo || lo < tp_lo) tp_lo = lo;
                if (!tp_hi || hi > tp_hi) tp_hi = hi;
            }
            if (tp_lo && h->tail_prefetch_mb > 0 && (size_t)(tp_hi - tp_lo) > ((size_t)h->tail_prefetch_mb << 20)) tp_hi = tp_lo + ((size_t)h->tail_prefetch_mb << 20);
        }
        hipError_t e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) return hipfail(e, "hipStreamBeginCapture");
        bool forked = false;
        int li = 0;
        for (const auto &l : h->plan.layers) {
            if (tp_lo && !forked && li >= h->tail_prefetch_at) {
                e = hipEventRecord(h->ev_fork, h->cap_stream);
                if (e == hipSuccess) e = hipStreamWaitEvent(h->side_stream, h->ev_fork, 0);
                if (e == hipSuccess) e = launch_touch_range(tp_lo, (size_t)(tp_hi - tp_lo) & ~(size_t)15, h->tail_prefetch, h->tail_prefetch_wgs,
                                                            reinterpret_cast<unsigned *>(h->ws + h->plan.counters_offset()), h->side_stream);
                if (e == hipSuccess) e = hipEventRecord(h->ev_join, h->side_stream);
                if (e != hipSuccess) { rc = hipfail(e, "tail_prefetch fork"); break; }
                forked = true;
            }
            rc = run_layer(h, l, feat_dev, cand_dev, cand_batch, out_dev, out_u8_dev, batch, h->cap_stream);
            if (rc) break;
            ++li;
        }
        if (forked) (void)hipStreamWaitEvent(h->cap_stream, h->ev_join, 0);      // the branch joins at the end: it never holds a layer up
        e = hipStreamEndCapture(h->cap_stream, &g->graph);
        if (rc) { g->reset(); return rc; }
        if (e != hipSuccess) { g->reset(); return hipfail(e, "hipStreamEndCapture"); }
        e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { g->reset(); return hipfail(e, "hipGraphInstantiate"); }
        g->key = key;
    }
    const hipError_t e = hipGraphLaunch(g->exec, s);
    if (e != hipSuccess) return hipfail(e, "hipGraphLaunch");
    return LSPF2F_OK;
}

int lspf2f_debug_poison(lspf2f_handle *h, unsigned char byte, void *hip_stream, unsigned *nonzero_counters)
{
    if (!h) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null handle");
    if (!h->ws) return fail(LSPF2F_ERR_STATE, "workspace not bound (lspf2f_bind_workspace)");
    const Plan &P = h->plan;
    if (h->ws_size < P.persistent_bytes()) return fail(LSPF2F_ERR_STATE, "workspace too small");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipError_t e = hipSuccess;
    // head of the workspace: [slot 0: per-person candidate cache][slot 1: per-forward candidate slot][arrival counters]; everything behind is scratch
    const size_t cc = P.cand_cache_bytes();
    if (!h->cand_cached) e = hipMemsetAsync(h->ws, byte, cc, s);
    if (e == hipSuccess) e = hipMemsetAsync(h->ws + cc, byte, cc, s);
    if (e == hipSuccess && h->ws_size > P.persistent_bytes()) e = hipMemsetAsync(h->ws + P.persistent_bytes(), byte, h->ws_size - P.persistent_bytes(), s);
    if (e != hipSuccess) return hipfail(e, "hipMemsetAsync (poison)");
    if (nonzero_counters) {
        *nonzero_counters = 0;
        if (h->counters_clean) {          // (before the first forward on this binding the counters are whatever the allocation held: nothing to check)
            std::vector<unsigned> host(Plan::kTileCounters);
            e = hipMemcpyAsync(host.data(), h->ws + P.counters_offset(), Plan::kTileCounters * sizeof(unsigned), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return hipfail(e, "read-back of the split-K arrival counters");
            for (unsigned v : host) *nonzero_counters += v != 0;
            return LSPF2F_OK;
        }
    }
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize (poison)");
    return LSPF2F_OK;
}

int lspf2f_set_candidates(lspf2f_handle *h, const float *cand_dev, void *hip_stream)
{
    if (!h) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null handle");
    if (!cand_dev) { h->cand_cached = nullptr; return LSPF2F_OK; }
    const Plan &P = h->plan;
    if (P.input_nc == P.feat_nc || P.feat_nc == 0) return fail(LSPF2F_ERR_UNSUPPORTED, "no candidate channels to cache");
    if (!h->blob || !h->ws) return fail(LSPF2F_ERR_STATE, "bind weights and workspace first");
    if (h->ws_size < P.cand_cache_bytes()) return fail(LSPF2F_ERR_STATE, "workspace too small");
    const LayerDesc &l = P.layers[0];
    FirstConvParams c{};
    c.feat = nullptr; c.cand = cand_dev; c.w = reinterpret_cast<const float *>(h->blob + l.w_off);
    c.out = reinterpret_cast<float *>(h->ws);
    c.B = 1; c.H = l.hs; c.W = l.hs; c.feat_nc = P.feat_nc; c.cand_nc = P.input_nc - P.feat_nc; c.cand_batch = 1;
    c.Cout = l.cout; c.ci_begin = P.feat_nc; c.ci_end = P.input_nc; c.base = nullptr; c.relu = 0;
    c.force_direct = h->first_direct;
    const hipError_t e = launch_first_conv(c, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return hipfail(e, "lspf2f_set_candidates launch");
    h->cand_cached = cand_dev;
    return LSPF2F_OK;
}

// Duration of a SUBSET of the forward's launches without host gaps: the selected layers are captured, in network order, into one graph
// that is replayed `reps` times between two ordinary events on `hip_stream` (events recorded by graph nodes cannot be timed on ROCm 7.2,
// so a kernel cannot be bracketed inside the replay of the whole forward).  part[i]: 0 = skip layer i, 1 = its main kernel(s) only (a
// split-K layer without its reduce launch), 2 = only its split-K reduce launch, 3 = everything the layer launches.  The layers read
// what the last forward left in the workspace (run one first); results are not meaningful, durations are.
int lspf2f_subset_timed(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch, float *out_dev, int batch,
                        void *hip_stream, const int *part, int reps, float *ms_per_replay, int *launches_per_replay)
{
    if (!part || !ms_per_replay || reps < 1) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null part / ms_per_replay, or reps < 1");
    int rc = check_forward_args(h, feat_dev, cand_dev, cand_batch, out_dev, batch);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if ((rc = clean_counters(h, s)) != 0) return rc;
    const int n = (int)h->plan.layers.size();
    if (!h->cap_stream) {
        const hipError_t e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return hipfail(e, "hipStreamCreateWithFlags");
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int launches = 0;
    hipError_t e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n && !rc && e == hipSuccess; ++i) {
        if (!part[i]) continue;
        h->timing_part = part[i];
        rc = run_layer(h, h->plan.layers[i], feat_dev, cand_dev, cand_batch, out_dev, nullptr, batch, h->cap_stream);
        ++launches;
    }
    h->timing_part = 3;
    const hipError_t e2 = hipStreamEndCapture(h->cap_stream, &graph);
    if (!rc && e != hipSuccess) rc = hipfail(e, "hipStreamBeginCapture");
    if (!rc && e2 != hipSuccess) rc = hipfail(e2, "hipStreamEndCapture");
    if (!rc && !launches) rc = fail(LSPF2F_ERR_INVALID_ARGUMENT, "empty selection");
    if (!rc && (e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) rc = hipfail(e, "hipGraphInstantiate");
    if (!rc && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail(LSPF2F_ERR_HIP, "hipEventCreate failed");
    if (!rc && (e = hipGraphLaunch(exec, s)) != hipSuccess) rc = hipfail(e, "hipGraphLaunch");          // warm-up replay
    if (!rc) {
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps && e == hipSuccess; ++r) e = hipGraphLaunch(exec, s);
        (void)hipEventRecord(e1, s);
        if (e != hipSuccess) rc = hipfail(e, "hipGraphLaunch");
    }
    if (!rc && (e = hipStreamSynchronize(s)) != hipSuccess) rc = hipfail(e, "hipStreamSynchronize");
    if (!rc) {
        float ms = 0.f;
        if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) rc = hipfail(e, "hipEventElapsedTime");
        *ms_per_replay = ms / (float)reps;
        if (launches_per_replay) *launches_per_replay = launches;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
}

int lspf2f_forward_timed(lspf2f_handle *h, const float *feat_dev, const float *cand_dev, int cand_batch,
                         float *out_dev, int batch, void *hip_stream, float *ms_per_layer)
{
    if (!ms_per_layer) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null ms_per_layer");
    int rc = check_forward_args(h, feat_dev, cand_dev, cand_batch, out_dev, batch);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if ((rc = clean_counters(h, s)) != 0) return rc;
    const int n = (int)h->plan.layers.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(LSPF2F_ERR_HIP, "hipEventCreate failed");
    (void)hipEventRecord(ev[0], s);
    for (int i = 0; i < n && !rc; ++i) {
        rc = run_layer(h, h->plan.layers[i], feat_dev, cand_dev, cand_batch, out_dev, nullptr, batch, s);
        (void)hipEventRecord(ev[i + 1], s);
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (!rc && e != hipSuccess) rc = hipfail(e, "hipStreamSynchronize");
    if (!rc)
        for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms_per_layer[i], ev[i], ev[i + 1]);
    for (auto &x : ev) (void)hipEventDestroy(x);
    return rc;
}

size_t lspf2f_conv3x3_scratch_bytes(int batch, int hs, int ws, int c0, int c1, int cout, int stride, int upsample,
                                    int tile_m, int tile_n, int split_k, int k_group, int dtype)
{
    const int ktc = dtype ? 64 : 32;
    if ((tile_m == 5001 || tile_m == 5002) && k_group == -1) {      // up-conv Winograd kernel: slabs at OUTPUT resolution + arrival counters
        const int sp = split_k > 0 ? split_k : 1;
        if (sp == 1) return 0;
        return (size_t)sp * batch * 4 * hs * hs * cout * sizeof(float) + (size_t)batch * (hs / 4) * (hs / 8) * (cout / (32 * (tile_m - 5000))) * sizeof(unsigned);
    }
    if (tile_m == 6001 && k_group == -1) {                           // Winograd F(4x4, 3x3) kernel: slabs + one arrival counter per (tile-block of 16 x 32 pixels, 32 channels)
        const int sp = split_k > 0 ? split_k : 1;
        if (sp == 1) return 0;
        return (size_t)sp * batch * hs * ws * cout * sizeof(float) + (size_t)batch * (hs / 16) * (ws / 32) * (cout / 32) * sizeof(unsigned);
    }
    if ((tile_m == 4001 || tile_m == 4002 || tile_m == 4003 || tile_m == 4004) && k_group == -1) {      // Winograd kernel: slabs + one arrival counter per (tile-block, channel group)
        if (tile_m == 4003 || tile_m == 4004) tile_m = 4001;         // (the register forms of nb = 1)
        const int sp = split_k > 0 ? split_k : 1;
        if (sp == 1) return 0;
        return (size_t)sp * batch * hs * ws * cout * sizeof(float) + (size_t)batch * (hs / 8) * (ws / 16) * (cout / (32 * (tile_m - 4000))) * sizeof(unsigned);
    }
    if (tile_m == 7064 || tile_m == 7032 || tile_m == 7164 || tile_m == 7132 || tile_m == 7116) return 0;      // patch-staged 16-bit kernels: no scratch
    (void)ws;
    const int ho = upsample ? 2 * hs : (stride == 2 ? (hs + 1) / 2 : hs);
    if ((tile_m == 16 || tile_m == 32) && tile_n == 16 && split_k == 2) {      // K-split full-K kernel: two partial tiles per tile + arrival counters
        const size_t pbk = tile_m / 16;
        return (size_t)batch * (ho * ho / (16 * pbk)) * (cout / 16) * (2 * pbk * 256 * sizeof(float) + sizeof(unsigned));
    }
    const int Mout = batch * ho * ho;
    const bool up4 = upsample == 2;
    const int M = up4 ? batch * hs * hs : Mout;
    int bm = tile_m, bn = tile_n, sp = split_k;
    if (!bm || !bn || !sp) {
        int a, b, c, g;
        const int kt = k_group == -4 ? 16 * (c0 / 4) / ktc : (up4 ? 4 : 9) * (c0 + c1) / ktc;      // -4: the 16 live (tap, quarter) pairs of a space-to-depth 4x4 / s2 conv
        choose_tiling(M, cout, kt, up4 ? 4 : 1, upsample == 1, dtype, &a, &b, &c, &g);
        if (!bm || !bn) { bm = a; bn = b; }
        if (!sp) sp = c;
    }
    (void)k_group;
    return sp > 1 ? (size_t)sp * Mout * cout * sizeof(float) : 0;
}

int lspf2f_conv3x3(const void *src0, const void *src1, const void *w_packed, const float *scale,
                   const float *shift, const void *residual, void *out, int batch, int hs, int ws, int c0,
                   int c1, int cout, int stride, int upsample, int relu, int tile_m, int tile_n, int split_k,
                   int k_group, int dtype, void *scratch, size_t scratch_bytes, void *hip_stream)
{
    if (dtype < 0 || dtype > 2) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
    const int ktc = dtype ? 64 : 32;
    if (!src0 || !w_packed || !out) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (hs != ws) return fail(LSPF2F_ERR_UNSUPPORTED, "square tensors only");
    const bool wino_tile = (tile_m == 4001 || tile_m == 4002 || tile_m == 4003 || tile_m == 4004 || tile_m == 5001 || tile_m == 5002 || tile_m == 6001) && k_group == -1;     // its K-step is 8 channels, checked by wino_supported()
    if (!wino_tile && ((c0 % ktc) || (c1 % ktc) || c0 <= 0 || c1 < 0 || (c1 > 0 && !src1)))
        return fail(LSPF2F_ERR_UNSUPPORTED, "channel counts must be multiples of 32 (fp32) / 64 (bf16, fp16)");
    if (cout % 4) return fail(LSPF2F_ERR_UNSUPPORTED, "cout must be a multiple of 4");
    if (stride != 1 && stride != 2) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "stride must be 1 or 2");
    if (upsample && stride != 1) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "upsample requires stride 1");
    if ((scale == nullptr) != (shift == nullptr)) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "scale and shift come together");
    if (k_group == -4) {
        // The masked 16-of-36 tap operand of a space-to-depth 4x4 / s2 conv ([cout][16][ci]) is an IMPLICIT-GEMM form only: checked here, before any other route
        // (Winograd, full-K, row / band kernels read w_packed as a dense 9-tap operand) can take the call
        if (dtype != 0 || stride != 1 || upsample || c1 || c0 % 4 || (c0 / 4) % ktc)
            return fail(LSPF2F_ERR_UNSUPPORTED, "k_group -4 (space-to-depth 4x4 / s2 taps): fp32, one source of 4 x ci channels, ci a multiple of 32");
        const bool gemm_tile = (tile_m == 0 && tile_n == 0) || ((tile_m == 32 || tile_m == 64 || tile_m == 128) && (tile_n == 64 || tile_n == 128));
        if (!gemm_tile) return fail(LSPF2F_ERR_UNSUPPORTED, "k_group -4 runs on the implicit GEMM only: tile 0x0 (planner's choice) or one of its tiles");
    }
    hipError_t e = hipSuccess;
    {
        // tile 1x1 forces the tiny-M single-launch kernel; tile 0x0 lets the planner's rule pick it
        SmallMParams q{};
        q.src = src0; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
        q.B = batch; q.Hs = hs; q.Ws = ws; q.Ho = upsample ? 2 * hs : (stride == 2 ? (hs + 1) / 2 : hs); q.Wo = q.Ho;
        q.Cin = c0; q.Cout = cout; q.stride = stride; q.up = upsample == 1; q.relu = relu; q.M = batch * q.Ho * q.Wo;
        q.dtype = dtype;
        const bool want = (tile_m == 1 && tile_n == 1) || (tile_m == 0 && tile_n == 0 && split_k == 0 && k_group != -4);
        if (want && c1 == 0 && upsample != 2 && smallm_supported(q)) {
            e = launch_smallm(q, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (tiny-M) launch");
            return LSPF2F_OK;
        }
        if (tile_m == 1 && tile_n == 1) return fail(LSPF2F_ERR_UNSUPPORTED, "tiny-M kernel does not support this shape");
    }
    {
        // tile (16 | 32) x 16 forces the full-K single-launch kernel; tile 0x0 + split 0 lets the planner's rule pick it
        const int ho_ = upsample ? 2 * hs : (stride == 2 ? (hs + 1) / 2 : hs);
        if (tile_m > 3000 && tile_n == 64 && k_group == -1) {     // 3000 + R: the sub-pixel up-conv row kernel, weights in its fragment order
            RowUpParams q{};
            q.src0 = src0; q.src1 = src1; q.w = w_packed; q.scale = scale; q.shift = shift; q.out = out;
            q.B = batch; q.H = hs; q.W = ws; q.R = tile_m - 3000; q.relu = relu; q.dtype = dtype;
            if (!rowup_layer(hs, c0, c1, cout, upsample == 2, dtype, false) || residual || hs != ws || !rowup_supported(q) || (q.R & 1))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the bf16 up-conv row kernel does not support this shape");
            e = launch_rowup(q, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (rowup) launch");
            return LSPF2F_OK;
        }
        if ((tile_m == 5001 || tile_m == 5002) && k_group == -1) {   // 5000 + nb: the up-conv Winograd kernel, w_packed in its fragment order; split_k = K splits
            const int sp = split_k > 0 ? split_k : 1, nbk = tile_m - 5000;
            const size_t slab = sp > 1 ? (size_t)sp * batch * 4 * hs * ws * cout * sizeof(float) : 0;
            WinoUpParams q{};
            q.src0 = static_cast<const float *>(src0); q.src1 = c1 ? static_cast<const float *>(src1) : nullptr; q.u = static_cast<const float *>(w_packed);
            q.scale = scale; q.shift = shift; q.out = static_cast<float *>(out);
            q.B = batch; q.Hs = hs; q.Ws = ws; q.C0 = c0; q.C1 = c1; q.N = cout; q.relu = relu; q.splits = sp;
            if (sp > 1) {
                const size_t ncnt = (size_t)batch * (hs / 4) * (ws / 8) * (cout / (32 * nbk));
                if (!scratch || scratch_bytes < slab + ncnt * sizeof(unsigned)) return fail(LSPF2F_ERR_STATE, "split-K scratch missing or too small");
                q.partial = static_cast<float *>(scratch);
                q.tile_cnt = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab);   // must be zero on entry; every launch leaves it zero
            }
            if (dtype != 0 || stride != 1 || !upsample || residual || hs != ws || !winoup_supported(q, nbk))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the up-conv Winograd kernel does not support this shape");
            e = launch_winoup(q, nbk, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (winoup) launch");
            return LSPF2F_OK;
        }
        if (tile_m == 6001 && k_group == -1) {                        // the Winograd F(4x4, 3x3) kernel, w_packed in the order of pack_wino4_weights(); split_k = K splits (0: 1)
            const int sp = split_k > 0 ? split_k : 1;
            const size_t slab = sp > 1 ? (size_t)sp * batch * hs * ws * cout * sizeof(float) : 0;
            const size_t ncnt = (size_t)batch * (hs / 16) * (ws / 32) * (cout / 32);
            WinoParams q{};
            q.src = static_cast<const float *>(src0); q.u = static_cast<const float *>(w_packed); q.scale = scale; q.shift = shift;
            q.residual = static_cast<const float *>(residual); q.out = static_cast<float *>(out);
            q.B = batch; q.H = hs; q.W = ws; q.C = c0; q.N = cout; q.relu = relu; q.splits = sp;
            if (sp > 1) {
                if (!scratch || scratch_bytes < slab + ncnt * sizeof(unsigned)) return fail(LSPF2F_ERR_STATE, "split-K scratch missing or too small");
                q.partial = static_cast<float *>(scratch);
                q.tile_cnt = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab);   // must be zero on entry; every launch leaves it zero
            }
#ifdef LSPF2F_WINO_STAMPS
            {   // stamps live behind slabs and counters when the caller's scratch has room for them
                const size_t used = (slab + ncnt * sizeof(unsigned) + 255) / 256 * 256;
                const size_t blocks = ncnt * (size_t)sp;
                if (scratch && scratch_bytes >= used + blocks * 4 * 8 * 8) q.stamps = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + used);
            }
#endif
            if (dtype != 0 || c1 != 0 || stride != 1 || upsample != 0 || hs != ws || !wino4_supported(q))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the Winograd F(4x4,3x3) kernel does not support this shape");
            e = launch_wino4(q, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (wino4) launch");
            return LSPF2F_OK;
        }
        if ((tile_m == 4001 || tile_m == 4002 || tile_m == 4003 || tile_m == 4004) && k_group == -1) {   // 4000 + nb: the Winograd kernel, w_packed in its fragment order; split_k = K splits (0: 1)
            const int ureg = tile_m == 4003 ? 1 : tile_m == 4004 ? 2 : 0;      // 4003: nb = 1 with the U fragments in registers (the form the plans take; 4004: four register sets), 4001: through LDS
            if (ureg) tile_m = 4001;
            const int sp = split_k > 0 ? split_k : 1;
            const size_t slab = sp > 1 ? (size_t)sp * batch * hs * ws * cout * sizeof(float) : 0;
            const size_t ncnt = (size_t)batch * (hs / 8) * (ws / 16) * (cout / (32 * (tile_m - 4000)));
            WinoParams q{};
            q.src = static_cast<const float *>(src0); q.u = static_cast<const float *>(w_packed); q.scale = scale; q.shift = shift;
            q.residual = static_cast<const float *>(residual); q.out = static_cast<float *>(out);
            q.B = batch; q.H = hs; q.W = ws; q.C = c0; q.N = cout; q.relu = relu; q.splits = sp; q.ureg = ureg;
            if (sp > 1) {
                if (!scratch || scratch_bytes < slab + ncnt * sizeof(unsigned)) return fail(LSPF2F_ERR_STATE, "split-K scratch missing or too small");
                q.partial = static_cast<float *>(scratch);
                q.tile_cnt = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab);   // must be zero on entry; every launch leaves it zero
            }
#ifdef LSPF2F_WINO_STAMPS
            {   // stamps live behind slabs and counters when the caller's scratch has room for them
                const size_t used = (slab + ncnt * sizeof(unsigned) + 255) / 256 * 256;
                const size_t blocks = ncnt * (size_t)sp;
                if (scratch && scratch_bytes >= used + blocks * 4 * 8 * 8) q.stamps = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + used);
            }
#endif
            if (dtype != 0 || c1 != 0 || stride != 1 || upsample != 0 || hs != ws || !wino_supported(q, tile_m - 4000))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the Winograd kernel does not support this shape");
            e = launch_wino(q, tile_m - 4000, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (wino) launch");
            return LSPF2F_OK;
        }
        if ((tile_m == 7164 || tile_m == 7132 || tile_m == 7116) && k_group != -4) {   // 7100 + tile width: the patch-staged kernel's sub-pixel up-conv form (conv3x3_patchup16); upsample == 2, w_packed = [4][cout][2][2][c0 + c1]
            PatchConvParams q{};
            q.src = src0; q.src1 = c1 ? src1 : nullptr; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
            q.B = batch; q.H = hs; q.W = ws; q.C = c0; q.C1 = c1; q.Cout = cout; q.relu = relu; q.dtype = dtype;
            if (stride != 1 || upsample != 2 || !patchup16_supported(q, tile_m - 7100, tile_n))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the patch-staged 16-bit up-conv kernel does not support this shape");
            e = launch_patchup16(q, tile_m - 7100, tile_n, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (patchup16) launch");
            return LSPF2F_OK;
        }
        if ((tile_m == 7064 || tile_m == 7032) && k_group != -4) {   // 7000 + tile width: the patch-staged 16-bit kernel (patch16.hip), tile_n = 128 | 64 channels per workgroup; igemm weight rows
            PatchConvParams q{};
            q.src = src0; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
            q.B = batch; q.H = hs; q.W = ws; q.C = c0; q.Cout = cout; q.relu = relu; q.dtype = dtype;
            q.deep = tile_n == 64;                  // tile_n 65 = 64 channels per workgroup in the FIRST form (copies in the load segment, 3-slot ring; A-B runs, the ablation and stamp builds)
            if (tile_n == 65) tile_n = 64;
#ifdef LSPF2F_ABLATE
            q.dbg = ablate_dbg();
#endif
#ifdef LSPF2F_PATCH_STAMPS
            if (scratch && scratch_bytes >= (size_t)4096 * 8 * 8 * 8) q.stamps = static_cast<unsigned long long *>(scratch);
#endif
            if (c1 != 0 || stride != 1 || upsample != 0 || !patch16_supported(q, tile_m - 7000, tile_n))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the patch-staged 16-bit kernel does not support this shape");
            e = launch_patch16(q, tile_m - 7000, tile_n, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (patch16) launch");
            return LSPF2F_OK;
        }
        if (tile_m == 2000 && k_group == -1) {     // the activation-stationary bf16 kernel of the 16x16 / 8x8 levels; weights in its fragment order
            BandConvParams q{};
            q.src = src0; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
            q.B = batch; q.W = hs; q.Cout = cout; q.relu = relu; q.dtype = dtype;
            if (!bandconv_layer(hs, c0, c1, cout, stride, upsample == 1, upsample == 2, dtype, false) || hs != ws || !bandconv_supported(q))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the bf16 band kernel does not support this shape");
            e = launch_bandconv(q, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (bandconv) launch");
            return LSPF2F_OK;
        }
        if (tile_m > 1000 && (tile_n == 64 || tile_n == 128) && k_group != -4) {   // 1000 + R: the weights-stationary bf16 kernel (tile_n channels in and out) with R output rows per strip
            RowConvParams q{};
            q.src = src0; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
            q.B = batch; q.H = hs; q.W = ws; q.C = tile_n; q.R = tile_m - 1000; q.relu = relu; q.dtype = dtype;
            q.wfrag = k_group == -1 ? 1 : 0;     // -1: w_packed is already in the row kernel's fragment order
            if (!rowconv_layer(hs, c0, c1, cout, stride, upsample == 1, upsample == 2, dtype, false) || c0 != tile_n || hs != ws || !rowconv_supported(q))
                return fail(LSPF2F_ERR_UNSUPPORTED, "the bf16 row kernel does not support this shape");
            e = launch_rowconv(q, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (rowconv) launch");
            return LSPF2F_OK;
        }
        if (dtype != 0 && (tile_m == 16 || tile_m == 32) && tile_n == 16 && k_group != -4) {      // the 16-bit full-K kernel (fullk16.hip); k_group -1: w_packed in its tile-blocked order
            FullK16Params q{};
            q.src0 = src0; q.src1 = c1 ? src1 : nullptr; q.w = w_packed; q.scale = scale; q.shift = shift; q.residual = residual; q.out = out;
            q.B = batch; q.Hs = hs; q.Ws = ws; q.Ho = ho_; q.Wo = ho_; q.C0 = c0; q.C1 = c1; q.Cout = cout;
            q.up = upsample == 1; q.relu = relu; q.stride = stride; q.dtype = dtype; q.wtile = k_group == -1 ? 1 : 0;
            if (upsample == 2 || hs != ws || !fullk16_supported(q, tile_m / 16)) return fail(LSPF2F_ERR_UNSUPPORTED, "the 16-bit full-K kernel does not support this shape");
            e = launch_fullk16(q, tile_m / 16, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (full-K, 16-bit) launch");
            return LSPF2F_OK;
        }
        int pb = 0;
        if ((tile_m == 16 || tile_m == 32) && tile_n == 16) pb = tile_m / 16;
        else if (tile_m == 0 && tile_n == 0 && split_k == 0 && k_group != -4)
            pb = fullk_choice(batch, hs, ho_, c0, c1, cout, stride, upsample == 1, upsample == 2, dtype);
        if (pb) {
            FullKParams q{};
            q.src0 = static_cast<const float *>(src0); q.src1 = c1 ? static_cast<const float *>(src1) : nullptr;
            q.w = static_cast<const float *>(w_packed); q.scale = scale; q.shift = shift;
            q.residual = static_cast<const float *>(residual); q.out = static_cast<float *>(out);
            q.B = batch; q.Hs = hs; q.Ws = ws; q.Ho = ho_; q.Wo = ho_; q.C0 = c0; q.C1 = c1; q.Cout = cout;
            q.up = upsample == 1; q.relu = relu; q.stride = stride;
            q.wtile = k_group == -1 ? 1 : 0;       // -1: w_packed is already in the full-K kernel's tile-blocked layout
            if (split_k == 2) {    // K halves: scratch = [2][tiles][256] floats + one zeroed counter per tile; a single source's w_packed is
                                                   // packed as two half-sources
                const size_t ntile = (size_t)batch * (ho_ * ho_ / (16 * pb)) * (cout / 16);     // 16 pb-pixel tiles x 16-channel slices
                const size_t slab = 2 * ntile * pb * 256 * sizeof(float);
                if (!scratch || scratch_bytes < slab + ntile * sizeof(unsigned))
                    return fail(LSPF2F_ERR_INVALID_ARGUMENT, "scratch too small for the K-split full-K kernel");
                q.split = 2; q.partial = static_cast<float *>(scratch);
                q.tile_cnt = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab);
            }
#ifdef LSPF2F_FULLK_STAMPS
            q.stamps = scratch_bytes >= (size_t)512 * 4 * 16 * 8 ? static_cast<unsigned long long *>(scratch) : nullptr;
#endif
            if (dtype != 0 || (stride != 1 && !(stride == 2 && q.split == 2)) || upsample == 2 || !fullk_supported(q, pb))
                return fail(LSPF2F_ERR_UNSUPPORTED, "full-K kernel does not support this shape");
            e = launch_fullk(q, pb, static_cast<hipStream_t>(hip_stream));
            if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 (full-K) launch");
            return LSPF2F_OK;
        }
    }
    IgemmParams p{};
    p.src0 = src0; p.src1 = c1 ? src1 : nullptr; p.w = w_packed; p.scale = scale; p.shift = shift;
    p.residual = residual; p.out = out;
    p.B = batch; p.Hs = hs; p.Ws = ws;
    p.Ho = upsample ? 2 * hs : (stride == 2 ? (hs + 1) / 2 : hs);
    p.Wo = p.Ho;
    if (upsample < 0 || upsample > 2) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "upsample must be 0, 1 or 2");
    p.C0 = c0; p.C1 = c1; p.Cin = c0 + c1; p.Cout = cout; p.stride = stride; p.relu = relu;
    p.up = upsample == 1; p.up4 = upsample == 2;
    p.Mout = batch * p.Ho * p.Wo;
    p.M = p.up4 ? batch * hs * ws : p.Mout;
    p.dtype = dtype;
    p.ktiles_total = (p.up4 ? 4 : 9) * p.Cin / ktc;
    if (k_group == -4) {
        // Conv2d(k4, s2, p1) as a 3x3 conv on the space-to-depth image (c0 = 4 ci, channel = (dy * 2 + dx) * ci + c): tap row ty reads sub-rows {1}, {0, 1}, {0} for
        // ty = 0, 1, 2 (columns alike), so 16 of the 36 (tap, quarter) pairs carry weights; w_packed = [cout][those 16 in tap-major order][ci]
        if (dtype != 0 || stride != 1 || upsample || c1 || c0 % 4 || (c0 / 4) % ktc) return fail(LSPF2F_ERR_UNSUPPORTED, "k_group -4 (space-to-depth 4x4 / s2 taps): fp32, one source of 4 x ci channels, ci a multiple of 32");
        static const unsigned sub[3] = {2u, 3u, 1u};          // live sub-rows of tap row 0, 1, 2 as a bit set over dy
        p.kmask = 0;
        for (int ty = 0; ty < 3; ++ty)
            for (int tx = 0; tx < 3; ++tx)
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx)
                        if (((sub[ty] >> dy) & 1u) && ((sub[tx] >> dx) & 1u)) p.kmask |= 1ull << ((ty * 3 + tx) * 4 + dy * 2 + dx);
        p.kblk = c0 / 4;
        p.ktiles_total = 16 * p.kblk / ktc;
        k_group = 0;
    }
    int bm = tile_m, bn = tile_n, sp = split_k, grp = k_group;
    {
        int a, b, c, g;
        choose_tiling(p.M, cout, p.ktiles_total, p.up4 ? 4 : 1, p.up != 0, dtype, &a, &b, &c, &g);
        if (!bm || !bn) { bm = a; bn = b; if (!grp) grp = g; }
        if (!sp) sp = c;
        if (!grp) grp = 1;
    }
    if (!igemm_group_supported(bm, bn, grp, p.up != 0)) return fail(LSPF2F_ERR_UNSUPPORTED, "tile shape / k_group not instantiated");
    if (sp < 1 || sp > p.ktiles_total) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "bad split_k");
    p.ktiles_per_split = (p.ktiles_total + sp - 1) / sp;
    sp = (p.ktiles_total + p.ktiles_per_split - 1) / p.ktiles_per_split;
    p.splits = sp;
    if (sp > 1) {
        if (!scratch || scratch_bytes < (size_t)sp * p.Mout * cout * sizeof(float))
            return fail(LSPF2F_ERR_STATE, "split-K scratch missing or too small");
        p.partial = static_cast<float *>(scratch);
    }
#ifdef LSPF2F_ABLATE
    p.dbg = ablate_dbg();   // tools/ablate.sh builds only
#endif
#ifdef LSPF2F_IGEMM_STAMPS
    if (sp == 1 && scratch && scratch_bytes >= (size_t)2048 * 4 * 16 * 8) p.stamps = static_cast<unsigned long long *>(scratch);
#endif
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    e = launch_igemm(p, bm, bn, grp, s);
    if (e == hipSuccess && sp > 1) e = launch_splitk_reduce(p, s);
    if (e != hipSuccess) return hipfail(e, "lspf2f_conv3x3 launch");
    return LSPF2F_OK;
}


// scratch of lspf2f_wino_chain: [split-K slabs][one arrival counter per (tile-block, channel group)][(nlayers - 1) x tile-blocks gate counters][1 give-up word]
static void wino_chain_scratch_layout(int nlayers, int batch, int hs, int c, int sp, size_t *slab, size_t *ncnt, size_t *narr)
{
    *slab = sp > 1 ? (size_t)sp * batch * hs * hs * c * sizeof(float) : 0;
    *ncnt = sp > 1 ? (size_t)batch * (hs / 8) * (hs / 16) * (c / 32) : 0;
    *narr = (size_t)(nlayers > 1 ? nlayers - 1 : 0) * batch * (hs / 8) * (hs / 16) * kWinoArriveStride;
}

size_t lspf2f_wino_chain_scratch_bytes(int nlayers, int batch, int hs, int c, int split_k)
{
    if (nlayers < 1 || batch < 1 || hs < 16 || c < 32) return 0;
    size_t slab, ncnt, narr;
    wino_chain_scratch_layout(nlayers, batch, hs, c, split_k > 0 ? split_k : 1, &slab, &ncnt, &narr);
    return slab + (ncnt + narr + 1) * sizeof(unsigned);
}

int lspf2f_wino_chain(int nlayers, const float *const *src, const float *const *u_packed, const float *const *scale, const float *const *shift,
                      const float *const *residual, float *const *out, const int *relu, int batch, int hs, int c, int split_k, int mode,
                      void *scratch, size_t scratch_bytes, void *hip_stream)
{
    if (nlayers < 1 || nlayers > kWinoChainMax) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "wino_chain: 1..4 layers");
    if (!src || !u_packed || !scale || !shift || !residual || !out || !relu) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "null argument");
    if (mode < 0 || mode > 4) return fail(LSPF2F_ERR_INVALID_ARGUMENT, "wino_chain: mode 0 (one launch per layer), 1 (one launch), 2 (one one-layer chain launch per layer), 3 (one launch, sc1 loads), 4 (one launch, plain loads and NO acquire: measurements only)");
    const int sp = split_k > 0 ? split_k : 1;
    size_t slab, ncnt, narr;
    wino_chain_scratch_layout(nlayers, batch, hs, c, sp, &slab, &ncnt, &narr);
    if (!scratch || scratch_bytes < slab + (ncnt + narr + 1) * sizeof(unsigned)) return fail(LSPF2F_ERR_STATE, "wino_chain: scratch missing or too small (lspf2f_wino_chain_scratch_bytes)");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    WinoParams q{};
    q.B = batch; q.H = hs; q.W = hs; q.C = c; q.N = c; q.splits = sp; q.ureg = 1; q.out_wt = 1; q.prio = 1;
    if (sp > 1) {
        q.partial = static_cast<float *>(scratch);
        q.tile_cnt = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab);
    }
    unsigned *arrive = reinterpret_cast<unsigned *>(static_cast<char *>(scratch) + slab) + ncnt;
    if (!wino_supported(q, 1)) return fail(LSPF2F_ERR_UNSUPPORTED, "the Winograd kernel does not support this shape");
    hipError_t e = hipSuccess;
    if (mode == 0) {
        for (int k = 0; k < nlayers && e == hipSuccess; ++k) {
            WinoParams p = q;
            p.src = src[k]; p.u = u_packed[k]; p.scale = scale[k]; p.shift = shift[k]; p.residual = residual[k]; p.out = out[k]; p.relu = relu[k];
            e = launch_wino(p, 1, s);
        }
    } else {
        WinoChainParams pc{};
        pc.c = q;
        pc.arrive = arrive; pc.fail = arrive + narr;
#ifdef LSPF2F_WINO_STAMPS
        {   // stamps live behind the counters when the caller's scratch has room for them: [blocks][4 waves][8]
            const size_t used = (slab + (ncnt + narr + 1) * sizeof(unsigned) + 255) / 256 * 256;
            const size_t blocks = (size_t)nlayers * batch * (hs / 8) * (hs / 16) * (c / 32) * sp;
            if (scratch_bytes >= used + blocks * 4 * 8 * 8) pc.c.stamps = reinterpret_cast<unsigned long long *>(static_cast<char *>(scratch) + used);
        }
#endif
        auto fill = [&](int slot, int k) {
            WinoChainLayer &l = pc.L[slot];
            l.src = src[k]; l.u = u_packed[k]; l.scale = scale[k]; l.shift = shift[k]; l.residual = residual[k]; l.out = out[k]; l.relu = relu[k];
        };
        pc.sc1_loads = mode == 3 ? 1 : mode == 4 ? 2 : 0;
        if (mode == 1 || mode == 3 || mode == 4) {
            pc.nlayers = nlayers;
            for (int k = 0; k < nlayers; ++k) fill(k, k);
            if (!wino_chain_supported(pc)) return fail(LSPF2F_ERR_UNSUPPORTED, "wino_chain: layer k must read layer k - 1's output, and no output may alias another tensor of the chain");
            e = launch_wino_chain(pc, s);
        } else {
            pc.nlayers = 1;
            for (int k = 0; k < nlayers && e == hipSuccess; ++k) { fill(0, k); e = launch_wino_chain(pc, s); }
        }
    }
    if (e != hipSuccess) return hipfail(e, "lspf2f_wino_chain launch");
    return LSPF2F_OK;
}

}  // extern "C"
