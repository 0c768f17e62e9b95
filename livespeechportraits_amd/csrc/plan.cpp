#include "plan.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace lspf2f {

static const double kBnEps = 1e-5;   // nn.BatchNorm2d default (networks.py never overrides it)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// fp32 -> 16-bit storage, round to nearest even: bf16 (dtype 1) or IEEE binary16 (dtype 2)
static uint16_t narrow16(float f, int dtype)
{
    if (dtype == 2) {
        const _Float16 h = (_Float16)f;
        uint16_t u;
        std::memcpy(&u, &h, 2);
        return u;
    }
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

void choose_tiling(int M, int N, int ktiles, int par, bool up9, int dtype, int *bm_out, int *bn_out, int *splits_out, int *group_out)
{
    // par = independent GEMM slices per launch (4 output parities in sub-pixel up-conv form)
    // up9: the 9-tap upsample gather form is instantiated for 64x64 (G = 1, 4) and 32x64 (G = 4) only
    //
    // Rules distilled from tools/tune_conv.py sweeps on MI355X (profiles/r01_tune_*.txt).  The kernel is
    // MFMA-bound and its throughput is flat (+-5 %) across tile shapes, so what matters is having
    // >= ~2 workgroups per CU (256 CUs): 64x128 when that still leaves >= 384 workgroups, else 64x64,
    // else 64x64 + split-K up to ~768 (fp32) / ~512 (bf16) workgroups with >= 4 K-tiles per split.  128-row tiles never won.
    const int want = 384;
    int bm = 64, bn = 64, splits = 1, group = 1;
    if (M <= 32) {
        // <= 4x4 spatial at batch 1 that the tiny-M kernel did not take: weight streaming, 4 K-tiles per
        // workgroup fetched in one step
        bm = 32; bn = 64; group = 4;
        splits = std::max(1, (ktiles + 3) / 4);
    } else if (M <= 64) {
        group = 4;
        splits = std::max(1, (ktiles + 3) / 4);
    } else {
        const long t128 = (long)par * ((M + 63) / 64) * ((N + 127) / 128);
        const long t64 = (long)par * ((M + 63) / 64) * ((N + 63) / 64);
        long tiles = t64;
        if (!up9 && N >= 128 && t128 >= want) { bn = 128; tiles = t128; }
        if (dtype != 0 && !up9) {
            // bf16: an MFMA step is 16x shorter, so the per-K-tile overhead (DMA issue, barrier) dominates
            // and the biggest tile that still gives every CU a workgroup wins by 10-35 %
            // (profiles/r01_tune_conv_bf16_b8.txt): 128 rows x (128 | 64) columns
            const int wn = N >= 128 ? 128 : 64;
            const long tbig = (long)par * ((M + 127) / 128) * ((N + wn - 1) / wn);
            if (tbig >= 256) { bm = 128; bn = wn; tiles = tbig; }
            // ... except where that tile count needs a K-split (256 .. 511 tiles) although 64 x 128 tiles fill the chip unsplit and K is short (<= 36 K-tiles: L3.down of an
            // 8-frame plan, 256 -> 512 at stride 2): 36.6 us with 128x128 x 2 splits + splitk_reduce, 32.8 us with 64x128 unsplit (profiles/r05_bf16_tilings.txt); at K = 4608
            // the 128-row tile keeps its lead (53.4 vs 59.2 us)
            if (tbig >= 256 && tbig < 512 && ktiles <= 36 && N >= 128 && t128 >= 512) { bm = 64; bn = 128; tiles = t128; }
            // (the sub-pixel up-conv in the same position -- L4.up of an 8-frame plan, 256 tiles of 128x128 x 2 K-splits + reduce -- measures 49.0 us on unsplit 64x128 tiles against 49.5 + 6.8 us
            // back to back, but the FORWARD is 0.8-1 % slower with it, A-B-A-B: profiles/r06_patch16_ab.txt (9); not taken)
        }
        if (tiles < (bm == 128 ? 512 : want)) {
            // workgroups to aim for when splitting K.  fp32: 768 beats 512 (batch 1 `large` 388.8-389.5 -> 392.3-393.2 frames/s,
            // `normal` 633.5-635.0 -> 638.5-640.9, batch 8 +0.4 %; 1024 is slightly below 768).  Splitting the 512-tile layers of the 128x128 level
            // in two as well (2 -> 4 waves per SIMD) was measured in round 2: the K loop gains what the 8.4-MB slabs cost, 52 -> 55 us per layer.  bf16 keeps 512 (768: batch 8
            // `large` -0.7 %, `normal` no change).  Same session, 3 runs per arm.
            // 16-bit, 64-row tiles, M >= 2048 and a long K (>= 72 K-tiles): L4.down / L5.up of an 8-frame plan, the two 16x16-level layers conv3x3_fullk16 does not take.
            // 256 tiles x 2 splits measured 46.7 / 29.8 us per layer, x 4 splits 43.1 / 27.3 (tools/sessions/gpu_r5_s7.sh, profiles/r05_bf16_tilings.txt); at 4 frames
            // (M = 1024) the finer split loses, so the rule stops there
            const bool long_k16 = dtype != 0 && bm == 64 && M >= 2048 && ktiles >= 72;
            const int target = dtype == 0 ? 768 : long_k16 ? 1024 : 512;
            splits = (int)((target + tiles - 1) / tiles);
            splits = std::min(splits, std::max(1, ktiles / 4));   // keep >= 4 K-tiles per split
            const int per = (ktiles + splits - 1) / splits;      // make every split non-empty
            splits = (ktiles + per - 1) / per;
        }
    }
    *bm_out = bm;
    *bn_out = bn;
    *splits_out = splits;
    *group_out = group;
}

int wino_choice(int batch, int ho, int cin, int cout, int *splits_out)
{
    // A workgroup = 32 Winograd tiles (8 x 16 output pixels) x 32 nb channels over the whole K (or a slice).  Two channel blocks per wave halve
    // the fragment reads per MFMA but cost half the workgroups: used when that still leaves two workgroups per CU.  Below ~1.5 workgroups
    // per CU the input channels are split (combined inside the launch), keeping >= 4 eight-channel steps per slice.
    const long ntb = (long)batch * (ho / 8) * (ho / 16);
    // 16x16 frames (2 tile-blocks each): measured 18.9 us against 17.4 us for the full-K kernel at 1 frame, 48.9 against 89.4 us for
    // igemm + split-K at 8 frames (tools/wino_sweep.py); the full-K kernel covers <= 2 frames, Winograd takes over from 4
    if (ho < 32 && ntb * (cout / 32) < 128) return 0;
    int nb = (cout % 64 == 0 && ntb * (cout / 64) >= 512) ? 2 : 1;
    const long wgs = ntb * (cout / (32 * nb));
    const int steps = cin / 8;
    int splits = 1;
    if (wgs < 384) {
        splits = (int)((512 + wgs - 1) / wgs);
        splits = std::min(splits, std::min(8, std::max(1, steps / 4)));
        const int per = (steps + splits - 1) / splits;
        splits = (steps + per - 1) / per;                  // every slice non-empty
    }
    WinoParams q{};
    q.B = batch; q.H = ho; q.W = ho; q.C = cin; q.N = cout; q.splits = 1;
    if (!wino_supported(q, nb)) return 0;
    if (splits > 1 && wgs > (long)Plan::kTileCounters) return 0;
    *splits_out = splits;
    return nb;
}

int wino4_choice(int batch, int ho, int cin, int cout, int *splits_out)
{
    // A workgroup = 32 tiles of 4 x 4 outputs (16 x 32 pixels) x 32 channels, ONE per CU (its ring slots take 112 KB of LDS): below ~one
    // workgroup per CU the input channels are split (combined inside the launch), keeping >= 4 eight-channel steps per slice.
    const long wgs = (long)batch * (ho / 16) * (ho / 32) * (cout / 32);
    const int steps = cin / 8;
    int splits = 1;
    if (wgs < 192) {
        splits = (int)((256 + wgs - 1) / wgs);
        splits = std::min(splits, std::min(8, std::max(1, steps / 4)));
        const int per = (steps + splits - 1) / splits;
        splits = (steps + per - 1) / per;                  // every slice non-empty
    }
    WinoParams q{};
    q.B = batch; q.H = ho; q.W = ho; q.C = cin; q.N = cout; q.splits = 1;
    if (!wino4_supported(q)) return 0;
    if (splits > 1 && wgs > (long)Plan::kTileCounters) return 0;
    *splits_out = splits;
    return 1;
}

int winoup_choice(int batch, int hs, int cin, int cout, int *splits_out, int force_nb, int target, bool small_level)
{
    // the up-conv that writes 16 x 16 (1024 -> 512 from 8 x 8) runs here from two frames up (tools/time_conv.py, per layer): 28.3 against 47.8 us for the full-K kernel at
    // two frames, 39.2 against 93.6 (implicit GEMM over the upsampled 9 taps) at four, 58.1 against 170.3 at eight.  One frame keeps the full-K kernel: 21.3 against 25.1 us
    // when timed alone, but the whole forward is 0.6 % SLOWER with it (653.2 vs 657.0 frames/s, A-B-A-B of two libraries: profiles/r04_fullk_small_levels_batch.txt)
    if (small_level && batch < 2) return 0;
    // a workgroup = 32 source pixels (8 x 16 output pixels) x 32 nb channels, THREE waves: two such workgroups leave a CU's four SIMDs with 2, 2, 1, 1
    // waves, four give every SIMD three.  So: one channel block per wave (115 registers, 28 KB of LDS: five fit) and K splits -- >= 8 eight-channel
    // steps each -- until there are ~4 workgroups per CU.  Measured, `large` fp32 (A-B-A-B, one session): nb 1 / 1024 workgroups 610.6 frames/s,
    // nb 1 / 512 605.6, nb 2 / 512 602.5, nb 2 / 1024 580.8, nb 1 / 1536 602.0, nb 1 / 2048 599.8; batch 8: nb 1 967.4, nb 2 945.8
    const long ntb = (long)batch * (hs / 4) * (hs / 8);
    const int nb = force_nb ? force_nb : 1;
    if ((nb != 1 && nb != 2) || cout % (32 * nb)) return 0;     // a forced nb the shape cannot take: the implicit GEMM keeps the layer
    const long wgs = ntb * (cout / (32 * nb));
    if (wgs <= 0) return 0;
    {   // the kernel's own shape limits (2-GiB buffer offsets, extents), like wino_choice: an unsupported shape falls back instead of failing at launch
        WinoUpParams q{};
        q.B = batch; q.Hs = hs; q.Ws = hs; q.C0 = cin; q.C1 = 0; q.N = cout; q.splits = 1;
        if (!winoup_supported(q, nb)) return 0;
    }
    const int steps = cin / 8;
    int splits = 1;
    if (wgs < target * 3 / 4) {
        splits = (int)((target + wgs - 1) / wgs);
        splits = std::min(splits, std::min(8, std::max(1, steps / 8)));
        const int per = (steps + splits - 1) / splits;
        splits = (steps + per - 1) / per;
    }
    if (splits > 1 && wgs > (long)Plan::kTileCounters) return 0;
    *splits_out = splits;
    return nb;
}

static void level_channels(int depth, int ngf, int input_nc, int output_nc, int *cin, int *inner, int *cout)
{
    // networks.py:557-570: innermost and the num_downs-5 middle blocks are ngf*8 -> ngf*8, then
    // ngf*4 -> ngf*8, ngf*2 -> ngf*4, ngf -> ngf*2, outermost output_nc/input_nc -> ngf.
    if (depth == 0) { *cin = input_nc; *inner = ngf; *cout = output_nc; return; }
    const int mo = std::min(1 << (depth - 1), 8), mi = std::min(1 << depth, 8);
    *cin = ngf * mo; *inner = ngf * mi; *cout = ngf * mo;
}

namespace {
struct Builder {
    Plan &p;
    explicit Builder(Plan &plan) : p(plan) {}

    void add_param(const std::string &key, std::vector<int64_t> dims)
    {
        ParamDesc d;
        d.key = key;
        d.dims = std::move(dims);
        p.param_index[key] = (int)p.params.size();
        p.params.push_back(std::move(d));
    }
    void add_bn(const std::string &key, int c)
    {
        add_param(key + ".weight", {c});
        add_param(key + ".bias", {c});
        add_param(key + ".running_mean", {c});
        add_param(key + ".running_var", {c});
    }
    int add_tensor(const std::string &name, int c, int h, int def)
    {
        TensorDesc t;
        t.name = name; t.c = c; t.h = h; t.def = def;
        p.tensors.push_back(t);
        return (int)p.tensors.size() - 1;
    }
    void use(int t, int layer) { if (t >= 0) p.tensors[t].last_use = std::max(p.tensors[t].last_use, layer); }

    int add_layer(LayerDesc l)
    {
        const int idx = (int)p.layers.size();
        use(l.src0, idx); use(l.src1, idx); use(l.res, idx);
        p.layers.push_back(std::move(l));
        return idx;
    }

    // ResidualBlock (networks.py:650-675): conv-BN-ReLU-conv-BN, += x, ReLU
    int res_block(const std::string &lname, const std::string &key, int x, int c, int h)
    {
        // with norm_layer = InstanceNorm2d the block keeps bias=False convs (networks.py:662-668) and the norms own no tensors
        const bool in = p.norm == 1;
        add_param(key + ".block.0.weight", {c, c, 3, 3});
        if (!in) add_bn(key + ".block.1", c);
        add_param(key + ".block.3.weight", {c, c, 3, 3});
        if (!in) add_bn(key + ".block.4", c);
        LayerDesc a;
        a.name = lname + ".a"; a.kind = kIgemm; a.src0 = x; a.cin = a.c0 = c; a.cout = c;
        a.hs = a.ho = h; a.stride = 1; a.relu = true;
        a.wkey = key + ".block.0.weight"; a.bnkey = in ? "" : key + ".block.1"; a.inorm = in;
        a.out = add_tensor(a.name, c, h, (int)p.layers.size());
        const int ta = a.out;
        add_layer(a);
        LayerDesc b;
        b.name = lname + ".b"; b.kind = kIgemm; b.src0 = ta; b.res = x; b.cin = b.c0 = c; b.cout = c;
        b.hs = b.ho = h; b.stride = 1; b.relu = true; b.residual = true;
        b.wkey = key + ".block.3.weight"; b.bnkey = in ? "" : key + ".block.4"; b.inorm = in;
        b.out = add_tensor(b.name, c, h, (int)p.layers.size());
        const int tb = b.out;
        add_layer(b);
        return tb;
    }

    // one skip block; x = tensor id of its input (-1: the API input), returns the tensor id of
    // model(x)'s up-side output (the second half of cat([x, model(x)])); -1 for the outermost.
    int level(int depth, const std::string &pfx, int x, int h_in)
    {
        const bool outer = depth == 0, innermost = depth == p.num_downs - 1;
        int cin, inner, cout;
        level_channels(depth, p.ngf, p.input_nc, p.output_nc, &cin, &inner, &cout);
        const int hd = h_in / 2;
        const std::string L = "L" + std::to_string(depth);
        int i = 0;
        auto key = [&](int idx) { return pfx + ".model." + std::to_string(idx); };

        LayerDesc d;
        d.name = L + ".down"; d.kind = outer ? kFirstConv : kIgemm; d.src0 = x;
        d.cin = d.c0 = cin; d.cout = inner; d.hs = h_in; d.ho = hd; d.stride = 2; d.relu = true;
        d.wkey = key(i) + ".weight";
        add_param(d.wkey, {inner, cin, 3, 3});
        const bool in = p.norm == 1;          // use_bias = norm_layer == nn.InstanceNorm2d (networks.py:590)
        if (in) { d.biaskey = key(i) + ".bias"; add_param(d.biaskey, {inner}); }
        ++i;
        if (!outer && !innermost) {
            if (in) d.inorm = true; else { d.bnkey = key(i); add_bn(d.bnkey, inner); }
            ++i;
        }
        ++i;  // ReLU
        d.out = add_tensor(d.name, inner, hd, (int)p.layers.size());
        int cur = d.out;
        add_layer(d);
        for (int r = 0; r < p.nres; ++r) { cur = res_block(L + ".d.res" + std::to_string(r), key(i), cur, inner, hd); ++i; }

        int below = -1;
        if (!innermost) { below = level(depth + 1, key(i), cur, hd); ++i; }
        ++i;  // Upsample

        LayerDesc u;
        u.name = L + ".up"; u.kind = outer ? kLastConv : kIgemm; u.src0 = cur; u.src1 = below;
        u.c0 = inner; u.c1 = innermost ? 0 : inner; u.cin = u.c0 + u.c1; u.cout = cout;
        u.hs = hd; u.ho = h_in; u.stride = 1; u.up = true; u.concat = !innermost;
        u.relu = !outer; u.tanh_out = outer;
        u.wkey = key(i) + ".weight";
        add_param(u.wkey, {cout, u.cin, 3, 3});
        if (in) { u.biaskey = key(i) + ".bias"; add_param(u.biaskey, {cout}); }
        ++i;
        if (!outer) {
            if (in) u.inorm = true; else { u.bnkey = key(i); add_bn(u.bnkey, cout); }
            i += 2;
        }
        if (outer) { u.out = -1; add_layer(u); return -1; }
        u.out = add_tensor(u.name, cout, h_in, (int)p.layers.size());
        cur = u.out;
        add_layer(u);
        for (int r = 0; r < p.nres; ++r) { cur = res_block(L + ".u.res" + std::to_string(r), key(i), cur, cout, h_in); ++i; }
        return cur;
    }
};
}  // namespace

namespace {
struct BatchLayout { size_t act_bytes, partial_bytes, stats_bytes; int groups_max; };
BatchLayout layout_for(const Plan &p, int batch, std::vector<size_t> *offsets, std::vector<LayerDesc> *tiled);
}  // namespace

std::string Plan::build(int variant_, int input_nc_, int feat_nc_, int output_nc_, int ngf_, int num_downs_,
                        int size_, bool keep, int dtype_, int norm_, int max_batch_forms)
{
    if (norm_ != 0 && norm_ != 1) return "norm must be 0 (BatchNorm2d, eval) or 1 (InstanceNorm2d)";
    if (norm_ == 1 && dtype_ != 0) return "the InstanceNorm variant is fp32 only";
    norm = norm_;
    if (dtype_ < 0 || dtype_ > 2) return "dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)";
    if (dtype_ != 0 && ngf_ % 64 != 0) return "16-bit storage needs ngf % 64 == 0 (a K-tile is 64 channels)";
    dtype = dtype_;
    if (variant_ != 0 && variant_ != 1)
        return "variant must be 0 (normal) or 1 (large); the 'small' U-Net (networks.py:680-769) is not supported";
    if (ngf_ <= 0 || ngf_ % 32 != 0) return "ngf must be a positive multiple of 32 (MFMA K-tile = 32 channels)";
    if (num_downs_ < 5 || num_downs_ > 12) return "num_downs must be in [5, 12] (networks.py:563)";
    if (size_ <= 0 || size_ % (1 << num_downs_) != 0) return "frame size must be a multiple of 2**num_downs";
    if (output_nc_ < 1 || output_nc_ > 4) return "output_nc must be in [1, 4]";
    if (input_nc_ < 1 || feat_nc_ < 0 || feat_nc_ > input_nc_) return "bad input_nc / feat_nc";
    variant = variant_; nres = variant_ == 1 ? 2 : 1;
    input_nc = input_nc_; feat_nc = feat_nc_; output_nc = output_nc_; ngf = ngf_; num_downs = num_downs_;
    size = size_; keep_intermediates = keep;
    layers.clear(); tensors.clear(); params.clear(); param_index.clear();
    Builder b(*this);
    b.level(0, "netG.model", -1, size);

    // up-convs of the compute-bound levels run in sub-pixel form: 4 parities x 2x2 taps (16/9 of the 9-tap weight bytes, 4/9 of the FLOPs); the
    // weight-streaming-bound small levels keep the 9-tap gather form.
    for (auto &l : layers) {
        l.up4 = l.kind == kIgemm && l.up && l.ho >= kUp4MinExtent;
        if (l.up4) l.up = false;
    }
    // Blob layout.  First every weight form some kernel of this dtype could read; then, when the handle's largest batch is known, the layout
    // is redone with ONLY the forms the plans of batch 1 .. max_batch take (the planner is run for each of them): a layer that every such plan
    // runs on the Winograd kernel does not carry its 9-tap rows, the 16x16 layers of a one-frame handle carry neither rows nor G g G^T, ...
    // `large` fp32 at 512x512: 1.22 GB with every form, 0.62 GB for max_batch = 1, 0.80 GB for max_batch = 8.
    assign_offsets(nullptr);
    if (max_batch_forms > 0 && !keep_all_forms) {
        std::vector<unsigned> used(layers.size(), 0u);
        for (int b = 1; b <= max_batch_forms; ++b) {
            std::vector<LayerDesc> t = layers;
            (void)layout_for(*this, b, nullptr, &t);
            for (size_t i = 0; i < layers.size(); ++i) used[i] |= forms_used(t[i], *this);
        }
        assign_offsets(&used);
    }
    planned_batch = 0;
    return "";
}

unsigned Plan::forms_used(const LayerDesc &l, const Plan &p)
{
    if (l.kind == kFirstConv) return kFormRows;
    if (l.kind == kLastConv) return kFormRows | (p.last_as_gemm(l) ? kFormGemmLast : 0u);      // rows: the direct kernels (fp32 plans; `lastconv_direct` of 16-bit ones)
    if (l.wino4) return kFormWino4;
    if (l.wino) return kFormWino;
    if (l.winoup) return kFormWinoUp;
    if (l.rowup) return kFormRowUp;
    if (l.bandconv) return kFormBand;
    if (l.rowconv) return kFormRow;
    if (l.fullk) return (p.dtype == 0 && (l.stride == 2 || (l.splits == 2 && !l.c1))) ? kFormFullK2 : kFormFullK;      // (16-bit plans: one tile-blocked form, fullk16.hip)
    return kFormRows;
}

void Plan::assign_offsets(const std::vector<unsigned> *used)
{
    size_t off = (size_t)blob_pad_kb * 1024;          // tools (tune key `blob_pad_kb`): where the weights sit relative to the workspace's channel interleave
    for (size_t li = 0; li < layers.size(); ++li) {
        LayerDesc &l = layers[li];
        const unsigned need = used ? (*used)[li] : ~0u;
        l.w_off = l.wgemm_off = l.wrl_off = l.wfk_off = l.wfk2_off = l.wwg_off = l.ww4_off = l.wwu_off = l.wru_off = l.wbc_off = l.wrc_off = -1;
        l.scale_off = l.shift_off = -1;
        if (need & kFormRows) {
            off = align_up(off, 256);
            l.w_off = (int64_t)off;
            off += (size_t)l.cout * l.cin * ((l.up4 || l.kind == kLastConv) ? 16 : 9) * (layer_weights_typed(l) ? elt() : sizeof(float));
        }
        if ((need & kFormGemmLast) && last_as_gemm(l)) {
            off = align_up(off, 256);
            l.wgemm_off = (int64_t)off;
            off += (size_t)4 * l.cout * 9 * l.cin * elt();
            if (l.c0 == 64 && l.c1 == 64 && l.cout == 3 && (l.hs % 64) == 0) {     // rowlast128 writes 12 columns per pixel (48-byte records): output_nc 3 only
                off = align_up(off, 256);
                l.wrl_off = (int64_t)off;
                off += (size_t)9 * 4 * 64 * 8 * elt();
            }
        }
        if (l.kind == kIgemm && fullk_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype)) {
            // the 16x16 / 8x8 layers in the tile-blocked layout of the full-K kernel (small-batch plans) ...
            if (need & kFormFullK) {
                off = align_up(off, 256);
                l.wfk_off = (int64_t)off;
                off += (size_t)l.cout * 9 * l.cin * sizeof(float);
            }
            if ((need & kFormFullK2) && l.c1 == 0 && l.c0 >= 256) {                  // ... and its K-split form, which reads a single source as two half-sources
                off = align_up(off, 256);
                l.wfk2_off = (int64_t)off;
                off += (size_t)l.cout * 9 * l.cin * sizeof(float);
            }
        }
        // 16-bit plans: the 8x8 / 4x4 / 2x2 layers in the tile-blocked order of the 16-bit full-K kernel (fullk16.hip)
        if (l.kind == kIgemm && dtype != 0 && (need & kFormFullK) && fullk16_levels && fullk16_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wfk_off = (int64_t)off;
            off += (size_t)l.cout * 9 * l.cin * elt();
        }
        // the stride-2 convs of the small levels on the K-split full-K kernel (tune key `fullk_s2=1`): only carried when that path is switched on
        if (l.kind == kIgemm && (need & kFormFullK2) && use_fullk_s2 && fullk_s2_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wfk2_off = (int64_t)off;                        // the only full-K copy of these layers: their single source as two half-sources
            off += (size_t)l.cout * 9 * l.cin * sizeof(float);
        }
        if (l.kind == kIgemm && (need & kFormWino) && use_wino && wino_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wwg_off = (int64_t)off;                         // 16/9 of the 9-tap bytes
            off += (size_t)16 * l.cout * l.cin * sizeof(float);
        }
        if (l.kind == kIgemm && (need & kFormWino4) && use_wino && use_wino4 && wino4_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.ww4_off = (int64_t)off;                         // 36/9 of the 9-tap bytes
            off += (size_t)36 * l.cout * l.cin * sizeof(float);
        }
        if (l.kind == kIgemm && (need & kFormWinoUp) && use_wino && use_winoup && winoup_layer(l.hs, l.c0, l.c1, l.cout, l.up4 || l.up, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wwu_off = (int64_t)off;
            off += (size_t)9 * l.cout * l.cin * sizeof(float);
        }
        if (l.kind == kIgemm && (need & kFormRowUp) && rowup_layer(l.hs, l.c0, l.c1, l.cout, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wru_off = (int64_t)off;
            off += (size_t)16 * l.cout * l.cin * elt();
        }
        if (l.kind == kIgemm && (need & kFormBand) && bandconv_layer(l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wbc_off = (int64_t)off;
            off += (size_t)l.cout * 9 * l.cin * elt();
        }
        if (l.kind == kIgemm && (need & kFormRow) && rowconv_layer(l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, dtype, l.inorm)) {
            off = align_up(off, 256);
            l.wrc_off = (int64_t)off;
            off += (size_t)l.cout * 9 * l.cin * elt();
        }
        if (!l.bnkey.empty() || !l.biaskey.empty()) {   // a conv bias travels as (scale 1, shift bias)
            off = align_up(off, 256);
            l.scale_off = (int64_t)off; off += (size_t)l.cout * sizeof(float);
            off = align_up(off, 256);
            l.shift_off = (int64_t)off; off += (size_t)l.cout * sizeof(float);
        }
    }
    blob_bytes = align_up(off, 256);
}



int64_t Plan::layer_flops(const LayerDesc &l) const { return 2ll * l.cout * l.cin * 9 * l.ho * l.ho; }

int64_t Plan::layer_act_bytes(const LayerDesc &l) const
{
    int64_t e = (int64_t)l.cin * l.hs * l.hs + (int64_t)l.cout * l.ho * l.ho;
    if (l.residual) e += (int64_t)l.cout * l.ho * l.ho;
    return e * (int64_t)elt();
}

namespace {
struct Arena {
    std::vector<std::pair<size_t, size_t>> free_;   // (offset, size), sorted by offset, coalesced
    size_t end = 0;
    size_t alloc(size_t n)
    {
        // best fit among free blocks
        int best = -1;
        for (int i = 0; i < (int)free_.size(); ++i)
            if (free_[i].second >= n && (best < 0 || free_[i].second < free_[best].second)) best = i;
        if (best >= 0) {
            const size_t off = free_[best].first;
            if (free_[best].second == n) free_.erase(free_.begin() + best);
            else { free_[best].first += n; free_[best].second -= n; }
            return off;
        }
        // grow; reuse a trailing free block if it touches the end
        if (!free_.empty() && free_.back().first + free_.back().second == end) {
            const size_t off = free_.back().first;
            free_.pop_back();
            end = off + n;
            return off;
        }
        const size_t off = end;
        end += n;
        return off;
    }
    void release(size_t off, size_t n)
    {
        auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(off, (size_t)0));
        it = free_.insert(it, {off, n});
        // coalesce with next / previous
        if (it + 1 != free_.end() && it->first + it->second == (it + 1)->first) {
            it->second += (it + 1)->second;
            free_.erase(it + 1);
        }
        if (it != free_.begin() && (it - 1)->first + (it - 1)->second == it->first) {
            (it - 1)->second += it->second;
            free_.erase(it);
        }
    }
};

BatchLayout layout_for(const Plan &p, int batch, std::vector<size_t> *offsets, std::vector<LayerDesc> *tiled)
{
    Arena a;
    std::vector<size_t> off(p.tensors.size(), 0);
    auto bytes_of = [&](const TensorDesc &t) {
        return align_up((size_t)t.c * t.h * t.h * (size_t)batch * p.elt(), 256);
    };
    size_t partial = 0;
    int groups_max = 0, cmax = 0;
    for (int li = 0; li < (int)p.layers.size(); ++li) {
        const LayerDesc &l = p.layers[li];
        if (l.out >= 0) off[l.out] = a.alloc(bytes_of(p.tensors[l.out]));
        if (!p.keep_intermediates)
            for (int t : {l.src0, l.src1, l.res})
                if (t >= 0 && p.tensors[t].last_use == li) {
                    // a tensor may appear twice among (src0, src1, res) only if the graph is
                    // malformed; the U-Net never does that
                    a.release(off[t], bytes_of(p.tensors[t]));
                }
        if (l.kind == kIgemm) {
            int bm, bn, splits, group;
            const int Mout = batch * l.ho * l.ho;
            const int M = l.up4 ? batch * l.hs * l.hs : Mout;
            choose_tiling(M, l.cout, (l.up4 ? 4 : 9) * l.cin / p.ktile_channels(), l.up4 ? 4 : 1, l.up, p.dtype, &bm, &bn, &splits, &group);
            const bool smallm = !l.up4 && smallm_eligible(M, l.cin, l.c1, l.cout, (size_t)batch * l.hs * l.hs * l.cin * 4, p.dtype == 0 ? p.smallm_kb : 64);
            if (smallm) { bm = bn = 1; splits = 1; group = 1; }
            // (every kernel choice below looks only at the offset of the weight form IT reads: the blob of a handle carries just the forms its
            // batch range uses, Plan::build)
            int fullk = smallm ? 0 : fullk_choice(batch, l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, p.dtype);
            bool fullk_k2 = false;
            if (fullk) {
                const bool want = p.use_fullk_split && fullk_split(batch, l.ho, l.c0, l.c1, l.cout, fullk, p.fullk_split_max_tiles);
                if (want && (l.c1 ? l.wfk_off >= 0 : l.wfk2_off >= 0)) fullk_k2 = true;
                else if (l.wfk_off < 0) fullk = 0;
            }
            const bool fullk_s2 = !fullk && !smallm && p.use_fullk_s2 && p.use_fullk_split && l.wfk2_off >= 0 &&
                                  fullk_s2_layer(l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, p.dtype, l.inorm) &&
                                  fullk_s2_choice(batch, l.hs, l.ho, l.c0, l.cout, p.use_fullk_s2) > 0;
            if (fullk_s2) { fullk = 1; bm = 16; bn = 16; splits = 2; group = 1; }
            else if (fullk) { bm = 16 * fullk; bn = 16; splits = fullk_k2 ? 2 : 1; group = 1; }
            // 16-bit plans: the same single-launch structure, whole K per workgroup, no split (fullk16.hip)
            const int fullk16 = (p.dtype != 0 && !smallm && l.wfk_off >= 0)
                                    ? fullk16_choice(batch, l.hs, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, p.dtype, p.fullk16_levels, p.fullk16_min_frames) : 0;
            if (fullk16) { fullk = fullk16; bm = 16 * fullk16; bn = 16; splits = 1; group = 1; }
            int wsplits = 1;
            int w4splits = 1;
            const int wino4 = (p.use_wino && p.use_wino4 && l.ww4_off >= 0 && !smallm) ? wino4_choice(batch, l.ho, l.cin, l.cout, &w4splits) : 0;
            const int wino = (!wino4 && p.use_wino && l.wwg_off >= 0 && !smallm) ? wino_choice(batch, l.ho, l.cin, l.cout, &wsplits) : 0;
            if (wino) { bm = 32; bn = 32 * wino; splits = wsplits; group = 1; }
            if (wino4) { bm = 32; bn = 32; splits = w4splits; group = 1; }
            int usplits = 1;
            const int winoup = (p.use_wino && p.use_winoup && l.wwu_off >= 0) ? winoup_choice(batch, l.hs, l.cin, l.cout, &usplits, p.winoup_nb, p.winoup_target, !l.up4) : 0;
            if (winoup) { bm = 32; bn = 32 * winoup; splits = usplits; group = 1; }
            const int rowconv = p.use_rowconv && l.wrc_off >= 0 ? rowconv_rows(batch, l.ho, l.ho, l.c0) : 0;
            if (rowconv) { bm = (l.c0 == 64 ? 64 : 32) * rowconv; bn = l.c0; splits = 1; group = 1; }
            int rowup = p.use_rowup && l.wru_off >= 0 ? rowup_rows(batch, l.hs, l.hs) : 0;
            if (rowup < 8) rowup = 0;     // short strips (1 frame: 4 rows + 2 halo steps) do not beat the implicit GEMM: 22.4 vs 22.9 us
            if (rowup) { bm = 32 * rowup; bn = 32; splits = 1; group = 1; }
            const bool bandconv = p.use_bandconv && l.wbc_off >= 0 && !smallm && !fullk16 &&
                                  (l.ho >= 8 ? (long)batch * (l.ho == 16 ? 4 : 2) * (l.cout / 32) >= p.bandconv_min_blocks
                                             : batch >= p.bandconv_min_frames_small);
            if (bandconv) { bm = l.ho == 16 ? 64 : 32; bn = 32; splits = 1; group = 1; }
            int pbn = 0;
            const int patch16 = (p.use_patch16 && !smallm && !fullk && !rowconv && !bandconv)
                                    ? patch16_choice(batch, l.ho, l.c0, l.c1, l.cout, l.stride, l.up, l.up4, p.dtype, l.inorm, p.patch16_min_blocks, &pbn) : 0;
            int patchup16 = 0;
            if (!patch16 && p.use_patch16 && p.use_patchup16 && !smallm && !fullk && !rowup && !winoup)
                patchup16 = patchup16_choice(batch, l.hs, l.c0, l.c1, l.cout, l.up4, p.dtype, l.inorm, p.patch16_min_blocks, &pbn);
            if (patch16 || patchup16) { bm = 256; bn = pbn; splits = 1; group = 1; }
            int route = kInNone;
            if (l.inorm) {
                // rows of one wave (32 per 32x32 tile row, bm / 2 waves... = bm / 2 for the 2x2-wave tiles) must stay inside one
                // frame for the epilogue sums; tiny levels do statistics + normalisation in one workgroup per channel slab
                const int hw = l.ho * l.ho, rhw = l.up4 ? l.hs * l.hs : hw;
                const int wave_rows = bm == 32 ? 32 : bm / 2;
                if (!smallm && !fullk && !wino && !wino4 && !winoup && splits == 1 && rhw >= 1024 && rhw % wave_rows == 0) {
                    route = kInFused;
                    groups_max = std::max(groups_max, (l.up4 ? 4 : 1) * rhw / wave_rows);
                } else if ((wino || winoup) && !wino4 && p.in_wino_stats) {
                    // one group per tile-block of 8 x 16 output pixels; also at 32 x 32 and below, where in_small would put a 512-channel frame on 16 workgroups
                    // (57.9 -> 33 us per layer at 32 x 32, batch 1)
                    route = kInWino;
                    groups_max = std::max(groups_max, hw / 128);
                } else if (hw <= p.in_small_max_hw) {
                    route = kInSmall;
                } else {
                    route = kInReduce;
                    groups_max = std::max(groups_max, (hw + 63) / 64);
                }
                cmax = std::max(cmax, l.cout);
            }
            if (tiled) {
                const long tiles = (long)(l.up4 ? 4 : 1) * ((M + bm - 1) / std::max(bm, 1)) * ((l.cout + bn - 1) / std::max(bn, 1));
                (*tiled)[li].wino = wino;
                (*tiled)[li].wino4 = wino4;
                (*tiled)[li].winoup = winoup;
                (*tiled)[li].fused_splitk = (wino || wino4 || winoup) ? splits > 1 : (p.dtype == 0 || p.fused_splitk16) && !rowconv && !bandconv && !rowup && !smallm && !fullk && !l.inorm && splits >= 2 && splits <= 8 && tiles <= (long)Plan::kTileCounters &&
                                            (size_t)splits * Mout * l.cout * sizeof(float) < (size_t)0x7fffffff;
                (*tiled)[li].bm = bm; (*tiled)[li].bn = bn; (*tiled)[li].splits = splits; (*tiled)[li].group = group;
                (*tiled)[li].smallm = smallm; (*tiled)[li].in_route = route; (*tiled)[li].fullk = fullk; (*tiled)[li].rowconv = rowconv; (*tiled)[li].bandconv = bandconv; (*tiled)[li].rowup = rowup; (*tiled)[li].patch16 = patch16 ? patch16 : patchup16;
            }
            if (splits > 1) partial = std::max(partial, (size_t)splits * Mout * l.cout * sizeof(float));
        }
        // the GEMM form of the last conv parks its [B][hs][hs][4*cout] fp32 result in the split-K scratch (idle by then)
        if (p.last_as_gemm(l)) partial = std::max(partial, (size_t)batch * l.hs * l.hs * 4 * l.cout * sizeof(float));
    }
    if (offsets) *offsets = off;
    // InstanceNorm statistics: per-group (sum d, sum d^2, shift) [3][batch][groups_max][cmax] and the finalised (mean, rstd) [2][batch][cmax]
    const size_t stats = cmax ? align_up(((size_t)3 * groups_max + 2) * batch * cmax * sizeof(float), 256) : 0;
    return {align_up(a.end, 256), align_up(partial, 256), stats, groups_max};
}
}  // namespace

void Plan::plan_batch(int batch)
{
    if (batch == planned_batch) return;
    std::vector<size_t> off;
    const BatchLayout bl = layout_for(*this, batch, &off, &layers);
    for (size_t i = 0; i < tensors.size(); ++i) tensors[i].offset = persistent_bytes() + off[i];
    act_bytes = persistent_bytes() + bl.act_bytes;
    partial_bytes = bl.partial_bytes;
    partial_offset = act_bytes;
    stats_bytes = bl.stats_bytes;
    stats_offset = partial_offset + partial_bytes;
    stats_groups_max = bl.groups_max;
    planned_batch = batch;
}

size_t Plan::workspace_bytes(int batch) const
{
    const BatchLayout bl = layout_for(*this, batch, nullptr, nullptr);
    return persistent_bytes() + bl.act_bytes + bl.partial_bytes + bl.stats_bytes;
}

std::string Plan::pack(void *blob, size_t bytes) const
{
    if (bytes < blob_bytes) return "packed blob buffer too small";
    for (const auto &pd : params)
        if (!pd.set) return "missing state-dict tensor: " + pd.key;
    std::memset(blob, 0, blob_bytes);
    char *base = static_cast<char *>(blob);
    auto get = [&](const std::string &k) -> const ParamDesc & { return params[param_index.at(k)]; };
    for (const auto &l : layers) {
        const float *W = get(l.wkey).data.data();           // OIHW
        // igemm-family weights are stored in the plan's dtype; staged in fp32 then narrowed (RNE) if bf16
        const bool narrow = dtype != 0 && layer_weights_typed(l);
        const size_t wcount = (size_t)l.cout * l.cin * ((l.up4 || l.kind == kLastConv) ? 16 : 9);
        // the row form ([co][tap][ci], sub-pixel rows, or the first conv's [ci][tap][co]) is always staged on the host: every other form is derived
        // from it (or from W), and it is copied into the blob only where a kernel reads it (w_off >= 0)
        std::vector<float> stage_buf(wcount);
        float *dst = stage_buf.data();
        std::vector<uint16_t> rows16;                        // the same values in the plan's 16-bit storage type (RNE), for the forms regrouped from them
        const int cin = l.cin, cout = l.cout;
        if ((l.kind == kIgemm && l.up4) || l.kind == kLastConv) {
            // sub-pixel form of Upsample(x2, nearest) + Conv3x3: output parity (py, px) only ever
            // sees 2x2 distinct source pixels, so the 3x3 taps that alias onto the same source
            // pixel are pre-summed (in double, rounded once):
            //   py = 0: source rows {y-1: ky 0} {y: ky 1,2}     py = 1: {y: ky 0,1} {y+1: ky 2}
            // layout [parity][co][a*2+b][ci]
            static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};   // [parity][a] -> k range
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    for (int co = 0; co < cout; ++co)
                        for (int ci = 0; ci < cin; ++ci) {
                            const float *w9 = W + ((size_t)co * cin + ci) * 9;
                            for (int a = 0; a < 2; ++a)
                                for (int b = 0; b < 2; ++b) {
                                    double acc = 0.0;
                                    for (int ky = lo[py][a]; ky <= hi[py][a]; ++ky)
                                        for (int kx = lo[px][b]; kx <= hi[px][b]; ++kx) acc += (double)w9[ky * 3 + kx];
                                    dst[(((size_t)(py * 2 + px) * cout + co) * 4 + a * 2 + b) * cin + ci] = (float)acc;
                                }
                        }
            if (l.wwu_off >= 0) pack_winoup_weights(W, cin, cout, reinterpret_cast<float *>(base + l.wwu_off));
            if (last_as_gemm(l) && l.wgemm_off >= 0) {
                // the same pre-summed taps as one 3x3 conv on the LOW-res source: output channel par*cout + co, tap
                // (a, b) of parity (py, px) sits at low-res offset (py - 1 + a, px - 1 + b); the other taps are zero
                uint16_t *g = reinterpret_cast<uint16_t *>(base + l.wgemm_off);
                for (int par = 0; par < 4; ++par)
                    for (int co = 0; co < cout; ++co)
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b) {
                                const int tap = ((par >> 1) + a) * 3 + (par & 1) + b;
                                for (int ci = 0; ci < cin; ++ci)
                                    g[(((size_t)par * cout + co) * 9 + tap) * cin + ci] = narrow16(dst[(((size_t)par * cout + co) * 4 + a * 2 + b) * cin + ci], dtype);
                            }
                if (l.wrl_off >= 0)     // (the blob is zero-filled, so the untouched taps of g are zeros)
                    pack_rowlast_weights(g, reinterpret_cast<uint16_t *>(base + l.wrl_off), 4 * cout);
            }
        } else if (l.kind == kIgemm) {
            // [co][tap][ci]  -- the implicit-GEMM B operand, K contiguous per output channel
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t)
                        dst[((size_t)co * 9 + t) * cin + ci] = W[((size_t)co * cin + ci) * 9 + t];
            if (l.wfk_off >= 0 && dtype == 0) pack_fullk_weights(dst, l.c0, l.c1 ? 2 : 1, cout, reinterpret_cast<float *>(base + l.wfk_off));
            if (l.wfk2_off >= 0) pack_fullk_weights(dst, l.c0 / 2, 2, cout, reinterpret_cast<float *>(base + l.wfk2_off));
            if (l.wwg_off >= 0) pack_wino_weights(W, cin, cout, reinterpret_cast<float *>(base + l.wwg_off));
            if (l.ww4_off >= 0) pack_wino4_weights(W, cin, cout, reinterpret_cast<float *>(base + l.ww4_off));
            if (l.wwu_off >= 0) pack_winoup_weights(W, cin, cout, reinterpret_cast<float *>(base + l.wwu_off));      // an up-conv below the sub-pixel extent that larger batches run on winoup3x3
        } else if (l.kind == kFirstConv) {
            // [ci][tap][co]  -- broadcast rows for the direct first-layer kernel
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t)
                        dst[((size_t)ci * 9 + t) * cout + co] = W[((size_t)co * cin + ci) * 9 + t];
        }
        if (narrow) {
            rows16.resize(wcount);
            for (size_t i = 0; i < wcount; ++i) rows16[i] = narrow16(stage_buf[i], dtype);    // round to nearest even
            if (l.w_off >= 0) std::memcpy(base + l.w_off, rows16.data(), wcount * sizeof(uint16_t));
        } else if (l.w_off >= 0) {
            std::memcpy(base + l.w_off, stage_buf.data(), wcount * sizeof(float));
        }
        if (l.wru_off >= 0)
            pack_rowup_weights(rows16.data(), reinterpret_cast<uint16_t *>(base + l.wru_off));
        if (l.wfk_off >= 0 && dtype != 0 && l.kind == kIgemm)     // 16-bit plans: the narrowed rows regrouped into the tile-blocked order of conv3x3_fullk16
            pack_fullk16_weights(rows16.data(), l.c0, l.c1 ? 2 : 1, cout, reinterpret_cast<uint16_t *>(base + l.wfk_off));
        if (l.wbc_off >= 0)
            pack_bandconv_weights(rows16.data(), reinterpret_cast<uint16_t *>(base + l.wbc_off), cout);
        if (l.wrc_off >= 0)     // the same bf16 values, regrouped into the MFMA A-fragments the row kernel keeps in registers
            pack_rowconv_weights(rows16.data(), reinterpret_cast<uint16_t *>(base + l.wrc_off), l.c0);
        if (!l.biaskey.empty()) {
            const float *bv = get(l.biaskey).data.data();
            float *sc = reinterpret_cast<float *>(base + l.scale_off);
            float *sh = reinterpret_cast<float *>(base + l.shift_off);
            for (int c = 0; c < cout; ++c) { sc[c] = 1.0f; sh[c] = bv[c]; }
        }
        if (!l.bnkey.empty()) {
            // eval-mode BatchNorm2d folded to y = x*scale + shift, applied AFTER accumulation
            // (same order as conv -> BN in the reference)
            const float *g = get(l.bnkey + ".weight").data.data();
            const float *b = get(l.bnkey + ".bias").data.data();
            const float *m = get(l.bnkey + ".running_mean").data.data();
            const float *v = get(l.bnkey + ".running_var").data.data();
            float *sc = reinterpret_cast<float *>(base + l.scale_off);
            float *sh = reinterpret_cast<float *>(base + l.shift_off);
            for (int c = 0; c < cout; ++c) {
                const double s = (double)g[c] / std::sqrt((double)v[c] + kBnEps);
                sc[c] = (float)s;
                sh[c] = (float)((double)b[c] - (double)m[c] * s);
            }
        }
    }
    return "";
}

}  // namespace lspf2f
