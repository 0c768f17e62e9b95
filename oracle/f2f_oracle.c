/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the arithmetic of
 * LiveSpeechPortraits' feature2face generator forward.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the library built from this file
 * (oracle/_build/libf2f_oracle.so); nothing under livespeechportraits_amd/ does.
 *
 * It is independent of PyTorch: the convolution, eval-mode BatchNorm, nearest upsample,
 * concatenation, residual add, ReLU and tanh are spelled out here with double-precision
 * accumulation and float storage between layers, so that a disagreement between the HIP path
 * and oracle/torch_oracle.py can be arbitrated without trusting ATen/oneDNN.
 *
 * Reference semantics restated (file:line under /root/reference; ops are torch.nn, the
 * reference's pinned dependency torch==1.7.1 -- cog.yaml:9):
 *   nn.Conv2d(k=3, p=1, s in {1,2}, bias=False)     models/networks.py:594-595, 611, 618, 627, 663, 666
 *       out[co][oy][ox] = sum_{ci,ky,kx} w[co][ci][ky][kx] * in[ci][oy*s+ky-1][ox*s+kx-1]  (zero pad)
 *   nn.BatchNorm2d eval, eps 1e-5                   models/networks.py:606-607, 664, 667
 *       y = (x - running_mean) / sqrt(running_var + eps) * weight + bias
 *   nn.Upsample(scale_factor=2, mode='nearest')     models/networks.py:610, 617, 626   out[y][x] = in[y>>1][x>>1]
 *   ResidualBlock: conv-BN-ReLU-conv-BN, += x, ReLU models/networks.py:670-675
 *   skip block: cat([x, model(x)], 1) / outermost   models/networks.py:642-646
 *   generator: tanh(model(x))                       models/networks.py:575-579
 *   level order / channel schedule                  models/networks.py:557-572, 592-640
 *
 * Parameter blob layout ("walk order", floats, no num_batches_tracked): for each level,
 * depth-first exactly as the state dict lists them:
 *   down conv OIHW, [BN: weight,bias,running_mean,running_var], res blocks
 *   (conv OIHW, BN x4, conv OIHW, BN x4), <submodule>, up conv OIHW, [BN x4], res blocks.
 *
 * Parity pin: checked against outputs of the reference itself frozen in tests/golden/
 * (tests/test_oracle.py); the reference has no tests of its own for this path.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F2F_BN_EPS 1e-5

typedef struct {
    const float *p;   /* cursor into the parameter blob */
    int nres, ngf, num_downs, input_nc, output_nc;
} f2f_ctx;

/* 3x3 conv, pad 1, NCHW, one frame. in: [cin][h][w], out: [cout][ho][wo] */
void f2f_conv3x3(const float *in, int cin, int h, int w, const float *wt, int cout, int stride,
                 float *out)
{
    const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)ho * wo);
#pragma omp for schedule(dynamic, 1)
        for (int co = 0; co < cout; ++co) {
            memset(acc, 0, sizeof(double) * (size_t)ho * wo);
            for (int ci = 0; ci < cin; ++ci) {
                const float *ip = in + (size_t)ci * h * w;
                const float *wp = wt + ((size_t)co * cin + ci) * 9;
                for (int ky = 0; ky < 3; ++ky) {
                    for (int kx = 0; kx < 3; ++kx) {
                        const double wv = wp[ky * 3 + kx];
                        for (int oy = 0; oy < ho; ++oy) {
                            const int iy = oy * stride + ky - 1;
                            if (iy < 0 || iy >= h) continue;
                            const float *irow = ip + (size_t)iy * w;
                            double *arow = acc + (size_t)oy * wo;
                            /* valid ox: 0 <= ox*stride+kx-1 < w */
                            int ox0 = 0;
                            while (ox0 < wo && ox0 * stride + kx - 1 < 0) ++ox0;
                            int ox1 = wo;
                            while (ox1 > ox0 && (ox1 - 1) * stride + kx - 1 >= w) --ox1;
                            if (stride == 1) {
                                const float *ir = irow + kx - 1;
                                for (int ox = ox0; ox < ox1; ++ox) arow[ox] += wv * (double)ir[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox)
                                    arow[ox] += wv * (double)irow[ox * stride + kx - 1];
                            }
                        }
                    }
                }
            }
            float *op = out + (size_t)co * ho * wo;
            for (int i = 0; i < ho * wo; ++i) op[i] = (float)acc[i];
        }
        free(acc);
    }
}

/* eval-mode BatchNorm2d in place; bn = [weight | bias | running_mean | running_var], each [c] */
void f2f_bn_eval(float *x, int c, int hw, const float *bn)
{
    const float *g = bn, *b = bn + c, *m = bn + 2 * c, *v = bn + 3 * c;
    for (int ch = 0; ch < c; ++ch) {
        const double inv = 1.0 / sqrt((double)v[ch] + F2F_BN_EPS);
        float *p = x + (size_t)ch * hw;
        for (int i = 0; i < hw; ++i)
            p[i] = (float)(((double)p[i] - (double)m[ch]) * inv * (double)g[ch] + (double)b[ch]);
    }
}

void f2f_relu(float *x, size_t n)
{
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* nearest x2: in [c][h][w] -> out [c][2h][2w] */
void f2f_upsample2(const float *in, int c, int h, int w, float *out)
{
    for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < 2 * h; ++y)
            for (int x = 0; x < 2 * w; ++x)
                out[((size_t)ch * 2 * h + y) * 2 * w + x] = in[((size_t)ch * h + (y >> 1)) * w + (x >> 1)];
}

static float *f2f_res(f2f_ctx *c, float *x, int ch, int h)
{
    const size_t n = (size_t)ch * h * h;
    float *t = (float *)malloc(n * sizeof(float));
    float *u = (float *)malloc(n * sizeof(float));
    f2f_conv3x3(x, ch, h, h, c->p, ch, 1, t);  c->p += (size_t)ch * ch * 9;
    f2f_bn_eval(t, ch, h * h, c->p);           c->p += 4 * (size_t)ch;
    f2f_relu(t, n);
    f2f_conv3x3(t, ch, h, h, c->p, ch, 1, u);  c->p += (size_t)ch * ch * 9;
    f2f_bn_eval(u, ch, h * h, c->p);           c->p += 4 * (size_t)ch;
    for (size_t i = 0; i < n; ++i) u[i] = u[i] + x[i];
    f2f_relu(u, n);
    free(t);
    free(x);
    return u;
}

static int imin(int a, int b) { return a < b ? a : b; }

/* one skip block; x: [cin][h][h] (not freed); returns a malloc'ed tensor:
 * outermost -> [output_nc][h][h]; otherwise cat([x, model(x)]) = [2*cin][h][h] */
static float *f2f_level(f2f_ctx *c, const float *x, int depth, int h)
{
    const int outer = depth == 0, inner_most = depth == c->num_downs - 1;
    const int mo = imin(1 << (depth > 0 ? depth - 1 : 0), 8), mi = imin(1 << depth, 8);
    const int cin = outer ? c->input_nc : c->ngf * mo;
    const int cmid = outer ? c->ngf : c->ngf * mi;
    const int cout = outer ? c->output_nc : c->ngf * mo;
    const int hd = h / 2;

    float *d = (float *)malloc((size_t)cmid * hd * hd * sizeof(float));
    f2f_conv3x3(x, cin, h, h, c->p, cmid, 2, d);  c->p += (size_t)cmid * cin * 9;
    if (!outer && !inner_most) { f2f_bn_eval(d, cmid, hd * hd, c->p); c->p += 4 * (size_t)cmid; }
    f2f_relu(d, (size_t)cmid * hd * hd);
    for (int r = 0; r < c->nres; ++r) d = f2f_res(c, d, cmid, hd);

    int cup = cmid;
    float *below = d;
    if (!inner_most) {
        below = f2f_level(c, d, depth + 1, hd);
        free(d);
        cup = 2 * cmid;
    }
    float *up = (float *)malloc((size_t)cup * h * h * sizeof(float));
    f2f_upsample2(below, cup, hd, hd, up);
    free(below);
    float *o = (float *)malloc((size_t)cout * h * h * sizeof(float));
    f2f_conv3x3(up, cup, h, h, c->p, cout, 1, o);  c->p += (size_t)cout * cup * 9;
    free(up);
    if (outer) return o;
    f2f_bn_eval(o, cout, h * h, c->p);  c->p += 4 * (size_t)cout;
    f2f_relu(o, (size_t)cout * h * h);
    for (int r = 0; r < c->nres; ++r) o = f2f_res(c, o, cout, h);

    float *cat = (float *)malloc((size_t)2 * cin * h * h * sizeof(float));
    memcpy(cat, x, (size_t)cin * h * h * sizeof(float));
    memcpy(cat + (size_t)cin * h * h, o, (size_t)cout * h * h * sizeof(float));
    free(o);
    return cat;
}

/* Whole generator. x: [batch][input_nc][size][size]; out: [batch][output_nc][size][size].
 * Returns the number of parameter floats consumed (caller checks it equals the blob length),
 * or -1 on bad arguments. apply_tanh = 0 returns the pre-tanh frame. */
long f2f_generator_forward(const float *params, int nres, int input_nc, int output_nc, int ngf,
                           int num_downs, int size, int batch, const float *x, float *out,
                           int apply_tanh)
{
    if (!params || !x || !out || nres < 1 || num_downs < 5 || size % (1 << num_downs)) return -1;
    long used = 0;
    for (int b = 0; b < batch; ++b) {
        f2f_ctx c = {params, nres, ngf, num_downs, input_nc, output_nc};
        float *o = f2f_level(&c, x + (size_t)b * input_nc * size * size, 0, size);
        const size_t n = (size_t)output_nc * size * size;
        for (size_t i = 0; i < n; ++i)
            out[(size_t)b * n + i] = apply_tanh ? (float)tanh((double)o[i]) : o[i];
        free(o);
        used = (long)(c.p - params);
    }
    return used;
}
