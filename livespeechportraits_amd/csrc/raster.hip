// gfx950: the landmark edge map of the render loop on the device (include/lspraster.h).
//
// One workgroup rasterises one 64-row band of one frame, in two phases.  Phase 1, one THREAD per edge of the frame's ~88-edge list:
// the sequential, per-edge part of the scan conversion -- the perpendicular offset (double sqrt / divide), clipping, the emulated 64-bit
// divisions behind the DDA steps and scanline slopes, the order in which the fill's two edge chains advance -- is evaluated once into a
// small plan in LDS (measured: done per wave instead, these ~2500 instructions per edge made the kernel 120 us whatever the edge length).
// Phase 2, one WAVE per edge: pixels and scanlines are closed forms of the plan -- DDA step k is (base + k, (minor + k * step) >> 16), row
// y0 + k of a fill piece has x_i + k * d_i on chain i, exactly what the sequential `x += dx` reaches -- and go to the 64 lanes.  The primitives only ever write one value, so the image is the union of their pixel sets and the order is irrelevant:
// lanes set bits of the band's bitmask in LDS (64 rows x W bits = 4 KB at W = 512) with ds_or.  The band is then expanded to the
// output tensor with coalesced 16-byte stores: the kernel's HBM traffic is the output itself (1 MiB fp32 per frame) plus ~1 KB of points.
//
// The scan conversion follows OpenCV 4.4.0's cv::line for thickness > 1 (modules/imgproc/src/drawing.cpp: ThickLine ->
// FillConvexPoly with a Line2 outline, Circle end caps) in 16.16 fixed point with 64-bit intermediates; doubles are used
// exactly where OpenCV uses them (the perpendicular offset and the clip intersections), with IEEE division / square root.
#include "../../include/lspraster.h"

#include <hip/hip_runtime.h>

#include <string>

namespace lspraster {

constexpr int SHIFT = 16;
constexpr long long ONE = 1ll << SHIFT;
constexpr int BAND = 64;                    // rows per workgroup

struct Band {
    unsigned *bits;                         // LDS, [BAND][words]
    int y0, rows, w, h, words;              // image rows [y0, y0 + rows), image size
};

__device__ __forceinline__ void span(const Band &b, int y, int xl, int xr)      // inclusive, already clipped to [0, w)
{
    const int r = y - b.y0;
    if ((unsigned)r >= (unsigned)b.rows || xl > xr) return;
    unsigned *row = b.bits + r * b.words;
    const int wl = xl >> 5, wr = xr >> 5;
    for (int wi = wl; wi <= wr; ++wi) {
        unsigned m = 0xffffffffu;
        if (wi == wl) m &= 0xffffffffu << (xl & 31);
        if (wi == wr) m &= 0xffffffffu >> (31 - (xr & 31));
        atomicOr(row + wi, m);
    }
}

__device__ __forceinline__ void dot(const Band &b, int x, int y)
{
    if ((unsigned)x < (unsigned)b.w && (unsigned)y < (unsigned)b.h) span(b, y, x, x);
}

struct P2 { long long x, y; };

// clip the segment to [0, width) x [0, height) (fixed-point extents); false = nothing left
__device__ bool clip(long long width, long long height, P2 &p1, P2 &p2)
{
    const long long right = width - 1, bottom = height - 1;
    auto code_x = [&](long long x) { return (int)(x < 0) + (int)(x > right) * 2; };
    int c1 = code_x(p1.x) + (int)(p1.y < 0) * 4 + (int)(p1.y > bottom) * 8;
    int c2 = code_x(p2.x) + (int)(p2.y < 0) * 4 + (int)(p2.y > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        if (c1 & 12) {
            const long long a = c1 < 8 ? 0 : bottom;
            p1.x += (long long)((double)(a - p1.y) * (double)(p2.x - p1.x) / (double)(p2.y - p1.y));
            p1.y = a;
            c1 = code_x(p1.x);
        }
        if (c2 & 12) {
            const long long a = c2 < 8 ? 0 : bottom;
            p2.x += (long long)((double)(a - p2.y) * (double)(p2.x - p1.x) / (double)(p2.y - p1.y));
            p2.y = a;
            c2 = code_x(p2.x);
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                const long long a = c1 == 1 ? 0 : right;
                p1.y += (long long)((double)(a - p1.x) * (double)(p2.y - p1.y) / (double)(p2.x - p1.x));
                p1.x = a;
                c1 = 0;
            }
            if (c2) {
                const long long a = c2 == 1 ? 0 : right;
                p2.y += (long long)((double)(a - p2.x) * (double)(p2.y - p1.y) / (double)(p2.x - p1.x));
                p2.x = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// ---- phase 1, one THREAD per edge: everything that is per-edge and sequential (the perpendicular offset, clipping, the 64-bit
// divisions behind the DDA steps and the scanline slopes, the order in which the fill's two chains change edge) becomes a plan ----
struct DdaRec {            // one outline piece: pixel k is (base + k, (minor + k * step) >> 16) (x-major) or transposed
    int valid, xmajor, count, base;
    long long minor, step;
    int ex, ey;            // the rounded far end point, drawn once
};
struct Piece {             // scanlines y0 .. y0 + n - 1 of the fill: chain i sits at x_i + k * d_i on row y0 + k
    int y0, n;
    long long x0, d0, x1, d1;
};
constexpr int MAX_PIECES = 6;      // every piece but the last ends at an edge change, and a quad has 4 edges to spend
struct EdgePlan {
    int skip, npieces, radius, hasquad;
    int cx0, cy0, cx1, cy1;
    DdaRec dda[4];
    Piece pc[MAX_PIECES];
};

__device__ void plan_dda(const Band &b, P2 a, P2 e, DdaRec &r)
{
    r.valid = 0;
    if (!clip((long long)b.w << SHIFT, (long long)b.h << SHIFT, a, e)) return;
    long long dx = e.x - a.x, dy = e.y - a.y;
    const long long ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    const bool xmajor = ax > ay;
    if (xmajor ? dx < 0 : dy < 0) {        // walk in the direction of the increasing major coordinate
        const P2 t = a; a = e; e = t;
        dx = -dx; dy = -dy;
    }
    r.step = xmajor ? (dy * ONE) / (ax | 1) : (dx * ONE) / (ay | 1);
    r.count = (int)((xmajor ? e.x - a.x : e.y - a.y) >> SHIFT);
    a.x += ONE >> 1;
    a.y += ONE >> 1;
    r.ex = (int)((e.x + (ONE >> 1)) >> SHIFT);
    r.ey = (int)((e.y + (ONE >> 1)) >> SHIFT);
    r.xmajor = xmajor ? 1 : 0;
    r.base = (int)((xmajor ? a.x : a.y) >> SHIFT);
    r.minor = xmajor ? a.y : a.x;
    r.valid = 1;
}

// convex quad in 16.16: outline through the DDA, interior by walking the left and right edge chains from the top vertex.  The chains
// change edge at a handful of event rows; between two events row y + k has x_i + k * d_i on chain i, exactly what k sequential
// `x += dx` give -- one Piece per stretch.
__device__ void plan_quad(const Band &b, const P2 (&v)[4], EdgePlan &pl)
{
    constexpr int N = 4;
    constexpr long long HALF = ONE >> 1;
    long long xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    int imin = 0;
    P2 prev = v[N - 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (v[i].y < ymin) { ymin = v[i].y; imin = i; }
        ymax = v[i].y > ymax ? v[i].y : ymax;
        xmax = v[i].x > xmax ? v[i].x : xmax;
        xmin = v[i].x < xmin ? v[i].x : xmin;
        plan_dda(b, prev, v[i], pl.dda[i]);
        prev = v[i];
    }
    xmin = (xmin + HALF) >> SHIFT; xmax = (xmax + HALF) >> SHIFT;
    ymin = (ymin + HALF) >> SHIFT; ymax = (ymax + HALF) >> SHIFT;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= b.w || (int)ymin >= b.h) return;
    if (ymax > b.h - 1) ymax = b.h - 1;
    int idx[2] = {imin, imin}, ye[2], y = (int)ymin, edges = N;
    const int di[2] = {1, N - 1};
    long long x[2] = {-ONE, -ONE}, dxr[2] = {0, 0};
    ye[0] = ye[1] = y;
    // vertex access with a runtime index: 4 entries, resolved with selects (no scratch)
    auto vx = [&](int i) { return i == 0 ? v[0].x : i == 1 ? v[1].x : i == 2 ? v[2].x : v[3].x; };
    auto vy = [&](int i) { return i == 0 ? v[0].y : i == 1 ? v[1].y : i == 2 ? v[2].y : v[3].y; };
    while (y <= (int)ymax && pl.npieces < MAX_PIECES) {
        // edge changes due at row y (the sequential algorithm checks them at every row; they can only fire at y == ye[i])
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (y < ye[i]) continue;
            int i0 = idx[i], i1 = i0 + di[i];
            if (i1 >= N) i1 -= N;
            while (edges-- > 0) {
                const int ty = (int)((vy(i1) + HALF) >> SHIFT);
                if (ty > y) {
                    const long long xs = vx(i0), xe = vx(i1);
                    ye[i] = ty;
                    dxr[i] = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                    x[i] = xs;
                    idx[i] = i1;
                    break;
                }
                i0 = i1;
                i1 += di[i];
                if (i1 >= N) i1 -= N;
            }
        }
        if (edges < 0) break;
        int ynext = (int)ymax + 1;
        if (ye[0] > y && ye[0] < ynext) ynext = ye[0];
        if (ye[1] > y && ye[1] < ynext) ynext = ye[1];
        if (ye[0] <= y || ye[1] <= y) ynext = y + 1;         // cannot happen (an exhausted chain ends the walk above); one row if it did
        Piece &q = pl.pc[pl.npieces++];
        q.y0 = y; q.n = ynext - y; q.x0 = x[0]; q.d0 = dxr[0]; q.x1 = x[1]; q.d1 = dxr[1];
        x[0] += (long long)(ynext - y) * dxr[0];
        x[1] += (long long)(ynext - y) * dxr[1];
        y = ynext;
    }
}

__device__ void plan_edge(const Band &b, int x0, int y0, int x1, int y1, int thickness, EdgePlan &pl)
{
    pl.npieces = 0; pl.hasquad = 0;
    pl.cx0 = x0; pl.cy0 = y0; pl.cx1 = x1; pl.cy1 = y1;
#pragma unroll
    for (int i = 0; i < 4; ++i) pl.dda[i].valid = 0;
    const P2 p0 = {(long long)x0 * ONE, (long long)y0 * ONE}, p1 = {(long long)x1 * ONE, (long long)y1 * ONE};
    const double dx = (double)(p0.x - p1.x) * (1.0 / (double)ONE), dy = (double)(p1.y - p0.y) * (1.0 / (double)ONE);
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    const int half = thickness << (SHIFT - 1);               // half the width, 16.16
    if (fabs(r) > 2.2204460492503131e-16) {
        r = ((double)half + (double)odd * (double)ONE * 0.5) / __dsqrt_rn(r);
        const long long ox = __double2ll_rn(dy * r), oy = __double2ll_rn(dx * r);      // round half to even, like cvRound
        const P2 q[4] = {{p0.x + ox, p0.y + oy}, {p0.x - ox, p0.y - oy}, {p1.x - ox, p1.y - oy}, {p1.x + ox, p1.y + oy}};
        pl.hasquad = 1;
        plan_quad(b, q, pl);
    }
    pl.radius = (half + (int)(ONE >> 1)) >> SHIFT;
}

// ---- phase 2, one WAVE per edge: the plan is wave-uniform, pixels and scanlines go to the lanes ----
__device__ void draw_dda(const Band &b, const DdaRec &r, int lane)
{
    if (!r.valid) return;
    if (lane == 0) dot(b, r.ex, r.ey);
    if (r.xmajor)
        for (int k = lane; k <= r.count; k += 64) dot(b, r.base + k, (int)((r.minor + k * r.step) >> SHIFT));
    else
        for (int k = lane; k <= r.count; k += 64) dot(b, (int)((r.minor + k * r.step) >> SHIFT), r.base + k);
}

__device__ void draw_piece(const Band &b, const Piece &q, int lane)
{
    constexpr long long HALF = ONE >> 1;
    // only the rows inside this band matter: start the lanes at the band's first row of the piece
    int k0 = b.y0 - q.y0;
    if (k0 < 0) k0 = 0;
    int k1 = b.y0 + b.rows - q.y0;
    if (k1 > q.n) k1 = q.n;
    for (int k = k0 + lane; k < k1; k += 64) {
        const int yy = q.y0 + k;
        if (yy < 0) continue;
        const long long xa = q.x0 + k * q.d0, xb = q.x1 + k * q.d1;
        const long long lo = xa > xb ? xb : xa, hi = xa > xb ? xa : xb;
        const int xl = (int)((lo + HALF) >> SHIFT), xr = (int)((hi + HALF) >> SHIFT);
        if (xr >= 0 && xl < b.w) span(b, yy, xl < 0 ? 0 : xl, xr >= b.w ? b.w - 1 : xr);
    }
}

// filled midpoint circle: horizontal spans; the walk is short (radius <= 16) and wave-uniform, lanes 0..3 write its four spans
__device__ void disc(const Band &b, int cx, int cy, int radius, int lane)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    auto row = [&](int y, int xl, int xr) {
        if ((unsigned)y >= (unsigned)b.h || xl >= b.w || xr < 0) return;
        span(b, y, xl < 0 ? 0 : xl, xr > b.w - 1 ? b.w - 1 : xr);
    };
    if (!(cx - radius < b.w && cx + radius >= 0 && cy - radius < b.h && cy + radius >= 0)) return;
    while (dx >= dy) {
        if (lane < 4) {
            const int ry = (lane & 2) ? dx : dy, rx = (lane & 2) ? dy : dx;      // lanes 0,1: rows cy -/+ dy, half-width dx; 2,3: rows cy -/+ dx, half-width dy
            row((lane & 1) ? cy + ry : cy - ry, cx - rx, cx + rx);
        }
        ++dy;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

struct Params {
    const void *points;
    const int *segments;
    float *out_f32;
    unsigned char *out_u8;
    int dtype, npoints, nseg, thickness, h, w;
};

__device__ __forceinline__ int coord(const Params &p, size_t i)
{
    if (p.dtype == LSPRASTER_POINTS_I32) return static_cast<const int *>(p.points)[i];
    if (p.dtype == LSPRASTER_POINTS_F32) return (int)static_cast<const float *>(p.points)[i];     // truncation toward zero
    return (int)static_cast<const double *>(p.points)[i];
}

constexpr int EDGE_CHUNK = 96;             // edges planned per round (41 KB of plans in LDS)

__global__ __launch_bounds__(256) void edge_map_band(const Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    EdgePlan *plans = reinterpret_cast<EdgePlan *>(lds);                       // [EDGE_CHUNK]
    unsigned *bits = reinterpret_cast<unsigned *>(lds + sizeof(EdgePlan) * EDGE_CHUNK);
    const int tid = threadIdx.x, lane = tid & 63, frame = blockIdx.y;
    Band b;
    b.bits = bits; b.w = p.w; b.h = p.h; b.words = p.w >> 5;
    b.y0 = blockIdx.x * BAND;
    b.rows = p.h - b.y0 < BAND ? p.h - b.y0 : BAND;
    for (int i = tid; i < BAND * b.words; i += 256) bits[i] = 0u;
    const size_t base = (size_t)frame * p.npoints * 2;
    const int reach = (p.thickness >> 1) + 2;                  // a primitive never leaves its end points' box by more than this
    for (int c0 = 0; c0 < p.nseg; c0 += EDGE_CHUNK) {
        const int nc = p.nseg - c0 < EDGE_CHUNK ? p.nseg - c0 : EDGE_CHUNK;
        __syncthreads();                                       // bits zeroed / the previous chunk's plans are no longer read
        if (tid < nc) {
            EdgePlan &pl = plans[tid];
            const int s = c0 + tid;
            const int ia = p.segments[2 * s], ib = p.segments[2 * s + 1];
            pl.skip = 1;
            if ((unsigned)ia < (unsigned)p.npoints && (unsigned)ib < (unsigned)p.npoints) {
                const int x0 = coord(p, base + 2 * ia), y0 = coord(p, base + 2 * ia + 1);
                const int x1 = coord(p, base + 2 * ib), y1 = coord(p, base + 2 * ib + 1);
                const int lo = (y0 < y1 ? y0 : y1) - reach, hi = (y0 < y1 ? y1 : y0) + reach;
                if (!(hi < b.y0 || lo >= b.y0 + b.rows)) {     // the edge touches this band
                    pl.skip = 0;
                    plan_edge(b, x0, y0, x1, y1, p.thickness, pl);
                }
            }
        }
        __syncthreads();
        for (int e = __builtin_amdgcn_readfirstlane(tid >> 6); e < nc; e += 4) {
            const EdgePlan &pl = plans[e];
            if (pl.skip) continue;
            if (pl.hasquad) {
#pragma unroll
                for (int i = 0; i < 4; ++i) draw_dda(b, pl.dda[i], lane);
                for (int i = 0; i < pl.npieces; ++i) draw_piece(b, pl.pc[i], lane);
            }
            disc(b, pl.cx0, pl.cy0, pl.radius, lane);
            disc(b, pl.cx1, pl.cy1, pl.radius, lane);
        }
    }
    __syncthreads();
    // expand the band: 4 pixels per thread and step
    const int quads = b.rows * (p.w >> 2);
    for (int q = tid; q < quads; q += 256) {
        const int r = q / (p.w >> 2), x = (q - r * (p.w >> 2)) * 4;
        const unsigned nib = (bits[r * b.words + (x >> 5)] >> (x & 31)) & 15u;
        const size_t o = ((size_t)frame * p.h + b.y0 + r) * p.w + x;
        if (p.out_f32)
            *reinterpret_cast<float4 *>(p.out_f32 + o) = make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)(nib >> 3));
        if (p.out_u8)
            *reinterpret_cast<unsigned *>(p.out_u8 + o) = (nib & 1u ? 0xffu : 0u) | (nib & 2u ? 0xff00u : 0u) | (nib & 4u ? 0xff0000u : 0u) | (nib & 8u ? 0xff000000u : 0u);
    }
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

}  // namespace lspraster

using namespace lspraster;

extern "C" {

const char *lspraster_last_error(void) { return g_err.c_str(); }

int lspraster_edge_maps(const void *points_dev, int point_dtype, int batch, int npoints, const int32_t *segments_dev,
                        int nsegments, int thickness, int height, int width, float *out_f32_dev, unsigned char *out_u8_dev,
                        void *hip_stream)
{
    if (!points_dev || !segments_dev) return fail(LSPRASTER_ERR_INVALID_ARGUMENT, "null argument");
    if (!out_f32_dev && !out_u8_dev) return fail(LSPRASTER_ERR_INVALID_ARGUMENT, "at least one of out_f32_dev / out_u8_dev is required");
    if (point_dtype < LSPRASTER_POINTS_I32 || point_dtype > LSPRASTER_POINTS_F64) return fail(LSPRASTER_ERR_INVALID_ARGUMENT, "unknown point_dtype");
    if (batch < 1 || batch > 65535 || npoints < 1 || nsegments < 0) return fail(LSPRASTER_ERR_INVALID_ARGUMENT, "bad batch / npoints / nsegments");
    if (thickness < 2 || thickness > LSPRASTER_MAX_THICKNESS)
        return fail(LSPRASTER_ERR_UNSUPPORTED, "thickness must be in 2..32 (thickness 1 is a different OpenCV routine; the reference uses 2)");
    if (height < 1 || width < 32 || width % 32 || width > LSPRASTER_MAX_WIDTH) return fail(LSPRASTER_ERR_UNSUPPORTED, "width must be a multiple of 32, <= 4096");
    Params p{points_dev, segments_dev, out_f32_dev, out_u8_dev, point_dtype, npoints, nsegments, thickness, height, width};
    const size_t smem = sizeof(EdgePlan) * EDGE_CHUNK + (size_t)BAND * (width / 32) * sizeof(unsigned);      // <= 41 KB + 32 KB
    hipLaunchKernelGGL(edge_map_band, dim3((height + BAND - 1) / BAND, batch), dim3(256), smem, static_cast<hipStream_t>(hip_stream), p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSPRASTER_ERR_HIP, std::string("edge_map_band launch: ") + hipGetErrorString(e));
    return LSPRASTER_OK;
}

}  // extern "C"
