// gfx950: the landmark edge map of the render loop on the device (include/lspraster.h).
//
// One workgroup rasterises one 64-row band of one frame.  A WAVE takes one edge (thick line) of the frame's ~88-edge list at a time:
// the set-up (perpendicular offset, clipping, the per-edge divisions) is wave-uniform, and the per-pixel / per-row work is spread over
// the 64 lanes in closed form -- a DDA step k is x0 + k, (y0 + k * step) >> 16; a scanline y of the fill is xs + (y - y_event) * dx
// on each chain, exactly what the sequential `x += dx` reaches -- so an edge costs a few hundred instructions instead of a loop over
// its length.  The primitives only ever write one value, so the image is the union of their pixel sets and the order is irrelevant:
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

// sub-pixel DDA between two 16.16 points: one pixel per step along the major axis, plus the rounded far end point.  Arguments are
// wave-uniform; step k goes to lane k % 64.
__device__ void dda(const Band &b, P2 a, P2 e, int lane)
{
    if (!clip((long long)b.w << SHIFT, (long long)b.h << SHIFT, a, e)) return;
    long long dx = e.x - a.x, dy = e.y - a.y;
    const long long ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    const bool xmajor = ax > ay;
    if (xmajor ? dx < 0 : dy < 0) {        // walk in the direction of the increasing major coordinate
        const P2 t = a; a = e; e = t;
        dx = -dx; dy = -dy;
    }
    const long long step = xmajor ? (dy * ONE) / (ax | 1) : (dx * ONE) / (ay | 1);
    const int count = (int)((xmajor ? e.x - a.x : e.y - a.y) >> SHIFT);
    a.x += ONE >> 1;
    a.y += ONE >> 1;
    if (lane == 0) dot(b, (int)((e.x + (ONE >> 1)) >> SHIFT), (int)((e.y + (ONE >> 1)) >> SHIFT));
    if (xmajor) {
        const int x0 = (int)(a.x >> SHIFT);
        for (int k = lane; k <= count; k += 64) dot(b, x0 + k, (int)((a.y + k * step) >> SHIFT));
    } else {
        const int y0 = (int)(a.y >> SHIFT);
        for (int k = lane; k <= count; k += 64) dot(b, (int)((a.x + k * step) >> SHIFT), y0 + k);
    }
}

// convex quad in 16.16: outline through the DDA, interior by walking the left and right edge chains from the top vertex.  The chains
// change edge at a handful of event rows (wave-uniform, simulated in order); the rows between two events go to the lanes.
__device__ void fill_quad(const Band &b, const P2 (&v)[4], int lane)
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
        dda(b, prev, v[i], lane);
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
    while (y <= (int)ymax) {
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
        // rows [y, ynext) see no further change: row y + k has x[i] + k * dxr[i] on chain i, as k sequential `x += dx` would give.
        // A chain whose edge list is exhausted keeps ye[i] <= y forever; the next check then drives `edges` below zero and ends the
        // walk, exactly as the row-by-row loop does -- so only ONE row may be emitted in that state.
        int ynext = (int)ymax + 1;
        if (ye[0] > y && ye[0] < ynext) ynext = ye[0];
        if (ye[1] > y && ye[1] < ynext) ynext = ye[1];
        if (ye[0] <= y || ye[1] <= y) ynext = y + 1;
        for (int k = lane; k < ynext - y; k += 64) {
            const int yy = y + k;
            if (yy < 0) continue;
            const long long x0 = x[0] + k * dxr[0], x1 = x[1] + k * dxr[1];
            const long long lo = x0 > x1 ? x1 : x0, hi = x0 > x1 ? x0 : x1;
            const int xa = (int)((lo + HALF) >> SHIFT), xb = (int)((hi + HALF) >> SHIFT);
            if (xb >= 0 && xa < b.w) span(b, yy, xa < 0 ? 0 : xa, xb >= b.w ? b.w - 1 : xb);
        }
        x[0] += (long long)(ynext - y) * dxr[0];
        x[1] += (long long)(ynext - y) * dxr[1];
        y = ynext;
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

__device__ void thick_line(const Band &b, int x0, int y0, int x1, int y1, int thickness, int lane)
{
    const P2 p0 = {(long long)x0 * ONE, (long long)y0 * ONE}, p1 = {(long long)x1 * ONE, (long long)y1 * ONE};
    const double dx = (double)(p0.x - p1.x) * (1.0 / (double)ONE), dy = (double)(p1.y - p0.y) * (1.0 / (double)ONE);
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    const int half = thickness << (SHIFT - 1);               // half the width, 16.16
    if (fabs(r) > 2.2204460492503131e-16) {
        r = ((double)half + (double)odd * (double)ONE * 0.5) / __dsqrt_rn(r);
        const long long ox = __double2ll_rn(dy * r), oy = __double2ll_rn(dx * r);      // round half to even, like cvRound
        const P2 q[4] = {{p0.x + ox, p0.y + oy}, {p0.x - ox, p0.y - oy}, {p1.x - ox, p1.y - oy}, {p1.x + ox, p1.y + oy}};
        fill_quad(b, q, lane);
    }
    const int radius = (half + (int)(ONE >> 1)) >> SHIFT;
    disc(b, x0, y0, radius, lane);
    disc(b, x1, y1, radius, lane);
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

__global__ __launch_bounds__(256) void edge_map_band(const Params p)
{
    extern __shared__ unsigned bits[];
    const int tid = threadIdx.x, frame = blockIdx.y;
    Band b;
    b.bits = bits; b.w = p.w; b.h = p.h; b.words = p.w >> 5;
    b.y0 = blockIdx.x * BAND;
    b.rows = p.h - b.y0 < BAND ? p.h - b.y0 : BAND;
    for (int i = tid; i < BAND * b.words; i += 256) bits[i] = 0u;
    __syncthreads();
    const size_t base = (size_t)frame * p.npoints * 2;
    const int reach = (p.thickness >> 1) + 2;                  // a primitive never leaves its end points' box by more than this
    const int lane = tid & 63;
    for (int s = __builtin_amdgcn_readfirstlane(tid >> 6); s < p.nseg; s += 4) {       // one edge per wave at a time
        const int ia = p.segments[2 * s], ib = p.segments[2 * s + 1];
        if ((unsigned)ia >= (unsigned)p.npoints || (unsigned)ib >= (unsigned)p.npoints) continue;
        const int x0 = coord(p, base + 2 * ia), y0 = coord(p, base + 2 * ia + 1);
        const int x1 = coord(p, base + 2 * ib), y1 = coord(p, base + 2 * ib + 1);
        const int lo = (y0 < y1 ? y0 : y1) - reach, hi = (y0 < y1 ? y1 : y0) + reach;
        if (hi < b.y0 || lo >= b.y0 + b.rows) continue;        // this edge does not touch the band
        thick_line(b, x0, y0, x1, y1, p.thickness, lane);
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
    const size_t smem = (size_t)BAND * (width / 32) * sizeof(unsigned);
    hipLaunchKernelGGL(edge_map_band, dim3((height + BAND - 1) / BAND, batch), dim3(256), smem, static_cast<hipStream_t>(hip_stream), p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSPRASTER_ERR_HIP, std::string("edge_map_band launch: ") + hipGetErrorString(e));
    return LSPRASTER_OK;
}

}  // extern "C"
