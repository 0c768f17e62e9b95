// gfx950: the landmark edge map of the render loop on the device (include/lspraster.h).
//
// One workgroup of 1024 threads rasterises one 64-row band of one frame.  Everything is one THREAD per work item, in three steps separated by
// barriers (the wave-per-edge form this replaces spent its 70 us walking 22 edges per wave one after the other, every field of an edge's plan a
// separate LDS round trip; phase stamps of the 256-thread form of this one -- profiles/r03_raster_stamps_thread_per_item.txt -- showed 60 of
// 76 us in the drawing loops at ~8 cycles per dependent instruction, one wave per SIMD: hence 16 waves per workgroup):
//   1. thread per edge: end points, band test, the perpendicular offset (double sqrt / divide, as OpenCV) -> the quad's four vertices;
//   2. thread per (edge, side) and per (edge, fill): the sequential part of the scan conversion -- clipping, the DDA step, the order in
//      which the fill's two edge chains advance and their slopes -- becomes a small plan in LDS.  The 64-bit divisions behind the steps
//      and slopes run as one IEEE double division plus an exact integer fix-up (div_exact below) instead of the ~100-instruction emulated
//      64-bit divide: |numerator| < 2^52, where the truncated double quotient is the integer quotient;
//   3. 12 primitives per edge (4 outline DDAs, up to 6 fill pieces, 2 end caps); their steps inside the band -- one outline pixel, one fill
//      scanline, one end cap -- are counted, prefix-summed and dealt to the 1024 threads in equal shares (thread per primitive left the launch
//      waiting for the one thread with the longest edge: 67k of 93k cycles, profiles/r03_raster_stamps_1024_threads.txt).  Pixels and scanlines are
//      closed forms of the plan -- DDA step k is (base + k, (minor + k * step) >> 16), row y0 + k of a fill piece has x_i + k * d_i on chain
//      i, exactly what the sequential `x += dx` reaches.  The primitives only ever write one value, so the image is the union of their pixel
//      sets and the order is irrelevant: threads set bits of the band's bitmask in LDS (64 rows x W bits = 4 KB at W = 512) with ds_or.
// The band is then expanded to the output tensor with coalesced 16-byte stores: the kernel's HBM traffic is the output itself (1 MiB fp32 per
// frame) plus ~1 KB of points.
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
constexpr int NT = 1024;                    // threads per workgroup: the kernel is a few thousand dependent instructions per thread, and 16 waves per CU
                                            // (4 per SIMD) is what lets the SIMDs interleave them (one wave per SIMD: ~8 cycles per instruction, 76 us per launch)

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

// num / den with C semantics (truncation toward zero) for den > 0, |num| < 2^52: both are exact doubles, and a quotient that is not an integer is at
// least 1 / |num| > 2^-52 (relative) away from the next one, so the correctly rounded double quotient truncates to the right integer already; the
// remainder test is a guard that costs four instructions (tests/test_raster.py runs the recipe against integer division, adversarial quotients
// included).  The kernel's numerators are (dy << 16) with |dy| <= 2^28 and 2 (xe - xs) + h with 16.16 coordinates of int32 points: < 2^50.
__device__ __forceinline__ long long div_exact(long long num, long long den)
{
    long long q = (long long)((double)num / (double)den);
    const long long r = num - q * den;
    if (num >= 0) { if (r < 0) --q; else if (r >= den) ++q; }
    else { if (r > 0) ++q; else if (r <= -den) --q; }
    return q;
}

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
    long long minor;
    int step;              // 16.16 slope of the minor coordinate: |minor delta| <= |major delta|, so it fits 17 bits + sign and k * step is one
                           // v_mad_i64_i32 instead of a 64 x 64 multiply per pixel
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
    r.step = (int)(xmajor ? div_exact(dy * ONE, ax | 1) : div_exact(dx * ONE, ay | 1));
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

// convex quad in 16.16, interior: walking the left and right edge chains from the top vertex.  The chains change edge at a handful of event
// rows; between two events row y + k has x_i + k * d_i on chain i, exactly what k sequential `x += dx` give -- one Piece per stretch.  (The
// outline goes through plan_dda, one side per thread.)
__device__ void plan_fill(const Band &b, const P2 (&v)[4], EdgePlan &pl)
{
    constexpr int N = 4;
    constexpr long long HALF = ONE >> 1;
    long long xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    int imin = 0;
    int np = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (v[i].y < ymin) { ymin = v[i].y; imin = i; }
        ymax = v[i].y > ymax ? v[i].y : ymax;
        xmax = v[i].x > xmax ? v[i].x : xmax;
        xmin = v[i].x < xmin ? v[i].x : xmin;
    }
    xmin = (xmin + HALF) >> SHIFT; xmax = (xmax + HALF) >> SHIFT;
    ymin = (ymin + HALF) >> SHIFT; ymax = (ymax + HALF) >> SHIFT;
    pl.npieces = 0;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= b.w || (int)ymin >= b.h) return;
    if (ymax > b.h - 1) ymax = b.h - 1;
    int idx[2] = {imin, imin}, ye[2], y = (int)ymin, edges = N;
    const int di[2] = {1, N - 1};
    long long x[2] = {-ONE, -ONE}, dxr[2] = {0, 0};
    ye[0] = ye[1] = y;
    // vertex access with a runtime index: 4 entries, resolved with selects (no scratch)
    auto vx = [&](int i) { return i == 0 ? v[0].x : i == 1 ? v[1].x : i == 2 ? v[2].x : v[3].x; };
    auto vy = [&](int i) { return i == 0 ? v[0].y : i == 1 ? v[1].y : i == 2 ? v[2].y : v[3].y; };
    while (y <= (int)ymax && np < MAX_PIECES) {
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
                    dxr[i] = div_exact((xe - xs) * 2 + (ty - y), 2 * (ty - y));
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
        Piece &q = pl.pc[np++];
        q.y0 = y; q.n = ynext - y; q.x0 = x[0]; q.d0 = dxr[0]; q.x1 = x[1]; q.d1 = dxr[1];
        x[0] += (long long)(ynext - y) * dxr[0];
        x[1] += (long long)(ynext - y) * dxr[1];
        y = ynext;
    }
    pl.npieces = np;
}

// step 1, thread per edge: the quad around the segment (OpenCV's ThickLine), the end caps' centre and radius
__device__ void plan_edge(int x0, int y0, int x1, int y1, int thickness, EdgePlan &pl, P2 (&q)[4])
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
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = p0;
    if (fabs(r) > 2.2204460492503131e-16) {
        r = ((double)half + (double)odd * (double)ONE * 0.5) / __dsqrt_rn(r);
        const long long ox = __double2ll_rn(dy * r), oy = __double2ll_rn(dx * r);      // round half to even, like cvRound
        q[0] = {p0.x + ox, p0.y + oy}; q[1] = {p0.x - ox, p0.y - oy}; q[2] = {p1.x - ox, p1.y - oy}; q[3] = {p1.x + ox, p1.y + oy};
        pl.hasquad = 1;
    }
    pl.radius = (half + (int)(ONE >> 1)) >> SHIFT;
}

// ---- step 3: the primitives' steps (one pixel of an outline, one scanline of a fill piece, one end cap) are numbered through, band-limited, and
// dealt to the threads in equal contiguous shares ----
// steps of a primitive inside this band: DDA = its far end point + pixels k = kbeg .. kend, piece = rows kbeg .. kend - 1
__device__ __forceinline__ void dda_range(const Band &b, const DdaRec &r, int &kbeg, int &kend)     // inclusive; empty when kend < kbeg
{
    kbeg = 0; kend = r.count;
    if (!r.xmajor) {                                          // y = base + k: only the rows of this band
        if (b.y0 - r.base > kbeg) kbeg = b.y0 - r.base;
        if (b.y0 + b.rows - 1 - r.base < kend) kend = b.y0 + b.rows - 1 - r.base;
    }
}
__device__ __forceinline__ void piece_range(const Band &b, const Piece &q, int &kbeg, int &kend)    // half-open
{
    kbeg = b.y0 - q.y0;
    if (kbeg < 0) kbeg = 0;
    kend = b.y0 + b.rows - q.y0;
    if (kend > q.n) kend = q.n;
}
__device__ __forceinline__ void dda_step(const Band &b, const DdaRec &r, int k)
{
    const int mn = (int)((r.minor + (long long)k * r.step) >> SHIFT);
    if (r.xmajor) dot(b, r.base + k, mn); else dot(b, mn, r.base + k);
}
__device__ __forceinline__ void piece_step(const Band &b, const Piece &q, int k)
{
    constexpr long long HALF = ONE >> 1;
    const int yy = q.y0 + k;
    if (yy < 0) return;
    const long long xa = q.x0 + k * q.d0, xb = q.x1 + k * q.d1;
    const long long lo = xa > xb ? xb : xa, hi = xa > xb ? xa : xb;
    const int xl = (int)((lo + HALF) >> SHIFT), xr = (int)((hi + HALF) >> SHIFT);
    if (xr >= 0 && xl < b.w) span(b, yy, xl < 0 ? 0 : xl, xr >= b.w ? b.w - 1 : xr);
}

// filled midpoint circle: horizontal spans, four per step of the walk
__device__ void disc(const Band &b, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    auto row = [&](int y, int xl, int xr) {
        if ((unsigned)y >= (unsigned)b.h || xl >= b.w || xr < 0) return;
        span(b, y, xl < 0 ? 0 : xl, xr > b.w - 1 ? b.w - 1 : xr);
    };
    if (!(cx - radius < b.w && cx + radius >= 0 && cy - radius < b.h && cy + radius >= 0)) return;
    if (cy + radius < b.y0 || cy - radius >= b.y0 + b.rows) return;
    while (dx >= dy) {
        row(cy - dy, cx - dx, cx + dx);
        row(cy + dy, cx - dx, cx + dx);
        row(cy - dx, cx - dy, cx + dy);
        row(cy + dx, cx - dy, cx + dy);
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

// phase stamps of tools/raster_stamps.py (builds with -DLSPRASTER_STAMPS only): shader clock at the phase boundaries of the first 64 workgroups
#ifdef LSPRASTER_STAMPS
__device__ unsigned long long g_stamps[64][8];
#define RSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 64) g_stamps[blockIdx.x + gridDim.x * blockIdx.y][i] = __builtin_readcyclecounter(); } while (0)
#else
#define RSTAMP(i) do {} while (0)
#endif

constexpr int EDGE_CHUNK = 96;             // edges planned per round (41 KB of plans + 6 KB of vertices in LDS)
constexpr int kScanBytes = (2 * 12 * EDGE_CHUNK + 4 + 2 * (NT / 64)) * 4;       // compacted (first step, primitive) lists of a chunk + wave totals
constexpr int kDiscSteps = 4;              // an end cap (four spans per step of its midpoint walk, radius + 1 steps) weighs this many outline pixels in the deal

__global__ __launch_bounds__(NT) void edge_map_band(const Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    EdgePlan *plans = reinterpret_cast<EdgePlan *>(lds);                       // [EDGE_CHUNK]
    P2 *verts = reinterpret_cast<P2 *>(lds + sizeof(EdgePlan) * EDGE_CHUNK);   // [EDGE_CHUNK][4]
    int *coff = reinterpret_cast<int *>(lds + (sizeof(EdgePlan) + 4 * sizeof(P2)) * EDGE_CHUNK);        // [12 * EDGE_CHUNK + 1] first step of the i-th non-empty primitive
    int *cprim = coff + 12 * EDGE_CHUNK + 4;                                                             // [12 * EDGE_CHUNK] which primitive that is
    int *wsum = cprim + 12 * EDGE_CHUNK;                                                                 // [2][NT / 64] wave totals
    unsigned *bits = reinterpret_cast<unsigned *>(lds + (sizeof(EdgePlan) + 4 * sizeof(P2)) * EDGE_CHUNK + kScanBytes);
    const int tid = threadIdx.x, frame = blockIdx.y;
    RSTAMP(0);
    Band b;
    b.bits = bits; b.w = p.w; b.h = p.h; b.words = p.w >> 5;
    b.y0 = blockIdx.x * BAND;
    b.rows = p.h - b.y0 < BAND ? p.h - b.y0 : BAND;
    for (int i = tid; i < BAND * b.words; i += NT) bits[i] = 0u;
    const size_t base = (size_t)frame * p.npoints * 2;
    const int reach = (p.thickness >> 1) + 2;                  // a primitive never leaves its end points' box by more than this
    for (int c0 = 0; c0 < p.nseg; c0 += EDGE_CHUNK) {
        const int nc = p.nseg - c0 < EDGE_CHUNK ? p.nseg - c0 : EDGE_CHUNK;
        __syncthreads();                                       // bits zeroed / the previous chunk's plans are no longer read
        // 1. thread per edge
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
                    P2 q[4];
                    plan_edge(x0, y0, x1, y1, p.thickness, pl, q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) verts[4 * tid + i] = q[i];
                }
            }
        }
        __syncthreads();
        RSTAMP(1);
        // 2. thread per (edge, outline side) and per (edge, fill); item = kind * nc + edge, so a wave mostly holds one kind
        for (int w = tid; w < 5 * nc; w += NT) {
            const int kind = w / nc, e = w - kind * nc;
            EdgePlan &pl = plans[e];
            if (pl.skip || !pl.hasquad) continue;
            if (kind < 4) {
                plan_dda(b, verts[4 * e + ((kind + 3) & 3)], verts[4 * e + kind], pl.dda[kind]);
            } else {
                const P2 v[4] = {verts[4 * e], verts[4 * e + 1], verts[4 * e + 2], verts[4 * e + 3]};
                plan_fill(b, v, pl);
            }
        }
        __syncthreads();
        RSTAMP(2);
        // 3. primitive w = kind * nc + edge (4 outline DDAs, up to 6 fill pieces, 2 end caps): its number of steps inside this band ...
        const int nprim = 12 * nc;
        auto steps_of = [&](int w) -> int {
            const int kind = w / nc, e = w - kind * nc;
            const EdgePlan &pl = plans[e];
            if (pl.skip) return 0;
            if (kind < 4) {
                if (!pl.hasquad || !pl.dda[kind].valid) return 0;
                int k0, k1;
                dda_range(b, pl.dda[kind], k0, k1);
                return 1 + (k1 >= k0 ? k1 - k0 + 1 : 0);       // the far end point is step 0
            }
            if (kind < 10) {
                if (!pl.hasquad || kind - 4 >= pl.npieces) return 0;
                int k0, k1;
                piece_range(b, pl.pc[kind - 4], k0, k1);
                return k1 > k0 ? k1 - k0 : 0;
            }
            const int cy = kind == 10 ? pl.cy0 : pl.cy1;
            return (cy + pl.radius < b.y0 || cy - pl.radius >= b.y0 + b.rows) ? 0 : kDiscSteps;
        };
        // ... exclusive prefix sums over the primitives, of their steps and of "has any" (thread: PER consecutive primitives; wave scans; wave totals
        // through LDS): the non-empty primitives -- a small minority in any one band -- are compacted into (primitive, first step) pairs
        constexpr int PER = (12 * EDGE_CHUNK + NT - 1) / NT, NW = NT / 64;
        int mine[PER], sum = 0, cnt = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int w = tid * PER + i;
            mine[i] = w < nprim ? steps_of(w) : 0;
            sum += mine[i];
            cnt += mine[i] > 0;
        }
        int incl = sum, cincl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d), cup = __shfl_up(cincl, d);
            if ((tid & 63) >= d) { incl += up; cincl += cup; }
        }
        if ((tid & 63) == 63) { wsum[tid >> 6] = incl; wsum[NW + (tid >> 6)] = cincl; }
        __syncthreads();
        int wbase = 0, total = 0, cbase = 0, ncomp = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int t = wsum[i], c = wsum[NW + i];
            wbase += i < (tid >> 6) ? t : 0; total += t;
            cbase += i < (tid >> 6) ? c : 0; ncomp += c;
        }
        int run = wbase + incl - sum, slot = cbase + cincl - cnt;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (mine[i] > 0) { cprim[slot] = tid * PER + i; coff[slot] = run; ++slot; }
            run += mine[i];
        }
        if (tid == 0) coff[ncomp] = total;
        __syncthreads();
        RSTAMP(6);
        // ... and every thread takes an equal contiguous share of the steps: ONE binary search for the compacted primitive its first step belongs
        // to, then a walk through the list (searching per primitive, or walking the uncompacted list with its runs of empty primitives, each cost
        // more than the drawing itself: profiles/r03_raster_stamps_dealt_steps.txt)
        const int share = (total + NT - 1) / NT;
        int s0 = tid * share;
        const int s1 = s0 + share < total ? s0 + share : total;
        if (s0 < s1) {
            int lo = 0, hi = ncomp;                           // the last i with coff[i] <= s0
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= s0) lo = mid; else hi = mid; }
            for (int i = lo; s0 < s1; ++i) {
                const int w = cprim[i], beg = coff[i];
                const int first = s0 - beg;
                int last = coff[i + 1] - beg;                 // steps [first, last) of primitive w
                if (last - first > s1 - s0) last = first + (s1 - s0);
                const int kind = w / nc, e = w - kind * nc;
                const EdgePlan &pl = plans[e];
                if (kind < 4) {
                    const DdaRec r = pl.dda[kind];
                    int k0, k1;
                    dda_range(b, r, k0, k1);
                    for (int j = first; j < last; ++j) {
                        if (j == 0) dot(b, r.ex, r.ey); else dda_step(b, r, k0 + j - 1);
                    }
                } else if (kind < 10) {
                    const Piece q = pl.pc[kind - 4];
                    int k0, k1;
                    piece_range(b, q, k0, k1);
                    for (int j = first; j < last; ++j) piece_step(b, q, k0 + j);
                } else if (first == 0) {                      // an end cap counts kDiscSteps steps; the first one draws it
                    disc(b, kind == 10 ? pl.cx0 : pl.cx1, kind == 10 ? pl.cy0 : pl.cy1, pl.radius);
                }
                s0 += last - first;
            }
        }
    }
    RSTAMP(3);
    __syncthreads();
    RSTAMP(4);
    // expand the band: 4 pixels per thread and step
    const int quads = b.rows * (p.w >> 2);
    for (int q = tid; q < quads; q += NT) {
        const int r = q / (p.w >> 2), x = (q - r * (p.w >> 2)) * 4;
        const unsigned nib = (bits[r * b.words + (x >> 5)] >> (x & 31)) & 15u;
        const size_t o = ((size_t)frame * p.h + b.y0 + r) * p.w + x;
        if (p.out_f32)
            *reinterpret_cast<float4 *>(p.out_f32 + o) = make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)(nib >> 3));
        if (p.out_u8)
            *reinterpret_cast<unsigned *>(p.out_u8 + o) = (nib & 1u ? 0xffu : 0u) | (nib & 2u ? 0xff00u : 0u) | (nib & 4u ? 0xff0000u : 0u) | (nib & 8u ? 0xff000000u : 0u);
    }
    RSTAMP(5);
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

}  // namespace lspraster

using namespace lspraster;

extern "C" {

const char *lspraster_last_error(void) { return g_err.c_str(); }

#ifdef LSPRASTER_STAMPS
int lspraster_debug_stamps(unsigned long long *host_64x8)      // tools/raster_stamps.py
{
    return hipMemcpyFromSymbol(host_64x8, HIP_SYMBOL(lspraster::g_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif

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
    const size_t smem = (sizeof(EdgePlan) + 4 * sizeof(P2)) * EDGE_CHUNK + kScanBytes + (size_t)BAND * (width / 32) * sizeof(unsigned);      // <= 57 KB + 32 KB
    if (smem > 64 * 1024) {                                    // widths above 2048 only
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&edge_map_band), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (ea != hipSuccess) return fail(LSPRASTER_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(ea));
    }
    hipLaunchKernelGGL(edge_map_band, dim3((height + BAND - 1) / BAND, batch), dim3(NT), smem, static_cast<hipStream_t>(hip_stream), p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSPRASTER_ERR_HIP, std::string("edge_map_band launch: ") + hipGetErrorString(e));
    return LSPRASTER_OK;
}

}  // extern "C"
