/* ORACLE (test infrastructure; never linked into or called by the product).
 *
 * CPU restatement of the landmark rasterisation of the render loop:
 *   datasets/face_dataset.py:276-323  get_data_test_mode -> get_feature_image -> draw_face_feature_maps /
 *                                     draw_shoulder_points: cv2.line(img, pt1, pt2, 255, 2) per edge of part_list (:34-42)
 * The arithmetic lives in a third-party dependency absent from /root/reference AND from this image: OpenCV
 * (requirements.txt pins opencv_python==4.4.0.40).  This file restates the PUBLISHED algorithm of OpenCV 4.4.0
 * modules/imgproc/src/drawing.cpp for an 8-bit single-channel image, LINE_8, shift 0:
 *   cv::line -> ThickLine (thickness > 1): quad of half-width thickness/2 in XY_SHIFT fixed point -> FillConvexPoly
 *   (outline with Line2 -- a 16.16 DDA after clipLine -- then the two-edge scanline fill), plus a filled Circle of radius
 *   (thickness + 1) / 2 at both end points.
 * PARITY UNPINNED: no OpenCV build exists here, the reference has no test vectors for this path, and the restatement was
 * written from the published algorithm, not executed against cv2.  The GPU kernel is held bit-exact to THIS file; this
 * file's agreement with cv2 itself is unverified.  Known-answer anchors checked in tests/test_raster.py: a horizontal
 * thickness-2 line covers three rows (the well-known OpenCV behaviour), and the radius-1 end cap is the 5-pixel plus.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

typedef struct { int64_t x, y; } pt2l;
typedef struct { uint8_t *data; int w, h; } img8;

static void put(img8 *im, int x, int y) { if (x >= 0 && x < im->w && y >= 0 && y < im->h) im->data[(size_t)y * im->w + x] = 255; }
static void hline(img8 *im, int y, int xl, int xr) { for (int x = xl; x <= xr; ++x) im->data[(size_t)y * im->w + x] = 255; }   /* callers clip */

/* cv::clipLine(Size2l, Point2l&, Point2l&) */
static int clip_line(int64_t width, int64_t height, pt2l *p1, pt2l *p2)
{
    int c1, c2;
    const int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    int64_t *x1 = &p1->x, *y1 = &p1->y, *x2 = &p2->x, *y2 = &p2->y;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* Line2: fixed-point (16.16) DDA between two sub-pixel points */
static void line2(img8 *im, pt2l pt1, pt2l pt2)
{
    int64_t dx, dy, ax, ay, i, j, x_step, y_step;
    int ecount;
    if (!clip_line((int64_t)im->w << XY_SHIFT, (int64_t)im->h << XY_SHIFT, &pt1, &pt2)) return;
    dx = pt2.x - pt1.x;
    dy = pt2.y - pt1.y;
    j = dx < 0 ? -1 : 0;
    ax = (dx ^ j) - j;
    i = dy < 0 ? -1 : 0;
    ay = (dy ^ i) - i;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        pt1.x ^= pt2.x & j; pt2.x ^= pt1.x & j; pt1.x ^= pt2.x & j;
        pt1.y ^= pt2.y & j; pt2.y ^= pt1.y & j; pt1.y ^= pt2.y & j;
        x_step = XY_ONE;
        y_step = (dy * XY_ONE) / (ax | 1);          /* (dy << XY_SHIFT) / (ax | 1): multiplication keeps a negative dy defined in C */
        ecount = (int)((pt2.x - pt1.x) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        pt1.x ^= pt2.x & i; pt2.x ^= pt1.x & i; pt1.x ^= pt2.x & i;
        pt1.y ^= pt2.y & i; pt2.y ^= pt1.y & i; pt1.y ^= pt2.y & i;
        x_step = (dx * XY_ONE) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((pt2.y - pt1.y) >> XY_SHIFT);
    }
    pt1.x += (XY_ONE >> 1);
    pt1.y += (XY_ONE >> 1);
    put(im, (int)((pt2.x + (XY_ONE >> 1)) >> XY_SHIFT), (int)((pt2.y + (XY_ONE >> 1)) >> XY_SHIFT));
    if (ax > ay) {
        pt1.x >>= XY_SHIFT;
        while (ecount >= 0) {
            put(im, (int)pt1.x, (int)(pt1.y >> XY_SHIFT));
            pt1.x++;
            pt1.y += y_step;
            ecount--;
        }
    } else {
        pt1.y >>= XY_SHIFT;
        while (ecount >= 0) {
            put(im, (int)(pt1.x >> XY_SHIFT), (int)pt1.y);
            pt1.x += x_step;
            pt1.y++;
            ecount--;
        }
    }
    (void)x_step;
}

/* FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT) */
static void fill_convex_poly(img8 *im, const pt2l *v, int npts)
{
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    const int shift = XY_SHIFT;
    const int delta = 1 << shift >> 1;
    int i, y, imin = 0, edges = npts;
    int64_t xmin, xmax, ymin, ymax;
    const int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
    pt2l p0 = v[npts - 1];
    xmin = xmax = v[0].x;
    ymin = ymax = v[0].y;
    for (i = 0; i < npts; i++) {
        pt2l p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        if (p.y > ymax) ymax = p.y;
        if (p.x > xmax) xmax = p.x;
        if (p.x < xmin) xmin = p.x;
        line2(im, p0, p);                       /* shift != 0: the outline goes through the sub-pixel DDA */
        p0 = p;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= im->w || (int)ymin >= im->h) return;
    if (ymax > im->h - 1) ymax = im->h - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((v[idx].y + delta) >> shift);
                    if (ty > y) {
                        const int64_t xs = v[idx0].x, xe = v[idx].x;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < im->w) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= im->w) xx2 = im->w - 1;
                hline(im, y, xx1, xx2);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* Circle(img, center, radius, color, fill = 1): midpoint circle, horizontal spans */
static void circle_filled(img8 *im, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    const int inside = cx >= radius && cx < im->w - radius && cy >= radius && cy < im->h - radius;
    while (dx >= dy) {
        int mask;
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (inside) {
            hline(im, y11, x11, x12);
            hline(im, y12, x11, x12);
            hline(im, y21, x21, x22);
            hline(im, y22, x21, x22);
        } else if (x11 < im->w && x12 >= 0 && y21 < im->h && y22 >= 0) {
            if (x11 < 0) x11 = 0;
            if (x12 > im->w - 1) x12 = im->w - 1;
            if ((unsigned)y11 < (unsigned)im->h) hline(im, y11, x11, x12);
            if ((unsigned)y12 < (unsigned)im->h) hline(im, y12, x11, x12);
            if (x21 < im->w && x22 >= 0) {
                if (x21 < 0) x21 = 0;
                if (x22 > im->w - 1) x22 = im->w - 1;
                if ((unsigned)y21 < (unsigned)im->h) hline(im, y21, x21, x22);
                if ((unsigned)y22 < (unsigned)im->h) hline(im, y22, x21, x22);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* ThickLine(img, p0, p1, color, thickness > 1, LINE_8, flags = 3, shift = 0) */
static void thick_line(img8 *im, int x0, int y0, int x1, int y1, int thickness)
{
    static const double INV_XY_ONE = 1. / XY_ONE;
    pt2l p0 = {(int64_t)x0 * XY_ONE, (int64_t)y0 * XY_ONE}, p1 = {(int64_t)x1 * XY_ONE, (int64_t)y1 * XY_ONE};
    pt2l pt[4], dp = {0, 0};
    const double dx = (double)(p0.x - p1.x) * INV_XY_ONE, dy = (double)(p1.y - p0.y) * INV_XY_ONE;
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    int i;
    thickness <<= XY_SHIFT - 1;
    if (fabs(r) > 2.2204460492503131e-16) {
        r = (thickness + odd * XY_ONE * 0.5) / sqrt(r);
        dp.x = (int64_t)lrint(dy * r);           /* cvRound: round half to even */
        dp.y = (int64_t)lrint(dx * r);
        pt[0].x = p0.x + dp.x; pt[0].y = p0.y + dp.y;
        pt[1].x = p0.x - dp.x; pt[1].y = p0.y - dp.y;
        pt[2].x = p1.x - dp.x; pt[2].y = p1.y - dp.y;
        pt[3].x = p1.x + dp.x; pt[3].y = p1.y + dp.y;
        fill_convex_poly(im, pt, 4);
    }
    for (i = 0; i < 2; i++) {
        const int cx = (int)((p0.x + (XY_ONE >> 1)) >> XY_SHIFT), cy = (int)((p0.y + (XY_ONE >> 1)) >> XY_SHIFT);
        circle_filled(im, cx, cy, (thickness + (XY_ONE >> 1)) >> XY_SHIFT);
        p0 = p1;
    }
}

/* One edge map: `segments` = nseg (a, b) index pairs into `points` = npoints (x, y) integer pairs (already int()-truncated,
 * face_dataset.py:301-302, :318-319).  img is zeroed here (np.zeros((h, w), np.uint8), :313). */
void raster_edge_map(const int32_t *points, int npoints, const int32_t *segments, int nseg, int thickness, int height, int width,
                     uint8_t *img)
{
    img8 im = {img, width, height};
    memset(img, 0, (size_t)height * width);
    for (int s = 0; s < nseg; ++s) {
        const int a = segments[2 * s], b = segments[2 * s + 1];
        if (a < 0 || a >= npoints || b < 0 || b >= npoints) continue;
        thick_line(&im, points[2 * a], points[2 * a + 1], points[2 * b], points[2 * b + 1], thickness);
    }
}
