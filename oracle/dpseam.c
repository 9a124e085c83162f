/*
 * dpseam.c -- CPU ORACLE (test infrastructure): the dynamic-programming seam finder.
 *
 * Restates, function by function, DpSeamFinder of the reference (R64 = /root/reference/stitching/
 * OpenCV2.4.11-Stitching-64bit/OpenCV2.4.11-Stitching):
 *
 *   find                   R64/src/seam_finders.cpp:163-189   (pair order: ImagePairLess, detail/seam_finders.hpp:117-139)
 *   process                :192-248      findComponents   :251-308      findEdges           :311-375
 *   resolveConflicts       :378-516      computeGradients :519-543      hasOnlyOneNeighbor  :546-552
 *   closeToContour         :555-575      getSeamTips      :578-676      computeCosts        :703-778
 *   estimateSeam           :781-909      updateLabelsUsingSeam :912-1045
 *
 * OpenCV 2.4.11 primitives used by it and restated here (parity unpinned, like the rest of this oracle -- the
 * reference ships no test for this class and cannot be built in this image -- EXCEPT Sobel's float sums and the 8-bit gray constants,
 * which were read off the reference's shipped opencv_imgproc2411.dll: oracle/REFERENCE_BINARY_AUDIT.md):
 *   cv::floodFill(Mat_<int>, seed, newVal)         4-connected region of pixels EQUAL to the seed pixel (loDiff = upDiff = 0)
 *   cv::partition(points, labels, ClosePoints(10)) classes = connected components of the predicate graph, numbered in the
 *                                                  order of their first member (core/operations.hpp: union-find, then "enumerate classes")
 *   std::sort + std::reverse on at most 32 pairs   MSVC 2010's std::sort is an insertion sort below 32 elements (_ISORT_MAX), i.e.
 *                                                  stable; the driver stitches at most 8 cameras = 28 pairs.  More than 32 pairs with
 *                                                  equal centre distances are ordered differently by an introsort: not restated.
 *   cvRound(double)                                round-half-even (cvtsd2si)
 *   cvtColor(BGR2GRAY) on CV_32F                   b * 0.114f + g * 0.587f + r * 0.299f, left to right
 *   Sobel(CV_32F, ksize 3, BORDER_REFLECT_101)     row pass (s[x+1] - s[x-1]  or  s[x-1] + 2 s[x] + s[x+1]) then column pass; the
 *                                                  three-term sums associated as 2.4.11's SSE bodies / scalar tails do, by
 *                                                  column position (see gradients()) -- COLOR_GRAD only
 *
 * Images are CV_32FC3 (what Stitcher / the command-line sample feed it: images_warped[i].convertTo(..., CV_32F),
 * CLI:488-489) or CV_8UC3, tightly packed, BGR; masks CV_8U tightly packed, updated in place.
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mst_oracle.h"

enum { FIRST = 1, SECOND = 2, INTERS = 4, INTERS_FIRST = 5, INTERS_SECOND = 6 };

typedef struct { int x, y; } Pt;
typedef struct { Pt *p; int n, cap; } PtVec;

static void pv_push(PtVec *v, int x, int y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 64;
        v->p = (Pt *)realloc(v->p, (size_t)v->cap * sizeof(Pt));
    }
    v->p[v->n].x = x; v->p[v->n].y = y; v->n++;
}

typedef struct {
    int cost_func;                       /* 0 COLOR, 1 COLOR_GRAD */
    int utlx, utly, uw, uh;              /* unionTl_, unionSize_ */
    uint8_t *mask1, *mask2, *contour1, *contour2;      /* uw x uh */
    int *labels;                         /* uw x uh */
    int ncomps;
    int *states;
    Pt *tls, *brs;
    PtVec *contours;
    int comps_cap;
    /* edges_: std::set<pair<int,int>> as a dense matrix (ordered iteration = row-major scan) */
    uint8_t *edge;                       /* ncomps x ncomps */
    float *gradx1, *grady1, *gradx2, *grady2;          /* image-sized, COLOR_GRAD */
} Dp;

/* cv::floodFill on a CV_32S image, 4-connectivity, zero tolerance: the connected region of pixels equal to the seed */
static void flood_fill_i32(int *img, int w, int h, int sx, int sy, int new_val)
{
    const int old = img[(size_t)sy * w + sx];
    if (old == new_val) return;
    size_t cap = 1024, n = 0;
    int *stack = (int *)malloc(cap * 2 * sizeof(int));
    stack[0] = sx; stack[1] = sy; n = 1;
    img[(size_t)sy * w + sx] = new_val;
    while (n) {
        --n;
        const int x = stack[2 * n], y = stack[2 * n + 1];
        static const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
        for (int k = 0; k < 4; ++k) {
            const int xx = x + dx[k], yy = y + dy[k];
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            if (img[(size_t)yy * w + xx] != old) continue;
            img[(size_t)yy * w + xx] = new_val;
            if (n == cap) { cap *= 2; stack = (int *)realloc(stack, cap * 2 * sizeof(int)); }
            stack[2 * n] = xx; stack[2 * n + 1] = yy; ++n;
        }
    }
    free(stack);
}

#define L(d, y, x) ((d)->labels[(size_t)(y) * (d)->uw + (x)])

static void comps_reserve(Dp *d, int n)
{
    if (n <= d->comps_cap) return;
    int cap = d->comps_cap ? d->comps_cap : 16;
    while (cap < n) cap *= 2;
    d->states = (int *)realloc(d->states, (size_t)cap * sizeof(int));
    d->tls = (Pt *)realloc(d->tls, (size_t)cap * sizeof(Pt));
    d->brs = (Pt *)realloc(d->brs, (size_t)cap * sizeof(Pt));
    d->contours = (PtVec *)realloc(d->contours, (size_t)cap * sizeof(PtVec));
    for (int i = d->comps_cap; i < cap; ++i) { d->contours[i].p = NULL; d->contours[i].n = 0; d->contours[i].cap = 0; }
    d->comps_cap = cap;
}

/* seam_finders.cpp:251-308 */
static void find_components(Dp *d)
{
    const int w = d->uw, h = d->uh;
    d->ncomps = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int m1 = d->mask1[(size_t)y * w + x], m2 = d->mask2[(size_t)y * w + x];
            L(d, y, x) = (m1 && m2) ? INT_MAX : (m1 ? INT_MAX - 1 : (m2 ? INT_MAX - 2 : 0));
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (L(d, y, x) >= INT_MAX - 2) {
                comps_reserve(d, d->ncomps + 1);
                d->states[d->ncomps] = L(d, y, x) == INT_MAX ? INTERS : (L(d, y, x) == INT_MAX - 1 ? FIRST : SECOND);
                flood_fill_i32(d->labels, w, h, x, y, ++d->ncomps);
                d->tls[d->ncomps - 1].x = x; d->tls[d->ncomps - 1].y = y;
                d->brs[d->ncomps - 1].x = x + 1; d->brs[d->ncomps - 1].y = y + 1;
                d->contours[d->ncomps - 1].n = 0;
            }
            if (L(d, y, x)) {
                const int l = L(d, y, x), ci = l - 1;
                if (x < d->tls[ci].x) d->tls[ci].x = x;
                if (y < d->tls[ci].y) d->tls[ci].y = y;
                if (x + 1 > d->brs[ci].x) d->brs[ci].x = x + 1;
                if (y + 1 > d->brs[ci].y) d->brs[ci].y = y + 1;
                if ((x == 0 || L(d, y, x - 1) != l) || (x == w - 1 || L(d, y, x + 1) != l) ||
                    (y == 0 || L(d, y - 1, x) != l) || (y == h - 1 || L(d, y + 1, x) != l))
                    pv_push(&d->contours[ci], x, y);
            }
        }
}

/* seam_finders.cpp:311-375: an edge (ci, cj) exists iff some contour pixel of one touches the other (the weights only
 * ever decide "> 0"); both directions are inserted */
static void find_edges(Dp *d)
{
    const int n = d->ncomps, w = d->uw, h = d->uh;
    d->edge = (uint8_t *)calloc((size_t)n * n + 1, 1);
    for (int ci = 0; ci < n; ++ci)
        for (int i = 0; i < d->contours[ci].n; ++i) {
            const int x = d->contours[ci].p[i].x, y = d->contours[ci].p[i].y, l = ci + 1;
            int o;
            if (x > 0 && (o = L(d, y, x - 1)) && o != l) { d->edge[(size_t)ci * n + o - 1] = 1; d->edge[(size_t)(o - 1) * n + ci] = 1; }
            if (y > 0 && (o = L(d, y - 1, x)) && o != l) { d->edge[(size_t)ci * n + o - 1] = 1; d->edge[(size_t)(o - 1) * n + ci] = 1; }
            if (x < w - 1 && (o = L(d, y, x + 1)) && o != l) { d->edge[(size_t)ci * n + o - 1] = 1; d->edge[(size_t)(o - 1) * n + ci] = 1; }
            if (y < h - 1 && (o = L(d, y + 1, x)) && o != l) { d->edge[(size_t)ci * n + o - 1] = 1; d->edge[(size_t)(o - 1) * n + ci] = 1; }
        }
}

static int has_only_one_neighbor(const Dp *d, int comp)
{
    int cnt = 0;
    for (int j = 0; j < d->ncomps; ++j) cnt += d->edge[(size_t)comp * d->ncomps + j];
    return cnt == 1;
}

static int close_to_contour(const Dp *d, int y, int x, const uint8_t *cm)
{
    for (int dy = -2; dy <= 2; ++dy) {
        if (y + dy < 0 || y + dy >= d->uh) continue;
        for (int dx = -2; dx <= 2; ++dx)
            if (x + dx >= 0 && x + dx < d->uw && cm[(size_t)(y + dy) * d->uw + x + dx]) return 1;
    }
    return 0;
}

static int cv_round_d(double v) { return (int)lrint(v); }       /* default rounding mode: half to even */

/* seam_finders.cpp:578-676 */
static int get_seam_tips(const Dp *d, int comp1, int comp2, Pt *p1, Pt *p2)
{
    const int w = d->uw, h = d->uh, l2 = comp2 + 1;
    PtVec sp = {0, 0, 0};
    for (int i = 0; i < d->contours[comp1].n; ++i) {
        const int x = d->contours[comp1].p[i].x, y = d->contours[comp1].p[i].y;
        if (close_to_contour(d, y, x, d->contour1) && close_to_contour(d, y, x, d->contour2) &&
            ((x > 0 && L(d, y, x - 1) == l2) || (y > 0 && L(d, y - 1, x) == l2) ||
             (x < w - 1 && L(d, y, x + 1) == l2) || (y < h - 1 && L(d, y + 1, x) == l2)))
            pv_push(&sp, x, y);
    }
    if (sp.n < 2) { free(sp.p); return 0; }
    /* cv::partition with ClosePoints(10): dist^2 < 100; classes numbered by their first member */
    int *lab = (int *)malloc((size_t)sp.n * sizeof(int));
    int *stack = (int *)malloc((size_t)sp.n * sizeof(int));
    for (int i = 0; i < sp.n; ++i) lab[i] = -1;
    int nlabels = 0;
    for (int i = 0; i < sp.n; ++i) {
        if (lab[i] >= 0) continue;
        int top = 0;
        stack[top++] = i; lab[i] = nlabels;
        while (top) {
            const int a = stack[--top];
            for (int b = 0; b < sp.n; ++b) {
                if (lab[b] >= 0) continue;
                const int ddx = sp.p[a].x - sp.p[b].x, ddy = sp.p[a].y - sp.p[b].y;
                if (ddx * ddx + ddy * ddy < 100) { lab[b] = nlabels; stack[top++] = b; }
            }
        }
        ++nlabels;
    }
    free(stack);
    int ok = 0;
    if (nlabels >= 2) {
        Pt *sum = (Pt *)calloc((size_t)nlabels, sizeof(Pt));
        int *cnt = (int *)calloc((size_t)nlabels, sizeof(int));
        for (int i = 0; i < sp.n; ++i) { sum[lab[i]].x += sp.p[i].x; sum[lab[i]].y += sp.p[i].y; cnt[lab[i]]++; }
        /* two most distant clusters (cy2 divides by size1: the reference's own slip, :634) */
        int idx[2] = {-1, -1};
        double max_dist = -1.7976931348623157e308;
        for (int i = 0; i < nlabels - 1; ++i)
            for (int j = i + 1; j < nlabels; ++j) {
                const double size1 = (double)cnt[i], size2 = (double)cnt[j];
                const double cx1 = cv_round_d(sum[i].x / size1), cy1 = cv_round_d(sum[i].y / size1);
                const double cx2 = cv_round_d(sum[j].x / size2), cy2 = cv_round_d(sum[j].y / size1);
                const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
                if (dist > max_dist) { max_dist = dist; idx[0] = i; idx[1] = j; }
            }
        Pt p[2];
        for (int k = 0; k < 2; ++k) {
            const double size = (double)cnt[idx[k]];
            const double cx = cv_round_d(sum[idx[k]].x / size), cy = cv_round_d(sum[idx[k]].y / size);
            double min_dist = 1.7976931348623157e308;
            p[k].x = p[k].y = 0;
            for (int j = 0; j < sp.n; ++j) {            /* members of the cluster, in their original order */
                if (lab[j] != idx[k]) continue;
                const double dist = (sp.p[j].x - cx) * (sp.p[j].x - cx) + (sp.p[j].y - cy) * (sp.p[j].y - cy);
                if (dist < min_dist) { min_dist = dist; p[k] = sp.p[j]; }
            }
        }
        *p1 = p[0]; *p2 = p[1];
        ok = 1;
        free(sum); free(cnt);
    }
    free(lab); free(sp.p);
    return ok;
}

typedef struct { const void *data; int w, h, is_float; } Img;

static float diff3(const Img *a, int y1, int x1, const Img *b, int y2, int x2)
{
    if (a->is_float) {
        const float *r1 = (const float *)a->data + ((size_t)y1 * a->w + x1) * 3;
        const float *r2 = (const float *)b->data + ((size_t)y2 * b->w + x2) * 3;
        const float d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
        return (d0 * d0 + d1 * d1) + d2 * d2;
    }
    const uint8_t *r1 = (const uint8_t *)a->data + ((size_t)y1 * a->w + x1) * 3;
    const uint8_t *r2 = (const uint8_t *)b->data + ((size_t)y2 * b->w + x2) * 3;
    const int d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
    return (float)(d0 * d0 + d1 * d1 + d2 * d2);
}

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

/* cvtColor(BGR2GRAY) + Sobel(CV_32F, dx / dy, ksize 3) :519-543 */
static void gradients(const Img *im, float **gx, float **gy)
{
    const int w = im->w, h = im->h;
    float *gray = (float *)malloc((size_t)w * h * sizeof(float));
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float b, g, r;
        if (im->is_float) {              /* RGB2Gray<float>, BGR order */
            const float *p = (const float *)im->data + 3 * i; b = p[0]; g = p[1]; r = p[2];
            gray[i] = b * 0.114f + g * 0.587f + r * 0.299f;
        } else {                         /* RGB2Gray<uchar>: 14-bit fixed point, the gray image is CV_8U (found by the second
                                          * restatement, tests/independent/dpseam_py.py: round 3 used the float form here) */
            const uint8_t *p = (const uint8_t *)im->data + 3 * i;
            (void)b; (void)g; (void)r;
            gray[i] = (float)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
        }
    }
    /* cv::Sobel(CV_32F -> CV_32F, ksize 3) = sepFilter2D with SymmRowSmallFilter<float, float, SymmRowSmallVec_32f> and
     * SymmColumnSmallFilter<Cast<float, float>, SymmColumnSmallVec_32f> (OpenCV 2.4.11 modules/imgproc/src/filter.cpp,
     * getLinearRowFilter / getLinearColumnFilter for ksize <= 5 / == 3).  The [-1 0 1] pass is one subtraction whatever code
     * runs it; the [1 2 1] pass is a sum of three floats whose ASSOCIATION depends on which loop of the filter produced the
     * element, and on non-integer data the forms differ by one ulp (round 4 used one form everywhere; on 8-bit sources every
     * form is exact).  With W = the image width (cn = 1), W8 = W & ~7, R = W - W8, as the x64 SSE2 build runs them:
     *   row [1 2 1]   x < W8                 SSE body, 8 per trip    a + ((m + m) + c)          _mm_add_ps(x0, _mm_add_ps(_mm_add_ps(x1, x1), x2))
     *                 W8 <= x < W8 + (R & ~1) scalar pairs            (a + m * 2) + c            S[-cn] + S[0]*2 + S[cn]
     *                 x == W - 1, R odd       generic tail            m * 2 + (c + a)            kx[0]*S[0] + kx[1]*(S[j] + S[-j])
     *   col [1 2 1]   x < W8                 SSE body                t + (b + (m + m))          _mm_add_ps(s0, _mm_add_ps(s4, _mm_add_ps(s2, s2)))
     *                 W8 <= x                 unrolled by four + rest  (t + m * 2) + b            S0[i] + S1[i]*2 + S2[i] + _delta
     * (a / c: left / right neighbour, t / b: row above / below, REFLECT_101 at the borders; _delta = 0 adds nothing).
     * PINNED against the reference's own binary: tools/ref_binary_audit.py disassembles the x64 Release opencv_imgproc2411.dll the
     * reference ships and runs, and prints the expression every store of these loops writes (oracle/REFERENCE_BINARY_AUDIT.md).
     * That audit corrected the first round-5 statement of this table, which gave the columns after the unrolled fours the generic
     * (t + b) + m * 2: in the shipped build every column past the SSE body is (t + m * 2) + b. */
    const int w8 = w & ~7, rem = w - w8;
    float *dxr = (float *)malloc((size_t)w * h * sizeof(float)), *smr = (float *)malloc((size_t)w * h * sizeof(float));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float a = gray[(size_t)y * w + reflect101(x - 1, w)], c = gray[(size_t)y * w + reflect101(x + 1, w)], m = gray[(size_t)y * w + x];
            dxr[(size_t)y * w + x] = c - a;
            float s;
            if (x < w8) { const float t = m + m; const float u = t + c; s = a + u; }
            else if (x < w8 + (rem & ~1)) { const float t = m * 2.f; const float u = a + t; s = u + c; }
            else { const float t = m * 2.f; const float u = c + a; s = t + u; }
            smr[(size_t)y * w + x] = s;
        }
    *gx = (float *)malloc((size_t)w * h * sizeof(float));
    *gy = (float *)malloc((size_t)w * h * sizeof(float));
    for (int y = 0; y < h; ++y) {
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            const float t = dxr[(size_t)ym * w + x], b = dxr[(size_t)yp * w + x], m = dxr[(size_t)y * w + x];
            float s;
            if (x < w8) { const float u = m + m; const float v = b + u; s = t + v; }
            else { const float u = m * 2.f; const float v = t + u; s = v + b; }
            (*gx)[(size_t)y * w + x] = s;
            (*gy)[(size_t)y * w + x] = smr[(size_t)yp * w + x] - smr[(size_t)ym * w + x];
        }
    }
    free(gray); free(dxr); free(smr);
}

/* seam_finders.cpp:703-778 */
static void compute_costs(const Dp *d, const Img *im1, const Img *im2, int tl1x, int tl1y, int tl2x, int tl2y, int comp,
                          float **cost_v, float **cost_h)
{
    const int l = comp + 1;
    const int rx = d->tls[comp].x, ry = d->tls[comp].y, rw = d->brs[comp].x - rx, rh = d->brs[comp].y - ry;
    const int dx1 = d->utlx - tl1x, dy1 = d->utly - tl1y, dx2 = d->utlx - tl2x, dy2 = d->utly - tl2y;
    const float bad = (255.f * 255.f + 255.f * 255.f) + 255.f * 255.f;
    float *cv = (float *)malloc((size_t)rh * (rw + 1) * sizeof(float));
    float *ch = (float *)malloc((size_t)(rh + 1) * rw * sizeof(float));
    for (int y = ry; y < ry + rh; ++y)
        for (int x = rx; x < rx + rw + 1; ++x) {
            float c = bad;
            /* (the reference reads labels_(y, roi.br().x) one column past the roi here: inside the union image whenever the roi
             * does not touch its right edge; at the edge it reads the next row's first pixel -- guarded, see DESIGN) */
            const int lab = x < d->uw ? L(d, y, x) : (y + 1 < d->uh ? L(d, y + 1, 0) : 0);
            if (lab == l && x > 0 && L(d, y, x - 1) == l) {
                const float cc = (diff3(im1, y + dy1, x + dx1 - 1, im2, y + dy2, x + dx2) +
                                  diff3(im1, y + dy1, x + dx1, im2, y + dy2, x + dx2 - 1)) / 2;
                if (d->cost_func == 0) c = cc;
                else {
                    const float g = fabsf(d->gradx1[(size_t)(y + dy1) * im1->w + x + dx1]) + fabsf(d->gradx1[(size_t)(y + dy1) * im1->w + x + dx1 - 1]) +
                                    fabsf(d->gradx2[(size_t)(y + dy2) * im2->w + x + dx2]) + fabsf(d->gradx2[(size_t)(y + dy2) * im2->w + x + dx2 - 1]) + 1.f;
                    c = cc / g;
                }
            }
            cv[(size_t)(y - ry) * (rw + 1) + (x - rx)] = c;
        }
    for (int y = ry; y < ry + rh + 1; ++y)
        for (int x = rx; x < rx + rw; ++x) {
            float c = bad;
            const int lab = y < d->uh ? L(d, y, x) : 0;
            if (lab == l && y > 0 && L(d, y - 1, x) == l) {
                const float cc = (diff3(im1, y + dy1 - 1, x + dx1, im2, y + dy2, x + dx2) +
                                  diff3(im1, y + dy1, x + dx1, im2, y + dy2 - 1, x + dx2)) / 2;
                if (d->cost_func == 0) c = cc;
                else {
                    const float g = fabsf(d->grady1[(size_t)(y + dy1) * im1->w + x + dx1]) + fabsf(d->grady1[(size_t)(y + dy1 - 1) * im1->w + x + dx1]) +
                                    fabsf(d->grady2[(size_t)(y + dy2) * im2->w + x + dx2]) + fabsf(d->grady2[(size_t)(y + dy2 - 1) * im2->w + x + dx2]) + 1.f;
                    c = cc / g;
                }
            }
            ch[(size_t)(y - ry) * rw + (x - rx)] = c;
        }
    *cost_v = cv; *cost_h = ch;
}

/* seam_finders.cpp:781-909; the seam comes back in union coordinates */
static int estimate_seam(const Dp *d, const Img *im1, const Img *im2, int tl1x, int tl1y, int tl2x, int tl2y, int comp,
                         Pt p1, Pt p2, PtVec *seam, int *is_horizontal)
{
    float *cv, *ch;
    compute_costs(d, im1, im2, tl1x, tl1y, tl2x, tl2y, comp, &cv, &ch);
    const int rx = d->tls[comp].x, ry = d->tls[comp].y, rw = d->brs[comp].x - rx, rh = d->brs[comp].y - ry;
    Pt src = {p1.x - rx, p1.y - ry}, dst = {p2.x - rx, p2.y - ry};
    const int l = comp + 1;
    int swapped = 0;
    *is_horizontal = abs(dst.x - src.x) > abs(dst.y - src.y);
    if (*is_horizontal) { if (src.x > dst.x) { Pt t = src; src = dst; dst = t; swapped = 1; } }
    else if (src.y > dst.y) { Pt t = src; src = dst; dst = t; swapped = 1; }
    uint8_t *control = (uint8_t *)calloc((size_t)rw * rh, 1), *reach = (uint8_t *)calloc((size_t)rw * rh, 1);
    float *cost = (float *)calloc((size_t)rw * rh, sizeof(float));
#define CV(y, x) cv[(size_t)(y) * (rw + 1) + (x)]
#define CH(y, x) ch[(size_t)(y) * rw + (x)]
#define RE(y, x) reach[(size_t)(y) * rw + (x)]
#define CO(y, x) cost[(size_t)(y) * rw + (x)]
    RE(src.y, src.x) = 1;
    CO(src.y, src.x) = 0.f;
    if (*is_horizontal) {
        for (int x = src.x + 1; x <= dst.x; ++x)
            for (int y = 0; y < rh; ++y) {
                if (L(d, y + ry, x + rx) != l) continue;
                float best = 0.f; int step = 0;
                /* min_element over (cost, step) pairs in this order: the first minimum, ties by the smaller step */
                if (RE(y, x - 1)) { best = CO(y, x - 1) + CH(y, x - 1); step = 1; }
                if (y > 0 && RE(y - 1, x - 1)) {
                    const float c = CO(y - 1, x - 1) + CH(y - 1, x - 1) + CV(y - 1, x);
                    if (!step || c < best) { best = c; step = 2; }
                }
                if (y < rh - 1 && RE(y + 1, x - 1)) {
                    const float c = CO(y + 1, x - 1) + CH(y + 1, x - 1) + CV(y, x);
                    if (!step || c < best) { best = c; step = 3; }
                }
                if (step) { CO(y, x) = best; control[(size_t)y * rw + x] = (uint8_t)step; RE(y, x) = 255; }
            }
    } else {
        for (int y = src.y + 1; y <= dst.y; ++y)
            for (int x = 0; x < rw; ++x) {
                if (L(d, y + ry, x + rx) != l) continue;
                float best = 0.f; int step = 0;
                if (RE(y - 1, x)) { best = CO(y - 1, x) + CV(y - 1, x); step = 1; }
                if (x > 0 && RE(y - 1, x - 1)) {
                    const float c = CO(y - 1, x - 1) + CV(y - 1, x - 1) + CH(y, x - 1);
                    if (!step || c < best) { best = c; step = 2; }
                }
                if (x < rw - 1 && RE(y - 1, x + 1)) {
                    const float c = CO(y - 1, x + 1) + CV(y - 1, x + 1) + CH(y, x);
                    if (!step || c < best) { best = c; step = 3; }
                }
                if (step) { CO(y, x) = best; control[(size_t)y * rw + x] = (uint8_t)step; RE(y, x) = 255; }
            }
    }
    int ok = RE(dst.y, dst.x) != 0;
    if (ok) {
        Pt p = dst;
        seam->n = 0;
        pv_push(seam, p.x + rx, p.y + ry);
        if (*is_horizontal) {
            while (p.x != src.x) {
                const int c = control[(size_t)p.y * rw + p.x];
                if (c == 2) p.y--; else if (c == 3) p.y++;
                p.x--;
                pv_push(seam, p.x + rx, p.y + ry);
            }
        } else {
            while (p.y != src.y) {
                const int c = control[(size_t)p.y * rw + p.x];
                if (c == 2) p.x--; else if (c == 3) p.x++;
                p.y--;
                pv_push(seam, p.x + rx, p.y + ry);
            }
        }
        if (!swapped)
            for (int i = 0, j = seam->n - 1; i < j; ++i, --j) { Pt t = seam->p[i]; seam->p[i] = seam->p[j]; seam->p[j] = t; }
    }
#undef CV
#undef CH
#undef RE
#undef CO
    free(cv); free(ch); free(control); free(reach); free(cost);
    return ok;
}

/* seam_finders.cpp:912-1045 */
static void update_labels_using_seam(Dp *d, int comp1, int comp2, const PtVec *seam, int is_horizontal)
{
    const int tx = d->tls[comp1].x, ty = d->tls[comp1].y, mw = d->brs[comp1].x - tx, mh = d->brs[comp1].y - ty;
    const int w = d->uw, h = d->uh;
    int *mask = (int *)calloc((size_t)mw * mh, sizeof(int));
#define M(y, x) mask[(size_t)(y) * mw + (x)]
    const PtVec *ct = &d->contours[comp1];
    for (int i = 0; i < ct->n; ++i) M(ct->p[i].y - ty, ct->p[i].x - tx) = 255;
    for (int i = 0; i < seam->n; ++i) M(seam->p[i].y - ty, seam->p[i].x - tx) = 255;
    const int l1 = comp1 + 1, l2 = comp2 + 1;
    int ncomps = 0;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x)
            if (!M(y, x) && L(d, y + ty, x + tx) == l1) flood_fill_i32(mask, mw, mh, x, y, ++ncomps);
    for (int i = 0; i < ct->n; ++i) {
        const int x = ct->p[i].x - tx, y = ct->p[i].y - ty;
        int ok = 0;
        static const int dx[8] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy[8] = {0, 0, -1, +1, -1, -1, +1, +1};
        for (int j = 0; j < 8; ++j) {
            const int c = x + dx[j], r = y + dy[j];
            if (c >= 0 && c < mw && r >= 0 && r < mh && M(r, c) && M(r, c) != 255) { ok = 1; M(y, x) = M(r, c); }
        }
        if (!ok) M(y, x) = 0;
    }
    for (int i = 0; i < seam->n; ++i) {
        const int x = seam->p[i].x - tx, y = seam->p[i].y - ty;
        if (is_horizontal) {
            if (y < mh - 1 && M(y + 1, x) && M(y + 1, x) != 255) M(y, x) = M(y + 1, x); else M(y, x) = 0;
        } else {
            if (x < mw - 1 && M(y, x + 1) && M(y, x + 1) != 255) M(y, x) = M(y, x + 1); else M(y, x) = 0;
        }
    }
    /* (mask values are 0 .. ncomps by now, or 255 where a contour pixel kept it: ncomps may exceed 255, so count by value) */
    const int nkeys = (ncomps > 255 ? ncomps : 255) + 1;
    int *connect2 = (int *)calloc((size_t)nkeys, sizeof(int)), *connect_other = (int *)calloc((size_t)nkeys, sizeof(int));
    for (int i = 0; i < ct->n; ++i) {
        const int x = ct->p[i].x, y = ct->p[i].y;
        const int key = M(y - ty, x - tx);
        if ((x > 0 && L(d, y, x - 1) == l2) || (y > 0 && L(d, y - 1, x) == l2) ||
            (x < w - 1 && L(d, y, x + 1) == l2) || (y < h - 1 && L(d, y + 1, x) == l2))
            connect2[key]++;
        if ((x > 0 && L(d, y, x - 1) != l1 && L(d, y, x - 1) != l2) || (y > 0 && L(d, y - 1, x) != l1 && L(d, y - 1, x) != l2) ||
            (x < w - 1 && L(d, y, x + 1) != l1 && L(d, y, x + 1) != l2) || (y < h - 1 && L(d, y + 1, x) != l1 && L(d, y + 1, x) != l2))
            connect_other[key]++;
    }
    /* isAdjComp is sized ncomps + 1 in the reference and indexed by every key of connect2 (keys 1 .. ncomps were inserted
     * up front; operator[] adds 0 and, past ncomps, 255): only keys <= ncomps can be read back below */
    uint8_t *is_adj = (uint8_t *)calloc((size_t)nkeys, 1);
    const double len = (double)ct->n;
    for (int k = 0; k <= ncomps; ++k) is_adj[k] = connect2[k] / len > 0.05 && connect_other[k] / len < 0.1;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x)
            if (M(y, x) && M(y, x) <= ncomps && is_adj[M(y, x)]) L(d, y + ty, x + tx) = l2;
#undef M
    free(mask); free(connect2); free(connect_other); free(is_adj);
}

/* seam_finders.cpp:378-516 */
static void resolve_conflicts(Dp *d, const Img *im1, const Img *im2, int tl1x, int tl1y, int tl2x, int tl2y,
                              uint8_t *m1, uint8_t *m2)
{
    const int w = d->uw, h = d->uh, n = d->ncomps;
    if (d->cost_func == 1) { gradients(im1, &d->gradx1, &d->grady1); gradients(im2, &d->gradx2, &d->grady2); }
    for (;;) {
        int c1 = 0, c2 = 0, conflict = 0;
        for (int a = 0; a < n && !conflict; ++a)
            for (int b = 0; b < n; ++b) {
                if (!d->edge[(size_t)a * n + b]) continue;
                if ((d->states[a] & INTERS) && (d->states[a] & ~INTERS) != d->states[b]) { c1 = a; c2 = b; conflict = 1; break; }
            }
        if (!conflict) break;
        const int l1 = c1 + 1, l2 = c2 + 1;
        if (has_only_one_neighbor(d, c1)) {
            for (int y = d->tls[c1].y; y < d->brs[c1].y; ++y)
                for (int x = d->tls[c1].x; x < d->brs[c1].x; ++x)
                    if (L(d, y, x) == l1) L(d, y, x) = l2;
            d->states[c1] = d->states[c2] == FIRST ? SECOND : FIRST;
        } else {
            Pt p1, p2;
            if (get_seam_tips(d, c1, c2, &p1, &p2)) {
                PtVec seam = {0, 0, 0};
                int horiz;
                if (estimate_seam(d, im1, im2, tl1x, tl1y, tl2x, tl2y, c1, p1, p2, &seam, &horiz))
                    update_labels_using_seam(d, c1, c2, &seam, horiz);
                free(seam.p);
            }
            d->states[c1] = d->states[c2] == FIRST ? INTERS_SECOND : INTERS_FIRST;
        }
        const int c[2] = {c1, c2}, lbl[2] = {l1, l2};
        for (int i = 0; i < 2; ++i) {
            const int x0 = d->tls[c[i]].x, x1 = d->brs[c[i]].x, y0 = d->tls[c[i]].y, y1 = d->brs[c[i]].y;
            d->tls[c[i]].x = d->tls[c[i]].y = INT_MAX;
            d->brs[c[i]].x = d->brs[c[i]].y = INT_MIN;
            d->contours[c[i]].n = 0;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    if (L(d, y, x) != lbl[i]) continue;
                    if (x < d->tls[c[i]].x) d->tls[c[i]].x = x;
                    if (y < d->tls[c[i]].y) d->tls[c[i]].y = y;
                    if (x + 1 > d->brs[c[i]].x) d->brs[c[i]].x = x + 1;
                    if (y + 1 > d->brs[c[i]].y) d->brs[c[i]].y = y + 1;
                    if ((x == 0 || L(d, y, x - 1) != lbl[i]) || (x == w - 1 || L(d, y, x + 1) != lbl[i]) ||
                        (y == 0 || L(d, y - 1, x) != lbl[i]) || (y == h - 1 || L(d, y + 1, x) != lbl[i]))
                        pv_push(&d->contours[c[i]], x, y);
                }
        }
        d->edge[(size_t)c1 * n + c2] = 0;
        d->edge[(size_t)c2 * n + c1] = 0;
    }
    const int dx1 = d->utlx - tl1x, dy1 = d->utly - tl1y, dx2 = d->utlx - tl2x, dy2 = d->utly - tl2y;
    for (int y = 0; y < im2->h; ++y)
        for (int x = 0; x < im2->w; ++x) {
            const int l = L(d, y - dy2, x - dx2);
            const int yy = y - dy2 + dy1, xx = x - dx2 + dx1;
            /* mask1.at(yy, xx): inside mask1 wherever the label is FIRST-owned (such pixels lie in image 1) */
            if (l > 0 && (d->states[l - 1] & FIRST) && yy >= 0 && yy < im1->h && xx >= 0 && xx < im1->w && m1[(size_t)yy * im1->w + xx])
                m2[(size_t)y * im2->w + x] = 0;
        }
    for (int y = 0; y < im1->h; ++y)
        for (int x = 0; x < im1->w; ++x) {
            const int l = L(d, y - dy1, x - dx1);
            const int yy = y - dy1 + dy2, xx = x - dx1 + dx2;
            if (l > 0 && (d->states[l - 1] & SECOND) && yy >= 0 && yy < im2->h && xx >= 0 && xx < im2->w && m2[(size_t)yy * im2->w + xx])
                m1[(size_t)y * im1->w + x] = 0;
        }
    if (d->cost_func == 1) { free(d->gradx1); free(d->grady1); free(d->gradx2); free(d->grady2); }
}

/* seam_finders.cpp:192-248 */
static void process_pair(int cost_func, const Img *im1, const Img *im2, int tl1x, int tl1y, int tl2x, int tl2y, uint8_t *m1, uint8_t *m2)
{
    const int itlx = tl1x > tl2x ? tl1x : tl2x, itly = tl1y > tl2y ? tl1y : tl2y;
    const int ibrx = tl1x + im1->w < tl2x + im2->w ? tl1x + im1->w : tl2x + im2->w;
    const int ibry = tl1y + im1->h < tl2y + im2->h ? tl1y + im1->h : tl2y + im2->h;
    if (itlx >= ibrx || itly >= ibry) return;
    Dp d;
    memset(&d, 0, sizeof(d));
    d.cost_func = cost_func;
    d.utlx = tl1x < tl2x ? tl1x : tl2x; d.utly = tl1y < tl2y ? tl1y : tl2y;
    const int ubrx = tl1x + im1->w > tl2x + im2->w ? tl1x + im1->w : tl2x + im2->w;
    const int ubry = tl1y + im1->h > tl2y + im2->h ? tl1y + im1->h : tl2y + im2->h;
    d.uw = ubrx - d.utlx; d.uh = ubry - d.utly;
    const size_t px = (size_t)d.uw * d.uh;
    d.mask1 = (uint8_t *)calloc(px, 1); d.mask2 = (uint8_t *)calloc(px, 1);
    d.contour1 = (uint8_t *)calloc(px, 1); d.contour2 = (uint8_t *)calloc(px, 1);
    d.labels = (int *)malloc(px * sizeof(int));
    for (int y = 0; y < im1->h; ++y) memcpy(d.mask1 + (size_t)(y + tl1y - d.utly) * d.uw + (tl1x - d.utlx), m1 + (size_t)y * im1->w, (size_t)im1->w);
    for (int y = 0; y < im2->h; ++y) memcpy(d.mask2 + (size_t)(y + tl2y - d.utly) * d.uw + (tl2x - d.utlx), m2 + (size_t)y * im2->w, (size_t)im2->w);
    for (int y = 0; y < d.uh; ++y)
        for (int x = 0; x < d.uw; ++x) {
            const uint8_t *mm[2] = {d.mask1, d.mask2};
            uint8_t *cc[2] = {d.contour1, d.contour2};
            for (int k = 0; k < 2; ++k) {
                const uint8_t *m = mm[k];
#define MK(yy, xx) m[(size_t)(yy) * d.uw + (xx)]
                if (MK(y, x) && ((x == 0 || !MK(y, x - 1)) || (x == d.uw - 1 || !MK(y, x + 1)) ||
                                 (y == 0 || !MK(y - 1, x)) || (y == d.uh - 1 || !MK(y + 1, x))))
                    cc[k][(size_t)y * d.uw + x] = 255;
#undef MK
            }
        }
    find_components(&d);
    find_edges(&d);
    resolve_conflicts(&d, im1, im2, tl1x, tl1y, tl2x, tl2y, m1, m2);
    for (int i = 0; i < d.comps_cap; ++i) free(d.contours[i].p);
    free(d.contours); free(d.states); free(d.tls); free(d.brs); free(d.edge);
    free(d.mask1); free(d.mask2); free(d.contour1); free(d.contour2); free(d.labels);
}

/* seam_finders.cpp:163-189.  images[i]: float32 (is_float) or uint8 BGR, sizes_wh[2i] x sizes_wh[2i+1], tight. */
/* computeGradients of one image (test hook): gx, gy are h x w floats */
void mo_dp_gradients(const void *image, int is_float, int w, int h, float *gx, float *gy)
{
    Img im;
    memset(&im, 0, sizeof im);
    im.data = image; im.is_float = is_float; im.w = w; im.h = h;
    float *x = NULL, *y = NULL;
    gradients(&im, &x, &y);
    memcpy(gx, x, (size_t)w * h * sizeof(float));
    memcpy(gy, y, (size_t)w * h * sizeof(float));
    free(x); free(y);
}

void mo_dp_seams(int n, const int *corners_xy, const int *sizes_wh, const void *const *images, int is_float, int cost_func,
                 uint8_t *const *masks)
{
    if (n <= 0) return;
    const int npairs = n * (n - 1) / 2;
    int *pi = (int *)malloc((size_t)(npairs + 1) * sizeof(int)), *pj = (int *)malloc((size_t)(npairs + 1) * sizeof(int));
    long *pd = (long *)malloc((size_t)(npairs + 1) * sizeof(long));
    int k = 0;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int c1x = corners_xy[2 * i] + sizes_wh[2 * i] / 2, c1y = corners_xy[2 * i + 1] + sizes_wh[2 * i + 1] / 2;
            const int c2x = corners_xy[2 * j] + sizes_wh[2 * j] / 2, c2y = corners_xy[2 * j + 1] + sizes_wh[2 * j + 1] / 2;
            pi[k] = i; pj[k] = j;
            pd[k] = (long)(int)((c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y));      /* Point::dot in int */
            ++k;
        }
    /* sort ascending (insertion sort: stable, see the header), then reverse */
    for (int a = 1; a < npairs; ++a) {
        const int ti = pi[a], tj = pj[a]; const long td = pd[a];
        int b = a - 1;
        while (b >= 0 && td < pd[b]) { pi[b + 1] = pi[b]; pj[b + 1] = pj[b]; pd[b + 1] = pd[b]; --b; }
        pi[b + 1] = ti; pj[b + 1] = tj; pd[b + 1] = td;
    }
    for (int a = npairs - 1; a >= 0; --a) {
        const int i0 = pi[a], i1 = pj[a];
        Img im1 = {images[i0], sizes_wh[2 * i0], sizes_wh[2 * i0 + 1], is_float};
        Img im2 = {images[i1], sizes_wh[2 * i1], sizes_wh[2 * i1 + 1], is_float};
        process_pair(cost_func, &im1, &im2, corners_xy[2 * i0], corners_xy[2 * i0 + 1], corners_xy[2 * i1], corners_xy[2 * i1 + 1],
                     masks[i0], masks[i1]);
    }
    free(pi); free(pj); free(pd);
}
