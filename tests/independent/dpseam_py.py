"""A SECOND, independent restatement of detail::DpSeamFinder -- test infrastructure (round 4).

Written from the reference's own source (`R64/src/seam_finders.cpp:163-1045`, `detail/seam_finders.hpp:105-154`,
`detail/util_inl.hpp:100-122`) in a separate pass WITHOUT consulting `oracle/dpseam.c`: plain Python containers stand in
for the C++ ones whose iteration order is the result --

* `std::sort` of the image pairs + `reverse` (`:175-181`): MSVC 2010 sorts fewer than 32 elements by insertion sort,
  which keeps equal keys in input order; a stable sort is used here (rigs of up to 8 cameras: 28 pairs),
* `cv::floodFill` with default arguments: the 4-connected region of pixels EQUAL to the seed (whatever their label),
* `cv::partition` (core/operations.hpp): connected components of the "closer than 10 px" graph, numbered by their
  first member in input order,
* `std::set<pair<int,int>>` in lexicographic order, `std::map<int,int>` keyed by component number,
* `*min_element` over `pair<float,int>`: smallest cost, ties to the smaller step number, the first such element.

OpenCV 2.4.11 primitives used by computeGradients (`:519-543`) are restated from the published algorithm: BGR2GRAY for
CV_8UC3 in 14-bit fixed point (1868 / 9617 / 4899, + 2^13, >> 14), for CV_32FC3 as b*0.114f + g*0.587f + r*0.299f
evaluated left to right in float; Sobel(ksize 3, CV_32F, BORDER_REFLECT_101) as the separable [-1 0 1] x [1 2 1]
filter, rows first, in float32 -- exact for 8-bit sources; for float sources the three-term sums are associated by column
position as the 2.4.11 filters' SSE bodies, scalar pair / unrolled loops and generic tails do (`_sum121_row`, `_sum121_col`;
round 5 -- before, this file used the scalar pair form everywhere and the oracle the generic tail's).

The reference's quirks are kept: `cy2 = cvRound(sum[j].y / size1)` (`:608`), contour pixels taking the LAST matching
neighbour's component (`:912-934`), flood fills that run over pixels of other labels inside the bounding box and the
relabelling that follows them (`:902-905, :1040-1043`).  computeCosts fills one column of costV and one row of costH
(`:734, :757`) that estimateSeam never reads -- with labels_ read one element past a row's end (`:736, :759`); they
are not computed here, which is why that out-of-bounds read needs no decision.
"""
import numpy as np

FIRST, SECOND, INTERS = 1, 2, 4
INTERS_FIRST, INTERS_SECOND = INTERS | FIRST, INTERS | SECOND
INT_MAX = 2 ** 31 - 1
COLOR, COLOR_GRAD = 0, 1
F = np.float32


def cv_round(v):
    return int(np.rint(np.float64(v)))          # round half to even, like cvtsd2si


def flood_fill(img, x, y, new_val):
    """cv::floodFill(img, Point(x, y), new_val): 4-connectivity, zero tolerance."""
    h, w = img.shape
    old = img[y, x]
    if old == new_val:
        return
    stack = [(x, y)]
    img[y, x] = new_val
    while stack:
        cx, cy = stack.pop()
        for nx, ny in ((cx - 1, cy), (cx + 1, cy), (cx, cy - 1), (cx, cy + 1)):
            if 0 <= nx < w and 0 <= ny < h and img[ny, nx] == old:
                img[ny, nx] = new_val
                stack.append((nx, ny))


def partition_close(points, min_dist):
    """cv::partition with ClosePoints(min_dist): classes numbered by first member."""
    n = len(points)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for i in range(n):
        for j in range(i + 1, n):
            dx, dy = points[i][0] - points[j][0], points[i][1] - points[j][1]
            if dx * dx + dy * dy < min_dist * min_dist:
                ri, rj = find(i), find(j)
                if ri != rj:
                    parent[rj] = ri
    cls, labels = {}, []
    for i in range(n):
        r = find(i)
        if r not in cls:
            cls[r] = len(cls)
        labels.append(cls[r])
    return labels


def bgr2gray(img):
    if img.dtype == np.uint8:
        b, g, r = (img[..., k].astype(np.int64) for k in range(3))
        return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    b, g, r = (img[..., k].astype(F) for k in range(3))
    return ((b * F(0.114) + g * F(0.587)).astype(F) + r * F(0.299)).astype(F)


def _reflect101(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * n - 2 - p
    return p


def _sum121_row(l, c, r, w):
    """[1 2 1] along a row as SymmRowSmallFilter<float, float, SymmRowSmallVec_32f> (2.4.11 filter.cpp) leaves it: the SSE body
    covers columns [0, w & ~7) with l + ((c + c) + r), the scalar pair loop the next even count with (l + c*2) + r, and a last odd
    column goes through the generic loop: kx[0]*S[0] + kx[1]*(S[+1] + S[-1]) = c*2 + (r + l)."""
    w8 = w & ~7
    pairs_end = w8 + ((w - w8) & ~1)
    out = np.empty_like(c)
    out[:, :w8] = (l[:, :w8] + ((c[:, :w8] + c[:, :w8]).astype(F) + r[:, :w8]).astype(F)).astype(F)
    out[:, w8:pairs_end] = ((l[:, w8:pairs_end] + (c[:, w8:pairs_end] * F(2)).astype(F)).astype(F) + r[:, w8:pairs_end]).astype(F)
    out[:, pairs_end:] = ((c[:, pairs_end:] * F(2)).astype(F) + (r[:, pairs_end:] + l[:, pairs_end:]).astype(F)).astype(F)
    return out


def _sum121_col(u, m, d, w):
    """[1 2 1] down a column as SymmColumnSmallFilter<Cast<float, float>, SymmColumnSmallVec_32f> leaves it in the x64 build the
    reference ships (oracle/REFERENCE_BINARY_AUDIT.md): SSE body on columns [0, w & ~7): u + (d + (m + m)); every column after it --
    the loop unrolled by four and what is left -- (u + m*2) + d."""
    w8 = w & ~7
    out = np.empty_like(m)
    out[:, :w8] = (u[:, :w8] + (d[:, :w8] + (m[:, :w8] + m[:, :w8]).astype(F)).astype(F)).astype(F)
    out[:, w8:] = ((u[:, w8:] + (m[:, w8:] * F(2)).astype(F)).astype(F) + d[:, w8:]).astype(F)
    return out


def sobel(gray, dx, dy):
    """cv::Sobel(gray, dst, CV_32F, dx, dy) with ksize 3, (dx, dy) = (1, 0) or (0, 1): rows first, then columns; the difference
    pass is a single subtraction, the smoothing pass associates by column position (see the two helpers)."""
    g = gray.astype(F)
    h, w = g.shape
    xi = [[_reflect101(x + k, w) for x in range(w)] for k in (-1, 0, 1)]
    yi = [[_reflect101(y + k, h) for y in range(h)] for k in (-1, 0, 1)]
    l, c, r = g[:, xi[0]], g[:, xi[1]], g[:, xi[2]]
    row = (r - l).astype(F) if dx == 1 else _sum121_row(l, c, r, w)
    u, m, d = row[yi[0], :], row[yi[1], :], row[yi[2], :]
    return (d - u).astype(F) if dy == 1 else _sum121_col(u, m, d, w)


class _Pair:
    """DpSeamFinder::process for one image pair; masks are modified in place."""

    def __init__(self, cost_func, image1, image2, tl1, tl2, mask1, mask2):
        self.cost_func = cost_func
        self.im1, self.im2, self.tl1, self.tl2, self.m1, self.m2 = image1, image2, tl1, tl2, mask1, mask2

    def run(self):
        im1, im2, tl1, tl2, mask1, mask2 = self.im1, self.im2, self.tl1, self.tl2, self.m1, self.m2
        itl = (max(tl1[0], tl2[0]), max(tl1[1], tl2[1]))
        ibr = (min(tl1[0] + im1.shape[1], tl2[0] + im2.shape[1]), min(tl1[1] + im1.shape[0], tl2[1] + im2.shape[0]))
        if itl[0] >= ibr[0] or itl[1] >= ibr[1]:
            return
        self.utl = (min(tl1[0], tl2[0]), min(tl1[1], tl2[1]))
        ubr = (max(tl1[0] + im1.shape[1], tl2[0] + im2.shape[1]), max(tl1[1] + im1.shape[0], tl2[1] + im2.shape[0]))
        self.uw, self.uh = ubr[0] - self.utl[0], ubr[1] - self.utl[1]
        uw, uh = self.uw, self.uh
        self.M1 = np.zeros((uh, uw), np.uint8)
        self.M2 = np.zeros((uh, uw), np.uint8)
        self.M1[tl1[1] - self.utl[1]:tl1[1] - self.utl[1] + mask1.shape[0], tl1[0] - self.utl[0]:tl1[0] - self.utl[0] + mask1.shape[1]] = mask1
        self.M2[tl2[1] - self.utl[1]:tl2[1] - self.utl[1] + mask2.shape[0], tl2[0] - self.utl[0]:tl2[0] - self.utl[0] + mask2.shape[1]] = mask2
        self.C1, self.C2 = self._contour_mask(self.M1), self._contour_mask(self.M2)
        self._find_components()
        self._find_edges()
        self._resolve_conflicts()

    @staticmethod
    def _contour_mask(m):
        h, w = m.shape
        on = m != 0
        p = np.zeros((h + 2, w + 2), bool)
        p[1:-1, 1:-1] = on
        inner = p[1:-1, :-2] & p[1:-1, 2:] & p[:-2, 1:-1] & p[2:, 1:-1]      # all four neighbours set (outside = unset)
        return np.where(on & ~inner, 255, 0).astype(np.uint8)

    def _is_contour(self, x, y, l):
        L, uw, uh = self.labels, self.uw, self.uh
        return (x == 0 or L[y, x - 1] != l) or (x == uw - 1 or L[y, x + 1] != l) or \
               (y == 0 or L[y - 1, x] != l) or (y == uh - 1 or L[y + 1, x] != l)

    def _find_components(self):
        uw, uh = self.uw, self.uh
        a, b = self.M1 != 0, self.M2 != 0
        L = np.zeros((uh, uw), np.int64)
        L[b] = INT_MAX - 2
        L[a] = INT_MAX - 1
        L[a & b] = INT_MAX
        self.labels = L
        self.states, self.tls, self.brs, self.contours = [], [], [], []
        ncomps = 0
        for y in range(uh):
            for x in range(uw):
                v = L[y, x]
                if v >= INT_MAX - 2:
                    self.states.append(INTERS if v == INT_MAX else FIRST if v == INT_MAX - 1 else SECOND)
                    ncomps += 1
                    flood_fill(L, x, y, ncomps)
                    self.tls.append([x, y])
                    self.brs.append([x + 1, y + 1])
                    self.contours.append([])
                l = int(L[y, x])
                if l:
                    ci = l - 1
                    self.tls[ci][0] = min(self.tls[ci][0], x); self.tls[ci][1] = min(self.tls[ci][1], y)
                    self.brs[ci][0] = max(self.brs[ci][0], x + 1); self.brs[ci][1] = max(self.brs[ci][1], y + 1)
                    if self._is_contour(x, y, l):
                        self.contours[ci].append((x, y))
        self.ncomps = ncomps

    def _neighbours(self, x, y):
        # the order the reference tests them in: left, up, right, down
        if x > 0:
            yield x - 1, y
        if y > 0:
            yield x, y - 1
        if x < self.uw - 1:
            yield x + 1, y
        if y < self.uh - 1:
            yield x, y + 1

    def _find_edges(self):
        L = self.labels
        w = {}
        for ci in range(self.ncomps):
            l = ci + 1
            for (x, y) in self.contours[ci]:
                for nx, ny in self._neighbours(x, y):
                    o = int(L[ny, nx])
                    if o and o != l:
                        w[(ci, o - 1)] = w.get((ci, o - 1), 0) + 1
                        w[(o - 1, ci)] = w.get((o - 1, ci), 0) + 1
        self.edges = {e for e, c in w.items() if c > 0}

    def _touches(self, x, y, pred):
        L = self.labels
        return any(pred(int(L[ny, nx])) for nx, ny in self._neighbours(x, y))

    def _resolve_conflicts(self):
        im1, im2, tl1, tl2, mask1, mask2 = self.im1, self.im2, self.tl1, self.tl2, self.m1, self.m2
        if self.cost_func == COLOR_GRAD:
            g1, g2 = bgr2gray(im1), bgr2gray(im2)
            self.gx1, self.gy1 = sobel(g1, 1, 0), sobel(g1, 0, 1)
            self.gx2, self.gy2 = sobel(g2, 1, 0), sobel(g2, 0, 1)
        L, S = self.labels, self.states
        while True:
            conflict = None
            for (c1, c2) in sorted(self.edges):
                if (S[c1] & INTERS) and (S[c1] & ~INTERS) != S[c2]:
                    conflict = (c1, c2)
                    break
            if conflict is None:
                break
            c1, c2 = conflict
            l1, l2 = c1 + 1, c2 + 1
            if sum(1 for e in self.edges if e[0] == c1) == 1:
                box = L[self.tls[c1][1]:self.brs[c1][1], self.tls[c1][0]:self.brs[c1][0]]
                box[box == l1] = l2
                S[c1] = SECOND if S[c2] == FIRST else FIRST
            else:
                tips = self._seam_tips(c1, c2)
                if tips is not None:
                    est = self._estimate_seam(c1, tips[0], tips[1])
                    if est is not None:
                        self._update_labels(c1, c2, est[0], est[1])
                S[c1] = INTERS_SECOND if S[c2] == FIRST else INTERS_FIRST
            for c, l in ((c1, l1), (c2, l2)):
                x0, x1, y0, y1 = self.tls[c][0], self.brs[c][0], self.tls[c][1], self.brs[c][1]
                self.tls[c] = [INT_MAX, INT_MAX]
                self.brs[c] = [-INT_MAX - 1, -INT_MAX - 1]
                self.contours[c] = []
                for y in range(y0, y1):
                    for x in range(x0, x1):
                        if L[y, x] == l:
                            self.tls[c][0] = min(self.tls[c][0], x); self.tls[c][1] = min(self.tls[c][1], y)
                            self.brs[c][0] = max(self.brs[c][0], x + 1); self.brs[c][1] = max(self.brs[c][1], y + 1)
                            if self._is_contour(x, y, l):
                                self.contours[c].append((x, y))
            self.edges.discard((c1, c2))
            self.edges.discard((c2, c1))
        dx1, dy1 = self.utl[0] - tl1[0], self.utl[1] - tl1[1]
        dx2, dy2 = self.utl[0] - tl2[0], self.utl[1] - tl2[1]
        st = np.array([0] + S, np.int64)
        h2, w2 = mask2.shape
        l2v = L[-dy2:-dy2 + h2, -dx2:-dx2 + w2]
        # mask1 at (y - dy2 + dy1, x - dx2 + dx1): the reference indexes it without a bounds check; inside the union rectangle it is
        # mask1_ (the padded copy) wherever that index falls inside mask1, and only pixels of components that lie inside mask1
        # (state FIRST comes from mask1) can pass the label test
        kill2 = (l2v > 0) & ((st[l2v] & FIRST) != 0) & (self.M1[-dy2:-dy2 + h2, -dx2:-dx2 + w2] != 0)
        h1, w1 = mask1.shape
        l1v = L[-dy1:-dy1 + h1, -dx1:-dx1 + w1]
        m2_after = self.M2.copy()
        m2_after[-dy2:-dy2 + h2, -dx2:-dx2 + w2][kill2] = 0
        mask2[kill2] = 0
        kill1 = (l1v > 0) & ((st[l1v] & SECOND) != 0) & (m2_after[-dy1:-dy1 + h1, -dx1:-dx1 + w1] != 0)
        mask1[kill1] = 0

    def _close(self, y, x, cm):
        return bool(cm[max(0, y - 2):min(self.uh, y + 3), max(0, x - 2):min(self.uw, x + 3)].any())

    def _seam_tips(self, c1, c2):
        l2 = c2 + 1
        special = [(x, y) for (x, y) in self.contours[c1]
                   if self._close(y, x, self.C1) and self._close(y, x, self.C2) and self._touches(x, y, lambda o: o == l2)]
        if len(special) < 2:
            return None
        labels = partition_close(special, 10)
        n = max(labels) + 1
        if n < 2:
            return None
        sx, sy, pts = [0] * n, [0] * n, [[] for _ in range(n)]
        for p, k in zip(special, labels):
            sx[k] += p[0]; sy[k] += p[1]; pts[k].append(p)
        best, idx = -np.finfo(np.float64).max, (-1, -1)
        for i in range(n - 1):
            for j in range(i + 1, n):
                s1, s2 = float(len(pts[i])), float(len(pts[j]))
                cx1, cy1 = cv_round(sx[i] / s1), cv_round(sy[i] / s1)
                cx2, cy2 = cv_round(sx[j] / s2), cv_round(sy[j] / s1)          # (size1: the reference's slip)
                d = float((cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2))
                if d > best:
                    best, idx = d, (i, j)
        out = []
        for k in idx:
            s = float(len(pts[k]))
            cx, cy = cv_round(sx[k] / s), cv_round(sy[k] / s)
            md, pick = np.finfo(np.float64).max, None
            for p in pts[k]:
                d = float((p[0] - cx) * (p[0] - cx) + (p[1] - cy) * (p[1] - cy))
                if d < md:
                    md, pick = d, p
            out.append(pick)
        return out[0], out[1]

    def _diff(self, y1, x1, y2, x2):
        a, b = self.im1[y1, x1], self.im2[y2, x2]
        if self.im1.dtype == np.uint8:
            d = a.astype(np.int64) - b.astype(np.int64)
            return F(int(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        d = (a.astype(F) - b.astype(F)).astype(F)
        q = (d * d).astype(F)
        return F(F(q[0] + q[1]) + q[2])

    def _cost_v(self, y, x, l):
        """costV at union coordinates (y, x), x inside the component's box."""
        L = self.labels
        if not (L[y, x] == l and x > 0 and L[y, x - 1] == l):
            return F(195075.0)
        dx1, dy1 = self.utl[0] - self.tl1[0], self.utl[1] - self.tl1[1]
        dx2, dy2 = self.utl[0] - self.tl2[0], self.utl[1] - self.tl2[1]
        cc = F(F(self._diff(y + dy1, x + dx1 - 1, y + dy2, x + dx2) + self._diff(y + dy1, x + dx1, y + dy2, x + dx2 - 1)) / F(2))
        if self.cost_func == COLOR:
            return cc
        g = F(F(F(F(abs(self.gx1[y + dy1, x + dx1]) + abs(self.gx1[y + dy1, x + dx1 - 1])) + abs(self.gx2[y + dy2, x + dx2])) +
                abs(self.gx2[y + dy2, x + dx2 - 1])) + F(1))
        return F(cc / g)

    def _cost_h(self, y, x, l):
        L = self.labels
        if not (L[y, x] == l and y > 0 and L[y - 1, x] == l):
            return F(195075.0)
        dx1, dy1 = self.utl[0] - self.tl1[0], self.utl[1] - self.tl1[1]
        dx2, dy2 = self.utl[0] - self.tl2[0], self.utl[1] - self.tl2[1]
        cc = F(F(self._diff(y + dy1 - 1, x + dx1, y + dy2, x + dx2) + self._diff(y + dy1, x + dx1, y + dy2 - 1, x + dx2)) / F(2))
        if self.cost_func == COLOR:
            return cc
        g = F(F(F(F(abs(self.gy1[y + dy1, x + dx1]) + abs(self.gy1[y + dy1 - 1, x + dx1])) + abs(self.gy2[y + dy2, x + dx2])) +
                abs(self.gy2[y + dy2 - 1, x + dx2])) + F(1))
        return F(cc / g)

    def _estimate_seam(self, comp, p1, p2):
        L = self.labels
        l = comp + 1
        rx, ry = self.tls[comp]
        rw, rh = self.brs[comp][0] - rx, self.brs[comp][1] - ry
        cv_cache, ch_cache = {}, {}

        def cV(y, x):          # roi coordinates
            k = (y, x)
            if k not in cv_cache:
                cv_cache[k] = self._cost_v(y + ry, x + rx, l)
            return cv_cache[k]

        def cH(y, x):
            k = (y, x)
            if k not in ch_cache:
                ch_cache[k] = self._cost_h(y + ry, x + rx, l)
            return ch_cache[k]
        src, dst = (p1[0] - rx, p1[1] - ry), (p2[0] - rx, p2[1] - ry)
        swapped = False
        horizontal = abs(dst[0] - src[0]) > abs(dst[1] - src[1])
        if horizontal:
            if src[0] > dst[0]:
                src, dst, swapped = dst, src, True
        elif src[1] > dst[1]:
            src, dst, swapped = dst, src, True
        control = np.zeros((rh, rw), np.uint8)
        reach = np.zeros((rh, rw), np.uint8)
        cost = np.zeros((rh, rw), F)
        reach[src[1], src[0]] = 1
        if horizontal:
            for x in range(src[0] + 1, dst[0] + 1):
                for y in range(rh):
                    steps = []
                    if L[y + ry, x + rx] == l:
                        if reach[y, x - 1]:
                            steps.append((F(cost[y, x - 1] + cH(y, x - 1)), 1))
                        if y > 0 and reach[y - 1, x - 1]:
                            steps.append((F(F(cost[y - 1, x - 1] + cH(y - 1, x - 1)) + cV(y - 1, x)), 2))
                        if y < rh - 1 and reach[y + 1, x - 1]:
                            steps.append((F(F(cost[y + 1, x - 1] + cH(y + 1, x - 1)) + cV(y, x)), 3))
                    if steps:
                        c, k = min(steps, key=lambda s: (float(s[0]), s[1]))
                        cost[y, x], control[y, x], reach[y, x] = c, k, 255
        else:
            for y in range(src[1] + 1, dst[1] + 1):
                for x in range(rw):
                    steps = []
                    if L[y + ry, x + rx] == l:
                        if reach[y - 1, x]:
                            steps.append((F(cost[y - 1, x] + cV(y - 1, x)), 1))
                        if x > 0 and reach[y - 1, x - 1]:
                            steps.append((F(F(cost[y - 1, x - 1] + cV(y - 1, x - 1)) + cH(y, x - 1)), 2))
                        if x < rw - 1 and reach[y - 1, x + 1]:
                            steps.append((F(F(cost[y - 1, x + 1] + cV(y - 1, x + 1)) + cH(y, x)), 3))
                    if steps:
                        c, k = min(steps, key=lambda s: (float(s[0]), s[1]))
                        cost[y, x], control[y, x], reach[y, x] = c, k, 255
        if not reach[dst[1], dst[0]]:
            return None
        px, py = dst
        seam = [(px + rx, py + ry)]
        if horizontal:
            while px != src[0]:
                k = control[py, px]
                if k == 2:
                    py -= 1
                elif k == 3:
                    py += 1
                px -= 1
                seam.append((px + rx, py + ry))
        else:
            while py != src[1]:
                k = control[py, px]
                if k == 2:
                    px -= 1
                elif k == 3:
                    px += 1
                py -= 1
                seam.append((px + rx, py + ry))
        if not swapped:
            seam.reverse()
        assert seam[0] == tuple(p1) and seam[-1] == tuple(p2)
        return seam, horizontal

    def _update_labels(self, c1, c2, seam, horizontal):
        L = self.labels
        tx, ty = self.tls[c1]
        mh, mw = self.brs[c1][1] - ty, self.brs[c1][0] - tx
        mask = np.zeros((mh, mw), np.int64)
        for (x, y) in self.contours[c1]:
            mask[y - ty, x - tx] = 255
        for (x, y) in seam:
            mask[y - ty, x - tx] = 255
        l1, l2 = c1 + 1, c2 + 1
        ncomps = 0
        for y in range(mh):
            for x in range(mw):
                if not mask[y, x] and L[y + ty, x + tx] == l1:
                    ncomps += 1
                    flood_fill(mask, x, y, ncomps)
        for (gx, gy) in self.contours[c1]:
            x, y = gx - tx, gy - ty
            ok = False
            for ddx, ddy in ((-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)):
                c, r = x + ddx, y + ddy
                if 0 <= c < mw and 0 <= r < mh and mask[r, c] and mask[r, c] != 255:
                    ok = True
                    mask[y, x] = mask[r, c]
            if not ok:
                mask[y, x] = 0
        for (gx, gy) in seam:
            x, y = gx - tx, gy - ty
            if horizontal:
                mask[y, x] = mask[y + 1, x] if (y < mh - 1 and mask[y + 1, x] and mask[y + 1, x] != 255) else 0
            else:
                mask[y, x] = mask[y, x + 1] if (x < mw - 1 and mask[y, x + 1] and mask[y, x + 1] != 255) else 0
        connect2 = {i: 0 for i in range(1, ncomps + 1)}
        other = {i: 0 for i in range(1, ncomps + 1)}
        for (gx, gy) in self.contours[c1]:
            m = int(mask[gy - ty, gx - tx])
            if self._touches(gx, gy, lambda o: o == l2):
                connect2[m] = connect2.get(m, 0) + 1
            if self._touches(gx, gy, lambda o: o != l1 and o != l2):
                other[m] = other.get(m, 0) + 1
        n = float(len(self.contours[c1]))
        adj = {k: (v / n > 0.05 and other.get(k, 0) / n < 0.1) for k, v in connect2.items() if 1 <= k <= ncomps}
        for y in range(mh):
            for x in range(mw):
                m = int(mask[y, x])
                if m and adj.get(m, False):
                    L[y + ty, x + tx] = l2


def dp_seams(corners, images, masks, cost="color"):
    """DpSeamFinder(cost)::find: returns new masks."""
    out = [np.array(m, np.uint8, copy=True) for m in masks]
    n = len(images)
    if n == 0:
        return out
    cf = {"color": COLOR, "color_grad": COLOR_GRAD}[cost]

    def dist(p):
        i, j = p
        c1 = (corners[i][0] + images[i].shape[1] // 2, corners[i][1] + images[i].shape[0] // 2)
        c2 = (corners[j][0] + images[j].shape[1] // 2, corners[j][1] + images[j].shape[0] // 2)
        return (c1[0] - c2[0]) ** 2 + (c1[1] - c2[1]) ** 2
    pairs = sorted([(i, j) for i in range(n - 1) for j in range(i + 1, n)], key=dist)      # stable, like an insertion sort
    pairs.reverse()
    for (i, j) in pairs:
        _Pair(cf, images[i], images[j], tuple(corners[i]), tuple(corners[j]), out[i], out[j]).run()
    return out
