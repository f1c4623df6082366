"""Polygon rasterisers the reference's dataset classes call in third-party libraries, restated from the published
algorithms of the pinned versions (requirement.yaml: scikit-image==0.18.2, opencv-python==4.5.3.56; neither is installed
here, so parity with the libraries themselves is UNPINNED -- tests check the algorithms' defining properties):

  * ``polygon(r, c, shape)``  = ``skimage.draw.polygon``  (datasets/echo.py:243-246: EchoNet volume tracings -> LV mask)
  * ``fill_poly(points, shape)`` = ``cv2.fillPoly(img, [points], 255)`` on a zero image (datasets/cardiac_uda.py:223-246:
    CardiacUDA contour pixels -> filled organ mask)

Host-side, numpy only: they run once per sample in front of the GPU input-formatting kernels.
"""
import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def polygon(r, c, shape=None):
    """Pixels whose centres lie inside the polygon with vertex rows `r` / columns `c`: scikit-image 0.18's
    ``_polygon`` -- bounding box from the vertices (clipped to `shape`), then the crossing-number test
    ``point_in_polygon`` (skimage/_shared/geometry.pxd) per pixel: an edge (i, j) toggles the state when
    ``(yp[i] <= y < yp[j] or yp[j] <= y < yp[i]) and x < (xp[j] - xp[i]) * (y - yp[i]) / (yp[j] - yp[i]) + xp[i]``.
    Returns (rr, cc) in row-major order, like the library."""
    r = np.asarray(r, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    if r.size == 0:
        return np.zeros(0, np.intp), np.zeros(0, np.intp)
    minr, maxr = int(max(0, r.min())), int(np.ceil(r.max()))
    minc, maxc = int(max(0, c.min())), int(np.ceil(c.max()))
    if shape is not None:
        maxr, maxc = min(shape[0] - 1, maxr), min(shape[1] - 1, maxc)
    if maxr < minr or maxc < minc:
        return np.zeros(0, np.intp), np.zeros(0, np.intp)
    ys = np.arange(minr, maxr + 1, dtype=np.float64)[:, None]
    xs = np.arange(minc, maxc + 1, dtype=np.float64)[None, :]
    inside = np.zeros((ys.shape[0], xs.shape[1]), dtype=bool)
    j = r.size - 1
    for i in range(r.size):
        yi, yj, xi, xj = r[i], r[j], c[i], c[j]
        if yi != yj:
            span = ((yi <= ys) & (ys < yj)) | ((yj <= ys) & (ys < yi))
            xcross = (xj - xi) * (ys - yi) / (yj - yi) + xi
            inside ^= span & (xs < xcross)
        j = i
    rr, cc = np.nonzero(inside)
    return rr + minr, cc + minc


def _line8(img, x0, y0, x1, y1, value):
    """cv::LineIterator, 8-connected (imgproc/drawing.cpp): the major axis advances every step, the minor axis when the
    error term is negative; both end points are drawn.  Points outside the image are skipped (cv::clipLine)."""
    h, w = img.shape
    dx, dy = x1 - x0, y1 - y0
    sx, sy = (-1 if dx < 0 else 1), (-1 if dy < 0 else 1)
    dx, dy = abs(dx), abs(dy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err = dx - 2 * dy
    x, y = x0, y0
    for _ in range(dx + 1):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = value
        minor = err < 0
        err += -2 * dy + (2 * dx if minor else 0)
        if steep:
            y += sy
            if minor:
                x += sx
        else:
            x += sx
            if minor:
                y += sy


def fill_poly(points, shape, value=255, out=None):
    """``cv2.fillPoly(img, [points], value)`` (lineType LINE_8, shift 0) on a (H, W) uint8 image: `points` is (n, 2) integer
    (x, y).  OpenCV's CollectPolyEdges draws every polygon edge as an 8-connected line and FillEdgeCollection fills the
    interior by the even-odd rule on scanlines y0 <= y < y1 of each non-horizontal edge, edge abscissae in 16.16 fixed
    point advanced by the truncated slope ``dx = (x1 - x0) / (y1 - y0)``, spans ``[ceil(xa), floor(xb)]`` between the
    1st/2nd, 3rd/4th, ... active edges (sorted by x, bubble-sorted after every scanline)."""
    H, W = shape
    img = np.zeros((H, W), dtype=np.uint8) if out is None else out
    pts = np.asarray(points, dtype=np.int64).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return img
    edges = []                                    # [y0, y1, x (16.16), dx]
    px, py = int(pts[-1, 0]) << XY_SHIFT, int(pts[-1, 1])
    for i in range(n):
        qx, qy = int(pts[i, 0]) << XY_SHIFT, int(pts[i, 1])
        _line8(img, (px + (XY_ONE >> 1)) >> XY_SHIFT, py, (qx + (XY_ONE >> 1)) >> XY_SHIFT, qy, value)
        if py != qy:
            num, den = qx - px, qy - py
            dx = abs(num) // abs(den) * (1 if (num >= 0) == (den >= 0) else -1)     # C++ integer division truncates
            edges.append([py, qy, px, dx] if py < qy else [qy, py, qx, dx])
        px, py = qx, qy
    if len(edges) < 2:
        return img
    edges.sort(key=lambda e: (e[0], e[2], e[3]))  # CmpEdges: y0, then x, then dx
    y_max = min(max(e[1] for e in edges), H)
    active, nxt, total = [], 0, len(edges)
    for y in range(edges[0][0], y_max):
        active = [e for e in active if e[1] != y]                 # an edge leaves when y reaches its lower end
        while nxt < total and edges[nxt][0] == y:                 # an edge enters in front of the first one with x >= its x
            e = edges[nxt]
            k = 0
            while k < len(active) and active[k][2] < e[2]:
                k += 1
            active.insert(k, e)
            nxt += 1
        for k in range(0, len(active) - 1, 2):
            a, b = active[k], active[k + 1]
            lo, hi = (b, a) if a[2] > b[2] else (a, b)
            x1, x2 = (lo[2] + XY_ONE - 1) >> XY_SHIFT, hi[2] >> XY_SHIFT
            if y >= 0 and x1 < W and x2 >= 0:
                img[y, max(x1, 0):min(x2, W - 1) + 1] = value
            a[2] += a[3]
            b[2] += b[3]
        active.sort(key=lambda e: e[2])                           # stable, like the bubble sort (swaps only when x is larger)
    return img
