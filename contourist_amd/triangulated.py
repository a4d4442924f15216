"""Triangulated 2-D contours -- host-side mirror of the reference's `contourist/triangulated.py`
(ContourGrid :77-118, DxDy2DContourGrid :121-140, DxDy2DContour :143-147, Grid2DContour :149-381,
contour_sequences_to_svg :16-54).

Same constructors, attributes and return types; the lattice walk itself (search_grid, find_initial_contour_pairs,
expand_contour_pairs, get_contour_sequences) runs on the GPU in one pass (csrc/cx_contour2d.hip through
_ffi.Context.contour2d).  There is no CPU fallback: without the HIP library the constructors raise.

Differences from the reference, all about its set-iteration-order artefacts:
  * a sample exactly equal to the isovalue counts as high only; the reference gives such a point both roles and
    resolves the resulting branches in set order,
  * a point is dropped when it is np.allclose to its predecessor on the polyline; the reference compares with the
    last point it kept, in the direction its walk happened to take,
  * polylines come out ordered by their first crossing; the reference's order and starting points follow set order.
"""
import numpy as np

from . import _ffi
from . import grid_field

adjacent_offsets = [(0, 1), (1, 1), (1, 0), (0, -1), (-1, -1), (-1, 0)]   # triangulated.py:10-12
adjacency_array = np.array(adjacent_offsets, dtype=int)

_DEFAULT_DEVICE = [0]


def set_default_device(device):
    _DEFAULT_DEVICE[0] = int(device)


def unpack_keys(keys, m):
    "crossing keys -> (i, j, bi, bj, level): lattice edge (i,j)-(bi,bj) and the index of the isovalue"
    keys = np.asarray(keys, dtype=np.int64)
    level = keys & 0xFFFF
    e = keys >> 16
    lin, d = e // 3, e % 3
    i, j = lin // m, lin % m
    return i, j, i + (d != 1), j + (d != 0), level


def lattice_samples(function, n, m):
    "fp32 samples of function(i, j) over 0 <= i < n, 0 <= j < m: one broadcast call if the function allows it"
    I, J = np.meshgrid(np.arange(n), np.arange(m), indexing="ij")
    try:
        out = np.asarray(function(I, J), dtype=np.float64)
        if out.shape == (n, m) and all(out[p] == float(function(int(p[0]), int(p[1]))) for p in ((0, 0), (n - 1, m - 1), (n // 2, m // 3))):
            return np.ascontiguousarray(out, dtype=np.float32)
    except Exception:
        pass
    out = np.zeros((n, m), dtype=np.float64)
    for i in range(n):
        for j in range(m):
            out[i, j] = function(i, j)
    return np.ascontiguousarray(out, dtype=np.float32)


def seed_rows(f, value, level, low_point, high_point):
    """the two seed rows (i, j, role, level) of one end point pair: bisection of find_initial_contour_pairs
    (triangulated.py:299-315).  f(i, j) is called like the reference calls it."""
    def straddles(lo, hi):
        return f(*lo) <= value and f(*hi) >= value
    low = tuple(int(x) for x in low_point)
    high = tuple(int(x) for x in high_point)
    if not straddles(low, high):
        low, high = high, low
        assert straddles(low, high), "bad end points " + repr((low, high))
    while max(abs(low[0] - high[0]), abs(low[1] - high[1])) > 1:
        mid = ((low[0] + high[0]) // 2, (low[1] + high[1]) // 2)
        if straddles(low, mid):
            high = mid
        else:
            assert straddles(mid, high)
            low = mid
    return [(low[0], low[1], 0, level), (high[0], high[1], 1, level)]


def split_chains(points, keys, chains, nlevels):
    "device output -> per level [(closed, points (k,2), keys (k,)), ...]"
    out = [[] for _ in range(nlevels)]
    for c in chains:
        a, b = int(c["first"]), int(c["first"]) + int(c["count"])
        out[int(c["level"])].append((bool(c["closed"]), points[a:b], keys[a:b]))
    return out


class Grid2DContour(object):
    """Grid2DContour(horizontal_n, vertical_m, function, value, segment_endpoints=None, callback=None)
    (triangulated.py:149-196).  `samples=` gives the lattice values directly (fp32 (n, m) array, or a CUDA/HIP
    torch tensor) instead of sampling `function` once per lattice point."""

    def __init__(self, horizontal_n, vertical_m, function, value, segment_endpoints=None, callback=None, samples=None,
                 device=None, context=None):
        n = self.n = int(horizontal_n)
        m = self.m = int(vertical_m)
        self.corner = np.array([n, m], dtype=int)
        self.z = value
        self.callback = callback
        self.device = _DEFAULT_DEVICE[0] if device is None else int(device)
        self._ctx = context
        if samples is None:
            assert function is not None, "need a function or samples"
            samples = lattice_samples(function, n, m)
        elif not grid_field._is_torch(samples):
            samples = np.ascontiguousarray(samples, dtype=np.float32)
        assert tuple(samples.shape) == (n, m), "samples must have shape (horizontal_n, vertical_m)"
        self.samples = samples
        if function is None:
            host = self._host_samples

            def function(i, j):
                return float(host()[int(i), int(j)])
        self.f = function
        self._host = None
        self.segment_endpoints = None if segment_endpoints is None else np.array(segment_endpoints, dtype=int).reshape(-1, 2, 2)
        self.contours = []
        self._raw = None

    def _host_samples(self):
        if self._host is None:
            s = self.samples
            self._host = s.detach().cpu().numpy() if grid_field._is_torch(s) else s
        return self._host

    def context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device)
        return self._ctx

    def check_callback(self):
        if self.callback:
            self.callback(self)

    def in_range(self, pair):
        return np.all(np.asarray(pair) >= 0) and np.all(np.asarray(pair) < self.corner)

    @property
    def end_points(self):
        """the seeds: the given end points, or (search_grid, triangulated.py:198-212) every crossing axis edge
        that starts at i < n-1, j < m-1, oriented low -> high"""
        if self.segment_endpoints is not None:
            return self.segment_endpoints
        A = self._host_samples().astype(np.float64)
        z = self.z
        out = []
        a = A[:self.n - 1, :self.m - 1]
        for (di, dj) in ((1, 0), (0, 1)):
            b = A[di:self.n - 1 + di, dj:self.m - 1 + dj]
            fwd = (a <= z) & (b >= z)
            bwd = ~fwd & (b <= z) & (a >= z)
            for (i, j) in np.argwhere(fwd):
                out.append(((int(i), int(j)), (int(i) + di, int(j) + dj)))
            for (i, j) in np.argwhere(bwd):
                out.append(((int(i) + di, int(j) + dj), (int(i), int(j))))
        return np.array(out, dtype=int).reshape(-1, 2, 2)

    def _extract(self, values, seeds, flags=0, mins_delta=None):
        s = self.samples
        ctx = self.context()
        if grid_field._is_torch(s):
            assert s.is_cuda and s.is_contiguous() and str(s.dtype) == "torch.float32", \
                "device samples must be a contiguous float32 tensor on the GPU"
            return ctx.contour2d(None, values, seeds, flags, mins_delta, device_ptr=s.data_ptr(), shape=(self.n, self.m))
        return ctx.contour2d(s, values, seeds, flags, mins_delta)

    def _seeds(self):
        if self.segment_endpoints is None or len(self.segment_endpoints) == 0:
            return None
        rows = []
        for (a, b) in self.segment_endpoints:
            rows.extend(seed_rows(self.f, self.z, 0, a, b))
        return np.array(rows, dtype=np.int32)

    def get_contour_sequences(self):
        "[(closed, points (k,2) float64 grid coordinates), ...]  (triangulated.py:226-297)"
        self.check_callback()
        pts, keys, chains, _ = self._extract([float(self.z)], self._seeds())
        self._raw = (pts, keys, chains)
        self.contours = [(closed, p) for (closed, p, _) in split_chains(pts, keys, chains, 1)[0]]
        self.check_callback()
        return self.contours

    @property
    def interpolated_contour_pairs(self):
        "{((i,j) low, (i,j) high): interpolated point} of the polylines kept, before the duplicate drop (triangulated.py:186)"
        pts, keys, _, _ = self._extract([float(self.z)], self._seeds(), _ffi.CX2_NO_DEDUPE)
        i, j, bi, bj, _ = unpack_keys(keys, self.m)
        A = self._host_samples()
        out = {}
        for k in range(len(keys)):
            a, b = (int(i[k]), int(j[k])), (int(bi[k]), int(bj[k]))
            if not A[a] < A[b]:
                a, b = b, a
            out[(a, b)] = pts[k]
        return out

    @property
    def triangle_triples(self):
        "the lattice triangles the polylines pass through, as frozensets of three points (triangulated.py:285-289)"
        if self._raw is None:
            self.get_contour_sequences()
        pts, keys, chains = self._raw
        i, j, bi, bj, _ = unpack_keys(keys, self.m)
        out = set()
        for c in chains:
            a, n = int(c["first"]), int(c["count"])
            idx = list(range(a, a + n)) + ([a] if c["closed"] and n > 2 else [])
            for u, v in zip(idx[:-1], idx[1:]):
                triple = frozenset([(int(i[u]), int(j[u])), (int(bi[u]), int(bj[u])), (int(i[v]), int(j[v])), (int(bi[v]), int(bj[v]))])
                if len(triple) == 3:
                    out.add(triple)
        return out


class ContourGrid(object):
    "what the 2-D and the 3-D world-coordinate facades share (triangulated.py:77-118)"

    def __init__(self, function_grid, value, segment_endpoints=None, linear_interpolate=True):
        self.grid, self.value = function_grid, value
        self.linear_interpolate = linear_interpolate
        self.segment_endpoints = segment_endpoints
        found = None
        if segment_endpoints is not None:
            for (start_xy, end_xy) in segment_endpoints:
                assert len(start_xy) == 2 and len(end_xy) == 2
            found = [ge for ge in (self.to_grid_endpoint(p, q) for (p, q) in segment_endpoints) if ge is not None]
        # no usable end point: the contour maker searches the grid itself (:104-106)
        self.contour_maker = self.get_contour_maker(found if found else None)
        self.grid_values = None

    def to_grid_endpoint(self, start_xy, end_xy):
        """the first pair of distinct lattice points around the two world points whose samples do not lie on the same
        side of the value (:109-118), or None"""
        g, level = self.grid, self.value
        around_start = list(g.surrounding_vertices(np.asarray(start_xy, dtype=float)))
        around_end = list(g.surrounding_vertices(np.asarray(end_xy, dtype=float)))
        for p in around_start:
            fp = g.grid_function(*p) - level
            for q in around_end:
                if np.array_equal(p, q):
                    continue
                if fp * (g.grid_function(*q) - level) <= 0:
                    return (p.copy(), q.copy())
        return None


def grid_lattice(grid):
    """(n, m, samples or None) of a 2-D FunctionGrid: the reference's lattice is 0 <= p < grid_dimensions
    (triangulated.py:125-127); a grid made from a sample array (FunctionGrid.from_array) is contoured whole."""
    assert grid.dimension == 2, "2-D grid expected"
    if getattr(grid, "array_backed", False):
        d = grid.dense_samples()
        return int(d.shape[0]), int(d.shape[1]), d
    n, m = (int(x) for x in grid.grid_dimensions)
    cached = getattr(grid, "_lattice2d", None)
    if cached is None:
        cached = np.ascontiguousarray(grid._evaluate((n, m)), dtype=np.float32)
        grid._lattice2d = cached
    return n, m, cached


class DxDy2DContourGrid(ContourGrid):

    def get_contour_maker(self, grid_endpoints):
        assert self.linear_interpolate, "non-linear interpolation not implemented yet for 2d"
        grid = self.grid
        n, m, samples = grid_lattice(grid)
        return Grid2DContour(n, m, grid.grid_function, self.value, grid_endpoints, samples=samples)

    def get_contour_sequences(self):
        "[(closed, points (k,2) float64 world coordinates), ...]  (triangulated.py:129-138)"
        self.grid_contours = self.contour_maker.get_contour_sequences()
        self.contours = [self.from_grid_contour(c) for c in self.grid_contours]
        return self.contours

    def from_grid_contour(self, contour):
        (closed, grid_points) = contour
        grid = self.grid
        return (closed, np.asarray(grid_points, dtype=float).reshape(-1, 2) * grid.delta + grid.mins)


class DxDy2DContour(DxDy2DContourGrid):

    def __init__(self, xmin, ymin, xmax, ymax, dx, dy, function, value, segment_endpoints=None):
        from . import field2d
        function_grid = field2d.Function2DGrid(xmin, ymin, xmax, ymax, dx, dy, function)
        DxDy2DContourGrid.__init__(self, function_grid, value, segment_endpoints)


def contour_sequences_to_svg(contour_sequences, html_width=300):
    "the contours as one SVG path each, in a viewBox that fits them (triangulated.py:16-54; same text)"
    every = np.concatenate([np.asarray(seq, dtype=float).reshape(-1, 2) for (_, seq) in contour_sequences], axis=0)
    lower, upper = every.min(axis=0), every.max(axis=0)
    (width, height) = upper - lower
    stroke = "%4.2f" % (0.01 * max(width, height),)
    paths = []
    for (closed, seq) in contour_sequences:
        steps = [("L" if k else "M") + "%4.2f %4.2f" % (p[0], p[1]) for k, p in enumerate(np.asarray(seq, dtype=float).reshape(-1, 2))]
        if closed:
            steps.append("Z")
        paths.append('<path stroke-width="%s" stroke="black" fill="none" d="%s" />' % (stroke, " ".join(steps)))
    scale = html_width * (1.0 / width)        # (this order of operations: the digits of the height match the reference's)
    head = '\n<svg height="%s" width="%s" viewBox="%s %s %s %s">\n' % (height * scale, html_width, lower[0], lower[1], width, height)
    return head + "\n".join(paths) + "\n</svg>\n"


def svg_demo():
    import math

    def f(x, y):
        return x * x + y * (y + 1) * (y - 1) - math.sin(2 * y * y + 4 * x)
    C = DxDy2DContour(-1, -1, 1, 1, 0.2, 0.2, f, 0.2)
    return contour_sequences_to_svg(C.get_contour_sequences())
