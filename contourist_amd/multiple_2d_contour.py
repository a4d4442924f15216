"""Several 2-D contour levels of one field -- host-side mirror of the reference's
`contourist/multiple_2d_contour.py` (Multiple2DContourGrid :7-75, Multiple2DContour :78-82,
Percentile2DContour :84-98, Linear2DContour :100-108).

The reference builds one DxDy2DContourGrid per value and walks the lattice once per value; here ALL values are
extracted in ONE pass over the samples on the GPU (csrc/cx_contour2d.hip: the levels each lattice edge crosses
are found by bisection in the sorted values, exactly like classify_endpoint_values :48-59).
"""
import numpy as np

from . import _ffi
from . import field2d
from . import triangulated


class Multiple2DContourGrid(object):

    def __init__(self, function_grid, values, segment_endpoints=()):
        self.grid = function_grid
        self.values = list(sorted(values))
        self.segment_end_points = segment_endpoints
        self.value_to_endpoints = None
        self.value_to_contour_sequences = None
        self._ctx = None

    def context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(triangulated._DEFAULT_DEVICE[0])
        return self._ctx

    def get_contours_dictionary(self):
        """{value: [(closed, points (k,2) world coordinates), ...], ...} for each of the values
        (multiple_2d_contour.py:17-30)."""
        grid = self.grid
        n, m, samples = triangulated.grid_lattice(grid)
        values = sorted(set(float(v) for v in self.values))
        index = {v: k for k, v in enumerate(values)}
        searched = self.classify_endpoints()
        rows = []
        for value in self.value_to_endpoints:
            k = index[float(value)]
            helper = triangulated.ContourGrid.__new__(triangulated.ContourGrid)
            helper.grid, helper.value = grid, value
            for (start_xy, end_xy) in self.value_to_endpoints[value]:
                ge = helper.to_grid_endpoint(start_xy, end_xy)
                if ge is not None:
                    rows.extend(triangulated.seed_rows(grid.grid_function, value, k, ge[0], ge[1]))
        flags = _ffi.CX2_SEARCH_SEEDS if searched else 0
        maker = triangulated.Grid2DContour(n, m, grid.grid_function, values[0], None, samples=samples, context=self.context())
        mins_delta = [grid.mins[0], grid.mins[1], grid.delta[0], grid.delta[1]]
        pts, keys, chains, _ = maker._extract(values, np.array(rows, dtype=np.int32).reshape(-1, 4) if rows else None, flags, mins_delta)
        per_level = triangulated.split_chains(pts, keys, chains, len(values))
        self.value_to_contour_sequences = {value: [(c, p) for (c, p, _) in per_level[index[float(value)]]] for value in self.value_to_endpoints}
        return self.value_to_contour_sequences

    def classify_endpoints(self):
        """value -> the given segments that straddle it (multiple_2d_contour.py:32-42).  Returns True when some value
        has none, i.e. when the reference falls back to its exhaustive grid search (:39-41): the crossings that
        search finds are generated on the device and are not listed here."""
        self.value_to_endpoints = dict((level, []) for level in self.values)
        field = self.grid.f
        for (p, q) in self.segment_end_points:
            self.classify_endpoint_values(p, field(*p), q, field(*q))
        return min((len(found) for found in self.value_to_endpoints.values()), default=1) == 0

    def classify_endpoint(self, startpoint, endpoint):
        "one segment: evaluate the field at both ends and file it under the levels in between (:44-46)"
        field = self.grid.f
        return self.classify_endpoint_values(startpoint, field(*startpoint), endpoint, field(*endpoint))

    def classify_endpoint_values(self, startpoint, f_start, endpoint, f_end):
        """file the segment, oriented from its lower to its higher end, under every level v with
        f(lower) <= v <= f(higher) (:48-59): a contiguous index range of the sorted levels"""
        lower, upper = (startpoint, endpoint) if not f_end < f_start else (endpoint, startpoint)
        levels = np.asarray(self.values, dtype=np.float64)
        first = int(np.searchsorted(levels, min(f_start, f_end), side="left"))
        last = int(np.searchsorted(levels, max(f_start, f_end), side="right"))
        for level in self.values[first:last]:
            self.value_to_endpoints[level].append((lower, upper))


def _grid2d(xmin, ymin, xmax, ymax, dx, dy, function):
    return field2d.Function2DGrid(xmin, ymin, xmax, ymax, dx, dy, function)


class Multiple2DContour(Multiple2DContourGrid):
    "levels given by the caller (multiple_2d_contour.py:78-82)"

    def __init__(self, xmin, ymin, xmax, ymax, dx, dy, function, values, segment_endpoints=()):
        Multiple2DContourGrid.__init__(self, _grid2d(xmin, ymin, xmax, ymax, dx, dy, function), values, segment_endpoints)


class Percentile2DContour(Multiple2DContourGrid):
    "levels = every (N / breakpoints)-th sample of the sorted lattice samples (multiple_2d_contour.py:84-98)"

    def __init__(self, xmin, ymin, xmax, ymax, dx, dy, function, breakpoints=10, segment_endpoints=()):
        self.function_grid = _grid2d(xmin, ymin, xmax, ymax, dx, dy, function)
        self.values = self.get_values(breakpoints)
        Multiple2DContourGrid.__init__(self, self.function_grid, self.values, segment_endpoints)

    def _samples(self):
        "the lattice samples the device contours (float64 view of the fp32 array)"
        return np.asarray(triangulated.grid_lattice(self.function_grid)[2], dtype=np.float64)

    def get_values(self, breakpoints):
        ordered = np.sort(self._samples(), axis=None)
        stride = int(ordered.size / breakpoints)
        return list(ordered[stride::stride])


class Linear2DContour(Percentile2DContour):
    "levels = multiples of (max - min) / breakpoints, as the reference computes them (multiple_2d_contour.py:100-108)"

    def get_values(self, breakpoints):
        samples = self._samples()
        step = (samples.max() - samples.min()) * (1.0 / breakpoints)
        return [step * k for k in range(1, breakpoints)]
