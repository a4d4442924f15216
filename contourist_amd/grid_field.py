"""Sampled scalar field on a regular n-D grid -- host-side mirror of the reference's
`contourist/grid_field.py` (class FunctionGrid, grid_field.py:8-118; iter_indices :120-137).

Same constructor, attributes and coordinate conventions as the reference, plus what the device
path needs: the field as ONE dense fp32 array (`dense_samples()`), which is what the HIP march
streams instead of calling `function` once per corner per use.

    world = grid * delta + mins                      (grid_field.py:89-93)
    grid_dimensions = int((maxes - mins)/delta) + 1  (grid_field.py:26-27, truncation)
    the march visits voxels 0 <= p < grid_dimensions, i.e. samples 0 .. grid_dimensions inclusive
    (tetrahedral.py:465-469), so the dense array has grid_dimensions + 1 samples per axis.
"""
import numpy as np


def iter_indices(shape, skip=1):
    "generate all index tuples of an array of shape `shape` (stride `skip`), last axis fastest."
    shape = tuple(int(n) for n in shape)
    if len(shape) == 0:
        yield ()
        return
    ranges = [range(0, n, skip) for n in shape]
    idx = [0] * len(shape)

    def rec(axis):
        if axis == len(shape):
            yield tuple(idx)
            return
        for v in ranges[axis]:
            idx[axis] = v
            for t in rec(axis + 1):
                yield t
    for t in rec(0):
        yield t


class FunctionGrid(object):
    """FunctionGrid(mins, maxes, delta, function, materialize=False, cache=False)  (grid_field.py:10)"""

    def __init__(self, mins, maxes, delta, function, materialize=False, cache=False):
        lower = np.asarray(mins, dtype=float).copy()
        assert lower.ndim == 1, "mins is one coordinate per axis"
        self.dimension = int(lower.shape[0])
        # maxes and delta may be scalars or one value per axis (the reference broadcasts them the same way, :11-16)
        self.mins = lower
        self.maxes = np.broadcast_to(np.asarray(maxes, dtype=float), lower.shape).copy()
        self.delta = np.broadcast_to(np.asarray(delta, dtype=float), lower.shape).copy()
        self.f = function
        self.cached, self.cache = cache, {}
        self.materialize, self.materialized_array = materialize, None
        self.grid_dimensions = self.to_grid_vertex(self.maxes) + 1          # int((maxes - mins) / delta) + 1  (:26-27)
        assert np.all(self.grid_dimensions >= 2), "grid must have dimensions greater than 2"
        self._dense = None            # fp32 samples, shape grid_dimensions + 1 (numpy or torch-on-GPU)
        if materialize:
            assert not cache, "do not cache and materialize at the same time."
            self.materialize_array()

    # ---- construction from samples (no Python callable on the hot path) ------------------------
    @classmethod
    def from_array(cls, samples, mins=None, delta=None):
        """Grid over an existing dense sample array (numpy, or a torch tensor already in HBM).
        samples[i,j,k] is the field at grid vertex (i,j,k); grid_dimensions = shape - 1."""
        shape = tuple(int(n) for n in samples.shape)
        dim = len(shape)
        mins = np.zeros(dim) if mins is None else np.array(mins, dtype=float)
        delta = np.ones(dim) if delta is None else np.array(delta, dtype=float) * np.ones(dim)
        # maxes chosen so that int((maxes-mins)/delta)+1 == shape-1 despite rounding
        maxes = mins + delta * (np.array(shape, dtype=float) - 2 + 0.5)
        self = cls.__new__(cls)
        self.mins, self.maxes, self.delta = mins, maxes, delta
        self.dimension = dim
        self.cached = False
        self.materialize = False
        self.materialized_array = None
        self.cache = {}
        self.grid_dimensions = np.array(shape, dtype=int) - 1
        assert np.all(self.grid_dimensions >= 1)
        self._dense = samples if _is_torch(samples) else np.ascontiguousarray(samples, dtype=np.float32)
        self.array_backed = True      # f cannot be evaluated outside the samples
        dense = self._dense

        def lookup(*xyz):
            g = np.rint((np.array(xyz, dtype=float) - mins) / delta).astype(int)
            return float(dense[tuple(int(x) for x in g)])
        self.f = lookup
        return self

    # ---- reference API ---------------------------------------------------------------------------
    def materialize_array(self):
        "dense float64 array of f over index tuples of shape grid_dimensions (grid_field.py:34-44)"
        shape = tuple(int(n) for n in self.grid_dimensions)
        full = self._evaluate(shape)
        self.materialized_array = full
        return full

    def to_grid_coordinates(self, xypoint):
        return (xypoint - self.mins) / self.delta

    def on_grid(self, grid_vertex):
        return np.all(grid_vertex >= 0) and np.all(grid_vertex <= self.grid_dimensions)

    def surrounding_vertices(self, xypoint, skip=1, grid_vertex=False):
        vertex0 = xypoint if grid_vertex else self.to_grid_vertex(xypoint)
        offset = np.zeros((self.dimension,), dtype=int)
        for index in range(2 ** self.dimension):
            for shift in range(self.dimension):
                offset[shift] = ((index >> shift) & 1) * skip
            yield vertex0 + offset

    def to_grid_vertex(self, xypoint):
        return np.array(self.to_grid_coordinates(xypoint), dtype=int)

    def from_grid_coordinates(self, xygrid):
        xygrid = np.array(xygrid, dtype=float)
        return (xygrid * self.delta) + self.mins

    def grid_function(self, *xy_grid):
        "field value at grid coordinates (grid_field.py:95-118)"
        xy_grid = tuple(xy_grid)
        all_ints = all(isinstance(x, (int, np.integer)) for x in xy_grid)
        m = self.materialized_array
        if m is not None and all_ints:
            try:
                return m[xy_grid]
            except IndexError:
                pass
        if self.cached and all_ints and xy_grid in self.cache:
            return self.cache[xy_grid]
        result = self.f(*self.from_grid_coordinates(xy_grid))
        if self.cached and all_ints:
            self.cache[tuple(int(x) for x in xy_grid)] = result
        return result

    def find_contour_crossing_grid_segments(self, value, skip=1):
        """(maxf, minf, [(vertex0, vertex1), ...]) for lattice segments to the 2^d-1 forward neighbours
        with (f0-value)*(f1-value) < 0 (grid_field.py:64-84), in the reference's order: lattice points in
        lexicographic order (iter_indices), their neighbours by offset index (surrounding_vertices: bit `shift` of the index
        moves axis `shift`).  Evaluated with array operations (the device march does not need this list; the
        seeded growth of search_for_endpoints(skip > 1) does, and it is order-sensitive).
        With skip > 1 the forward neighbour of the last strided point lies up to skip-1 steps beyond the grid: the
        reference evaluates f there; so does this for a callable field (a sample array cannot be evaluated outside
        itself: those segments are left out)."""
        gd = tuple(int(n) for n in self.grid_dimensions)
        dim = self.dimension
        ext = skip - 1 if (skip > 1 and not getattr(self, "array_backed", False)) else 0
        if ext:
            S = np.asarray(self._evaluate(tuple(n + 1 + ext for n in gd)), dtype=np.float64)
        else:
            S = np.asarray(self.dense_samples_host(), dtype=np.float64)
        base = tuple(slice(0, gd[a], skip) for a in range(dim))
        f0 = S[base]
        maxf = minf = None
        hits = []
        for index in range(1, 2 ** dim):
            off = [((index >> shift) & 1) * skip for shift in range(dim)]
            sl = tuple(slice(off[a], off[a] + gd[a], skip) for a in range(dim))
            f1 = S[sl]
            # without samples beyond the grid the forward neighbour of the last strided index is missing
            common = tuple(slice(0, min(f0.shape[a], f1.shape[a])) for a in range(dim))
            a0, a1 = f0[common], f1[common]
            if a0.size:
                hi = max(a0.max(), a1.max())
                lo = min(a0.min(), a1.min())
                maxf = hi if maxf is None else max(maxf, hi)
                minf = lo if minf is None else min(minf, lo)
            for idx in np.argwhere((a0 - value) * (a1 - value) < 0):
                rank = int(np.ravel_multi_index(tuple(idx), f0.shape))
                v0 = idx * skip
                hits.append((rank, index, v0.astype(int), (v0 + off).astype(int)))
        hits.sort(key=lambda h: (h[0], h[1]))
        return (maxf, minf, [(h[2], h[3]) for h in hits])

    # ---- dense samples for the device path -----------------------------------------------------
    def _evaluate(self, shape, first=0):
        """f over index tuples first .. first+shape-1 -> float64 array.  Tries one broadcast call
        f(X, Y, Z) first; falls back to one Python call per sample."""
        axes = [self.mins[a] + self.delta[a] * (np.arange(shape[a], dtype=float) + first) for a in range(self.dimension)]
        mesh = np.meshgrid(*axes, indexing="ij")
        try:
            out = np.asarray(self.f(*mesh), dtype=float)
            if out.shape == tuple(shape):
                # spot-check the broadcast result against scalar calls
                probe = [tuple(0 for _ in shape), tuple(n - 1 for n in shape), tuple(n // 2 for n in shape)]
                if all(np.isclose(out[p], float(self.f(*[m[p] for m in mesh])), rtol=1e-12, atol=0) or
                       out[p] == float(self.f(*[m[p] for m in mesh])) for p in probe):
                    return out
        except Exception:
            pass
        out = np.zeros(shape, dtype=float)
        for idx in iter_indices(shape):
            out[idx] = self.f(*[axes[a][idx[a]] for a in range(self.dimension)])
        return out

    def dense_samples(self, margin=0):
        """fp32 samples at grid vertices 0..grid_dimensions inclusive (numpy array or CUDA/HIP tensor).
        margin > 0 (callable f only): vertices -margin .. grid_dimensions+margin, the lattice the reference reaches
        when a seed voxel lies on the rim of the grid (tetrahedral.py:396-441 does not range-check seed voxels)."""
        if margin:
            assert not getattr(self, "array_backed", False), "a sample array cannot be evaluated outside itself"
            shape = tuple(int(n) + 1 + 2 * margin for n in self.grid_dimensions)
            e = self._evaluate(shape, first=-margin)
            self._dense64_margin = (margin, np.ascontiguousarray(e, dtype=np.float64) if e.size <= self.DENSE64_LIMIT else None)
            return np.ascontiguousarray(e, dtype=np.float32)
        if self._dense is None:
            shape = tuple(int(n) + 1 for n in self.grid_dimensions)
            e = self._evaluate(shape)
            self._dense = np.ascontiguousarray(e, dtype=np.float32)
            if e.size <= self.DENSE64_LIMIT:
                self._dense64 = np.ascontiguousarray(e, dtype=np.float64)
        return self._dense

    DENSE64_LIMIT = 1 << 26    # samples; beyond this the float64 originals (8 B each) are not kept

    def dense_samples64(self, margin=0):
        """float64 values of a CALLABLE field at the vertices of dense_samples(margin), or None (array-backed field, or too
        large to keep): what the reference interpolates its crossings on (tetrahedral.py:471-487)"""
        if getattr(self, "array_backed", False):
            return None
        if margin:
            m = getattr(self, "_dense64_margin", None)
            if m is None or m[0] != margin:
                self.dense_samples(margin)
                m = self._dense64_margin
            return m[1]
        self.dense_samples()
        return getattr(self, "_dense64", None)

    def dense_samples_host(self):
        d = self.dense_samples()
        if _is_torch(d):
            return d.detach().cpu().numpy()
        return d


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"
