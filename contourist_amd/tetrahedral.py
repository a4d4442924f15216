"""3-D isosurfaces by marching tetrahedra on MI355X -- host-side mirror of the reference's
`contourist/tetrahedral.py` (Delta3DContour :50-87, TriangulatedIsosurfaces :89-101,
Grid3DContour :104-107, GridContour3d :514-621).

Same class names, constructor arguments, methods and return conventions as the reference; the
work happens in hand-written HIP kernels behind the C ABI (include/contourist_hip.h):

    search_for_endpoints()        -> Level-0 march on the device (crossing search, active voxels,
                                     tetrahedra classification, edge interpolation, triangle emit)
    get_points_and_triangles()    -> device post-passes (weld, tiny collapse, clean, orient) and
                                     one download; returns (points, triangles)

Differences a caller can observe (see DESIGN.md "Boundary"):
  * points is a (V,3) float64 ndarray and triangles a (T,3) int32 ndarray (rows sorted); both
    index / iterate like the reference's list of arrays / sorted list of tuples.
  * the march is a dense scan: EVERY component of the isosurface inside the grid is returned,
    not only those reachable from the seed segments.
  * flatten=True (serial LP decimation)
    are not offered on the device path and raise NotImplementedError.
"""
import itertools

import numpy as np

from . import _ffi
from . import grid_field
from . import surface_geometry

# cube corners, Kuhn tetrahedra and the 26-neighbourhood, in the reference's order
# (tetrahedral.py:20-47): corner c = 4*di + 2*dj + dk, tetrahedra = monotone paths A..H.
CUBE = np.array(list(itertools.product((0, 1), repeat=3)), dtype=int)
_A, _B, _C, _D, _E, _F, _G, _H = (tuple(c) for c in CUBE)
TETRAHEDRA = np.array([[_A, _H, _B, _D], [_A, _H, _D, _C], [_A, _H, _C, _G],
                       [_A, _H, _G, _E], [_A, _H, _E, _F], [_A, _H, _F, _B]], dtype=int)
OFFSETS = np.array([o for o in itertools.product((-1, 0, 1), repeat=3) if any(o)], dtype=int)

_DEFAULT_DEVICE = [0]


def set_default_device(device):
    _DEFAULT_DEVICE[0] = int(device)


class GridContour3d(object):
    """Device-backed counterpart of GridContour3d (tetrahedral.py:514-621) working in GRID coordinates.

    corner   = voxels per axis (the reference's `corner`); the sample array has corner+1 per axis.
    samples  = dense fp32 array (numpy) or torch tensor on the GPU, shape corner+1.
    """

    # One extraction addresses its samples and edges with 32-bit ids: 2^29 samples at most (include/contourist_hip.h, cx_grid_upload).
    # A larger volume -- 1024^3 fp32 is 4 GB of the 288 GB of an MI355X -- is marched slab by slab along axis 0 on the same context
    # (each slab with one plane of its upper neighbour, global edge ids, as the ranks of distributed.level1_slabs do) and
    # post-processed as ONE mesh.  Lowered by the tests to force the slab path on small volumes.
    MAX_SAMPLES_PER_EXTRACTION = 1 << 29

    def __init__(self, corner, samples, value, segment_endpoints=None, linear_interpolate=True,
                 callback=None, device=None, diagonal="cpython310", context=None, voxel_range=None, function=None, samples64=None):
        self.corner = np.array(corner, dtype=int)
        assert self.corner.shape == (3,), "dimension must be 3"
        if segment_endpoints is not None:
            for (p1, p2) in segment_endpoints:
                assert len(p1) == 3
                assert len(p2) == 3
        if not linear_interpolate and function is None:
            raise NotImplementedError("linear_interpolate=False re-evaluates the function between the lattice points "
                                      "(tetrahedral.py:488-505): it needs the callable, a sample array is not enough")
        self.dimension = 3
        self.linear_interpolate = bool(linear_interpolate)
        self.function = function      # f(i, j, k) in the lattice coordinates of `samples` (linear_interpolate=False only)
        self._f_broadcasts = None
        self.end_points = segment_endpoints
        self.voxel_range = voxel_range    # in_range box of the seeded growth (lo, hi); None = the whole array
        self.keep_in_range = False        # True: every voxel of the box is kept, the end points only add seed voxels outside it
        self.origin = (0, 0, 0)           # lattice coordinates of sample (0,0,0) in the reference's grid (hash order of the diagonals)
        self.grid_shift = 0               # the sample array starts this many lattice steps before the reference's grid
        self.value = float(value)
        self.callback = callback
        self.flatten = False
        self.smooth = None
        self.samples = samples
        self.samples64 = samples64    # float64 originals of a callable's samples (the reference interpolates on these), or None
        shape = tuple(int(n) for n in samples.shape)
        assert samples64 is None or tuple(samples64.shape) == shape
        assert shape == tuple(int(c) + 1 for c in self.corner), (shape, self.corner)
        self.shape = shape
        self.device = _DEFAULT_DEVICE[0] if device is None else int(device)
        self.flags = {"cpython310": _ffi.CX_DIAG_CPYTHON310, "canonical": _ffi.CX_DIAG_CANONICAL}[diagonal]
        self._ctx = context
        self._counts = None
        self._post = None

    # -- device plumbing ----------------------------------------------------------------------------
    def context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device)
        return self._ctx

    def _bind_grid(self):
        ctx = self.context()
        s = self.samples
        if grid_field._is_torch(s):
            assert s.is_cuda and s.is_contiguous() and str(s.dtype) == "torch.float32", \
                "device samples must be a contiguous float32 tensor on the GPU"
            ctx.adopt_device_grid(s.data_ptr(), self.shape, keepalive=s)
        else:
            ctx.upload_grid(s)
        ctx.shadow_grid_f64(self.samples64)

    def _in_slabs(self):
        "more samples than one extraction addresses: the volume goes through the device slab by slab"
        return int(np.prod(self.shape, dtype=np.int64)) > int(self.MAX_SAMPLES_PER_EXTRACTION)

    def _slab_planes(self):
        "planes of axis 0 per slab, so that a slab plus its halo plane stays within one extraction"
        per_plane = int(self.shape[1]) * int(self.shape[2])
        planes = int(self.MAX_SAMPLES_PER_EXTRACTION) // per_plane - 1
        if planes < 2:
            raise ValueError("a plane of %d x %d samples leaves no room for a slab of two planes and its halo in one extraction "
                             "(%d samples)" % (self.shape[1], self.shape[2], int(self.MAX_SAMPLES_PER_EXTRACTION)))
        return planes

    @staticmethod
    def _slab_bounds(n0, planes):
        """[(i0, i1), ...]: consecutive ranges of planes covering [0, n0), `planes` each (every slab but the last is marched with plane
        i1 as its halo); a single trailing plane joins the slab before it (alone it would hold no voxel; planes + 1 still fit one
        extraction, that slab has no halo)"""
        bounds, at = [], 0
        while at < n0:
            end = min(n0, at + planes)
            if n0 - end == 1:
                end = n0
            bounds.append((at, end))
            at = end
        return bounds

    def _post_in_slabs(self, clean=True):
        """Level 0 slab by slab + Level 1 of the assembled mesh (the single-process form of distributed.level1_slabs: same slabs,
        same global edge ids, same post-pass on ONE mesh, so components, weld buckets and the max-x rule see the whole surface).
        Leaves the Level-1 mesh in the context (download_level1 / level1_torch) and returns the post-pass counts."""
        from . import distributed
        if (self.end_points is not None and len(self.end_points)) or self.voxel_range is not None or not self.linear_interpolate \
                or tuple(self.origin) != (0, 0, 0) or self.grid_shift:
            raise NotImplementedError("a volume of more than %d samples is marched in slabs: exhaustive search, linear interpolation and "
                                      "an array without a rim only (no end points, no voxel range)" % int(self.MAX_SAMPLES_PER_EXTRACTION))
        ctx = self.context()
        s, s64 = self.samples, self.samples64
        on_device = grid_field._is_torch(s)
        if on_device:
            assert s.is_cuda and s.is_contiguous() and str(s.dtype) == "torch.float32", \
                "device samples must be a contiguous float32 tensor on the GPU"
        else:
            s = np.ascontiguousarray(s, dtype=np.float32)
        n0 = int(self.shape[0])
        planes = self._slab_planes()
        parts = []
        totals = dict(n_cells=0, n_vertices=0, n_triangles=0, n_border_voxels=0)
        bounds = self._slab_bounds(n0, planes)
        for (i0, i1) in bounds:
            has_halo = i1 < n0
            local = s[i0:i1 + (1 if has_halo else 0)]
            ctx.set_origin(i0, 0, 0)
            if on_device:
                ctx.adopt_device_grid(local.data_ptr(), tuple(int(n) for n in local.shape), keepalive=s)
            else:
                ctx.upload_grid(local)
            ctx.shadow_grid_f64(None if s64 is None else s64[i0:i1 + (1 if has_halo else 0)])
            counts = ctx.extract3d(self.value, self.flags)
            _xyz32, keys, tris = ctx.download_level0(counts)
            xyz = ctx.level0_points_f64(counts)            # global coordinates (the origin is added before interpolating)
            parts.append(distributed.local_to_global(xyz, keys, tris, tuple(local.shape), i0, i1 - i0, has_halo, True))
            for k in totals:
                totals[k] += int(counts[k])
        ctx.set_origin(0, 0, 0)
        keys, xyz, tris = distributed.assemble(parts)
        self._slab_counts = dict(totals, n_slabs=len(parts), n_vertices=int(len(keys)), n_triangles=int(len(tris)))
        ctx.set_reference_corner((0, 0, 0))
        return ctx.postprocess3d_mesh(xyz, tris, [int(c) for c in self.corner], (0 if clean else 1) | _ffi.CX_MESH_OF_THE_MARCH, self.smooth or 0.0)

    def march(self, force=False):
        "Level 0 on the device (idempotent). returns the counts dict."
        if self._in_slabs():
            raise NotImplementedError("a volume of more than %d samples has no single Level-0 extraction: get_points_and_triangles() "
                                      "marches it in slabs" % int(self.MAX_SAMPLES_PER_EXTRACTION))
        if self._counts is None or force:
            self._bind_grid()
            self.context().set_origin(*self.origin)
            self._counts = self.context().extract3d(self.value, self.flags)
            self._post = None
        return self._counts

    def level0(self):
        """Level-0 mesh as host arrays: dict(xyz (V,3) f32 grid coords, keys (V,) u32 edge ids,
        triangles (T,3) i32 wound low->high, counts)."""
        counts = self.march()
        xyz, keys, tris = self.context().download_level0(counts)
        return dict(xyz=xyz, keys=keys, triangles=tris, counts=counts)

    # -- reference API -------------------------------------------------------------------------------
    def extract_surface_geometry(self, clean=True):
        "SurfaceGeometry of the welded, cleaned and oriented mesh in grid coordinates (tetrahedral.py:604-621)"
        if self.flatten:
            raise NotImplementedError("flatten=True (lp_tools decimation) is outside the device path")
        if self._in_slabs():
            if self.smooth:
                assert self.smooth > 0 and self.smooth <= 1
            ctx = self.context()
            if self._post is None:
                self._post = self._post_in_slabs(clean)
            pts, tris = ctx.download_level1(self._post)
            return surface_geometry.SurfaceGeometry._from_device(pts, tris, ctx)
        self.march()
        if self.flatten:
            raise NotImplementedError("flatten=True (lp_tools decimation) is outside the device path")
        if self.smooth:
            assert self.smooth > 0 and self.smooth <= 1
        ctx = self.context()
        if self._post is None:
            # explicit end points restrict the result to the components the reference's breadth-first search
            # reaches from them (tetrahedral.py:396-463); None = every component (exhaustive search_for_endpoints)
            if self.end_points is not None and len(self.end_points):
                self.seeded = ctx.select_seeded(self.end_points, self.voxel_range, self.keep_in_range)
            if self.voxel_range is not None:   # the array has a margin: Level-1 scales of the reference's own grid
                lo, hi = self.voxel_range
                ctx.set_reference_corner([int(h) - int(l) for l, h in zip(lo, hi)])
            else:
                ctx.set_reference_corner((0, 0, 0))
            if self.linear_interpolate:
                self._post = ctx.postprocess3d(0 if clean else 1, self.smooth or 0.0)
            else:
                self._post = self._postprocess_refined(ctx, clean)
        # an array with a rim (grid_shift, origin = -grid_shift) is post-processed in the reference's own lattice
        # coordinates on the device (weld buckets truncate towards zero there as in the reference): nothing to shift back
        pts, tris = ctx.download_level1(self._post)
        return surface_geometry.SurfaceGeometry._from_device(pts, tris, ctx)

    def _ensure_post(self, clean=True):
        "Level 1 on the device, left there (no download); returns the context"
        if self._in_slabs():
            ctx = self.context()
            if self._post is None:
                self._post = self._post_in_slabs(clean)
            return ctx
        self.march()
        ctx = self.context()
        if self._post is None:
            if self.end_points is not None and len(self.end_points):
                self.seeded = ctx.select_seeded(self.end_points, self.voxel_range, self.keep_in_range)
            if self.voxel_range is not None:
                lo, hi = self.voxel_range
                ctx.set_reference_corner([int(h) - int(l) for l, h in zip(lo, hi)])
            else:
                ctx.set_reference_corner((0, 0, 0))
            self._post = ctx.postprocess3d(0 if clean else 1, self.smooth or 0.0) if self.linear_interpolate else self._postprocess_refined(ctx, clean)
        return ctx

    def write_mesh(self, path, fmt="ply", mins=None, delta=None, clean=True):
        """the welded, cleaned, oriented mesh as a binary file written STRAIGHT FROM THE DEVICE BUFFERS (cx_level1_write: no
        (points, triangles) arrays on the host) -- the step every caller of the reference takes next (html_demo.py:118-161).
        fmt "ply" | "gltf_bin"; mins / delta: world = grid * delta + mins.  Faces in device order (the Python API sorts rows)."""
        ctx = self._ensure_post(clean)
        return ctx.write_level1(path, fmt, mins, delta)

    # -- linear_interpolate=False (tetrahedral.py:488-505) ------------------------------------------------------
    def _feval(self, P):
        "function at the rows of P (N,3), float64: one broadcast call if the function allows it, else one call per row"
        P = np.asarray(P, dtype=np.float64).reshape(-1, 3)
        if len(P) == 0:
            return np.zeros(0)
        if self._f_broadcasts is not False:
            try:
                out = np.asarray(self.function(P[:, 0], P[:, 1], P[:, 2]), dtype=np.float64)
                if out.shape != (len(P),):
                    self._f_broadcasts = False
                elif self._f_broadcasts is None:     # first time: spot-check the broadcast result against scalar calls
                    probe = [0, len(P) // 2, len(P) - 1]
                    self._f_broadcasts = all(abs(out[k] - float(self.function(*P[k]))) <= 1e-12 * max(1.0, abs(out[k])) for k in probe)
                if self._f_broadcasts:
                    return out
            except Exception:
                self._f_broadcasts = False
        return np.array([float(self.function(*row)) for row in P], dtype=np.float64)

    def _refined_points(self, keys):
        """the reference's float64 crossing points with its regula-falsi refinement: contour_pair_interpolation
        (tetrahedral.py:471-512) with linear_interpolate == False, iterations = 5, for every crossing edge at once"""
        lo, hi = unpack_edge_ids(keys, self.shape)
        return refined_crossing_points(self._feval, self.value, lo, hi)

    def _postprocess_refined(self, ctx, clean):
        """Level 1 on refined points: the Level-0 mesh comes to the host, its points are replaced by the reference's
        refined ones (the caller's function is Python), and the same device post-pass runs on the result"""
        L = self.level0()
        keys, tris = L["keys"].astype(np.int64), L["triangles"].astype(np.int64)
        tkeep, vkeep = ctx.seeded_masks(L["counts"])
        tris = tris[tkeep]
        used = np.zeros(len(keys), dtype=bool)
        used[tris.ravel()] = True
        used &= vkeep | used
        ids = np.nonzero(used)[0]
        order = ids[np.argsort(keys[ids], kind="stable")]                    # ascending edge id: index == priority
        renum = -np.ones(len(keys), dtype=np.int64)
        renum[order] = np.arange(len(order))
        pts = self._refined_points(keys[order])
        if self.grid_shift and len(pts):
            pts = pts - float(self.grid_shift)          # the reference's own lattice coordinates (weld buckets, output)
        if self.voxel_range is not None:
            lo, hi = self.voxel_range
            corner = [int(h) - int(l) for l, h in zip(lo, hi)]
        else:
            corner = [int(c) for c in self.corner]
        return ctx.postprocess3d_mesh(pts, renum[tris], corner, (0 if clean else 1) | _ffi.CX_MESH_OF_THE_MARCH, self.smooth or 0.0)

    def get_points_and_triangles(self, clean=True, device=False):
        """(grid_points (V,3) float64, triangles (T,3) int32 sorted rows)  (tetrahedral.py:528-552).
        device=True: the same mesh as torch tensors ON THE GPU (points float64, triangles int32 in device order, each row wound
        as the host rows are) -- no download; for consumers that live on the device (cx_level1_device_ptrs)."""
        if device:
            ctx = self._ensure_post(clean)
            if self.callback:
                self.callback(self)
            return ctx.level1_torch(copy=True)
        geometry = self.extract_surface_geometry(clean)
        if self.callback:
            self.callback(self)
        return (geometry.vertices, geometry.oriented_triangles)

    extract_points_and_triangles = get_points_and_triangles

    # lazily derived views of the reference's bookkeeping (tetrahedral.py:158-169)
    @property
    def interpolated_contour_pairs(self):
        "{((i,j,k) low, (i,j,k) high): grid xyz} as the reference keeps it -- host side, for inspection only"
        L = self.level0()
        lo, hi = unpack_edge_ids(L["keys"], self.shape)
        S = np.asarray(self.samples if not grid_field._is_torch(self.samples) else self.samples.cpu().numpy())
        swap = S[tuple(lo.T)] > S[tuple(hi.T)]
        a = np.where(swap[:, None], hi, lo)
        b = np.where(swap[:, None], lo, hi)
        return {(tuple(int(x) for x in p), tuple(int(x) for x in q)): np.array(c, dtype=float)
                for p, q, c in zip(a, b, L["xyz"])}


def unpack_edge_ids(keys, shape):
    "edge id (lin << 3 | d) -> (lower lattice point (V,3), upper lattice point (V,3))"
    keys = np.asarray(keys).astype(np.int64)
    lin, d = keys >> 3, keys & 7
    i, r = np.divmod(lin, shape[1] * shape[2])
    j, k = np.divmod(r, shape[2])
    lo = np.stack([i, j, k], axis=1)
    hi = lo + np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
    return lo, hi


def bisect_endpoints(lattice_function, value, pairs, lo, hi, dropped=None):
    """The bisection of GridContour.find_initial_voxels (tetrahedral.py:408-423) on the host, for end points that lie OUTSIDE the
    sampled array: the reference evaluates its callable wherever the end points are (its own demos pass (20,20,20) on a 12^3
    grid and (100,100,100) in world coordinates, html_demo.py:147-161, 277-282) and halves the lattice segment until the two
    points are neighbours.  Same swaps, same assert, same integer midpoints, the callable in float64.  A pair inside the box
    [lo, hi] per axis is handed on untouched (the device bisects it on the samples, as before); a pair that leaves the box is
    replaced by the neighbouring pair its bisection ends at.  A pair whose bisection ENDS outside the box (the crossing it brackets
    is not in the sampled array, rim included) is left out and appended to `dropped` if given: the reference would emit the
    triangles of that one voxel out there -- evaluating its callable where this build has no samples -- and grow nothing from it
    (in_range, tetrahedral.py:465-469); the device cannot, and refusing the whole call for it (CX_ERR_INVALID, as round 3 did)
    is worse than leaving the stray voxel out.  -> list of (low_point, high_point) int arrays"""
    out = []
    lo, hi = np.asarray(lo), np.asarray(hi)
    for a, b in pairs:
        a, b = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
        inside = np.all(a >= lo) and np.all(a <= hi) and np.all(b >= lo) and np.all(b <= hi)
        if inside:
            out.append((a, b))
            continue
        low_point, high_point = a, b
        low_value, high_value = float(lattice_function(*low_point)), float(lattice_function(*high_point))
        if low_value > value or high_value < value:
            low_point, low_value, high_point, high_value = high_point, high_value, low_point, low_value
        assert low_value <= value and high_value >= value, "Bad end points " + repr((low_point, low_value, high_point, high_value, value))
        while np.any(np.abs(low_point - high_point) > 1):
            mid_point = (low_point + high_point) // 2
            if float(lattice_function(*mid_point)) < value:
                low_point = mid_point
            else:
                high_point = mid_point
        if np.all(low_point >= lo) and np.all(low_point <= hi) and np.all(high_point >= lo) and np.all(high_point <= hi):
            out.append((low_point, high_point))
        elif dropped is not None:
            dropped.append((low_point, high_point))
    return out


def Grid3DContour(horizontal_n, vertical_m, forward_l, function, value, segment_endpoints,
                  linear_interpolate=True, callback=None, device=None):
    """Grid3DContour(n, m, l, function, value, segment_endpoints, ...)  (tetrahedral.py:104-107).
    `function(i, j, k)` over grid coordinates is sampled once into a dense (n+1, m+1, l+1) array;
    pass a numpy array / GPU tensor of that shape instead of a callable to skip the sampling."""
    corner = (int(horizontal_n), int(vertical_m), int(forward_l))
    if callable(function):
        g = grid_field.FunctionGrid([0, 0, 0], [c - 0.5 for c in corner], [1, 1, 1], function)
        assert tuple(g.grid_dimensions) == corner
        if segment_endpoints is not None and len(segment_endpoints):
            # explicit end points: the reference does not range-check the voxels it starts from and evaluates the function
            # one lattice step outside the grid (tetrahedral.py:396-441): sample that rim too, grow inside the grid only
            m = 1
            segment_endpoints = bisect_endpoints(function, float(value), segment_endpoints, [-m] * 3, [c + m for c in corner])
            shifted = [(np.asarray(a, dtype=int) + m, np.asarray(b, dtype=int) + m) for (a, b) in segment_endpoints]
            maker = GridContour3d(tuple(c + 2 * m for c in corner), g.dense_samples(margin=m), value, shifted, linear_interpolate, callback,
                                  device, voxel_range=((m, m, m), tuple(c + m for c in corner)),
                                  function=lambda i, j, k: function(i - m, j - m, k - m), samples64=g.dense_samples64(margin=m))
            maker.origin = (-m, -m, -m)
            maker.grid_shift = m
            return maker
        samples = g.dense_samples()
    else:
        samples = function
    return GridContour3d(corner, samples, value, segment_endpoints, linear_interpolate, callback, device,
                         function=function if callable(function) else None,
                         samples64=g.dense_samples64() if callable(function) else None)


class Delta3DContour(object):
    """World-coordinate facade (tetrahedral.py:50-87 on top of triangulated.ContourGrid :79-118)."""

    linear_interpolate = True
    flatten = False
    minimum_ratio = None
    minimum_extent = None
    smooth = None
    device = None

    def __init__(self, function_grid, value, segment_endpoints=None, linear_interpolate=True):
        self.linear_interpolate = linear_interpolate
        self.grid = function_grid
        self.value = value
        self.segment_endpoints = segment_endpoints
        grid_endpoints = None
        if segment_endpoints is not None:
            grid_endpoints = []
            for (start_xy, end_xy) in segment_endpoints:
                # the reference asserts len == 2 here (a 2-D leftover, triangulated.py:96) which makes
                # non-empty 3-D endpoints unusable at HEAD; both 2- and 3-vectors are accepted here.
                assert len(start_xy) == len(end_xy) == self.grid.dimension
                grid_endpoint = self.to_grid_endpoint(start_xy, end_xy)
                if grid_endpoint is not None:
                    grid_endpoints.append(grid_endpoint)
            if len(grid_endpoints) < 1:
                grid_endpoints = None
        self.contour_maker = self.get_contour_maker(grid_endpoints)
        self.grid_values = None

    def to_grid_endpoint(self, start_xy, end_xy):
        "first pair of surrounding lattice points whose values straddle the isovalue (triangulated.py:109-118)"
        grid = self.grid
        value = self.value
        for start_grid in grid.surrounding_vertices(np.array(start_xy, dtype=float)):
            for end_grid in grid.surrounding_vertices(np.array(end_xy, dtype=float)):
                if not np.all(start_grid == end_grid):
                    if (grid.grid_function(*start_grid) - value) * (grid.grid_function(*end_grid) - value) <= 0:
                        return (start_grid, end_grid)
        return None

    def get_contour_maker(self, grid_endpoints, rim=True):
        grid = self.grid
        self.grid_endpoints = grid_endpoints
        if self.flatten:
            raise NotImplementedError("flatten=True (lp_tools decimation) is outside the device path")
        gd = np.array([int(n) for n in grid.grid_dimensions])
        self._grid_shift = 0
        if grid_endpoints and rim and not getattr(grid, "array_backed", False):
            # explicit end points on a callable field: the reference does not range-check its seed voxels and
            # evaluates f one lattice step outside the grid (tetrahedral.py:396-441); sample that rim too and
            # keep the breadth-first growth inside the reference's grid
            m = 1
            grid_endpoints = bisect_endpoints(self._lattice_function(0), float(self.value), grid_endpoints, [-m] * 3, list(gd + m))
            shifted = [(np.asarray(a, dtype=int) + m, np.asarray(b, dtype=int) + m) for (a, b) in grid_endpoints]
            result = GridContour3d(tuple(gd + 2 * m), grid.dense_samples(margin=m), self.value, shifted,
                                   linear_interpolate=self.linear_interpolate, device=self.device,
                                   voxel_range=((m, m, m), tuple(gd + m)), function=self._lattice_function(m),
                                   samples64=grid.dense_samples64(margin=m))
            result.origin = (-m, -m, -m)      # the CPython-order diagonals hash the reference's own lattice coordinates
            result.grid_shift = m
            self._grid_shift = m
        else:
            result = GridContour3d(tuple(gd), grid.dense_samples(), self.value,
                                   grid_endpoints, linear_interpolate=self.linear_interpolate, device=self.device,
                                   function=self._lattice_function(0), samples64=grid.dense_samples64())
        result.flatten = self.flatten
        result.smooth = self.smooth
        return result

    def _lattice_function(self, shift):
        """f over the lattice coordinates of the sample array (which may start `shift` steps before the grid): what
        FunctionGrid.grid_function does (grid_field.py:95-118, world = grid * delta + mins), usable on arrays.
        None for a grid made from a sample array."""
        grid = self.grid
        if getattr(grid, "array_backed", False):
            return None
        mins, delta, f = grid.mins, grid.delta, grid.f

        def lattice_f(i, j, k):
            return f((i - shift) * delta[0] + mins[0], (j - shift) * delta[1] + mins[1], (k - shift) * delta[2] + mins[2])
        return lattice_f

    def write_mesh(self, path, fmt="ply"):
        "binary mesh file in WORLD coordinates straight from the device buffers (GridContour3d.write_mesh)"
        if fmt == "gltf":
            from . import mesh_io
            return mesh_io.write_gltf_device(self, path)
        return self.contour_maker.write_mesh(path, fmt, self.grid.mins, self.grid.delta)

    def search_for_endpoints(self, skip=1):
        """Reference: crossing search over every skip-th lattice point + new contour maker (tetrahedral.py:74-81,
        grid_field.py:64-84).  skip == 1 (the canonical call): the device march contains the exhaustive search and
        every component is returned; `grid_endpoints` is derived lazily from the crossing edges it found.
        skip > 1: the coarse search runs on the dense samples, its segments seed the reference's breadth-first
        growth (cx_select_seeded3d), so components the coarse lattice misses stay out, as in the reference."""
        self._skip = skip
        if skip > 1:
            (maxf, minf, segments) = self.grid.find_contour_crossing_grid_segments(self.value, skip)
            self.grid_values = (minf, maxf)
            # coarse segments lie inside the grid: no rim of extra samples needed (rim=False)
            self.contour_maker = self.get_contour_maker(segments if len(segments) else None, rim=False)
            self.contour_maker.march()
            self.grid_endpoints = segments
            return
        rim = None if getattr(self.grid, "array_backed", False) else self._rim_segments()
        if rim is not None and len(rim):
            # the surface reaches the rim of the grid: the reference starts from every crossing segment without
            # range-checking the voxels it starts from (tetrahedral.py:396-441), so the voxels one step OUTSIDE the grid
            # next to the crossing segments on the rim get triangles too.  One extra sample all around, every voxel of
            # the grid kept, the rim segments as seeds
            self.contour_maker = self.get_contour_maker(rim, rim=True)
            self.contour_maker.keep_in_range = True
        else:
            self.contour_maker = self.get_contour_maker(None)
        if not self.contour_maker._in_slabs():      # (a volume beyond one extraction is marched slab by slab in get_points_and_triangles)
            self.contour_maker.march()
        self.grid_endpoints = _LazyEndpoints(self.contour_maker, skip, getattr(self, "_grid_shift", 0), self.grid.grid_dimensions)

    def _rim_segments(self, shell=2):
        """the crossing lattice segments of find_contour_crossing_grid_segments (grid_field.py:64-84: from every lattice
        point 0 <= p < grid_dimensions to its 7 forward neighbours, strict sign change) that lie within `shell` lattice
        steps of the rim of the grid, IN THE REFERENCE'S ORDER (points in index order, last axis fastest; neighbours in
        the order of surrounding_vertices :52-62, first axis fastest).  None if no segment touches the rim itself.
        The order matters: the reference walks its end points one after the other with one shared `visited` set
        (tetrahedral.py:396-441), and which voxel next to the rim a point picks depends on what was visited before."""
        gd = np.array([int(n) for n in self.grid.grid_dimensions])
        S = np.asarray(self.grid.dense_samples_host(), dtype=np.float64)     # vertices 0 .. gd inclusive
        return rim_crossing_segments(S, gd, float(self.value), shell)

    def get_points_and_triangles(self, device=False):
        """(points in world coordinates, triangles)  (tetrahedral.py:83-87).  device=True: torch tensors on the GPU -- the world
        transform grid * delta + mins (grid_field.py:89-93) is applied there too; nothing comes to the host."""
        if device:
            import torch
            (grid_points, triangles) = self.contour_maker.get_points_and_triangles(device=True)
            delta = torch.as_tensor(np.asarray(self.grid.delta, dtype=np.float64), device=grid_points.device)
            mins = torch.as_tensor(np.asarray(self.grid.mins, dtype=np.float64), device=grid_points.device)
            return (grid_points * delta + mins, triangles)
        (grid_points, triangles) = self.contour_maker.get_points_and_triangles()      # (the maker undoes its own shift)
        points = self.grid.from_grid_coordinates(grid_points) if len(grid_points) else np.zeros((0, 3))
        return (points, triangles)


class _LazyEndpoints(object):
    "sequence of (vertex0, vertex1) crossing lattice segments, materialised on first use"

    def __init__(self, maker, skip, shift=0, grid_dimensions=None):
        self._maker, self._skip, self._list = maker, skip, None
        self._shift, self._gd = int(shift), grid_dimensions

    def _get(self):
        if self._list is None:
            L = self._maker.level0()
            lo, hi = unpack_edge_ids(L["keys"], self._maker.shape)
            if self._shift:      # the array carries a rim of extra samples: back to the grid's own lattice
                lo, hi = lo - self._shift, hi - self._shift
                inside = np.all(lo >= 0, axis=1) & np.all(lo < np.asarray(self._gd, dtype=int), axis=1)
                lo, hi = lo[inside], hi[inside]
            if self._skip > 1:
                keep = np.all(lo % self._skip == 0, axis=1)
                lo, hi = lo[keep], hi[keep]
            self._list = list(zip(lo, hi))
        return self._list

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, n):
        return self._get()[n]


class TriangulatedIsosurfaces(Delta3DContour):
    """TriangulatedIsosurfaces(mins, maxes, delta, function, value, segment_endpoints, ...)
    (tetrahedral.py:89-101).  `function` may be a callable f(x, y, z) in world coordinates (sampled
    once, vectorised when it broadcasts) or a dense sample array / GPU tensor of shape
    grid_dimensions+1."""

    def __init__(self, mins, maxes, delta, function, value, segment_endpoints,
                 linear_interpolate=True, flatten=False, minimum_ratio=None, minimum_extent=None,
                 smooth=None, device=None):
        self.flatten = flatten
        self.smooth = smooth
        self.device = device
        if minimum_ratio is not None:
            self.minimum_ratio = minimum_ratio
        if minimum_extent is not None:
            self.minimum_extent = minimum_extent
        if callable(function):
            grid = grid_field.FunctionGrid(mins, maxes, delta, function)
        else:
            grid = grid_field.FunctionGrid.from_array(function, mins, delta)
        Delta3DContour.__init__(self, grid, value, segment_endpoints, linear_interpolate=linear_interpolate)


class MultiLevelIsosurfaces(object):
    """Several isovalues of ONE field (BASELINE.json config 5; the reference has this only in 2-D,
    contourist/multiple_2d_contour.py:17-75).  The dense samples are bound to the device once and ALL levels are
    marched in one call: one pass over the samples classifies every level (`cx_extract3d_levels`), then the
    vertex / triangle stages and the Level-1 post-pass run per level.  `levels()` yields (value, points, triangles)
    in ascending value order; every level equals what a TriangulatedIsosurfaces of that value returns."""

    def __init__(self, mins, maxes, delta, function, values, device=None, diagonal="cpython310"):
        self.values = sorted(float(v) for v in values)
        if callable(function):
            self.grid = grid_field.FunctionGrid(mins, maxes, delta, function)
        else:
            self.grid = grid_field.FunctionGrid.from_array(function, mins, delta)
        self.device = _DEFAULT_DEVICE[0] if device is None else int(device)
        self.flags = {"cpython310": _ffi.CX_DIAG_CPYTHON310, "canonical": _ffi.CX_DIAG_CANONICAL}[diagonal]
        self._ctx = _ffi.Context(self.device)
        self.counts = None

    def levels(self, clean=True):
        samples = self.grid.dense_samples()
        corner = tuple(int(n) for n in self.grid.grid_dimensions)
        shape = tuple(int(n) for n in samples.shape)
        ctx = self._ctx
        if shape[2] < 4:          # rows shorter than 4 samples take the shape-agnostic kernel: one level at a time
            for v in self.values:
                maker = GridContour3d(corner, samples, v, None, context=ctx)
                grid_points, triangles = maker.get_points_and_triangles(clean)
                yield (v, self.grid.from_grid_coordinates(grid_points) if len(grid_points) else np.zeros((0, 3)), triangles)
            return
        if grid_field._is_torch(samples):
            ctx.adopt_device_grid(samples.data_ptr(), shape, keepalive=samples)
        else:
            ctx.upload_grid(samples)
        ctx.set_origin(0, 0, 0)
        ctx.set_reference_corner((0, 0, 0))
        self.counts = ctx.extract3d_levels(self.values, self.flags)
        for n, v in enumerate(self.values):
            ctx.select_level(n)
            post = ctx.postprocess3d(0 if clean else 1)
            grid_points, triangles = ctx.download_level1(post)
            geometry = surface_geometry.SurfaceGeometry._from_device(grid_points, triangles, ctx)   # sorted rows, as the reference returns them
            points = self.grid.from_grid_coordinates(geometry.vertices) if len(grid_points) else np.zeros((0, 3))
            yield (v, points, geometry.oriented_triangles)


def rim_crossing_segments(S, gd, v, shell=2):
    """the crossing lattice segments of find_contour_crossing_grid_segments (grid_field.py:64-84: from every lattice point
    0 <= p < grid_dimensions to its 2^d - 1 forward neighbours, strict sign change) that lie within `shell` lattice steps of
    the rim of the grid, IN THE REFERENCE'S ORDER (points in index order, last axis fastest; neighbours in the order of
    surrounding_vertices :52-62, first axis fastest).  Any dimension.  S: samples at the vertices 0 .. gd inclusive.
    None if no segment touches the rim itself."""
    gd = np.asarray(gd, dtype=int)
    d = len(gd)
    base = S[tuple(slice(0, int(n)) for n in gd)]
    near = np.zeros(tuple(gd), dtype=bool)        # lower end within the shell
    for a in range(d):
        sl = [slice(None)] * d
        sl[a] = slice(0, shell + 1)
        near[tuple(sl)] = True
        sl[a] = slice(max(int(gd[a]) - 1 - shell, 0), int(gd[a]))
        near[tuple(sl)] = True
    rows, touches = [], False
    for index in range(1, 2 ** d):
        o = np.array([(index >> a) & 1 for a in range(d)])          # first axis fastest (:58-61)
        nb = S[tuple(slice(int(o[a]), int(o[a]) + int(gd[a])) for a in range(d))]
        hit = np.argwhere(((base - v) * (nb - v) < 0) & near)
        if len(hit) == 0:
            continue
        upper = hit + o
        touches = touches or bool(np.any(hit == 0) or np.any(upper == gd))
        lin = np.ravel_multi_index(tuple(hit.T), tuple(int(n) for n in gd))
        rows.append(np.concatenate([lin[:, None], np.full((len(hit), 1), index), hit, upper], axis=1))
    if not rows or not touches:
        return None
    R = np.concatenate(rows, axis=0)
    R = R[np.lexsort((R[:, 1], R[:, 0]))]
    return [(r[2:2 + d].astype(int), r[2 + d:2 + 2 * d].astype(int)) for r in R]


def refined_crossing_points(feval, z, lo, hi):
    """contour_pair_interpolation (tetrahedral.py:471-512) with linear_interpolate == False, iterations = 5, for many crossing
    edges at once, any dimension.  feval(P (N,d) float64) -> f at the rows; lo, hi: (N,d) integer lattice points of the edges
    (the coordinates f expects).  -> (N,d) float64 points in those coordinates."""
    low_a, high_a = np.asarray(lo).astype(np.float64), np.asarray(hi).astype(np.float64)
    lo_f, hi_f = low_a.copy(), high_a.copy()
    flow, fhigh = feval(low_a), feval(high_a)
    swap = flow > fhigh                                                  # :478-480
    low_a[swap], high_a[swap] = hi_f[swap], lo_f[swap]
    flow, fhigh = np.where(swap, fhigh, flow), np.where(swap, flow, fhigh)
    crosses = (flow <= z) & (fhigh >= z)
    den = 1.0 * (fhigh - flow)
    flat = np.abs(den) <= 1e-8                                           # np.allclose(denominator, 0)
    ratio = np.where(flat, 0.5, (z - flow) / np.where(flat, 1.0, den))
    P = low_a + ratio[:, None] * (high_a - low_a)
    P[~crosses] = low_a[~crosses]                                        # ":509 temporary hack": interpolated = low_a
    active = crosses.copy()
    fint = np.full(len(P), z, dtype=np.float64)
    if active.any():
        fint[active] = feval(P[active])
    for _ in range(5):                                                   # iterations=5
        close_f = np.abs(fint - z) <= 1e-8 + 1e-5 * abs(z)
        close_p = np.all(np.abs(low_a - high_a) <= 1e-8 + 1e-5 * np.abs(high_a), axis=1)
        active = active & ~close_f & ~close_p
        if not active.any():
            break
        below = active & (fint < z)
        above = active & ~(fint < z)
        low_a[below], flow[below] = P[below], fint[below]
        high_a[above], fhigh[above] = P[above], fint[above]
        r = (z - flow[active]) * 1.0 / (fhigh[active] - flow[active])
        P[active] = low_a[active] + r[:, None] * (high_a[active] - low_a[active])
        fint[active] = feval(P[active])
    return P
