"""4-D isosurfaces ("morphing" 3-D surfaces) by marching pentatopes on MI355X -- host-side mirror of
the reference's `contourist/pentatopes.py` (Delta4DContour :42-68, MorphingIsoSurfaces :71-89,
GridContour4D :92-444).

Device kernels: the hyper-voxel march -- 16-corner border test,
24 pentatopes per hyper-voxel, 1-4 / 2-3 splits into tetrahedra, 4-D edge interpolation (cx_extract4d);
the post-steps of find_tetrahedra -- bin_times, drop_instant_tetrahedra, tiny collapse
(cx_postprocess4d); the slicing of tetrahedra into morph triangles and their time-aware orientation
(cx_morph_triangles).  Row B6, evaluating the surface at a time t, is MorphTriangles.triangles_at
(the consumer-side lerp of misc/morph_triangles.js).
"""
import itertools

import numpy as np

from . import _ffi
from . import grid_field
from . import morph_geometry
from . import tetrahedral


def generate_simplex_vertices(permutation):
    vertex = [0] * len(permutation)
    yield vertex[:]
    for index in permutation:
        vertex[index] = 1
        yield vertex[:]


# the 24 pentatopes (monotone lattice paths, one per axis permutation), the 16 hypercube corners and the
# 80-neighbourhood, in the reference's order (pentatopes.py:15-39)
PENTATOPES = np.array([list(generate_simplex_vertices(p)) for p in itertools.permutations(range(4))], dtype=int)
HYPERCUBE = np.array(list(itertools.product((0, 1), repeat=4)), dtype=int)
OFFSETS4D = np.array([o for o in itertools.product((-1, 0, 1), repeat=4) if any(o)], dtype=int)


def unpack_edge_ids4(keys, shape):
    "edge id (lin << 4 | d) -> (lower lattice point (V,4), upper lattice point (V,4))"
    keys = np.asarray(keys).astype(np.int64)
    lin, d = keys >> 4, keys & 15
    l = lin % shape[3]
    r = lin // shape[3]
    k = r % shape[2]
    r //= shape[2]
    j = r % shape[1]
    i = r // shape[1]
    lo = np.stack([i, j, k, l], axis=1)
    hi = lo + np.stack([(d >> 3) & 1, (d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
    return lo, hi


class GridContour4D(object):
    """device-backed counterpart of GridContour4D (pentatopes.py:92-444), grid coordinates.

    segment_endpoints: lattice point pairs that straddle the value.  None or empty: every component of the 4-D level
    set (what the reference returns after search_for_endpoints()); otherwise find_tetrahedra() keeps the components the
    reference's search reaches from them (80-neighbour growth: tetrahedral.py:396-463 with OFFSETS4D,
    pentatopes.py:32-39; on the device: cx_select_seeded4d).  march() always returns the whole Level-0 mesh."""

    def __init__(self, corner, samples, value, segment_endpoints=None, linear_interpolate=True, callback=None,
                 device=None, diagonal="cpython310", context=None, voxel_range=None, origin=(0, 0, 0, 0), function=None):
        self.corner = np.array(corner, dtype=int)
        assert self.corner.shape == (4,), "dimension should be 4 " + repr(self.corner.shape)
        self.linear_interpolate = linear_interpolate
        self.function = function              # f over the REFERENCE's lattice coordinates (linear_interpolate=False re-evaluates it)
        self._f_broadcasts = None
        self.dimension = 4
        self.value = float(value)
        if callable(samples):
            self.function = samples
            # GridContour4D(corner, function, value, segment_endpoints): the reference's own signature (pentatopes.py:92-100;
            # its test0 demo, :528-551).  f over lattice coordinates is sampled once -- with explicit end points one lattice
            # step beyond the grid as well, where the reference puts seed voxels (tetrahedral.py:396-441)
            gd = tuple(int(c) for c in self.corner)
            g = grid_field.FunctionGrid([0.0] * 4, [c - 0.5 for c in gd], [1.0] * 4, samples)
            assert tuple(int(n) for n in g.grid_dimensions) == gd
            if segment_endpoints is not None and len(segment_endpoints):
                m = 1
                segment_endpoints = [(np.asarray(a, dtype=int) + m, np.asarray(b, dtype=int) + m) for (a, b) in segment_endpoints]
                samples = g.dense_samples(margin=m)
                voxel_range = ((m,) * 4, tuple(n + m for n in gd))
                origin = (-m,) * 4
                self.corner = self.corner + 2 * m
            else:
                samples = g.dense_samples()
        self.end_points = segment_endpoints
        self.samples = samples
        self.shape = tuple(int(n) for n in samples.shape)
        assert self.shape == tuple(int(c) + 1 for c in self.corner), (self.shape, self.corner)
        self.device = tetrahedral._DEFAULT_DEVICE[0] if device is None else int(device)
        self.flags = {"cpython310": _ffi.CX_DIAG_CPYTHON310, "canonical": _ffi.CX_DIAG_CANONICAL}[diagonal]
        self._ctx = context
        self._counts = None
        # an array with a rim of samples around the reference's grid (Delta4DContour.get_contour_maker): the lattice point of
        # sample [0,0,0,0] (negative) and the in_range box of the seeded growth = the reference's grid, in array coordinates
        self.origin = tuple(int(o) for o in origin)
        self.voxel_range = voxel_range
        self.keep_in_range = False        # True: every hyper-voxel of the box is kept, the end points only add seed voxels outside it

    def context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(self.device)
        return self._ctx

    def march(self):
        """Level 0: the hyper-voxel march alone.  returns dict(xyzt (V,4) f32 grid coords, keys (V,) u32 edge ids,
        tetrahedra (T,4) i32, counts)"""
        ctx = self.context()
        ctx.set_origin4d(*self.origin)
        s = self.samples
        if grid_field._is_torch(s):
            assert s.is_cuda and s.is_contiguous() and str(s.dtype) == "torch.float32"
            ctx.adopt_device_grid4d(s.data_ptr(), self.shape, keepalive=s)
        else:
            ctx.upload_grid4d(s)
        self._counts = ctx.extract4d(self.value, self.flags)
        xyzt, keys, tets = ctx.download_level0_4d(self._counts)
        if any(self.origin):        # array coordinates -> the reference's lattice (the edge ids stay those of the array)
            xyzt = xyzt + np.asarray(self.origin, dtype=np.float32)
        return dict(xyzt=xyzt, keys=keys, tetrahedra=tets, counts=self._counts)

    def find_tetrahedra(self, nbins=100):
        """GridContour4D.find_tetrahedra (pentatopes.py:101-125) on the device: march, bin_times(nbins),
        drop_instant_tetrahedra, remove_tiny_simplices(1e-3).  returns dict(points4d (V,4) float64 grid
        coordinates, keys (V,) edge ids, tetrahedra (T,4) int32, counts)"""
        L = self.march()
        ctx = self.context()
        self._interp_vertices = None
        if self.end_points is not None and len(self.end_points):
            self.seeded = ctx.select_seeded4d(self.end_points, self.voxel_range, self.keep_in_range)
            # the reference only ever interpolates the edges of the hyper-voxels its search reaches (interpolated_contour_pairs,
            # tetrahedral.py:176-188): those are the vertices of the tetrahedra the selection keeps.  collect_morph_triangles
            # hands exactly these on as points4d (pentatopes.py:316-322), so that MorphTriangles.to_json -- whose shift, scale
            # and counts come from ALL points it holds (morph_geometry.py:91-125) -- writes what the reference writes.
            keep = ctx.seeded4d_mask(L["counts"]).astype(bool)
            self._interp_vertices = np.unique(L["tetrahedra"][keep].reshape(-1)) if keep.any() else np.zeros(0, dtype=np.int64)
        if self.linear_interpolate:
            post = ctx.postprocess4d(nbins)
        else:
            post = ctx.postprocess4d(nbins, points=self._refined_points(L["keys"]))
        pts, tets = ctx.download_level1_4d(post)
        self.post_counts = post
        return dict(points4d=pts, keys=L["keys"], tetrahedra=tets, counts=post)

    # -- linear_interpolate=False (tetrahedral.py:488-505) ------------------------------------------------------
    def _feval(self, P):
        "function at the rows of P (N,4), float64: one broadcast call if the function allows it, else one call per row"
        P = np.asarray(P, dtype=np.float64).reshape(-1, 4)
        if len(P) == 0:
            return np.zeros(0)
        if self._f_broadcasts is not False:
            try:
                out = np.asarray(self.function(P[:, 0], P[:, 1], P[:, 2], P[:, 3]), dtype=np.float64)
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
        """the reference's float64 crossing points with its regula-falsi refinement (contour_pair_interpolation,
        tetrahedral.py:471-512, linear_interpolate == False), in the reference's lattice, one per Level-0 vertex"""
        if self.function is None:
            raise NotImplementedError("linear_interpolate=False needs the function itself (it is evaluated between the lattice points)")
        lo, hi = unpack_edge_ids4(keys, self.shape)
        shift = np.asarray(self.origin, dtype=np.int64)
        return tetrahedral.refined_crossing_points(self._feval, self.value, lo + shift, hi + shift)

    def collect_morph_triangles(self, epsilon=1e-7):
        """slice the tetrahedra into morph triangles and orient them (pentatopes.py:314-368) on the device.
        Call find_tetrahedra() first (as Delta4DContour.collect_morph_triangles does).
        returns morph_geometry.MorphTriangles in GRID coordinates."""
        assert epsilon == 1e-7, "the device path implements the reference's default epsilon"
        if getattr(self, "post_counts", None) is None:
            self.find_tetrahedra()
        pts, segs, tris, ncomp = self.context().morph_triangles()
        self.n_components = ncomp
        used = getattr(self, "_interp_vertices", None)
        if used is not None:       # explicit end points: only the points the reference's search interpolated (see find_tetrahedra)
            renum = -np.ones(len(pts), dtype=np.int64)
            renum[used] = np.arange(len(used))
            segs = renum[np.asarray(segs, dtype=np.int64)]
            assert segs.min(initial=0) >= 0, "a morph segment uses a vertex outside the seeded selection"
            pts = pts[used]
        self.morph_vertex_ids = np.arange(len(pts)) if used is None else used     # points4d[n] is Level-0 vertex morph_vertex_ids[n]
        return morph_geometry.MorphTriangles(pts, segs, tris)


    def triangles_at(self, t, download=True):
        """surface at grid time t, evaluated on the device from the morph triangles of collect_morph_triangles()
        (misc/morph_triangles.js:26-140; same result as MorphTriangles.triangles_at on the host).
        -> (points (P,3) float64 grid coordinates, triangles (Q,3) int32), or the two counts with download=False"""
        if getattr(self, "n_components", None) is None:
            self.collect_morph_triangles()
        return self.context().morph_eval(t, download)

    def triangles_at_many(self, times, download=True):
        """the surfaces at all `times` in one set of launches (cx_morph_eval_many: the per-t isosurface stream the viewer plays
        frame by frame, misc/morph_triangles.js:117-204) -> list of (points, triangles), each what triangles_at(t) returns;
        with download=False the (n, 2) array of counts (Context.morph_eval_device_ptrs(i) gives the device addresses)"""
        if getattr(self, "n_components", None) is None:
            self.collect_morph_triangles()
        return self.context().morph_eval_many(times, download)


class Delta4DContour(tetrahedral.Delta3DContour):
    "world-coordinate facade (pentatopes.py:42-68)"

    def get_contour_maker(self, grid_endpoints, rim=True):
        grid = self.grid
        self.grid_endpoints = grid_endpoints
        gd = tuple(int(n) for n in grid.grid_dimensions)
        if grid_endpoints is not None and len(grid_endpoints) and rim and not getattr(grid, "array_backed", False):
            # explicit end points on a callable field: the reference does not range-check its seed voxels and evaluates f one
            # lattice step outside the grid (tetrahedral.py:396-441; its own test0 call, pentatopes.py:528-551, starts in two
            # such hyper-voxels): sample that rim too, keep the breadth-first growth inside the reference's grid
            m = 1
            shifted = [(np.asarray(a, dtype=int) + m, np.asarray(b, dtype=int) + m) for (a, b) in grid_endpoints]
            return GridContour4D(tuple(n + 2 * m for n in gd), grid.dense_samples(margin=m), self.value, shifted,
                                 linear_interpolate=self.linear_interpolate, device=self.device,
                                 voxel_range=((m,) * 4, tuple(n + m for n in gd)), origin=(-m,) * 4, function=self._lattice_function())
        return GridContour4D(gd, grid.dense_samples(), self.value, grid_endpoints,
                             linear_interpolate=self.linear_interpolate, device=self.device, function=self._lattice_function())

    def _lattice_function(self):
        "f over the grid's lattice coordinates (grid_field.py:95-118: world = grid * delta + mins); None for a grid made from samples"
        grid = self.grid
        if getattr(grid, "array_backed", False):
            return None
        mins, delta, f = grid.mins, grid.delta, grid.f

        def lattice_f(i, j, k, l):
            return f(i * delta[0] + mins[0], j * delta[1] + mins[1], k * delta[2] + mins[2], l * delta[3] + mins[3])
        return lattice_f

    def search_for_endpoints(self, skip=1):
        """skip == 1: every component (the dense march contains the exhaustive search).  skip > 1: the coarse crossing
        search of the reference (grid_field.py:64-84 over every skip-th lattice point) runs on the dense samples and its
        segments seed the 80-neighbour growth (cx_select_seeded4d): components the coarse lattice misses stay out."""
        if skip > 1:
            (maxf, minf, segments) = self.grid.find_contour_crossing_grid_segments(self.value, skip)
            self.grid_values = (minf, maxf)
            self.contour_maker = self.get_contour_maker(segments if len(segments) else None, rim=False)   # coarse segments lie inside the grid
            self.grid_endpoints = segments
            return
        rim = None
        if not getattr(self.grid, "array_backed", False):
            gd = [int(n) for n in self.grid.grid_dimensions]
            rim = tetrahedral.rim_crossing_segments(np.asarray(self.grid.dense_samples_host(), dtype=np.float64), gd, float(self.value))
        if rim is not None and len(rim):
            # the surface reaches the rim of the grid: the reference starts from every crossing segment without range-checking
            # the hyper-voxels it starts from (tetrahedral.py:396-441), so the ones one step OUTSIDE the grid next to the
            # crossing segments on the rim get tetrahedra too.  One extra sample all around, every hyper-voxel of the grid
            # kept, the rim segments as seeds (as Delta3DContour.search_for_endpoints does in 3-D)
            self.contour_maker = self.get_contour_maker(rim, rim=True)
            self.contour_maker.keep_in_range = True
            return
        self.contour_maker = self.get_contour_maker(None)

    def collect_morph_triangles(self):
        "MorphTriangles in world coordinates (pentatopes.py:64-68)"
        contour_maker = self.contour_maker
        contour_maker.find_tetrahedra()
        grid_morph_triangles = contour_maker.collect_morph_triangles()
        return grid_morph_triangles.from_grid_coordinates(self.grid)

    def triangles_at(self, t):
        """surface at WORLD time t (device evaluation of the morph triangles, misc/morph_triangles.js:26-140):
        (points (P,3) world coordinates, triangles (Q,3) int32)"""
        grid = self.grid
        tg = (float(t) - float(grid.mins[-1])) / float(grid.delta[-1])
        pts, tris = self.contour_maker.triangles_at(tg)
        if len(pts):
            pts = pts * np.asarray(grid.delta[:3], dtype=float) + np.asarray(grid.mins[:3], dtype=float)
        return pts, tris

    def to_json(self):
        "pentatopes.py:85-89"
        morph_triangles = self.collect_morph_triangles()
        return morph_triangles.to_json(min_value=self.grid.mins[-1], max_value=self.grid.maxes[-1])


class MorphingIsoSurfaces(Delta4DContour):
    """MorphingIsoSurfaces(mins, maxes, delta, function, value, segment_endpoints, ...)  (pentatopes.py:71-89).
    linear_interpolate=False is honoured here; the reference's constructor loses the flag (the base constructor it calls last
    resets it to True, pentatopes.py:82-83: there a caller has to set the attribute again before search_for_endpoints)."""

    def __init__(self, mins, maxes, delta, function, value, segment_endpoints, linear_interpolate=True, flatten=False,
                 minimum_ratio=None, minimum_extent=None, smooth=None, device=None):
        self.flatten = flatten
        self.smooth = smooth
        self.device = device
        self.linear_interpolate = linear_interpolate
        if callable(function):
            self.grid = grid_field.FunctionGrid(mins, maxes, delta, function)
        else:
            self.grid = grid_field.FunctionGrid.from_array(function, mins, delta)
        Delta4DContour.__init__(self, self.grid, value, segment_endpoints, linear_interpolate=linear_interpolate)
