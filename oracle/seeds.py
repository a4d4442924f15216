"""TEST INFRASTRUCTURE (oracle) -- seeded voxel selection of the 3-D and the 4-D march, restated from the reference
(the 4-D march reuses the same methods with the 80 offsets of pentatopes.py:32-39).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Reference (contourist/tetrahedral.py):
  OFFSETS               :41-47    the 26 neighbour offsets in (i, j, k) lexicographic order
  border_voxel          :383-394  not np.allclose(value, corners) and min <= value <= max
  find_initial_voxels   :396-441  bisection of each end point pair, the point's own voxel or its first border
                                  neighbour, one shared `visited` set
  expand_voxels         :443-463  breadth-first growth over in-range border voxels
  in_range              :465-469  0 <= voxel < corner
The reference evaluates f outside the array for voxels on its rim; a dense array cannot: such voxels are not
border voxels here (the same rule as the device code)."""
import itertools

import numpy as np

OFFSETS = [o for o in itertools.product((-1, 0, 1), repeat=3) if o != (0, 0, 0)]
BOX = list(itertools.product((0, 1), repeat=3))
_OFFS = {3: OFFSETS, 4: [o for o in itertools.product((-1, 0, 1), repeat=4) if any(o)]}
_BOX = {3: BOX, 4: list(itertools.product((0, 1), repeat=4))}


def border_voxel(A, value, p):
    p = tuple(int(x) for x in p)
    dim = A.ndim
    if any(x < 0 for x in p) or any(p[a] + 1 >= A.shape[a] for a in range(dim)):
        return False
    vals = np.array([float(A[tuple(p[a] + b[a] for a in range(dim))]) for b in _BOX[dim]], dtype=np.float64)
    if np.allclose(value, vals):
        return False
    return vals.min() <= value and vals.max() >= value


def initial_voxels(A, value, end_points):
    visited, new = set(), set()
    dim = A.ndim
    for (low_point, high_point) in np.asarray(end_points, dtype=np.int64).reshape(-1, 2, dim):
        low_point, high_point = low_point.copy(), high_point.copy()
        low_value, high_value = float(A[tuple(low_point)]), float(A[tuple(high_point)])
        if low_value > value or high_value < value:
            low_point, low_value, high_point, high_value = high_point, high_value, low_point, low_value
        assert low_value <= value and high_value >= value, "Bad end points"
        while np.any(np.abs(low_point - high_point) > 1):
            mid = (low_point + high_point) // 2
            if float(A[tuple(mid)]) < value:
                low_point = mid
            else:
                high_point = mid
        for point in (low_point, high_point):
            t = tuple(int(x) for x in point)
            if t in visited:
                continue
            visited.add(t)
            if border_voxel(A, value, t):
                new.add(t)
                continue
            for o in _OFFS[dim]:
                q = tuple(t[a] + o[a] for a in range(dim))
                if q in visited:
                    continue
                visited.add(q)
                if border_voxel(A, value, q):
                    new.add(q)
                    break
    return new


def expand(A, value, seeds, lo=None, hi=None):
    """lo <= voxel < hi is the reference's in_range box (default: the whole array)"""
    corner = np.array(A.shape) - 1 if hi is None else np.minimum(np.array(hi), np.array(A.shape) - 1)
    dim = A.ndim
    lo = np.zeros(dim, dtype=int) if lo is None else np.maximum(np.array(lo), 0)
    surface, visited, horizon = set(), set(seeds), set(seeds)
    while horizon:
        nxt = set()
        for v in horizon:
            surface.add(v)
            for o in _OFFS[dim]:
                q = tuple(v[a] + o[a] for a in range(dim))
                if q in visited or np.any(np.array(q) < lo) or np.any(np.array(q) >= corner):
                    continue
                visited.add(q)
                if border_voxel(A, value, q):
                    nxt.add(q)
        horizon = nxt
    return surface


def triangle_voxels(keys, tris, shape):
    """lower corner of the voxel that emitted each Level-0 triangle: the componentwise minimum over the lattice
    end points of its three edges (every Kuhn tetrahedron contains corner 0 of its voxel)"""
    keys = np.asarray(keys, dtype=np.int64)
    lin, d = keys >> 3, keys & 7
    q = np.stack([lin // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], axis=1)
    return q[np.asarray(tris)].min(axis=1)


def select(A, value, end_points, keys, tris, lo=None, hi=None):
    "mask over the Level-0 triangles: emitted by a voxel the reference's seeded search reaches"
    surf = expand(A, value, initial_voxels(A, value, end_points), lo, hi)
    vox = triangle_voxels(keys, tris, A.shape)
    return np.array([tuple(v) in surf for v in vox], dtype=bool), surf


def tetrahedron_voxels(keys, tets, shape):
    """lower corner of the hyper-voxel that emitted each Level-0 tetrahedron of the 4-D march: the componentwise
    minimum over the owners of its four edges (every tetrahedron touches all five corners of its pentatope, and every
    pentatope contains corner 0 of its hypercube)"""
    keys = np.asarray(keys, dtype=np.int64)
    lin = keys >> 4
    s3, s2, s1 = shape[3], shape[2] * shape[3], shape[1] * shape[2] * shape[3]
    q = np.stack([lin // s1, (lin // s2) % shape[1], (lin // s3) % shape[2], lin % shape[3]], axis=1)
    return q[np.asarray(tets)].min(axis=1)


def select4d(A, value, end_points, keys, tets):
    "mask over the Level-0 tetrahedra: emitted by a hyper-voxel the reference's seeded search reaches"
    surf = expand(A, value, initial_voxels(A, value, end_points))
    vox = tetrahedron_voxels(keys, tets, A.shape)
    return np.array([tuple(int(x) for x in v) in surf for v in vox], dtype=bool), surf
