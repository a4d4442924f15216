"""Morphing triangles of a 4-D isosurface -- host-side mirror of the reference's
`contourist/morph_geometry.py` class MorphTriangles (:5-125).  The slicing of tetrahedra
(MorphGeometry.add_tetrahedron, :155-192) and the time-aware orientation (:49-89) run on the device
(cx_morph_triangles); this class only carries the result and its coordinate / JSON conversions.

    points4d                  (V,4) float64   x, y, z, t
    segment_point_indices     (S,2) int       point indices, low t -> high t      (morph_geometry.py:12-18)
    triangle_segment_indices  (T,3) int       segment indices, consistently wound (morph_geometry.py:49-59)
A vertex of a triangle at time t is the point at parameter (t - t_lo)/(t_hi - t_lo) of its segment; the
triangle exists while t lies inside all three segment intervals (misc/morph_triangles.js:26-140).
"""
import numpy as np


class MorphTriangles(object):

    def __init__(self, points4d, segment_point_indices, triangle_segment_indices):
        self.points4d = points4d = np.array(points4d, dtype=float).reshape(-1, 4)
        t_values = points4d[:, -1]
        self.max_value = t_values.max() if len(t_values) else 0.0
        self.min_value = t_values.min() if len(t_values) else 0.0
        seg = np.array(segment_point_indices, dtype=np.int64).reshape(-1, 2)
        if len(seg):
            swap = points4d[seg[:, 0], -1] > points4d[seg[:, 1], -1]
            seg = np.where(swap[:, None], seg[:, ::-1], seg)
        self.segment_point_indices = seg
        self.triangle_segment_indices = np.array(triangle_segment_indices, dtype=np.int64).reshape(-1, 3)

    def from_grid_coordinates(self, grid):
        "same triangles with points in world coordinates (morph_geometry.py:24-26)"
        points4d = grid.from_grid_coordinates(self.points4d) if len(self.points4d) else self.points4d
        return MorphTriangles(points4d, self.segment_point_indices, self.triangle_segment_indices)

    def triangles_at(self, t):
        """(points (P,3), triangles (Q,3)) of the surface at time t: the consumer-side evaluation of
        misc/morph_triangles.js:26-140 (lerp along each segment; a triangle is visible while t is inside all
        three of its segments' intervals)."""
        P, S, T = self.points4d, self.segment_point_indices, self.triangle_segment_indices
        if len(T) == 0:
            return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
        lo, hi = P[S[:, 0], 3], P[S[:, 1], 3]
        inside = (lo <= t) & (t <= hi)
        vis = inside[T].all(axis=1)
        lam = np.where(hi > lo, (t - lo) / np.where(hi > lo, hi - lo, 1.0), 0.0)
        pos = P[S[:, 0], :3] + lam[:, None] * (P[S[:, 1], :3] - P[S[:, 0], :3])
        used = np.unique(T[vis].reshape(-1))
        remap = -np.ones(len(S), dtype=np.int64)
        remap[used] = np.arange(len(used))
        return pos[used], remap[T[vis]]

    def to_json(self, min_value=None, max_value=None, maxint=999999, epsilon=1e-4):
        """the compact integer JSON that misc/morph_triangles.js reads (wire format of morph_geometry.py:91-125):
        positions quantised to 0..maxint per axis with their shift and scale, then the flattened index lists"""
        t_lo = self.min_value if min_value is None else max(min_value, self.min_value)
        t_hi = self.max_value if max_value is None else min(max_value, self.max_value)
        P = np.asarray(self.points4d, dtype=np.float64)
        shift = P.min(axis=0)
        scale = np.maximum(P.max(axis=0) - shift, epsilon) / maxint
        quantised = ((P - shift[None, :]) * (1.0 / scale)[None, :]).astype(int)

        def vector(values):
            return "[" + ", ".join("%s" % (x,) for x in values) + "]"
        fields = [("description", '"Ordered 4d morphing triangles."'),
                  ("max_value", "%s" % (t_hi,)), ("min_value", "%s" % (t_lo,)),
                  ("counts", vector((len(P), len(self.segment_point_indices), len(self.triangle_segment_indices)))),
                  ("shift", vector(shift)), ("scale", vector(scale)),
                  ("positions", flatten_json_list(quantised)),
                  ("segments", flatten_json_list(self.segment_point_indices)),
                  ("triangles", flatten_json_list(self.triangle_segment_indices))]
        return "{\n" + ",\n".join('"%s": %s' % field for field in fields) + "\n}"


def flatten_json_list(sequence, fmt=str):
    "rows joined by commas, one row per line, in one pair of brackets (morph_geometry.py:127-128)"
    return "[" + ",\n".join(",".join(map(fmt, row)) for row in sequence) + "]"
