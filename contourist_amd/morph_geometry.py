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
        "compact integer JSON consumed by misc/morph_triangles.js (morph_geometry.py:91-125)"
        L = []
        a = L.append
        a("{\n")
        a('"description": "Ordered 4d morphing triangles.",\n')
        min_value = self.min_value if min_value is None else max(min_value, self.min_value)
        max_value = self.max_value if max_value is None else min(max_value, self.max_value)
        a('"max_value": %s,\n' % (max_value,))
        a('"min_value": %s,\n' % (min_value))
        points = self.points4d
        segments = self.segment_point_indices
        triangles = self.triangle_segment_indices
        a('"counts": [%s, %s, %s],\n' % (len(points), len(segments), len(triangles),))
        maxima = points.max(axis=0)
        minima = points.min(axis=0)
        diff = np.maximum(maxima - minima, epsilon)
        a('"shift": [%s, %s, %s, %s],\n' % tuple(minima))
        scale = diff / maxint
        a('"scale": [%s, %s, %s, %s],\n' % tuple(scale))
        invscale = (1.0 / scale).reshape((1, 4))
        positions = ((points - minima.reshape(1, 4)) * invscale).astype(int)
        a('"positions": %s,\n' % (flatten_json_list(positions),))
        a('"segments": %s,\n' % (flatten_json_list(segments),))
        a('"triangles": %s\n' % (flatten_json_list(triangles),))
        a("}")
        return "".join(L)


def flatten_json_list(sequence, fmt=str):
    return "[%s]" % (",\n".join(",".join(fmt(y) for y in x) for x in sequence),)
