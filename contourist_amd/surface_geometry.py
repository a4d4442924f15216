"""Triangle-mesh cleanup and orientation -- host-side mirror of the reference's
`contourist/surface_geometry.py` (class SurfaceGeometry :4-140); the passes run on the device
(cx_surface_geometry / cx_postprocess3d in include/contourist_hip.h).

    SurfaceGeometry(vertices, triangles)
        .clean_triangles()    drop zero-area triangles, merge their coincident vertices (:14-50)
        .orient_triangles()   per connected component, wind outward as seen from the vertex with
                              the largest x (:52-140)
    attributes: vertices, triangles, oriented_triangles, vertex_map, input_vertices, input_triangles
"""
import numpy as np

from . import _ffi


class SurfaceGeometry(object):

    def __init__(self, vertices, triangles, context=None, device=0):
        self.input_vertices = vertices
        self.input_triangles = triangles
        self.vertices = vertices
        self.triangles = triangles
        self.oriented_triangles = triangles
        self.vertex_map = tuple(range(len(vertices)))
        self._ctx = context
        self._device = device
        self._cleaned = False

    @classmethod
    def _from_device(cls, points, triangles, context):
        "wrap a mesh that the device post-passes already cleaned and oriented"
        self = cls(points, triangles, context)
        self.oriented_triangles = _sorted_rows(triangles)
        self.triangles = self.oriented_triangles
        self._cleaned = True
        return self

    def _context(self):
        if self._ctx is None:
            self._ctx = _ffi.Context(self._device)
        return self._ctx

    def _arrays(self):
        pts = np.asarray([np.asarray(p, dtype=np.float64) for p in self.vertices], dtype=np.float64).reshape(-1, 3)
        tris = np.asarray([tuple(t) for t in self.triangles if len(tuple(t)) == 3], dtype=np.int32).reshape(-1, 3)
        return pts, tris

    def clean_triangles(self):
        "Remove area 0 triangles and duplicate vertices on area 0 triangles -> (vertices, triangles)"
        pts = np.asarray([np.asarray(p, dtype=np.float64) for p in self.input_vertices], dtype=np.float64).reshape(-1, 3)
        tris = np.asarray([tuple(t) for t in self.input_triangles], dtype=np.int32).reshape(-1, 3)
        p2, t2 = self._context().surface_geometry(pts, tris, do_clean=2)   # clean only
        self.vertices = p2
        self.triangles = t2
        self.oriented_triangles = t2
        self.vertex_map = None
        self._cleaned = True
        return (p2, t2)

    def orient_triangles(self, compatible_triangle_test=None):
        "Orient triangles so cross product of triangle vectors points outwards -> sorted (T,3) rows"
        if compatible_triangle_test is not None:
            raise NotImplementedError("compatible_triangle_test callbacks cannot run on the device; "
                                      "the 4-D morph path has its own time-overlap orientation kernel")
        pts, tris = self._arrays()
        p2, t2 = self._context().surface_geometry(pts, tris, do_clean=0)
        self.oriented_triangles = _sorted_rows(t2)
        return self.oriented_triangles


def _sorted_rows(tris):
    tris = np.asarray(tris).reshape(-1, 3)
    if len(tris) == 0:
        return tris
    return tris[np.lexsort((tris[:, 2], tris[:, 1], tris[:, 0]))]
