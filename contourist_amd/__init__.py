"""contourist_amd -- MI355X (gfx950) isosurface extraction behind contourist's Python API.

Modules mirror the reference package for the tetrahedral / pentatope voxel-march path:
    grid_field        FunctionGrid                       (reference contourist/grid_field.py)
    tetrahedral       TriangulatedIsosurfaces, Delta3DContour, Grid3DContour, GridContour3d
                                                         (reference contourist/tetrahedral.py)
    surface_geometry  SurfaceGeometry                    (reference contourist/surface_geometry.py)
    pentatopes, morph_geometry   MorphingIsoSurfaces, MorphTriangles       (reference contourist/pentatopes.py, morph_geometry.py)
    triangulated, multiple_2d_contour, field2d   2-D contour lines at one / several isovalues
                                                         (reference contourist/triangulated.py, multiple_2d_contour.py, field2d.py)
All compute runs in hand-written HIP kernels loaded through ctypes (contourist_amd/_ffi.py);
there is no CPU fallback.
"""
__version__ = "0.1.0"
