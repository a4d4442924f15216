"""CPU: host-side behaviour of the reference-shaped API that needs no device: constructor argument
handling, assertions where the reference asserts, loud failure without a GPU (no CPU fallback)."""
import numpy as np
import pytest

from contourist_amd import _ffi, grid_field, pentatopes, tetrahedral


def sphere(x, y, z):
    return x * x + y * y + z * z


def test_constructors_and_attributes():
    d = 0.25
    S = tetrahedral.TriangulatedIsosurfaces([-1] * 3, [1] * 3, [d] * 3, sphere, 0.5, [])
    assert isinstance(S.grid, grid_field.FunctionGrid) and S.value == 0.5
    assert S.grid_endpoints is None                       # [] -> grid search (triangulated.py:100-102)
    assert isinstance(S.contour_maker, tetrahedral.GridContour3d)
    assert tuple(S.contour_maker.corner) == tuple(S.grid.grid_dimensions)
    assert S.contour_maker.samples.shape == tuple(n + 1 for n in S.grid.grid_dimensions)
    # 3-D endpoints are accepted (the reference's ctor asserts len == 2, a 2-D leftover)
    T = tetrahedral.TriangulatedIsosurfaces([-1] * 3, [1] * 3, [d] * 3, sphere, 0.5, [((0, 0, 0), (1, 1, 1))])
    assert T.grid_endpoints is not None and len(T.grid_endpoints) == 1
    p, q = T.grid_endpoints[0]
    assert (T.grid.grid_function(*p) - 0.5) * (T.grid.grid_function(*q) - 0.5) <= 0
    assert tetrahedral.CUBE.shape == (8, 3) and tetrahedral.TETRAHEDRA.shape == (6, 4, 3) and tetrahedral.OFFSETS.shape == (26, 3)
    assert pentatopes.PENTATOPES.shape == (24, 5, 4) and pentatopes.HYPERCUBE.shape == (16, 4) and pentatopes.OFFSETS4D.shape == (80, 4)


def test_reference_asserts_and_unsupported_options():
    A = np.zeros((5, 5, 5), dtype=np.float32)
    with pytest.raises(AssertionError):                    # tetrahedral.py:526 sanity_check
        tetrahedral.GridContour3d((4, 4), A[0], 0.0)
    with pytest.raises(AssertionError):                    # tetrahedral.py:153-155 endpoint dimension
        tetrahedral.GridContour3d((4, 4, 4), A, 0.0, [((0, 0), (1, 1))])
    with pytest.raises(NotImplementedError):
        tetrahedral.GridContour3d((4, 4, 4), A, 0.0, linear_interpolate=False)
    with pytest.raises(NotImplementedError):
        tetrahedral.TriangulatedIsosurfaces([0] * 3, [4] * 3, [1] * 3, A, 0.0, [], flatten=True)
    with pytest.raises(AssertionError):
        pentatopes.GridContour4D((4, 4, 4), A, 0.0)


def test_no_gpu_fails_loudly():
    """on a host without a HIP device the product path raises; it never falls back to a CPU implementation"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.CxError):
        _ffi.Context(0)
    S = tetrahedral.TriangulatedIsosurfaces([-1] * 3, [1] * 3, [0.5] * 3, sphere, 0.5, [])
    with pytest.raises(_ffi.CxError):
        S.search_for_endpoints()


def test_bisect_endpoints_leaves_out_pairs_that_end_outside_the_samples():
    """an end point pair far outside the grid whose crossing is ALSO outside the sampled array must not reach the
    device (cx_select_seeded3d would reject the whole call); pairs that end inside are bisected exactly as the reference does
    (tetrahedral.py:408-423)."""
    import numpy as np
    from contourist_amd import tetrahedral

    def f(i, j, k):                     # a plane at i = 40.5: the crossing is far outside a 12^3 grid
        return float(i) - 40.5
    dropped = []
    out = tetrahedral.bisect_endpoints(f, 0.0, [((0, 0, 0), (100, 100, 100)), ((3, 3, 3), (4, 3, 3))], [-1] * 3, [13] * 3, dropped)
    assert len(out) == 1 and tuple(out[0][0]) == (3, 3, 3)            # the inside pair is handed on untouched
    assert len(dropped) == 1 and np.all(np.abs(dropped[0][0] - dropped[0][1]) <= 1) and dropped[0][0][0] == 40

    def g(i, j, k):                     # a plane at i = 5.5: the far pair's bisection ends inside the grid
        return float(i) - 5.5
    out = tetrahedral.bisect_endpoints(g, 0.0, [((0, 0, 0), (100, 100, 100))], [-1] * 3, [13] * 3)
    assert len(out) == 1 and out[0][0][0] == 5 and out[0][1][0] == 6


def test_slab_bounds_of_a_volume_beyond_one_extraction():
    """GridContour3d._slab_bounds / _slab_planes (host logic of the slab path for volumes of more than 2^29 samples): the slabs cover
    every plane once, in order; none is a single plane (it would hold no voxel); a slab plus its halo plane fits one extraction"""
    from contourist_amd import tetrahedral
    B = tetrahedral.GridContour3d._slab_bounds
    for n0 in range(2, 60):
        for planes in range(2, 20):
            b = B(n0, planes)
            assert b[0][0] == 0 and b[-1][1] == n0
            assert all(b[k][1] == b[k + 1][0] for k in range(len(b) - 1))
            assert all(i1 - i0 >= 2 or n0 < 2 for (i0, i1) in b)
            # with its halo plane (all but the last) a slab holds at most planes + 1 planes
            assert all((i1 - i0) + (1 if i1 < n0 else 0) <= planes + 1 for (i0, i1) in b)

    class Fake(object):
        MAX_SAMPLES_PER_EXTRACTION = 1 << 29
        shape = (1056, 720, 720)
    assert tetrahedral.GridContour3d._in_slabs(Fake()) and not tetrahedral.GridContour3d._in_slabs(type("S", (Fake,), {"shape": (512, 512, 512)})())
    planes = tetrahedral.GridContour3d._slab_planes(Fake())
    assert (planes + 1) * 720 * 720 <= (1 << 29) < (planes + 2) * 720 * 720
    small = type("T", (Fake,), {"shape": (4, 30000, 30000)})()
    with pytest.raises(ValueError):
        tetrahedral.GridContour3d._slab_planes(small)
