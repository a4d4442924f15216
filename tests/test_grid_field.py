"""CPU: host-side FunctionGrid mirror -- the transform / cache / materialise semantics the reference pins in
contourist/test/test_field2d.py:22-61 (there in 2-D through field2d.Function2DGrid, a thin wrapper over
grid_field.FunctionGrid), plus the dense-sample view the device path uses."""
import numpy as np
import pytest

from contourist_amd import grid_field


def function(x, y):
    return (x + 100) * 1000 + (y + 100)


@pytest.mark.parametrize("materialize,cache", [(False, False), (False, True), (True, False)])
def test_transforms_cache_materialize(materialize, cache):
    # the reference test builds Function2DGrid(xmin=-10, ymin=-20, xmax=30, ymax=50, dx=10, dy=20)
    grid = grid_field.FunctionGrid([-10, -20], [30, 50], [10.0, 20.0], function, materialize, cache)
    for iteration in (1, 2):
        assert np.allclose(grid.to_grid_coordinates(np.array((-10, -20))), (0, 0))
        assert np.allclose(grid.from_grid_coordinates((0, 0)), (-10, -20))
        assert np.allclose(grid.to_grid_coordinates(np.array((0, 0))), (1, 1))
        assert np.allclose(grid.from_grid_coordinates((1, 1)), (0, 0))
        assert np.allclose(grid.grid_function(0, 0), 90080)
        assert np.allclose(grid.grid_function(4, 3), 130140)
        S = set(tuple(int(v) for v in x) for x in grid.surrounding_vertices(np.array((5, 5))))
        assert S == set([(1, 2), (1, 1), (2, 1), (2, 2)])
    assert tuple(grid.grid_dimensions) == (5, 4)
    if cache:
        assert grid.cache == {(0, 0): 90080.0, (4, 3): 130140.0}
    else:
        assert grid.cache == {}
    if materialize:
        expect = [[90080.0, 90100.0, 90120.0, 90140.0], [100080.0, 100100.0, 100120.0, 100140.0],
                  [110080.0, 110100.0, 110120.0, 110140.0], [120080.0, 120100.0, 120120.0, 120140.0],
                  [130080.0, 130100.0, 130120.0, 130140.0]]
        assert np.allclose(grid.materialized_array, expect)
    else:
        assert grid.materialized_array is None


def test_dense_samples_and_crossing_search_match_the_oracle():
    from oracle import level0
    rng = np.random.RandomState(1)
    A = rng.standard_normal((9, 10, 11)).astype(np.float32)
    G = grid_field.FunctionGrid.from_array(A, mins=[1.0, 2.0, 3.0], delta=[0.5, 0.25, 2.0])
    assert tuple(G.grid_dimensions) == (8, 9, 10)
    assert G.dense_samples() is not None and G.dense_samples().shape == A.shape
    assert np.allclose(G.from_grid_coordinates((2, 4, 1)), (2.0, 3.0, 5.0))
    assert G.grid_function(2, 4, 1) == pytest.approx(float(A[2, 4, 1]))
    maxf, minf, segs = G.find_contour_crossing_grid_segments(0.1)
    n, mn, mx = level0.count_crossings(A, 0.1)           # grid_field.py:64-84 restated in C
    assert len(segs) == n and minf == pytest.approx(mn) and maxf == pytest.approx(mx)
    for v0, v1 in segs[:50]:
        assert (A[tuple(v0)] - 0.1) * (A[tuple(v1)] - 0.1) < 0 and np.all(v1 - v0 >= 0) and (v1 - v0).max() == 1


def test_callable_is_sampled_vectorised_or_pointwise():
    d = 3.0 / 32
    G = grid_field.FunctionGrid([-1.5] * 3, [1.5 - d] * 3, [d] * 3, lambda x, y, z: x * x + y * y + z * z)
    S = G.dense_samples()
    assert S.shape == (33, 33, 33) and S.dtype == np.float32
    assert S[0, 0, 0] == pytest.approx(3 * 1.5 ** 2) and S[16, 16, 16] == pytest.approx(0.0)

    def scalar_only(x, y, z):            # does not broadcast: falls back to one call per sample
        if isinstance(x, np.ndarray):
            raise TypeError("scalars only")
        return float(x) + 10 * float(y) + 100 * float(z)
    H = grid_field.FunctionGrid([0, 0, 0], [2, 2, 2], [1, 1, 1], scalar_only)
    assert H.dense_samples().shape == (4, 4, 4) and H.dense_samples()[1, 2, 3] == pytest.approx(321.0)


def test_iter_indices_order():
    assert list(grid_field.iter_indices((2, 3))) == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
    assert list(grid_field.iter_indices((4,), 2)) == [(0,), (2,)]


def test_bad_grid_asserts_like_the_reference():
    with pytest.raises(AssertionError):
        grid_field.FunctionGrid([0, 0], [0.5, 0.5], [1, 1], lambda x, y: 0.0)      # grid_field.py:28
    with pytest.raises(AssertionError):
        grid_field.FunctionGrid([0, 0], [4, 4], [1, 1], lambda x, y: 0.0, materialize=True, cache=True)   # :31


def test_crossing_segments_in_the_references_order():
    """find_contour_crossing_grid_segments against what the real reference listed (tests/golden/crossing_segments.npz, by
    oracle/make_goldens_segments.py): same segments in the SAME order -- also with skip > 1, where the forward neighbour
    of the last strided lattice point lies beyond the grid and the reference evaluates the callable there"""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    from contourist_amd import grid_field
    from oracle.make_goldens_segments import field, MINS, MAXES, DELTA, VALUE
    G = np.load(os.path.join(GOLDEN_DIR, "crossing_segments.npz"))
    for skip in (1, 2, 3):
        g = grid_field.FunctionGrid(MINS, MAXES, DELTA, field)
        maxf, minf, segs = g.find_contour_crossing_grid_segments(VALUE, skip)
        got = np.array([list(p) + list(q) for p, q in segs], dtype=np.int32).reshape(-1, 6)
        assert np.array_equal(got, G["skip%d" % skip])
        assert np.allclose([maxf, minf], G["range%d" % skip], rtol=1e-6)
