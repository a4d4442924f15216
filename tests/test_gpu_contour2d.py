"""GPU: 2-D contour lines (csrc/cx_contour2d.hip through the C ABI) against the oracle (oracle/contour2d.py),
against polylines written by the real reference (tests/golden2d), and -- at sizes the oracle cannot reach --
through properties of the polylines themselves."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden2d", "*.npz")))


def smooth2(shape, seed, passes):
    rng = np.random.RandomState(seed)
    B = rng.standard_normal(shape)
    for _ in range(passes):
        for ax in range(2):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    return (B / B.std()).astype(np.float32)


def device_chains(A, values, seeds=None, flags=0, mins_delta=None, ctx=None):
    "per level [(closed, points, pairs (k,4) oriented low -> high)]"
    from contourist_amd import _ffi, triangulated
    ctx = ctx or _ffi.Context(0)
    pts, keys, chains, npairs = ctx.contour2d(A, values, seeds, flags, mins_delta)
    i, j, bi, bj, lvl = triangulated.unpack_keys(keys, A.shape[1])
    flip = ~(A[i, j] < A[bi, bj])
    pairs = np.stack([np.where(flip, bi, i), np.where(flip, bj, j), np.where(flip, i, bi), np.where(flip, j, bj)], axis=1)
    out = [[] for _ in values]
    for c in chains:
        a, b = int(c["first"]), int(c["first"]) + int(c["count"])
        assert np.all(lvl[a:b] == c["level"])
        out[int(c["level"])].append((bool(c["closed"]), pts[a:b], pairs[a:b]))
    return out, npairs


FIELDS = {
    "noise_61x47": (smooth2((61, 47), 5, 2), [-0.8, -0.1, 0.0, 0.4, 1.3]),
    "noise_rough_33x90": (smooth2((33, 90), 6, 0), [-0.5, 0.5]),
    "two_by_two": (np.array([[0.0, 1.0], [2.0, 3.0]], dtype=np.float32), [0.5, 1.5, 2.5]),
    "thin_2x40": (smooth2((2, 40), 7, 1), [0.0]),
    "thin_40x2": (smooth2((40, 2), 8, 1), [0.0]),
    "levels_equal_samples": (np.round(smooth2((20, 22), 9, 2) * 4).astype(np.float32) / 4, [-0.5, 0.25, 0.5]),
}


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_every_chain_equals_oracle(name):
    """all polylines before the duplicate drop: the same sequences of lattice pairs (exactly, modulo rotation and
    direction) and the same points (float64, 1e-12) as the restated reference under the build's tie rule"""
    from contourist_amd import _ffi
    from oracle import contour2d as o2
    A, values = FIELDS[name]
    got, npairs = device_chains(A, values, None, _ffi.CX2_ALL_CHAINS | _ffi.CX2_NO_DEDUPE)
    total = 0
    for k, v in enumerate(values):
        want = o2.contours(A, v, "all", "build", dedupe=False)
        assert o2.canonical_keys(got[k]) == o2.canonical_keys(want), "level %r" % v
        where = {}
        for _, p, keys in want:
            for row, q in zip(keys, p):
                where[tuple(int(x) for x in row)] = q
        for _, p, keys in got[k]:
            for row, q in zip(keys, p):
                assert np.max(np.abs(where[tuple(int(x) for x in row)] - q)) <= 1e-12
            total += len(p)
    assert total == npairs


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_search_seeds_and_duplicate_drop(name):
    "default mode (grid-search seeds, duplicates dropped) == oracle; the drop is the predecessor rule on the raw chains"
    from contourist_amd import _ffi
    from oracle import contour2d as o2
    A, values = FIELDS[name]
    raw, _ = device_chains(A, values, None, _ffi.CX2_NO_DEDUPE)
    got, _ = device_chains(A, values, None, 0)
    for k, v in enumerate(values):
        want = o2.contours(A, v, None, "build", dedupe=False)
        assert o2.canonical_keys(raw[k]) == o2.canonical_keys(want), "level %r" % v
        assert len(raw[k]) == len(got[k])
        for (c0, p0, k0), (c1, p1, k1) in zip(raw[k], got[k]):
            keep = o2.drop_close_to_previous(p0)
            assert np.array_equal(p0[keep], p1) and np.array_equal(k0[keep], k1)
            if not c0:
                assert c1 == o2._close(p1[0], p1[-1])
            else:
                assert c1 or o2._close(p0[0], p0[-1])


@pytest.mark.parametrize("fn", GOLDEN, ids=[os.path.basename(f)[:-4] for f in GOLDEN])
def test_api_equals_reference_goldens(fn):
    "the mirrored classes return the polylines the real reference returned (canonical form, 1e-7)"
    from contourist_amd import triangulated, multiple_2d_contour, grid_field
    from oracle import contour2d as o2
    z = np.load(fn)
    A, ref = z["A"], o2.unpack_golden(z)
    n, m = A.shape

    def f(i, j):
        return float(A[int(i), int(j)])
    if "world" in z.files:
        x0, y0, dx, dy = z["world"]
        grid = grid_field.FunctionGrid((x0, y0), (x0 + dx * (n - 1 + 0.25), y0 + dy * (m - 1 + 0.25)), (dx, dy),
                                       lambda x, y: f(round((x - x0) / dx), round((y - y0) / dy)))
        for v, seqs in ref:
            got = triangulated.DxDy2DContourGrid(grid, v).get_contour_sequences()
            assert o2.canonical(got) == o2.canonical(seqs)
    elif "end_points" in z.files:
        for v, seqs in ref:
            G = triangulated.Grid2DContour(n, m, f, v, [[tuple(a), tuple(b)] for a, b in z["end_points"].tolist()])
            assert o2.canonical(G.get_contour_sequences()) == o2.canonical(seqs)
    elif len(ref) == 1:
        v, seqs = ref[0]
        G = triangulated.Grid2DContour(n, m, f, v, None)
        got = G.get_contour_sequences()
        assert o2.canonical(got) == o2.canonical(seqs)
        assert all(isinstance(c, bool) and p.dtype == np.float64 and p.shape[1] == 2 for c, p in got)
    else:
        grid = grid_field.FunctionGrid((0.0, 0.0), (n - 1 + 0.25, m - 1 + 0.25), (1.0, 1.0), lambda x, y: f(round(x), round(y)))
        M = multiple_2d_contour.Multiple2DContourGrid(grid, [v for v, _ in ref])
        d = M.get_contours_dictionary()
        assert sorted(d) == sorted(v for v, _ in ref)
        for v, seqs in ref:
            assert o2.canonical(d[v]) == o2.canonical(seqs), "level %r" % v


def test_seeded_growth_equals_oracle():
    "explicit end points: bisection on the host, growth groups on the device"
    from contourist_amd import triangulated
    from oracle import contour2d as o2
    A = smooth2((50, 44), 15, 3)
    v = 0.2
    L = o2.Lattice(A, v, "build")
    crossing = L.search_grid()
    for pick in ([3], [0, len(crossing) // 2], [len(crossing) - 1, 7, 11]):
        eps = [crossing[k] for k in pick]
        # make the pairs long so that the bisection has work to do
        eps_far = [((0, 0), (A.shape[0] - 1, A.shape[1] - 1))] if (A[0, 0] - v) * (A[-1, -1] - v) < 0 else []
        for seeds in (eps, eps + eps_far):
            G = triangulated.Grid2DContour(A.shape[0], A.shape[1], None, v, seeds, samples=A)
            want = o2.contours(A, v, seeds, "build")
            assert o2.canonical(G.get_contour_sequences()) == o2.canonical([(c, p) for c, p, _ in want])
            assert 0 < len(G.get_contour_sequences()) <= len(o2.contours(A, v, "all", "build"))


def test_attributes_of_the_contour_maker():
    from contourist_amd import triangulated
    from oracle import contour2d as o2
    z = np.load([f for f in GOLDEN if "circle_24x20" in f][0])
    A = z["A"]
    G = triangulated.Grid2DContour(24, 20, None, 30.0, None, samples=A)
    pairs = G.interpolated_contour_pairs
    want = o2.Lattice(A, 30.0, "build").all_pairs()
    assert set(pairs) == set(want)
    assert all(np.max(np.abs(np.array(want[k]) - pairs[k])) <= 1e-12 for k in want)
    triples = G.triangle_triples
    assert len(triples) == len(pairs) and all(len(t) == 3 for t in triples)
    assert len(G.end_points) > 0 and G.in_range((23, 19)) and not G.in_range((24, 0))
    svg = triangulated.contour_sequences_to_svg(G.get_contour_sequences())
    assert svg.count("<path") == 1 and " Z" in svg


def test_device_resident_samples_and_context_reuse():
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi, triangulated
    A = smooth2((300, 257), 16, 3)
    t = torch.from_numpy(A).cuda()
    ctx = _ffi.Context(0)
    a, _ = device_chains(A, [0.0, 0.7], ctx=ctx)
    G = triangulated.Grid2DContour(300, 257, None, 0.0, None, samples=t, context=ctx)
    got = G._extract([0.0, 0.7], None)
    want = ctx.contour2d(A, [0.0, 0.7])
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    # a smaller problem after a larger one on the same context
    small, _ = device_chains(A[:40, :30].copy(), [0.0], ctx=ctx)
    fresh, _ = device_chains(A[:40, :30].copy(), [0.0])
    assert len(small[0]) == len(fresh[0]) and all(np.array_equal(x[1], y[1]) for x, y in zip(small[0], fresh[0]))


def test_empty_and_invalid_inputs():
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    A = smooth2((16, 16), 17, 1)
    pts, keys, chains, npairs = ctx.contour2d(A, [100.0])
    assert len(pts) == 0 and len(chains) == 0 and npairs == 0
    with pytest.raises(_ffi.CxError):
        ctx.contour2d(A, [1.0, 1.0])
    with pytest.raises(_ffi.CxError):
        ctx.contour2d(A, [2.0, 1.0])
    with pytest.raises(_ffi.CxError):
        ctx.contour2d(A[:1], [0.0])
    B = A.copy()
    B[3, 4] = np.nan
    B[9, 9] = np.inf
    pts, keys, chains, npairs = ctx.contour2d(B, [0.0], None, _ffi.CX2_ALL_CHAINS)
    assert npairs > 0 and int(chains["count"].sum()) == len(pts)
    # more than 2^29 crossings in one call are refused (the 32-bit scan would wrap at 2^32: the total is also counted in 64 bits)
    W = np.random.RandomState(3).standard_normal((2048, 2048)).astype(np.float32)
    with pytest.raises(_ffi.CxError) as e:
        ctx.contour2d(W, np.linspace(-3.0, 3.0, 1000))
    assert e.value.code == -6 and "crossings" in str(e.value)
    # ... and more levels than the LDS copy holds are searched in global memory, same result
    many = np.linspace(-2.0, 2.0, 1500)
    few = many[700:710]
    a = ctx.contour2d(A, many, None, _ffi.CX2_ALL_CHAINS)
    b = ctx.contour2d(A, few, None, _ffi.CX2_ALL_CHAINS)
    sel = np.isin(a[2]["level"], np.arange(700, 710))
    assert sel.sum() == len(b[2]) and np.array_equal(a[2]["count"][sel], b[2]["count"])
    pa = np.concatenate([a[0][c["first"]:c["first"] + c["count"]] for c in a[2][sel]]) if sel.any() else np.zeros((0, 2))
    assert np.array_equal(pa, b[0])
    # seeds outside the lattice or on the wrong side are ignored
    seeds = np.array([[-1, 0, 0, 0], [16, 3, 1, 0], [2, 2, 0, 5], [2, 2, 7, 0]], dtype=np.int32)
    pts, keys, chains, _ = ctx.contour2d(A, [0.0], seeds)
    assert len(chains) == 0


def test_large_field_many_levels_properties():
    """2048 x 3072 samples, 16 levels: every crossing is on exactly one polyline; consecutive points share a lattice
    triangle; open polylines end on the rim of the lattice; closed ones return to a neighbour of their start"""
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi, triangulated
    g = torch.Generator(device="cuda").manual_seed(99)
    t = torch.randn((2048, 3072), device="cuda", generator=g)
    for _ in range(6):
        t = 0.25 * torch.roll(t, 1, 0) + 0.5 * t + 0.25 * torch.roll(t, -1, 0)
        t = 0.25 * torch.roll(t, 1, 1) + 0.5 * t + 0.25 * torch.roll(t, -1, 1)
    t = (t / t.std()).contiguous()
    n, m = t.shape
    values = np.linspace(-2.0, 2.0, 16)
    ctx = _ffi.Context(0)
    pts, keys, chains, npairs = ctx.contour2d(None, values, None, _ffi.CX2_ALL_CHAINS | _ffi.CX2_NO_DEDUPE, device_ptr=t.data_ptr(), shape=(n, m))
    assert len(pts) == npairs and len(np.unique(keys)) == npairs
    # number of crossings == what the samples say (f < z <= f' on every lattice edge, three directions)
    A = t.double()
    V = torch.from_numpy(values).cuda()
    expect = 0
    for (di, dj) in ((1, 0), (0, 1), (1, 1)):
        a, b = A[:n - di, :m - dj], A[di:, dj:]
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        expect += int((torch.searchsorted(V, hi.contiguous(), right=True) - torch.searchsorted(V, lo.contiguous(), right=True)).sum())
    assert npairs == expect
    first, count = chains["first"].astype(np.int64), chains["count"].astype(np.int64)
    assert first[0] == 0 and np.all(first[1:] == first[:-1] + count[:-1]) and first[-1] + count[-1] == npairs
    i, j, bi, bj, lvl = triangulated.unpack_keys(keys, m)
    assert np.all(lvl == np.repeat(chains["level"], count))
    # consecutive crossings lie on two edges of one lattice triangle: together they touch exactly 3 lattice points
    same = np.ones(npairs, dtype=bool)
    same[first] = False
    k = np.nonzero(same)[0]
    pa = np.stack([i[k - 1] * m + j[k - 1], bi[k - 1] * m + bj[k - 1], i[k] * m + j[k], bi[k] * m + bj[k]], axis=1)
    pa.sort(axis=1)
    assert np.all((np.diff(pa, axis=1) != 0).sum(axis=1) == 2)
    step = np.abs(pts[k] - pts[k - 1]).max(axis=1)
    assert step.max() <= 1.0 + 1e-9
    # open chains start and end on the rim
    last = first + count - 1
    opened = chains["closed"] == 0
    for idx in (first[opened], last[opened]):
        on_rim = (np.minimum(i[idx], bi[idx]) == 0) | (np.maximum(i[idx], bi[idx]) == n - 1) | (np.minimum(j[idx], bj[idx]) == 0) | \
                 (np.maximum(j[idx], bj[idx]) == m - 1)
        assert np.all(on_rim)
    closed = ~opened
    gap = np.abs(pts[first[closed]] - pts[last[closed]]).max(axis=1)
    assert gap.max() <= 1.0 + 1e-9
    # the same job with the duplicate drop and the grid-search seeds loses nothing but near-duplicate points (np.allclose scales with the coordinate: 0.03 lattice units at 3000)
    pts2, keys2, chains2, _ = ctx.contour2d(None, values, None, 0, device_ptr=t.data_ptr(), shape=(n, m))
    assert len(chains) - 1 <= len(chains2) <= len(chains) and len(pts2) <= len(pts) and len(pts2) > 0.95 * len(pts)


def test_reference_demo_and_level_classes():
    """the reference's own svg demo (triangulated.py:56-61) and the two classes that choose their own levels
    (multiple_2d_contour.py:84-108) run through the mirrored API"""
    import math
    from contourist_amd import triangulated, multiple_2d_contour
    from oracle import contour2d as o2
    svg = triangulated.svg_demo()
    assert svg.count("<path") == 3 and "viewBox" in svg          # the reference finds 3 open contours on this grid

    def f(x, y):
        return x * x + y * (y + 1) * (y - 1) - math.sin(2 * y * y + 4 * x)
    for cls, n in ((multiple_2d_contour.Linear2DContour, 5), (multiple_2d_contour.Percentile2DContour, 4)):
        M = cls(-1, -1, 1, 1, 0.05, 0.05, f, breakpoints=n)
        d = M.get_contours_dictionary()
        assert sorted(d) == sorted(set(float(v) for v in M.values))
        nn, mm, A = triangulated.grid_lattice(M.grid)
        for v in d:
            want = [(c, p * M.grid.delta + M.grid.mins) for c, p, _ in o2.contours(A, float(v), None, "build")]
            assert o2.canonical(d[v]) == o2.canonical(want), "level %r" % v


def test_random_fields_levels_and_seeds_equal_oracle():
    """60 random cases: shapes 2..40, rough to smooth, some quantised (many samples equal to a level), 1..5 levels,
    a third of them with explicit end points: polylines == oracle, pair for pair"""
    from contourist_amd import _ffi
    from oracle import contour2d as o2
    ctx = _ffi.Context(0)
    rng = np.random.RandomState(2024)
    for case in range(60):
        n, m = int(rng.randint(2, 41)), int(rng.randint(2, 41))
        A = smooth2((n, m), 3000 + case, int(rng.randint(0, 4))) if n * m > 4 else rng.standard_normal((n, m)).astype(np.float32)
        if rng.rand() < 0.3:
            A = (np.round(A * 4) / 4).astype(np.float32)
        raw = rng.uniform(-1.2, 1.2, size=int(rng.randint(1, 6)))
        values = sorted(set((np.round(raw * 4) / 4).tolist())) if rng.rand() < 0.5 else sorted(set(np.round(raw, 3).tolist()))
        values = [float(v) for v in values]
        seeded = rng.rand() < 0.33
        for k, v in enumerate(values):
            L = o2.Lattice(A, v, "build")
            eps = None
            if seeded:
                cand = L.search_grid()
                if not cand:
                    continue
                pick = rng.choice(len(cand), size=min(len(cand), 1 + int(rng.randint(0, 3))), replace=False)
                eps = [cand[int(p)] for p in pick]
            want = o2.contours(A, v, eps, "build", dedupe=False)
            seeds = None
            if eps is not None:
                rows = []
                for (a, b) in eps:
                    lo, hi = L.seed_points(a, b)
                    rows += [(lo[0], lo[1], 0, 0), (hi[0], hi[1], 1, 0)]
                seeds = np.array(rows, dtype=np.int32)
            got, _ = device_chains(A, [v], seeds, _ffi.CX2_NO_DEDUPE, ctx=ctx)
            assert o2.canonical_keys(got[0]) == o2.canonical_keys(want), "case %d shape %s level %r seeded %s" % (case, A.shape, v, seeded)
