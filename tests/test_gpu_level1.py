"""GPU: API-level (Level-1) parity of the device post-passes with the oracle's canonical restatement
and with the outputs of the real reference (goldens), through the reference-shaped Python API."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, golden_names

pytestmark = pytest.mark.gpu

WELL_BEHAVED = ("sphere32", "inv_sphere20", "shells24", "blobs27", "noise24_v0", "noise32_v0")


def run_api(G):
    from contourist_amd import tetrahedral
    A, v = G["A"], float(G["value"])
    mins = G["mins"] if "mins" in G.files else np.zeros(3)
    delta = G["delta"] if "delta" in G.files else np.ones(3)
    smooth = float(G["smooth"]) if "smooth" in G.files else None
    S = tetrahedral.TriangulatedIsosurfaces(list(mins), None, list(delta), A, v, [], smooth=smooth)
    S.search_for_endpoints()
    points, triangles = S.get_points_and_triangles()
    grid_points = (np.asarray(points) - mins) / delta if len(points) else np.zeros((0, 3))
    return S, np.asarray(points), grid_points, np.asarray(triangles)


@pytest.mark.parametrize("name", golden_names())
def test_api_matches_oracle_and_reference(name):
    from oracle import level0, postpass
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    S, points, grid_points, triangles = run_api(G)
    post = S.contour_maker._post
    O = level0.march3d(A, v, diag_mode=1)
    smooth = float(G["smooth"]) if "smooth" in G.files else None
    postpass.set_compare_scale(1e8 if smooth else None)
    L1 = postpass.level1_from_level0(level0.edge_keys_from_pairs(O["pairs"], A.shape), O["xyz"], O["tris"], corner, smooth=smooth)
    # counts through the pipeline: identical to the oracle's canonical pipeline
    assert post["n_after_weld"] == L1["n_after_weld"] == int(G["n_tris_after_weld"])
    assert post["n_after_tiny"] == L1["n_after_tiny"]
    assert len(triangles) == len(L1["triangles"])
    assert triangles.shape[1] == 3 and points.shape[1] == 3
    assert triangles.min(initial=0) >= 0 and triangles.max(initial=-1) < len(points)
    # device vs oracle: same canonical choices => same triangles and winding (ambiguous components excused)
    cmp = postpass.compare_level1(L1, grid_points, triangles, corner, reach=0)
    assert not cmp["missing"] and not cmp["extra"] and not cmp["winding"], {k: (len(x) if isinstance(x, list) else x) for k, x in cmp.items()}
    assert cmp["excused_rows"] == 0
    # coordinates are the reference's float64 interpolation, bit for bit (every output point is a Level-0 point)
    if smooth is None:
        ref_pts = set(map(tuple, np.round(O["xyz"], 12).tolist()))
        got_pts = set(map(tuple, np.round(grid_points, 12).tolist()))
        assert got_pts <= ref_pts
    else:
        # smoothed coordinates: same values as the oracle's up to the order of the float64 sums
        def rows(P):          # order by coordinates rounded to 1e-7 so that 1e-15 differences cannot reorder rows
            R = np.round(P, 7)
            return P[np.lexsort((R[:, 2], R[:, 1], R[:, 0]))]
        a, b = rows(grid_points), rows(L1["grid_points"])
        assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=1e-9)
    # device vs the real reference's output
    reach = 2 * int(postpass.expander_for(corner).max()) if smooth else 2
    cmpr = postpass.compare_level1(L1, G["l1_grid_points"], G["l1_triangles"], corner, reach=reach)
    assert not cmpr["missing"] and not cmpr["extra"] and not cmpr["winding"]
    if name in WELL_BEHAVED:
        if L1["comp_flags"].max(initial=0) == 0 and len(L1["sites"]) == 0:
            # orientable manifold components with an unambiguous start: nothing may need excusing
            assert cmp["excused_winding"] == 0 and cmpr["excused_winding"] == 0
        else:
            # triangle order on the device varies from run to run (atomics), and with it the flood-fill
            # order on the few non-manifold patches welding creates; bounded, never the bulk
            assert cmp["excused_winding"] <= 0.01 * len(triangles) + 8
        # Level-1 canonical forms (bucket triples + winding) are then literally identical
        ref = postpass.canonical_level1(G["l1_grid_points"], G["l1_triangles"], corner)
        got = postpass.canonical_level1(grid_points, triangles, corner)
        if len(L1["sites"]) == 0:
            assert np.array_equal(ref, got)
    # world coordinates (grid_field.py:89-93)
    if "l1_points" in G.files and len(L1["sites"]) == 0 and len(points) == len(G["l1_points"]):
        def wrows(P):
            R = np.round(P, 7)
            return P[np.lexsort((R[:, 2], R[:, 1], R[:, 0]))]
        assert np.allclose(wrows(points), wrows(G["l1_points"]), rtol=0, atol=1e-9 if smooth else 1e-12)


def test_config1_sphere_counts():
    """BASELINE.json configs[0]: 32^3 sphere -> 6386 points / 12768 triangles, Euler characteristic 2"""
    from contourist_amd import tetrahedral
    d = 3.0 / 32
    S = tetrahedral.TriangulatedIsosurfaces([-1.5] * 3, [1.5 - d] * 3, [d] * 3, lambda x, y, z: x * x + y * y + z * z, 1.0, [])
    S.search_for_endpoints()
    points, triangles = S.get_points_and_triangles()
    assert len(points) == 6386 and len(triangles) == 12768
    edges = set()
    for t in triangles:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges.add((min(a, b), max(a, b)))
    assert len(points) - len(edges) + len(triangles) == 2
    # outward orientation: normals point away from the origin
    P = points[triangles]
    n = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    assert np.all(np.einsum("ij,ij->i", n, P.mean(axis=1)) > 0)
    assert np.allclose(np.linalg.norm(points, axis=1), 1.0, atol=0.01)


def test_surface_geometry_standalone():
    """SurfaceGeometry(vertices, triangles).clean_triangles()/orient_triangles() on a caller's mesh with
    scrambled winding"""
    from contourist_amd import surface_geometry
    from oracle import level0, postpass
    G = np.load(os.path.join(GOLDEN_DIR, "shells24.npz"))
    O = level0.march3d(G["A"], float(G["value"]), diag_mode=1)
    rng = np.random.RandomState(0)
    tris = O["tris"].copy()
    flip = rng.rand(len(tris)) < 0.5
    tris[flip] = tris[flip][:, ::-1]
    sg = surface_geometry.SurfaceGeometry(list(O["xyz"]), [tuple(t) for t in tris])
    got = np.asarray(sg.orient_triangles())
    want, label, flags = postpass.orient(O["xyz"], O["tris"])
    assert flags.max(initial=0) == 0

    def canon(T):
        T = np.asarray(T)
        first = np.argmin(T, axis=1)
        idx = (first[:, None] + np.arange(3)[None, :]) % 3
        R = np.take_along_axis(T, idx, axis=1)
        return R[np.lexsort((R[:, 2], R[:, 1], R[:, 0]))]
    assert np.array_equal(canon(got), canon(want))
    # clean: a mesh with an injected zero-area triangle and a duplicate vertex
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=float)
    tri = [(0, 1, 2), (0, 1, 3), (1, 4, 2), (0, 2, 3), (4, 2, 3)]
    sg2 = surface_geometry.SurfaceGeometry(list(pts), tri)
    v2, t2 = sg2.clean_triangles()
    assert len(t2) == 4 and len(v2) == 4      # (1,4,2) dropped, vertices 1 and 4 merged


def test_multi_level_matches_single_levels():
    """config 5 shape: several isovalues of one resident grid == the same levels extracted one by one"""
    from contourist_amd import tetrahedral
    G = np.load(os.path.join(GOLDEN_DIR, "noise32_v0.npz"))
    A = G["A"]
    values = [0.4, -0.5, 0.0]
    M = tetrahedral.MultiLevelIsosurfaces([0, 0, 0], None, [1, 1, 1], A, values)
    out = list(M.levels())
    assert [o[0] for o in out] == sorted(values)
    for v, pts, tris in out:
        S = tetrahedral.TriangulatedIsosurfaces([0, 0, 0], None, [1, 1, 1], A, v, [])
        S.search_for_endpoints()
        p2, t2 = S.get_points_and_triangles()
        assert len(pts) == len(p2) and len(tris) == len(t2)
        a = np.array(sorted(map(tuple, np.round(pts, 9).tolist())))
        b = np.array(sorted(map(tuple, np.round(p2, 9).tolist())))
        assert np.array_equal(a, b)
    assert abs(len(out[1][2]) - len(G["l1_triangles"])) <= 0.002 * len(G["l1_triangles"])   # v=0: the golden's level


def test_random_fields_level1_equals_oracle():
    """24 random closed-interior fields (shapes 9..19 per axis, rough to smooth, random isovalues): the device's weld /
    tiny collapse / clean / orient == the oracle's canonical pipeline -- same counts after every stage, the same
    triangles as weld-bucket triples, the same windings"""
    from contourist_amd import _ffi
    from oracle import level0, postpass
    rng = np.random.RandomState(77)
    ctx = _ffi.Context(0)
    for case in range(24):
        shape = tuple(int(x) for x in rng.randint(9, 20, size=3))
        B = rng.standard_normal(shape)
        for _ in range(int(rng.randint(0, 4))):
            for ax in range(3):
                B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
        B = (B / B.std()).astype(np.float32)
        fill = np.float32(B.min() - 1.0)
        for ax in range(3):
            sl = [slice(None)] * 3
            for idx in (0, 1, -1, -2):
                sl[ax] = idx
                B[tuple(sl)] = fill
        v = float(np.round(rng.uniform(-0.9, 0.9), 3))
        ctx.upload_grid(B)
        ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310)
        post = ctx.postprocess3d(0)
        pts, tris = ctx.download_level1(post)
        O = level0.march3d(B, v, diag_mode=1)
        if len(O["tris"]) == 0:
            assert post["n_triangles"] == 0
            continue
        corner = np.array(shape) - 1
        ko = level0.edge_keys_from_pairs(O["pairs"], shape)
        L1 = postpass.level1_from_level0(ko, O["xyz"], O["tris"], corner)
        where = "case %d shape %s v=%g" % (case, shape, v)
        assert post["n_after_weld"] == L1["n_after_weld"] and post["n_after_tiny"] == L1["n_after_tiny"], where
        assert len(tris) == len(L1["triangles"]), where
        cmp = postpass.compare_level1(L1, pts, tris, corner, reach=0)
        assert not cmp["missing"] and not cmp["extra"] and not cmp["winding"], where
        assert cmp["excused_rows"] == 0, where


def test_open_mesh_of_separate_triangles():
    """17 000 triangles without a common edge: three distinct edges per triangle, the worst case of the orientation stage's edge
    table (cx_post.hip, cxp_edge_table_size) -- every triangle its own component, nothing lost"""
    from contourist_amd import _ffi
    nt = 17000
    rng = np.random.RandomState(2)
    base = rng.rand(nt, 1, 3) * 200.0 + 10.0
    pts = (base + np.array([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.25], [0.0, 1.0, 0.5]]])).reshape(-1, 3)
    tris = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
    ctx = _ffi.Context(0)
    post = ctx.postprocess3d_mesh(pts, tris, [255, 255, 255])
    assert post["n_triangles"] == nt and post["n_vertices"] == 3 * nt and post["n_components"] == nt
    p1, t1 = ctx.download_level1(post)
    # every triangle wound so that its normal_x is positive (each is its own component: surface_geometry.py:99-103)
    n = np.cross(p1[t1[:, 1]] - p1[t1[:, 0]], p1[t1[:, 2]] - p1[t1[:, 0]])
    assert np.all(n[:, 0] > 0)


def _level1_digest(extra_env):
    """Level 1 of a 72^3 noise field in a child process (debug knobs are only read by a process started with CX_DEBUG=1):
    -> (n_vertices, n_triangles, n_components, sha1 of the oriented triangles written as edge ids)"""
    import hashlib  # noqa: F401
    import subprocess
    import sys
    code = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from contourist_amd import _ffi
from test_gpu_sharded_level1 import canon
A = np.random.RandomState(11).rand(72, 72, 72).astype(np.float32)
ctx = _ffi.Context(0)
ctx.upload_grid(A)
ctx.extract3d(0.5, 1)
post = ctx.postprocess3d(0)
pts, tris = ctx.download_level1(post)
keys = ctx.download_level1_keys(post).astype(np.int64)
print("DIGEST", post["n_vertices"], post["n_triangles"], post["n_components"], hashlib.sha1(canon(keys, tris).tobytes()).hexdigest())
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("DIGEST")]
    assert out.returncode == 0 and lines, out.stderr.decode()[-2000:]
    return lines[0].split()[1:]


def test_linking_paths_agree():
    """the orientation stage of the march's own meshes links most edges inside blocks of triangles and sends the rest through a
    global table that is first sized small (cx_post.hip, cxp_k_edges_block).  The same mesh must come out when that table is too
    small and the stage is repeated with the full one (CX_EDGE_TABLE_TINY), with the full table at once (CX_EDGE_TABLE_FULL), and
    with every edge through the global table as for a caller's mesh (CX_LINK_GLOBAL): same components, same winding."""
    base = _level1_digest({})
    assert int(base[1]) > 100000 and int(base[2]) >= 1
    for knob in ("CX_EDGE_TABLE_TINY", "CX_EDGE_TABLE_FULL", "CX_LINK_GLOBAL"):
        assert _level1_digest({"CX_DEBUG": "1", knob: "1"}) == base, knob


def test_level1_device_export_equals_download():
    """cx_level1_device_ptrs / get_points_and_triangles(device=True): the Level-1 mesh as torch tensors ON THE GPU is, bit for
    bit, the mesh cx_level1_download brings to the host (tetrahedral.py:83-87, 528-552 without the trip)."""
    torch = pytest.importorskip("torch")
    from contourist_amd import tetrahedral

    def f(x, y, z):
        return np.sin(3.1 * x) * np.cos(2.3 * y) + 0.7 * z * z - 0.35 * x * y
    S = tetrahedral.TriangulatedIsosurfaces([-1.0, -1.2, -0.9], [1.1, 1.0, 1.2], [0.05, 0.055, 0.06], f, 0.21, [])
    S.search_for_endpoints()
    pts_h, tris_h = S.get_points_and_triangles()
    pts_d, tris_d = S.get_points_and_triangles(device=True)
    assert pts_d.is_cuda and tris_d.is_cuda and pts_d.dtype == torch.float64 and tris_d.dtype == torch.int32
    assert tuple(pts_d.shape) == tuple(np.asarray(pts_h).shape) and len(tris_h) > 1000
    assert np.array_equal(pts_d.cpu().numpy().view(np.uint64), np.asarray(pts_h, dtype=np.float64).view(np.uint64))
    td = tris_d.cpu().numpy()
    assert np.array_equal(td[np.lexsort((td[:, 2], td[:, 1], td[:, 0]))], np.asarray(tris_h))
    # grid coordinates from the maker, and the raw pointers / counts of the C ABI
    gp_h, gt_h = S.contour_maker.get_points_and_triangles()
    gp_d, gt_d = S.contour_maker.get_points_and_triangles(device=True)
    assert np.array_equal(gp_d.cpu().numpy().view(np.uint64), np.asarray(gp_h, dtype=np.float64).view(np.uint64))
    pp, tp, nv, nt = S.contour_maker.context().level1_device_ptrs()
    assert pp and tp and nv == len(gp_h) and nt == len(gt_h)
    view_p, view_t = S.contour_maker.context().level1_torch(copy=False)
    assert view_p.data_ptr() == pp and view_t.data_ptr() == tp and torch.equal(view_p, gp_d)


def test_level1_equals_oracle_at_mid_size():
    """a smooth field of 72 x 80 x 88 samples (the bench field's generator, 260-350 k triangles, several components): the triangles of one
    edge lie in different 512-triangle blocks of the linking kernel, the global edge table and the component passes are in play --
    weld / tiny collapse / clean / orient == the oracle's canonical pipeline: same counts after every stage, the same triangles as
    weld-bucket triples, the same windings (tetrahedral.py:190-215, 353-375, surface_geometry.py:14-140)"""
    from contourist_amd import _ffi, synthetic
    from oracle import level0, postpass
    shape = (72, 80, 88)
    A = synthetic.smooth_noise_host(shape, 4321, 180)
    ctx = _ffi.Context(0)
    try:
        for v in (0.0, 0.6):
            ctx.upload_grid(A)
            c = ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310)
            post = ctx.postprocess3d(0)
            pts, tris = ctx.download_level1(post)
            O = level0.march3d(A, v, diag_mode=1)
            assert c["n_triangles"] == len(O["tris"]) and len(O["tris"]) > 20000
            corner = np.array(shape) - 1
            ko = level0.edge_keys_from_pairs(O["pairs"], shape)
            L1 = postpass.level1_from_level0(ko, O["xyz"], O["tris"], corner)
            assert post["n_after_weld"] == L1["n_after_weld"] and post["n_after_tiny"] == L1["n_after_tiny"], v
            assert len(tris) == len(L1["triangles"]), v
            cmp = postpass.compare_level1(L1, pts, tris, corner, reach=0)
            assert not cmp["missing"] and not cmp["extra"] and not cmp["winding"], (v, cmp)
    finally:
        ctx.close()
