"""GPU: binary mesh files written straight from the Level-1 device buffers (cx_level1_write) against the host writers fed with
the downloaded arrays: the PLY is byte for byte what mesh_io.write_ply writes (faces in device order), the glTF payload holds the
float32 world positions and the indices, with the accessor bounds of the positions as written."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["sphere32", "noise24_v0"])
def test_device_fed_ply_and_gltf(name, tmp_path):
    from contourist_amd import tetrahedral, mesh_io
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    S = tetrahedral.TriangulatedIsosurfaces(G["mins"], None, G["delta"], G["A"], float(G["value"]), [])
    S.search_for_endpoints()
    p_dev = str(tmp_path / "device.ply")
    info = mesh_io.write_ply_device(S, p_dev)
    # the same mesh through the array API, faces as the device holds them
    ctx = S.contour_maker.context()
    pts_grid, tris_dev = ctx.download_level1(S.contour_maker._post)
    world = S.grid.from_grid_coordinates(pts_grid)
    p_host = str(tmp_path / "host.ply")
    mesh_io.write_ply(p_host, world, tris_dev)
    assert open(p_dev, "rb").read() == open(p_host, "rb").read()
    # what the writer reports: counts, bytes and the bounds of the positions (both formats)
    assert info["n_vertices"] == len(world) and info["n_triangles"] == len(tris_dev) and info["bytes"] == os.path.getsize(p_dev)
    assert np.array_equal(info["min"], world.min(axis=0)) and np.array_equal(info["max"], world.max(axis=0))
    # a path that cannot be written: an error, and no partial file left behind
    from contourist_amd import _ffi
    bad = str(tmp_path / "no_such_dir" / "x.ply")
    with pytest.raises(_ffi.CxError):
        ctx.write_level1(bad, "ply", S.grid.mins, S.grid.delta)
    assert not os.path.exists(bad)
    P, T = mesh_io.read_ply(p_dev)
    pts_api, tris_api = S.get_points_and_triangles()
    assert np.array_equal(P, np.asarray(pts_api)) and len(T) == len(tris_api)
    assert sorted(map(tuple, T.tolist())) == sorted(map(tuple, np.asarray(tris_api).tolist()))
    # glTF
    g_dev = str(tmp_path / "device.gltf")
    mesh_io.write_gltf_device(S, g_dev)
    doc = json.load(open(g_dev))
    blob = open(str(tmp_path / "device.bin"), "rb").read()
    nv, nt = len(world), len(tris_dev)
    assert doc["buffers"][0]["byteLength"] == len(blob) == nv * 12 + nt * 12
    pos = np.frombuffer(blob[:nv * 12], dtype="<f4").reshape(nv, 3)
    idx = np.frombuffer(blob[nv * 12:], dtype="<u4").reshape(nt, 3)
    assert np.array_equal(pos, world.astype(np.float32)) and np.array_equal(idx, tris_dev.astype(np.uint32))
    assert np.allclose(doc["accessors"][0]["min"], pos.min(axis=0)) and np.allclose(doc["accessors"][0]["max"], pos.max(axis=0))
    assert doc["accessors"][1]["count"] == nt * 3


def test_large_mesh_in_several_chunks(tmp_path):
    "more than one staging chunk per section (1 M records): 160^3 noise -> ~2.5 M triangles"
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi, synthetic, mesh_io
    A = synthetic.smooth_noise_torch((160, 160, 160), 7, 60, torch.device("cuda", 0))
    ctx = _ffi.Context(0)
    try:
        ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
        c = ctx.extract3d(0.0, 1)
        post = ctx.postprocess3d(0)
        assert post["n_triangles"] > 1 << 20
        path = str(tmp_path / "big.ply")
        info = ctx.write_level1(path, "ply", [0.5, -1.0, 2.0], [0.25, 0.5, 0.125])
        pts, tris = ctx.download_level1(post)
        P, T = mesh_io.read_ply(path)
        assert info["n_vertices"] == len(pts) and info["n_triangles"] == len(tris) and info["bytes"] == os.path.getsize(path)
        assert np.array_equal(P, pts * np.array([0.25, 0.5, 0.125]) + np.array([0.5, -1.0, 2.0])) and np.array_equal(T, tris)
    finally:
        ctx.close()
