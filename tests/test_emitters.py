"""CPU: three.js emitters (host-side formatting, no device needed)"""
import json

import numpy as np


def test_emit_three_json_roundtrip():
    from contourist_amd import html_demo
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    tris = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    D = json.loads(html_demo.emit_three_json((pts, tris)))
    assert D["faces"] == [0, 0, 1, 2, 0, 0, 2, 3]
    assert D["vertices"] == pts.reshape(-1).tolist()
    page = html_demo.grid_html_page((pts, tris), title="t")
    assert "THREE.Face3" in page and "[0, 1, 2]" in page


def test_ply_and_gltf_writers_round_trip(tmp_path):
    import json
    from contourist_amd import mesh_io
    rng = np.random.RandomState(3)
    P = rng.standard_normal((17, 3))
    T = rng.randint(0, 17, size=(29, 3)).astype(np.int32)
    path = mesh_io.write_ply(str(tmp_path / "m.ply"), P, T)
    P2, T2 = mesh_io.read_ply(path)
    assert np.array_equal(P, P2) and np.array_equal(T, T2)
    g = mesh_io.write_gltf_bin(str(tmp_path / "m.gltf"), P, T)
    doc = json.load(open(g))
    blob = open(str(tmp_path / "m.bin"), "rb").read()
    assert doc["buffers"][0]["byteLength"] == len(blob) == 17 * 12 + 29 * 3 * 4
    pos = np.frombuffer(blob[:17 * 12], dtype="<f4").reshape(17, 3)
    idx = np.frombuffer(blob[17 * 12:], dtype="<u4").reshape(29, 3)
    assert np.allclose(pos, P.astype(np.float32)) and np.array_equal(idx, T)
