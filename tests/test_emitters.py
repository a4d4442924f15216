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
