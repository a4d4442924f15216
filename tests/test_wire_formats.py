"""The two wire formats of the path, against the BYTES the real reference wrote (tests/golden_wire/*.txt.gz, made by
oracle/make_goldens_wire.py from the reference's own writers): MorphTriangles.to_json (morph_geometry.py:91-125) on the
reference's morph triangles of the 4-D fixtures, and html_demo.emit_three_json (html_demo.py:147-161) on its Level-1 meshes."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT

WIRE = os.path.join(ROOT, "tests", "golden_wire")
G4 = os.path.join(ROOT, "tests", "golden4d")


def golden_text(name):
    with gzip.open(os.path.join(WIRE, name), "rb") as f:
        return f.read().decode("ascii")


@pytest.mark.parametrize("name,tag", [("two_blobs_seeded_12x12x12x7", "all"), ("two_blobs_seeded_12x12x12x7", "clipped"),
                                      ("paraboloid_11x11x11x9", "clipped")])
def test_to_json_bytes(name, tag):
    from contourist_amd import morph_geometry
    G = np.load(os.path.join(G4, name + ".npz"))
    MT = morph_geometry.MorphTriangles(G["mt_points4d"], G["mt_segments"], G["mt_triangles"])
    lo, hi = float(MT.min_value), float(MT.max_value)
    kw = {} if tag == "all" else dict(min_value=lo + 0.25 * (hi - lo), max_value=hi - 0.125 * (hi - lo), maxint=4095)
    assert MT.to_json(**kw) == golden_text("%s.to_json.%s.txt.gz" % (name, tag))


@pytest.mark.parametrize("name", ["two_dots", "tiny_amp16", "inv_sphere20"])
def test_emit_three_json_bytes(name):
    from contourist_amd import html_demo
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    pts = [np.array(p, dtype=np.float64) for p in G["l1_points"]]
    tris = [tuple(int(x) for x in t) for t in G["l1_triangles"]]
    want = golden_text("%s.three.json.txt.gz" % name)
    assert html_demo.emit_three_json((pts, tris)) == want
    # and with the arrays the device path hands back ((V,3) float64 / (T,3) int32)
    P = np.asarray(G["l1_points"], dtype=np.float64).reshape(-1, 3)
    T = np.asarray(G["l1_triangles"], dtype=np.int32).reshape(-1, 3)
    assert html_demo.emit_three_json((P, T)) == want
