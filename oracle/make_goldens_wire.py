#!/usr/bin/env python3
"""oracle/make_goldens_wire.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Golden BYTES of the reference's two wire formats, written by the REAL reference's own writers:
  * MorphTriangles.to_json (contourist/morph_geometry.py:91-125) on the morph triangles the reference produced for the
    4-D fixtures (tests/golden4d/*.npz: mt_points4d / mt_segments / mt_triangles), whole range and a clipped range;
  * html_demo.emit_three_json (contourist/html_demo.py:147-161) on the reference's own Level-1 meshes of two 3-D
    fixtures (tests/golden/*.npz: l1_points / l1_triangles).
Output: tests/golden_wire/*.txt.gz (the exact strings, gzip-compressed; the largest 4-D fixtures are left out to keep the
repository small) -- data, not code."""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden_wire")


class _Mesh(object):
    "what emit_three_json asks of a grid contour"

    def __init__(self, points, triangles):
        self.points, self.triangles = points, triangles

    def get_points_and_triangles(self):
        return (self.points, self.triangles)


def main():
    from oracle import make_goldens4d
    make_goldens4d.reference_modules4d()            # translated copy of the package on sys.path, numpy aliases set
    from contourist import morph_geometry, html_demo
    os.makedirs(OUT, exist_ok=True)
    g4 = os.path.join(ROOT, "tests", "golden4d")
    for name in sorted(os.listdir(g4)):
        if not name.endswith(".npz"):
            continue
        G = np.load(os.path.join(g4, name))
        if "mt_points4d" not in G or name[:-4] not in ("two_blobs_seeded_12x12x12x7", "paraboloid_11x11x11x9"):
            continue
        MT = morph_geometry.MorphTriangles(G["mt_points4d"], [tuple(int(x) for x in s) for s in G["mt_segments"]],
                                           [tuple(int(x) for x in t) for t in G["mt_triangles"]])
        lo, hi = float(MT.min_value), float(MT.max_value)
        for tag, kw in (("all", {}), ("clipped", dict(min_value=lo + 0.25 * (hi - lo), max_value=hi - 0.125 * (hi - lo), maxint=4095))):
            if tag == "all" and name.startswith("paraboloid"):
                continue
            text = MT.to_json(**kw)
            with gzip.GzipFile(os.path.join(OUT, "%s.to_json.%s.txt.gz" % (name[:-4], tag)), "wb", mtime=0) as f:
                f.write(text.encode("ascii"))
        print(name, len(text))
    g3 = os.path.join(ROOT, "tests", "golden")
    for name in ("two_dots", "tiny_amp16", "inv_sphere20"):
        G = np.load(os.path.join(g3, name + ".npz"))
        pts = [np.array(p, dtype=np.float64) for p in G["l1_points"]]
        tris = [tuple(int(x) for x in t) for t in G["l1_triangles"]]
        text = html_demo.emit_three_json(_Mesh(pts, tris))
        with gzip.GzipFile(os.path.join(OUT, "%s.three.json.txt.gz" % name), "wb", mtime=0) as f:
            f.write(text.encode("ascii"))
        print(name, len(text))


if __name__ == "__main__":
    main()
