#!/usr/bin/env python3
"""oracle/make_goldens_py2_demos.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

The reference COMMITTED outputs of its own demo calls: misc/centered.js (html_demo.emit_three_json of test_json /
test_centered, html_demo.py:147-161, 231-238) and misc/sphere.html, torus.html, wave.html, hyperbola.html
(html_demo.grid_html_page of test_sphere / test_torus / test_wave / test_hyperbola, html_demo.py:240-282), written by the
Python-2 code of its time.  This script reads the NUMBERS out of those files -- the vertex coordinates and the index
triples -- into tests/golden_demos/py2_<name>.npz: an independent pin of the Level-1 API next to the fixtures the present
checkout produces under Python 3 (make_goldens_demos.py).  Nothing of the files' markup or script text is stored.

Revision drift is printed (and asserted by tests/test_demo_outputs.py): where the committed output and today's checkout
differ, the difference is the reference's own (hash order of Python 2 vs 3 sets, later edits of the post-pass), not ours."""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/misc"
OUT = os.path.normpath(os.path.join(HERE, "..", "tests", "golden_demos"))


def parse_three_json(text):
    "emit_three_json: {'faces': [0, a, b, c, 0, ...], 'vertices': [x, y, z, ...]}"
    d = json.loads(text)
    f = np.array(d["faces"], dtype=np.int64).reshape(-1, 4)
    assert np.all(f[:, 0] == 0)
    return np.array(d["vertices"], dtype=np.float64).reshape(-1, 3), f[:, 1:].copy()


def parse_html(text):
    "grid_html_page: var vertices = [[x, y, z], ...]; var indices = [[a, b, c], ...];"
    mv = re.search(r"var\s+vertices\s*=\s*(\[\[.*?\]\])\s*;", text, re.S)
    mi = re.search(r"var\s+indices\s*=\s*(\[\[.*?\]\])\s*;", text, re.S)
    assert mv and mi
    return np.array(json.loads(mv.group(1)), dtype=np.float64).reshape(-1, 3), np.array(json.loads(mi.group(1)), dtype=np.int64).reshape(-1, 3)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fname, parser in (("centered", "centered.js", parse_three_json), ("sphere", "sphere.html", parse_html),
                                ("torus", "torus.html", parse_html), ("wave", "wave.html", parse_html),
                                ("hyperbola", "hyperbola.html", parse_html)):
        pts, tris = parser(open(os.path.join(REF, fname)).read())
        assert tris.min() >= 0 and tris.max() < len(pts)
        np.savez_compressed(os.path.join(OUT, "py2_" + name + ".npz"), points=pts, triangles=tris.astype(np.int32))
        line = "%-10s committed by the reference: %6d points %6d triangles" % (name, len(pts), len(tris))
        now = os.path.join(OUT, name + ".npz")
        if os.path.exists(now):
            G = np.load(now)
            line += " | today's checkout (Python 3): %6d points %6d triangles" % (len(G["points"]), len(G["triangles"]))
        print(line, flush=True)


if __name__ == "__main__":
    main()
