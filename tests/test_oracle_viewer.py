"""CPU: oracle/morph_eval.py (the oracle cx_morph_eval is compared with on the GPU) against what the
reference's OWN viewer code computed -- misc/morph_triangles.js, lines cut out and run unchanged under node by
oracle/make_goldens_viewer.py on the bytes MorphTriangles.to_json wrote (tests/golden_wire), results in
tests/golden_viewer/viewer_*.npz.  Which lines of the viewer each assertion covers is stated at the assertion."""
import glob
import gzip
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = sorted(glob.glob(os.path.join(ROOT, "tests", "golden_viewer", "viewer_*.npz")))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[7:-4] for p in FIX])
def test_morph_eval_oracle_equals_the_reference_viewer(path):
    from oracle import morph_eval
    G = np.load(path)
    name = os.path.basename(path)[7:-4]
    stem, mode = name.rsplit("_", 1)
    data = json.loads(gzip.open(os.path.join(ROOT, "tests", "golden_wire", "%s.to_json.%s.txt.gz" % (stem, mode))).read().decode("utf8"))
    segments = np.array(data["segments"], dtype=np.int64).reshape(-1, 2)
    triangles = np.array(data["triangles"], dtype=np.int64).reshape(-1, 3)
    # morph_triangles.js:26-42: positions = shift + scale * position, per coordinate (the oracle takes them as its input)
    raw = np.array(data["positions"], dtype=np.float64).reshape(-1, 4)
    P = np.array(data["shift"], dtype=np.float64) + np.array(data["scale"], dtype=np.float64) * raw
    assert np.array_equal(P, G["positions"])
    # :49-50 epsilon = 1e-7 * (max_value - min_value)
    assert float(G["epsilon"]) == (float(data["max_value"]) - float(data["min_value"])) * 1.0 * 1e-7
    # :53-84 per-triangle interval: the common extent of the three segments, dropped if a segment has no time extent or the
    # extent is empty
    tr_min, tr_max, valid = morph_eval.triangle_intervals(P, segments, triangles)
    kept = ~np.isnan(G["tr_min"])
    assert np.array_equal(valid, kept)
    assert np.array_equal(tr_min[valid], G["tr_min"][kept]) and np.array_equal(tr_max[valid], G["tr_max"][kept])
    # :86-88 order of the kept triangles by tr_min (V8's sort is stable: ties stay in index order)
    order = [i for _, i in sorted((tr_min[i], i) for i in range(len(triangles)) if valid[i])]
    assert order == G["order"].tolist()
    for n, t in enumerate(G["times"].tolist()):
        S = morph_eval.surface_at(P, segments, triangles, t, float(data["min_value"]), float(data["max_value"]))
        # :109-149 start_transition: evaluated at min_t = current_t + epsilon (the driver sets current_t = t - epsilon); active =
        # triangles with tr_min <= min_t < tr_max, in tr_min order; the scan stops at the first triangle that starts later
        assert abs(float(G["min_t_%d" % n]) - t) <= 2 * float(G["epsilon"])
        assert S["active"] == G["active_%d" % n].tolist()
        # :179-204 one vertex per segment in order of first use; :156-178 interpolate_points_3d at min_t, bit for bit
        assert S["segment_ids"] == G["segment_ids_%d" % n].tolist()
        assert np.array_equal(S["faces"], G["faces_%d" % n])
        St = morph_eval.surface_at(P, segments, triangles, float(G["min_t_%d" % n]), float(data["min_value"]), float(data["max_value"]))
        assert St["active"] == S["active"]
        assert np.array_equal(St["points"], G["start_points_%d" % n])
        # the morph target of the interval (the same segments at max_t): interpolate_points_3d again, incl. its clamps
        eps = float(G["epsilon"])
        ends = np.array([morph_eval.interpolate_points_3d(P[segments[s][0]], P[segments[s][1]], float(G["max_t_%d" % n]), eps) for s in S["segment_ids"]])
        assert np.array_equal(ends.reshape(-1, 3), G["end_points_%d" % n])


def test_fixtures_present():
    assert len(FIX) >= 3
