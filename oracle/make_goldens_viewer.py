#!/usr/bin/env python3
"""oracle/make_goldens_viewer.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference and `node` exist.

Pins oracle/morph_eval.py on the reference's OWN viewer code: misc/morph_triangles.js builds the surface at
a time t from the morph-triangle JSON (MorphTriangles.to_json).  The file as a whole needs three.js and a browser; its
arithmetic does not.  At generation time this script reads the file where it lies, cuts out -- by line range, each range
checked against anchor texts so that a different revision fails loudly --
   A  lines 6-12     the time range and the shift / scale of the data
   B  lines 14-105   positions (shifted, scaled), epsilon, segments, triangles, the per-triangle intervals [tr_min, tr_max]
                     (:53-84), their order (:86-88), the interval variables
   C  lines 109-149  the body of start_transition() up to the geometry: the active triangles and the interval [min_t, max_t]
   D  lines 156-178  interpolate_points_3d
and runs those lines UNCHANGED under node inside a small driver (written to a temporary directory outside the repository)
that feeds them the JSON of the wire fixtures (tests/golden_wire/*.to_json.*.txt.gz: bytes written by the real reference),
sets the viewer's clock to a list of times and prints what the lines computed.  The driver's own part is the loop over the
active triangles' segments in order of first use (the reference does that inside THREE.Geometry calls, :179-204).
Output: tests/golden_viewer/viewer_<fixture>.npz -- numbers only."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/misc/morph_triangles.js"
WIRE = os.path.normpath(os.path.join(HERE, "..", "tests", "golden_wire"))
OUT = os.path.normpath(os.path.join(HERE, "..", "tests", "golden_viewer"))

SLICES = {   # name: (first line, last line, text the first line must contain, text the last line must contain)
    "A": (6, 12, 'var max_value = morph_triangle_data["max_value"]', "var ticking = false"),
    "B": (14, 105, "var unflatten_list = function", "var transition_start = null"),
    "C": (109, 149, "if (current_t + epsilon > max_value)", "}"),
    "D": (156, 178, "var interpolate_points_3d = function(p_early, p_late, t_value)", "}"),
}

DRIVER = """
var fs = require('fs');
var morph_triangle_data = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
var times = JSON.parse(process.argv[3]);
var duration = 1.0;
%(A)s
%(B)s
%(D)s
var scan = function() {
%(C)s
    return active_triangles;
};
var out = {epsilon: epsilon, min_value: min_value, max_value: max_value, positions: positions,
           triangle_order: triangle_order, triangle_max: triangle_max, evals: []};
for (var n = 0; n < times.length; n++) {
    current_t = times[n] - epsilon;          // start_transition() then evaluates at min_t = current_t + epsilon
    var active = scan();
    var seen = {}, seg_ids = [], start_points = [], end_points = [], faces = [];
    for (var i = 0; i < active.length; i++) {
        var tsegments = triangles[active[i]];
        var face = [];
        for (var j = 0; j < tsegments.length; j++) {
            var index = tsegments[j];
            if (!(index in seen)) {
                seen[index] = seg_ids.length;
                seg_ids.push(index);
                var segment = segments[index];
                start_points.push(interpolate_points_3d(positions[segment[0]], positions[segment[1]], min_t).slice(0, 3));
                end_points.push(interpolate_points_3d(positions[segment[0]], positions[segment[1]], max_t).slice(0, 3));
            }
            face.push(seen[index]);
        }
        faces.push(face);
    }
    out.evals.push({t: times[n], min_t: min_t, max_t: max_t, active: active, segment_ids: seg_ids,
                    start_points: start_points, end_points: end_points, faces: faces});
}
process.stdout.write(JSON.stringify(out));
"""


def slices():
    lines = open(SRC).read().split("\n")
    out = {}
    for name, (a, b, first, last) in SLICES.items():
        assert first in lines[a - 1], (name, lines[a - 1])
        assert last in lines[b - 1], (name, lines[b - 1])
        out[name] = "\n".join(lines[a - 1:b])
    return out


def main():
    S = slices()
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="cx_viewer_") as tmp:
        drv = os.path.join(tmp, "driver.js")
        open(drv, "w").write(DRIVER % S)
        for fname in sorted(os.listdir(WIRE)):
            if ".to_json." not in fname:
                continue
            data = json.loads(gzip.open(os.path.join(WIRE, fname)).read().decode("utf8"))
            jpath = os.path.join(tmp, "data.json")
            json.dump(data, open(jpath, "w"))
            lo, hi = float(data["min_value"]), float(data["max_value"])
            times = [lo + (hi - lo) * f for f in (0.003, 0.11, 0.26, 0.41, 0.5, 0.63, 0.77, 0.93)]
            r = subprocess.run(["node", drv, jpath, json.dumps(times)], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            R = json.loads(r.stdout)
            nt = len(data["triangles"]) // 3
            tr_min = np.full(nt, np.nan)
            tr_max = np.full(nt, np.nan)
            # triangle_order is sorted by tr_min; triangle_max was pushed in triangle order of the KEPT triangles
            kept = sorted(i for _, i in R["triangle_order"])
            for i, tmax in zip(kept, R["triangle_max"]):
                tr_max[i] = tmax
            for tmin, i in R["triangle_order"]:
                tr_min[i] = tmin
            arrays = dict(epsilon=R["epsilon"], min_value=R["min_value"], max_value=R["max_value"],
                          positions=np.array(R["positions"], dtype=np.float64), tr_min=tr_min, tr_max=tr_max,
                          order=np.array([i for _, i in R["triangle_order"]], dtype=np.int64), times=np.array(times))
            for n, E in enumerate(R["evals"]):
                arrays["min_t_%d" % n] = E["min_t"]; arrays["max_t_%d" % n] = E["max_t"]
                arrays["active_%d" % n] = np.array(E["active"], dtype=np.int64)
                arrays["segment_ids_%d" % n] = np.array(E["segment_ids"], dtype=np.int64)
                arrays["start_points_%d" % n] = np.array(E["start_points"], dtype=np.float64).reshape(-1, 3)
                arrays["end_points_%d" % n] = np.array(E["end_points"], dtype=np.float64).reshape(-1, 3)
                arrays["faces_%d" % n] = np.array(E["faces"], dtype=np.int64).reshape(-1, 3)
            name = fname.replace(".txt.gz", "").replace(".to_json.", "_")
            np.savez_compressed(os.path.join(OUT, "viewer_" + name + ".npz"), **arrays)
            print("%-44s %6d triangles, %5d kept; active at the 8 times: %s" % (name, nt, len(kept), [len(E["active"]) for E in R["evals"]]), flush=True)


if __name__ == "__main__":
    main()
