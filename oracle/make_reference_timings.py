"""TEST / BENCH INFRASTRUCTURE -- not part of the product path.

The real reference (pure Python, single thread) cannot travel to the GPU box, so its throughput is
recorded HERE, from the wall times `oracle/make_goldens.py` stored in every fixture when it ran the real
reference (`t_total_s` = search_for_endpoints() + get_points_and_triangles(), tetrahedral.py:74-87), next
to the C restatement (`oracle/march_oracle.c`, Level 0 only) timed on the same arrays on the same core.
bench.py quotes both and scales the port's rate on the GPU box's host by the ratio
(SURVEY.md 8(d) i-iii).  Output: tests/golden/reference_timings.json."""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import build as obuild, level0
    obuild.build()
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        d = np.load(path, allow_pickle=True)
        if "t_total_s" not in d or "A" not in d:
            continue
        A = np.ascontiguousarray(d["A"], dtype=np.float32)
        v = float(d["value"])
        level0.march3d(A, v, diag_mode=1)
        reps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            level0.march3d(A, v, diag_mode=1)
            reps += 1
        t_port = (time.perf_counter() - t0) / reps
        t_ref = float(d["t_total_s"])
        rows.append({"fixture": os.path.basename(path), "samples": int(A.size), "reference_s": t_ref, "port_s": t_port,
                     "reference_Mvoxels_s": A.size / t_ref / 1e6, "port_Mvoxels_s": A.size / t_port / 1e6,
                     "port_over_reference": t_ref / t_port})
    ref = float(np.median([r["reference_Mvoxels_s"] for r in rows]))
    ratio = float(np.median([r["port_over_reference"] for r in rows]))
    out = {"where": "build container, 1 core of an 8-core Intel Xeon @ 2.10 GHz, CPython 3.10.12, numpy 2.2.6",
           "what": "real reference: TriangulatedIsosurfaces(...).search_for_endpoints() + get_points_and_triangles() "
                   "(Level 0 + Level 1, single thread, pure Python); port: oracle/march_oracle.c Level 0, single thread",
           "reference_Mvoxels_s_median": ref, "port_over_reference_median": ratio, "fixtures": rows}
    with open(os.path.join(ROOT, "tests", "golden", "reference_timings.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("reference_Mvoxels_s_median", "port_over_reference_median")}))


if __name__ == "__main__":
    main()
