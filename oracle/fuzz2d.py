#!/usr/bin/env python3
"""oracle/fuzz2d.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Random small fields and levels: polylines of the REAL reference (contourist/triangulated.py via make_goldens2d)
against the restatement oracle/contour2d.py under the build's rule.  Round 1: 40 fields, 88 levels, 87 identical;
the one difference is which of two np.allclose points of a polyline is dropped (it depends on the direction the
reference's walk happened to take, triangulated.py:268)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens2d as mg          # noqa: E402
from oracle import contour2d as o2   # noqa: E402


def main(nfields=40):
    bad, tot = [], 0
    for seed in range(nfields):
        rng = np.random.RandomState(100 + seed)
        n, m = rng.randint(6, 30), rng.randint(6, 30)
        A = mg.smooth2((n, m), 1000 + seed, rng.randint(0, 4))
        values = sorted(set(np.round(rng.uniform(-1.5, 1.5, size=rng.randint(1, 4)), 3).tolist()))
        for v, seqs in mg.run_reference2d(A, values):
            tot += 1
            mine = o2.contours(A, v, None, "build")
            if o2.canonical(seqs) != o2.canonical([(c, p) for c, p, _ in mine]):
                bad.append((seed, v))
    print("levels %d, identical %d, different %s" % (tot, tot - len(bad), bad))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
