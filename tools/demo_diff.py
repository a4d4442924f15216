"""GPU: which triangles of a reference demo (tests/golden_demos/<name>.npz) the device path does not reproduce, and where
they lie.  usage: python tools/demo_diff.py wave"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_demos import calls, GD   # noqa: E402
from contourist_amd import tetrahedral   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "wave"
G = np.load(os.path.join(GD, name + ".npz"))
make, side = calls(tetrahedral)[name]
obj = make()
if side is None:
    obj.search_for_endpoints()
pts, tris = obj.get_points_and_triangles()
pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def cents(P, T, nd=3):
    c = (P[T[:, 0]] + P[T[:, 1]] + P[T[:, 2]]) / 3.0
    return [tuple(r) for r in np.round(c, nd).tolist()]


a = collections.Counter(cents(G["points"], G["triangles"]))
b = collections.Counter(cents(pts, tris))
only_a = sorted((a - b).elements())
only_b = sorted((b - a).elements())
print(name, "reference", len(G["triangles"]), "device", len(tris), "reference-only", len(only_a), "device-only", len(only_b))
print("reference points", len(G["points"]), "device points", len(pts))
for c in only_a[:60]:
    print("  ref only", c)
for c in only_b[:60]:
    print("  dev only", c)
# vertex sets
pa = {tuple(np.round(p, 6)) for p in np.asarray(G["points"], dtype=np.float64)}
pb = {tuple(np.round(p, 6)) for p in pts}
print("points only in reference", len(pa - pb), "only in device", len(pb - pa))
for p in sorted(pa - pb)[:40]:
    print("  ref point", p)
for p in sorted(pb - pa)[:40]:
    print("  dev point", p)
