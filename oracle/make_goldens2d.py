#!/usr/bin/env python3
"""oracle/make_goldens2d.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

2-D contour goldens from the REAL reference (contourist/triangulated.py, multiple_2d_contour.py).
field2d.py is Python-2 source (implicit relative import), so the package is copied to a temp dir OUTSIDE
the repo, translated there with lib2to3 and imported from there; only numbers are
written to tests/golden2d/*.npz: the sample array, the isovalues, the end points and the polylines the
reference returned.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.normpath(os.path.join(HERE, "..", "tests", "golden2d"))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..")))
_TMP = [None]


def reference_modules2d():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference not present")
    if _TMP[0] is None:
        tmp = tempfile.mkdtemp(prefix="contourist_ref2d_")
        shutil.copytree(os.path.join(REFERENCE, "contourist"), os.path.join(tmp, "contourist"))
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "field2d.py", "pentatopes.py", "morph_geometry.py",
                               "html_demo.py", "lasso.py"], cwd=os.path.join(tmp, "contourist"),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _TMP[0] = tmp
    np.int = int
    np.float = float
    np.sometrue = np.any
    sys.dont_write_bytecode = True
    if _TMP[0] not in sys.path:
        sys.path.insert(0, _TMP[0])
    from contourist import triangulated, field2d, multiple_2d_contour
    triangulated.field2d = field2d   # triangulated.DxDy2DContour names field2d without importing it (:140)
    return triangulated, field2d, multiple_2d_contour


def smooth2(shape, seed, passes):
    rng = np.random.RandomState(seed)
    B = rng.standard_normal(shape)
    for _ in range(passes):
        for ax in range(2):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    return (B / B.std()).astype(np.float32)


def fields2d():
    F = {}
    I, J = np.meshgrid(np.arange(24.0), np.arange(20.0), indexing="ij")
    F["circle_24x20"] = dict(A=((I - 11.3) ** 2 + (J - 9.6) ** 2).astype(np.float32), values=[30.0])
    x = -1.0 + 0.2 * np.arange(11)
    X, Y = np.meshgrid(x, x, indexing="ij")
    F["demo_11x11"] = dict(A=(X * X + Y * (Y + 1) * (Y - 1) - np.sin(2 * Y * Y + 4 * X)).astype(np.float32), values=[0.2])
    F["noise_40x33"] = dict(A=smooth2((40, 33), 11, 2), values=[0.1])
    F["noise_levels_48x37"] = dict(A=smooth2((48, 37), 21, 3), values=[-1.1, -0.4, 0.0, 0.35, 0.9, 1.7])
    # (a field FULL of samples equal to the isovalue -- e.g. integers contoured at an integer -- makes the reference's
    # result depend on its set iteration order and is not a fixture; Percentile2DContour's levels ARE samples, one
    # tie per level, and that is covered below)
    P = smooth2((36, 31), 14, 3)
    srt = np.sort(P.astype(np.float64).flatten())
    skip = int(srt.size / 5)
    F["percentile_36x31"] = dict(A=P, values=[float(srt[k]) for k in range(skip, srt.size, skip)])
    # the last lattice point alone on its side: its three pairs are not found by the reference's grid search
    C = np.zeros((9, 8), dtype=np.float32)
    C[8, 7] = 2.0
    C[2:4, 2:5] = 3.0
    F["corner_9x8"] = dict(A=C, values=[1.0])
    # two separate loops, seeded growth reaches one of them
    S = np.minimum((I - 6.2) ** 2 + (J - 6.4) ** 2, (I - 16.7) ** 2 + (J - 12.3) ** 2).astype(np.float32)
    F["seeded_two_loops_24x20"] = dict(A=S, values=[9.0], end_points=[[[6, 6], [6, 12]]])
    F["demo_world_11x11"] = dict(A=F["demo_11x11"]["A"], values=[0.2], world=[-1.0, -1.0, 0.2, 0.2])
    F["seeded_far_24x20"] = dict(A=S, values=[9.0], end_points=[[[17, 12], [2, 19]], [[16, 13], [23, 13]]])
    return F


def run_reference2d(A, values, end_points=None, world=None):
    """[(value, [(closed, points[k,2]), ...]), ...] in grid coordinates (unit delta, zero mins)"""
    triangulated, field2d, multiple = reference_modules2d()
    A = np.ascontiguousarray(A, dtype=np.float32)
    n, m = A.shape

    def f(i, j):
        return float(A[int(i), int(j)])
    out = []
    if world is not None:
        # DxDy2DContourGrid over a FunctionGrid: world coordinates (triangulated.py:121-138)
        x0, y0, dx, dy = world
        grid = field2d.Function2DGrid(x0, y0, x0 + dx * (n - 1 + 0.25), y0 + dy * (m - 1 + 0.25), dx, dy,
                                      lambda x, y: f(round((x - x0) / dx), round((y - y0) / dy)))
        assert tuple(grid.grid_dimensions) == (n, m)
        for v in values:
            C = triangulated.DxDy2DContourGrid(grid, v)
            out.append((v, [(bool(c), np.array(p, dtype=np.float64)) for c, p in C.get_contour_sequences()]))
        return out
    if end_points is not None:
        for v in values:
            G = triangulated.Grid2DContour(n, m, f, v, [[tuple(a), tuple(b)] for a, b in end_points])
            out.append((v, [(bool(c), np.array(p, dtype=np.float64)) for c, p in G.get_contour_sequences()]))
        return out
    if len(values) == 1:
        G = triangulated.Grid2DContour(n, m, f, values[0], None)
        return [(values[0], [(bool(c), np.array(p, dtype=np.float64)) for c, p in G.get_contour_sequences()])]
    grid = field2d.Function2DGrid(0.0, 0.0, n - 1 + 0.25, m - 1 + 0.25, 1.0, 1.0, lambda x, y: f(round(x), round(y)))
    assert tuple(grid.grid_dimensions) == (n, m)
    M = multiple.Multiple2DContourGrid(grid, values)
    d = M.get_contours_dictionary()
    return [(v, [(bool(c), np.array(p, dtype=np.float64)) for c, p in d[v]]) for v in sorted(d)]


def pack(result):
    values, level, closed, offsets, pts = [], [], [], [0], []
    for li, (v, seqs) in enumerate(result):
        values.append(v)
        for c, p in seqs:
            level.append(li)
            closed.append(c)
            pts.append(p.reshape(-1, 2))
            offsets.append(offsets[-1] + len(p))
    return dict(values=np.array(values, dtype=np.float64), level=np.array(level, dtype=np.int32), closed=np.array(closed, dtype=np.uint8),
                offsets=np.array(offsets, dtype=np.int64), points=np.concatenate(pts) if pts else np.zeros((0, 2)))


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, spec in fields2d().items():
        res = run_reference2d(spec["A"], spec["values"], spec.get("end_points"), spec.get("world"))
        d = pack(res)
        d["A"] = spec["A"]
        if "world" in spec:
            d["world"] = np.array(spec["world"], dtype=np.float64)
        if "end_points" in spec:
            d["end_points"] = np.array(spec["end_points"], dtype=np.int32)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **d)
        print(name, [(v, len(s), sum(len(p) for _, p in s)) for v, s in res])


if __name__ == "__main__":
    main()
