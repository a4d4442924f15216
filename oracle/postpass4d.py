"""oracle/postpass4d.py -- TEST INFRASTRUCTURE ONLY.

Restatement of GridContour4D.find_tetrahedra's post-steps on a Level-0 4-D mesh:
    bin_times(nbins=100)              contourist/pentatopes.py:162-169
    drop_instant_tetrahedra(1e-7)     contourist/pentatopes.py:171-189
    remove_tiny_simplices(1e-3)       contourist/tetrahedral.py:353-375  (called at pentatopes.py:125)
Canonical choice for the tiny collapse as in oracle/postpass.py (group root = smallest edge key).
"""
import numpy as np

from .postpass import UnionFind


def bin_times(xyzt, corner, nbins=100):
    xyzt = np.array(xyzt, dtype=np.float64)
    min_interval = corner[-1] * (1.0 / nbins)
    bins = np.trunc(xyzt[:, 3] / min_interval)       # int(tvalue / min_interval)
    xyzt[:, 3] = bins * min_interval
    return xyzt


def drop_instant(xyzt, tets, epsilon=1e-7):
    tets = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    if len(tets) == 0:
        return tets
    T = xyzt[tets][:, :, 3]
    return tets[~((T.max(axis=1) - T.min(axis=1)) < epsilon)]


def tiny_collapse4(keys, xyzt, tets, corner, epsilon=1e-3):
    keys = np.asarray(keys, dtype=np.int64)
    xyzt = np.array(xyzt, dtype=np.float64)
    tets = np.asarray(tets, dtype=np.int64).reshape(-1, 4)
    if len(tets) == 0:
        return xyzt, tets
    P = xyzt[tets]
    delta = (P.max(axis=1) - P.min(axis=1)) * (1.0 / np.asarray(corner, dtype=np.float64))
    tiny = delta.max(axis=1) < epsilon
    if tiny.any():
        order = np.argsort(keys, kind="stable")
        rank = np.empty(len(keys), dtype=np.int64)
        rank[order] = np.arange(len(keys))
        uf = UnionFind(len(keys))
        for t in tets[tiny]:
            r = rank[t]
            for s in (1, 2, 3):
                uf.union(int(r[0]), int(r[s]))
        for v in np.unique(tets[tiny].reshape(-1)):
            xyzt[v] = xyzt[order[uf.find(int(rank[v]))]]
    return xyzt, tets[~tiny]


def find_tetrahedra_post(keys, xyzt, tets, corner):
    "returns dict(xyzt, tets, n_after_drop, n_after_tiny)"
    x1 = bin_times(xyzt, corner)
    t1 = drop_instant(x1, tets)
    x2, t2 = tiny_collapse4(keys, x1, t1, corner)
    return dict(xyzt_binned=x1, xyzt=x2, tets=t2, n_after_drop=len(t1), n_after_tiny=len(t2))
