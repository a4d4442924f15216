"""oracle/level0_4d.py -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of march4d_oracle.c + canonical forms."""
import ctypes

import numpy as np

from . import level0 as _l0


def _lib():
    L = _l0.lib()
    if not getattr(L, "_cx4d", False):
        i64 = ctypes.c_int64
        L.oracle_march4d.restype = ctypes.c_int
        L.oracle_march4d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p]
        L._cx4d = True
    return L


def march4d(A, value, diag_mode=1, origin=(0, 0, 0, 0), vcap=None, tcap=None):
    """Level-0 dense 4-D march.  returns dict(pairs (V,8) int32 [low ijkl, high ijkl], xyzt (V,4) float64,
    tets (T,4) int64 vertex indices, nborder, nborder_mixed)"""
    A = np.ascontiguousarray(A, dtype=np.float32)
    assert A.ndim == 4
    shape = np.array(A.shape, dtype=np.int64)
    org = np.array(origin, dtype=np.int64)
    vcap = vcap or max(4096, A.size // 2)
    tcap = tcap or 8 * vcap
    while True:
        pairs = np.zeros((vcap, 8), dtype=np.int32)
        xyzt = np.zeros((vcap, 4), dtype=np.float64)
        tets = np.zeros((tcap, 4), dtype=np.int64)
        counts = np.zeros(4, dtype=np.int64)
        rc = _lib().oracle_march4d(A.ctypes.data, shape.ctypes.data, float(value), int(diag_mode), org.ctypes.data,
                                   pairs.ctypes.data, xyzt.ctypes.data, vcap, tets.ctypes.data, tcap, counts.ctypes.data)
        if rc != 0:
            raise MemoryError("oracle_march4d")
        nv, nt, nb, nbm = (int(c) for c in counts)
        if nv <= vcap and nt <= tcap:
            return dict(pairs=pairs[:nv].copy(), xyzt=xyzt[:nv].copy(), tets=tets[:nt].copy(), nborder=nb, nborder_mixed=nbm)
        vcap = max(2 * vcap, nv + 1)
        tcap = max(2 * tcap, nt + 1)


def edge_keys4(pairs, shape):
    "unordered lattice edge -> int64 key = linear_index(lower endpoint)*16 + dir, dir = 8di+4dj+2dk+dl in 1..15"
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 8)
    if len(pairs) == 0:
        return np.zeros(0, dtype=np.int64)
    lo = np.minimum(pairs[:, :4], pairs[:, 4:])
    hi = np.maximum(pairs[:, :4], pairs[:, 4:])
    d = hi - lo
    assert d.min() >= 0 and d.max() <= 1
    lin = ((lo[:, 0] * shape[1] + lo[:, 1]) * shape[2] + lo[:, 2]) * shape[3] + lo[:, 3]
    return lin * 16 + (d[:, 0] * 8 + d[:, 1] * 4 + d[:, 2] * 2 + d[:, 3])


def canonical4(keys, xyzt, tets):
    "vertices sorted by key; tetrahedra as sorted rows of 4 keys, rows lexsorted"
    keys = np.asarray(keys, dtype=np.int64)
    order = np.argsort(keys, kind="stable")
    tk = np.sort(keys[np.asarray(tets, dtype=np.int64)], axis=1) if len(tets) else np.zeros((0, 4), np.int64)
    if len(tk):
        tk = tk[np.lexsort(tk.T[::-1])]
    return keys[order], np.asarray(xyzt)[order], tk


def pentatope_groups(tet_keys, shape):
    """comparison modulo the 2-3 split choice: group tetrahedra by the pentatope (5 lattice points) they came
    from; rows = sorted lattice points + sorted union of edge keys, padded with -1"""
    tet_keys = np.asarray(tet_keys, dtype=np.int64).reshape(-1, 4)
    groups = {}
    for row in tet_keys:
        pts = set()
        for key in row:
            lin, d = int(key) >> 4, int(key) & 15
            l = lin % shape[3]; r = lin // shape[3]
            k = r % shape[2]; r //= shape[2]
            j = r % shape[1]; i = r // shape[1]
            a = (i, j, k, l)
            b = (i + ((d >> 3) & 1), j + ((d >> 2) & 1), k + ((d >> 1) & 1), l + (d & 1))
            pts.add(a); pts.add(b)
        assert len(pts) == 5, "tetrahedron does not span a pentatope"
        groups.setdefault(tuple(sorted(pts)), set()).update(int(x) for x in row)
    out = []
    for pent, ks in groups.items():
        ks = sorted(ks)
        assert len(ks) in (4, 6), len(ks)
        out.append([c for p in pent for c in p] + ks + [-1] * (6 - len(ks)))
    return np.array(sorted(out), dtype=np.int64)
