"""oracle/level0.py -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper around march_oracle.c (Level-0 march) and order-independent canonical forms
used to compare the HIP path, this oracle and the reference goldens.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = _build.build()
        L = ctypes.CDLL(path)
        i64 = ctypes.c_int64
        L.oracle_march3d.restype = ctypes.c_int
        L.oracle_march3d.argtypes = [ctypes.c_void_p, i64, i64, i64, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, i64,
                                     ctypes.c_void_p]
        L.oracle_march3d_origin.restype = ctypes.c_int
        L.oracle_march3d_origin.argtypes = [ctypes.c_void_p, i64, i64, i64, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p]
        L.oracle_count_crossings3d.restype = i64
        L.oracle_count_crossings3d.argtypes = [ctypes.c_void_p, i64, i64, i64, ctypes.c_double,
                                               ctypes.c_void_p]
        L.oracle_py_tuplehash.restype = ctypes.c_uint64
        L.oracle_py_tuplehash.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_py_set8_slots.restype = None
        L.oracle_py_set8_slots.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _LIB = L
    return _LIB


def count_crossings(A, value):
    A = np.ascontiguousarray(A, dtype=np.float32)
    mm = np.zeros(2, dtype=np.float64)
    n = lib().oracle_count_crossings3d(A.ctypes.data, *A.shape, float(value), mm.ctypes.data)
    return int(n), float(mm[0]), float(mm[1])


def march3d(A, value, diag_mode=1, vcap=None, tcap=None, origin=(0, 0, 0)):
    """Level-0 dense march of fp32 array A at isovalue `value`.

    returns dict(pairs (V,6) int32 [low ijk, high ijk], xyz (V,3) float64 grid coords,
                 tris (T,3) int64 vertex indices, nborder int)
    diag_mode 1 reproduces the reference's CPython-3.10 set iteration order (quad diagonal)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    assert A.ndim == 3
    if vcap is None:
        vcap = int(count_crossings(A, value)[0] * 1.25) + 4096
    if tcap is None:
        tcap = 3 * vcap
    while True:
        pairs = np.zeros((vcap, 6), dtype=np.int32)
        xyz = np.zeros((vcap, 3), dtype=np.float64)
        tris = np.zeros((tcap, 3), dtype=np.int64)
        counts = np.zeros(4, dtype=np.int64)
        org = np.array(origin, dtype=np.int64)
        rc = lib().oracle_march3d_origin(A.ctypes.data, *A.shape, float(value), int(diag_mode), org.ctypes.data,
                                         pairs.ctypes.data, xyz.ctypes.data, vcap, tris.ctypes.data, tcap,
                                         counts.ctypes.data)
        if rc != 0:
            raise MemoryError("oracle_march3d")
        nv, nt, nb, nbm = (int(c) for c in counts)
        if nv <= vcap and nt <= tcap:
            return dict(pairs=pairs[:nv].copy(), xyz=xyz[:nv].copy(), tris=tris[:nt].copy(), nborder=nb, nborder_mixed=nbm)
        vcap = max(vcap * 2, nv + 1)
        tcap = max(tcap * 2, nt + 1)


# ---- canonical forms -------------------------------------------------------------------------

def edge_keys_from_pairs(pairs, shape):
    """unordered lattice edge -> int64 key = linear_index(lexicographically smaller endpoint)*8 + dir,
    dir = 4*di+2*dj+dk in 1..7 (the device's vertex id)."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 6)
    if len(pairs) == 0:
        return np.zeros(0, dtype=np.int64)
    lo = np.minimum(pairs[:, :3], pairs[:, 3:])
    hi = np.maximum(pairs[:, :3], pairs[:, 3:])
    d = hi - lo
    assert d.min() >= 0 and d.max() <= 1, "not a Kuhn edge"
    lin = (lo[:, 0] * shape[1] + lo[:, 1]) * shape[2] + lo[:, 2]
    return lin * 8 + (d[:, 0] * 4 + d[:, 1] * 2 + d[:, 2])


def key_to_pair(keys, shape):
    keys = np.asarray(keys, dtype=np.int64)
    lin, d = keys >> 3, keys & 7
    i, r = np.divmod(lin, shape[1] * shape[2])
    j, k = np.divmod(r, shape[2])
    a = np.stack([i, j, k], axis=1)
    b = a + np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
    return a, b


def canonical_level0(keys, xyz, tris):
    """sort vertices by key; triangles -> rows of 3 keys, each row sorted, rows lexsorted.
    returns (keys_sorted, xyz_sorted, tri_keys_sorted)"""
    keys = np.asarray(keys, dtype=np.int64)
    order = np.argsort(keys, kind="stable")
    tk = np.sort(keys[np.asarray(tris, dtype=np.int64)], axis=1) if len(tris) else np.zeros((0, 3), np.int64)
    if len(tk):
        tk = tk[np.lexsort((tk[:, 2], tk[:, 1], tk[:, 0]))]
    return keys[order], np.asarray(xyz)[order], tk


def tet_polygons(tri_keys, shape):
    """Level-0 comparison 'modulo the quad diagonal': group triangles by the tetrahedron they
    came from (the 4 lattice points their 3 edges touch) and return the sorted set of
    (tet lattice points..., polygon edge keys...) rows, padded with -1."""
    tri_keys = np.asarray(tri_keys, dtype=np.int64).reshape(-1, 3)
    if len(tri_keys) == 0:
        return np.zeros((0, 8), dtype=np.int64)
    a, b = key_to_pair(tri_keys.reshape(-1), shape)
    la = (a[:, 0] * shape[1] + a[:, 1]) * shape[2] + a[:, 2]
    lb = (b[:, 0] * shape[1] + b[:, 1]) * shape[2] + b[:, 2]
    pts = np.stack([la, lb], axis=1).reshape(-1, 6)
    groups = {}
    for row, tk in zip(pts, tri_keys):
        tet = tuple(sorted(set(int(x) for x in row)))
        assert len(tet) == 4, "triangle does not span a tetrahedron"
        groups.setdefault(tet, set()).update(int(x) for x in tk)
    out = []
    for tet, ks in groups.items():
        ks = sorted(ks)
        assert len(ks) in (3, 4)
        out.append(list(tet) + ks + [-1] * (4 - len(ks)))
    out = np.array(sorted(out), dtype=np.int64)
    return out
