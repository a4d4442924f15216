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


# ---- rows B4 / B5: morph triangles -----------------------------------------------------------------
def collect_morph_triangles(keys, xyzt, tets, epsilon=1e-7):
    """GridContour4D.collect_morph_triangles (pentatopes.py:314-368) with
    MorphGeometry.triangulate_tetrahedron_at_midpoints / add_tetrahedron / interpolate_pair_3d
    (morph_geometry.py:145-237) and MorphTriangles.__init__ (morph_geometry.py:7-22).
    Canonical numbering: vertices in ascending edge-key order (the reference numbers them in dict order,
    which decides the split of the 4-segment slices -- not contractual).
    returns dict(keys, points4d, segments (S,2) low-t -> high-t, triangles (T,3) segment indices (unoriented))"""
    keys = np.asarray(keys, dtype=np.int64)
    order = np.argsort(keys, kind="stable")
    inv = np.empty(len(keys), dtype=np.int64)
    inv[order] = np.arange(len(keys))
    keys = keys[order]
    V = np.asarray(xyzt, dtype=np.float64)[order]
    tets = inv[np.asarray(tets, dtype=np.int64).reshape(-1, 4)] if len(tets) else np.zeros((0, 4), np.int64)
    tvals = V[:, 3]
    tri_pairs = set()
    for tet in tets:
        a, b, c, d = sorted(int(x) for x in tet)
        ts = sorted(tvals[i] for i in (a, b, c, d))
        prev = None
        for cur in ts:
            if prev is not None and (cur - prev) > 1e-4:          # morph_geometry.py:150
                mid = 0.5 * (cur + prev)
                spanning = []
                for (i, j) in ((a, b), (a, c), (a, d), (b, c), (b, d), (c, d)):
                    v1, v2 = tvals[i], tvals[j]
                    if v1 > v2:
                        v1, v2 = v2, v1
                    if mid + 1e-5 < v1 or mid - 1e-5 > v2:          # morph_geometry.py:218
                        continue
                    spanning.append((i, j))
                if len(spanning) == 3:
                    tri_pairs.add(frozenset(spanning))
                elif len(spanning) == 4:                            # morph_geometry.py:176-186
                    pair1 = spanning[0]
                    pair2 = None
                    for p in spanning[1:]:
                        if not (set(p) & set(pair1)):
                            pair2 = p
                    for p in spanning:
                        if p != pair1 and p != pair2:
                            tri_pairs.add(frozenset([pair1, pair2, p]))
            prev = cur
    t_eps = epsilon * (tvals.max() - tvals.min()) if len(tvals) else 0.0
    kept = [t for t in tri_pairs if all(abs(tvals[i] - tvals[j]) > t_eps for (i, j) in t)]    # pentatopes.py:336-348
    seg_set = sorted(set(p for t in kept for p in t))
    seg_index = {p: n for n, p in enumerate(seg_set)}
    segments = np.array([(j, i) if tvals[i] > tvals[j] else (i, j) for (i, j) in seg_set], dtype=np.int64).reshape(-1, 2)
    triangles = np.array([sorted(seg_index[p] for p in t) for t in kept], dtype=np.int64).reshape(-1, 3)
    if len(triangles):
        triangles = triangles[np.lexsort(triangles.T[::-1])]
    return dict(keys=keys, points4d=V, segments=segments, triangles=triangles)


def orient_morph_triangles(M):
    """MorphTriangles.orient_triangles (morph_geometry.py:49-89): SurfaceGeometry.orient_triangles on the
    3-D midpoints of the segments, propagating only between triangles whose time ranges overlap."""
    from . import postpass
    V, segs, tris = M["points4d"], M["segments"], M["triangles"]
    if len(tris) == 0:
        return tris, np.zeros(0, np.int64), np.zeros(0, np.uint8)
    tv = V[:, 3]
    lo_t, hi_t = tv[segs[:, 0]], tv[segs[:, 1]]
    tmin = np.maximum(tv.min(), lo_t[tris].max(axis=1))
    tmax = np.minimum(tv.max(), hi_t[tris].min(axis=1))
    mid = 0.5 * (V[segs[:, 0], :3] + V[segs[:, 1], :3])

    def compatible(t1, t2):
        return max(tmin[t1], tmin[t2]) < min(tmax[t1], tmax[t2])
    return postpass.orient(mid, tris, compatible)


def morph_polygons(point_keys, segments, triangles):
    """order-independent form of a morph-triangle set, blind to the numbering-dependent split of 4-segment
    slices: every triangle -> the full slice polygon it belongs to (3 or 4 segments as key pairs)."""
    point_keys = np.asarray(point_keys, dtype=np.int64)
    seg_keys = [tuple(sorted((int(point_keys[i]), int(point_keys[j])))) for i, j in np.asarray(segments)]
    polys = set()
    for t in np.asarray(triangles):
        ps = [seg_keys[s] for s in t]
        verts = sorted(set(k for p in ps for k in p))
        assert len(verts) == 4
        deg = {k: sum(k in p for p in ps) for k in verts}
        if max(deg.values()) == 3:
            polys.add(tuple(sorted(ps)))
        else:
            # path x - y - z - w : sides {y, w} and {x, z}; polygon = all 4 segments between the sides
            ends = [k for k in verts if deg[k] == 1]
            mids = [k for k in verts if deg[k] == 2]
            x = ends[0]
            y = [k for k in mids if tuple(sorted((x, k))) in ps][0]
            z = [k for k in mids if k != y][0]
            w = ends[1]
            side1, side2 = (y, w), (x, z)
            polys.add(tuple(sorted(tuple(sorted((p, q))) for p in side1 for q in side2)))
    return polys


def winding_agreement(keys_a, segs_a, tris_a, keys_b, segs_b, tris_b):
    "(#triangles present in both as unordered segment triples, #of those with the same winding)"
    def canon(keys, segs, tris):
        sk = [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in np.asarray(segs)]
        out = {}
        for t in np.asarray(tris):
            tr = [sk[s] for s in t]
            m = min(range(3), key=lambda n: tr[n])
            out[frozenset(tr)] = (tr[m], tr[(m + 1) % 3], tr[(m + 2) % 3])
        return out
    da, db = canon(keys_a, segs_a, tris_a), canon(keys_b, segs_b, tris_b)
    common = set(da) & set(db)
    return len(common), sum(da[k] == db[k] for k in common)
