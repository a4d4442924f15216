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


def orient_morph_triangles(M, shuffle=None):
    """MorphTriangles.orient_triangles (morph_geometry.py:49-89): SurfaceGeometry.orient_triangles on the
    3-D midpoints of the segments, propagating only between triangles whose time ranges overlap.
    shuffle: see postpass.orient."""
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
    return postpass.orient(mid, tris, compatible, shuffle)


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


def winding_excuses(keys_a, segs_a, tris_a, keys_b, segs_b, tris_b, points4d_b):
    """Where do two windings of the same morph triangles differ, and may they?  a = the build, b = the reference (or its
    restatement), triangles matched as unordered triples of segments (segments as sorted pairs of edge keys).

    The reference orients by flood fill (morph_geometry.py:49-66 on surface_geometry.py:52-140): across a segment that is
    shared by exactly TWO triangles with overlapping time ranges the relative winding is forced, so two consistent
    windings can only differ by whole patches bounded by segments with three or more triangles, by time-incompatible
    pairs, or by the rim.  Returns dict:
      common, agree            matched triangles / of those wound the same way
      forced_breaks            manifold, time-compatible segments whose two triangles have DIFFERENT agreement status --
                               must be 0: anything else is a real winding error on one side
      patches                  list of (size, agree?) of the patches of equal status connected through such segments
      nonmanifold_segments     segments with >= 3 triangles (where the reference's own traversal order decides)"""
    def seg_keys(keys, segs):
        return [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in np.asarray(segs)]

    def canon(sk, tris):
        out = {}
        for t in np.asarray(tris):
            tr = [sk[s] for s in t]
            m = min(range(3), key=lambda n: tr[n])
            out[frozenset(tr)] = (tr[m], tr[(m + 1) % 3], tr[(m + 2) % 3])
        return out
    ska, skb = seg_keys(keys_a, segs_a), seg_keys(keys_b, segs_b)
    da, db = canon(ska, tris_a), canon(skb, tris_b)
    common = sorted(set(da) & set(db), key=lambda k: sorted(k))
    status = {k: da[k] == db[k] for k in common}
    # time range of every reference triangle (compute_triangle_stats, morph_geometry.py:68-89)
    P = np.asarray(points4d_b, dtype=np.float64)
    segs_b = np.asarray(segs_b)
    tmin_all, tmax_all = float(P[:, 3].min()), float(P[:, 3].max())
    trange = {}
    for t in np.asarray(tris_b):
        lo, hi = tmin_all, tmax_all
        for s in t:
            i, j = segs_b[s]
            a, b = sorted((P[i, 3], P[j, 3]))
            lo, hi = max(lo, a), min(hi, b)
        trange[frozenset(skb[s] for s in t)] = (lo, hi)
    by_segment = {}
    for k in db:
        for s in k:
            by_segment.setdefault(s, []).append(k)
    def crowded(ts):   # some time at which three or more triangles hang on the segment
        for z in ts:
            for w in ts:
                lo, hi = max(trange[z][0], trange[w][0]), min(trange[z][1], trange[w][1])
                if lo < hi and sum(1 for u in ts if trange[u][0] < 0.5 * (lo + hi) < trange[u][1]) >= 3:
                    return True
        return False
    nonmanifold = [s for s, ts in by_segment.items() if len(ts) >= 3 and crowded(ts)]
    parent = {k: k for k in common}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    forced_breaks = 0
    for s, ts in by_segment.items():
        # two triangles on a segment force each other's winding when their time ranges overlap and, at a time both are
        # there, no third triangle hangs on the segment (the time slice is a manifold surface along this edge)
        for x in range(len(ts)):
            for y in range(x + 1, len(ts)):
                if ts[x] not in status or ts[y] not in status:
                    continue
                (l0, h0), (l1, h1) = trange[ts[x]], trange[ts[y]]
                lo, hi = max(l0, l1), min(h0, h1)
                if not (lo < hi):
                    continue          # not time compatible: the flood fill does not cross here
                mid = 0.5 * (lo + hi)
                if sum(1 for z in ts if trange[z][0] < mid < trange[z][1]) != 2:
                    continue          # three or more at that time: the reference's traversal order decides
                if status[ts[x]] != status[ts[y]]:
                    forced_breaks += 1
                else:
                    parent[find(ts[x])] = find(ts[y])
    sizes = {}
    for k in common:
        r = find(k)
        sizes.setdefault(r, [0, status[k]])[0] += 1
    return dict(common=len(common), agree=sum(status.values()), forced_breaks=forced_breaks,
                patches=sorted((n, bool(a)) for n, a in sizes.values()), nonmanifold_segments=len(nonmanifold))


def order_dependent_triangles(M, runs=6, seed=5):
    """triangles of the morph set M (dict as collect_morph_triangles returns) whose winding the reference's flood fill
    does NOT determine: it comes out differently when the triangles hanging on a crowded segment are visited in another
    order (the reference iterates a Python set there).  -> set of frozensets of segment key pairs"""
    keys = M["keys"]
    sk = [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in np.asarray(M["segments"])]

    def canon(tris):
        out = {}
        for t in np.asarray(tris):
            tr = [sk[s] for s in t]
            m = min(range(3), key=lambda n: tr[n])
            out[frozenset(tr)] = (tr[m], tr[(m + 1) % 3], tr[(m + 2) % 3])
        return out
    base = canon(orient_morph_triangles(M)[0])
    varying = set()
    rng = np.random.RandomState(seed)
    for _ in range(runs):
        other = canon(orient_morph_triangles(M, rng)[0])
        varying.update(k for k in base if other[k] != base[k])
    return varying


def disagreements(keys_a, segs_a, tris_a, keys_b, segs_b, tris_b):
    "triangles present in both sets (as unordered segment triples) that are wound differently -> set of frozensets"
    def canon(keys, segs, tris):
        sk = [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in np.asarray(segs)]
        out = {}
        for t in np.asarray(tris):
            tr = [sk[s] for s in t]
            m = min(range(3), key=lambda n: tr[n])
            out[frozenset(tr)] = (tr[m], tr[(m + 1) % 3], tr[(m + 2) % 3])
        return out
    da, db = canon(keys_a, segs_a, tris_a), canon(keys_b, segs_b, tris_b)
    return set(k for k in set(da) & set(db) if da[k] != db[k])


def forced_pair_violations(keys, segs, tris, points4d):
    """pairs of triangles that hang on one segment, share a time range and are alone on the segment at a time inside it
    (the time slice is a manifold surface along that edge), but run the segment in the SAME direction: an inconsistent
    winding of the surface at that time.  -> (number of such pairs, number of forced pairs examined)"""
    sk = [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in np.asarray(segs)]
    P = np.asarray(points4d, dtype=np.float64)
    segs = np.asarray(segs)
    tris = np.asarray(tris)
    lo_all, hi_all = float(P[:, 3].min()), float(P[:, 3].max())
    tr = []
    for t in tris:
        lo, hi = lo_all, hi_all
        for s in t:
            a, b = sorted((P[segs[s][0], 3], P[segs[s][1], 3]))
            lo, hi = max(lo, a), min(hi, b)
        tr.append((lo, hi))
    by_segment = {}
    for n, t in enumerate(tris):
        for pos in range(3):
            # direction in which triangle n runs segment t[pos]: towards t[pos+1]
            by_segment.setdefault(int(t[pos]), []).append((n, int(t[(pos + 1) % 3])))
    bad = examined = 0
    for s, users in by_segment.items():
        for x in range(len(users)):
            for y in range(x + 1, len(users)):
                (n0, nxt0), (n1, nxt1) = users[x], users[y]
                lo, hi = max(tr[n0][0], tr[n1][0]), min(tr[n0][1], tr[n1][1])
                if not lo < hi:
                    continue
                mid = 0.5 * (lo + hi)
                if sum(1 for (n, _) in users if tr[n][0] < mid < tr[n][1]) != 2:
                    continue
                examined += 1
                # consistent: one triangle goes s -> next, the other previous -> s, i.e. the third corners differ in role;
                # with triangles as cyclic triples of SEGMENTS (vertices of the slice), a shared vertex s proves nothing by
                # itself -- the shared EDGE is (s, other shared segment): compare through the second shared vertex
                t0, t1 = [int(v) for v in tris[n0]], [int(v) for v in tris[n1]]
                shared = [v for v in t0 if v in t1]
                if len(shared) != 2:
                    continue
                a, b = shared
                d0 = (t0.index(b) - t0.index(a)) % 3 == 1
                d1 = (t1.index(b) - t1.index(a)) % 3 == 1
                if d0 == d1:
                    bad += 1
    return bad, examined
