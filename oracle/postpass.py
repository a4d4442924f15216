"""oracle/postpass.py -- TEST INFRASTRUCTURE ONLY.

Python/numpy restatement of the reference's mesh post-passes ("Level 1", SURVEY.md 8a rows
A6-A11) applied to a Level-0 mesh (vertices keyed by lattice edge, triangles as index triples):

    weld            GridContour.quantize_interpolations   contourist/tetrahedral.py:190-215
    tiny collapse   GridContour.remove_tiny_simplices     contourist/tetrahedral.py:353-375
    extract         GridContour3d.extract_surface_geometry contourist/tetrahedral.py:604-621
    clean           SurfaceGeometry.clean_triangles       contourist/surface_geometry.py:14-50
    orient          SurfaceGeometry.orient_triangles      contourist/surface_geometry.py:52-140

The reference processes Python sets/dicts in hash order, which makes a few choices order
dependent (which bucket member represents a welded vertex, the tiny-collapse merge point,
sequential side effects).  This restatement fixes a CANONICAL order (ascending edge key) for
those choices -- the same one the HIP path implements -- and the goldens record whether the
reference's own output is order-invariant for each fixture (oracle/make_goldens.py).
"""
import numpy as np


def _trunc_int(x):
    return np.trunc(x).astype(np.int64)


def expander_for(corner, divisions=10000):
    """tetrahedral.py:192  expander = ((divisions * 1.0) / corner).astype(int)"""
    return _trunc_int((divisions * 1.0) / np.asarray(corner, dtype=np.float64))


def weld_buckets(xyz, corner):
    """tetrahedral.py:196  quantized = (interpolation * expander).astype(int)"""
    return _trunc_int(np.asarray(xyz, dtype=np.float64) * expander_for(corner))


class UnionFind(object):
    def __init__(self, n):
        self.p = np.arange(n, dtype=np.int64)

    def find(self, x):
        p = self.p
        r = x
        while p[r] != r:
            r = p[r]
        while p[x] != r:
            p[x], x = r, p[x]
        return r

    def union(self, a, b):
        a, b = self.find(a), self.find(b)
        if a == b:
            return
        if a < b:          # root = smallest member
            self.p[b] = a
        else:
            self.p[a] = b

    def roots(self):
        return np.array([self.find(i) for i in range(len(self.p))], dtype=np.int64)


def _tri_priority(keys, tri):
    return tuple(sorted(int(keys[v]) for v in tri))


def weld(keys, xyz, tris, corner):
    """A6.  keys (V,) edge keys, xyz (V,3) grid coords, tris (T,3) vertex indices.
    Canonical representative of a bucket = member with the LARGEST edge key (the members of a bucket are crossings
    on the upward edges of one lattice point; the largest key is the most diagonal edge, which is what the reference's
    order-dependent "last inserted wins" (:198-203) picks most often: fixture coarse_sphere_r12_corner511).
    returns (rep (V,) vertex index of each vertex's representative, tris' (T',3))
    Triangles whose three representatives are not distinct are dropped (:209-211); triangles that
    become the same vertex SET are merged (set semantics :206-211) -- canonical survivor = the one
    with the lexicographically smallest sorted triple of ORIGINAL edge keys; it keeps its winding."""
    keys = np.asarray(keys, dtype=np.int64)
    q = weld_buckets(xyz, corner)
    rep = np.arange(len(keys), dtype=np.int64)
    best = {}
    for v in np.argsort(-keys, kind="stable"):
        b = (int(q[v, 0]), int(q[v, 1]), int(q[v, 2]))
        if b not in best:
            best[b] = int(v)
        rep[v] = best[b]
    winners = {}
    for t in np.asarray(tris, dtype=np.int64).reshape(-1, 3):
        m = (int(rep[t[0]]), int(rep[t[1]]), int(rep[t[2]]))
        if len(set(m)) < 3:
            continue
        s = tuple(sorted(m))
        pr = _tri_priority(keys, t)
        if s not in winners or pr < winners[s][0]:
            winners[s] = (pr, m)
    out = np.array([w[1] for w in winners.values()], dtype=np.int64).reshape(-1, 3)
    tprio = np.array([w[0] for w in winners.values()], dtype=np.int64).reshape(-1, 3)
    return rep, out, tprio


def tiny_mask(xyz, tris, corner, epsilon=1e-4):
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(tris) == 0:
        return np.zeros(0, dtype=bool)
    P = np.asarray(xyz, dtype=np.float64)[tris]
    delta = (P.max(axis=1) - P.min(axis=1)) * (1.0 / np.asarray(corner, dtype=np.float64))
    return delta.max(axis=1) < epsilon


def tiny_collapse(keys, xyz, tris, corner, epsilon=1e-4):
    """A7.  drop triangles whose bounding box, scaled by 1/corner, is smaller than epsilon in every
    axis (:360-365) and move their three vertices onto one point (:368-370).
    Canonical form: the tiny test uses the coordinates BEFORE any move; vertices linked by tiny
    triangles form groups (union-find); every member takes the coordinates of the group's
    smallest-edge-key member.  returns (xyz', tris')"""
    keys = np.asarray(keys, dtype=np.int64)
    xyz = np.array(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    invcorner = 1.0 / np.asarray(corner, dtype=np.float64)
    if len(tris) == 0:
        return xyz, tris
    P = xyz[tris]                                    # (T,3,3)
    delta = (P.max(axis=1) - P.min(axis=1)) * invcorner
    tiny = delta.max(axis=1) < epsilon
    if tiny.any():
        # union by key order: relabel vertices by key rank so "smallest root" == smallest key
        order = np.argsort(keys, kind="stable")
        rank = np.empty(len(keys), dtype=np.int64)
        rank[order] = np.arange(len(keys))
        uf = UnionFind(len(keys))
        for t in tris[tiny]:
            r = rank[t]
            uf.union(int(r[0]), int(r[1]))
            uf.union(int(r[0]), int(r[2]))
        involved = np.unique(tris[tiny].reshape(-1))
        for v in involved:
            root_rank = uf.find(int(rank[v]))
            xyz[v] = xyz[order[root_rank]]
    return xyz, tris[~tiny]


def tiny_sites(xyz, tris, corner, epsilon=1e-4):
    """coordinates of the vertices of tiny triangles (where the reference's order-dependent
    merge-point choice may move a vertex by less than one weld bucket)"""
    xyz = np.asarray(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(tris) == 0:
        return np.zeros((0, 3))
    P = xyz[tris]
    delta = (P.max(axis=1) - P.min(axis=1)) / np.asarray(corner, dtype=np.float64)
    tiny = delta.max(axis=1) < epsilon
    return P[tiny].reshape(-1, 3)


def extract(xyz, tris):
    """A8.  compact the vertices used by triangles (:605-614). returns (xyz_used, tris_renumbered, used_ids)"""
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    used = np.unique(tris.reshape(-1))
    remap = -np.ones(len(xyz), dtype=np.int64)
    remap[used] = np.arange(len(used))
    return np.asarray(xyz)[used], remap[tris], used


def clean(xyz, tris, tprio=None):
    """A9.  SurfaceGeometry.clean_triangles: omit triangles with np.allclose(cross((A-C),(B-C)), 0)
    (:33-34) and merge those of their vertex pairs that are np.allclose (:38-43).
    Canonical form: all merges are collected first (union-find, root = smallest index), then every
    triangle is remapped; triangles left with fewer than 3 distinct vertices are dropped (the
    reference drops them in orient_triangles, surface_geometry.py:63) and equal vertex sets merged.
    returns (xyz', tris') with vertices compacted in first-use order."""
    xyz = np.asarray(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(tris) == 0:
        return xyz[:0], tris
    A, B, C = xyz[tris[:, 0]], xyz[tris[:, 1]], xyz[tris[:, 2]]
    cross = np.cross(A - C, B - C)
    degenerate = np.all(np.abs(cross) <= 1e-8, axis=1)          # allclose(cross, 0): rtol*|0| = 0
    uf = UnionFind(len(xyz))

    def close(i, j):   # np.allclose(p_i, p_j): |a-b| <= 1e-8 + 1e-5*|b|
        return bool(np.all(np.abs(xyz[i] - xyz[j]) <= 1e-8 + 1e-5 * np.abs(xyz[j])))
    for a, b, c in tris[degenerate]:
        for (i, j) in ((a, b), (a, c), (b, c)):
            if close(i, j):
                uf.union(int(i), int(j))
    root = uf.roots()
    kept = {}
    if tprio is None:
        tprio = np.sort(tris, axis=1)
    for t, pr in zip(tris[~degenerate], np.asarray(tprio)[~degenerate]):
        m = (int(root[t[0]]), int(root[t[1]]), int(root[t[2]]))
        if len(set(m)) < 3:
            continue
        s = tuple(sorted(m))
        pr = tuple(int(x) for x in pr)
        if s not in kept or pr < kept[s][0]:      # equal vertex sets: smallest original priority triple survives
            kept[s] = (pr, m)
    out = np.array([w[1] for w in kept.values()], dtype=np.int64).reshape(-1, 3)
    x2, t2, _ = extract(xyz, out)
    return x2, t2


def orient(xyz, tris, compatible=None, shuffle=None):
    """A10.  SurfaceGeometry.orient_triangles, restated faithfully (input winding is IGNORED):
    while unoriented triangles remain: take the vertex with the largest (x, index) among them
    (:79), among its unoriented triangles the one with the largest |cross(a-b, a-c)[0]| (:88-94,
    ties: last wins in the reference's set order; here the largest triangle id), wind it so that
    component is positive (:99-103) and flood-fill across shared edges so that a neighbour traverses
    the shared edge in the opposite direction (:110-138).
    returns (tris_oriented (T,3) in input order, component_label (T,), comp_flags (ncomp,) uint8)
    comp_flags bit 0: the component is not an orientable edge-manifold (an edge with more than two
    triangles, or the flood fill met a conflicting orientation) -> the reference's result depends on
    its traversal order; bit 1: the start rule is ambiguous (several vertices share the largest x, or
    several of their triangles share the largest |cross_x|, and they do not all imply the same
    winding, or cross_x is exactly 0) -> the reference's result depends on its vertex numbering."""
    # shuffle: a numpy RandomState -- the triangles of an edge are visited in a random order, as the reference's
    # `for triangle in triangles` over a SET does from run to run (surface_geometry.py:117); running the fill with several
    # orders shows which windings its traversal order decides (edges with three or more triangles)
    xyz = np.asarray(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    T = len(tris)
    if T == 0:
        return tris, np.zeros(0, np.int64), np.zeros(0, np.uint8)
    edge_tris = {}
    vert_tris = {}
    for t in range(T):
        a, b, c = (int(x) for x in tris[t])
        for v in (a, b, c):
            vert_tris.setdefault(v, []).append(t)
        for e in ((a, b), (b, c), (a, c)):
            edge_tris.setdefault((min(e), max(e)), []).append(t)
    if shuffle is not None:
        for e in edge_tris:
            if len(edge_tris[e]) > 2:
                shuffle.shuffle(edge_tris[e])
    orientation = {}
    label = -np.ones(T, dtype=np.int64)
    unoriented = set(range(T))
    ncomp = 0
    while unoriented:
        vs = set()
        for t in unoriented:
            vs.update(int(x) for x in tris[t])
        vmax = max((xyz[i][0], i) for i in vs)[1]
        initial, maxdotx = None, 0.0
        for t in sorted(vert_tris[vmax]):
            if t not in unoriented:
                continue
            a, b, c = (xyz[i] for i in tris[t])
            dotx = np.cross(a - b, a - c)[0]
            if abs(dotx) >= abs(maxdotx):
                maxdotx, initial = dotx, t
        o = tuple(int(x) for x in tris[initial])
        a, b, c = (xyz[i] for i in o)
        if np.cross(a - b, a - c)[0] < 0:
            o = tuple(reversed(o))
        stack = [(initial, o)]
        while stack:
            t, o = stack.pop()
            orientation[t] = o
            label[t] = ncomp
            unoriented.discard(t)
            a, b, c = o
            for (i1, i2) in ((c, b), (b, a), (a, c)):
                for t2 in edge_tris[(min(i1, i2), max(i1, i2))]:
                    if t2 != t and t2 not in orientation and (compatible is None or compatible(t2, t)):
                        (i3,) = set(int(x) for x in tris[t2]) - {i1, i2}
                        stack.append((t2, (i1, i2, i3)))
        ncomp += 1
    out = np.array([orientation[t] for t in range(T)], dtype=np.int64).reshape(-1, 3)
    flags = np.zeros(ncomp, dtype=np.uint8)
    # bit 0: non-manifold / non-orientable
    for e, ts in edge_tris.items():
        if len(ts) > 2:
            flags[label[ts[0]]] |= 1
        elif len(ts) == 2:
            def direction(t):
                o = orientation[t]
                for n in range(3):
                    if {o[n], o[(n + 1) % 3]} == set(e):
                        return (o[n], o[(n + 1) % 3])
            if direction(ts[0]) == direction(ts[1]):
                flags[label[ts[0]]] |= 1
    # bit 1: ambiguous start
    for cid in range(ncomp):
        members = np.nonzero(label == cid)[0]
        vs = np.unique(tris[members].reshape(-1))
        xmax = xyz[vs, 0].max()
        for v in vs[xyz[vs, 0] == xmax]:
            cand = [t for t in vert_tris[int(v)] if label[t] == cid]
            dots = []
            for t in cand:
                a, b, c = (xyz[i] for i in orientation[t])
                dots.append(np.cross(a - b, a - c)[0])
            m = max(abs(d) for d in dots)
            for d in dots:
                if abs(d) >= m * (1 - 1e-12) and d <= 0:
                    flags[cid] |= 2
    return out, label, flags


_QSCALE = [None]     # when set: compare by coordinates rounded to 1/scale instead of by weld buckets


def set_compare_scale(scale):
    """smoothed meshes put many coordinates exactly ON weld-bucket boundaries (symmetric averages), where the
    order of a float64 sum decides the bucket; such fixtures are compared by rounded coordinates instead"""
    _QSCALE[0] = scale


def _compare_ids(points, corner):
    if _QSCALE[0] is None:
        return weld_buckets(points, corner)
    return np.rint(np.asarray(points, dtype=np.float64) * _QSCALE[0]).astype(np.int64)


def canonical_level1(grid_points, triangles, corner):
    """order-independent Level-1 form: every triangle as a triple of weld-bucket ids
    of its (final) vertex coordinates, rotated so the smallest bucket comes first (winding kept),
    rows sorted.  returns int64 array (T, 9)."""
    grid_points = np.asarray(grid_points, dtype=np.float64).reshape(-1, 3)
    triangles = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)
    if len(triangles) == 0:
        return np.zeros((0, 9), dtype=np.int64)
    q = _compare_ids(grid_points, corner)
    # scalar bucket id for ordering
    big = int(q.max()) + 2
    sid = (q[:, 0] * big + q[:, 1]) * big + q[:, 2]
    s = sid[triangles]                                  # (T,3)
    first = np.argmin(s, axis=1)
    idx = (first[:, None] + np.arange(3)[None, :]) % 3
    rot = np.take_along_axis(triangles, idx, axis=1)
    rows = q[rot].reshape(-1, 9)
    order = np.lexsort(rows.T[::-1])
    return rows[order]


def smooth_interpolations(xyz, tris, factor):
    """GridContour.smooth_interpolations (tetrahedral.py:329-351): every vertex of a triangle moves by `factor`
    towards the mean of the vertices of its triangles (the vertex itself included, each neighbour once)."""
    xyz = np.array(xyz, dtype=np.float64)
    adj = {}
    for t in np.asarray(tris, dtype=np.int64).reshape(-1, 3):
        for v in t:
            adj.setdefault(int(v), set()).update(int(x) for x in t)
    new = xyz.copy()
    for v, nb in adj.items():
        avg = xyz[sorted(nb)].mean(axis=0)
        new[v] = xyz[v] - factor * (xyz[v] - avg)
    return new


def level1_from_level0(keys, xyz, tris, corner, smooth=None):
    """full canonical post-pass chain on a Level-0 mesh (input winding is irrelevant).
    returns dict(grid_points, triangles, n_after_weld, n_after_tiny, flipped)"""
    keys = np.asarray(keys, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    # canonical numbering: vertices in ascending edge-key order, so "smallest index" == "smallest key"
    order = np.argsort(keys, kind="stable")
    inv = np.empty(len(keys), dtype=np.int64)
    inv[order] = np.arange(len(keys))
    keys, xyz, tris = keys[order], xyz[order], (inv[tris] if len(tris) else tris)
    rep, t1, p1 = weld(keys, xyz, tris, corner)
    n_after_weld = len(t1)
    if smooth:
        xyz = smooth_interpolations(xyz, t1, smooth)
    # where the reference's hash-order artefacts can act: welded groups (which member represents
    # the bucket), tiny triangles (merge point), zero-area triangles (merge direction)
    welded = np.nonzero(rep != np.arange(len(rep)))[0]
    group = np.unique(np.concatenate([welded, rep[welded]])) if len(welded) else np.zeros(0, np.int64)
    sites = np.concatenate([xyz[group].reshape(-1, 3), tiny_sites(xyz, t1, corner).reshape(-1, 3)], axis=0)
    keep = ~tiny_mask(xyz, t1, corner)
    xyz2, t2 = tiny_collapse(keys, xyz, t1, corner)
    p2 = p1[keep]
    n_after_tiny = len(t2)
    x3, t3, _ = extract(xyz2, t2)
    # vertices of zero-area triangles: clean_triangles merges them in hash order (not contractual)
    if len(t3):
        Pc = x3[t3]
        crossc = np.cross(Pc[:, 0] - Pc[:, 2], Pc[:, 1] - Pc[:, 2])
        degenerate = np.all(np.abs(crossc) <= 1e-8, axis=1)
        sites = np.concatenate([sites.reshape(-1, 3), Pc[degenerate].reshape(-1, 3)], axis=0)
    x4, t4 = clean(x3, t3, p2)
    t5, label, comp_flags = orient(x4, t4)
    return dict(grid_points=x4, triangles=t5, n_after_weld=n_after_weld, n_after_tiny=n_after_tiny,
                labels=label, comp_flags=comp_flags, sites=sites)


def _unoriented(rows):
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 3, 3)
    out = []
    for r in rows:
        out.append(tuple(sorted(tuple(int(x) for x in b) for b in r)))
    return out


def compare_level1(oracle_L1, other_points, other_tris, corner, reach=2):
    """Compare another Level-1 mesh (the reference's golden, or the HIP path's output) with the
    oracle's canonical Level-1 result `oracle_L1` (dict of level1_from_level0).

    Contract: same triangles as triples of weld buckets, same winding.  Not
    contractual, hence excused: (a) rows touching a site where the reference's hash order picks a
    weld representative / merge point (`sites`); (b) the winding of components whose orientation the
    reference itself only fixes through traversal order or vertex numbering (comp_flags).
    returns dict(missing, extra: unexcused unoriented rows on one side only;
                 winding: unexcused winding mismatches; excused_rows, excused_winding)"""
    o_rows = canonical_level1(oracle_L1["grid_points"], oracle_L1["triangles"], corner)
    x_rows = canonical_level1(other_points, other_tris, corner)
    # component flags of the oracle's triangles, keyed by unoriented row
    q = _compare_ids(oracle_L1["grid_points"], corner)
    o_tris = np.asarray(oracle_L1["triangles"], dtype=np.int64).reshape(-1, 3)
    flag_of = {}
    for t, lab in zip(o_tris, oracle_L1["labels"]):
        key = tuple(sorted(tuple(int(x) for x in q[v]) for v in t))
        flag_of[key] = int(oracle_L1["comp_flags"][lab])
    o_un, x_un = _unoriented(o_rows), _unoriented(x_rows)
    o_map = dict(zip(o_un, map(tuple, o_rows.tolist())))
    x_map = dict(zip(x_un, map(tuple, x_rows.tolist())))
    sites = oracle_L1["sites"]
    sb = _compare_ids(np.asarray(sites, dtype=np.float64).reshape(-1, 3), corner) if len(sites) else np.zeros((0, 3), np.int64)
    if _QSCALE[0] is not None:
        reach = int(reach * _QSCALE[0] / float(np.min(expander_for(corner)))) + 1

    def near_site(un):
        if len(sb) == 0:
            return False
        b = np.array(un, dtype=np.int64).reshape(3, 3)
        d = np.abs(b[:, None, :] - sb[None, :, :]).max(axis=2)
        return bool((d <= reach).any())
    missing = [u for u in o_map if u not in x_map]
    extra = [u for u in x_map if u not in o_map]
    res = dict(excused_rows=0, excused_winding=0)
    res["missing"] = [u for u in missing if not near_site(u)]
    res["extra"] = [u for u in extra if not near_site(u)]
    res["excused_rows"] = len(missing) + len(extra) - len(res["missing"]) - len(res["extra"])
    winding = []
    for u, row in o_map.items():
        if u in x_map and x_map[u] != row:
            if flag_of.get(u, 0) != 0 or near_site(u):
                res["excused_winding"] += 1
            else:
                winding.append(u)
    res["winding"] = winding
    res["n_oracle"], res["n_other"] = len(o_rows), len(x_rows)
    return res


def compare_canonical(ref_rows, got_rows, sites, corner, reach=2):
    """Level-1 comparison that is blind to the NON-contractual choices of the reference
    (tiny-collapse merge point): rows present on one side only are excused when one
    of their three weld buckets lies within `reach` buckets (per axis) of a tiny-collapse site.
    returns (unexcused rows only in ref, unexcused rows only in got, number of excused rows)"""
    rs = set(map(tuple, np.asarray(ref_rows).tolist()))
    gs = set(map(tuple, np.asarray(got_rows).tolist()))
    only_r, only_g = rs - gs, gs - rs
    if not only_r and not only_g:
        return [], [], 0
    sb = weld_buckets(np.asarray(sites, dtype=np.float64).reshape(-1, 3), corner) if len(sites) else np.zeros((0, 3), np.int64)

    def excused(row):
        if len(sb) == 0:
            return False
        b = np.array(row, dtype=np.int64).reshape(3, 3)
        d = np.abs(b[:, None, :] - sb[None, :, :]).max(axis=2)       # (3, nsites)
        return bool((d <= reach).any())
    bad_r = [r for r in only_r if not excused(r)]
    bad_g = [g for g in only_g if not excused(g)]
    return bad_r, bad_g, len(only_r) + len(only_g) - len(bad_r) - len(bad_g)
