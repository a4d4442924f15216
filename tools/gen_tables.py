#!/usr/bin/env python3
"""Generate contourist_amd/csrc/cx_tables.h : lookup tables of the 6-tetrahedra
(Kuhn) voxel march and of the 24-pentatope hypercube march.

The tables are derived from first principles here (monotone lattice paths of
the unit cube / hypercube); the only facts taken from the reference are the
ORDER in which it lists cube corners, tetrahedra and pentatopes, because that
order is the insertion order into its low/high point sets and therefore decides
the quad diagonal (reference contourist/tetrahedral.py:20-39, 561-595 and
contourist/pentatopes.py:15-30, 223-291).

Conventions
-----------
3-D corner index  c = 4*di + 2*dj + dk      (array axes (i,j,k), k fastest)
edge              (c1, d)                   c1 = lower corner (bit subset of c2), d = c1 ^ c2 in 1..7
                                            (edge direction); edge id = index in the list of 19 edges
A grid vertex q "owns" the 7 edges q -> q+d, so e names (owner corner, direction).
tet pattern       bit m set <=> tet vertex m (reference order) has f < value.
Triangles are wound so that (p1-p0)x(p2-p0) points from the low side (f<value)
to the high side.
"""
import itertools
import os
import sys

import numpy as np

# --- 3-D ------------------------------------------------------------------
# reference order of tetrahedra: [A,H,B,D] [A,H,D,C] [A,H,C,G] [A,H,G,E] [A,H,E,F] [A,H,F,B]
# with A..H = corners 0..7 in (di,dj,dk) binary order (tetrahedral.py:20-39).
TETS = [
    [0, 7, 1, 3],
    [0, 7, 3, 2],
    [0, 7, 2, 6],
    [0, 7, 6, 4],
    [0, 7, 4, 5],
    [0, 7, 5, 1],
]


def corner_xyz(c):
    return np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1], dtype=float)


# the 19 edges of the Kuhn-triangulated voxel: (owner corner c1, direction d), c1 a strict bit-subset of c1|d
EDGES = [(c1, d) for c1 in range(7) for d in range(1, 8) if (c1 & d) == 0]
EDGE_ID = {e: n for n, e in enumerate(EDGES)}
assert len(EDGES) == 19


def edge_ref(c1, c2):
    lo, hi = (c1, c2) if (c1 & c2) == c1 else (c2, c1)
    assert (lo & hi) == lo and lo != hi, (c1, c2)
    return EDGE_ID[(lo, lo ^ hi)]


def edge_mid(c1, c2, t=0.5):
    return corner_xyz(c1) * (1 - t) + corner_xyz(c2) * t


def wind(tri_edges, low, high):
    """orient a triangle given as 3 (c_low, c_high) corner pairs so the normal points low->high."""
    # use a generic (non-midpoint) interpolation so nothing is accidentally degenerate
    ts = [0.37, 0.52, 0.61]
    pts = [edge_mid(a, b, t) for (a, b), t in zip(tri_edges, ts)]
    n = np.cross(pts[1] - pts[0], pts[2] - pts[0])
    g = np.mean([corner_xyz(c) for c in high], axis=0) - np.mean([corner_xyz(c) for c in low], axis=0)
    s = float(np.dot(n, g))
    assert abs(s) > 1e-9, (tri_edges, s)
    if s < 0:
        tri_edges = [tri_edges[0], tri_edges[2], tri_edges[1]]
    return tri_edges


def tet_entry(tet, pattern, variant):
    """-> list of triangles, each a list of 3 edge refs (wound low->high)."""
    low = [tet[m] for m in range(4) if (pattern >> m) & 1]      # insertion order
    high = [tet[m] for m in range(4) if not (pattern >> m) & 1]
    if not low or not high:
        return []
    least, most = (low, high) if len(low) <= len(high) else (high, low)
    tris = []
    if len(least) == 1:
        a = least[0]
        b, c, d = most
        tris.append([(a, b), (a, c), (a, d)])
    else:
        a, b = least
        c, d = most
        if variant:            # exactly one of the two 2-sets iterates in swapped order
            a, b = b, a
        tris.append([(a, d), (a, c), (b, c)])
        tris.append([(a, d), (b, d), (b, c)])
    out = []
    for tri in tris:
        # each pair joins a low and a high corner; normalise pair to (low, high)
        pairs = [(p, q) if p in low else (q, p) for (p, q) in tri]
        pairs = wind(pairs, low, high)
        out.append([edge_ref(p, q) for (p, q) in pairs])
    return out


def pack_tris(tris):
    word = len(tris) << 30
    for n, tri in enumerate(tris):
        t = tri[0] | (tri[1] << 5) | (tri[2] << 10)
        word |= t << (15 * n)
    return word


# --- 4-D ------------------------------------------------------------------
# corner index c = 8*di + 4*dj + 2*dk + dl  (pentatopes.py:28-30);
# pentatope n = monotone path for the n-th permutation of itertools.permutations(range(4))
# where permutation entry "index" sets coordinate axis "index" (pentatopes.py:15-26).
def pentatopes():
    out = []
    for perm in itertools.permutations(range(4)):
        v = [0, 0, 0, 0]
        path = [0]
        for axis in perm:
            v[axis] = 1
            path.append(8 * v[0] + 4 * v[1] + 2 * v[2] + v[3])
        out.append(path)
    return out


def edge_ref4(c1, c2):
    lo, hi = (c1, c2) if (c1 & c2) == c1 else (c2, c1)
    assert (lo & hi) == lo and lo != hi
    return (lo << 4) | (lo ^ hi)       # 8 bits: owner corner (4) | direction 1..15 (4)


def pent_entry(pent, pattern, perm_id):
    """-> list of tetrahedra (each 4 edge refs).  2-3 case: perm_id selects the iteration
    order of the 2-set (bit 0: swapped) and of the 3-set (perm_id>>1 in 0..5, index into
    itertools.permutations(range(3))) -- reference pentatopes.py:246-291."""
    low = [pent[m] for m in range(5) if (pattern >> m) & 1]
    high = [pent[m] for m in range(5) if not (pattern >> m) & 1]
    if not low or not high:
        return []
    least, most = (low, high) if len(low) <= len(high) else (high, low)
    if len(least) == 1:
        a = least[0]
        return [[edge_ref4(*p) for p in orient_tet(pent, pattern, [(a, x) for x in most])]]
    a, b = least
    if perm_id & 1:
        a, b = b, a
    order3 = list(itertools.permutations(range(3)))[perm_id >> 1]
    c, d, e = (most[order3[0]], most[order3[1]], most[order3[2]])
    ac, ad, ae = (a, c), (a, d), (a, e)
    bc, bd, be = (b, c), (b, d), (b, e)
    tets = [(ac, be, ad, bd), (ac, be, ad, ae), (ac, be, bd, bc)]
    return [[edge_ref4(*p) for p in orient_tet(pent, pattern, t)] for t in tets]


def corner_xyzt(c):
    return [(c >> 3) & 1, (c >> 2) & 1, (c >> 1) & 1, c & 1]


def orient_tet(pent, pattern, tet):
    """the four crossing edges of a tetrahedron in the order that makes det[p1-p0, p2-p0, p3-p0, gradient] POSITIVE for the
    linear interpolant of the pentatope -- whatever the sample values, as long as they have the signs of `pattern` (checked on
    several random assignments).  The march then emits every tetrahedron wound from low to high without looking at the samples:
    tetrahedra with coincident vertices (samples EQUAL to the isovalue), where a determinant computed from the data vanishes,
    get the winding of their non-degenerate neighbours.  (Exact rational arithmetic.)"""
    from fractions import Fraction
    import random
    rng = random.Random(12345 + pattern * 131 + sum(pent))
    sign = None
    for trial in range(4):
        f = {}
        for m, c in enumerate(pent):
            mag = Fraction(rng.randint(1, 1000), rng.randint(1, 1000)) + 1
            f[c] = -mag if (pattern >> m) & 1 else mag
        # gradient: the path sets one axis per step
        g = [Fraction(0)] * 4
        for m in range(4):
            a, b = corner_xyzt(pent[m]), corner_xyzt(pent[m + 1])
            axis = [k for k in range(4) if a[k] != b[k]][0]
            g[axis] = f[pent[m + 1]] - f[pent[m]]
        pts = []
        for (c1, c2) in tet:
            t = (0 - f[c1]) / (f[c2] - f[c1])
            p1, p2 = corner_xyzt(c1), corner_xyzt(c2)
            pts.append([Fraction(p1[k]) + t * (p2[k] - p1[k]) for k in range(4)])
        rows = [[pts[k][a] - pts[0][a] for a in range(4)] for k in (1, 2, 3)] + [g]
        d = det4(rows)
        assert d != 0
        sg = 1 if d > 0 else -1
        assert sign is None or sign == sg, "orientation depends on the sample values?"
        sign = sg
    tet = list(tet)
    if sign < 0:
        tet[2], tet[3] = tet[3], tet[2]
    return tet


def det4(m):
    def det3(r):
        return (r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) +
                r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]))
    total = 0
    for col in range(4):
        minor = [[row[k] for k in range(4) if k != col] for row in m[1:]]
        total += (-1) ** col * m[0][col] * det3(minor)
    return total


def main(out_path):
    L = []
    a = L.append
    a("// GENERATED by tools/gen_tables.py -- do not edit.")
    a("// Lookup tables of the Kuhn 6-tetrahedra voxel march and the 24-pentatope hypercube march.")
    a("// Table semantics and the reference lines they restate: see tools/gen_tables.py.")
    a("#pragma once")
    a("#include <stdint.h>")
    a("")
    a("// tet vertex m of tet t -> cube corner (reference order, tetrahedral.py:32-39)")
    a("#define CX_TET_CORNERS_INIT { \\")
    for t in TETS:
        a("  {%d,%d,%d,%d}, \\" % tuple(t))
    a("}")
    a("")
    a("// the 19 voxel edges: {owner corner c1, direction d}; edge id = index into this list")
    a("#define CX_EDGES_INIT { \\")
    a("  " + ",".join("{%d,%d}" % e for e in EDGES) + " \\")
    a("}")
    a("")
    a("// [tet][pattern][variant] : bits 0..14 triangle 0 (3 x 5-bit edge ids), 15..29 triangle 1, 30..31 count")
    a("#define CX_TET_TRIS_INIT { \\")
    ntri_by_mask = []
    for t in TETS:
        rows = []
        for p in range(16):
            rows.append("{0x%xu,0x%xu}" % (pack_tris(tet_entry(t, p, 0)), pack_tris(tet_entry(t, p, 1))))
        a("  {" + ",".join(rows) + "}, \\")
    a("}")
    a("")
    a("// [tet][pattern][variant][triangle] : 3 x 6 bits (owner corner c1 | direction d << 3) of the triangle's")
    a("// voxel edges, wound like CX_TET_TRIS_INIT; bits 30..31 of both words: triangle count of the entry")
    a("#define CX_TET_TRIS_CD_INIT { \\")
    for t in TETS:
        rows = []
        for p in range(16):
            ent = []
            for v in range(2):
                tris = tet_entry(t, p, v)
                words = []
                for n in range(2):
                    w = len(tris) << 30
                    if n < len(tris):
                        for s_, eid in enumerate(tris[n]):
                            c1, d = EDGES[eid]
                            w |= (c1 | (d << 3)) << (6 * s_)
                    words.append("0x%xu" % w)
                ent.append("{" + ",".join(words) + "}")
            rows.append("{" + ",".join(ent) + "}")
        a("  {" + ",".join(rows) + "}, \\")
    a("}")
    a("")
    # per-voxel triangle count (no tolerance skips): sign mask bit c set <=> corner c low
    for mask in range(256):
        n = 0
        for t in TETS:
            p = sum((((mask >> t[m]) & 1) << m) for m in range(4))
            n += len(tet_entry(t, p, 0))
        ntri_by_mask.append(n)
    a("// triangles emitted by a voxel with corner sign mask m (bit c set <=> f(corner c) < value), no tolerance skips")
    a("#define CX_VOXEL_NTRI_INIT { \\")
    for r in range(0, 256, 32):
        a("  " + ",".join(str(x) for x in ntri_by_mask[r:r + 32]) + ", \\")
    a("}")
    a("")
    with open(out_path, "w") as f:
        f.write("\n".join(L) + "\n")
    print("wrote", out_path, "max ntri/voxel", max(ntri_by_mask))
    # 4-D tables go to their own header (large)
    out4 = out_path.replace("cx_tables.h", "cx_tables4d.h")
    L = []
    a = L.append
    a("// GENERATED by tools/gen_tables.py -- do not edit.")
    a("#pragma once")
    a("#include <stdint.h>")
    a("")
    P = pentatopes()
    a("// pentatope vertex m of pentatope n -> hypercube corner (pentatopes.py:15-30)")
    a("#define CX_PENT_CORNERS_INIT { \\")
    for p in P:
        a("  {%d,%d,%d,%d,%d}, \\" % tuple(p))
    a("}")
    a("")
    a("// [pentatope][pattern(32)][perm_id(12)] : up to 3 tetrahedra x 4 edge refs (8 bit each) = 96 bits in 2 x u64,")
    a("// word0 = tets 0,1 ; word1 bits 0..31 = tet 2, bits 32..33 = count")
    a("#define CX_PENT_TETS_INIT { \\")
    for p in P:
        rows = []
        for pat in range(32):
            cols = []
            for perm_id in range(12):
                tets = pent_entry(p, pat, perm_id)
                w0 = w1 = 0
                for n, tet in enumerate(tets):
                    word = tet[0] | (tet[1] << 8) | (tet[2] << 16) | (tet[3] << 24)
                    if n == 0:
                        w0 |= word
                    elif n == 1:
                        w0 |= word << 32
                    else:
                        w1 |= word
                w1 |= len(tets) << 32
                cols.append("{0x%xULL,0x%xULL}" % (w0, w1))
            rows.append("{" + ",".join(cols) + "}")
        a("  {" + ",".join(rows) + "}, \\")
    a("}")
    a("")
    # the same table FACTORED (small enough for LDS): the tetrahedra of a pattern / permutation in pentatope-LOCAL terms
    # (edges as pairs of local vertices 0..4, 10 of them), a per-pentatope map local edge -> edge ref, and the parity of
    # the pentatope's axis permutation, which flips the orientation (swap of the last two refs)
    pairs = [(x, y) for x in range(5) for y in range(x + 1, 5)]
    pair_id = dict((pr, k) for k, pr in enumerate(pairs))
    ref_local = {}
    for n, p in enumerate(P):
        for (x, y) in pairs:
            ref_local[(n, edge_ref4(p[x], p[y]))] = pair_id[(x, y)]
    def local_words(n, pat, perm_id):
        return [[ref_local[(n, r)] for r in tet] for tet in pent_entry(P[n], pat, perm_id)]
    base = [[local_words(0, pat, perm_id) for perm_id in range(12)] for pat in range(32)]
    parity = []
    for n in range(len(P)):
        flips = set()
        for pat in range(32):
            for perm_id in range(12):
                got = local_words(n, pat, perm_id)
                assert len(got) == len(base[pat][perm_id])
                for tg, tb in zip(got, base[pat][perm_id]):
                    if tg == tb:
                        flips.add(0)
                    else:
                        assert tg == [tb[0], tb[1], tb[3], tb[2]], (n, pat, perm_id, tg, tb)
                        flips.add(1)
        assert len(flips) == 1, (n, flips)
        parity.append(flips.pop())
    a("// [pattern(32)][perm_id(12)]: up to 3 tetrahedra x 4 LOCAL edges, an edge = x | y << 2 with x < y the local vertices")
    a("// (5 bits), 20 bits per tetrahedron, bits 60..61 = count; oriented for pentatope 0")
    a("#define CX_PENT_LOCAL_INIT { \\")
    for pat in range(32):
        cols = []
        for perm_id in range(12):
            w = 0
            for k, tet in enumerate(base[pat][perm_id]):
                for s_, e in enumerate(tet):
                    x, y = pairs[e]
                    w |= (x | (y << 2)) << (20 * k + 5 * s_)
            w |= len(base[pat][perm_id]) << 60
            cols.append("0x%xULL" % w)
        a("  {" + ",".join(cols) + "}, \\")
    a("}")
    a("// [pentatope][local edge] -> edge ref (owner corner << 4 | direction)")
    a("#define CX_PENT_EDGE_REF_INIT { \\")
    for n, p in enumerate(P):
        a("  {" + ",".join("0x%x" % edge_ref4(p[x], p[y]) for (x, y) in pairs) + "}, \\")
    a("}")
    a("// bit n: pentatope n has the orientation opposite to pentatope 0 (swap the last two refs of every tetrahedron)")
    a("#define CX_PENT_FLIP_MASK 0x%xu" % sum(b << n for n, b in enumerate(parity)))
    a("")
    with open(out4, "w") as f:
        f.write("\n".join(L) + "\n")
    print("wrote", out4)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    default = os.path.join(here, "..", "contourist_amd", "csrc", "cx_tables.h")
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.normpath(default))
