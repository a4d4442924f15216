"""oracle/contour2d.py -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's 2-D contour path).

Only tests/, __graft_entry__.smoke() and bench cpu_baseline legs may import this module; the product
(contourist_amd) never does.  Parity pinned: tests/golden2d/*.npz were produced by the real reference
(oracle/make_goldens2d.py) and tests/test_oracle2d_vs_golden.py checks this restatement against them.

Restated from contourist/triangulated.py (file:line of the reference):
    adjacent_offsets                          :10-12    the six neighbours of a lattice point (the lattice is
                                                        triangulated with the (1,1) diagonal)
    adjacent_pairs                            :63-75    the <= 4 pairs that share a triangle and an end point
    Grid2DContour.contour_pair_interpolation  :339-353  pair (low, high) exists iff f(low) <= z <= f(high);
                                                        ratio = (z-flow)/(fhigh-flow), 0.5 if |den| <= 1e-8
    find_all_adjacent_contour_pairs           :355-381  pairs around a point in one role; neighbours must be in
                                                        range 0 <= p < (n, m) (:333-334), the point itself is not checked
    find_initial_contour_pairs                :299-320  end point pairs are bisected until adjacent
    expand_contour_pairs                      :322-331  breadth-first growth over shared (point, role)
    get_contour_sequences                     :226-297  walk over adjacencies, consecutive np.allclose points dropped
and contourist/multiple_2d_contour.py:
    classify_endpoint_values                  :48-59    levels crossed by a segment: bisect_left(f_start) .. bisect_right(f_end)

The reference pops its work sets in CPython set order; this restatement pops the smallest element.  For
fields where no sample equals an isovalue every pair has at most two adjacencies and the result (as a set of
polylines modulo rotation/reversal) does not depend on that order.  `rule="build"` classifies samples equal to
the isovalue as high only (f < z is low, like the 3-D march), which is what the HIP path implements; after the
consecutive-duplicate drop it yields the same polylines as the reference rule when a tie vertex has one low
and one high arc of neighbours.
"""
import bisect

import numpy as np

ADJACENT_OFFSETS = [(0, 1), (1, 1), (1, 0), (0, -1), (-1, -1), (-1, 0)]   # triangulated.py:10-12


def _close(a, b):
    "np.allclose(a, b) for 2-vectors: |a-b| <= 1e-8 + 1e-5*|b| in every coordinate"
    return all(abs(x - y) <= 1e-8 + 1e-5 * abs(y) for x, y in zip(a, b))


class Lattice(object):
    def __init__(self, A, z, rule="reference"):
        self.A = np.asarray(A)
        self.n, self.m = self.A.shape
        self.z = float(z)
        self.rule = rule

    def f(self, p):
        return float(self.A[p[0], p[1]])

    def in_range(self, p):   # :333-334
        return 0 <= p[0] < self.n and 0 <= p[1] < self.m

    def interpolation(self, low, high):   # :339-353
        flow, fhigh = self.f(low), self.f(high)
        ok = (flow <= self.z <= fhigh) if self.rule == "reference" else (flow < self.z <= fhigh)
        if not ok:
            return None
        ratio = 0.5
        den = 1.0 * (fhigh - flow)
        if not abs(den) <= 1e-8:
            ratio = (self.z - flow) / den
        return (low[0] + ratio * (high[0] - low[0]), low[1] + ratio * (high[1] - low[1]))

    def around(self, location, is_low):   # :355-381
        out = {}
        for (di, dj) in ADJACENT_OFFSETS:
            q = (location[0] + di, location[1] + dj)
            if not self.in_range(q):
                continue
            pair = (location, q) if is_low else (q, location)
            p = self.interpolation(*pair)
            if p is not None:
                out[pair] = p
        return out

    def all_pairs(self):
        "every pair of the lattice (what the growth reaches when every crossing is a seed)"
        out = {}
        for i in range(self.n):
            for j in range(self.m):
                out.update(self.around((i, j), True))
        return out

    def search_grid(self):   # :198-212
        "the seeds of the exhaustive search: crossing axis edges that start at i < n-1, j < m-1"
        out = []
        for i in range(self.n - 1):
            for j in range(self.m - 1):
                for p1 in ((i + 1, j), (i, j + 1)):
                    if self._straddle((i, j), p1) is not None:
                        out.append(((i, j), p1))
                    elif self._straddle(p1, (i, j)) is not None:
                        out.append((p1, (i, j)))
        return out

    def seed_points(self, low, high):   # :299-320 (the bisection)
        low, high = tuple(int(x) for x in low), tuple(int(x) for x in high)
        if self._straddle(low, high) is None:
            low, high = high, low
            assert self._straddle(low, high) is not None, "bad end points " + repr((low, high))
        while max(abs(low[0] - high[0]), abs(low[1] - high[1])) > 1:
            mid = ((low[0] + high[0]) // 2, (low[1] + high[1]) // 2)
            if self._straddle(low, mid) is not None:
                high = mid
            else:
                assert self._straddle(mid, high) is not None
                low = mid
        return low, high

    def _straddle(self, low, high):
        # the reference calls contour_pair_interpolation here; only "is not None" is used.  Seeds are chosen with
        # the reference's non-strict test under both rules (the build does the same, csrc/cx_contour2d.hip)
        flow, fhigh = self.f(low), self.f(high)
        return True if (flow <= self.z <= fhigh) else None

    def _seed_pairs(self, low, high):
        """pairs around the two seed points (:316-317).  Build rule: a seed point equal to the isovalue is a high
        point, whichever role the reference's test gave it."""
        low_is_low = True if self.rule == "reference" else (self.f(low) < self.z)
        new = self.around(low, low_is_low)
        new.update(self.around(high, False))
        return new

    def grow(self, end_points):   # :299-331
        found = {}
        horizon = set()
        for (low, high) in end_points:
            low, high = self.seed_points(low, high)
            new = self._seed_pairs(low, high)
            assert len(new) > 0 or self.rule != "reference"
            found.update(new)
            horizon.update(new)
        while horizon:
            new = {}
            for (low, high) in horizon:
                new.update(self.around(low, True))
                new.update(self.around(high, False))
            horizon = set(p for p in new if p not in found)
            found.update(new)
        return found


def adjacent_pairs(low, high):   # :63-75
    n = len(ADJACENT_OFFSETS)
    li = ADJACENT_OFFSETS.index((low[0] - high[0], low[1] - high[1]))
    hi = ADJACENT_OFFSETS.index((high[0] - low[0], high[1] - low[1]))
    for (hshift, lshift) in [(-1, 0), (1, 0), (0, -1), (0, 1)]:
        ol = ADJACENT_OFFSETS[(li + lshift) % n]
        oh = ADJACENT_OFFSETS[(hi + hshift) % n]
        yield ((high[0] + ol[0], high[1] + ol[1]), (low[0] + oh[0], low[1] + oh[1]))


def sequences(interpolated, dedupe=True):   # :226-297, work sets popped smallest-first
    "[(closed, [(pair, point), ...]), ...] from {pair: point}"
    adjacencies = {}
    for pair in interpolated:
        adjacencies[pair] = sorted(p for p in adjacent_pairs(*pair) if p in interpolated)
    edge_pairs = set(p for p in adjacencies if len(adjacencies[p]) < 2)
    unvisited = set(interpolated)
    highs, lows = {}, {}
    for pair in unvisited:
        highs.setdefault(pair[1], set()).add(pair)
        lows.setdefault(pair[0], set()).add(pair)

    def remove_pair(p):
        unvisited.discard(p)
        edge_pairs.discard(p)
    contours = []
    while unvisited:
        chain = []
        if edge_pairs:
            pair = min(edge_pairs)
            edge_pairs.discard(pair)
            unvisited.discard(pair)
            closed = False
        else:
            pair = min(unvisited)
            unvisited.discard(pair)
            closed = True
        while pair is not None:
            (low, high) = pair
            for hp in list(highs.get(low, [])):
                remove_pair(hp)
            for lp in list(lows.get(high, [])):
                remove_pair(lp)
            p = interpolated[pair]
            if len(chain) == 0 or not dedupe or not _close(chain[-1][1], p):
                chain.append((pair, p))
            nxt = None
            for adjacent in adjacencies[pair]:
                if adjacent in unvisited:
                    remove_pair(adjacent)
                    nxt = adjacent
                    break
            pair = nxt
        if _close(chain[0][1], chain[-1][1]):
            closed = True
        contours.append((closed, chain))
    return contours


def levels_of_segment(values, f_start, f_end, rule="reference"):   # multiple_2d_contour.py:48-59
    if f_end < f_start:
        f_start, f_end = f_end, f_start
    if rule == "reference":
        return range(bisect.bisect_left(values, f_start), bisect.bisect_right(values, f_end))
    return range(bisect.bisect_right(values, f_start), bisect.bisect_right(values, f_end))


def contours(A, value, end_points=None, rule="reference", dedupe=True):
    """[(closed, points[k,2] float64 grid coordinates, pairs[k,4] int), ...] for one isovalue.
    end_points None: the reference's grid search then its seeded growth; "all": every pair of the lattice;
    else the seeded growth from the given end points."""
    L = Lattice(A, value, rule)
    if isinstance(end_points, str) and end_points == "all":
        found = L.all_pairs()
    else:
        seeds = L.search_grid() if end_points is None else end_points
        found = L.grow(seeds) if len(seeds) else {}
    out = []
    for closed, chain in sequences(found, dedupe):
        pts = np.array([p for (_, p) in chain], dtype=np.float64).reshape(-1, 2)
        keys = np.array([pair[0] + pair[1] for (pair, _) in chain], dtype=np.int64).reshape(-1, 4)
        out.append((closed, pts, keys))
    return out


def unpack_golden(z):
    "[(value, [(closed, points), ...]), ...] from a tests/golden2d file"
    out = [(float(v), []) for v in z["values"]]
    for c in range(len(z["level"])):
        out[int(z["level"][c])][1].append((bool(z["closed"][c]), z["points"][z["offsets"][c]:z["offsets"][c + 1]]))
    return out


def canonical_keys(contour_list):
    "order-free form of the pair sequences: [(closed, ((li,lj,hi,hj), ...)), ...] rotated/reversed to the smallest pair first"
    out = []
    for item in contour_list:
        closed, rows = bool(item[0]), [tuple(int(x) for x in r) for r in np.asarray(item[2]).reshape(-1, 4)]
        if closed and len(rows) > 2:
            k = rows.index(min(rows))
            fwd = rows[k:] + rows[:k]
            rows = min(fwd, [fwd[0]] + fwd[1:][::-1])
        elif rows and rows[-1] < rows[0]:
            rows = rows[::-1]
        out.append((closed, tuple(rows)))
    return sorted(out)


def drop_close_to_previous(points):
    "the build's duplicate rule: a point np.allclose to its predecessor on the polyline is dropped"
    points = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    keep = [True] + [not _close(points[k - 1], points[k]) for k in range(1, len(points))]
    return np.array(keep, dtype=bool)


def canonical(contour_list, decimals=7):
    """order-free form of [(closed, points), ...]: every polyline rotated/reversed to start at its smallest
    point, the list sorted; points rounded to `decimals`."""
    out = []
    for item in contour_list:
        closed, pts = bool(item[0]), np.round(np.asarray(item[1], dtype=np.float64).reshape(-1, 2), decimals) + 0.0
        rows = [tuple(r) for r in pts.tolist()]
        if closed and len(rows) > 1 and rows[0] == rows[-1]:
            rows = rows[:-1]
        if closed and len(rows) > 2:
            k = rows.index(min(rows))
            fwd = rows[k:] + rows[:k]
            bwd = [fwd[0]] + fwd[1:][::-1]
            rows = min(fwd, bwd)
        elif rows and rows[-1] < rows[0]:
            rows = rows[::-1]
        out.append((closed, tuple(rows)))
    return sorted(out)
