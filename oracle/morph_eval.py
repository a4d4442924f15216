"""TEST INFRASTRUCTURE -- not part of the product path.

CPU restatement of the consumer side of the 4-D path: the surface at a time t from morph triangles, as the reference's
viewer builds it (misc/morph_triangles.js).  Plain Python loops over the fixtures' few thousand triangles.

  triangle intervals      misc/morph_triangles.js:53-84   a triangle lives on the COMMON extent of its three segments:
                                                           tr_min = max of the segments' t_low, tr_max = min of their t_high;
                                                           a segment without time extent (t_low == t_high) kills the
                                                           triangle, t_low > t_high is an error; kept if tr_min < tr_max
  active set at a time    :117-147                        triangles with tr_min <= t and tr_max > t  (a triangle whose
                                                           interval starts after t ends the scan: `triangle_min_t > min_t`)
  point on a segment      :156-178 interpolate_points_3d  ratio = (t - e_t) / (l_t - e_t) if l_t - e_t > epsilon else 0.5;
                                                           ratio + epsilon < 0 -> early point, ratio - epsilon > 1 -> late
                                                           point, else early + ratio * (late - early);
                                                           epsilon = 1e-7 * (max_value - min_value)  (:49-50)
  geometry                :179-204                        one vertex per segment of an active triangle, numbered by first use
The viewer evaluates at the start `min_t` of each transition interval; `surface_at` takes that time as `t`.
Parity pin: none of the reference's tests covers the viewer, so the reference ITSELF is run: oracle/make_goldens_viewer.py cuts
the arithmetic lines out of misc/morph_triangles.js (the ranges above) and runs them unchanged under node on the bytes
MorphTriangles.to_json wrote; tests/test_oracle_viewer.py holds this restatement to those results bit for bit (intervals, order,
active sets, first-use numbering, interpolated points at both ends of the interval).  It is what tests/test_gpu_level0_4d.py
compares cx_morph_eval with (so the device is no longer compared with the package's own numpy)."""
import numpy as np


def triangle_intervals(points4d, segments, triangles):
    """-> (tr_min, tr_max, valid) per triangle (morph_triangles.js:53-84)"""
    P = np.asarray(points4d, dtype=np.float64)
    nt = len(triangles)
    tr_min = np.zeros(nt)
    tr_max = np.zeros(nt)
    valid = np.zeros(nt, dtype=bool)
    for i, tri in enumerate(triangles):
        lo_hi = None
        ok = True
        for s in tri:
            a, b = segments[s]
            t_low, t_high = P[a][3], P[b][3]
            if t_low < t_high:
                if lo_hi is None:
                    lo_hi = [t_low, t_high]
                else:
                    if lo_hi[0] < t_low:
                        lo_hi[0] = t_low
                    if lo_hi[1] > t_high:
                        lo_hi[1] = t_high
            else:
                if t_low > t_high:
                    raise ValueError("segment in triangle has negative time dimension.")
                ok = False
                break
        if ok and lo_hi is not None and lo_hi[0] < lo_hi[1]:
            tr_min[i], tr_max[i], valid[i] = lo_hi[0], lo_hi[1], True
    return tr_min, tr_max, valid


def interpolate_points_3d(p_early, p_late, t_value, epsilon):
    "morph_triangles.js:156-178"
    e_t, l_t = p_early[3], p_late[3]
    ratio = 0.5
    diff = l_t - e_t
    if diff > epsilon:
        ratio = (t_value - e_t) * 1.0 / diff
    if ratio + epsilon < 0:
        return [p_early[0], p_early[1], p_early[2]]
    if ratio - epsilon > 1:
        return [p_late[0], p_late[1], p_late[2]]
    return [p_early[i] + ratio * (p_late[i] - p_early[i]) for i in range(3)]


def surface_at(points4d, segments, triangles, t, min_value=None, max_value=None):
    """the geometry the viewer builds for time t: dict(active = triangle indices in the viewer's order (by tr_min, stable),
    segment_ids = segments in order of first use, points = their 3-D points at t, faces = index triples into those)"""
    P = np.asarray(points4d, dtype=np.float64)
    if min_value is None:
        min_value = float(P[:, 3].min())
    if max_value is None:
        max_value = float(P[:, 3].max())
    epsilon = (max_value - min_value) * 1.0 * 1e-7
    tr_min, tr_max, valid = triangle_intervals(P, segments, triangles)
    order = sorted((tr_min[i], i) for i in range(len(triangles)) if valid[i])
    active = []
    for (tmin, i) in order:
        if tmin > t:
            break
        if tr_max[i] > t:
            active.append(i)
    vertex_index_map = {}
    points, faces = [], []
    for i in active:
        face = []
        for s in triangles[i]:
            s = int(s)
            if s not in vertex_index_map:
                a, b = segments[s]
                vertex_index_map[s] = len(points)
                points.append(interpolate_points_3d(P[a], P[b], t, epsilon))
            face.append(vertex_index_map[s])
        faces.append(face)
    seg_ids = sorted(vertex_index_map, key=vertex_index_map.get)
    return dict(active=active, segment_ids=seg_ids, points=np.array(points, dtype=np.float64).reshape(-1, 3),
                faces=np.array(faces, dtype=np.int64).reshape(-1, 3))


def canonical(segment_ids, points, faces):
    "order-independent form: per triangle its three (segment id, point) corners, winding kept up to rotation"
    out = []
    for f in faces:
        ids = [int(segment_ids[k]) for k in f]
        r = ids.index(min(ids))
        rot = [f[(r + k) % 3] for k in range(3)]
        out.append(tuple((int(segment_ids[k]),) + tuple(float(x) for x in points[k]) for k in rot))
    return sorted(out)
