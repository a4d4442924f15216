"""CPU: rank 0's part of the sharded Level 1 (distributed.merge_shard_components): labels of the same triangle seen from two
slabs name one component; per component the start triangle is the candidate with the largest (x, vertex edge id, |normal_x|)
(surface_geometry.py:79-94, ties by edge id) and its sign decides the flip (:99-103).  Synthetic lists, no GPU."""
import numpy as np

from contourist_amd import distributed as cd


def lists_of(tri, cand):
    "tri: [(k0,k1,k2,label,cls)], cand: [(label,x,vkey,nx,neg,has)]"
    tri = np.array(tri, dtype=np.int64).reshape(-1, 5)
    cand = np.array(cand, dtype=np.float64).reshape(-1, 6)
    return dict(tri_keys=tri[:, :3], tri_label=tri[:, 3].astype(np.uint32), tri_class=tri[:, 4].astype(np.uint8),
                cand_label=cand[:, 0].astype(np.uint32), cand_x=cand[:, 1], cand_vertex_key=cand[:, 2].astype(np.int64),
                cand_nx=cand[:, 3], cand_negative=cand[:, 4].astype(np.uint8), cand_has=cand[:, 5].astype(np.uint8))


def test_chain_over_three_ranks_and_a_separate_component():
    # component X runs through ranks 0-1-2 (labels 7 / 3 and 4 / 9); rank 1 holds it in TWO local pieces (3, 4) that only meet
    # through rank 2; component Y lives on the 0|1 boundary only (labels 20 / 21)
    r0 = lists_of([(10, 11, 12, 7, 2), (13, 14, 15, 7, 4),          # own triangle next to rank 1, copy of rank 1's triangle
                   (50, 51, 52, 20, 2), (53, 54, 55, 20, 4)],
                  [(7, 5.0, 100, 1.0, 0, 1), (20, 4.0, 400, 1.0, 1, 1)])
    r1 = lists_of([(10, 11, 12, 3, 3), (13, 14, 15, 3, 1),
                   (30, 31, 32, 4, 2), (33, 34, 35, 4, 4), (36, 37, 38, 3, 2), (39, 40, 41, 3, 4),
                   (50, 51, 52, 21, 3), (53, 54, 55, 21, 1)],
                  [(3, 9.0, 200, 2.0, 0, 1), (4, 9.5, 210, 0.5, 0, 1), (21, 4.5, 410, 1.0, 0, 1)])
    r2 = lists_of([(30, 31, 32, 9, 3), (33, 34, 35, 9, 1), (36, 37, 38, 9, 3), (39, 40, 41, 9, 1)],
                  [(9, 14.0, 300, 3.0, 1, 1)])
    out, stats = cd.merge_shard_components([r0, r1, r2])
    assert stats["unmatched"] == 0 and stats["components"] == 2 and stats["pairs"] == 8
    flips = [dict(zip(l.tolist(), f.tolist())) for l, f in out]
    # X: the candidate of rank 2 has the largest x (14.0) and a negative normal -> everything of X flips, on every rank
    assert flips[0][7] == 1 and flips[1][3] == 1 and flips[1][4] == 1 and flips[2][9] == 1
    # Y: rank 1's candidate (x = 4.5) wins over rank 0's (4.0): no flip
    assert flips[0][20] == 0 and flips[1][21] == 0


def test_ties_go_to_the_larger_edge_id_then_the_larger_normal():
    a = lists_of([(1, 2, 3, 0, 2), (4, 5, 6, 0, 4)], [(0, 8.0, 77, 1.0, 1, 1)])
    b = lists_of([(1, 2, 3, 5, 3), (4, 5, 6, 5, 1)], [(5, 8.0, 78, 0.1, 0, 1)])
    out, _ = cd.merge_shard_components([a, b])
    assert out[0][1].tolist() == [0] and out[1][1].tolist() == [0]          # same x: vertex 78 > 77 decides
    b = lists_of([(1, 2, 3, 5, 3), (4, 5, 6, 5, 1)], [(5, 8.0, 77, 0.1, 0, 1)])
    out, _ = cd.merge_shard_components([a, b])
    assert out[0][1].tolist() == [1] and out[1][1].tolist() == [1]          # same vertex from both sides: |normal_x| 1.0 > 0.1


def test_a_triangle_only_one_side_knows_is_reported():
    a = lists_of([(1, 2, 3, 0, 2)], [(0, 1.0, 1, 1.0, 0, 1)])
    b = lists_of([(1, 2, 4, 5, 3)], [(5, 2.0, 2, 1.0, 0, 0)])
    out, stats = cd.merge_shard_components([a, b])
    assert stats["unmatched"] == 2 and stats["pairs"] == 0
    assert out[0][0].tolist() == [0] and out[1][0].tolist() == []            # a component without any candidate gets no answer


def test_layout_covers_every_cell_once():
    for n0 in (16, 41, 512):
        for world in (1, 2, 3, 5, 8):
            if n0 // world < 3:
                continue
            owned = []
            for r in range(world):
                lay = cd.shard_layout(n0, world, r)
                owned += list(range(lay["e0"] + lay["own_lo"], lay["e0"] + lay["own_hi"]))
                assert lay["e1"] - lay["e0"] - 1 >= lay["own_hi"] and lay["own_lo"] == (cd.SHARD_LAYERS if r else 0)
            assert owned == list(range(n0 - 1))
