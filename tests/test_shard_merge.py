"""CPU: rank 0's part of the sharded Level 1 (distributed.merge_shard_components): labels of the same triangle seen from two
slabs name one component; per component the start triangle is the candidate with the largest (x, vertex edge id, |normal_x|)
(surface_geometry.py:79-94, ties by edge id) and its sign decides the flip (:99-103).  Synthetic lists, no GPU."""
import numpy as np

from contourist_amd import distributed as cd


def small(pairs, cand, unmatched=0):
    "pairs: [(my label, lower neighbour's label)], cand: [(label, x, vkey, nx, neg, has)]"
    cand = np.array(cand, dtype=np.float64).reshape(-1, 6)
    return dict(pairs=np.array(pairs, dtype=np.int64).reshape(-1, 2), unmatched=unmatched,
                cand_label=cand[:, 0].astype(np.uint32), cand_x=cand[:, 1], cand_vertex_key=cand[:, 2].astype(np.int64),
                cand_nx=cand[:, 3], cand_negative=cand[:, 4].astype(np.uint8), cand_has=cand[:, 5].astype(np.uint8))


def test_chain_over_three_ranks_and_a_separate_component():
    # component X runs through ranks 0-1-2 (labels 7 / 3 and 4 / 9); rank 1 holds it in TWO local pieces (3, 4) that only meet
    # through rank 2; component Y lives on the 0|1 boundary only (labels 20 / 21)
    r0 = small([], [(7, 5.0, 100, 1.0, 0, 1), (20, 4.0, 400, 1.0, 1, 1)])
    r1 = small([(3, 7), (21, 20)], [(3, 9.0, 200, 2.0, 0, 1), (4, 9.5, 210, 0.5, 0, 1), (21, 4.5, 410, 1.0, 0, 1)])
    r2 = small([(9, 3), (9, 4)], [(9, 14.0, 300, 3.0, 1, 1)])
    out, stats = cd.merge_shard_components([r0, r1, r2])
    assert stats["unmatched"] == 0 and stats["components"] == 2 and stats["pairs"] == 4
    flips = [dict(zip(l.tolist(), f.tolist())) for l, f in out]
    # X: the candidate of rank 2 has the largest x (14.0) and a negative normal -> everything of X flips, on every rank
    assert flips[0][7] == 1 and flips[1][3] == 1 and flips[1][4] == 1 and flips[2][9] == 1
    # Y: rank 1's candidate (x = 4.5) wins over rank 0's (4.0): no flip
    assert flips[0][20] == 0 and flips[1][21] == 0


def test_ties_go_to_the_larger_edge_id_then_the_larger_normal():
    a = small([], [(0, 8.0, 77, 1.0, 1, 1)])
    b = small([(5, 0)], [(5, 8.0, 78, 0.1, 0, 1)])
    out, _ = cd.merge_shard_components([a, b])
    assert out[0][1].tolist() == [0] and out[1][1].tolist() == [0]          # same x: vertex 78 > 77 decides
    b = small([(5, 0)], [(5, 8.0, 77, 0.1, 0, 1)])
    out, _ = cd.merge_shard_components([a, b])
    assert out[0][1].tolist() == [1] and out[1][1].tolist() == [1]          # same vertex from both sides: |normal_x| 1.0 > 0.1


def test_components_without_a_candidate_and_reported_mismatches():
    a = small([], [(0, 1.0, 1, 1.0, 0, 1)], unmatched=2)
    b = small([], [(5, 2.0, 2, 1.0, 0, 0)], unmatched=1)          # a component made of copies only: no own triangle, no candidate
    out, stats = cd.merge_shard_components([a, b])
    assert stats["unmatched"] == 3 and stats["pairs"] == 0
    assert out[0][0].tolist() == [0] and out[1][0].tolist() == []            # ... and no answer for it


def test_pair_labels_lines_the_two_lists_up():
    "a rank's own boundary triangles against the neighbour's copies of them: any order, distinct label pairs, strangers counted"
    import torch
    rng = np.random.RandomState(3)
    h = torch.from_numpy(rng.randint(-2 ** 62, 2 ** 62, size=1000).astype(np.int64))
    own_label = torch.from_numpy((np.arange(1000) % 7).astype(np.int32))
    copy_label = torch.from_numpy((np.arange(1000) % 7 + 100).astype(np.int32))
    perm = torch.from_numpy(rng.permutation(1000))
    pairs, unmatched = cd.pair_labels(h, own_label, h[perm], copy_label[perm])
    assert unmatched == 0 and sorted(map(tuple, pairs.tolist())) == [(k, k + 100) for k in range(7)]
    # one triangle the neighbour does not know, one copy of a triangle this rank does not own
    pairs, unmatched = cd.pair_labels(h, own_label, torch.cat([h[perm][:-1], torch.tensor([12345], dtype=torch.int64)]), copy_label[perm])
    assert unmatched == 2 and len(pairs) == 7
    pairs, unmatched = cd.pair_labels(h[:0], own_label[:0], h[:5], copy_label[:5])
    assert unmatched == 5 and pairs.shape == (0, 2)


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
