"""CPU: the demo outputs the reference itself COMMITTED (misc/centered.js, sphere.html, torus.html, wave.html, hyperbola.html:
emit_three_json / grid_html_page of its own demo calls under Python 2; numbers extracted by oracle/make_goldens_py2_demos.py into
tests/golden_demos/py2_*.npz) against the fixtures today's checkout produces under Python 3 (tests/golden_demos/*.npz,
oracle/make_goldens_demos.py): an independent pin of the Level-1 API -- order-independent quantities only (counts, vertex
sets), because which quad diagonal a 2-2 tetrahedron gets is the hash order of the interpreter that ran it.

Revision drift found and recorded here:
* sphere / hyperbola: the committed vertex LISTS carry 61 / 82 repeated points (an earlier extract_surface_geometry numbered
  them per use); as SETS they are exactly today's points, and the triangle counts are equal;
* torus: identical counts and identical points;
* wave: the committed page is test_wave(side=6, scale=0.2) -- recovered from its own points (the field is linear in z, so every
  crossing on a z edge satisfies z = 1.1 + sin(((x-6)^2 + (y-6)^2) * 0.2) to 1e-14), not today's default (side=20, scale=0.02);
  with those parameters today's checkout returns the same 1023 points and 1900 triangles, 1474 of them the same triangles
  (the other 426 are the other diagonal of a quad);
* centered: the committed file is the SEEDED call of test_json (end points (0,0,0)-(100,100,100)); today's constructor
  asserts on 3-D end points (triangulated.py:96), so the Python-3 fixture is the exhaustive search instead (more sheets).
  6 of its 1010 points (12 triangles) lie in voxels one step outside the grid along +y: an earlier in_range let the growth
  step there; today's `point < corner` does not (tests/test_gpu_demos.py compares the device with the remaining 1004)."""
import os

import numpy as np
import pytest

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_demos")


def point_set(P, T=None, digits=9):
    P = np.asarray(P, dtype=np.float64)
    if T is not None:
        P = P[np.unique(np.asarray(T).reshape(-1))]
    return set(map(tuple, np.round(P, digits).tolist()))


@pytest.mark.parametrize("name", ["sphere", "torus", "hyperbola"])
def test_committed_output_equals_todays_checkout(name):
    old = np.load(os.path.join(GD, "py2_" + name + ".npz"))
    new = np.load(os.path.join(GD, name + ".npz"))
    assert len(old["triangles"]) == len(new["triangles"])
    a, b = point_set(old["points"], old["triangles"]), point_set(new["points"], new["triangles"])
    assert a == b
    assert len(b) == len(new["points"])                 # today's list has no repeated point
    if name == "torus":
        assert len(old["points"]) == len(new["points"])


def test_committed_wave_parameters_recovered():
    "every crossing of the committed wave on a z edge lies on z = 1.1 + sin(((x - 6)^2 + (y - 6)^2) * 0.2)"
    P = np.load(os.path.join(GD, "py2_wave.npz"))["points"]
    on_z = (np.abs(P[:, 0] - np.round(P[:, 0])) < 1e-12) & (np.abs(P[:, 1] - np.round(P[:, 1])) < 1e-12) & (np.abs(P[:, 2] - np.round(P[:, 2])) > 1e-9)
    assert on_z.sum() > 100
    z = 1.1 + np.sin(((P[on_z, 0] - 6) ** 2 + (P[on_z, 1] - 6) ** 2) * 0.2)
    assert np.max(np.abs(z - P[on_z, 2])) < 1e-12
    assert P[:, :2].max() == 12.0 and P[:, :2].min() == 0.0


def test_committed_centered_is_a_subset_of_the_exhaustive_search():
    "the seeded call reaches some of the sheets of the exhaustive one: (almost) all of its points are among today's"
    old = point_set(np.load(os.path.join(GD, "py2_centered.npz"))["points"], digits=6)
    new = point_set(np.load(os.path.join(GD, "centered.npz"))["points"], digits=6)
    assert len(old & new) >= 0.99 * len(old)
