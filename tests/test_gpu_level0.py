"""GPU: Level-0 parity of the HIP march (through the C ABI) with the oracle and with the goldens
produced by the real reference.  Bit-exact for edge sets, orientation and triangle index sets;
vertex coordinates within 1e-6 relative (fp32 device arithmetic vs the reference's float64)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6      # BASELINE.json north_star: vertex coords within 1e-6 relative fp32
ABS_FLOOR = 1e-6    # + absolute floor for coordinates near 0 (in voxel units)


@pytest.fixture(scope="module")
def ctx():
    from contourist_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()


def hip_level0(ctx, A, v, flags):
    ctx.upload_grid(A)
    counts = ctx.extract3d(v, flags)
    xyz, keys, tris = ctx.download_level0(counts)
    return counts, xyz, keys.astype(np.int64), tris.astype(np.int64)


def check_against_oracle(ctx, A, v, flags, diag_mode):
    from oracle import level0
    shape = A.shape
    counts, xyz, keys, tris = hip_level0(ctx, A, v, flags)
    O = level0.march3d(A, v, diag_mode=diag_mode)
    ko = level0.edge_keys_from_pairs(O["pairs"], shape)
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    ch = level0.canonical_level0(keys, xyz, tris)
    assert counts["n_vertices"] == len(ko) and counts["n_triangles"] == len(O["tris"])
    assert counts["n_border_voxels"] == O["nborder_mixed"]     # voxels with a sign change that border_voxel() accepts
    assert len(np.unique(keys)) == len(keys)
    assert np.array_equal(co[0], ch[0])                                  # crossing-edge set, exact
    err = np.abs(ch[1].astype(np.float64) - co[1])
    assert np.all(err <= REL_TOL * np.abs(co[1]) + ABS_FLOOR), err.max()
    assert np.array_equal(co[2], ch[2])                                  # triangle key triples, exact
    # winding: normal must point from low to high, checked geometrically where the triangle has area
    order = np.argsort(ko)
    pos = {int(k): n for n, k in enumerate(ko[order])}
    P = O["xyz"][order]
    pairs = O["pairs"][order].astype(np.float64)
    grad = pairs[:, 3:] - pairs[:, :3]
    idx = np.vectorize(pos.get)(keys[tris]) if len(tris) else np.zeros((0, 3), int)
    p0, p1, p2 = P[idx[:, 0]], P[idx[:, 1]], P[idx[:, 2]]
    n = np.cross(p1 - p0, p2 - p0)
    g = grad[idx[:, 0]] + grad[idx[:, 1]] + grad[idx[:, 2]]
    s = np.einsum("ij,ij->i", n, g)
    area = np.linalg.norm(n, axis=1)
    assert np.all(s[area > 1e-9] > 0)
    return counts


@pytest.mark.parametrize("name", golden_names())
def test_golden_fields_match_oracle_and_reference(ctx, name):
    from contourist_amd import _ffi
    from oracle import level0
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A, v = G["A"], float(G["value"])
    check_against_oracle(ctx, A, v, _ffi.CX_DIAG_CPYTHON310, 1)
    check_against_oracle(ctx, A, v, _ffi.CX_DIAG_CANONICAL, 0)
    # and directly against the reference's own Level-0 snapshot
    counts, xyz, keys, tris = hip_level0(ctx, A, v, _ffi.CX_DIAG_CPYTHON310)
    kr = level0.edge_keys_from_pairs(G["l0_pairs"], A.shape)
    cr = level0.canonical_level0(kr, G["l0_xyz"], G["l0_tris"])
    ch = level0.canonical_level0(keys, xyz, tris)
    assert np.array_equal(cr[0], ch[0]) and np.array_equal(cr[2], ch[2])
    assert np.all(np.abs(ch[1] - cr[1]) <= REL_TOL * np.abs(cr[1]) + ABS_FLOOR)
    assert counts["n_border_voxels"] <= len(G["surface_voxels"])   # == unless samples equal the isovalue exactly


@pytest.mark.parametrize("shape,seed", [((5, 7, 9), 1), ((2, 2, 2), 2), ((2, 9, 3), 3), ((17, 16, 65), 4),
                                        ((33, 31, 64), 5), ((64, 64, 64), 6), ((40, 24, 136), 7)])
def test_random_fields_ragged_shapes(ctx, shape, seed):
    """open-boundary white noise (surface cuts every array face), odd and minimum sizes"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(seed)
    A = rng.standard_normal(shape).astype(np.float32)
    for v in (0.0, 0.3):
        check_against_oracle(ctx, A, v, _ffi.CX_DIAG_CPYTHON310, 1)


@pytest.mark.parametrize("shape,seed", [((6, 5, 4), 11), ((6, 5, 5), 12), ((6, 5, 6), 13), ((6, 5, 7), 14), ((9, 6, 257), 15),
                                        ((5, 9, 258), 16), ((4, 5, 259), 17), ((3, 18, 513), 18), ((20, 20, 255), 19),
                                        ((3, 6, 1024), 20), ((4, 5, 1030), 25)])   # (rows longer than the k * P2 table of the triangle kernels: 64-bit hashes)
def test_rows_not_multiple_of_four(ctx, shape, seed):
    """rows whose length is not a multiple of 4 (16-byte loads only 4-byte aligned, the lane at the end of a row
    shifts its samples into place), one to three k-segments"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(seed)
    A = rng.standard_normal(shape).astype(np.float32)
    check_against_oracle(ctx, A, 0.1, _ffi.CX_DIAG_CPYTHON310, 1)
    check_against_oracle(ctx, A, 0.1, _ffi.CX_DIAG_CANONICAL, 0)


@pytest.mark.parametrize("shape,seed", [((5, 7, 9), 21), ((17, 16, 65), 22), ((33, 31, 64), 23), ((20, 20, 255), 24)])
def test_generic_kernel_on_request(ctx, shape, seed):
    """the shape-agnostic classify kernel (rows shorter than 4 samples need it; CX_KERNEL_GENERIC forces it)"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(seed)
    A = rng.standard_normal(shape).astype(np.float32)
    check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_GENERIC, 1)


def test_generic_kernel_many_reservations(ctx):
    """a grid large enough that the generic kernel's workgroups reserve output ranges in arbitrary order:
    the triangle kernel then meets waves whose cells belong to different reservations"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(31)
    A = rng.standard_normal((72, 70, 68)).astype(np.float32)
    ctx.reserve(400000, 1200000, 2400000)
    for rep in range(3):
        check_against_oracle(ctx, A, 0.8, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_GENERIC, 1)


_LONG_TASKS_CHILD = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
from contourist_amd import _ffi
import test_gpu_level0 as T
ctx = _ffi.Context(0)
rng = np.random.RandomState(41)
A = rng.standard_normal((128, 32, 256)).astype(np.float32)
c = T.check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CPYTHON310, 1)
T.check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CANONICAL, 0)
ctx.close()
print("LONG_TASKS_OK", c["n_cells"], c["n_vertices"], c["n_triangles"])
"""


def test_long_tasks_flush_queue_and_batch_records():
    """A dense surface with the longest streaming tasks the kernel supports: the tuning knob CX_TASKS=2 (honoured only in a
    process STARTED with CX_DEBUG=1, hence the child process) makes every streaming wave cover 15 planes of 4 x 256 cells of
    white noise, i.e. queue ~15 000 cells: the LDS-staged queue (1024 entries) is flushed about fifteen times per wave, the
    wave closes ~30 batches, and the vertex stage's waves start inside batches (cx_skip_rounds).  Checked against the oracle
    exactly like every other case (both diagonal modes)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CX_DEBUG="1", CX_TASKS="2")
    r = subprocess.run([sys.executable, "-c", _LONG_TASKS_CHILD.format(root=root, tests=os.path.join(root, "tests"))],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("LONG_TASKS_OK")]
    assert line, r.stdout[-2000:]
    n_cells = int(line[0].split()[1])
    assert n_cells > 900000      # ~all 127 x 31 x 255 voxels (+ boundary cells) are active: the queues were long


_ENTRY_KERNEL_CHILD = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
from contourist_amd import _ffi
import test_gpu_level0 as T
ctx = _ffi.Context(0)
rng = np.random.RandomState(91)
n = 0
for shape, v in (((33, 31, 64), 0.0), ((20, 20, 255), 0.3), ((64, 48, 260), -0.2), ((9, 40, 12), 0.1)):
    A = rng.standard_normal(shape).astype(np.float32)
    for _ in range(2):
        for ax in range(3):
            A = (0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)).astype(np.float32)
    A = (A / A.std()).astype(np.float32)
    n += T.check_against_oracle(ctx, A, v, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_STAGED, 1)["n_triangles"]
    T.check_against_oracle(ctx, A, v, _ffi.CX_DIAG_CANONICAL | _ffi.CX_KERNEL_STAGED, 0)
    assert ctx.level0_path() == 1
# samples exactly at the isovalue and within the reference's tolerances: batches on the per-cell path
B = np.round(rng.standard_normal((12, 13, 14)) * 2) / 2
T.check_against_oracle(ctx, B.astype(np.float32), 0.5, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_STAGED, 1)
C = (100.0 + rng.standard_normal((12, 12, 12)) * 1.5e-3).astype(np.float32)
T.check_against_oracle(ctx, C, 100.0, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_STAGED, 1)
# dense white noise: every wave closes many batches, the triangle stage's waves start inside batches
D = rng.standard_normal((64, 32, 256)).astype(np.float32)
T.check_against_oracle(ctx, D, 0.0, _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_STAGED, 1)
ctx.close()
print("ENTRY_KERNEL_OK", n)
"""


def test_entry_walking_triangle_kernel():
    """cx_k_emit_triangles_e (the triangle stage that walks queue entries and needs no cell records; not the default because it
    measured slower) against the oracle, in a child process started with CX_DEBUG=1 CX_K2_ENTRIES=1: smooth
    and white-noise fields, both diagonal modes, samples on the isovalue and inside the reference's tolerances"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CX_DEBUG="1", CX_K2_ENTRIES="1")
    r = subprocess.run([sys.executable, "-c", _ENTRY_KERNEL_CHILD.format(root=root, tests=os.path.join(root, "tests"))],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert any(ln.startswith("ENTRY_KERNEL_OK") for ln in r.stdout.splitlines()), r.stdout[-2000:]


def test_empty_and_full(ctx):
    from contourist_amd import _ffi
    A = np.ones((8, 8, 8), dtype=np.float32)
    ctx.upload_grid(A)
    for v in (0.5, 2.0, 1.0):     # all high, all low, all exactly equal (f == v is HIGH)
        c = ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310)
        assert c["n_vertices"] == 0 and c["n_triangles"] == 0 and c["n_cells"] == 0


def test_values_equal_to_isovalue_and_tolerances(ctx):
    """samples exactly at the isovalue, and fields within the reference's np.allclose tolerances"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(9)
    A = np.round(rng.standard_normal((12, 13, 14)) * 2) / 2
    check_against_oracle(ctx, A.astype(np.float32), 0.5, _ffi.CX_DIAG_CPYTHON310, 1)
    B = (rng.standard_normal((12, 12, 12)) * 3e-9).astype(np.float32)          # |f - v| ~ 1e-8 absolute tolerance
    check_against_oracle(ctx, B, 0.0, _ffi.CX_DIAG_CPYTHON310, 1)
    C = (100.0 + rng.standard_normal((12, 12, 12)) * 1.5e-3).astype(np.float32)  # ~ 1e-5 relative tolerance
    check_against_oracle(ctx, C, 100.0, _ffi.CX_DIAG_CPYTHON310, 1)
    check_against_oracle(ctx, C, 100.0007, _ffi.CX_DIAG_CPYTHON310, 1)           # isovalue not representable in fp32


def test_capacity_growth(ctx):
    from contourist_amd import _ffi
    rng = np.random.RandomState(3)
    A = rng.standard_normal((48, 48, 48)).astype(np.float32)    # ~all voxels active: exceeds default buffers
    check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CPYTHON310, 1)


def test_context_reuse_across_shapes_and_kernels(ctx):
    """one context, grids of different shapes and both classify paths in turn: buffers are regrown or reused,
    results stay exact"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(51)
    seq = [((24, 20, 36), 0), ((9, 40, 12), _ffi.CX_KERNEL_GENERIC), ((40, 33, 65), 0), ((6, 6, 6), 0),
           ((24, 20, 36), _ffi.CX_KERNEL_GENERIC), ((50, 12, 260), 0), ((9, 40, 12), 0)]
    for shape, extra in seq:
        A = rng.standard_normal(shape).astype(np.float32)
        check_against_oracle(ctx, A, 0.4, _ffi.CX_DIAG_CPYTHON310 | extra, 1)


def test_fraction_stream():
    """The stream kernel computing the interpolation fractions itself and handing them to the vertex stage (cx_params::tq, round 4:
    no sample gathers in the vertex stage; bit-identical, but measured slower and therefore off -- DESIGN.md section 4) against the
    oracle, in a child process started with CX_DEBUG=1 CX_TQ=1: the same fields as the entry-walking kernel's test -- smooth and
    white noise (white noise overflows the waves' regions of fractions: those waves take the per-cell path), both diagonal modes,
    samples on the isovalue and inside the reference's tolerances."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CX_DEBUG="1", CX_TQ="1")
    r = subprocess.run([sys.executable, "-c", _ENTRY_KERNEL_CHILD.format(root=root, tests=os.path.join(root, "tests"))],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("ENTRY_KERNEL_OK")], r.stdout[-2000:]


def test_negative_zero_samples_at_isovalue_zero(ctx):
    """samples of -0.0 (np.round of small negative numbers) at isovalue 0.0: `-0.0 < 0.0` is false in the reference (and in the
    oracle), so such a sample is NOT below the isovalue -- the stream kernel reads the comparison off the sign bit of f - v and has to
    get this case right (found by tools/fuzz_gpu.py: inconsistent meshes, indices out of range).  Every kernel path, both diagonal
    modes; the fields hold 5-12 % zeros of either sign."""
    from contourist_amd import _ffi
    rng = np.random.RandomState(2024)
    for shape in ((5, 10, 17), (23, 14, 4), (12, 23, 5), (33, 20, 70), (40, 27, 300)):
        A = rng.standard_normal(shape)
        for _ in range(2):
            for ax in range(3):
                A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
        A = (np.round(A / A.std() * 4) / 4).astype(np.float32)
        assert np.any(np.signbit(A) & (A == 0)) and np.any(~np.signbit(A) & (A == 0))
        for extra in (0, _ffi.CX_KERNEL_STAGED, _ffi.CX_KERNEL_TILED, _ffi.CX_KERNEL_FUSED, _ffi.CX_KERNEL_GENERIC):
            check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CPYTHON310 | extra, 1)
        check_against_oracle(ctx, A, 0.0, _ffi.CX_DIAG_CANONICAL, 0)
        check_against_oracle(ctx, A, -0.0, _ffi.CX_DIAG_CPYTHON310, 1)
