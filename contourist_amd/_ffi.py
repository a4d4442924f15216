"""ctypes binding of libcontourist_hip.so (C ABI: include/contourist_hip.h).

The HIP library is the product path: if it is missing or cannot be loaded this module raises --
there is no CPU fallback anywhere in `contourist_amd`.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcontourist_hip.so")
if os.environ.get("CX_DEBUG") == "1" and os.environ.get("CX_LIB_PATH"):   # A/B builds of the tools (tools/variants.sh)
    LIB_PATH = os.environ["CX_LIB_PATH"]

CX_OK = 0
CX_ERR_CAPACITY = -5
CX_DIAG_CANONICAL = 0
CX_DIAG_CPYTHON310 = 1
CX_KERNEL_GENERIC = 0x100
CX_KERNEL_STAGED = 0x200
CX_KERNEL_FUSED = 0x400
CX_KERNEL_TILED = 0x800

# every symbol include/contourist_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "cx_ctx_create", "cx_ctx_destroy", "cx_last_error", "cx_set_stream", "cx_synchronize",
    "cx_grid_upload", "cx_grid_adopt_device", "cx_grid_shadow_f64", "cx_set_origin", "cx_reserve",
    "cx_extract3d", "cx_extract3d_async", "cx_counts_get", "cx_extract3d_levels", "cx_levels_select", "cx_level0_path", "cx_level0_download", "cx_level0_device_ptrs", "cx_level0_device_records", "cx_level0_download_records",
    "cx_postprocess3d", "cx_postprocess3d_ex", "cx_level0_points_f64", "cx_postprocess3d_mesh", "cx_select_seeded3d", "cx_select_seeded3d_ex", "cx_seeded_masks_download", "cx_set_reference_corner", "cx_level1_download", "cx_level1_device_ptrs", "cx_level1_download_keys", "cx_postprocess3d_shard_begin", "cx_postprocess3d_shard_boundary", "cx_postprocess3d_shard_candidates", "cx_postprocess3d_shard_finish", "cx_level1_write", "cx_surface_geometry",
    "cx_grid4d_upload", "cx_grid4d_adopt_device", "cx_set_origin4d", "cx_extract4d", "cx_extract4d_async", "cx_counts4d_get", "cx_select_seeded4d", "cx_select_seeded4d_ex", "cx_seeded_mode", "cx_halo_exchange", "cx_rccl_unique_id", "cx_rccl_comm_init", "cx_rccl_comm_destroy", "cx_rccl_available", "cx_rccl_comm_share", "cx_slab_step", "cx_seeded4d_mask_download", "cx_level0_4d_download", "cx_postprocess4d", "cx_postprocess4d_points", "cx_level1_4d_download", "cx_morph_triangles", "cx_morph_download", "cx_morph_eval", "cx_morph_eval_download", "cx_morph_eval_many", "cx_morph_eval_many_download", "cx_morph_eval_many_device_ptrs", "cx_morph_eval_many_download_all",
    "cx_contour2d_extract", "cx_contour2d_download",
    "cx_timing_enable", "cx_timing_read", "cx_measure_read_bandwidth", "cx_debug_stamps", "cx_version",
]
CX_MESH_OF_THE_MARCH = 8      # cx_postprocess3d_mesh flags bit 3: the triangles are ones the march emitted (at most two per edge before merges)
CX2_ALL_CHAINS = 1
CX2_NO_DEDUPE = 2
CX2_SEARCH_SEEDS = 4


class CxCounts(ctypes.Structure):
    _fields_ = [("n_cells", ctypes.c_int64), ("n_vertices", ctypes.c_int64),
                ("n_triangles", ctypes.c_int64), ("n_border_voxels", ctypes.c_int64)]


class CxCounts2D(ctypes.Structure):
    _fields_ = [("n_points", ctypes.c_uint32), ("n_chains", ctypes.c_uint32), ("n_pairs", ctypes.c_uint32), ("n_levels", ctypes.c_uint32)]


CHAIN2D_DTYPE = np.dtype([("level", np.int32), ("closed", np.int32), ("first", np.uint32), ("count", np.uint32)])


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def _share_hip_runtime_with_torch():
    """One process must not hold two HIP runtimes.  A PyTorch-ROCm wheel ships its own libamdhip64.so and names it
    without the version, so the dynamic loader does not recognise ROCm's libamdhip64.so.7 (loaded for this
    library) as the same thing: torch.cuda then finds "no HIP GPUs".  If such a wheel is installed, load ITS
    runtime first -- this library's NEEDED libamdhip64.so.7 binds to it by soname -- whether or not torch gets
    imported later.  Without torch the ROCm copy on the library's RUNPATH is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass   # best effort: the library itself still loads (against ROCm's runtime)


def load():
    """load (once) and type the shared library; raises HipLibraryMissing loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `python -m contourist_amd.build` (hipcc --offload-arch=gfx950). "
            "contourist_amd has no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, u32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_double
    L.cx_version.restype = ctypes.c_char_p
    L.cx_version.argtypes = []
    L.cx_last_error.restype = ctypes.c_char_p
    L.cx_last_error.argtypes = [vp]
    sigs = {
        "cx_ctx_create": [ctypes.c_int, ctypes.POINTER(vp)],
        "cx_ctx_destroy": [vp],
        "cx_set_stream": [vp, vp],
        "cx_synchronize": [vp],
        "cx_grid_upload": [vp, vp, i64, i64, i64],
        "cx_grid_adopt_device": [vp, vp, i64, i64, i64],
        "cx_grid_shadow_f64": [vp, vp, i64, i64, i64],
        "cx_set_origin": [vp, i64, i64, i64],
        "cx_reserve": [vp, i64, i64, i64],
        "cx_extract3d": [vp, dbl, u32, ctypes.POINTER(CxCounts)],
        "cx_extract3d_async": [vp, dbl, u32],
        "cx_counts_get": [vp, ctypes.POINTER(CxCounts)],
        "cx_level0_path": [vp, ctypes.POINTER(ctypes.c_int)],
        "cx_extract3d_levels": [vp, vp, ctypes.c_int32, u32, vp],
        "cx_levels_select": [vp, ctypes.c_int32],
        "cx_level0_download": [vp, vp, vp],
        "cx_level0_device_ptrs": [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "cx_level0_device_records": [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "cx_level0_download_records": [vp, vp, vp],
        "cx_postprocess3d": [vp, u32, vp],
        "cx_postprocess3d_ex": [vp, u32, dbl, vp],
        "cx_level0_points_f64": [vp, vp],
        "cx_postprocess3d_mesh": [vp, vp, i64, vp, i64, vp, u32, dbl, vp],
        "cx_select_seeded3d": [vp, vp, i64, vp, vp],
        "cx_select_seeded3d_ex": [vp, vp, i64, vp, u32, vp],
        "cx_seeded_masks_download": [vp, vp, vp],
        "cx_set_reference_corner": [vp, i64, i64, i64],
        "cx_level1_download": [vp, vp, vp],
        "cx_level1_device_ptrs": [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64)],
        "cx_level1_download_keys": [vp, vp],
        "cx_postprocess3d_shard_begin": [vp, u32, i64, i64, vp, vp, vp, vp],
        "cx_postprocess3d_shard_boundary": [vp, ctypes.c_int, vp, vp],
        "cx_postprocess3d_shard_candidates": [vp, vp, vp, vp, vp, vp, vp],
        "cx_postprocess3d_shard_finish": [vp, vp, vp, i64, vp],
        "cx_level1_write": [vp, ctypes.c_int, ctypes.c_char_p, vp, vp],
        "cx_surface_geometry": [vp, vp, ctypes.POINTER(i64), vp, ctypes.POINTER(i64), ctypes.c_int],
        "cx_debug_stamps": [vp, i64, vp],
        "cx_grid4d_upload": [vp, vp, i64, i64, i64, i64],
        "cx_grid4d_adopt_device": [vp, vp, i64, i64, i64, i64],
        "cx_set_origin4d": [vp, i64, i64, i64, i64],
        "cx_extract4d": [vp, dbl, u32, ctypes.POINTER(CxCounts)],
        "cx_extract4d_async": [vp, dbl, u32],
        "cx_counts4d_get": [vp, ctypes.POINTER(CxCounts)],
        "cx_select_seeded4d": [vp, vp, i64, vp],
        "cx_select_seeded4d_ex": [vp, vp, i64, vp, ctypes.c_uint32, vp],
        "cx_seeded4d_mask_download": [vp, vp],
        "cx_seeded_mode": [vp, vp],
        "cx_halo_exchange": [vp, vp, ctypes.c_int, ctypes.c_int, vp, i64, i64],
        "cx_rccl_unique_id": [vp],
        "cx_rccl_comm_init": [vp, vp, ctypes.c_int, ctypes.c_int],
        "cx_rccl_comm_destroy": [vp],
        "cx_rccl_available": [],
        "cx_rccl_comm_share": [vp, vp],
        "cx_slab_step": [vp, vp, i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_uint32],
        "cx_level0_4d_download": [vp, vp, vp, vp],
        "cx_postprocess4d": [vp, ctypes.c_int32, vp],
        "cx_postprocess4d_points": [vp, ctypes.c_int32, vp, vp],
        "cx_level1_4d_download": [vp, vp, vp],
        "cx_morph_triangles": [vp, vp],
        "cx_morph_download": [vp, vp, vp, vp],
        "cx_morph_eval": [vp, dbl, vp],
        "cx_morph_eval_download": [vp, vp, vp],
        "cx_morph_eval_many": [vp, vp, ctypes.c_int32, vp],
        "cx_morph_eval_many_download": [vp, ctypes.c_int32, vp, vp],
        "cx_morph_eval_many_device_ptrs": [vp, ctypes.c_int32, vp, vp],
        "cx_morph_eval_many_download_all": [vp, vp, vp],
        "cx_contour2d_extract": [vp, vp, ctypes.c_int, i64, i64, vp, ctypes.c_int32, vp, i64, u32, vp, ctypes.POINTER(CxCounts2D)],
        "cx_contour2d_download": [vp, vp, vp, vp],
        "cx_timing_enable": [vp, ctypes.c_int],
        "cx_timing_read": [vp, ctypes.POINTER(dbl), ctypes.POINTER(ctypes.c_int)],
        "cx_measure_read_bandwidth": [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(dbl)],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        fn.argtypes = args
    _lib = L
    return L


class CxError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "contourist_hip error %d: %s" % (code, message))
        self.code = code


class Context(object):
    """One device + one HIP stream (cx_ctx).  Not thread safe."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = ctypes.c_void_p()
        rc = self.lib.cx_ctx_create(int(device), ctypes.byref(h))
        if rc != CX_OK:
            raise CxError(rc, "cx_ctx_create(device=%d) failed (no HIP device?)" % device)
        self.handle = h
        self.device = int(device)
        self._keep = None      # keeps an adopted tensor / uploaded array alive
        self.shape = None
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != CX_OK:
            raise CxError(rc, self.lib.cx_last_error(self.handle).decode("utf-8", "replace"))

    def set_stream(self, stream):
        self._check(self.lib.cx_set_stream(self.handle, ctypes.c_void_p(int(stream) if stream else 0)))

    def synchronize(self):
        self._check(self.lib.cx_synchronize(self.handle))

    def upload_grid(self, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        assert a.ndim == 3, "3-D sample array expected"
        self._check(self.lib.cx_grid_upload(self.handle, a.ctypes.data, *a.shape))
        self.shape = tuple(int(n) for n in a.shape)
        self._keep = None

    def adopt_device_grid(self, device_ptr, shape, keepalive=None):
        assert len(shape) == 3
        self._check(self.lib.cx_grid_adopt_device(self.handle, ctypes.c_void_p(int(device_ptr)), *[int(n) for n in shape]))
        self.shape = tuple(int(n) for n in shape)
        self._keep = keepalive

    def shadow_grid_f64(self, array=None):
        """float64 originals of the bound samples (None drops them): Level 1 interpolates the crossings on these, as the
        reference does on the float64 values of its callable (tetrahedral.py:471-487)"""
        if array is None:
            self._check(self.lib.cx_grid_shadow_f64(self.handle, None, 0, 0, 0))
            return
        a = np.ascontiguousarray(array, dtype=np.float64)
        assert tuple(a.shape) == tuple(self.shape), (a.shape, self.shape)
        self._check(self.lib.cx_grid_shadow_f64(self.handle, a.ctypes.data, *a.shape))

    def set_origin(self, o0=0, o1=0, o2=0):
        self._check(self.lib.cx_set_origin(self.handle, int(o0), int(o1), int(o2)))

    def set_origin4d(self, o0=0, o1=0, o2=0, o3=0):
        "lattice coordinates of sample [0,0,0,0] in the reference's grid (negative: the array carries a rim of extra samples)"
        self._check(self.lib.cx_set_origin4d(self.handle, int(o0), int(o1), int(o2), int(o3)))

    def reserve(self, max_cells=0, max_vertices=0, max_triangles=0):
        self._check(self.lib.cx_reserve(self.handle, int(max_cells), int(max_vertices), int(max_triangles)))

    def extract3d(self, value, flags=CX_DIAG_CPYTHON310):
        c = CxCounts()
        self._check(self.lib.cx_extract3d(self.handle, float(value), int(flags), ctypes.byref(c)))
        return dict(n_cells=c.n_cells, n_vertices=c.n_vertices, n_triangles=c.n_triangles,
                    n_border_voxels=c.n_border_voxels)

    def extract3d_levels(self, values, flags=CX_DIAG_CPYTHON310):
        """all isovalues of `values` in one call (one pass over the samples for all of them) -> list of counts dicts;
        select_level(i) then makes level i the current extraction for download_level0 / postprocess3d / ..."""
        vals = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        out = (CxCounts * len(vals))()
        self._check(self.lib.cx_extract3d_levels(self.handle, vals.ctypes.data, len(vals), int(flags), ctypes.cast(out, ctypes.c_void_p)))
        return [dict(n_cells=c.n_cells, n_vertices=c.n_vertices, n_triangles=c.n_triangles, n_border_voxels=c.n_border_voxels) for c in out]

    def select_level(self, index):
        self._check(self.lib.cx_levels_select(self.handle, int(index)))

    def extract3d_async(self, value, flags=CX_DIAG_CPYTHON310):
        self._check(self.lib.cx_extract3d_async(self.handle, float(value), int(flags)))

    def counts(self):
        c = CxCounts()
        self._check(self.lib.cx_counts_get(self.handle, ctypes.byref(c)))
        return dict(n_cells=c.n_cells, n_vertices=c.n_vertices, n_triangles=c.n_triangles,
                    n_border_voxels=c.n_border_voxels)

    def download_level0(self, counts):
        """-> (xyz (V,3) float32 grid coordinates, edge ids (V,) uint32, triangles (T,3) int32)"""
        nv, nt = int(counts["n_vertices"]), int(counts["n_triangles"])
        verts = np.empty((nv, 4), dtype=np.float32)
        tris = np.empty((nt, 3), dtype=np.int32)
        self._check(self.lib.cx_level0_download(self.handle, verts.ctypes.data, tris.ctypes.data))
        keys = verts[:, 3].copy().view(np.uint32)
        return verts[:, :3].copy(), keys, tris

    def download_level0_records(self, counts):
        """-> (edge ids (V,) uint32, t (V,) float32: the crossing sits at q + t d on the lattice edge the id names, triangles (T,3) int32):
        the 8-byte records as the march leaves them (no expansion to coordinates)"""
        nv, nt = int(counts["n_vertices"]), int(counts["n_triangles"])
        recs = np.empty((nv, 2), dtype=np.uint32)
        tris = np.empty((nt, 3), dtype=np.int32)
        self._check(self.lib.cx_level0_download_records(self.handle, recs.ctypes.data, tris.ctypes.data))
        return recs[:, 0].copy(), recs[:, 1].copy().view(np.float32), tris

    def set_reference_corner(self, corner=(0, 0, 0)):
        self._check(self.lib.cx_set_reference_corner(self.handle, *[int(c) for c in corner]))

    def select_seeded(self, endpoints, voxel_range=None, all_in_range=False, parallel=False):
        """restrict the next post-passes to the components reached from `endpoints` (n x 2 x 3 lattice points);
        voxel_range = (lo[3], hi[3]) in_range box of the growth (default: the whole array);
        -> dict(seed_voxels, groups_kept, triangles_kept)"""
        ep = np.ascontiguousarray(np.asarray(endpoints, dtype=np.int64).reshape(-1, 6), dtype=np.int32)
        out = np.zeros(4, dtype=np.int64)
        box = None
        if voxel_range is not None:
            box = np.ascontiguousarray(np.asarray(voxel_range, dtype=np.int64).reshape(6), dtype=np.int32)
        self._check(self.lib.cx_select_seeded3d_ex(self.handle, ep.ctypes.data, int(len(ep)),
                                                   box.ctypes.data if box is not None else None, (1 if all_in_range else 0) | (2 if parallel else 0), out.ctypes.data))
        return dict(seed_voxels=int(out[0]), groups_kept=int(out[1]), triangles_kept=int(out[2]))

    def seeded_masks(self, counts):
        "(triangle mask, vertex mask) of the last select_seeded as bool arrays (all True without a selection)"
        tk = np.empty(int(counts["n_triangles"]), dtype=np.uint8)
        vk = np.empty(int(counts["n_vertices"]), dtype=np.uint8)
        self._check(self.lib.cx_seeded_masks_download(self.handle, tk.ctypes.data, vk.ctypes.data))
        return tk.astype(bool), vk.astype(bool)

    def postprocess3d(self, flags=0, smooth=0.0):
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.cx_postprocess3d_ex(self.handle, int(flags), float(smooth or 0.0), out.ctypes.data))
        return dict(n_vertices=int(out[0]), n_triangles=int(out[1]), n_after_weld=int(out[2]),
                    n_after_tiny=int(out[3]), n_components=int(out[4]))

    def level0_points_f64(self, counts):
        "float64 coordinates of the Level-0 vertices as the reference interpolates them, in download_level0 order"
        pts = np.empty((int(counts["n_vertices"]), 3), dtype=np.float64)
        self._check(self.lib.cx_level0_points_f64(self.handle, pts.ctypes.data))
        return pts

    def postprocess3d_mesh(self, points, triangles, corner, flags=0, smooth=0.0):
        """Level 1 of an assembled Level-0 mesh (vertices in ascending global edge-id order, float64 grid coordinates of
        the whole volume, triangles as indices) -> the same dict as postprocess3d; fetch with download_level1"""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        tris = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        c3 = np.ascontiguousarray(corner, dtype=np.int64).reshape(3)
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.cx_postprocess3d_mesh(self.handle, pts.ctypes.data, len(pts), tris.ctypes.data, len(tris), c3.ctypes.data,
                                                   int(flags), float(smooth or 0.0), out.ctypes.data))
        return dict(n_vertices=int(out[0]), n_triangles=int(out[1]), n_after_weld=int(out[2]),
                    n_after_tiny=int(out[3]), n_components=int(out[4]))

    def download_level1(self, counts):
        pts = np.empty((int(counts["n_vertices"]), 3), dtype=np.float64)
        tris = np.empty((int(counts["n_triangles"]), 3), dtype=np.int32)
        self._check(self.lib.cx_level1_download(self.handle, pts.ctypes.data, tris.ctypes.data))
        return pts, tris

    def level1_device_ptrs(self):
        "(points pointer, triangles pointer, n_vertices, n_triangles) of the Level-1 mesh ON THE DEVICE (no copy; valid until the next post-pass)"
        pp, tp, nv, nt = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.cx_level1_device_ptrs(self.handle, ctypes.byref(pp), ctypes.byref(tp), ctypes.byref(nv), ctypes.byref(nt)))
        return pp.value or 0, tp.value or 0, nv.value, nt.value

    def level1_torch(self, copy=True):
        """the Level-1 mesh as torch tensors on the context's device: points (V,3) float64, triangles (T,3) int32 (device order).
        copy=True: tensors that own their memory (one device-to-device copy); copy=False: views of the context's buffers,
        valid only until the next post-pass / extraction on this context."""
        import torch
        pp, tp, nv, nt = self.level1_device_ptrs()
        dev = torch.device("cuda", self.device)

        class _View(object):
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}
        pts = torch.as_tensor(_View(pp, (nv, 3), "<f8"), device=dev) if nv else torch.zeros((0, 3), dtype=torch.float64, device=dev)
        tris = torch.as_tensor(_View(tp, (nt, 3), "<i4"), device=dev) if nt else torch.zeros((0, 3), dtype=torch.int32, device=dev)
        if copy:
            pts, tris = pts.clone(), tris.clone()
        else:
            pts._cx_keep = tris._cx_keep = self
        return pts, tris

    def download_level1_keys(self, counts):
        "edge ids (local to the marched array) of the vertices of download_level1, in its order"
        keys = np.empty(int(counts["n_vertices"]), dtype=np.uint32)
        self._check(self.lib.cx_level1_download_keys(self.handle, keys.ctypes.data))
        return keys

    def shard_begin(self, own_lo, own_hi, flags=0):
        """sharded Level 1, local part (cx_postprocess3d_shard_begin + _candidates) -> dict(counts, n_own_lower, n_upper_copies,
        cand_label / cand_x / cand_vertex_key / cand_nx / cand_negative / cand_has (C,)); the boundary lists stay on the device
        (shard_boundary)"""
        out = np.zeros(8, dtype=np.int64)
        n1, n4, nc = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self.lib.cx_postprocess3d_shard_begin(self.handle, int(flags), int(own_lo), int(own_hi), out.ctypes.data,
                                                          ctypes.addressof(n1), ctypes.addressof(n4), ctypes.addressof(nc)))
        C = int(nc.value)
        L = dict(cand_label=np.zeros(C, np.uint32), cand_x=np.zeros(C, np.float64), cand_vertex_key=np.zeros(C, np.uint32),
                 cand_nx=np.zeros(C, np.float64), cand_negative=np.zeros(C, np.uint8), cand_has=np.zeros(C, np.uint8))
        if C:
            self._check(self.lib.cx_postprocess3d_shard_candidates(self.handle, *[L[k].ctypes.data for k in (
                "cand_label", "cand_x", "cand_vertex_key", "cand_nx", "cand_negative", "cand_has")]))
        L.update(n_own_lower=int(n1.value), n_upper_copies=int(n4.value), counts=dict(n_after_weld=int(out[2]), n_after_tiny=int(out[3])))
        return L

    def shard_boundary(self, which, n, torch_device=None):
        """one boundary list of shard_begin (which = 1: own triangles next to the lower neighbour, 4: copies of the upper
        neighbour's first layer) -> (hash int64 (n,), label int32 (n,)): torch tensors on `torch_device` (the lists need not
        leave the GPU), numpy arrays without one.  The hashes are 64-bit patterns; as int64 they sort the same way on every rank."""
        n = int(n)
        if torch_device is not None:
            import torch
            h = torch.empty(n, dtype=torch.int64, device=torch_device)
            lab = torch.empty(n, dtype=torch.int32, device=torch_device)
            hp, lp = h.data_ptr(), lab.data_ptr()
        else:
            h = np.empty(n, dtype=np.int64)
            lab = np.empty(n, dtype=np.int32)
            hp, lp = h.ctypes.data, lab.ctypes.data
        if n:
            self._check(self.lib.cx_postprocess3d_shard_boundary(self.handle, int(which), ctypes.c_void_p(hp), ctypes.c_void_p(lp)))
        return h, lab

    def shard_finish(self, labels, flips):
        "the agreed flips for the components that reach a neighbour -> counts dict for download_level1 / download_level1_keys"
        lab = np.ascontiguousarray(labels, dtype=np.uint32).reshape(-1)
        fl = np.ascontiguousarray(flips, dtype=np.uint8).reshape(-1)
        assert len(lab) == len(fl)
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.cx_postprocess3d_shard_finish(self.handle, lab.ctypes.data if len(lab) else None, fl.ctypes.data if len(fl) else None,
                                                           len(lab), out.ctypes.data))
        return dict(n_vertices=int(out[0]), n_triangles=int(out[1]), n_components=int(out[4]))

    def write_level1(self, path, fmt="ply", mins=None, delta=None):
        """the Level-1 mesh of the last post-pass as a binary file written straight from the device buffers (no numpy arrays):
        fmt "ply" (float64 positions, int32 faces) or "gltf_bin" (float32 positions + uint32 indices); mins / delta: world
        coordinates = grid * delta + mins.  -> dict(n_vertices, n_triangles, bytes, min, max)"""
        md = None
        if mins is not None or delta is not None:
            md = np.ascontiguousarray(np.concatenate([np.asarray(mins if mins is not None else [0, 0, 0], dtype=np.float64).reshape(3),
                                                      np.asarray(delta if delta is not None else [1, 1, 1], dtype=np.float64).reshape(3)]))
        info = np.zeros(9, dtype=np.float64)
        self._check(self.lib.cx_level1_write(self.handle, {"ply": 0, "gltf_bin": 1}[fmt], os.fsencode(path),
                                             None if md is None else md.ctypes.data, info.ctypes.data))
        return dict(n_vertices=int(info[0]), n_triangles=int(info[1]), bytes=int(info[2]), min=info[3:6].copy(), max=info[6:9].copy())

    def surface_geometry(self, points, triangles, do_clean):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        tris = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3).copy()
        nv = ctypes.c_int64(len(pts))
        nt = ctypes.c_int64(len(tris))
        self._check(self.lib.cx_surface_geometry(self.handle, pts.ctypes.data, ctypes.byref(nv),
                                                 tris.ctypes.data, ctypes.byref(nt), int(bool(do_clean))))
        return pts[:nv.value].copy(), tris[:nt.value].copy()

    # ---- 4-D ---------------------------------------------------------------------------------
    def upload_grid4d(self, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        assert a.ndim == 4, "4-D sample array expected"
        self._check(self.lib.cx_grid4d_upload(self.handle, a.ctypes.data, *a.shape))
        self.shape4 = tuple(int(n) for n in a.shape)

    def adopt_device_grid4d(self, device_ptr, shape, keepalive=None):
        assert len(shape) == 4
        self._check(self.lib.cx_grid4d_adopt_device(self.handle, ctypes.c_void_p(int(device_ptr)), *[int(n) for n in shape]))
        self.shape4 = tuple(int(n) for n in shape)
        self._keep4 = keepalive

    def extract4d(self, value, flags=CX_DIAG_CPYTHON310):
        c = CxCounts()
        self._check(self.lib.cx_extract4d(self.handle, float(value), int(flags), ctypes.byref(c)))
        return dict(n_cells=c.n_cells, n_vertices=c.n_vertices, n_tetrahedra=c.n_triangles, n_border_voxels=c.n_border_voxels)

    def extract4d_async(self, value, flags=CX_DIAG_CPYTHON310):
        "the 4-D march enqueued on this context's stream, not waited for; counts4d() waits and returns what extract4d returns"
        self._check(self.lib.cx_extract4d_async(self.handle, float(value), int(flags)))

    def counts4d(self):
        c = CxCounts()
        self._check(self.lib.cx_counts4d_get(self.handle, ctypes.byref(c)))
        return dict(n_cells=c.n_cells, n_vertices=c.n_vertices, n_tetrahedra=c.n_triangles, n_border_voxels=c.n_border_voxels)

    def select_seeded4d(self, endpoints, voxel_range=None, all_in_range=False, parallel=False):
        """restrict the 4-D post-pass to the components the reference's seeded search reaches from the lattice end
        point pairs [(i0,j0,k0,l0), (i1,j1,k1,l1)] -> dict(seed_voxels, groups_kept, tetrahedra_kept).
        voxel_range = (lo[4], hi[4]): in_range box of the growth in array coordinates (default: the whole array); seed
        voxels outside it are kept and grow one step into it (an array with a rim around the reference's grid).
        all_in_range: every hyper-voxel inside the box is kept, the end points only add seed voxels outside it."""
        ep = np.ascontiguousarray(np.asarray(endpoints, dtype=np.int64).reshape(-1, 8), dtype=np.int32)
        out = np.zeros(4, dtype=np.int64)
        box = None
        if voxel_range is not None:
            box = np.ascontiguousarray(np.asarray(voxel_range, dtype=np.int64).reshape(8), dtype=np.int32)
        self._check(self.lib.cx_select_seeded4d_ex(self.handle, ep.ctypes.data, int(len(ep)), box.ctypes.data if box is not None else None,
                                                  (1 if all_in_range else 0) | (2 if parallel else 0), out.ctypes.data))
        return dict(seed_voxels=int(out[0]), groups_kept=int(out[1]), tetrahedra_kept=int(out[2]))

    def halo_exchange(self, rccl_comm, rank, world, local_ptr, n_own, plane_samples):
        """send plane 0 of the device buffer to rank-1, receive plane n_own from rank+1 (RCCL, on the context's stream); rccl_comm:
        the caller's ncclComm_t as an integer / ctypes pointer.  (contourist_amd.distributed does this step with torch.distributed.)"""
        self._check(self.lib.cx_halo_exchange(self.handle, rccl_comm, int(rank), int(world), local_ptr, int(n_own), int(plane_samples)))

    def rccl_unique_id(self):
        "128 bytes of a fresh ncclUniqueId (one rank calls this; the bytes go to every rank, e.g. by torch.distributed.broadcast)"
        buf = np.zeros(128, dtype=np.uint8)
        self._check(self.lib.cx_rccl_unique_id(buf.ctypes.data))
        return buf

    def rccl_comm_init(self, id128, rank, world):
        "collective over the ranks: a communicator owned by this context (cx_halo_exchange / slab_step with rccl_comm == NULL)"
        buf = np.ascontiguousarray(id128, dtype=np.uint8)
        assert buf.size == 128
        self._check(self.lib.cx_rccl_comm_init(self.handle, buf.ctypes.data, int(rank), int(world)))

    def rccl_available(self):
        "local, no collective: can this process run the C-side halo exchange (an RCCL copy is loaded and exports what is needed)?"
        return self.lib.cx_rccl_available() == CX_OK

    def rccl_comm_share(self, owner):
        "use `owner`'s communicator (another context of this rank, which keeps the ownership and must outlive this one)"
        self._check(self.lib.cx_rccl_comm_share(self.handle, owner.handle))
        self._comm_owner = owner

    def rccl_comm_destroy(self):
        self._check(self.lib.cx_rccl_comm_destroy(self.handle))
        self._comm_owner = None

    def slab_step(self, local_ptr, n_own, n1, n2, rank, world, value, flags=CX_DIAG_CPYTHON310, keepalive=None):
        """one rank's whole step for one volume in ONE call: adopt the device buffer (n_own planes + room for the halo plane unless
        this is the last rank), halo exchange on this context's stream with its own communicator, extraction enqueued behind it"""
        self._check(self.lib.cx_slab_step(self.handle, ctypes.c_void_p(int(local_ptr)), int(n_own), int(n1), int(n2), int(rank), int(world),
                                          float(value), int(flags)))
        self.shape = (int(n_own) + (1 if rank + 1 < world else 0), int(n1), int(n2))
        self._keep = keepalive

    def seeded_mode(self):
        "how the last seeded selection ran its end points: 'sequential' (the reference's shared visited set) or 'parallel'"
        m = ctypes.c_int(0)
        self._check(self.lib.cx_seeded_mode(self.handle, ctypes.byref(m)))
        return "parallel" if m.value else "sequential"

    def seeded4d_mask(self, counts):
        "mask over the Level-0 tetrahedra left by select_seeded4d (uint8, 1 = kept)"
        keep = np.zeros(int(counts["n_tetrahedra"]), dtype=np.uint8)
        self._check(self.lib.cx_seeded4d_mask_download(self.handle, keep.ctypes.data))
        return keep

    def download_level0_4d(self, counts):
        "-> (xyzt (V,4) float32, edge ids (V,) uint32, tetrahedra (T,4) int32)"
        nv, nt = int(counts["n_vertices"]), int(counts["n_tetrahedra"])
        verts = np.empty((nv, 4), dtype=np.float32)
        keys = np.empty(nv, dtype=np.uint32)
        tets = np.empty((nt, 4), dtype=np.int32)
        self._check(self.lib.cx_level0_4d_download(self.handle, verts.ctypes.data, keys.ctypes.data, tets.ctypes.data))
        return verts, keys, tets

    def postprocess4d(self, nbins=100, points=None):
        """bin_times / drop_instant / tiny collapse on the device; points: (n_vertices, 4) float64 replacing the linearly
        interpolated crossing points (linear_interpolate=False: refined on the host), in the reference's lattice"""
        out = np.zeros(8, dtype=np.int64)
        if points is not None:
            pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 4)
            self._check(self.lib.cx_postprocess4d_points(self.handle, int(nbins), pts.ctypes.data, out.ctypes.data))
        else:
            self._check(self.lib.cx_postprocess4d(self.handle, int(nbins), out.ctypes.data))
        return dict(n_vertices=int(out[0]), n_tetrahedra=int(out[1]), n_after_drop=int(out[2]), n_after_tiny=int(out[3]))

    def download_level1_4d(self, counts):
        pts = np.empty((int(counts["n_vertices"]), 4), dtype=np.float64)
        tets = np.empty((int(counts["n_tetrahedra"]), 4), dtype=np.int32)
        self._check(self.lib.cx_level1_4d_download(self.handle, pts.ctypes.data, tets.ctypes.data))
        return pts, tets

    def morph_triangles(self):
        "-> (points4d (V,4) float64, segments (S,2) int32 low t -> high t, triangles (T,3) int32 oriented)"
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.cx_morph_triangles(self.handle, out.ctypes.data))
        pts = np.empty((int(out[0]), 4), dtype=np.float64)
        segs = np.empty((int(out[1]), 2), dtype=np.int32)
        tris = np.empty((int(out[2]), 3), dtype=np.int32)
        self._check(self.lib.cx_morph_download(self.handle, pts.ctypes.data, segs.ctypes.data, tris.ctypes.data))
        return pts, segs, tris, int(out[4])

    def morph_eval(self, t, download=True):
        """surface at time t from the last morph_triangles() -> (points (P,3) float64, triangles (Q,3) int32)
        or, with download=False, just the counts (the mesh stays on the device)"""
        out = np.zeros(2, dtype=np.int64)
        self._check(self.lib.cx_morph_eval(self.handle, float(t), out.ctypes.data))
        if not download:
            return int(out[0]), int(out[1])
        pts = np.empty((int(out[0]), 3), dtype=np.float64)
        tris = np.empty((int(out[1]), 3), dtype=np.int32)
        self._check(self.lib.cx_morph_eval_download(self.handle, pts.ctypes.data, tris.ctypes.data))
        return pts, tris

    def morph_eval_many(self, times, download=True, out=None):
        """the surfaces at all `times` from the last morph_triangles() in one set of launches (cx_morph_eval_many)
        -> list of (points (P,3) float64, triangles (Q,3) int32), one per time, each what morph_eval(t) returns;
        with download=False the counts array (n,2) int64 (the meshes stay on the device: morph_eval_device_ptrs(i)).
        out = (points (>= sum P, 3) float64, triangles (>= sum Q, 3) int32): arrays to download into (a consumer that plays stream
        after stream keeps them: fresh arrays cost their page faults, ~20 ms for the 428 MB of config 4's 64 surfaces against 11 ms)"""
        _out = out
        ts = np.ascontiguousarray(times, dtype=np.float64).reshape(-1)
        out = np.zeros((len(ts), 2), dtype=np.int64)
        self._check(self.lib.cx_morph_eval_many(self.handle, ts.ctypes.data, len(ts), out.ctypes.data))
        if not download:
            return out
        # one transfer for all surfaces (cx_morph_eval_many_download_all); the surfaces are views into the two arrays
        counts = out
        npts, ntri = int(counts[:, 0].sum()), int(counts[:, 1].sum())
        pts_all = tris_all = None
        if _out is not None:
            pts_all, tris_all = _out
            ok = (pts_all.dtype == np.float64 and tris_all.dtype == np.int32 and pts_all.flags.c_contiguous and tris_all.flags.c_contiguous
                  and pts_all.ndim == 2 and tris_all.ndim == 2 and pts_all.shape[1] == 3 and tris_all.shape[1] == 3
                  and len(pts_all) >= npts and len(tris_all) >= ntri)
            if not ok:
                raise ValueError("out: C-contiguous (>= %d, 3) float64 and (>= %d, 3) int32 arrays expected" % (npts, ntri))
        else:
            pts_all = np.empty((npts, 3), dtype=np.float64)
            tris_all = np.empty((ntri, 3), dtype=np.int32)
        self._check(self.lib.cx_morph_eval_many_download_all(self.handle, pts_all.ctypes.data, tris_all.ctypes.data))
        res, p0, t0 = [], 0, 0
        for i in range(len(ts)):
            npi, nti = int(out[i, 0]), int(out[i, 1])
            res.append((pts_all[p0:p0 + npi], tris_all[t0:t0 + nti]))
            p0 += npi; t0 += nti
        return res

    def morph_eval_device_ptrs(self, i):
        "device addresses (points float64 x 3, triangles int32 x 3) of surface i of the last morph_eval_many / morph_eval; 0 for an empty one"
        p, t = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self.lib.cx_morph_eval_many_device_ptrs(self.handle, int(i), ctypes.byref(p), ctypes.byref(t)))
        return int(p.value or 0), int(t.value or 0)

    def contour2d(self, samples, values, seeds=None, flags=0, mins_delta=None, device_ptr=None, shape=None):
        """polylines of a 2-D sample array at every isovalue of `values` (ascending, distinct)
        -> (points (P,2) float64, keys (P,) int64, chains (C,) CHAIN2D_DTYPE, n_pairs).
        samples: fp32 numpy array (n, m), or None with device_ptr + shape for samples already in HBM.
        seeds: (S,4) int32 rows (i, j, role, level index), or None for the reference's grid search."""
        vals = np.ascontiguousarray(values, dtype=np.float64)
        if device_ptr is None:
            arr = np.ascontiguousarray(samples, dtype=np.float32)
            assert arr.ndim == 2
            n, m = arr.shape
            ptr, on_device = arr.ctypes.data, 0
        else:
            n, m = (int(x) for x in shape)
            ptr, on_device = int(device_ptr), 1
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int32).reshape(-1, 4)
        md = None if mins_delta is None else np.ascontiguousarray(mins_delta, dtype=np.float64).reshape(4)
        c = CxCounts2D()
        self._check(self.lib.cx_contour2d_extract(self.handle, ptr, on_device, n, m, vals.ctypes.data, len(vals),
                                                  None if sd is None or len(sd) == 0 else sd.ctypes.data, 0 if sd is None else len(sd),
                                                  int(flags), None if md is None else md.ctypes.data, ctypes.byref(c)))
        pts = np.empty((c.n_points, 2), dtype=np.float64)
        keys = np.empty((c.n_points,), dtype=np.int64)
        chains = np.empty((c.n_chains,), dtype=CHAIN2D_DTYPE)
        self._check(self.lib.cx_contour2d_download(self.handle, pts.ctypes.data, keys.ctypes.data, chains.ctypes.data))
        return pts, keys, chains, int(c.n_pairs)

    # what cx_timing_read's slots measure, with the algorithmic bytes of each stage (bench.py)
    def level0_path(self):
        "kernels of the last extraction: 0 generic classify + triangle stage, 1 staged pipeline, 2 stream + scan + fused emit, 3 stream + scan + tile emit + boundary"
        p = ctypes.c_int()
        self._check(self.lib.cx_level0_path(self.handle, ctypes.byref(p)))
        return p.value

    def kernel_names(self):
        path = self.level0_path()
        if path == 2:
            return [("stream_ms", "cx_k_stream"), ("scan_ms", "cx_k_scan_list"), ("cells_ms", "cx_k_cell_info+cx_k_emit_mesh")]
        if path == 1:
            return [("stream_ms", "cx_k_stream"), ("scan_ms", "cx_k_scan_list"),
                    ("cells_ms", "cx_k_emit_vertices"), ("emit_ms", "cx_k_emit_triangles_q")]
        if path == 3:
            return [("stream_ms", "cx_k_stream"), ("scan_ms", "cx_k_scan_list"),
                    ("cells_ms", "cx_k_tile_emit"), ("emit_ms", "cx_k_tile_boundary")]
        return [("stream_ms", "cx_k_classify_generic"), ("emit_ms", "cx_k_emit_triangles")]

    def vertex_stage_bytes(self, counts):
        if self.level0_path() in (2, 3):     # fused emit / tile emit: 8 B per vertex record + 12 B per triangle written by ONE kernel
            return 8.0 * counts["n_vertices"] + 12.0 * counts["n_triangles"]
        # what the stage must leave: 8-byte vertex records.  (Its cell records and per-entry words are the pipeline's own
        # intermediates -- traffic, not algorithmic bytes.)
        return 8.0 * counts["n_vertices"]

    def triangle_stage_bytes(self, counts):
        if self.level0_path() == 3:          # the boundary kernel: the triangles of ~13 % of the voxels (counted with the tile kernel above)
            return 0.0
        return 12.0 * counts["n_triangles"]

    def measure_read_bandwidth(self, device_ptr, nbytes, reps=5):
        "GB/s of a plain streaming read of nbytes at device_ptr (best of reps launches)"
        g = ctypes.c_double()
        self._check(self.lib.cx_measure_read_bandwidth(self.handle, ctypes.c_void_p(int(device_ptr)), int(nbytes), int(reps), ctypes.byref(g)))
        return g.value

    def timing_enable(self, on=True):
        self._check(self.lib.cx_timing_enable(self.handle, int(bool(on))))

    def timing_read(self):
        ms = (ctypes.c_double * 8)()
        n = ctypes.c_int()
        self._check(self.lib.cx_timing_read(self.handle, ms, ctypes.byref(n)))
        return dict(classify_ms=ms[0], emit_ms=ms[1], total_ms=ms[2], stream_ms=ms[3], scan_ms=ms[4],
                    cells_ms=ms[5], n=n.value)
