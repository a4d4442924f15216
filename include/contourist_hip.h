/*
 * contourist_hip.h -- C ABI of the MI355X (gfx950) isosurface extractor.
 *
 * The reference (AaronWatters/contourist) is pure Python and has no FFI; its boundary for this
 * path is a Python class API.  Each entry point below names the reference interface whose work
 * it replaces (paths relative to the reference checkout, contourist/...).  The Python host side
 * (contourist_amd/) mirrors the reference's classes 1:1 and calls these through ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions: every function returns 0 (CX_OK) or a negative cx_status, never throws, keeps no
 * global state.  One cx_ctx == one device + one HIP stream; contexts are independent and a
 * context must not be used from two threads at once.  The caller owns every host buffer; the
 * library owns every device buffer it allocates until cx_ctx_destroy (an adopted grid pointer
 * stays the caller's).  Sample arrays are C-ordered fp32, A[n0][n1][n2], last axis fastest;
 * array axes (0,1,2) are the reference's (x,y,z).
 */
#ifndef CONTOURIST_HIP_H
#define CONTOURIST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cx_ctx cx_ctx;

typedef enum {
    CX_OK = 0,
    CX_ERR_INVALID = -1,     /* bad argument */
    CX_ERR_HIP = -2,         /* a HIP runtime call failed; see cx_last_error */
    CX_ERR_NOMEM = -3,       /* device or host allocation failed */
    CX_ERR_STATE = -4,       /* call out of order (no grid, no extraction yet, ...) */
    CX_ERR_CAPACITY = -5,    /* output buffers too small; counts say what is needed */
    CX_ERR_UNSUPPORTED = -6  /* e.g. more than 2^29 samples in one grid (32-bit vertex ids) */
} cx_status;

/* flags of cx_extract3d */
#define CX_DIAG_CANONICAL 0u   /* 2-2 tetrahedron: quad split follows the tetrahedron's vertex order */
#define CX_DIAG_CPYTHON310 1u  /* quad split reproduces CPython 3.10 set iteration order, i.e. the
                                  reference as it runs today (tetrahedral.py:592-595) */
#define CX_KERNEL_GENERIC 0x100u /* force the shape-agnostic classify kernel (default: auto) */
#define CX_KERNEL_STAGED 0x200u  /* emit through the staged kernels (vertex stage + words per queue entry + triangle stage): the default */
#define CX_KERNEL_FUSED 0x400u   /* emit vertices and triangles in ONE kernel over the cell queues (vertex indices of neighbour
                                    cells found through the queues, no table per sample, no cell records); same mesh, same
                                    numbering.  An extraction with samples inside the reference's np.allclose tolerances whose
                                    rules drop vertices is sent through the staged kernels automatically */
#define CX_KERNEL_TILED 0x800u   /* emit vertices and triangles tile by tile (one workgroup per tile of the streaming pass, the
                                    neighbour cells' vertex indices kept in LDS; the voxels on a tile's last row / last sample
                                    column by a small second kernel): same mesh, same numbering, ~0.35 GB less HBM traffic per
                                    512^3 extraction.  Sent through the staged kernels automatically when a sample sits inside the
                                    reference's np.allclose tolerances or a tile holds more surface cells than the LDS words */

typedef struct {
    int64_t n_cells;         /* lattice cells with a sign change among their corners */
    int64_t n_vertices;      /* interpolated edge crossings == len(interpolated_contour_pairs) */
    int64_t n_triangles;     /* == len(simplex_sets) before quantize_interpolations */
    int64_t n_border_voxels; /* voxels with a sign change for which GridContour.border_voxel() is true */
} cx_counts;

/* ---- context --------------------------------------------------------------------------------- */
int cx_ctx_create(int device_id, cx_ctx** out);
int cx_ctx_destroy(cx_ctx* ctx);
const char* cx_last_error(cx_ctx* ctx);
/* run on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int cx_set_stream(cx_ctx* ctx, void* hip_stream);
int cx_synchronize(cx_ctx* ctx);

/* ---- the sampled field ------------------------------------------------------------------------
 * Replaces grid_field.FunctionGrid.materialize_array / grid_function (grid_field.py:34-44, 95-118):
 * the dense fp32 array the march reads instead of calling f(x,y,z) per corner per use. */
int cx_grid_upload(cx_ctx* ctx, const float* host, int64_t n0, int64_t n1, int64_t n2);
int cx_grid_adopt_device(cx_ctx* ctx, const void* device_ptr, int64_t n0, int64_t n1, int64_t n2);
/* float64 originals of the bound fp32 samples (same dimensions; NULL drops them).  The reference interpolates a
 * crossing on the float64 values of its callable (tetrahedral.py:471-487); with these bound, Level 1
 * (cx_postprocess3d, cx_level0_points_f64) does the same, so crossings next to a weld-bucket boundary
 * (surface_geometry.py:30-48) fall on the reference's side of it.  The march itself stays on fp32. */
int cx_grid_shadow_f64(cx_ctx* ctx, const double* host, int64_t n0, int64_t n1, int64_t n2);

/* the grid is a sub-block of a larger volume starting at lattice point (o0,o1,o2) (slab partitions).
 * Only CX_DIAG_CPYTHON310 depends on it: the reference's set order hashes ABSOLUTE lattice
 * coordinates (tetrahedral.py:567-575), so slabs must hash global coordinates to agree with the
 * undivided volume.  Edge ids stay local; the caller adds (o0*n1*n2 + ...) << 3.
 * Negative origins are allowed (|o| < 2^30): an array that carries a rim of samples around the reference's grid starts
 * at lattice point -1 (CPython's hash(-1) == -2 is reproduced).  cx_level0_points_f64 adds the origin as well. */
int cx_set_origin(cx_ctx* ctx, int64_t o0, int64_t o1, int64_t o2);

/* optional: pre-size the output buffers (cells / vertices / triangles); 0 keeps the default. */
int cx_reserve(cx_ctx* ctx, int64_t max_cells, int64_t max_vertices, int64_t max_triangles);

/* ---- Level 0: the voxel march -------------------------------------------------------------------
 * Replaces, for a dense grid, in one pass over the samples:
 *   FunctionGrid.find_contour_crossing_grid_segments   grid_field.py:64-84     (crossing edges)
 *   GridContour.find_initial_voxels/expand_voxels/border_voxel  tetrahedral.py:383-469 (active voxels)
 *   GridContour3d.enumerate_voxel_triangles / enumerate_tetrahedron_triangles  tetrahedral.py:554-595
 *   GridContour.add_simplex / interpolate_pair / contour_pair_interpolation    tetrahedral.py:176-188, 471-512
 * Result (device resident): 8-byte vertex records {uint32 edge id, fp32 t}: the crossing sits at q + t*d on the lattice edge
 * q -> q+d, edge id = (linear index of the owning lattice point q << 3) | direction(1..7, = 4di+2dj+dk), t = (v - f(q)) /
 * (f(q+d) - f(q)) -- what contour_pair_interpolation keeps per pair (tetrahedral.py:471-512: the pair and its ratio); and
 * triangles as int32 index triples wound so the normal points from f<value to f>=value.  Level 1 recomputes every point in
 * float64 from the grid and the id alone; fp32 grid coordinates {x, y, z, bits(edge id)} are expanded from the records on
 * request (cx_level0_download, cx_level0_device_ptrs).
 * cx_extract3d enqueues the kernels and returns the counts (one device->host copy);
 * the _async form only enqueues.  Returns CX_ERR_CAPACITY (with valid counts) if a buffer was too
 * small: call cx_reserve with the counts and extract again. */
int cx_extract3d(cx_ctx* ctx, double value, uint32_t flags, cx_counts* out);
int cx_extract3d_async(cx_ctx* ctx, double value, uint32_t flags);
int cx_counts_get(cx_ctx* ctx, cx_counts* out);
/* Several isovalues of ONE grid in one call (BASELINE config 5; the reference classifies against all sorted levels at once only
 * in 2-D, multiple_2d_contour.py:17-30, 48-59).  The pass over the samples runs once for all levels (the levels' workgroups
 * stream a tile side by side: it comes from HBM once); scan, vertex and triangle stages then run per level, so every level's
 * mesh is bit for bit the cx_extract3d mesh of that isovalue.  values: nlevels (1..64) isovalues in any order;
 * flags: CX_DIAG_CANONICAL / CX_DIAG_CPYTHON310; out_counts: nlevels entries (may be NULL).  Every level keeps its own
 * device buffers, sized to its surface.  Needs rows of at least 4 samples (CX_ERR_UNSUPPORTED otherwise).
 * cx_levels_select makes level `index` the context's current extraction: cx_level0_download, cx_level0_points_f64,
 * cx_postprocess3d*, cx_select_seeded3d* ... then act on that level (level 0 is selected on return).  The levels stay valid
 * until the next cx_extract3d* / cx_grid_* call on the context. */
int cx_extract3d_levels(cx_ctx* ctx, const double* values, int32_t nlevels, uint32_t flags, cx_counts* out_counts);
int cx_levels_select(cx_ctx* ctx, int32_t index);
/* which kernels produced the last extraction: 0 generic classify + triangle stage, 1 staged pipeline (stream, scan, vertex
 * stage, triangle stage), 2 stream, scan + fused emit kernel -- for measurement (bench.py names its kernels by it) */
int cx_level0_path(cx_ctx* ctx, int* path);
/* copy the Level-0 mesh to host: verts = n_vertices*4 floats, tris = n_triangles*3 int32 */
int cx_level0_download(cx_ctx* ctx, float* verts_xyzk, int32_t* tris);
/* device pointers of the Level-0 mesh (valid until the next extract / reserve / destroy / call of this function): the
 * vertex records expanded to float4 {x, y, z, bits(edge id)} (enqueued on the context's stream) and the index triples */
int cx_level0_device_ptrs(cx_ctx* ctx, void** verts_xyzk, void** tris);
/* device pointers of the Level-0 buffers as the march leaves them: n_vertices x {uint32 edge id, fp32 t}, n_triangles x 3
 * int32 (no copy, no kernel; valid until the next extract / reserve / destroy) */
int cx_level0_device_records(cx_ctx* ctx, void** vertex_records, void** tris);
/* the same to host memory: vertex_records = n_vertices x 2 uint32 {edge id, bits of the fp32 fraction t}, tris as above -- half the
 * bytes of cx_level0_download for a host that keeps (pair, ratio) as the reference does (tetrahedral.py:471-512) */
int cx_level0_download_records(cx_ctx* ctx, uint32_t* vertex_records, int32_t* tris);

/* ---- Level 1: mesh post-passes -------------------------------------------------------------------
 * Replaces GridContour.quantize_interpolations (tetrahedral.py:190-215), remove_tiny_simplices
 * (:353-375), GridContour3d.extract_surface_geometry (:604-621), SurfaceGeometry.clean_triangles
 * (surface_geometry.py:14-50) and SurfaceGeometry.orient_triangles (:52-140) on the device.
 * corner = grid_dimensions of the reference (= n-1 per axis).  flags bit 0: skip clean_triangles
 * (the reference's clean=False).  out_counts (8 x int64): [0] vertices, [1] triangles,
 * [2] triangles after weld, [3] triangles after tiny collapse, [4] connected components. */
int cx_postprocess3d(cx_ctx* ctx, uint32_t flags, int64_t* out_counts);
/* same with GridContour.smooth_interpolations(smooth) (tetrahedral.py:329-351, 547-550) between the weld and
 * the tiny collapse; 0 < smooth <= 1, smooth == 0 disables it */
int cx_postprocess3d_ex(cx_ctx* ctx, uint32_t flags, double smooth, int64_t* out_counts);
/* Seeded selection.  Replaces GridContour.find_initial_voxels / expand_voxels (tetrahedral.py:396-463): the
 * reference only enumerates the surface voxels it reaches breadth-first (26 neighbours) from its end point
 * pairs; the dense march finds every surface voxel, and this call restricts the NEXT cx_postprocess3d* calls
 * (until the next extraction) to the triangles of the voxel groups reached from the given pairs.
 * endpoints_ijk: n x 6 int32 lattice points (i0,j0,k0, i1,j1,k1) whose samples straddle the isovalue; they are
 * bisected and mapped to seed voxels exactly as the reference does.  out_counts (4 x int64): [0] seed voxels,
 * [1] voxel groups kept, [2] triangles kept, [3] rejected pairs (the reference asserts on those; CX_ERR_INVALID).
 * range_lo_hi: NULL, or 6 int32 (lo[3], hi[3]) = the reference's in_range box (tetrahedral.py:465-469) when the
 * sample array carries a margin around the reference's grid (the reference evaluates its callable outside the
 * grid for seed voxels on the rim; growth stays inside the box).
 * Not calling it keeps every component, as the reference's exhaustive search_for_endpoints() does. */
/* The Level-1 scales (weld buckets int(10000/corner), tiny-simplex extent 1/corner) use corner = n-1 per axis.
 * When the sample array carries a margin around the reference's grid, hand over the reference's own corner
 * (voxels per axis); 0 restores the default. */
int cx_set_reference_corner(cx_ctx* ctx, int64_t c0, int64_t c1, int64_t c2);
/* The one exchange step of a volume split into slabs along array axis 0 (no counterpart in the single-process reference; SURVEY
 * section 8e): rank r owns n_own planes and marches them with ONE halo plane, the first plane of rank r+1.  local_planes: device
 * buffer of (n_own + 1) * plane_samples floats (n_own on the last rank); sends plane 0 to rank-1 and receives plane n_own from
 * rank+1 in one RCCL group on the context's stream, so extractions enqueued afterwards are ordered behind it.  rccl_comm: the
 * caller's ncclComm_t; RCCL is not linked but looked up in the calling process (CX_ERR_UNSUPPORTED if there is none).
 * world == 1: nothing to do.  (The Python host uses torch.distributed for the same step: contourist_amd/distributed.py.) */
int cx_halo_exchange(cx_ctx* ctx, void* rccl_comm, int rank, int world, float* local_planes, int64_t n_own, int64_t plane_samples);
/* A communicator owned by the context, for hosts that have none to hand over: cx_rccl_unique_id on ONE rank (128 bytes,
 * ncclGetUniqueId), the bytes carried to every rank by the host's own means, cx_rccl_comm_init (ncclCommInitRank, collective)
 * on every rank; cx_halo_exchange with rccl_comm == NULL then uses it.  Only an RCCL copy ALREADY loaded in the process is
 * used (CX_ERR_UNSUPPORTED otherwise): a PyTorch host carries its own. */
int cx_rccl_unique_id(uint8_t* out128);
int cx_rccl_comm_init(cx_ctx* ctx, const uint8_t* id128, int rank, int world);
int cx_rccl_comm_destroy(cx_ctx* ctx);
/* CX_OK if this process can run the C-side exchange (an RCCL copy is loaded and exports what is needed), CX_ERR_UNSUPPORTED if not.
 * Local, no collective: the host asks every rank and agrees on the answers BEFORE the collective cx_rccl_comm_init, so that a
 * rank without RCCL cannot leave the others waiting inside ncclCommInitRank. */
int cx_rccl_available(void);
/* the contexts of one rank (several extractions in flight) share ONE communicator: `ctx` uses `owner`'s from now on (the owner
 * must outlive it and keeps the ownership; cx_rccl_comm_destroy on `ctx` only drops the reference).  Exchanges are issued by the
 * host's single thread in volume order, each on the calling context's stream. */
int cx_rccl_comm_share(cx_ctx* ctx, cx_ctx* owner);
/* One rank's whole step for one volume of a slab-partitioned stream of volumes, as ONE call: adopt the device buffer
 * (n_own planes of n1 x n2 samples, followed by room for the halo plane unless rank == world - 1), exchange the halo on the
 * context's stream with the context's own communicator, enqueue cx_extract3d_async behind it.  (A 64-plane slab of a 512^3
 * volume is ~50 us of GPU time: the host side of a step has to be cheaper than that.) */
int cx_slab_step(cx_ctx* ctx, float* local_planes, int64_t n_own, int64_t n1, int64_t n2, int rank, int world, double value, uint32_t flags);
int cx_select_seeded3d(cx_ctx* ctx, const int32_t* endpoints_ijk, int64_t n, const int32_t* range_lo_hi, int64_t* out_counts);
/* flags CX_SEED_ALL_IN_RANGE: every voxel inside range_lo_hi is kept (the exhaustive search_for_endpoints() of the
 * reference, tetrahedral.py:74-81) and the end points only add the seed voxels OUTSIDE it: the reference does not
 * range-check the voxels it starts from (:396-441), so a surface that reaches the rim of the grid gets triangles in
 * the voxels one step outside, next to the crossing lattice segments on the rim. */
#define CX_SEED_ALL_IN_RANGE 1u
/* CX_SEED_PARALLEL: one thread per end point pair without the reference's shared `visited` set (what more than 65 536
 * pairs get anyway); each point then yields its own voxel or its first border neighbour */
#define CX_SEED_PARALLEL 2u
int cx_select_seeded3d_ex(cx_ctx* ctx, const int32_t* endpoints_ijk, int64_t n, const int32_t* range_lo_hi, uint32_t flags,
                          int64_t* out_counts);
/* the masks of the last selection on the host (n_triangles and n_vertices bytes; all ones without a selection): for
 * callers that refine the Level-0 mesh on the host before cx_postprocess3d_mesh -- linear_interpolate=False
 * re-evaluates the caller's function between the lattice points (tetrahedral.py:488-505) */
int cx_seeded_masks_download(cx_ctx* ctx, uint8_t* tri_keep, uint8_t* vert_keep);
/* how the last cx_select_seeded3d* / cx_select_seeded4d* call ran its end points: 0 = one after the other with the reference's
 * shared `visited` set (tetrahedral.py:396-441; up to 65 536 pairs in 3-D, 16 384 in 4-D), 1 = one thread per pair
 * (CX_SEED_PARALLEL, or more pairs than that): adjacent candidate voxels may then be picked differently where pairs collide */
int cx_seeded_mode(cx_ctx* ctx, int* mode);
/* Level 1 of a mesh assembled by the caller -- the way several GPUs share one volume: every rank marches its slab
 * (cx_extract3d), takes the float64 coordinates the reference would have interpolated (cx_level0_points_f64: nv*3
 * doubles in the order of cx_level0_download, in the grid coordinates of the whole volume = lattice point + origin of
 * cx_set_origin; tetrahedral.py:471-487) and its triangles, the owner of the result
 * concatenates them in ASCENDING GLOBAL EDGE-ID ORDER (ids are global by formula; the index of a vertex is then its
 * priority in the canonical choices) and runs the same weld / tiny collapse / clean / orient on them:
 * corner = voxels per axis of the WHOLE volume, coordinates in its grid coordinates, triangles wound as the march
 * wound them (flags bit 2 set: windings are arbitrary, propagate them like cx_surface_geometry); flags bit 0 as in
 * (flags bit 3 set: the triangles are ones the march emitted -- slab meshes assembled on the host, a Level-0 mesh with refined points --, i.e.
 * an edge lies on at most two of them until the post-pass merges something into one of its ends: the components are then linked block by
 * block in LDS as in cx_postprocess3d instead of through the full edge table; a mesh of any other origin must leave it clear)
 * cx_postprocess3d.  Results through cx_level1_download. */
int cx_level0_points_f64(cx_ctx* ctx, double* points_xyz);
int cx_postprocess3d_mesh(cx_ctx* ctx, const double* points_xyz, int64_t nv, const int32_t* tris, int64_t nt, const int64_t* corner3,
                          uint32_t flags, double smooth, int64_t* out_counts);
/* copy the Level-1 mesh to host: points = nv*3 doubles (grid coordinates), tris = nt*3 int32 */
int cx_level1_download(cx_ctx* ctx, double* points_xyz, int32_t* tris);
/* device pointers of the same mesh (no copy): points = n_vertices*3 doubles (grid coordinates), tris = n_triangles*3 int32, valid
 * until the next post-pass / extraction / destroy on the context; the context's stream is synchronised before they are handed out.
 * For consumers on the GPU: what get_points_and_triangles() returns (tetrahedral.py:83-87, 528-552) without the trip to the host. */
int cx_level1_device_ptrs(cx_ctx* ctx, void** points_xyz, void** tris, int64_t* n_vertices, int64_t* n_triangles);
/* edge id ((linear index of the lower lattice point << 3) | direction, local to the marched array) of every vertex of
 * cx_level1_download, in its order: the representative that survived weld, tiny collapse and clean-up -- the identity of a
 * Level-1 vertex across slabs (after cx_postprocess3d_mesh: the index of the input vertex). */
int cx_level1_download_keys(cx_ctx* ctx, uint32_t* keys);

/* ---- Level 1 sharded over slabs (the reference is one process: tetrahedral.py:190-215, 353-375,
 * surface_geometry.py:14-140).  The context holds the extraction of a slab marched together with TWO layers of cells of
 * each neighbour (cx_set_origin: where the local array starts in the whole volume; cx_set_reference_corner: the corner of
 * the whole volume).  begin: weld / tiny collapse / clean-up of everything local, components of the own and first-layer
 * triangles.  What the ranks must agree on follows the slab BOUNDARY: _boundary hands out (hash of the three original edge
 * ids in the whole volume's numbering, component label) for this slab's own triangles next to its lower neighbour
 * (which = 1) and for its copies of the upper neighbour's first layer (which = 4) -- rank r's list 4 and rank r+1's list 1
 * are the same triangles, sorted by hash they pair the labels up; hash / label may be DEVICE pointers.  _candidates: per
 * component that reaches a neighbour, the start-triangle candidate among the own triangles (surface_geometry.py:79-103).
 * finish: the agreed flip per such component; afterwards cx_level1_download / _keys / cx_level1_write hand out the slab's
 * OWN part of the mesh in whole-volume coordinates.  own_lo / own_hi: own cell layers [lo, hi) of the local array.
 * flags as cx_postprocess3d. */
int cx_postprocess3d_shard_begin(cx_ctx* ctx, uint32_t flags, int64_t own_lo, int64_t own_hi, int64_t* out_counts8,
                                 int64_t* n_own_lower, int64_t* n_upper_copies, int64_t* n_candidates);
int cx_postprocess3d_shard_boundary(cx_ctx* ctx, int which, uint64_t* hash, uint32_t* label);
int cx_postprocess3d_shard_candidates(cx_ctx* ctx, uint32_t* cand_label, double* cand_x, uint32_t* cand_vertex_key, double* cand_nx,
                                      uint8_t* cand_negative, uint8_t* cand_has);
int cx_postprocess3d_shard_finish(cx_ctx* ctx, const uint32_t* labels, const uint8_t* flips, int64_t n, int64_t* out_counts8);

/* Binary mesh file straight from the Level-1 device buffers -- the step right after get_points_and_triangles() for every caller
 * of the reference (html_demo.py:118-161), without materialising the mesh in the caller's address space: the file's records are
 * laid out on the device and streamed through pinned staging buffers.  format CX_FILE_PLY: binary little-endian PLY, float64
 * x y z per vertex, faces as uchar 3 + three int32 (what contourist_amd.mesh_io.write_ply writes from host arrays, byte for
 * byte); CX_FILE_GLTF_BIN: the .bin payload of a glTF 2.0 mesh, float32 positions followed by uint32 indices.
 * mins_delta: NULL (grid coordinates) or {min_x, min_y, min_z, delta_x, delta_y, delta_z} for FunctionGrid.from_grid_coordinates
 * (grid_field.py:89-93).  out_info (9 doubles, may be NULL): vertices, triangles, bytes written, min xyz, max xyz of the
 * positions as written (glTF accessor bounds; PLY: not computed).  Face order is the device's (the Python API sorts the rows). */
#define CX_FILE_PLY 0
#define CX_FILE_GLTF_BIN 1
int cx_level1_write(cx_ctx* ctx, int format, const char* path, const double* mins_delta, double* out_info);

/* ---- standalone SurfaceGeometry operator ---------------------------------------------------------
 * SurfaceGeometry(vertices, triangles).clean_triangles() / .orient_triangles()
 * (surface_geometry.py:6-12, 14-50, 52-140) on caller-supplied host arrays.
 * mode 0 = orient_triangles only, 1 = clean_triangles then orient_triangles, 2 = clean_triangles only.
 * Outputs are written in place: *nv / *nt are updated, points (nv*3 doubles) and tris (nt*3 int32)
 * are overwritten with the cleaned / oriented mesh (vertices compacted to those still in use). */
int cx_surface_geometry(cx_ctx* ctx, double* points_xyz, int64_t* nv, int32_t* tris, int64_t* nt,
                        int mode);

/* ---- 4-D: marching pentatopes ------------------------------------------------------------------------
 * Replaces, for a dense grid A[n0][n1][n2][n3] (last axis = t, fastest), GridContour4D's hyper-voxel
 * march: find_initial_voxels/expand_voxels/border_voxel with the 16-corner HYPERCUBE (pentatopes.py:94-95,
 * tetrahedral.py:383-469), enumerate_voxel_tetrahedra / enumerate_pentatope_tetrahedra
 * (pentatopes.py:216-291) and the edge interpolation (tetrahedral.py:471-512).
 * Result: vertex records float4 {x,y,z,t} in grid coordinates, edge ids
 * (linear index of the lower lattice point << 4) | direction (1..15 = 8di+4dj+2dk+dl), and
 * tetrahedra as 4 int32 vertex indices (cx_counts.n_triangles counts tetrahedra here).
 * CX_DIAG_CPYTHON310 reproduces the reference's 2-3 split (set order of 4-tuples, pentatopes.py:255-256). */
int cx_grid4d_upload(cx_ctx* ctx, const float* host, int64_t n0, int64_t n1, int64_t n2, int64_t n3);
int cx_grid4d_adopt_device(cx_ctx* ctx, const void* device_ptr, int64_t n0, int64_t n1, int64_t n2, int64_t n3);
int cx_set_origin4d(cx_ctx* ctx, int64_t o0, int64_t o1, int64_t o2, int64_t o3);
int cx_extract4d(cx_ctx* ctx, double value, uint32_t flags, cx_counts* out);
/* The same march enqueued on the context's stream and not waited for (one volume after the other on two contexts, as
 * cx_extract3d_async / cx_counts_get for the 3-D path); cx_counts4d_get waits for it, validates the counters and, if a buffer was too
 * small, grows it and runs the march again synchronously.  CX_ERR_STATE without an extraction in flight. */
int cx_extract4d_async(cx_ctx* ctx, double value, uint32_t flags);
int cx_counts4d_get(cx_ctx* ctx, cx_counts* out);
/* Seeded selection in 4-D, the counterpart of cx_select_seeded3d: GridContour4D(corner, function, value, endpoints)
 * (pentatopes.py:92-100) grows from its end points with the methods it inherits (tetrahedral.py:396-463) over the 80
 * neighbours of pentatopes.py:32-39.  endpoints_ijkl: n x 8 int32 lattice points (i0,j0,k0,l0, i1,j1,k1,l1) whose
 * samples straddle the isovalue.  Call between cx_extract4d and cx_postprocess4d; restricts the post-pass (until the
 * next extraction) to the tetrahedra of the hyper-voxel groups reached.  out_counts (4 x int64): [0] seed voxels,
 * [1] groups kept, [2] tetrahedra kept, [3] rejected pairs (CX_ERR_INVALID). */
int cx_select_seeded4d(cx_ctx* ctx, const int32_t* endpoints_ijkl, int64_t n, int64_t* out_counts);
/* The same with the reference's in_range box (tetrahedral.py:465-469) given explicitly: range_lo_hi = lo[4], hi[4] in array
 * coordinates, lo <= hyper-voxel < hi (NULL: the whole array).  For an array that carries a rim of samples around the reference's
 * grid (origin -1, cx_set_origin4d): the growth stays inside the grid, seed voxels in the rim are kept and grow one step into it
 * -- the reference does not range-check the voxels it starts from (tetrahedral.py:396-441, pentatopes.py:528-551).
 * flags: CX_SEED_ALL_IN_RANGE as for cx_select_seeded3d_ex (the exhaustive search on a surface that leaves the grid). */
int cx_select_seeded4d_ex(cx_ctx* ctx, const int32_t* endpoints_ijkl, int64_t n, const int32_t* range_lo_hi, uint32_t flags, int64_t* out_counts);
/* the selection as a mask over the Level-0 tetrahedra (n_tetrahedra bytes, 1 = kept) */
int cx_seeded4d_mask_download(cx_ctx* ctx, uint8_t* tet_keep);
int cx_level0_4d_download(cx_ctx* ctx, float* verts_xyzt, uint32_t* edge_ids, int32_t* tets);
/* GridContour4D.find_tetrahedra's post-steps on the device: bin_times(nbins) (pentatopes.py:162-169),
 * drop_instant_tetrahedra(1e-7) (:171-189), remove_tiny_simplices(1e-3) (tetrahedral.py:353-375).
 * out_counts (8 x int64): [0] vertices (unchanged numbering), [1] surviving tetrahedra,
 * [2] after drop_instant, [3] after the tiny collapse.  Download: points = nv*4 doubles, tets = nt*4 int32. */
int cx_postprocess4d(cx_ctx* ctx, int32_t nbins, int64_t* out_counts);
/* the same on points handed over by the caller: n_vertices * 4 doubles in the order of cx_level0_4d_download, in the reference's
 * lattice -- linear_interpolate=False re-evaluates the caller's function between the lattice points (tetrahedral.py:488-505),
 * which is Python; NULL = cx_postprocess4d */
int cx_postprocess4d_points(cx_ctx* ctx, int32_t nbins, const double* points_xyzt, int64_t* out_counts);
int cx_level1_4d_download(cx_ctx* ctx, double* points_xyzt, int32_t* tets);
/* GridContour4D.collect_morph_triangles (pentatopes.py:314-368) + MorphTriangles.orient_triangles
 * (morph_geometry.py:49-89) on the tetrahedra left by cx_postprocess4d: every tetrahedron is sliced at the
 * midpoints between its distinct vertex times into 1-2 triangles whose corners are SEGMENTS (pairs of 4-D
 * points, stored low t -> high t); triangles are oriented per connected component, propagating only between
 * triangles whose time ranges overlap.  out_counts (8 x int64): [0] points, [1] segments, [2] triangles,
 * [4] components.  Download: points nv*4 doubles, segments ns*2 int32 (point indices), triangles nt*3 int32
 * (segment indices). */
int cx_morph_triangles(cx_ctx* ctx, int64_t* out_counts);
int cx_morph_download(cx_ctx* ctx, double* points_xyzt, int32_t* segments, int32_t* triangles);
/* The surface at time t from the morph triangles of the last cx_morph_triangles: the consumer-side evaluation of
 * misc/morph_triangles.js:26-140 (a triangle is visible while t is inside the intervals of all three of its
 * segments; corners = points of the segments at t).  out_counts (2 x int64): [0] points, [1] triangles.
 * Download: points np*3 doubles, triangles nt*3 int32 (indices into those points, original triangle order). */
int cx_morph_eval(cx_ctx* ctx, double t, int64_t* out_counts);
int cx_morph_eval_download(cx_ctx* ctx, double* points_xyz, int32_t* triangles);
/* The same for n_times times in ONE set of launches -- the per-t isosurface stream the viewer plays frame by frame
 * (misc/morph_triangles.js:117-204 once per frame; here: all frames of a stream at once).  times: any order, repeats allowed.
 * out_counts (n_times x 2 x int64): per time [0] points, [1] triangles.  Surface i is exactly what cx_morph_eval(times[i]) returns
 * (same points, same index triples, same order); the surfaces lie one behind the other in device memory.
 * _download: surface i of the last call (points np*3 doubles, triangles nt*3 int32, indices local to the surface);
 * _device_ptrs: where surface i lies on the device (NULL for an empty one) -- valid until the next evaluation on this context.
 * cx_morph_triangles leaves segments and triangles sorted by their start time, so that the triangles (and segments) that exist at one
 * time are a window of ids: a call costs what its windows hold, not what the whole morph holds. */
int cx_morph_eval_many(cx_ctx* ctx, const double* times, int32_t n_times, int64_t* out_counts);
int cx_morph_eval_many_download(cx_ctx* ctx, int32_t i, double* points_xyz, int32_t* triangles);
int cx_morph_eval_many_device_ptrs(cx_ctx* ctx, int32_t i, void** points_xyz, void** triangles);
/* every surface of the last call in ONE transfer: the points of surface 0, 1, ... one behind the other (sum of the point counts x 3
 * doubles), the triangles likewise (indices local to their surface) */
int cx_morph_eval_many_download_all(cx_ctx* ctx, double* points_xyz, int32_t* triangles);

/* ---- 2-D contour lines at several isovalues ------------------------------------------------------
 * Replaces triangulated.Grid2DContour (search_grid :198-212, find_initial_contour_pairs :299-320,
 * expand_contour_pairs :322-331, get_contour_sequences :226-297; triangulated.py) for one sample array and ALL
 * isovalues of multiple_2d_contour.Multiple2DContourGrid.get_contours_dictionary (multiple_2d_contour.py:17-30,
 * level lookup by bisection as in classify_endpoint_values :48-59) in one pass over the samples.
 * samples: fp32 A[n][m] (host pointer, or device pointer when on_device != 0; array axes (0,1) are the reference's
 * (x,y)); values: ascending, distinct.  A sample equal to an isovalue counts as high.
 * seeds: nseeds x (i, j, role, level index) int32 -- the polylines grown from the pairs around lattice point (i,j)
 * as their low (role 0) or high (role 1) end are kept, exactly the reference's seeded growth; nseeds == 0: the seeds
 * of the reference's own grid search (every crossing axis edge that starts at i < n-1, j < m-1).
 * Limits: n*m <= 2^30 samples, at most 65535 values, fewer than 2^29 crossings (all levels together) per call
 * (CX_ERR_UNSUPPORTED otherwise: contour fewer levels per call).
 * mins_delta: {min_x, min_y, delta_x, delta_y} for FunctionGrid.from_grid_coordinates (grid_field.py:89-93), or NULL
 * for grid coordinates.
 * Output (cx_contour2d_download): points n_points x 2 float64, polyline by polyline; keys n_points int64 =
 * ((3*(i*m+j) + d) << 16) | level index for the crossing of the lattice edge from (i,j) in direction d (0: (1,0),
 * 1: (0,1), 2: (1,1)); chains: one cx_chain2d per polyline, ordered by their first crossing's id. */
#define CX2_ALL_CHAINS 1u   /* keep every polyline (no seeded growth) */
#define CX2_NO_DEDUPE 2u    /* keep points that are np.allclose to their predecessor (triangulated.py:268) */
#define CX2_SEARCH_SEEDS 4u /* with explicit seeds: add the seeds of the grid search too (multiple_2d_contour.py:39-41) */
typedef struct {
    uint32_t n_points;   /* points in the polylines returned */
    uint32_t n_chains;   /* polylines returned */
    uint32_t n_pairs;    /* crossings found on the whole lattice, all levels */
    uint32_t n_levels;
} cx_counts2d;
typedef struct {
    int32_t level;       /* index into values */
    int32_t closed;      /* the `closed` flag of get_contour_sequences */
    uint32_t first;      /* first point */
    uint32_t count;      /* number of points */
} cx_chain2d;
int cx_contour2d_extract(cx_ctx* ctx, const float* samples, int on_device, int64_t n, int64_t m, const double* values, int32_t nvalues,
                         const int32_t* seeds, int64_t nseeds, uint32_t flags, const double* mins_delta, cx_counts2d* out);
int cx_contour2d_download(cx_ctx* ctx, double* points_xy, int64_t* keys, cx_chain2d* chains);

/* ---- measurement ----------------------------------------------------------------------------------
 * When enabled, every extract records HIP events around its kernels on the context's stream.
 * cx_timing_read synchronises and returns the summed milliseconds since the last reset:
 * ms[0] classify + vertex stage (all of its kernels), ms[1] triangle emit kernel, ms[2] whole extract,
 * ms[3] stream kernel, ms[4] scan kernel, ms[5] cell / vertex emit kernel (staged pipeline; the generic
 * classify kernel is reported in ms[3]), ms[6..7] reserved.  *n = number of extracts accumulated. */
int cx_timing_enable(cx_ctx* ctx, int on);
int cx_timing_read(cx_ctx* ctx, double ms[8], int* n);
/* what a plain streaming READ of `bytes` bytes at device_ptr reaches on this device (16-byte loads, grid-stride; best of `reps`
 * launches on the context's stream, HIP events): the measured ceiling beside the 8 TB/s data-sheet figure that bench.py quotes
 * the stream kernel against ("also report measured stream-read BW on the box").  No counterpart in the reference. */
int cx_measure_read_bandwidth(cx_ctx* ctx, const void* device_ptr, int64_t bytes, int reps, double* out_GBps);

/* diagnostics: per-wave s_memtime stamps of the stream kernel (4 words per wave: [0] start, [1] end; [2..3]
 * unused).  words > 0 allocates, host != NULL copies out, 0/NULL frees. */
int cx_debug_stamps(cx_ctx* ctx, int64_t words, unsigned long long* host);

/* library build info: "gfx950;<git describe or date>" */
const char* cx_version(void);

#ifdef __cplusplus
}
#endif
#endif
