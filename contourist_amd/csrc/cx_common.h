// cx_common.h -- shared declarations of the gfx950 isosurface extractor (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/contourist_hip.h"
#include "cx_tables.h"

// debug / tuning switches are honoured only in a process started with CX_DEBUG=1 (tools/, never the product path)
#include <cstdlib>
#include <cstring>
static inline bool cx_debug_enabled() {
    static const bool on = [] { const char* e = getenv("CX_DEBUG"); return e && strcmp(e, "1") == 0; }();
    return on;
}
static inline uint32_t cx_debug_knob(const char* name, uint32_t dflt) {
    if (!cx_debug_enabled()) return dflt;
    const char* e = getenv(name);
    return (e && atoi(e) > 0) ? (uint32_t)atoi(e) : dflt;
}

// ---- exact 32-bit division by a runtime-constant divisor (host computes the magic) -------------
struct cx_fdiv {
    uint32_t mul, sh1, sh2, d;
};

static inline cx_fdiv cx_fdiv_make(uint32_t d) {
    cx_fdiv f;
    uint32_t L = 0;
    while ((1ull << L) < d) L++;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    f.sh1 = L < 1 ? L : 1;
    f.sh2 = L > 0 ? L - 1 : 0;
    f.d = d;
    return f;
}

__device__ __forceinline__ uint32_t cx_div(uint32_t n, const cx_fdiv& f) {
    uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// Level-0 vertex record: 8 bytes.  x = edge id = (linear index of the owning lattice point q << 3) | direction d (1..7 = 4di+2dj+dk),
// y = bits of the fp32 fraction t in (v - f(q)) / (f(q+d) - f(q)): the crossing sits at q + t*d.  Level 1 recomputes the point in
// float64 from the grid and the id alone (cxp_k_vertices_f64, as the reference does: tetrahedral.py:471-512); callers that want
// fp32 grid coordinates get them expanded on request (cx_k_expand_verts: cx_level0_download / cx_level0_device_ptrs).
typedef uint2 cx_vrec;

// ---- parameters of one extraction ---------------------------------------------------------------
struct cx_params {
    const float* grid;     // n0*n1*n2 fp32 samples
    uint32_t n0, n1, n2;
    uint32_t nsamples;     // n0*n1*n2  (<= 2^29)
    cx_fdiv div_plane;     // / (n1*n2)
    cx_fdiv div_row;       // / n2
    float vcmp;            // smallest fp32 >= value :  (double)f < value  <=>  f < vcmp
    float near_abs;        // fp32 screen: |f - vcmp| <= near_abs is a superset of both float64 np.allclose tests
    float vhi, vlo;        // isovalue as an unevaluated fp32 sum (vhi + vlo == value to ~2^-48 relative)
    double value;          // isovalue (float64, as the reference computes)
    double tol_value;      // 1e-8 + 1e-5*|value|   (np.allclose(values, value), tetrahedral.py:576)
    uint32_t flags;
    uint32_t org0, org1, org2;   // lattice offset of this array inside a larger volume (hash order only)
    // outputs
    uint64_t* celltab;     // [nsamples] per lattice cell that owns a vertex: (crossing mask << 32) | first vertex index
    cx_vrec* verts;        // [vcap]  {edge id, bits(fp32 fraction t from the owning lattice point)}
    uint4* cells;          // [ccap]  {lin, sign|tetskip<<8|ntri<<16|emask<<24, tri base, first own vertex}
    int32_t* tris;         // [tcap*3]
    uint32_t vcap, ccap, tcap;
    uint32_t* counters;    // [0] cells [1] verts [2] tris [3] border voxels
    unsigned long long* stamps;   // diagnostic builds only: 4 s_memtime stamps per classify wave (else null)
    // staged pipeline (stream -> scan -> vertices -> triangles)
    uint32_t* queue;          // [nwaves * wcap] packed active-cell entries, one region per streaming wave
    struct cx_brec* brec;     // [nwaves * bcap] batch records of each streaming wave
    struct cx_wsum* wsum;     // [nwaves] what each streaming wave found
    struct cx_wbase* wbase;   // [nwaves] first vertex / triangle / record / batch index of each wave
    struct cx_bdesc* flat;    // [fcap] all batches, self-contained
    uint32_t fcap;
    // fused emit (cx_k_emit_mesh): vertex indices of neighbour cells are looked up through the stream kernel's queues --
    // per streaming wave CX_SWP plane steps of 64 lanes each
    uint32_t* qa;             // [nwaves][CX_SWP][64] (position in the wave's queue of the lane's first active cell << 16) | its active cells, bit 4r+m
    uint32_t* info;           // [queue size] per queue entry: first vertex of the cell relative to its wave's first | crossing mask >> 1 << 24
    const uint8_t* hbytes;    // [nsamples] CPython set-order code of each lattice point (CX_DIAG_CPYTHON310), or null
    uint64_t* info64;         // [queue size] staged kernels, per queue entry: (crossing mask << 32) | first vertex index -- the dense
                              // successor of `celltab` (which only the generic classify kernel still fills)
    cx_fdiv div_ci;           // / (cell planes per task)
    uint32_t* chunksum;       // [ceil(nwaves / 256)][8] totals (v, t, c, b, nb, near) of every 256 streaming waves, added up by the stream kernel
    uint32_t fused;           // 1: the fused emit kernel follows (no per-cell table, no cell records)
    uint32_t* rstart;         // [nvw] vertex stage: the batch in which wave m's share of the rounds starts (written by the scan kernel)
    uint32_t nvw;             // waves of the vertex stage (4 x its grid)
    uint32_t qlimit;          // queue entries a streaming wave may store (T.wcap; less when levels share a pool)
    uint32_t* kstart;         // [nkw] triangle stage: the batch in which wave m's share of the rounds starts (as rstart for the vertex stage)
    uint32_t nkw;             // waves of the triangle stage (4 x its grid)
    uint32_t write_records;   // 1: the vertex stage also writes the 16-byte cell records (seeded selection, the record-walking triangle kernel)
    // the stream of interpolation fractions (round 4): the stream kernel, which has the samples, computes t = (v - f(q)) / (f(q+d) - f(q))
    // of every crossing and writes them per streaming wave in the order the vertex stage numbers the vertices; the vertex stage reads
    // them back in order instead of gathering samples from half of the grid's cache lines.  null: the vertex stage gathers (levels, fused).
    float* tq;                // [nwaves * wcap] one region per streaming wave, filled from the front
    uint32_t tlimit;          // fractions a streaming wave may store (T.wcap); a wave with more hands its cells to the per-cell path
    // tile emit path (cx_tile3d.h): the words (crossing mask << 32 | first vertex) of the cells on the j- and k-faces of the tiles,
    // and the voxels that need them (a neighbour cell in the next tile in j or k)
    uint64_t* fj;             // [n0][4 * njg][n2]: rows j % 8 == 0 (slot 2 (j / 8)) and j % 8 == 7 (slot 2 (j / 8) + 1)
    uint64_t* fk;             // [n0][n1][2 * nks]: samples k % 256 == 0 (slot 2 ks) and k % 256 == 255 (slot 2 ks + 1)
    uint4* bnd;               // [2 nblocks][T.bndcap] (per half tile) boundary records {lin, sign|tetskip<<8|ntri<<16|emask<<24, first triangle, first vertex}
    uint32_t* bndn;           // [2 nblocks] how many
    uint32_t* torder;         // [3][2 nblocks] half tiles by class of work (most queue entries first; empty ones in none), filled by the scan kernel
    uint32_t tile_cap;        // info words a workgroup of the tile kernel holds in LDS (queue entries of its tile + the next chunk's first plane)
};
#ifndef CX_SWP
#define CX_SWP 16u            // plane slots per streaming wave: cell planes per task (ci) + 1, ci <= 15
#endif

struct cx_wsum {   // 32 bytes
    uint32_t nb;           // batches
    uint32_t v, t, c, b;   // vertices, triangles, cell records, border voxels of the wave's cells
    uint32_t nq;           // queued cells
    uint32_t near;         // a sample of the wave's region lies inside the tolerance screen: per-cell path
    uint32_t nr;           // rounds of 64 queued cells, batch by batch (sum of ceil(n / 64))
};
struct cx_wbase {  // 16 bytes
    uint32_t v, t, c, boff;
};
struct cx_brec {   // 32 bytes: cells [qoff, qoff+n) of the wave's queue and what precedes them in the wave
    uint32_t qoff, n, vpre, tpre, cpre, near, pad0, pad1;
};

struct cx_bdesc {  // 48 bytes: one batch as the emit kernels need it
    uint32_t w;            // streaming wave
    uint32_t qofs, n;      // its cells: queue[qofs .. qofs+n)
    uint32_t vbase, tbase, cbase;   // first vertex / triangle / cell record
    uint32_t near;
    uint32_t rbase;        // rounds of 64 cells in the batches before this one
    uint32_t p, j0, k0;    // the streaming wave's tile: first plane, first row, first sample (what cx_tile_of computes from w)
    uint32_t nsteps;       // planes of cells in the tile
};

// launch geometry of the staged pipeline: workgroup -> (k segment of 256 samples, group of 16 rows, chunk of ci planes)
struct cx_task {
    uint32_t ci;           // cell planes per task
    uint32_t nks, njg, nic;
    uint32_t nblocks;      // nks * njg * nic
    uint32_t chunk;        // tasks per XCD (grid = 8 * chunk workgroups)
    uint32_t wcap;         // queue entries per wave (every cell of its task)
    uint32_t bcap;         // batch records per wave
    uint32_t bndcap;       // boundary records per half tile (tile emit path): the voxels of its last row and last sample column
    cx_fdiv div_nks, div_njg;   // / nks, / njg: a streaming wave's tile from its number (triangle stage)
};

// debug / ablation flags (timing experiments only; results are wrong when set)
#define CX_DBG_PHASE_A_ONLY 0x10000u   // classify kernel: stream + queue, skip phase B
#define CX_DBG_COUNT_ONLY 0x20000u     // phase B: count pass only
#define CX_DBG_NO_CELLTAB 0x40000u     // phase B: skip the per-cell table store
#define CX_DBG_NO_VERTS 0x80000u       // phase B: skip vertex record stores
#define CX_DBG_NO_CELLS 0x100000u      // phase B: skip cell record stores
#define CX_DBG_NO_LOOKUP 0x200000u     // emit kernel: skip neighbour table lookups
#define CX_DBG_NO_TRIS 0x400000u       // emit kernel: skip triangle stores
#define CX_DBG_NO_EMIT 0x800000u       // skip the emit kernel
#define CX_DBG_NO_NEAR 0x2000000u      // stream kernel: ignore the tolerance screen (never take the per-cell path)
#define CX_DBG_HASH64 0x4000000u       // triangle kernels: corner hashes in full 64-bit arithmetic (no table, no 32-bit fast form)
#define CX_DBG_NO_VLOADS 0x1000000u    // phase B (packed entries): no sample loads for the vertices

enum { CX_CNT_CELLS = 0, CX_CNT_VERTS = 1, CX_CNT_TRIS = 2, CX_CNT_BORDER = 3, CX_CNT_BATCHES = 4,
       CX_CNT_NEAR = 5,   // streaming waves that met a sample inside the tolerance screen (their cells take the per-cell path)
       CX_CNT_ROUNDS = 6, // rounds of 64 queued cells over all batches (what the vertex stage divides among its waves)
       CX_CNT_OVERFLOW = 7, // a streaming wave found more cells than its slice of a shared queue pool holds (cx_extract3d_levels)
       CX_CNT_TILEOVF = 8,  // tile emit path: a tile holds more active cells than a workgroup's LDS words (the host runs the staged kernels instead)
       CX_CNT_TCLS = 9,     // ... 9, 10, 11: half tiles in each class of P.torder (zeroed by the stream kernel, counted by the scan kernel)
       CX_CNT_WORDS = 16 };

// device tables (defined in cx_march3d.hip)
extern __device__ __constant__ uint8_t cx_d_tet_corners[6][4];
extern __device__ __constant__ uint32_t cx_d_tet_tris[6][16][2];
extern __device__ __constant__ uint8_t cx_d_voxel_ntri[256];

// kernel launchers (cx_march3d.hip)
void cx_launch_classify_generic(const cx_params& P, hipStream_t s);
bool cx_fast_classify_supported(const cx_params& P);
bool cx_fast_classify_supported_dims(int64_t n2, const float* grid);
cx_task cx_fast_task(uint32_t n0, uint32_t n1, uint32_t n2);
void cx_launch_stream(const cx_params& P, const cx_task& T, hipStream_t s);
void cx_launch_scan_waves(const cx_params& P, const cx_task& T, hipStream_t s);
void cx_launch_stream_levels(const cx_params* host_params, const cx_task& T, uint32_t nlevels, hipStream_t s);   // parameters by value, 8 levels per launch
void cx_launch_scan_levels(const cx_params* device_params, const cx_task& T, uint32_t nlevels, hipStream_t s);
void cx_launch_emit_vertices(const cx_params& P, const cx_task& T, hipStream_t s);
uint32_t cx_vertex_stage_waves(const cx_params& P);
uint32_t cx_triangle_stage_waves(const cx_params& P);   // -> cx_params::nvw (the scan kernel needs it before the vertex stage runs)
void cx_launch_emit_triangles(const cx_params& P, const uint64_t* hash_xy, hipStream_t s);
void cx_launch_emit_triangles_q(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s);
void cx_launch_emit_triangles_e(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s);
void cx_launch_emit_mesh(const cx_params& P, const cx_task& T, hipStream_t s);
void cx_launch_tile_emit(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s);   // cx_k_tile_emit + cx_k_tile_boundary
uint32_t cx_tile_cap_default();
void cx_launch_hash_bytes(uint8_t* table, const uint64_t* hash_xy, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t org2, hipStream_t s);
void cx_launch_expand_verts(const cx_vrec* recs, float4* out, uint32_t n, uint32_t n1, uint32_t n2, hipStream_t s);
void cx_launch_hash_xy(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t org0, uint32_t org1, hipStream_t s);
