// cx_common.h -- shared declarations of the gfx950 isosurface extractor (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/contourist_hip.h"
#include "cx_tables.h"

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
    float4* verts;         // [vcap]  {x,y,z,bits(edge id)}
    uint4* cells;          // [ccap]  {lin, sign|tetskip<<8|ntri<<16|emask<<24, tri base, first own vertex}
    int32_t* tris;         // [tcap*3]
    uint32_t vcap, ccap, tcap;
    uint32_t* counters;    // [0] cells [1] verts [2] tris [3] border voxels
    unsigned long long* stamps;   // diagnostic builds only: 4 s_memtime stamps per classify wave (else null)
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

enum { CX_CNT_CELLS = 0, CX_CNT_VERTS = 1, CX_CNT_TRIS = 2, CX_CNT_BORDER = 3, CX_CNT_WORDS = 8 };

// device tables (defined in cx_march3d.hip)
extern __device__ __constant__ uint8_t cx_d_tet_corners[6][4];
extern __device__ __constant__ uint32_t cx_d_tet_tris[6][16][2];
extern __device__ __constant__ uint8_t cx_d_voxel_ntri[256];

// kernel launchers (cx_march3d.hip)
void cx_launch_classify_generic(const cx_params& P, hipStream_t s);
bool cx_fast_classify_supported(const cx_params& P);
void cx_launch_classify_fast(const cx_params& P, hipStream_t s);
void cx_launch_emit_triangles(const cx_params& P, const uint64_t* hash_xy, hipStream_t s);
void cx_launch_hash_xy(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t org0, uint32_t org1, hipStream_t s);
