// cx_state4.h -- parameters and per-context state of the 4-D march (host side).
#pragma once
#include "cx_ctx.h"

struct cx_params4 {
    const float* grid;
    uint32_t n0, n1, n2, n3, nsamples;
    cx_fdiv div3, div2, div1;
    float vcmp, near_abs, vhi, vlo;
    double value, tol_value;
    uint32_t flags;
    uint32_t org[4];
    // where the vertices of a lattice cell are, without a table of one entry per sample (rounds 1-3: 8 bytes per sample, 1 GiB on config 4):
    // the cells of one bitmap word (32 cells of a row) sit next to each other in the queue, so
    //     queue position of cell (row, l) = items[row * nw3 + l / 32].x + popc(items[...].y & bits below l % 32)
    // and the cells kernel leaves (crossing mask << 32 | first vertex) per QUEUE ENTRY, densely
    uint2* items;              // [nrows * nw3] {queue position of the word's first active cell, the word's active cells}; written for words with one
    uint64_t* info;            // [qcap] per queue entry
    float4* verts;
    uint32_t* vkeys;
    uint4* cells;
    int32_t* tets;
    uint32_t vcap, ccap, tcap;
    uint32_t* counters;
    unsigned long long* counters_tb;   // (border voxels << 32) | tetrahedra: reserved by the cells kernel, in a cache line of its own
    const uint64_t* hash_xyz;
    const uint64_t* lut;
    uint32_t* queue;           // linear indices of the cells with a sign change among their corners
    uint32_t qcap;
    uint4* rounds;             // per 64 queue entries: first record, records, first tetrahedron, tetrahedra
    // sign bitmap: bit l%32 of word [row(i,j,k)][l/32] <=> sample < isovalue
    uint32_t* signbits;
    uint32_t nw3;              // words per row
    uint32_t nrows;            // n0*n1*n2
    cx_fdiv div_w, div_r2, div_r1;   // / nw3, / (n1*n2), / n2
};
#define CX4_CNT_QUEUE 6   // counter word: cells in the queue
void cx_launch_signbits4d(const cx_params4& P, hipStream_t s);
void cx_launch_classify4d(const cx_params4& P, hipStream_t s);
void cx_launch_emit_tets(const cx_params4& P, hipStream_t s);
void cx_launch_hash_xyz(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t n2, const uint32_t org[4], hipStream_t s);
const uint64_t* cx_pent_lut_device();

struct cx_state4 {
    const float* grid = nullptr;
    float* grid_owned = nullptr;
    size_t grid_owned_bytes = 0;
    int64_t n[4] = {0, 0, 0, 0};
    uint2* items = nullptr;
    size_t items_cap = 0;
    uint64_t* info = nullptr;
    size_t info_cap = 0;
    float4* verts = nullptr;
    uint32_t* vkeys = nullptr;
    size_t vkeys_cap = 0;
    uint4* cells = nullptr;
    int32_t* tets = nullptr;
    uint32_t vcap = 0, ccap = 0, tcap = 0;
    uint64_t* hash_xyz = nullptr;
    size_t hash_cap = 0;
    uint32_t* queue = nullptr;
    uint4* rounds = nullptr;
    size_t rounds_cap = 0;
    uint32_t qcap = 0;
    uint32_t* signbits = nullptr;
    size_t signbits_cap = 0;
    int64_t hash_key[7] = {-1, -1, -1, -1, -1, -1, -1};
    int64_t origin[4] = {0, 0, 0, 0};
    bool extracted = false;
    bool post_valid = false;
    bool pending = false;            // a cx_extract4d_async is enqueued, its counters not yet looked at (cx_counts4d_get)
    double pending_value = 0.0;
    uint32_t pending_flags = 0;
    double value = 0.0;
    cx_counts counts = {0, 0, 0, 0};
    // seeded selection (cx_select_seeded4d): mask over the Level-0 tetrahedra, valid until the next extraction
    uint8_t* tet_keep = nullptr;
    size_t keep_cap = 0;
    bool keep_valid = false;
};

