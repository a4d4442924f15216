/*
 * oracle/march4d_oracle.c  --  TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C, single-threaded CPU restatement of contourist's 4-D marching-pentatopes hyper-voxel
 * march on a dense fp32 sample array A[n0][n1][n2][n3] ("Level 0 / 4-D": the state of the
 * reference after enumerate_voxel_tetrahedra() and before bin_times()).  Paths below are relative
 * to /root/reference/contourist/.
 *
 *   PENTATOPES / HYPERCUBE                      pentatopes.py:15-30
 *   GridContour4D.enumerate_pentatope_tetrahedra pentatopes.py:223-291
 *   GridContour.border_voxel (16 corners)        tetrahedral.py:383-394 with box = HYPERCUBE
 *   GridContour.contour_pair_interpolation       tetrahedral.py:471-512
 *
 * Parity pin: tests/test_oracle4d_vs_golden.py against fixtures produced by the real reference
 * (oracle/make_goldens4d.py; the reference's Python-2 modules are translated with lib2to3 into a
 * temp dir outside the repo).  The reference has no test of its own for this path.
 * Dense scan instead of the reference's seeded BFS, as in march_oracle.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int allclose1(double a, double b) { return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b); }

static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static uint64_t py_tuplehash(const int64_t* t, int n) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    for (int i = 0; i < n; i++) {
        int64_t x = t[i];
        uint64_t lane = (x == -1) ? (uint64_t)(int64_t)-2 : (uint64_t)x;
        acc += lane * P2;
        acc = rotl64(acc, 31);
        acc *= P1;
    }
    acc += (uint64_t)n ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) acc = 1546275796ULL;
    return acc;
}

static void py_set8_slots(const uint64_t* h, int m, int* slot_of) {
    int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int n = 0; n < m; n++) {
        uint64_t perturb = h[n];
        unsigned i = (unsigned)(h[n] & 7);
        while (used[i]) {
            perturb >>= 5;
            i = (unsigned)((i * 5 + 1 + perturb) & 7);
        }
        used[i] = 1;
        slot_of[n] = (int)i;
    }
}

/* order[] = iteration order of a set built by inserting pts[0..m-1] (CPython 3.10, <= 5 items) */
static void set_iter_order(const int64_t pts[][4], int m, int* order, int emulate, const int64_t* origin) {
    for (int i = 0; i < m; i++) order[i] = i;
    if (!emulate) return;
    uint64_t h[5];
    int slot[5];
    for (int i = 0; i < m; i++) {
        int64_t g[4];
        for (int d = 0; d < 4; d++) g[d] = pts[i][d] + origin[d];
        h[i] = py_tuplehash(g, 4);
    }
    py_set8_slots(h, m, slot);
    for (int i = 1; i < m; i++) {
        int o = order[i], j = i - 1;
        while (j >= 0 && slot[order[j]] > slot[o]) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = o;
    }
}

typedef struct {
    int64_t* keys;
    int64_t* vals;
    int64_t cap;
} emap_t;

static int emap_init(emap_t* m, int64_t want) {
    int64_t cap = 1024;
    while (cap < 2 * want) cap <<= 1;
    m->cap = cap;
    m->keys = (int64_t*)malloc(sizeof(int64_t) * cap);
    m->vals = (int64_t*)malloc(sizeof(int64_t) * cap);
    if (!m->keys || !m->vals) return -1;
    for (int64_t i = 0; i < cap; i++) m->keys[i] = -1;
    return 0;
}
static int64_t emap_find(emap_t* m, int64_t key, int* found) {
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
    int64_t i = (int64_t)(h >> 20) & (m->cap - 1);
    while (m->keys[i] != -1 && m->keys[i] != key) i = (i + 1) & (m->cap - 1);
    *found = (m->keys[i] == key);
    return i;
}

typedef struct {
    const float* A;
    int64_t n[4];
    double value;
    int emulate;
    int64_t origin[4];
    emap_t map;
    int32_t* vert_pairs; /* [vcap][8] low point, high point */
    double* vert_xyzt;   /* [vcap][4] */
    int64_t vcap, nverts;
    int64_t* tets;       /* [tcap][4] vertex indices in add_simplex pair order */
    int64_t tcap, ntets;
    int64_t nborder, nborder_mixed;
} march4_t;

static double sample4(const march4_t* M, const int64_t* p) {
    return (double)M->A[((p[0] * M->n[1] + p[1]) * M->n[2] + p[2]) * M->n[3] + p[3]];
}

/* contour_pair_interpolation(swap=True), linear (tetrahedral.py:471-512) */
static int64_t interpolate_pair4(march4_t* M, const int64_t* p0, const int64_t* p1) {
    const int64_t* lowp = p0;
    const int64_t* highp = p1;
    double flow = sample4(M, p0), fhigh = sample4(M, p1);
    if (flow > fhigh) {
        const int64_t* tp = lowp; lowp = highp; highp = tp;
        double tf = flow; flow = fhigh; fhigh = tf;
    }
    const int64_t* a = p0;
    const int64_t* b = p1;
    for (int d = 0; d < 4; d++) {
        if (p0[d] != p1[d]) {
            if (p0[d] > p1[d]) { a = p1; b = p0; }
            break;
        }
    }
    int64_t lin = ((a[0] * M->n[1] + a[1]) * M->n[2] + a[2]) * M->n[3] + a[3];
    int64_t dir = ((b[0] - a[0]) << 3) | ((b[1] - a[1]) << 2) | ((b[2] - a[2]) << 1) | (b[3] - a[3]);
    int64_t key = lin * 16 + dir;
    int found;
    int64_t slot = emap_find(&M->map, key, &found);
    if (found) return M->map.vals[slot];
    int64_t idx = M->nverts++;
    if (idx < M->vcap) {
        double z = M->value, x[4];
        for (int d = 0; d < 4; d++) x[d] = (double)lowp[d];
        if (flow <= z && fhigh >= z) {
            double ratio = 0.5;
            double denominator = 1.0 * (fhigh - flow);
            if (!allclose1(denominator, 0.0)) ratio = (z - flow) / denominator;
            for (int d = 0; d < 4; d++) x[d] = (double)lowp[d] + ratio * ((double)highp[d] - (double)lowp[d]);
        }
        for (int d = 0; d < 4; d++) {
            M->vert_pairs[idx * 8 + d] = (int32_t)lowp[d];
            M->vert_pairs[idx * 8 + 4 + d] = (int32_t)highp[d];
            M->vert_xyzt[idx * 4 + d] = x[d];
        }
        M->map.keys[slot] = key;
        M->map.vals[slot] = idx;
    }
    return idx;
}

/* add_simplex for dimension 4: a tetrahedron of 4 interpolated pairs (tetrahedral.py:176-182) */
static void add_simplex4(march4_t* M, const int64_t* pr[8]) {
    int64_t v[4];
    for (int s = 0; s < 4; s++) v[s] = interpolate_pair4(M, pr[2 * s], pr[2 * s + 1]);
    int64_t t = M->ntets++;
    if (t < M->tcap)
        for (int s = 0; s < 4; s++) M->tets[t * 4 + s] = v[s];
}

/* GridContour4D.enumerate_pentatope_tetrahedra (pentatopes.py:223-291) */
static void enumerate_pentatope(march4_t* M, const int64_t pent[5][4]) {
    int64_t low[5][4], high[5][4];
    int nlow = 0, nhigh = 0, all_close = 1;
    for (int m = 0; m < 5; m++) {
        double pvalue = sample4(M, pent[m]);
        if (pvalue < M->value) memcpy(low[nlow++], pent[m], 4 * sizeof(int64_t));
        else memcpy(high[nhigh++], pent[m], 4 * sizeof(int64_t));
        if (!allclose1(pvalue, M->value)) all_close = 0;
    }
    if (nlow == 0 || nhigh == 0 || all_close) return;
    int64_t(*least)[4] = low;
    int64_t(*most)[4] = high;
    int nleast = nlow, nmost = nhigh;
    if (nleast > nmost) {
        least = high; most = low;
        nleast = nhigh; nmost = nlow;
    }
    int ol[5], om[5];
    set_iter_order(least, nleast, ol, M->emulate, M->origin);
    set_iter_order(most, nmost, om, M->emulate, M->origin);
    if (nleast == 1) { /* :246-250 */
        const int64_t* a = least[0];
        const int64_t* pr[8] = {a, most[om[0]], a, most[om[1]], a, most[om[2]], a, most[om[3]]};
        add_simplex4(M, pr);
    } else { /* :255-291 */
        const int64_t* a = least[ol[0]];
        const int64_t* b = least[ol[1]];
        const int64_t* c = most[om[0]];
        const int64_t* d = most[om[1]];
        const int64_t* e = most[om[2]];
        const int64_t* t1[8] = {a, c, b, e, a, d, b, d}; /* (ac, be, ad, bd) */
        const int64_t* t2[8] = {a, c, b, e, a, d, a, e}; /* (ac, be, ad, ae) */
        const int64_t* t3[8] = {a, c, b, e, b, d, b, c}; /* (ac, be, bd, bc) */
        add_simplex4(M, t1);
        add_simplex4(M, t2);
        add_simplex4(M, t3);
    }
}

/* pentatope n = monotone lattice path for the n-th permutation of itertools.permutations(range(4)),
 * permutation entry "axis" raises coordinate axis "axis" (pentatopes.py:15-26) */
static void pentatope_paths(int paths[24][5][4]) {
    int n = 0;
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < 4; c++)
                for (int d = 0; d < 4; d++) {
                    if (a == b || a == c || a == d || b == c || b == d || c == d) continue;
                    int perm[4] = {a, b, c, d};
                    int v[4] = {0, 0, 0, 0};
                    memcpy(paths[n][0], v, sizeof(v));
                    for (int s = 0; s < 4; s++) {
                        v[perm[s]] = 1;
                        memcpy(paths[n][s + 1], v, sizeof(v));
                    }
                    n++;
                }
}

int oracle_march4d(const float* A, const int64_t* shape, double value, int diag_mode, const int64_t* origin,
                   int32_t* vert_pairs, double* vert_xyzt, int64_t vcap, int64_t* tets, int64_t tcap, int64_t* counts) {
    march4_t M;
    memset(&M, 0, sizeof(M));
    M.A = A;
    for (int d = 0; d < 4; d++) { M.n[d] = shape[d]; M.origin[d] = origin ? origin[d] : 0; }
    M.value = value;
    M.emulate = diag_mode;
    M.vert_pairs = vert_pairs; M.vert_xyzt = vert_xyzt; M.vcap = vcap;
    M.tets = tets; M.tcap = tcap;
    if (emap_init(&M.map, vcap) != 0) return -1;
    int paths[24][5][4];
    pentatope_paths(paths);
    for (int64_t i = 0; i + 1 < M.n[0]; i++)
        for (int64_t j = 0; j + 1 < M.n[1]; j++)
            for (int64_t k = 0; k + 1 < M.n[2]; k++)
                for (int64_t l = 0; l + 1 < M.n[3]; l++) {
                    const int64_t p[4] = {i, j, k, l};
                    /* border_voxel with the 16 corners of HYPERCUBE */
                    double fmin = 0, fmax = 0;
                    int all_close = 1, nlow = 0;
                    for (int c = 0; c < 16; c++) {
                        int64_t q[4] = {i + ((c >> 3) & 1), j + ((c >> 2) & 1), k + ((c >> 1) & 1), l + (c & 1)};
                        double f = sample4(&M, q);
                        if (c == 0 || f < fmin) fmin = f;
                        if (c == 0 || f > fmax) fmax = f;
                        if (!allclose1(value, f)) all_close = 0;
                        if (f < value) nlow++;
                    }
                    if (all_close || !(fmin <= value && fmax >= value)) continue;
                    M.nborder++;
                    if (nlow > 0 && nlow < 16) M.nborder_mixed++;
                    for (int n = 0; n < 24; n++) {
                        int64_t pent[5][4];
                        for (int m = 0; m < 5; m++)
                            for (int d = 0; d < 4; d++) pent[m][d] = p[d] + paths[n][m][d];
                        enumerate_pentatope(&M, pent);
                    }
                    if (M.nverts > M.vcap) M.nverts = M.vcap + 1;
                }
    counts[0] = M.nverts;
    counts[1] = M.ntets;
    counts[2] = M.nborder;
    counts[3] = M.nborder_mixed;
    free(M.map.keys);
    free(M.map.vals);
    return 0;
}
