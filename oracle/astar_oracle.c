/*
 * astar_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see nmpc_oracle.h).
 *
 * CPU restatement of the reference's kinodynamic A* front end (second half), statement by statement:
 *   KinodynamicAstar::search            src/resilient_planner/path_searching/src/kinodynamic_astar.cpp:17-287
 *   estimateHeuristic / cubic / quartic :322-357, :426-501
 *   computeShotTraj                     :359-424
 *   retrievePath / getKinoTraj          :308-320, :648-695
 *   posToIndex / stateTransit           :815-820, :828-845   (f_ext enters the primitives at :838)
 *   the caller's search / retry rule    plan_manage/src/nmpc_solver.cpp:154-207 (NMPCSolver::getKinoPath)
 * and of what it asks the occupancy map:
 *   OccMap::checkState / getlineGrids   src/resilient_planner/occ_grid/src/occ_map.cpp:645-718
 *   OccMap::getVoxelState / isInMap / isInLocalMap / posToIndex   :45-116
 *   RayCaster::setInput / step          src/resilient_planner/occ_grid/src/raycast.cpp:263-365 (signum / mod / intbound :6-29)
 * The open set is a std::priority_queue of node pointers ordered by f_score (kinodynamic_astar.h:52-57) whose keys the search
 * modifies IN PLACE without re-heapifying (:220-226, :263-272); the order in which nodes are popped therefore depends on the
 * binary-heap algorithms of the C++ library, which are restated here (libstdc++ bits/stl_heap.h: __push_heap, __adjust_heap,
 * __pop_heap).  The closed / expanded set is an exact map from the voxel index to the node (kinodynamic_astar.h:72-103).
 *
 * PARITY UNPINNED: path_searching needs ROS, Eigen and boost, none of which is in the image, and the reference holds no test
 * vectors for it.  Deviations, all stated here:
 *  - cbrt / acos / cos (cubic(), :439-456) are the fdlibm algorithms written out below (det_cbrt, det_acos, det_cos; each within
 *    1 ulp of glibc's, which the reference links -- tests/test_oracle_astar.py checks that) and pow(t, j), j <= 3, is 1, t, t t,
 *    t t t: the HIP kernel uses the same sequences of IEEE operations, so that the two agree to the bit and the order in which
 *    nodes leave the heap cannot depend on the last bit of a libm call;
 *  - sums of three products (dot products, squared norms) are evaluated left to right without contraction (this file is
 *    compiled with -ffp-contract=off; Eigen's own evaluation order for 3-vectors is the same up to its unrolling);
 *  - `dynamic` is always false (the reference never passes true) and the uninitialised PathNode::time / time_origin_ are not
 *    modelled: they only feed timeToIndex(), whose result is unused when !dynamic;
 *  - RayCaster::step has no exit when a ray misses its end cell; a 4096-step cap returns "occupied" instead of hanging.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "astar_oracle.h"

/* ------------------------------------------------------------------ deterministic elementary functions (fdlibm) */
/* Eigen's unrolled reduction of a fixed-size 3-vector: a0 b0 + (a1 b1 + a2 b2) */
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

static inline uint32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double from_words(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }

double orc_det_cbrt(double x) /* fdlibm s_cbrt.c */
{
    const uint32_t B1 = 715094163u, B2 = 696219795u;
    const double C = 5.42857142857142815906e-01, D = -7.05306122448979611050e-01, E = 1.41428571428571436819e+00,
                 F = 1.60714285714285720630e+00, G = 3.57142857142857150787e-01;
    uint32_t hx = hi_word(x);
    const uint32_t sign = hx & 0x80000000u;
    hx ^= sign;
    if (hx >= 0x7ff00000u) return x + x;
    if ((hx | lo_word(x)) == 0) return x;
    x = from_words(hx, lo_word(x)); /* |x| */
    double t;
    if (hx < 0x00100000u) {
        t = from_words(0x43500000u, 0);
        t *= x;
        t = from_words(hi_word(t) / 3 + B2, 0);
    } else
        t = from_words(hx / 3 + B1, 0);
    double r = t * t / x;
    double s = C + r * t;
    t *= G + F / (s + E + D / s);
    t = from_words(hi_word(t) + 1u, 0);
    s = t * t;
    r = x / s;
    const double w = t + t;
    r = (r - t) / (w + r);
    t = t + t * r;
    return from_words(hi_word(t) | sign, lo_word(t));
}

double orc_det_acos(double x) /* fdlibm e_acos.c */
{
    const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const uint32_t hx = hi_word(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | lo_word(x)) == 0) return (hx >> 31) ? pi + 2.0 * pio2_lo : 0.0;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000u) {
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx >> 31) {
        const double z = (one + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {
        const double z = (one - x) * 0.5;
        const double s = sqrt(z);
        const double df = from_words(hi_word(s), 0);
        const double c = (z - df * df) / (s + df);
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}

/* cos for |x| < ~1e5 (cubic() evaluates it on [0, 5 pi / 3]): two-constant Cody-Waite reduction by pi/2, the products formed
 * exactly inside fma(), then the fdlibm kernels k_sin.c / k_cos.c on [-pi/4, pi/4] */
double orc_det_cos(double x)
{
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double b = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -b : b;
}

/* ------------------------------------------------------------------ occupancy map queries (occ_map.cpp) */
typedef struct {
    const orc_astar_params *P;
    double resolution_inv;
} Map;

static void map_pos_to_index(const Map *m, const double pos[3], int id[3]) /* occ_map.cpp:71-75 */
{
    for (int i = 0; i < 3; i++) id[i] = (int)floor((pos[i] - m->P->origin[i]) * m->resolution_inv);
}
static int map_voxel_state(const Map *m, const double pos[3]) /* occ_map.cpp:95-106 */
{
    const orc_astar_params *P = m->P;
    int id[3];
    map_pos_to_index(m, pos, id);
    if (!((id[0] | (P->grid[0] - 1 - id[0]) | id[1] | (P->grid[1] - 1 - id[1]) | id[2] | (P->grid[2] - 1 - id[2])) >= 0)) return -1; /* :66-69 */
    if (P->use_local) { /* isInLocalMap, :45-57: min_id / max_id already clamped by the caller's generator the way :48-55 do */
        if (!(((id[0] - P->local_min[0]) | (P->local_max[0] - id[0]) | (id[1] - P->local_min[1]) | (P->local_max[1] - id[1]) |
               (id[2] - P->local_min[2]) | (P->local_max[2] - id[2])) >= 0))
            return 0;
    }
    return P->occ[(size_t)id[0] * P->grid[1] * P->grid[2] + (size_t)id[1] * P->grid[2] + id[2]] ? 1 : 0;
}

static int signum_i(int x) { return x == 0 ? 0 : (x < 0 ? -1 : 1); } /* raycast.cpp:6-9 */
static double mod_d(double value, double modulus) { return fmod(fmod(value, modulus) + modulus, modulus); } /* :11-14 */
static double intbound(double s, double ds) /* :16-29 */
{
    if (ds < 0) return intbound(-s, -ds);
    s = mod_d(s, 1);
    return (1 - s) / ds;
}

/* getlineGrids (occ_map.cpp:686-718) fused with the loop over its result in checkState (:675-683): returns 1 as soon as a cell
 * of the segment is not free.  The cells come in the reference's order (the cells the ray visits, its end cell last). */
static int line_hits(const Map *m, const double s_p[3], const double e_p[3])
{
    const double res = m->P->resolution;
    const double start[3] = {s_p[0] / res, s_p[1] / res, s_p[2] / res}, end[3] = {e_p[0] / res, e_p[1] / res, e_p[2] / res};
    /* RayCaster::setInput, raycast.cpp:263-311 */
    int x = (int)floor(start[0]), y = (int)floor(start[1]), z = (int)floor(start[2]);
    const int endX = (int)floor(end[0]), endY = (int)floor(end[1]), endZ = (int)floor(end[2]);
    const double dx = endX - x, dy = endY - y, dz = endZ - z;
    const int stepX = signum_i((int)dx), stepY = signum_i((int)dy), stepZ = signum_i((int)dz);
    double tMaxX = intbound(start[0], dx), tMaxY = intbound(start[1], dy), tMaxZ = intbound(start[2], dz);
    const double tDeltaX = ((double)stepX) / dx, tDeltaY = ((double)stepY) / dy, tDeltaZ = ((double)stepZ) / dz;
    const int need_ray = !(stepX == 0 && stepY == 0 && stepZ == 0);
    double tmp[3];
    if (need_ray) {
        for (int guard = 0;; guard++) { /* while (raycaster.step(ray_pt)), :313-365: the cell is reported, then the ray advances */
            if (x == endX && y == endY && z == endZ) break;
            if (guard >= 4096) return 1;
            tmp[0] = (double)x * res + res / 2.0; tmp[1] = (double)y * res + res / 2.0; tmp[2] = (double)z * res + res / 2.0; /* :697-701 */
            if (map_voxel_state(m, tmp) != 0) return 1;
            if (tMaxX < tMaxY) {
                if (tMaxX < tMaxZ) { x += stepX; tMaxX += tDeltaX; }
                else { z += stepZ; tMaxZ += tDeltaZ; }
            } else {
                if (tMaxY < tMaxZ) { y += stepY; tMaxY += tDeltaY; }
                else { z += stepZ; tMaxZ += tDeltaZ; }
            }
        }
    }
    /* "check end", :704-717 */
    tmp[0] = floor(end[0]) * res + res / 2.0; tmp[1] = floor(end[1]) * res + res / 2.0; tmp[2] = floor(end[2]) * res + res / 2.0;
    return map_voxel_state(m, tmp) != 0;
}

/* OccMap::checkState (occ_map.cpp:645-684): 1 = free */
static int check_state(const Map *m, const double pos[3], const double vel[3], double inflate_ratio)
{
    double vel_hor[2] = {vel[0], vel[1]};
    const double v_hor_norm = sqrt(vel_hor[0] * vel_hor[0] + vel_hor[1] * vel_hor[1]);
    if (v_hor_norm < 1e-4) { vel_hor[0] = 1; vel_hor[1] = 1; }
    double cw[2] = {vel_hor[1], -vel_hor[0]}; /* r_m * vel_hor, r_m = [0 1; -1 0] */
    const double n2 = cw[0] * cw[0] + cw[1] * cw[1];
    if (n2 > 0.0) { const double nn = sqrt(n2); cw[0] = cw[0] / nn; cw[1] = cw[1] / nn; } /* Eigen normalized() */
    cw[0] = cw[0] * m->P->ego_r * inflate_ratio; cw[1] = cw[1] * m->P->ego_r * inflate_ratio;
    const double cw_edge[3] = {pos[0] + cw[0], pos[1] + cw[1], pos[2]}, ccw_edge[3] = {pos[0] - cw[0], pos[1] - cw[1], pos[2]};
    if (line_hits(m, cw_edge, ccw_edge)) return 0;
    const double up[3] = {pos[0], pos[1], pos[2] + m->P->ego_h * inflate_ratio}, down[3] = {pos[0], pos[1], pos[2] - m->P->ego_h * inflate_ratio};
    if (line_hits(m, up, down)) return 0;
    return 1;
}

/* ------------------------------------------------------------------ the search (kinodynamic_astar.cpp) */
#define IN_CLOSE_SET 'a'
#define IN_OPEN_SET 'b'
#define NOT_EXPAND 'c'

typedef struct {
    int index[3];
    double state[6], g_score, f_score, input[3], duration;
    int parent; /* node id, -1 = NULL */
    char node_state;
} PathNode;

typedef struct {
    const orc_astar_params *P;
    Map map;
    double inv_resolution, external_acc[3], end_pt[3];
    PathNode *pool;
    int use_node_num, iter_num;
    int *heap, heap_size;           /* std::priority_queue<PathNodePtr, vector, NodeComparator> */
    int64_t *hkeys; int *hvals; int hcap; /* NodeHashTable::data_3d_ as an open-addressing table (exact map semantics) */
    int is_shot_succ; double coef_shot[12], t_shot;
} Search;

static int64_t pack_index(const int id[3]) { return (((int64_t)(id[0] + (1 << 20))) << 42) | (((int64_t)(id[1] + (1 << 20))) << 21) | (int64_t)(id[2] + (1 << 20)); }
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static int hash_find(const Search *S, const int id[3])
{
    const int64_t key = pack_index(id);
    for (uint64_t h = mix64((uint64_t)key) & (uint64_t)(S->hcap - 1);; h = (h + 1) & (uint64_t)(S->hcap - 1)) {
        if (S->hvals[h] < 0) return -1;
        if (S->hkeys[h] == key) return S->hvals[h];
    }
}
static void hash_insert(Search *S, const int id[3], int node) /* unordered_map::insert: keeps an existing entry */
{
    const int64_t key = pack_index(id);
    for (uint64_t h = mix64((uint64_t)key) & (uint64_t)(S->hcap - 1);; h = (h + 1) & (uint64_t)(S->hcap - 1)) {
        if (S->hvals[h] < 0) { S->hkeys[h] = key; S->hvals[h] = node; return; }
        if (S->hkeys[h] == key) return;
    }
}

/* NodeComparator (kinodynamic_astar.h:52-57) */
static int node_comp(const Search *S, int n1, int n2) { return S->pool[n1].f_score > S->pool[n2].f_score; }
static void heap_push_hole(Search *S, int holeIndex, int topIndex, int value) /* std::__push_heap */
{
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && node_comp(S, S->heap[parent], value)) {
        S->heap[holeIndex] = S->heap[parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    S->heap[holeIndex] = value;
}
static void heap_push(Search *S, int node) { S->heap[S->heap_size++] = node; heap_push_hole(S, S->heap_size - 1, 0, node); }
static void heap_pop(Search *S) /* std::pop_heap + pop_back */
{
    if (S->heap_size > 1) {
        const int last = S->heap_size - 1, value = S->heap[last];
        S->heap[last] = S->heap[0];
        const int len = last; /* __adjust_heap(first, 0, len, value) */
        int holeIndex = 0, secondChild = 0;
        while (secondChild < (len - 1) / 2) {
            secondChild = 2 * (secondChild + 1);
            if (node_comp(S, S->heap[secondChild], S->heap[secondChild - 1])) secondChild--;
            S->heap[holeIndex] = S->heap[secondChild];
            holeIndex = secondChild;
        }
        if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
            secondChild = 2 * (secondChild + 1);
            S->heap[holeIndex] = S->heap[secondChild - 1];
            holeIndex = secondChild - 1;
        }
        heap_push_hole(S, holeIndex, 0, value);
    }
    S->heap_size--;
}

static void pos_to_index(const Search *S, const double pt[3], int idx[3]) /* :815-820 */
{
    for (int i = 0; i < 3; i++) idx[i] = (int)floor((pt[i] - S->P->origin[i]) * S->inv_resolution);
}

static void state_transit(const Search *S, const double state0[6], double state1[6], const double um[3], double tau) /* :828-845 */
{
    double um_with_disturb[3];
    for (int i = 0; i < 3; i++) um_with_disturb[i] = um[i] + S->external_acc[i];
    const double t2 = tau * tau; /* pow(tau, 2) */
    for (int i = 0; i < 3; i++) {
        /* phi_ * state0 + integral: the zero entries of phi_ contribute exact zeros */
        state1[i] = (state0[i] + tau * state0[3 + i]) + 0.5 * t2 * um_with_disturb[i];
        state1[3 + i] = state0[3 + i] + tau * um_with_disturb[i];
    }
}

static int cubic(double a, double b, double c, double d, double dts[3]) /* :426-459 */
{
    const double a2 = b / a, a1 = c / a, a0 = d / a;
    const double Q = (3 * a1 - a2 * a2) / 9;
    const double R = (9 * a1 * a2 - 27 * a0 - 2 * a2 * a2 * a2) / 54;
    const double D = Q * Q * Q + R * R;
    if (D > 0) {
        const double S = orc_det_cbrt(R + sqrt(D)), T = orc_det_cbrt(R - sqrt(D));
        dts[0] = -a2 / 3 + (S + T);
        return 1;
    } else if (D == 0) {
        const double S = orc_det_cbrt(R);
        dts[0] = -a2 / 3 + S + S;
        dts[1] = -a2 / 3 - S;
        return 2;
    } else {
        const double theta = orc_det_acos(R / sqrt(-Q * Q * Q));
        dts[0] = 2 * sqrt(-Q) * orc_det_cos(theta / 3) - a2 / 3;
        dts[1] = 2 * sqrt(-Q) * orc_det_cos((theta + 2 * M_PI) / 3) - a2 / 3;
        dts[2] = 2 * sqrt(-Q) * orc_det_cos((theta + 4 * M_PI) / 3) - a2 / 3;
        return 3;
    }
}

static int quartic(double a, double b, double c, double d, double e, double dts[4]) /* :461-501 */
{
    const double a3 = b / a, a2 = c / a, a1 = d / a, a0 = e / a;
    double ys[3];
    cubic(1, -a2, a1 * a3 - 4 * a0, 4 * a2 * a0 - a1 * a1 - a3 * a3 * a0, ys);
    const double y1 = ys[0];
    const double r = a3 * a3 / 4 - a2 + y1;
    if (r < 0) return 0;
    const double R = sqrt(r);
    double D, E;
    if (R != 0) {
        D = sqrt(0.75 * a3 * a3 - R * R - 2 * a2 + 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
        E = sqrt(0.75 * a3 * a3 - R * R - 2 * a2 - 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
    } else {
        D = sqrt(0.75 * a3 * a3 - 2 * a2 + 2 * sqrt(y1 * y1 - 4 * a0));
        E = sqrt(0.75 * a3 * a3 - 2 * a2 - 2 * sqrt(y1 * y1 - 4 * a0));
    }
    int n = 0;
    if (!isnan(D)) { dts[n++] = -a3 / 4 + R / 2 + D / 2; dts[n++] = -a3 / 4 + R / 2 - D / 2; }
    if (!isnan(E)) { dts[n++] = -a3 / 4 - R / 2 + E / 2; dts[n++] = -a3 / 4 - R / 2 - E / 2; }
    return n;
}

static double estimate_heuristic(const Search *S, const double x1[6], const double x2[6], double *optimal_time) /* :322-357 */
{
    const orc_astar_params *P = S->P;
    double dp[3], v0[3], v1[3];
    for (int i = 0; i < 3; i++) { dp[i] = x2[i] - x1[i]; v0[i] = x1[3 + i]; v1[i] = x2[3 + i]; }
    /* Eigen reduces a fixed-size 3-vector as a0 b0 + (a1 b1 + a2 b2) (redux_novec_unroller splits [0, 3) into [0, 1) and [1, 3)):
     * dp.dot(dp), (v0 + v1).dot(dp), v0.dot(v0) + v0.dot(v1) + v1.dot(v1) of kinodynamic_astar.cpp:329-331 in that order */
    const double vs[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    const double c1 = -36 * dot3(dp, dp);
    const double c2 = 24 * dot3(vs, dp);
    const double c3 = -4 * ((dot3(v0, v0) + dot3(v0, v1)) + dot3(v1, v1));
    const double c4 = 0, c5 = P->w_time;
    double ts[5];
    int nt = quartic(c5, c4, c3, c2, c1, ts);
    const double v_max = P->max_vel;
    const double t_bar = fmax(fmax(fabs(x1[0] - x2[0]), fabs(x1[1] - x2[1])), fabs(x1[2] - x2[2])) / v_max;
    ts[nt++] = t_bar;
    double cost = 100000000, t_d = t_bar;
    for (int i = 0; i < nt; i++) {
        const double t = ts[i];
        if (t < t_bar) continue;
        const double c = -c1 / (3 * t * t * t) - c2 / (2 * t * t) - c3 / t + P->w_time * t;
        if (c < cost) { cost = c; t_d = t; }
    }
    *optimal_time = t_d;
    return 1.0 * (1 + P->tie_breaker) * cost;
}

static int in_range_open(const orc_astar_params *P, const double p[3]) /* :152-154 (strict on both sides) */
{
    return !(p[0] <= P->origin[0] || p[0] >= P->map_size[0] * 0.5 || p[1] <= P->origin[1] || p[1] >= P->map_size[1] * 0.5 ||
             p[2] <= 0.1 || p[2] >= P->map_size[2] * 0.5);
}

static int compute_shot_traj(Search *S, const double state1[6], const double state2[6], double time_to_goal) /* :359-424 */
{
    const orc_astar_params *P = S->P;
    const double t_d = time_to_goal;
    double coef[12]; /* coef(dim, j), j = power of t */
    for (int dim = 0; dim < 3; dim++) {
        const double p0 = state1[dim], dp = state2[dim] - p0, v0 = state1[3 + dim], v1 = state2[3 + dim], dv = v1 - v0;
        const double a = 1.0 / 6.0 * (-12.0 / (t_d * t_d * t_d) * (dp - v0 * t_d) + 6 / (t_d * t_d) * dv);
        const double b = 0.5 * (6.0 / (t_d * t_d) * (dp - v0 * t_d) - 2 / t_d * dv);
        coef[dim * 4 + 3] = a; coef[dim * 4 + 2] = b; coef[dim * 4 + 1] = v0; coef[dim * 4 + 0] = p0;
    }
    const double t_delta = t_d / 10;
    for (double time = t_delta; time <= t_d; time += t_delta) {
        const double t[4] = {1.0, time, time * time, time * time * time};
        double coord[3], vel[3];
        for (int dim = 0; dim < 3; dim++) {
            const double *c = coef + dim * 4;
            coord[dim] = ((c[0] * t[0] + c[1] * t[1]) + c[2] * t[2]) + c[3] * t[3];
            /* (Tm * poly1d).dot(t), Tm = the derivative matrix: entries (c1, 2 c2, 3 c3, 0) */
            vel[dim] = ((c[1] * t[0] + (2 * c[2]) * t[1]) + (3 * c[3]) * t[2]) + 0.0 * t[3];
        }
        if (coord[0] < P->origin[0] || coord[0] >= P->map_size[0] * 0.5 || coord[1] < P->origin[1] || coord[1] >= P->map_size[1] * 0.5 ||
            coord[2] < 0.1 || coord[2] >= P->map_size[2] * 0.5)
            return 0;
        if (!check_state(&S->map, coord, vel, 1.5)) return 0;
    }
    memcpy(S->coef_shot, coef, sizeof coef);
    S->t_shot = t_d;
    S->is_shot_succ = 1;
    return 1;
}

int orc_astar_search(const orc_astar_params *P, const double start_pt[3], const double start_v[3], const double start_a[3],
                     const double end_pt[3], const double end_v[3], int init, const double external_acc[3], orc_astar_result *out)
{
    Search Sv, *S = &Sv;
    memset(S, 0, sizeof *S);
    S->P = P; S->map.P = P; S->map.resolution_inv = 1.0 / P->resolution;
    S->inv_resolution = 1.0 / P->resolution;
    memcpy(S->external_acc, external_acc, sizeof S->external_acc);
    memcpy(S->end_pt, end_pt, sizeof S->end_pt);
    S->pool = (PathNode *)malloc((size_t)P->allocate_num * sizeof(PathNode));
    S->heap = (int *)malloc((size_t)P->allocate_num * sizeof(int));
    S->hcap = 1;
    while (S->hcap < 2 * P->allocate_num) S->hcap <<= 1;
    S->hkeys = (int64_t *)malloc((size_t)S->hcap * sizeof(int64_t));
    S->hvals = (int *)malloc((size_t)S->hcap * sizeof(int));
    for (int i = 0; i < S->hcap; i++) S->hvals[i] = -1;
    int status = ORC_ASTAR_NO_PATH, terminate_node = -1;

    /* ---------- initialize ---------- (:25-46) */
    int cur = 0;
    PathNode *cur_node = &S->pool[0];
    cur_node->parent = -1;
    for (int i = 0; i < 3; i++) { cur_node->state[i] = start_pt[i]; cur_node->state[3 + i] = start_v[i]; }
    pos_to_index(S, start_pt, cur_node->index);
    cur_node->g_score = 0.0;
    cur_node->input[0] = cur_node->input[1] = cur_node->input[2] = 0.0; cur_node->duration = 0.0; /* (uninitialised in the reference; never read: every walk stops at the node without a parent) */
    double end_state[6], time_to_goal;
    int end_index[3];
    for (int i = 0; i < 3; i++) { end_state[i] = end_pt[i]; end_state[3 + i] = end_v[i]; }
    pos_to_index(S, end_pt, end_index);
    cur_node->f_score = P->lambda_heu * estimate_heuristic(S, cur_node->state, end_state, &time_to_goal);
    cur_node->node_state = IN_OPEN_SET;
    heap_push(S, 0);
    S->use_node_num += 1;
    hash_insert(S, cur_node->index, 0);

    int init_search = init;
    const int tolerance = (int)ceil(1 / P->resolution);
    int *tmp_expand = (int *)malloc((size_t)P->max_expand * sizeof(int));

    /* ---------- search loop ---------- (:54-282) */
    while (S->heap_size > 0) {
        cur = S->heap[0];
        cur_node = &S->pool[cur];
        const int near_end = abs(cur_node->index[0] - end_index[0]) <= tolerance && abs(cur_node->index[1] - end_index[1]) <= tolerance &&
                             abs(cur_node->index[2] - end_index[2]) <= tolerance;
        double dst[3] = {cur_node->state[0] - start_pt[0], cur_node->state[1] - start_pt[1], cur_node->state[2] - start_pt[2]};
        const int reach_horizon = sqrt(dot3(dst, dst)) >= P->horizon; /* (cur_node->state.head(3) - start_pt).norm() */
        if (reach_horizon || near_end) {
            terminate_node = cur;
            if (near_end) {
                estimate_heuristic(S, cur_node->state, end_state, &time_to_goal);
                compute_shot_traj(S, cur_node->state, end_state, time_to_goal);
                if (cur_node->parent < 0 && !S->is_shot_succ) status = ORC_ASTAR_NO_PATH;
                else if (!S->is_shot_succ) status = ORC_ASTAR_REACH_END_BUT_SHOT_FAILS;
                else status = ORC_ASTAR_REACH_END;
            } else
                status = ORC_ASTAR_REACH_HORIZON;
            goto done;
        }
        /* ---------- pop node and add to close set ---------- */
        heap_pop(S);
        cur_node->node_state = IN_CLOSE_SET;
        S->iter_num += 1;

        /* ---------- init state propagation ---------- (:108-137) */
        const double res = 1 / 2.0, time_res = 1 / 1.0, time_res_init = 1 / 8.0;
        double inputs[ORC_ASTAR_MAX_INPUTS][3], durations[ORC_ASTAR_MAX_DURATIONS];
        int n_in = 0, n_dur = 0;
        if (init_search) {
            memcpy(inputs[n_in++], start_a, 3 * sizeof(double));
            for (double tau = time_res_init * P->init_max_tau; tau <= P->init_max_tau; tau += time_res_init * P->init_max_tau)
                if (n_dur < ORC_ASTAR_MAX_DURATIONS) durations[n_dur++] = tau;
        } else {
            for (double ax = -P->max_acc; ax <= P->max_acc + 1e-3; ax += P->max_acc * res)
                for (double ay = -P->max_acc; ay <= P->max_acc + 1e-3; ay += P->max_acc * res)
                    for (double az = -P->max_acc; az <= P->max_acc + 1e-3; az += P->max_acc * res)
                        if (n_in < ORC_ASTAR_MAX_INPUTS) { inputs[n_in][0] = ax; inputs[n_in][1] = ay; inputs[n_in][2] = az; n_in++; }
            for (double tau = time_res * P->max_tau; tau <= P->max_tau; tau += time_res * P->max_tau)
                if (n_dur < ORC_ASTAR_MAX_DURATIONS) durations[n_dur++] = tau;
        }
        const double *cur_state = cur_node->state;
        int n_tmp = 0;

        /* ---------- state propagation loop ---------- (:140-281) */
        for (int i = 0; i < n_in; i++)
            for (int j = 0; j < n_dur; j++) {
                init_search = 0;
                const double *um = inputs[i];
                const double tau = durations[j];
                double pro_state[6];
                state_transit(S, cur_state, pro_state, um, tau);
                if (!in_range_open(P, pro_state)) continue;
                int pro_id[3];
                pos_to_index(S, pro_state, pro_id);
                int pro_node = hash_find(S, pro_id);
                if (pro_node >= 0 && S->pool[pro_node].node_state == IN_CLOSE_SET) continue;
                if (fabs(pro_state[3]) > P->max_vel || fabs(pro_state[4]) > P->max_vel || fabs(pro_state[5]) > P->max_vel) continue;
                if (pro_id[0] == cur_node->index[0] && pro_id[1] == cur_node->index[1] && pro_id[2] == cur_node->index[2]) continue; /* :179-184, !dynamic */
                int is_occ = 0;
                for (int k = 1; k <= P->check_num; ++k) {
                    const double dt = tau * (double)k / (double)P->check_num;
                    double xt[6];
                    state_transit(S, cur_state, xt, um, dt);
                    if (!check_state(&S->map, xt, xt + 3, 1.5)) { is_occ = 1; break; }
                }
                if (is_occ) continue;
                double ttg;
                const double tmp_g_score = (dot3(um, um) + P->w_time) * tau + cur_node->g_score; /* um.squaredNorm() */
                const double tmp_f_score = tmp_g_score + P->lambda_heu * estimate_heuristic(S, pro_state, end_state, &ttg);
                int prune = 0;
                for (int q = 0; q < n_tmp; ++q) {
                    PathNode *expand_node = &S->pool[tmp_expand[q]];
                    if (pro_id[0] == expand_node->index[0] && pro_id[1] == expand_node->index[1] && pro_id[2] == expand_node->index[2]) {
                        prune = 1;
                        if (tmp_f_score < expand_node->f_score) {
                            expand_node->f_score = tmp_f_score;
                            expand_node->g_score = tmp_g_score;
                            memcpy(expand_node->state, pro_state, sizeof pro_state);
                            memcpy(expand_node->input, um, 3 * sizeof(double));
                            expand_node->duration = tau;
                        }
                        break;
                    }
                }
                if (!prune) {
                    if (pro_node < 0) {
                        pro_node = S->use_node_num;
                        PathNode *pn = &S->pool[pro_node];
                        memcpy(pn->index, pro_id, sizeof pro_id);
                        memcpy(pn->state, pro_state, sizeof pro_state);
                        pn->f_score = tmp_f_score;
                        pn->g_score = tmp_g_score;
                        memcpy(pn->input, um, 3 * sizeof(double));
                        pn->duration = tau;
                        pn->parent = cur;
                        pn->node_state = IN_OPEN_SET;
                        heap_push(S, pro_node);
                        hash_insert(S, pro_id, pro_node);
                        if (n_tmp < P->max_expand) tmp_expand[n_tmp++] = pro_node;
                        S->use_node_num += 1;
                        if (S->use_node_num == P->allocate_num) { status = ORC_ASTAR_NO_PATH; terminate_node = -1; goto done; } /* "run out of memory", :255-259 */
                    } else if (S->pool[pro_node].node_state == IN_OPEN_SET) {
                        PathNode *pn = &S->pool[pro_node];
                        if (tmp_g_score < pn->g_score) {
                            memcpy(pn->state, pro_state, sizeof pro_state);
                            pn->f_score = tmp_f_score;
                            pn->g_score = tmp_g_score;
                            memcpy(pn->input, um, 3 * sizeof(double));
                            pn->duration = tau;
                            pn->parent = cur;
                        }
                    }
                }
            }
    }
    status = ORC_ASTAR_NO_PATH; /* open set empty */
    terminate_node = -1;
done:
    out->status = status;
    out->use_node_num = S->use_node_num;
    out->iter_num = S->iter_num;
    out->is_shot_succ = S->is_shot_succ;
    memcpy(out->coef_shot, S->coef_shot, sizeof out->coef_shot);
    out->t_shot = S->t_shot;
    out->n_path = 0;
    if (terminate_node >= 0) { /* retrievePath, :308-320 */
        int n = 0;
        for (int c = terminate_node; c >= 0; c = S->pool[c].parent) n++;
        if (n > ORC_ASTAR_MAX_PATH) { /* more nodes than the result holds: not returned at all (NO_PATH: the planner keeps its path), like the device */
            n = 0;
            status = ORC_ASTAR_NO_PATH;
            out->status = status;
        }
        out->n_path = n;
        int c = terminate_node;
        for (int q = n - 1; q >= 0; q--, c = S->pool[c].parent) {
            memcpy(out->path_state[q], S->pool[c].state, 6 * sizeof(double));
            memcpy(out->path_input[q], S->pool[c].input, 3 * sizeof(double));
            out->path_duration[q] = S->pool[c].duration;
            out->path_node[q] = c;
        }
    }
    free(tmp_expand); free(S->pool); free(S->heap); free(S->hkeys); free(S->hvals);
    return status;
}

/* getKinoTraj (:648-695) on a result: the search part backwards from the last node, reversed, then the shot */
int orc_astar_kino_traj(const orc_astar_params *P, const double external_acc[3], const orc_astar_result *r, double delta_t, double *pts, int cap)
{
    Search Sv;
    memset(&Sv, 0, sizeof Sv);
    Sv.P = P;
    memcpy(Sv.external_acc, external_acc, sizeof Sv.external_acc);
    int n = 0;
    double *tmp = (double *)malloc((size_t)cap * 3 * sizeof(double));
    for (int q = r->n_path - 1; q >= 1; q--) { /* node = path_nodes_[q], node->parent = path_nodes_[q - 1] */
        const double *ut = r->path_input[q], *x0 = r->path_state[q - 1];
        const double duration = r->path_duration[q];
        for (double t = duration; t >= -1e-5; t -= delta_t) {
            double xt[6];
            state_transit(&Sv, x0, xt, ut, t);
            if (n < cap) { tmp[3 * n] = xt[0]; tmp[3 * n + 1] = xt[1]; tmp[3 * n + 2] = xt[2]; }
            n++;
        }
    }
    const int ns = n < cap ? n : cap;
    for (int i = 0; i < ns; i++) memcpy(pts + 3 * i, tmp + 3 * (ns - 1 - i), 3 * sizeof(double)); /* reverse */
    free(tmp);
    n = ns;
    if (r->is_shot_succ) {
        for (double t = delta_t; t <= r->t_shot; t += delta_t) {
            const double tt[4] = {1.0, t, t * t, t * t * t};
            double coord[3];
            for (int dim = 0; dim < 3; dim++) {
                const double *c = r->coef_shot + dim * 4;
                coord[dim] = ((c[0] * tt[0] + c[1] * tt[1]) + c[2] * tt[2]) + c[3] * tt[3];
            }
            int differs = 1;
            if (n >= 1 && n <= cap) {
                const double *b = pts + 3 * (n - 1);
                const double dx = b[0] - coord[0], dy = b[1] - coord[1], dz = b[2] - coord[2];
                differs = sqrt(dx * dx + dy * dy + dz * dz) > 0.0;
            }
            if (n < 1 || differs) {
                if (n < cap) { pts[3 * n] = coord[0]; pts[3 * n + 1] = coord[1]; pts[3 * n + 2] = coord[2]; }
                n++;
            }
        }
    }
    return n;
}

/* Test helper: replays the inputs of a path (node q applied from the state reached so far, for its duration) under the external
 * acceleration `external_acc` from the path's first state, sampling every primitive check_num times like the search does
 * (kinodynamic_astar.cpp:190-199).  Returns -1 when every sample is free and in range, else the index of the first path node
 * whose primitive collides or leaves the map. */
int orc_astar_replay(const orc_astar_params *P, const double external_acc[3], const orc_astar_result *r)
{
    Search Sv;
    memset(&Sv, 0, sizeof Sv);
    Sv.P = P; Sv.map.P = P; Sv.map.resolution_inv = 1.0 / P->resolution;
    memcpy(Sv.external_acc, external_acc, sizeof Sv.external_acc);
    double cur[6];
    memcpy(cur, r->path_state[0], sizeof cur);
    for (int q = 1; q < r->n_path; q++) {
        for (int k = 1; k <= P->check_num; ++k) {
            double xt[6];
            state_transit(&Sv, cur, xt, r->path_input[q], r->path_duration[q] * (double)k / (double)P->check_num);
            if (!in_range_open(P, xt) || !check_state(&Sv.map, xt, xt + 3, 1.5)) return q;
        }
        double nx[6];
        state_transit(&Sv, cur, nx, r->path_input[q], r->path_duration[q]);
        memcpy(cur, nx, sizeof cur);
    }
    return -1;
}

/* NMPCSolver::getKinoPath's search part (nmpc_solver.cpp:154-207): search with the continuous initial state (init = true), on
 * NO_PATH once more with the full primitive set (init = false); then kino_path_ = getKinoTraj(Ts).  Returns the status of
 * the search that produced the path (NO_PATH: no path, kino_size = 0). */
int orc_astar_plan(const orc_astar_params *P, const double start_pt[3], const double start_v[3], const double start_a[3],
                   const double end_pt[3], const double end_v[3], int init, const double external_acc[3], double Ts,
                   double *kino_path, int cap, int *kino_size, orc_astar_result *res, int *retried, const double *retry_pt, const double *retry_v)
{
    int status = orc_astar_search(P, start_pt, start_v, start_a, end_pt, end_v, init, external_acc, res);
    *retried = 0;
    if (status == ORC_ASTAR_NO_PATH && init) {
        *retried = 1; /* "retry searching with discontinuous initial state": from the odometry state (start_pt_, start_v_), :190-193 */
        status = orc_astar_search(P, retry_pt ? retry_pt : start_pt, retry_v ? retry_v : start_v, start_a, end_pt, end_v, 0, external_acc, res);
    }
    *kino_size = status == ORC_ASTAR_NO_PATH ? 0 : orc_astar_kino_traj(P, external_acc, res, Ts, kino_path, cap);
    return status;
}

void orc_astar_batch(int B, const orc_astar_params *P, const double *start_pt, const double *start_v, const double *start_a,
                     const double *end_pt, const double *end_v, int init, const double *external_acc, double Ts, double *kino_path,
                     int cap, int *kino_size, int *status, orc_astar_result *res, int *retried, int nthreads, const double *retry_pt, const double *retry_v)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; b++) {
        orc_astar_result local, *r = res ? res + b : &local;
        int rt = 0;
        status[b] = orc_astar_plan(P, start_pt + 3 * b, start_v + 3 * b, start_a + 3 * b, end_pt + 3 * b, end_v + 3 * b, init,
                                   external_acc + 3 * b, Ts, kino_path + (size_t)b * cap * 3, cap, kino_size + b, r, &rt,
                                   retry_pt ? retry_pt + 3 * b : 0, retry_v ? retry_v + 3 * b : 0);
        if (retried) retried[b] = rt;
    }
}
