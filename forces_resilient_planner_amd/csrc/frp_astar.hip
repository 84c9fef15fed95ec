// frp_astar.hip -- SURVEY 8f row f-4, second half: the kinodynamic A* front end for B planners, on the device.
//
// Replaces, per planner, what NMPCSolver::getKinoPath runs on a goal / replan (plan_manage/src/nmpc_solver.cpp:154-215):
//   KinodynamicAstar::search     path_searching/src/kinodynamic_astar.cpp:17-287  (primitives with the external acceleration,
//                                stateTransit :828-845, f_ext at :838; heuristic :322-357, :426-501; one-shot :359-424)
//   the retry with the full primitive set on NO_PATH (nmpc_solver.cpp:190-207)
//   KinodynamicAstar::getKinoTraj(Ts)   :648-695  -> kino_path_, exactly the input of frp_nmpc_reference_batch
// and the occupancy queries of OccMap::checkState (occ_grid/src/occ_map.cpp:645-718, raycast.cpp:263-365).
//
// One workgroup (sixteen wavefronts, see NT) = one planner.  A search is a chain of expansions; inside an expansion
//   * one lane pops the open set: a binary heap of (f, node) pairs in HBM that reproduces std::priority_queue's
//     __push_heap / __adjust_heap step by step -- the reference changes keys in place without re-heapifying
//     (kinodynamic_astar.cpp:220-226, :263-272), so the pop order is defined by those algorithms and nothing else; since round 4
//     the pop runs on the last wavefront WHILE the others stage the window and run phase 1 (it touches nothing they read);
//   * all lanes stage the occupancy columns around the node in LDS: the map is bit-packed once per call, one 64-bit word per
//     (x, y) column (z = bit), and a 64 x 64-column window (32 KB) covers every cell a primitive of this node can touch;
//   * thread = primitive (125 inputs x 1 duration; 1 x 8 for the first expansion of a continuous start): state transit, range /
//     closed-set / velocity / same-voxel tests; thread = (primitive, collision sample) for the check_num samples through the
//     staged window; thread = surviving primitive for the cost and the quartic heuristic; then every survivor finds the first
//     survivor of its voxel;
//   * the first wavefront commits the survivors in input order -- node creation, in-place updates, heap pushes -- which is where the
//     reference's sequential semantics live: it walks the survivors' bit mask (~17 of 125 primitives), takes their values through
//     v_readlane, and climbs a push with all its ancestors fetched at once (round 4: 91 -> 47 us per expansion, 1104 -> 2130
//     searches/s on the pillar world, profiles/r04_astar_bench.jsonl; what is left is the collision samples, ~800 instructions
//     each x ~900 per expansion = the FP64 issue rate of one CU).
// The closed / expanded set is an open-addressing hash from the voxel index to the node (exact map semantics).  Arithmetic
// follows oracle/astar_oracle.c operation by operation (this file is compiled with -ffp-contract=off; cbrt / acos / cos are
// the same fdlibm sequences), so node order, node count and path samples agree with it to the bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdint.h>
#include "../../include/frp_nmpc.h"

namespace frp {
namespace astar {

constexpr int MAX_CAND = 128;    // primitives per expansion (125 in the reference's configuration)
constexpr int MAX_PATH = FRP_ASTAR_MAX_PATH;
constexpr int WIN = 64;          // window of staged occupancy columns: WIN x WIN words
// Threads per planner.  A batch ends with its longest search (a chain of expansions, 10 k of them in the pillar world of
// tests/tools/astar_bench.py against a mean of 650), so what counts is the latency of ONE expansion, not planners in flight: sixteen
// wavefronts on a CU of their own deal the ~900 collision samples of an expansion in one pass (measured, us per expansion of the
// longest search: 256 threads 65, 512 59, 768 50, 1024 47; 4096 trivial searches of an empty world 13.0 ms against 10.4 at 256).
#ifndef FRP_ASTAR_NT
#define FRP_ASTAR_NT 1024
#endif
constexpr int NT = FRP_ASTAR_NT;
constexpr char IN_CLOSE_SET = 'a', IN_OPEN_SET = 'b';

struct Node { // 128 bytes
    double state[6], g, f, input[3], duration;
    long long key;
    int parent, heap_pos, node_state, pad[3];
};
static_assert(sizeof(Node) == 128, "node record");
struct HeapEnt { double f; int id, pad; };
struct HashEnt { long long key; int val, pad; };

struct Args {
    frp_nmpc_astar p;
    const unsigned long long *packed; // [gx][gy] occupancy columns (z = bit), or null when gz > 64
    int fast, off[3];                 // ray cell -> voxel index is an integer offset (checked on the host)
    Node *nodes; HeapEnt *heap; HashEnt *hash; // per-planner areas of allocate_num / allocate_num / hcap entries
    int hcap;
};

// ------------------------------------------------------------------ deterministic elementary functions (fdlibm; see astar_oracle.c)
__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); }
__device__ __forceinline__ unsigned lo_word(double x) { return (unsigned)(unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double from_words(unsigned hi, unsigned lo) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }

__device__ double det_cbrt(double x)
{
    const unsigned B1 = 715094163u, B2 = 696219795u;
    const double C = 5.42857142857142815906e-01, D = -7.05306122448979611050e-01, E = 1.41428571428571436819e+00,
                 F = 1.60714285714285720630e+00, G = 3.57142857142857150787e-01;
    unsigned hx = hi_word(x);
    const unsigned sign = hx & 0x80000000u;
    hx ^= sign;
    if (hx >= 0x7ff00000u) return x + x;
    if ((hx | lo_word(x)) == 0) return x;
    x = from_words(hx, lo_word(x));
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

__device__ double det_acos(double x)
{
    const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const unsigned hx = hi_word(x), ix = hx & 0x7fffffffu;
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

__device__ double det_cos(double x)
{
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double s = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double c = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double b = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -b : b;
}

// ------------------------------------------------------------------ per-planner context
struct Ctx {
    const frp_nmpc_astar *P;
    const unsigned char *occ;
    const unsigned long long *packed;
    const unsigned long long *win; // LDS window, WIN x WIN columns starting at (wx0, wy0)
    int wx0, wy0, use_win, win_r; // win_r: the window holds the columns [wx0, wx0 + 2 win_r] x [wy0, wy0 + 2 win_r]
    int lmin[3], lmax[3], use_local;
    int fast, off[3]; // fast: the voxel of ray cell c is c + off on every axis (verified on the host for |c| <= CELL_SAFE)
    double res_inv, ext[3];
    double res_y, cn, cn_y; // div_rcp of the resolution and of check_num (divisors of every collision sample)
    // the box of RAY CELLS whose state is a bit of the staged window -- inside the map, inside the local range, inside the window,
    // inside the range the integer cell -> voxel offset was verified for -- as first cell, extent (0 = empty) and window / bit origin
    int fx0, fy0, fz0, fwx, fwy, fwz;
    unsigned fxn, fyn, fzn;
};
constexpr int CELL_SAFE = 4096;

// the state of the voxel with map index (i0, i1, i2): getVoxelState after its posToIndex
__device__ __forceinline__ int voxel_state_idx(const Ctx &c, int i0, int i1, int i2)
{
    const frp_nmpc_astar *P = c.P;
    if (!((i0 | (P->grid[0] - 1 - i0) | i1 | (P->grid[1] - 1 - i1) | i2 | (P->grid[2] - 1 - i2)) >= 0)) return -1;
    if (c.use_local &&
        !(((i0 - c.lmin[0]) | (c.lmax[0] - i0) | (i1 - c.lmin[1]) | (c.lmax[1] - i1) | (i2 - c.lmin[2]) | (c.lmax[2] - i2)) >= 0))
        return 0;
    if (c.packed) {
        const int wx = i0 - c.wx0, wy = i1 - c.wy0;
        const unsigned long long w = (c.use_win && ((wx | (2 * c.win_r - wx) | wy | (2 * c.win_r - wy)) >= 0)) ? c.win[wx * WIN + wy]
                                                                                                  : c.packed[(size_t)i0 * P->grid[1] + i1];
        return (int)((w >> i2) & 1ull);
    }
    return c.occ[((size_t)i0 * P->grid[1] + i1) * P->grid[2] + i2] ? 1 : 0;
}

// OccMap::getVoxelState (occ_map.cpp:95-106): -1 outside the map, 0 free (or outside the local range), 1 occupied
__device__ __forceinline__ int voxel_state(const Ctx &c, double px, double py, double pz)
{
    const frp_nmpc_astar *P = c.P;
    return voxel_state_idx(c, (int)floor((px - P->origin[0]) * c.res_inv), (int)floor((py - P->origin[1]) * c.res_inv),
                           (int)floor((pz - P->origin[2]) * c.res_inv));
}
// the voxel state of ray cell (x, y, z) (getlineGrids turns the cell into its centre x res + res / 2 and getVoxelState back
// into an index: with `fast` that round trip is the verified integer offset)
__device__ __forceinline__ int cell_state(const Ctx &c, int x, int y, int z)
{
    if (c.fast && ((x + CELL_SAFE) | (CELL_SAFE - x) | (y + CELL_SAFE) | (CELL_SAFE - y) | (z + CELL_SAFE) | (CELL_SAFE - z)) >= 0)
        return voxel_state_idx(c, x + c.off[0], y + c.off[1], z + c.off[2]);
    const double res = c.P->resolution;
    return voxel_state(c, (double)x * res + res / 2.0, (double)y * res + res / 2.0, (double)z * res + res / 2.0);
}

// a / b the way the compiler expands an FP64 division (rcp, two Newton steps on the reciprocal, q0 = a y, one residual step), split so
// that the reciprocal of a divisor that is used again -- the map resolution: twelve times per collision sample -- is formed once.
// Without the range scaling and the special-case fix-up of that expansion: for finite operands whose quotient is far from the
// denormal and overflow ranges (coordinates over a resolution, a vector over its norm, a sample index over check_num) the result is
// the same bits, i.e. the correctly rounded quotient the reference's CPU division produces (tests/test_gpu_astar.py: bit-identical).
__device__ __forceinline__ double div_rcp(double b)
{
    const double y0 = __builtin_amdgcn_rcp(b);
    const double y1 = __builtin_fma(y0, __builtin_fma(-b, y0, 1.0), y0);
    return __builtin_fma(y1, __builtin_fma(-b, y1, 1.0), y1);
}
__device__ __forceinline__ double div_y(double a, double b, double y)
{
    const double q0 = a * y;
    const double q = __builtin_fma(__builtin_fma(-b, q0, a), y, q0);
    return a == 0.0 ? a : q; // (+-0 / b keeps its sign)
}

// cell_state for the cells of the box above: three range tests, one LDS read (everything else: the general path)
__device__ __forceinline__ int cell_state_box(const Ctx &c, int x, int y, int z)
{
    const unsigned ux = (unsigned)(x - c.fx0), uy = (unsigned)(y - c.fy0), uz = (unsigned)(z - c.fz0);
    if (ux < c.fxn && uy < c.fyn && uz < c.fzn) return (int)((c.win[(ux + c.fwx) * WIN + (uy + c.fwy)] >> (uz + c.fwz)) & 1ull);
    return cell_state(c, x, y, z);
}
// (re)compute the box for the window at (wx0, wy0) -- uniform over the workgroup, a few scalar operations per expansion
__device__ __forceinline__ void set_cell_box(Ctx &c)
{
    c.fxn = c.fyn = c.fzn = 0;
    if (!(c.fast && c.packed && c.use_win)) return;
    const frp_nmpc_astar *P = c.P;
    int lo[3] = {c.wx0, c.wy0, 0}, hi[3] = {c.wx0 + 2 * c.win_r, c.wy0 + 2 * c.win_r, P->grid[2] - 1};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        lo[i] = max(lo[i], 0); hi[i] = min(hi[i], P->grid[i] - 1);
        if (c.use_local) { lo[i] = max(lo[i], c.lmin[i]); hi[i] = min(hi[i], c.lmax[i]); }
        lo[i] = max(lo[i] - c.off[i], -CELL_SAFE); hi[i] = min(hi[i] - c.off[i], CELL_SAFE); // ray cells
    }
    if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2] || hi[2] + c.off[2] > 63) return;
    c.fx0 = lo[0]; c.fy0 = lo[1]; c.fz0 = lo[2];
    c.fxn = (unsigned)(hi[0] - lo[0] + 1); c.fyn = (unsigned)(hi[1] - lo[1] + 1); c.fzn = (unsigned)(hi[2] - lo[2] + 1);
    c.fwx = lo[0] + c.off[0] - c.wx0; c.fwy = lo[1] + c.off[1] - c.wy0; c.fwz = lo[2] + c.off[2];
}

__device__ __forceinline__ int signum_i(int x) { return x == 0 ? 0 : (x < 0 ? -1 : 1); }
// mod(value, 1) = fmod(fmod(value, 1) + 1, 1) (raycast.cpp:11-14); fmod(v, 1) = v - trunc(v) exactly (the fraction of a double is a double)
__device__ __forceinline__ double mod1(double v) { const double f = v - trunc(v); const double t = f + 1.0; return t - trunc(t); }
__device__ __forceinline__ double intbound(double s, double ds)
{
    if (ds < 0) { s = -s; ds = -ds; }
    s = mod1(s);
    return (1 - s) / ds;
}

// getlineGrids (occ_map.cpp:686-718) + the scan over its cells in checkState: 1 = some cell of the segment is not free
__device__ int line_hits(const Ctx &c, double s0, double s1, double s2, double e0, double e1, double e2)
{
    const double res = c.P->resolution, ry = c.res_y;
    const double a0 = div_y(s0, res, ry), a1 = div_y(s1, res, ry), a2 = div_y(s2, res, ry), b0 = div_y(e0, res, ry), b1 = div_y(e1, res, ry),
                 b2 = div_y(e2, res, ry);
    int x = (int)floor(a0), y = (int)floor(a1), z = (int)floor(a2);
    const int endX = (int)floor(b0), endY = (int)floor(b1), endZ = (int)floor(b2);
    const double dx = endX - x, dy = endY - y, dz = endZ - z;
    const int stepX = signum_i((int)dx), stepY = signum_i((int)dy), stepZ = signum_i((int)dz);
    if (stepX == 0 && stepY == 0) {
        // a vertical segment visits the cells z .. endZ of one column (the traversal below would only ever step in z)
        const int zlo = z < endZ ? z : endZ, zhi = z < endZ ? endZ : z;
        for (int zz = zlo; zz <= zhi; zz++)
            if (cell_state_box(c, x, y, zz) != 0) return 1;
        return 0;
    }
    // tMax = intbound(a, d) = (1 - mod1(+-a)) / |d| and tDelta = step / d = 1 / |d| share the reciprocal of |d|.  An axis the segment
    // does not move on (d = 0) has tMax = (1 - s) / 0 = +inf exactly (s = mod1(.) < 1) and a tDelta that is never added.
    auto axis = [](double a, double d, double &tmax, double &tdelta) {
        const double ad = fabs(d), adn = d == 0.0 ? 1.0 : ad;
        const double y = div_rcp(adn), s = mod1(d < 0 ? -a : a);
        tmax = (d == 0.0 && a == a) ? __builtin_huge_val() : div_y(1 - s, adn, y); // (a NaN coordinate stays a NaN, as in (1 - s) / 0)
        tdelta = div_y(1.0, adn, y);
    };
    double tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ;
    axis(a0, dx, tMaxX, tDeltaX); axis(a1, dy, tMaxY, tDeltaY); axis(a2, dz, tMaxZ, tDeltaZ);
    for (int guard = 0;; guard++) {
        if (x == endX && y == endY && z == endZ) break;
        if (guard >= 4096) return 1;
        if (cell_state_box(c, x, y, z) != 0) return 1;
        if (tMaxX < tMaxY) {
            if (tMaxX < tMaxZ) { x += stepX; tMaxX += tDeltaX; }
            else { z += stepZ; tMaxZ += tDeltaZ; }
        } else {
            if (tMaxY < tMaxZ) { y += stepY; tMaxY += tDeltaY; }
            else { z += stepZ; tMaxZ += tDeltaZ; }
        }
    }
    return cell_state_box(c, endX, endY, endZ) != 0; // "check end": the cell floor(end), occ_map.cpp:704-717
}

// OccMap::checkState (occ_map.cpp:645-684): 1 = free
__device__ int check_state(const Ctx &c, const double pos[3], const double vel[3], double inflate_ratio)
{
    double vh0 = vel[0], vh1 = vel[1];
    const double v_hor_norm = sqrt(vh0 * vh0 + vh1 * vh1);
    if (v_hor_norm < 1e-4) { vh0 = 1; vh1 = 1; }
    double cw0 = vh1, cw1 = -vh0;
    const double n2 = cw0 * cw0 + cw1 * cw1;
    if (n2 > 0.0) { const double nn = sqrt(n2), ny = div_rcp(nn); cw0 = div_y(cw0, nn, ny); cw1 = div_y(cw1, nn, ny); }
    cw0 = cw0 * c.P->ego_r * inflate_ratio; cw1 = cw1 * c.P->ego_r * inflate_ratio;
    if (line_hits(c, pos[0] + cw0, pos[1] + cw1, pos[2], pos[0] - cw0, pos[1] - cw1, pos[2])) return 0;
    if (line_hits(c, pos[0], pos[1], pos[2] + c.P->ego_h * inflate_ratio, pos[0], pos[1], pos[2] - c.P->ego_h * inflate_ratio)) return 0;
    return 1;
}

__device__ __forceinline__ double rl_f64(double v, int l) // v_readlane of a double (l wave-uniform)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l)));
}

// position of the n-th (0-based) set bit of m, n < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n)
{
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int cnt = __popcll(m & (((1ull << w) - 1ull) << pos));
        if (n >= cnt) { n -= cnt; pos += w; }
    }
    return pos;
}

__device__ __forceinline__ void state_transit(const Ctx &c, const double s0[6], double s1[6], const double um[3], double tau)
{
    const double t2 = tau * tau;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double ud = um[i] + c.ext[i];
        s1[i] = (s0[i] + tau * s0[3 + i]) + 0.5 * t2 * ud;
        s1[3 + i] = s0[3 + i] + tau * ud;
    }
}

// Eigen evaluates the reduction of a fixed-size 3-vector as a0 b0 + (a1 b1 + a2 b2) (redux_novec_unroller splits [0, 3) into [0, 1) and
// [1, 3)); the search is steered by comparisons of nearly equal costs, so the order of these sums is part of the restatement.  The
// 4-term dots of the shot polynomials (dynamic-size VectorXd: packet reductions that depend on the build's SIMD width) stay left
// to right: parity with the reference at that level is structural, not bitwise.
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

// cubic(a, b, c, d).front() (kinodynamic_astar.cpp:426-459): quartic() uses only the first root the reference computes
__device__ double cubic_first(double a, double b, double c, double d)
{
    const double a2 = b / a, a1 = c / a, a0 = d / a;
    const double Q = (3 * a1 - a2 * a2) / 9;
    const double R = (9 * a1 * a2 - 27 * a0 - 2 * a2 * a2 * a2) / 54;
    const double D = Q * Q * Q + R * R;
    if (D > 0) {
        const double S = det_cbrt(R + sqrt(D)), T = det_cbrt(R - sqrt(D));
        return -a2 / 3 + (S + T);
    } else if (D == 0) {
        const double S = det_cbrt(R);
        return -a2 / 3 + S + S;
    } else {
        const double theta = det_acos(R / sqrt(-Q * Q * Q));
        return 2 * sqrt(-Q) * det_cos(theta / 3) - a2 / 3;
    }
}

// quartic (:461-501): the roots in the reference's order; an absent root is reported as -1 (the caller skips t < t_bar)
__device__ void quartic(double a, double b, double c, double d, double e, double &r0, double &r1, double &r2, double &r3)
{
    r0 = r1 = r2 = r3 = -1.0;
    const double a3 = b / a, a2 = c / a, a1 = d / a, a0 = e / a;
    const double y1 = cubic_first(1, -a2, a1 * a3 - 4 * a0, 4 * a2 * a0 - a1 * a1 - a3 * a3 * a0);
    const double r = a3 * a3 / 4 - a2 + y1;
    if (r < 0) return;
    const double R = sqrt(r);
    double D, E;
    if (R != 0) {
        D = sqrt(0.75 * a3 * a3 - R * R - 2 * a2 + 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
        E = sqrt(0.75 * a3 * a3 - R * R - 2 * a2 - 0.25 * (4 * a3 * a2 - 8 * a1 - a3 * a3 * a3) / R);
    } else {
        D = sqrt(0.75 * a3 * a3 - 2 * a2 + 2 * sqrt(y1 * y1 - 4 * a0));
        E = sqrt(0.75 * a3 * a3 - 2 * a2 - 2 * sqrt(y1 * y1 - 4 * a0));
    }
    if (!(D != D)) { r0 = -a3 / 4 + R / 2 + D / 2; r1 = -a3 / 4 + R / 2 - D / 2; }
    if (!(E != E)) { r2 = -a3 / 4 - R / 2 + E / 2; r3 = -a3 / 4 - R / 2 - E / 2; }
}

// estimateHeuristic (kinodynamic_astar.cpp:322-357)
__device__ double estimate_heuristic(const frp_nmpc_astar *P, const double x1[6], const double x2[6], double *optimal_time)
{
    double dp[3], v0[3], v1[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { dp[i] = x2[i] - x1[i]; v0[i] = x1[3 + i]; v1[i] = x2[3 + i]; }
    // (dot3: Eigen's unrolled reduction of a fixed-size 3-vector, a0 b0 + (a1 b1 + a2 b2) -- what dp.dot(dp), v0.dot(v1), .norm() compile to)
    const double vs[3] = {v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2]};
    const double c1 = -36 * dot3(dp, dp);
    const double c2 = 24 * dot3(vs, dp);
    const double c3 = -4 * ((dot3(v0, v0) + dot3(v0, v1)) + dot3(v1, v1));
    double ts[5];
    quartic(P->w_time, 0, c3, c2, c1, ts[0], ts[1], ts[2], ts[3]);
    const double t_bar = fmax(fmax(fabs(x1[0] - x2[0]), fabs(x1[1] - x2[1])), fabs(x1[2] - x2[2])) / P->max_vel;
    ts[4] = t_bar;
    double cost = 100000000, t_d = t_bar;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const double t = ts[i];
        if (t < t_bar) continue; // (also skips the roots that do not exist)
        const double c = -c1 / (3 * t * t * t) - c2 / (2 * t * t) - c3 / t + P->w_time * t;
        if (c < cost) { cost = c; t_d = t; }
    }
    *optimal_time = t_d;
    return 1.0 * (1 + P->tie_breaker) * cost;
}

__device__ __forceinline__ long long pack_index(int i0, int i1, int i2)
{
    return (((long long)(i0 + (1 << 20))) << 42) | (((long long)(i1 + (1 << 20))) << 21) | (long long)(i2 + (1 << 20));
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ int hash_find(HashEnt *h, int hcap, long long key)
{
    for (unsigned long long s = mix64((unsigned long long)key) & (unsigned long long)(hcap - 1);; s = (s + 1) & (unsigned long long)(hcap - 1)) {
        const int v = h[s].val;
        if (v < 0) return -1;
        if (h[s].key == key) return v;
    }
}
__device__ void hash_insert(HashEnt *h, int hcap, long long key, int node)
{
    for (unsigned long long s = mix64((unsigned long long)key) & (unsigned long long)(hcap - 1);; s = (s + 1) & (unsigned long long)(hcap - 1)) {
        if (h[s].val < 0) { h[s].key = key; h[s].val = node; return; }
        if (h[s].key == key) return;
    }
}

// 1 (shipped): every push of an expansion fetches its own ancestor chain -- the round-4 commit.  0: the round-5 experiment (VERDICT r04
// item 7: the ancestor chains of ALL pushes of an expansion fetched in one trip and the pushes replayed on the cached copy).  Measured on
// one box (tools/dbg/astar_ab2.sh, pillar world, 1024 planners): the cached walk 2309 searches/s, 43.6 us per expansion of the longest
// search; this one 2424 and 41.6 -- the union of the chains is ~4x the loads of one chain and the replay's bookkeeping costs what the
// saved trips return (profile build: commit walk 13.2 k cycles per expansion against 12.3 k).  Kept for A/B runs.
#ifndef FRP_ASTAR_SLOW_PUSH
#define FRP_ASTAR_SLOW_PUSH 1
#endif
// std::__push_heap with NodeComparator (f_score greater = lower priority); entries carry their node's current f
__device__ void heap_push_hole(HeapEnt *heap, Node *nodes, int hole, int top, double vf, int vid)
{
    int parent = (hole - 1) / 2;
    while (hole > top && heap[parent].f > vf) {
        const double pf = heap[parent].f; const int pid = heap[parent].id;
        heap[hole].f = pf; heap[hole].id = pid; nodes[pid].heap_pos = hole;
        hole = parent;
        parent = (hole - 1) / 2;
    }
    heap[hole].f = vf; heap[hole].id = vid; nodes[vid].heap_pos = hole;
}
// The same push executed by a whole wavefront (all lanes call it with wave-uniform arguments): lane j fetches the j-th ancestor of
// the hole -- index ((hole + 1) >> j) - 1 -- so the whole climb costs ONE trip to the L2 instead of one per level (a climb is a chain
// of dependent loads otherwise: ~550 cycles each, 2 levels on average); lane 0 then does exactly heap_push_hole's stores.
__device__ __forceinline__ void heap_push_hole_wave(HeapEnt *heap, Node *nodes, int hole, double vf, int vid)
{
    const int ln = threadIdx.x & 63;
    const int anc = ((hole + 1) >> ln) - 1; // lane 0: the hole itself (unused), lane 1: its parent, ...
    const bool has = ln >= 1 && ln < 32 && anc >= 0;
    const double af = has ? heap[anc].f : 0.0;
    const int aid = has ? heap[anc].id : 0;
    int j = 1;
    while (hole > 0) {
        const double pf = rl_f64(af, j);
        if (!(pf > vf)) break;
        const int pid = __builtin_amdgcn_readlane(aid, j);
        if (ln == 0) { heap[hole].f = pf; heap[hole].id = pid; nodes[pid].heap_pos = hole; }
        hole = (hole - 1) / 2;
        j++;
    }
    if (ln == 0) { heap[hole].f = vf; heap[hole].id = vid; nodes[vid].heap_pos = hole; }
}
__device__ void heap_pop(HeapEnt *heap, Node *nodes, int &size) // std::pop_heap + pop_back
{
    if (size > 1) {
        const int last = size - 1;
        const double vf = heap[last].f; const int vid = heap[last].id;
        const int len = last;
        int hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (heap[second].f > heap[second - 1].f) second--;
            const double cf = heap[second].f; const int cid = heap[second].id;
            heap[hole].f = cf; heap[hole].id = cid; nodes[cid].heap_pos = hole;
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            const double cf = heap[second - 1].f; const int cid = heap[second - 1].id;
            heap[hole].f = cf; heap[hole].id = cid; nodes[cid].heap_pos = hole;
            hole = second - 1;
        }
        heap_push_hole(heap, nodes, hole, 0, vf, vid);
    }
    size--;
}

#ifdef FRP_ASTAR_PROFILE // cycles per phase of a search, written into the last path_nodes row of the planner
#define APROF(i) do { const long long tn_ = clock64(); if (threadIdx.x == 0) sh.prof[i] += tn_ - pt_; pt_ = tn_; } while (0)
#define APROF_DECL() long long pt_ = clock64()
#else
#define APROF(i)
#define APROF_DECL()
#endif
struct Shared {
    long long prof[11];
    unsigned long long alive[2]; // survivors of phase 1 as bit masks over the primitives 0..63, 64..127
    unsigned long long vx_key[256]; // first-survivor-of-a-voxel search: open-addressing table voxel key -> lowest candidate index
    int vx_min[256];
    unsigned long long win[WIN * WIN];
    double c_state[MAX_CAND][6], c_g[MAX_CAND], c_f[MAX_CAND], c_um[MAX_CAND][3], c_tau[MAX_CAND];
    long long c_key[MAX_CAND];
    int c_pre[MAX_CAND], c_surv[MAX_CAND], c_leader[MAX_CAND], c_created[MAX_CAND], c_winner[MAX_CAND];
    double cur_state[6], cur_g, end_state[6], coef_shot[12], t_shot;
    int cur, cur_index[3], cur_parent, heap_size, use_node_num, iter_num, status, terminate, n_cand, shot_ok, shot_fail, is_shot_succ;
    int path_ids[MAX_PATH], n_path, n_in, n_dur_init;
    double in_list[MAX_CAND][3], dur_init[16], dur_full;
};

// One search (KinodynamicAstar::search); leaves status / terminate / shot in the shared block.
__device__ __forceinline__ void run_search(const Args &a, Shared &sh, Ctx &ctx, int b, bool init, bool retry)
{
    const frp_nmpc_astar *P = &a.p;
    const int lane = threadIdx.x;
    const int A = P->allocate_num;
    Node *nodes = a.nodes + (size_t)b * A;
    HeapEnt *heap = a.heap + (size_t)b * A;
    HashEnt *hash = a.hash + (size_t)b * a.hcap;
    // (the repeated search starts from the odometry state when the caller gives one: nmpc_solver.cpp:190-193)
    const double *start_pt = (retry && P->retry_pt ? P->retry_pt : P->start_pt) + 3 * b, *start_v = (retry && P->retry_vel ? P->retry_vel : P->start_vel) + 3 * b,
                 *start_a = P->start_acc + 3 * b;
    const double *end_pt = P->end_pt + 3 * b, *end_v = P->end_vel + 3 * b;
    // reset(): expanded_nodes_.clear(), open set emptied, counters zeroed
    for (int i = lane; i < a.hcap; i += NT) hash[i].val = -1;
    __syncthreads();
    if (lane == 0) {
        sh.heap_size = 0; sh.use_node_num = 0; sh.iter_num = 0; sh.is_shot_succ = 0; sh.status = FRP_ASTAR_NO_PATH; sh.terminate = -1;
        double st[6], ttg;
        int idx[3];
        for (int i = 0; i < 3; i++) { st[i] = start_pt[i]; st[3 + i] = start_v[i]; sh.end_state[i] = end_pt[i]; sh.end_state[3 + i] = end_v[i]; }
        for (int i = 0; i < 3; i++) idx[i] = (int)floor((start_pt[i] - P->origin[i]) * ctx.res_inv);
        const double f0 = P->lambda_heu * estimate_heuristic(P, st, sh.end_state, &ttg);
        for (int i = 0; i < 6; i++) nodes[0].state[i] = st[i];
        nodes[0].g = 0.0; nodes[0].f = f0; nodes[0].parent = -1; nodes[0].node_state = IN_OPEN_SET;
        nodes[0].key = pack_index(idx[0], idx[1], idx[2]);
        nodes[0].input[0] = nodes[0].input[1] = nodes[0].input[2] = 0.0; nodes[0].duration = 0.0;
        heap[0].f = f0; heap[0].id = 0; nodes[0].heap_pos = 0;
        sh.heap_size = 1; sh.use_node_num = 1;
        hash_insert(hash, a.hcap, nodes[0].key, 0);
    }
    __syncthreads();
    int end_index[3];
    for (int i = 0; i < 3; i++) end_index[i] = (int)floor((end_pt[i] - P->origin[i]) * ctx.res_inv);
    const int tolerance = (int)ceil(1 / P->resolution);
    bool init_search = init;
    APROF_DECL();

    for (;;) {
        if (sh.heap_size == 0) { if (lane == 0) { sh.status = FRP_ASTAR_NO_PATH; sh.terminate = -1; } break; }
        // ---- the node with the lowest f
        if (lane == 0) {
            const int cur = heap[0].id;
            sh.cur = cur;
            for (int i = 0; i < 6; i++) sh.cur_state[i] = nodes[cur].state[i];
            sh.cur_g = nodes[cur].g; sh.cur_parent = nodes[cur].parent;
            const long long key = nodes[cur].key; // PathNode::index (set at creation, :239)
            sh.cur_index[0] = (int)((key >> 42) & 0x1fffff) - (1 << 20);
            sh.cur_index[1] = (int)((key >> 21) & 0x1fffff) - (1 << 20);
            sh.cur_index[2] = (int)(key & 0x1fffff) - (1 << 20);
        }
        __syncthreads();
        APROF(0);
        const int cur = sh.cur;
        const bool near_end = abs(sh.cur_index[0] - end_index[0]) <= tolerance && abs(sh.cur_index[1] - end_index[1]) <= tolerance &&
                              abs(sh.cur_index[2] - end_index[2]) <= tolerance;
        const double dst[3] = {sh.cur_state[0] - start_pt[0], sh.cur_state[1] - start_pt[1], sh.cur_state[2] - start_pt[2]};
        const bool reach_horizon = sqrt(dot3(dst, dst)) >= P->horizon;
        // ---- pop node and add to close set (kinodynamic_astar.cpp:108-112) -- by the first lane of the LAST wavefront, while the
        // others stage the window and run phase 1: the sift-down is a chain of dependent loads that touches nothing those read
        // (the heap and heap_pos fields only), and its result is not needed before the commit
        const bool terminating = reach_horizon || near_end;
        constexpr int NW = NT - 64; // lanes that share the parallel work of this stretch
        if (!terminating && lane == NW) {
            int hs = sh.heap_size;
            heap_pop(heap, nodes, hs);
            sh.heap_size = hs;
            nodes[cur].node_state = IN_CLOSE_SET;
            sh.iter_num += 1;
        }
        // ---- stage the occupancy columns around the node (also used by the one-shot check below)
        if (ctx.packed) {
            // (only the square of columns a primitive of this node can reach -- win_r columns either way, see astar_kernel -- is
            // staged; a lookup outside it reads the packed map itself, so the extent is a cost, never a correctness, matter)
            const int wr = ctx.win_r, side = 2 * wr + 1;
            const int wx0 = sh.cur_index[0] - wr, wy0 = sh.cur_index[1] - wr;
            for (int i = lane; i < side * side && lane < NW; i += NW) {
                const int ix = i / side, iy = i - ix * side;
                const int gx = wx0 + ix, gy = wy0 + iy;
                sh.win[ix * WIN + iy] = (gx >= 0 && gx < P->grid[0] && gy >= 0 && gy < P->grid[1]) ? ctx.packed[(size_t)gx * P->grid[1] + gy] : 0ull;
            }
            ctx.wx0 = wx0; ctx.wy0 = wy0; ctx.use_win = 1;
            set_cell_box(ctx);
        }
        if (terminating) {
            __syncthreads();
            APROF(1);
            if (near_end) {
                // one-shot trajectory: estimateHeuristic for its duration, computeShotTraj's ten samples on ten lanes
                double ttg;
                estimate_heuristic(P, sh.cur_state, sh.end_state, &ttg);
                const double t_d = ttg;
                double coef[12];
#pragma unroll
                for (int dim = 0; dim < 3; dim++) {
                    const double p0 = sh.cur_state[dim], dp = sh.end_state[dim] - p0, v0 = sh.cur_state[3 + dim], v1 = sh.end_state[3 + dim], dv = v1 - v0;
                    const double ca = 1.0 / 6.0 * (-12.0 / (t_d * t_d * t_d) * (dp - v0 * t_d) + 6 / (t_d * t_d) * dv);
                    const double cb = 0.5 * (6.0 / (t_d * t_d) * (dp - v0 * t_d) - 2 / t_d * dv);
                    coef[dim * 4 + 3] = ca; coef[dim * 4 + 2] = cb; coef[dim * 4 + 1] = v0; coef[dim * 4 + 0] = p0;
                }
                if (lane == 0) sh.shot_fail = 0;
                __syncthreads();
                // sample times by repeated addition, like the loop `for (time = t_delta; time <= t_d; time += t_delta)`
                const double t_delta = t_d / 10;
                double time = t_delta;
                int nsamp = 0;
                for (double tt = t_delta; tt <= t_d && nsamp < NT; tt += t_delta) { if (nsamp == lane) time = tt; nsamp++; }
                if (lane < nsamp) {
                    const double t[4] = {1.0, time, time * time, time * time * time};
                    double coord[3], vel[3];
#pragma unroll
                    for (int dim = 0; dim < 3; dim++) {
                        const double *cc = coef + dim * 4;
                        coord[dim] = ((cc[0] * t[0] + cc[1] * t[1]) + cc[2] * t[2]) + cc[3] * t[3];
                        vel[dim] = ((cc[1] * t[0] + (2 * cc[2]) * t[1]) + (3 * cc[3]) * t[2]) + 0.0 * t[3];
                    }
                    int bad = 0;
                    if (coord[0] < P->origin[0] || coord[0] >= P->map_size[0] * 0.5 || coord[1] < P->origin[1] || coord[1] >= P->map_size[1] * 0.5 ||
                        coord[2] < 0.1 || coord[2] >= P->map_size[2] * 0.5)
                        bad = 1;
                    else if (!check_state(ctx, coord, vel, 1.5))
                        bad = 1;
                    if (bad) atomicOr(&sh.shot_fail, 1);
                }
                __syncthreads();
                if (lane == 0) {
                    if (!sh.shot_fail) {
                        sh.is_shot_succ = 1; sh.t_shot = t_d;
#pragma unroll
                        for (int i = 0; i < 12; i++) sh.coef_shot[i] = coef[i];
                    }
                    if (sh.cur_parent < 0 && !sh.is_shot_succ) sh.status = FRP_ASTAR_NO_PATH;
                    else if (!sh.is_shot_succ) sh.status = FRP_ASTAR_REACH_END_BUT_SHOT_FAILS;
                    else sh.status = FRP_ASTAR_REACH_END;
                    sh.terminate = cur;
                }
            } else if (lane == 0) {
                sh.status = FRP_ASTAR_REACH_HORIZON;
                sh.terminate = cur;
            }
            break;
        }
        // ---- primitives of this expansion: the continuous start uses its own acceleration with eight durations, every other
        // node the full input grid with one duration (kinodynamic_astar.cpp:116-137); candidate index = input-major order
        const bool use_init = init_search;
        init_search = false;
        const int n_cand = use_init ? sh.n_dur_init : sh.n_in;
        APROF(2);
        // phase 1, thread = primitive: state transit and the tests that need no map
        for (int c = lane; c < MAX_CAND; c += NT) {
            int surv = 0, pre = -1;
            long long key = 0;
            double pro[6], um[3] = {0.0, 0.0, 0.0}, tau = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) pro[i] = 0.0;
            if (c < n_cand) {
                if (use_init) { um[0] = start_a[0]; um[1] = start_a[1]; um[2] = start_a[2]; tau = sh.dur_init[c]; }
                else { um[0] = sh.in_list[c][0]; um[1] = sh.in_list[c][1]; um[2] = sh.in_list[c][2]; tau = sh.dur_full; }
                state_transit(ctx, sh.cur_state, pro, um, tau);
                surv = 1;
                if (pro[0] <= P->origin[0] || pro[0] >= P->map_size[0] * 0.5 || pro[1] <= P->origin[1] || pro[1] >= P->map_size[1] * 0.5 ||
                    pro[2] <= 0.1 || pro[2] >= P->map_size[2] * 0.5)
                    surv = 0;
                const int i0 = (int)floor((pro[0] - P->origin[0]) * ctx.res_inv), i1 = (int)floor((pro[1] - P->origin[1]) * ctx.res_inv),
                          i2 = (int)floor((pro[2] - P->origin[2]) * ctx.res_inv);
                key = pack_index(i0, i1, i2);
                if (surv) {
                    pre = hash_find(hash, a.hcap, key);
                    if (pre >= 0 && nodes[pre].node_state == IN_CLOSE_SET) surv = 0;
                }
                if (surv && (fabs(pro[3]) > P->max_vel || fabs(pro[4]) > P->max_vel || fabs(pro[5]) > P->max_vel)) surv = 0;
                if (surv && i0 == sh.cur_index[0] && i1 == sh.cur_index[1] && i2 == sh.cur_index[2]) surv = 0;
            }
            sh.c_surv[c] = surv; sh.c_pre[c] = pre; sh.c_key[c] = key; sh.c_tau[c] = tau; sh.c_g[c] = 0.0; sh.c_f[c] = 0.0; sh.c_leader[c] = c;
            static_assert(MAX_CAND == 128 && NT - 64 >= MAX_CAND, "one primitive per thread of the first two wavefronts, the last wavefront pops meanwhile");
            const unsigned long long bal = __ballot(surv != 0); // (c = lane: wavefront w holds the primitives 64 w .. 64 w + 63)
            if ((c & 63) == 0) sh.alive[c >> 6] = bal;
#pragma unroll
            for (int i = 0; i < 6; i++) sh.c_state[c][i] = pro[i];
            sh.c_um[c][0] = um[0]; sh.c_um[c][1] = um[1]; sh.c_um[c][2] = um[2];
        }
        __syncthreads();
        APROF(3);
        // phase 2, thread = (primitive, collision sample): check_num samples per surviving primitive (kinodynamic_astar.cpp:190-199;
        // the reference stops at the first colliding sample -- the verdict of a primitive is the same)
        // The samples of the SURVIVING primitives only are dealt over the threads (the a-th survivor = the a-th set bit of the
        // masks): a dead primitive would hold check_num lanes idle through the longest ray cast of its wavefront.
        // With eight or more wavefronts the first two leave the samples to the others and evaluate cost and heuristic of the phase-1
        // survivors meanwhile (the quartic of estimateHeuristic is a 4 k-cycle dependency chain on at most 128 lanes; for a primitive
        // that turns out to collide the values are simply never used).
        constexpr bool HEUR_BESIDE = NT >= 512;
        constexpr int CW0 = HEUR_BESIDE ? MAX_CAND : 0, CWN = NT - CW0; // first lane and number of lanes of the collision phase
        const unsigned long long al0 = sh.alive[0], al1 = sh.alive[1];
        const int n_al0 = __popcll(al0), n_alive = n_al0 + __popcll(al1);
        auto cost_and_heuristic = [&](int c) {
            double ttg;
            const double um0 = sh.c_um[c][0], um1 = sh.c_um[c][1], um2 = sh.c_um[c][2];
            const double g = ((um0 * um0 + (um1 * um1 + um2 * um2)) + P->w_time) * sh.c_tau[c] + sh.cur_g; // um.squaredNorm(): see dot3
            double pro[6];
#pragma unroll
            for (int i = 0; i < 6; i++) pro[i] = sh.c_state[c][i];
            sh.c_g[c] = g;
            sh.c_f[c] = g + P->lambda_heu * estimate_heuristic(P, pro, sh.end_state, &ttg);
        };
        if (HEUR_BESIDE && lane < n_cand && sh.c_surv[lane]) cost_and_heuristic(lane);
        // samples 0 .. CWN-1 go to the collision lanes; what is left over (more than CWN / check_num phase-1 survivors) is dealt over
        // ALL lanes -- the heuristic lanes take theirs when they are done, which is sooner than a second pass of the collision lanes
        const int n_samples = n_alive * P->check_num;
        for (int t = lane >= CW0 ? lane - CW0 : CWN + lane; t < n_samples; t = t < CWN ? CWN + lane : t + NT) {
            const int ai = t / P->check_num, k = t - ai * P->check_num + 1;
            const int c = ai < n_al0 ? nth_set_bit(al0, ai) : 64 + nth_set_bit(al1, ai - n_al0);
            const double um[3] = {sh.c_um[c][0], sh.c_um[c][1], sh.c_um[c][2]};
            const double dt = div_y(sh.c_tau[c] * (double)k, ctx.cn, ctx.cn_y);
            double xt[6];
            state_transit(ctx, sh.cur_state, xt, um, dt);
            if (!check_state(ctx, xt, xt + 3, 1.5)) sh.c_leader[c] = -1; // (c_leader doubles as the collision flag until phase 3 has read it)
        }
        __syncthreads();
        APROF(4);
        // phase 3, thread = primitive: cost and heuristic of the survivors
        if (lane < MAX_CAND) { // (all 128 lanes of the first two wavefronts: the survivor masks below are ballots)
            const int c = lane;
            int surv = c < n_cand ? sh.c_surv[c] : 0;
            if (surv && sh.c_leader[c] == -1) surv = 0;
            if (surv && !HEUR_BESIDE) cost_and_heuristic(c);
            if (c < n_cand) sh.c_surv[c] = surv;
            static_assert(MAX_CAND == 128, "two 64-bit survivor masks");
            const unsigned long long sb = __ballot(surv != 0); // (n_cand <= 128 <= NT: c = lane, wavefront w holds the primitives 64 w ..)
            if (c < MAX_CAND && (c & 63) == 0) sh.alive[c >> 6] = sb;
        }
        __syncthreads();
        APROF(5);
        // ---- the first survivor of every voxel (tmp_expand_nodes' lookup, kinodynamic_astar.cpp:210-230): every survivor enters
        // its voxel into a 256-slot table in LDS (compare-and-swap on the key, minimum on the candidate index) and reads the
        // minimum back -- instead of scanning the candidates before it (~64 LDS reads and compares per thread)
        int vslot = -1;
        for (int c = lane; c < n_cand; c += NT) {
            if (!sh.c_surv[c]) continue;
            const unsigned long long key = (unsigned long long)sh.c_key[c];
            for (unsigned h = (unsigned)mix64(key) & 255u;; h = (h + 1) & 255u) {
                const unsigned long long old = atomicCAS(&sh.vx_key[h], ~0ull, key);
                if (old == ~0ull || old == key) { atomicMin(&sh.vx_min[h], c); vslot = (int)h; break; }
            }
        }
        __syncthreads();
        for (int c = lane; c < n_cand; c += NT) sh.c_leader[c] = vslot >= 0 ? sh.vx_min[vslot] : c;
        __syncthreads();
        if (vslot >= 0) { sh.vx_key[vslot] = ~0ull; sh.vx_min[vslot] = 0x7fffffff; } // (cleared by its users for the next expansion; ordered by the barriers of the commit)
        APROF(6);
        // ---- commit in input order (kinodynamic_astar.cpp:232-278).  Thread 0 walks the survivors and takes the decisions that
        // depend on their order -- which primitive a node keeps, node numbers, heap pushes, keys changed in place --; the node
        // records themselves are written afterwards, one thread per touched node, from the winning primitive.
        // The voxel hash is only read in phase 1, so its inserts -- a chain of dependent probes in HBM per new node, like the heap
        // pushes -- run beside thread 0's loop on the first lane of the second wavefront (node numbers are the running count).
        int use_new = 0, hs_new = 0;
        bool out_of_memory = false;
        // (both walks visit the SURVIVORS only, by the bit masks phase 3 left: a typical expansion has ~17 of them among the 125
        // primitives -- scanning all of them cost the walking lane 29-35 k cycles of dependent LDS round trips.  And what the walk
        // needs of a survivor -- leader, open node, g, f -- is read by the whole wavefront at once, lane = primitive (two per lane),
        // and handed to the walk through v_readlane: the decisions are wave-uniform, lane 0 of the wavefront does the memory side.)
        const int wv = lane >> 6, ln = lane & 63;
        if (wv == 1) { // second wavefront: the voxel hash of the new nodes (node numbers are the running count)
            const int le0 = sh.c_leader[ln], le1 = sh.c_leader[64 + ln], pr0 = sh.c_pre[ln], pr1 = sh.c_pre[64 + ln];
            int use = sh.use_node_num;
            bool full = false;
            for (int w = 0; w < 2 && !full; w++) {
                const unsigned long long mw = sh.alive[w];
                unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mw >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mw);
                for (; m && !full; m &= m - 1) {
                    const int cl = __builtin_ctzll(m), c = 64 * w + cl;
                    const int L = __builtin_amdgcn_readlane(w ? le1 : le0, cl), pre = __builtin_amdgcn_readlane(w ? pr1 : pr0, cl);
                    if (pre >= 0 || L != c) continue;
                    if (ln == 0) hash_insert(hash, a.hcap, sh.c_key[c], use);
                    if (++use == A) full = true;
                }
            }
        }
        if (wv == 0) {
            const int le0 = sh.c_leader[ln], le1 = sh.c_leader[64 + ln], pr0 = sh.c_pre[ln], pr1 = sh.c_pre[64 + ln];
            const double g0 = sh.c_g[ln], g1 = sh.c_g[64 + ln], f0 = sh.c_f[ln], f1 = sh.c_f[64 + ln];
            // g of the open nodes the survivors may re-parent: fetched by all lanes at once (one trip to the L2 instead of one per
            // survivor on the walking lane).  Nothing writes a node's g before the walk is over, and voxel groups are disjoint.
            const bool od0 = pr0 >= 0 && ln < n_cand && sh.c_surv[ln], od1 = pr1 >= 0 && 64 + ln < n_cand && sh.c_surv[64 + ln];
            const double og0 = od0 ? nodes[pr0].g : 0.0, og1 = od1 ? nodes[pr1].g : 0.0;
            int use = sh.use_node_num, hs = sh.heap_size;
            // Round 5: the heap side of the walk without a trip to the L2 per push / per changed key.  The pushes of an expansion go to
            // the consecutive holes hs, hs + 1, ...: the union of their ancestor chains is small (level j above n holes: n / 2^j + 2
            // entries), so it is fetched ONCE, lane = (level, offset), together with the heap positions of the open nodes the survivors
            // may re-key; the walk then replays std::__push_heap on that copy -- every store it makes to the heap goes to memory (lane 0,
            // nothing waits for it) AND to the lanes that hold that heap index (a compare-select; an index cached at two levels, when the
            // holes straddle a depth boundary, stays consistent that way), and every entry it moves updates the tracked position of
            // the node it belongs to.  Up to 16 new nodes and a heap of at least 16 entries (no new hole is another's ancestor);
            // anything else takes the per-push path (heap_push_hole_wave, one trip per push).
            const bool nw0 = ln < n_cand && sh.c_surv[ln] && pr0 < 0 && le0 == ln, nw1 = 64 + ln < n_cand && sh.c_surv[64 + ln] && pr1 < 0 && le1 == 64 + ln;
            const int n_new = __builtin_popcountll(__ballot(nw0)) + __builtin_popcountll(__ballot(nw1));
            const int h0 = __builtin_amdgcn_readfirstlane(hs);
            const bool fast = n_new >= 1 && n_new <= 16 && h0 >= 16 && h0 + n_new < (1 << 24) && !FRP_ASTAR_SLOW_PUSH;
            // lane -> (level, offset): level 1: 10 slots, 2: 6, 3: 4, 4: 3, 5..24: 2 each (63 lanes)
            const int lv = ln < 10 ? 1 : ln < 16 ? 2 : ln < 20 ? 3 : ln < 23 ? 4 : 5 + ((ln - 23) >> 1);
            const int lo_ = ln < 10 ? ln : ln < 16 ? ln - 10 : ln < 20 ? ln - 16 : ln < 23 ? ln - 20 : ((ln - 23) & 1);
            int cidx = -1;
            if (fast && ln < 63) {
                const int lo = ((h0 + 1) >> lv) - 1, hi = ((h0 + n_new) >> lv) - 1;
                if (lo + lo_ >= 0 && lo + lo_ <= hi) cidx = lo + lo_;
            }
            double cf = cidx >= 0 ? heap[cidx].f : 0.0;
            int cid = cidx >= 0 ? heap[cidx].id : -1;
            // tracked heap positions: of the open nodes the survivors may re-key (lane = candidate) and of the nodes created here (lane = leader)
            int hp0 = od0 ? nodes[pr0].heap_pos : -1, hp1 = od1 ? nodes[pr1].heap_pos : -1;
            int np0 = -1, np1 = -1; // ids of the nodes created in this expansion, lane = their leader (cr0 / cr1 hold open nodes' ids too)
            auto slot_of = [&](int j, int idx) { // wave-uniform: the lane that caches heap index idx as a level-j ancestor
                const int lo = ((h0 + 1) >> j) - 1;
                const int base = j == 1 ? 0 : j == 2 ? 10 : j == 3 ? 16 : j == 4 ? 20 : 23 + 2 * (j - 5);
                return base + idx - lo;
            };
            auto moved = [&](int pid, int hole) { // entry of node pid now sits at heap index hole
                hp0 = (od0 && pr0 == pid) ? hole : hp0; hp1 = (od1 && pr1 == pid) ? hole : hp1;
                hp0 = (np0 == pid) ? hole : hp0; hp1 = (np1 == pid) ? hole : hp1;
            };
            auto rekey = [&](int pos, double f) { // heap[pos].f = f, in memory and in the copy
                if (ln == 0) heap[pos].f = f;
                cf = cidx == pos ? f : cf;
            };
            auto push_fast = [&](int h, double vf, int vid) {
                int hole = h, j = 1;
                while (hole > 0) {
                    const int idx = (hole - 1) >> 1;
                    const int sl = slot_of(j, idx);
                    const double pf = rl_f64(cf, sl);
                    if (!(pf > vf)) break;
                    const int pid = __builtin_amdgcn_readlane(cid, sl);
                    if (ln == 0) { heap[hole].f = pf; heap[hole].id = pid; nodes[pid].heap_pos = hole; }
                    const bool up = cidx == hole;
                    cf = up ? pf : cf; cid = up ? pid : cid;
                    moved(pid, hole);
                    hole = idx; j++;
                }
                if (ln == 0) { heap[hole].f = vf; heap[hole].id = vid; nodes[vid].heap_pos = hole; }
                const bool up = cidx == hole;
                cf = up ? vf : cf; cid = up ? vid : cid;
                return hole;
            };
            APROF(8);
            // the walk's own bookkeeping -- value, node and winning primitive of every voxel group -- lives in registers, lane = primitive
            // (two per lane), read with v_readlane and written with a compare-select: an LDS array would cost the walking lane a
            // round trip per access (the lesson of the survivor walk itself); the arrays the node writes below read are stored once, after it
            double gv0 = 0.0, gv1 = 0.0;
            int cr0 = 0, cr1 = 0, wn0 = -1, wn1 = -1;
            auto get_gv = [&](int i) { return i < 64 ? rl_f64(gv0, i) : rl_f64(gv1, i - 64); };
            auto get_cr = [&](int i) { return i < 64 ? __builtin_amdgcn_readlane(cr0, i) : __builtin_amdgcn_readlane(cr1, i - 64); };
            auto set_grp = [&](int i, double v, int win) { // group i: value and winner
                if (i < 64) { gv0 = ln == i ? v : gv0; wn0 = ln == i ? win : wn0; }
                else { gv1 = ln == i - 64 ? v : gv1; wn1 = ln == i - 64 ? win : wn1; }
            };
            auto set_cr = [&](int i, int nid) {
                if (i < 64) cr0 = ln == i ? nid : cr0;
                else cr1 = ln == i - 64 ? nid : cr1;
            };
            for (int w = 0; w < 2; w++) {
                const unsigned long long mw = sh.alive[w];
                unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mw >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mw);
                for (; m && !out_of_memory; m &= m - 1) {
                    const int cl = __builtin_ctzll(m), c = 64 * w + cl;
                    const int L = __builtin_amdgcn_readlane(w ? le1 : le0, cl), pre = __builtin_amdgcn_readlane(w ? pr1 : pr0, cl);
                    const double g = rl_f64(w ? g1 : g0, cl), f = rl_f64(w ? f1 : f0, cl), og = rl_f64(w ? og1 : og0, cl);
                    // (winner of the voxel's group: the primitive whose values the node ends up with; -1 until one is better than the open node)
                    if (pre >= 0) { // a node of this voxel is in the open set: keep the cheaper way to it
                        const int nid = pre;
                        if (L == c) { set_grp(c, og, -1); set_cr(c, nid); }
                        if (g < get_gv(L)) { // the key changes in place: no re-heapify (as in the reference)
                            if (fast) rekey(__builtin_amdgcn_readlane(w ? hp1 : hp0, cl), f);
                            else if (ln == 0) heap[nodes[nid].heap_pos].f = f;
                            set_grp(L, g, c);
                        }
                    } else if (L == c) { // new node
                        const int nid = use;
                        hs++;
                        if (fast) {
                            if (w) np1 = ln == cl ? nid : np1; else np0 = ln == cl ? nid : np0;
                            const int at = push_fast(hs - 1, f, nid);
                            if (w) hp1 = ln == cl ? at : hp1; else hp0 = ln == cl ? at : hp0;
                        } else heap_push_hole_wave(heap, nodes, hs - 1, f, nid); // (the whole wavefront: see there)
                        set_cr(c, nid); set_grp(c, f, c);
                        use++;
                        if (use == A) out_of_memory = true; // "run out of memory", kinodynamic_astar.cpp:255-259
                    } else if (f < get_gv(L)) { // a node of this voxel was created earlier in this expansion: keep the lower f
                        const int nl = get_cr(L);
                        if (fast) rekey(L < 64 ? __builtin_amdgcn_readlane(hp0, L) : __builtin_amdgcn_readlane(hp1, L - 64), f);
                        else if (ln == 0) heap[nodes[nl].heap_pos].f = f;
                        set_grp(L, f, c);
                    }
                }
            }
            APROF(9);
            sh.c_created[ln] = cr0; sh.c_created[64 + ln] = cr1; sh.c_winner[ln] = wn0; sh.c_winner[64 + ln] = wn1;
            use_new = use; hs_new = hs;
        }
        __syncthreads();
        APROF(10);
        if (lane == 0) { // (written back behind the barrier: the other lane reads the node count while this one loops)
            sh.use_node_num = use_new; sh.heap_size = hs_new;
            if (out_of_memory) { sh.status = FRP_ASTAR_NO_PATH; sh.terminate = -1; sh.heap_size = -1; }
        }
        for (int c = lane; c < n_cand; c += NT) {
            if (!sh.c_surv[c] || sh.c_leader[c] != c || sh.c_winner[c] < 0) continue;
            const int w = sh.c_winner[c], nid = sh.c_created[c];
            Node *nd = nodes + nid;
#pragma unroll
            for (int i = 0; i < 6; i++) nd->state[i] = sh.c_state[w][i];
            nd->g = sh.c_g[w]; nd->f = sh.c_f[w];
#pragma unroll
            for (int i = 0; i < 3; i++) nd->input[i] = sh.c_um[w][i];
            nd->duration = sh.c_tau[w];
            if (sh.c_pre[c] < 0) { nd->key = sh.c_key[c]; nd->node_state = IN_OPEN_SET; nd->parent = cur; }
            else nd->parent = cur; // (an open node that found a cheaper parent, :263-272)
        }
        // (the planner's pool, heap and hash are touched by this wavefront only: the lanes of one CU share its L1, so the
        //  workgroup-scope ordering of the barrier is all the other lanes need to see lane 0's stores)
        __syncthreads();
        APROF(7);
        if (sh.heap_size < 0) break;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void pack_map_kernel(const unsigned char *occ, int gx, int gy, int gz, unsigned long long *packed)
{
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= (size_t)gx * gy) return;
    const unsigned char *c = occ + col * gz;
    unsigned long long w = 0;
    for (int z = 0; z < gz; z++) w |= (unsigned long long)(c[z] ? 1 : 0) << z;
    packed[col] = w;
}

__global__ __launch_bounds__(NT) void astar_kernel(Args a)
{
    __shared__ Shared sh;
    const frp_nmpc_astar *P = &a.p;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = P->allocate_num;
    if (P->active && !P->active[b]) return; // this planner keeps its path (no replan requested)
    Ctx ctx;
    ctx.P = P; ctx.occ = P->occ; ctx.packed = a.packed; ctx.win = sh.win; ctx.wx0 = 0; ctx.wy0 = 0; ctx.use_win = 0;
    ctx.res_inv = 1.0 / P->resolution;
    ctx.res_y = div_rcp(P->resolution); ctx.cn = (double)P->check_num; ctx.cn_y = div_rcp(ctx.cn);
    ctx.fxn = ctx.fyn = ctx.fzn = 0; ctx.fx0 = ctx.fy0 = ctx.fz0 = ctx.fwx = ctx.fwy = ctx.fwz = 0;
    {   // reach of a primitive per axis: |v| tau + (max_acc + |f_ext|) tau^2 / 2 with |v| <= max_vel (nodes beyond it are never
        // created), plus the inflated ego radius of checkState and two cells of slack
        const double tau = fmax(P->max_tau, P->init_max_tau);
        double fm = fmax(fabs(P->external_acc[3 * b]), fabs(P->external_acc[3 * b + 1]));
        if (!(fm == fm)) fm = 0.0;
        const double reach = P->max_vel * tau + 0.5 * (P->max_acc + fm) * tau * tau + 1.5 * P->ego_r;
        int r = (int)ceil(reach / P->resolution) + 2;
        ctx.win_r = r < 1 ? 1 : (r > (WIN - 1) / 2 ? (WIN - 1) / 2 : r);
    }
    ctx.fast = a.fast; ctx.off[0] = a.off[0]; ctx.off[1] = a.off[1]; ctx.off[2] = a.off[2];
    ctx.use_local = P->local_box ? 1 : 0;
    for (int i = 0; i < 3; i++) {
        ctx.lmin[i] = P->local_box ? P->local_box[6 * b + i] : 0;
        ctx.lmax[i] = P->local_box ? P->local_box[6 * b + 3 + i] : 0;
        ctx.ext[i] = P->external_acc[3 * b + i];
    }
    if (lane < 11) sh.prof[lane] = 0;
    for (int i = lane; i < 256; i += NT) { sh.vx_key[i] = ~0ull; sh.vx_min[i] = 0x7fffffff; } // (every expansion leaves the table empty again)
    if (lane == 0) {
        // the primitive lists, by the reference's own loops (kinodynamic_astar.cpp:119-137): repeated addition, the same comparisons
        const double res = 1 / 2.0, time_res = 1 / 1.0, time_res_init = 1 / 8.0;
        int n = 0;
        for (double tau = time_res_init * P->init_max_tau; tau <= P->init_max_tau; tau += time_res_init * P->init_max_tau)
            if (n < 16) sh.dur_init[n++] = tau;
        sh.n_dur_init = n;
        n = 0;
        for (double ax = -P->max_acc; ax <= P->max_acc + 1e-3; ax += P->max_acc * res)
            for (double ay = -P->max_acc; ay <= P->max_acc + 1e-3; ay += P->max_acc * res)
                for (double az = -P->max_acc; az <= P->max_acc + 1e-3; az += P->max_acc * res)
                    if (n < MAX_CAND) { sh.in_list[n][0] = ax; sh.in_list[n][1] = ay; sh.in_list[n][2] = az; n++; }
        sh.n_in = n;
        sh.dur_full = time_res * P->max_tau; // (time_res = 1: the loop over durations has exactly one pass)
    }
    __syncthreads();
    Node *nodes = a.nodes + (size_t)b * A;
    int retried = 0;
    for (int attempt = 0; attempt < 2; attempt++) { // the second pass is the retry with the discontinuous initial state (nmpc_solver.cpp:190-207)
        run_search(a, sh, ctx, b, attempt == 0 && P->init_search != 0, attempt == 1);
        if (!(sh.status == FRP_ASTAR_NO_PATH && P->init_search && attempt == 0)) break;
        retried = 1;
        __syncthreads();
    }
    // ---- retrievePath (:308-320)
    if (lane == 0) {
        int n = 0;
        int n_full = 0;
        if (sh.status != FRP_ASTAR_NO_PATH && sh.terminate >= 0) {
            for (int c = sh.terminate; c >= 0; c = nodes[c].parent) n++;
            n_full = n;
            if (n > MAX_PATH) {
                // a path of more nodes than path_ids holds would lose its ROOT (the walk starts at the terminate node): it is not
                // returned at all -- NO_PATH, the planner keeps the path it has (nmpc_solver.cpp:209-213) -- and stats[3] = -(nodes)
                n = 0;
                sh.status = FRP_ASTAR_NO_PATH;
            } else {
                int c = sh.terminate;
                for (int q = n - 1; q >= 0; q--, c = nodes[c].parent) sh.path_ids[q] = c;
            }
        }
        sh.n_path = n;
        P->status[b] = sh.status;
        if (P->stats) { P->stats[4 * b] = sh.use_node_num; P->stats[4 * b + 1] = sh.iter_num; P->stats[4 * b + 2] = retried; P->stats[4 * b + 3] = n_full > MAX_PATH ? -n_full : n; }
    }
    __syncthreads();
    const int status = sh.status;
    const int n_path = sh.n_path;
    if (P->path_nodes) {
        double *o = P->path_nodes + (size_t)b * MAX_PATH * 11;
        for (int q = lane; q < n_path; q += NT) {
            const int c = sh.path_ids[q];
            for (int i = 0; i < 6; i++) o[q * 11 + i] = nodes[c].state[i];
            for (int i = 0; i < 3; i++) o[q * 11 + 6 + i] = nodes[c].input[i];
            o[q * 11 + 9] = nodes[c].duration;
            o[q * 11 + 10] = (double)c;
        }
    }
#ifdef FRP_ASTAR_PROFILE
    if (P->path_nodes && lane < 11) P->path_nodes[((size_t)b * MAX_PATH + MAX_PATH - 1) * 11 + lane] = (double)sh.prof[lane];
#endif
    // ---- getKinoTraj(Ts) (:648-695): the search part is generated backwards in the reference and reversed; here the samples
    // are counted first so that each one lands at its forward position (more samples than K: the first K are kept, stats[3] < 0)
    if (lane == 0) {
        const double delta_t = P->Ts;
        double *out = P->kino_path + (size_t)b * P->K * 3;
        int total = 0;
        for (int q = n_path - 1; q >= 1; q--) {
            const double duration = nodes[sh.path_ids[q]].duration;
            for (double t = duration; t >= -1e-5; t -= delta_t) total++;
        }
        int w = 0; // index in the reference's backward order
        for (int q = n_path - 1; q >= 1; q--) {
            const int c = sh.path_ids[q], pc = sh.path_ids[q - 1];
            double x0[6], ut[3];
            for (int i = 0; i < 6; i++) x0[i] = nodes[pc].state[i];
            for (int i = 0; i < 3; i++) ut[i] = nodes[c].input[i];
            const double duration = nodes[c].duration;
            for (double t = duration; t >= -1e-5; t -= delta_t) {
                double xt[6];
                state_transit(ctx, x0, xt, ut, t);
                const int pos = total - 1 - w;
                if (pos < P->K && status != FRP_ASTAR_NO_PATH) { out[3 * pos] = xt[0]; out[3 * pos + 1] = xt[1]; out[3 * pos + 2] = xt[2]; }
                w++;
            }
        }
        int n = total;
        if (status != FRP_ASTAR_NO_PATH && sh.is_shot_succ) {
            double last[3] = {0.0, 0.0, 0.0};
            if (n >= 1 && n <= P->K) { last[0] = out[3 * (n - 1)]; last[1] = out[3 * (n - 1) + 1]; last[2] = out[3 * (n - 1) + 2]; }
            bool have_last = n >= 1 && n <= P->K;
            for (double t = delta_t; t <= sh.t_shot; t += delta_t) {
                const double tt[4] = {1.0, t, t * t, t * t * t};
                double coord[3];
                for (int dim = 0; dim < 3; dim++) {
                    const double *cc = sh.coef_shot + dim * 4;
                    coord[dim] = ((cc[0] * tt[0] + cc[1] * tt[1]) + cc[2] * tt[2]) + cc[3] * tt[3];
                }
                bool differs = true;
                if (have_last) {
                    const double dx = last[0] - coord[0], dy = last[1] - coord[1], dz = last[2] - coord[2];
                    differs = sqrt(dx * dx + dy * dy + dz * dz) > 0.0;
                }
                if (n < 1 || differs) {
                    if (n < P->K) { out[3 * n] = coord[0]; out[3 * n + 1] = coord[1]; out[3 * n + 2] = coord[2]; last[0] = coord[0]; last[1] = coord[1]; last[2] = coord[2]; have_last = true; }
                    else have_last = false;
                    n++;
                }
            }
        }
        // (no path: getKinoPath returns before kino_path_ is assigned, nmpc_solver.cpp:195-198 -- the planner keeps what it had)
        if (status != FRP_ASTAR_NO_PATH) P->kino_size[b] = n < P->K ? n : P->K;
        if (P->stats && n > P->K) P->stats[4 * b + 3] = -sh.n_path; // truncated: more samples than K
    }
}

} // namespace astar
} // namespace frp

extern "C" {

size_t frp_nmpc_astar_workspace_bytes(const frp_nmpc_astar *p)
{
    if (!p || p->B <= 0 || p->allocate_num <= 1) return 0;
    int hcap = 1;
    while (hcap < 2 * p->allocate_num) hcap <<= 1;
    const size_t cols = (size_t)p->grid[0] * p->grid[1];
    return cols * sizeof(unsigned long long) + 256 +
           (size_t)p->B * ((size_t)p->allocate_num * (sizeof(frp::astar::Node) + sizeof(frp::astar::HeapEnt)) + (size_t)hcap * sizeof(frp::astar::HashEnt));
}

int frp_nmpc_astar_batch(const frp_nmpc_astar *p, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace frp::astar;
    if (!p || p->B <= 0 || !p->occ || !p->start_pt || !p->start_vel || !p->start_acc || !p->end_pt || !p->end_vel || !p->external_acc ||
        !p->kino_path || !p->kino_size || !p->status || !workspace)
        return FRP_ERR_ARG;
    if (p->grid[0] <= 0 || p->grid[1] <= 0 || p->grid[2] <= 0 || !(p->resolution > 0.0) || p->allocate_num < 2 || p->check_num < 1 || p->K < 1 ||
        !(p->max_tau > 0.0) || !(p->init_max_tau > 0.0) || !(p->max_acc > 0.0) || !(p->max_vel > 0.0) || !(p->Ts > 0.0))
        return FRP_ERR_ARG;
    { // the input grid must fit the per-expansion candidate slots (125 in the reference's configuration)
        int n1 = 0;
        for (double ax = -p->max_acc; ax <= p->max_acc + 1e-3; ax += p->max_acc * 0.5) n1++;
        if (n1 * n1 * n1 > MAX_CAND) return FRP_ERR_ARG;
    }
    if (workspace_bytes < frp_nmpc_astar_workspace_bytes(p)) return FRP_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Args a;
    a.p = *p;
    int hcap = 1;
    while (hcap < 2 * p->allocate_num) hcap <<= 1;
    a.hcap = hcap;
    const size_t cols = (size_t)p->grid[0] * p->grid[1];
    char *w = static_cast<char *>(workspace);
    unsigned long long *packed = reinterpret_cast<unsigned long long *>(w);
    w += (cols * sizeof(unsigned long long) + 255) / 256 * 256;
    a.nodes = reinterpret_cast<Node *>(w); w += (size_t)p->B * p->allocate_num * sizeof(Node);
    a.heap = reinterpret_cast<HeapEnt *>(w); w += (size_t)p->B * p->allocate_num * sizeof(HeapEnt);
    a.hash = reinterpret_cast<HashEnt *>(w);
    // getlineGrids hands cell centres (c res + res / 2) to getVoxelState, which floors ((centre - origin) / res) again: when that
    // round trip is c + off for every cell index a ray can reach, the kernel adds the offset instead (same results by this check)
    a.fast = 1;
    {
        const double res = p->resolution, res_inv = 1.0 / p->resolution;
        for (int i = 0; i < 3; i++) {
            a.off[i] = (int)std::floor(((0.0 * res + res / 2.0) - p->origin[i]) * res_inv);
            for (int c = -CELL_SAFE; c <= CELL_SAFE && a.fast; c++)
                if ((int)std::floor((((double)c * res + res / 2.0) - p->origin[i]) * res_inv) != c + a.off[i]) a.fast = 0;
        }
    }
    a.packed = nullptr;
    if (p->grid[2] <= 64) { // one 64-bit word per (x, y) column
        hipLaunchKernelGGL(pack_map_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, st, p->occ, p->grid[0], p->grid[1], p->grid[2], packed);
        a.packed = packed;
    }
    hipLaunchKernelGGL(astar_kernel, dim3(p->B), dim3(NT), 0, st, a);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

} // extern "C"
