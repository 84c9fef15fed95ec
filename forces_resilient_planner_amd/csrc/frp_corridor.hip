// frp_corridor.hip -- SURVEY 8f row f-3: corridor generation and selection (NMPCSolver::getSikangConst over the
// horizon, nmpc_solver.cpp:288-332, on top of DecompROS' EllipsoidDecomp3D::dilate) batched on the device.  Inputs
// are the stage references (row f-4 / the caller), the tube matrices of frp_nmpc_tube_batch (f-2) and the obstacle
// cloud; outputs are exactly the polytope inputs of frp_nmpc_pack_batch (f-1): poly_A, poly_b, poly_nfaces,
// poly_index.
//
// Reference (src/ThirdParty/DecompROS/decomp_ros_utils/include/ unless noted):
//   getSikangConst                   plan_manage/src/nmpc_solver.cpp:288-332
//   EllipsoidDecomp3D::dilate / get_constraints   decomp_util/ellipsoid_decomp.h:47-90
//   DecompBase::set_obs / find_polyhedron         decomp_util/decomp_base.h:33-38, 63-83
//   LineSegment::dilate / add_local_bbox / find_ellipsoid (3D)   decomp_util/line_segment.h:31-35, 47-85, 136-211
//   Ellipsoid::dist / closest_point / closest_hyperplane         decomp_geometry/ellipsoid.h:19-58
//   Polyhedron::inside, LinearConstraint(p0, planes)             decomp_geometry/polyhedron.h:51-58, 98-118
//   vec3_to_rotation                 decomp_geometry/geometric_utils.h:27-35;  epsilon_  decomp_basis/data_type.h:129
//
// Mapping.  One 256-thread workgroup per planner walks the stages in order (a stage reuses the polytope made for an
// earlier stage while its inflated tube ellipsoid fits, so the stage loop is sequential by definition).  A
// decomposition is a sequence of scans -- "keep the points that ..., and find the one closest to the ellipsoid
// centre in the ellipsoid's metric" -- where the reference rebuilds std::vectors:
//   * the first scan reads the whole cloud once (CR_UNROLL 64-point words in flight per wave) -- or, when the caller
//     supplies the uniform grid of frp_nmpc_cloud_grid_build, only the cell rows under the box's axis-aligned hull
//     (scan_grid, a separate kernel instantiation) --, tests the local box in the box's own frame, and appends the indices of the in-box points to a dense list in LDS (one LDS atomic
//     per CR_UNROLL words; the order of the list is irrelevant because minima are tie-broken by cloud index, which
//     is the reference's first-minimum rule on its order-preserving lists);
//   * every later scan runs over that list (typically 10 % of the cloud) with all lanes busy; point lists are bit
//     masks over list positions, one 64-bit word per 64 positions, produced by wave ballots; word g is always read
//     and written by the same wave, empty words are skipped 64 at a time with a ballot.  Boxes holding more than
//     CR_LIST points fall back to masks over the cloud itself;
//   * "filter with the new ellipsoid, then take the closest of what is left" uses the same distances, so both happen
//     in ONE scan; the workgroup minimum carries the winner's coordinates (one barrier, double-buffered);
//   * the 3x3 algebra of an ellipsoid update is wave-uniform: thread 0 does it and publishes the result through LDS
//     (struct Uni), so it costs the scanning waves no registers; the hyperplane of a cut is a dozen operations and
//     is computed by every lane from the winner the reduction hands out (no publish, no extra barrier).
// Memory-side work: 24 bytes per cloud point per decomposition, then 24-byte gathers of in-box points from L2 (the
// cloud is shared by the planners of a fleet); lists of up to CR_TILE * 256 points live in registers (scan_tile).
// Measured (profiles/r01_corridor_bench.json): first scan + list + register-tile fill 32 us on the plain cloud, 16 us
// through the uniform grid of frp_nmpc_cloud_grid_build (scan_grid), then ~20-30 scans of ~1.9 us per decomposition.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/frp_nmpc.h"

// Which point a round picks is decided by comparisons of nearly equal distances (after find_ellipsoid the obstacle points that shaped the
// ellipsoid sit at distance 1 +- 1 ulp), so every kernel of this file must compute the SAME bits from the same inputs.  Left to the
// compiler, a * b + c is fused or not depending on the code around it (round 5: the shell kernel visited two touching points in the other
// order -- same rows, swapped); hence no implicit contraction anywhere in this file, and the per-point expressions (metric distance,
// box frame projection, side of a cut) written once, with their fused multiply-adds spelled out.
#pragma clang fp contract(off)

namespace frp {

// a0 * x + a1 * y + a2 * z with the rounding every kernel uses
__device__ __forceinline__ double dot3(double a0, double a1, double a2, double x, double y, double z)
{
    return __builtin_fma(a2, z, __builtin_fma(a1, y, a0 * x));
}
// on which side of the cut (q, n) the point lies (decomp_base.h:74-78: kept while negative)
__device__ __forceinline__ double cut_side(const double n[3], const double q[3], double x, double y, double z)
{
    return dot3(n[0], n[1], n[2], x - q[0], y - q[1], z - q[2]);
}

#ifndef FRP_CR_WAVES
#define FRP_CR_WAVES 4
#endif
constexpr int CR_WAVES = FRP_CR_WAVES, CR_THREADS = 64 * CR_WAVES, CR_UNROLL = 4, CR_BATCH = 2;
constexpr double CR_EPS = 1e-10; // epsilon_, data_type.h:129

struct M3 { double m[9]; };

__device__ __forceinline__ M3 mul(const M3 &a, const M3 &b)
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
__device__ __forceinline__ M3 transpose(const M3 &a)
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * j + i];
    return r;
}
__device__ __forceinline__ M3 inverse(const M3 &a) // cofactors / determinant
{
    const double *m = a.m;
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double inv = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    M3 r;
    r.m[0] = c00 * inv; r.m[1] = (m[2] * m[7] - m[1] * m[8]) * inv; r.m[2] = (m[1] * m[5] - m[2] * m[4]) * inv;
    r.m[3] = c01 * inv; r.m[4] = (m[0] * m[8] - m[2] * m[6]) * inv; r.m[5] = (m[2] * m[3] - m[0] * m[5]) * inv;
    r.m[6] = c02 * inv; r.m[7] = (m[1] * m[6] - m[0] * m[7]) * inv; r.m[8] = (m[0] * m[4] - m[1] * m[3]) * inv;
    return r;
}
__device__ __forceinline__ M3 quat_to_rot(double w, double x, double y, double z)
{
    M3 r;
    r.m[0] = 1 - 2 * (y * y + z * z); r.m[1] = 2 * (x * y - w * z);     r.m[2] = 2 * (x * z + w * y);
    r.m[3] = 2 * (x * y + w * z);     r.m[4] = 1 - 2 * (x * x + z * z); r.m[5] = 2 * (y * z - w * x);
    r.m[6] = 2 * (x * z - w * y);     r.m[7] = 2 * (y * z + w * x);     r.m[8] = 1 - 2 * (x * x + y * y);
    return r;
}
__device__ __forceinline__ M3 rot_diag_rot(const M3 &R, double a0, double a1, double a2) // R diag(a) R'
{
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[3 * i + j] = R.m[3 * i] * a0 * R.m[3 * j] + R.m[3 * i + 1] * a1 * R.m[3 * j + 1] + R.m[3 * i + 2] * a2 * R.m[3 * j + 2];
    return r;
}
__device__ __forceinline__ void tmul(const M3 &R, const double v[3], double o[3]) // o = R' v
{
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = R.m[j] * v[0] + R.m[3 + j] * v[1] + R.m[6 + j] * v[2];
}
// Ellipsoid::dist (ellipsoid.h:19-21), squared, with C^-1 precomputed.  The scans are FP64-VALU-bound and a square
// root is half of their arithmetic, so it is taken only where the reference's threshold needs it (1 - dist >
// epsilon_); "dist <= 1" and the ordering of distances are the same on the squares.
__device__ __forceinline__ double ell_dist2(const M3 &Ci, const double d[3], double x, double y, double z)
{
    const double u = x - d[0], v = y - d[1], w = z - d[2];
    const double a = dot3(Ci.m[0], Ci.m[1], Ci.m[2], u, v, w), b = dot3(Ci.m[3], Ci.m[4], Ci.m[5], u, v, w), c = dot3(Ci.m[6], Ci.m[7], Ci.m[8], u, v, w);
    return dot3(a, b, c, a, b, c);
}

struct Best { double dist; int idx; double x, y, z; }; // candidate closest point: SQUARED metric distance, cloud index, coordinates

__device__ __forceinline__ bool before(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

// Wave minima on the DPP network (quad_perm, row_half_mirror, row_mirror, then v_readlane across the four rows):
// a ds_bpermute butterfly costs an LDS round trip per step, and this reduction runs once per scan.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = dpp_i32<CTRL>((int)(unsigned)b), hi = dpp_i32<CTRL>((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_min_f64(double v)
{
    v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0x140>(v));
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16 * k) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 16 * k)));
    return fmin(fmin(r[0], r[1]), fmin(r[2], r[3]));
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, dpp_i32<0xB1>(v)); v = min(v, dpp_i32<0x4E>(v)); v = min(v, dpp_i32<0x141>(v)); v = min(v, dpp_i32<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// minimum over the workgroup in (distance, index) order; idx = INT_MAX when no point was alive.  The winner's
// coordinates travel with it, so nobody has to fetch the point again.  s_red is double-buffered by `phase`: one barrier.
__device__ Best block_min(const Best &mine, Best *s_red, int &phase)
{
    const double d = wave_min_f64(mine.dist);
    const int i = wave_min_i32(mine.dist == d ? mine.idx : 0x7fffffff);
    Best *buf = s_red + phase * CR_WAVES;
    phase ^= 1;
    const int lane = threadIdx.x & 63;
    if (i == 0x7fffffff ? lane == 0 : mine.idx == i) buf[threadIdx.x >> 6] = mine; // the lane that owns the wave's minimum
    __syncthreads();
    Best r = buf[0];
#pragma unroll
    for (int w = 1; w < CR_WAVES; ++w)
        if (before(buf[w].dist, buf[w].idx, r.dist, r.idx)) r = buf[w];
    return r;
}

struct Scan {             // what a scan iterates over
    const double *pts;    // the planner's cloud [.][3]
    const uint32_t *list; // nullptr: positions are cloud indices; else positions index this list of cloud indices
    int Pn, W;            // positions, 64-position words
};

#ifndef FRP_CR_TILE
#define FRP_CR_TILE 5
#endif
#ifndef FRP_CR_WPE
#define FRP_CR_WPE 3
#endif
// Register tile of 5 words per wave (1280 points per planner) at three workgroups per CU: a decomposition is a chain of
// ~20 latency-bound scans (reduction, barrier, thread-0 algebra), so a third resident workgroup per CU is worth more than the
// 8-word tile that needs 256 VGPRs (full tick, 4096 planners: 1.32 -> 1.09 ms; 4 per CU spills too much: 1.28; two-wave
// workgroups: 1.27-1.52).
constexpr int CR_TILE = FRP_CR_TILE; // 64-position words per wave held in registers (CR_TILE * CR_THREADS points per planner)
#ifndef FRP_CR_LIST
#define FRP_CR_LIST 8192
#endif
constexpr int CR_LIST = FRP_CR_LIST; // capacity of the in-box index list (LDS); larger boxes fall back to cloud positions

// Wave-uniform state of the running decomposition.  It lives in LDS and is advanced by thread 0 only, so the 3x3
// algebra costs no registers in the scanning waves: a scan loads just the 9 + 3 (+ 6) doubles it needs.
struct Uni {
    double Ri[9], Rf[9], Ci[9], CC[9]; // initial / final ellipsoid frame, C^-1, C^-1 C^-T
    double mid[3], ax[3];              // ellipsoid centre (p1 + p2) / 2, semi-axes
    double box[12][3];                 // local box: points 0..5, outward normals 6..11
    double frame[3][3], p1[3], len;    // the same box as a frame at p1: axes dir_h, dir, dir_v; segment length
    int rows, overflow, count;         // rows emitted, > F rows seen, in-box points appended to the list
};

__device__ __forceinline__ M3 ld3(const double *p) { M3 r; for (int k = 0; k < 9; ++k) r.m[k] = p[k]; return r; }
__device__ __forceinline__ void st3(double *p, const M3 &a) { for (int k = 0; k < 9; ++k) p[k] = a.m[k]; }

#ifdef FRP_CORRIDOR_PROFILE
__device__ long long g_prof[2];
#endif

enum { KEEP_OUTSIDE = 0, KEEP_INSIDE = 1, KEEP_ALL = 2, KEEP_BEHIND_PLANE = 3 };

// One pass: out = { points of `in` that satisfy MODE }, returns the kept point closest to the centre in the metric
// of u.Ci (first minimum in list order).  Word g of a mask is always handled by wave g % CR_WAVES.
template <int MODE>
__device__ __forceinline__ Best scan(const Scan &s, const uint64_t *in, uint64_t *out, const Uni &u, Best *s_red, int &phase,
                                     const double *pq = nullptr, const double *pn = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef FRP_CORRIDOR_PROFILE
    long long ts0 = wall_clock64();
#endif
    const M3 Ci = ld3(u.Ci);
    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
    double q[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (MODE == KEEP_BEHIND_PLANE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { q[k] = pq[k]; n[k] = pn[k]; } // the hyperplane, computed by every lane from the reduction's winner
    }
    Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
    // Most words of a list are empty.  Each lane fetches one of the wave's next 64 words, a ballot tells which are
    // not, and only those are visited -- one LDS round trip per 4096 points instead of one per 64.
    for (int base = wave; base < s.W; base += CR_WAVES * 64) {
        const int gm = base + lane * CR_WAVES;
        const uint64_t wm = gm < s.W ? in[gm] : 0;
        if (out != in && gm < s.W && wm == 0) out[gm] = 0;
        uint64_t nz = __ballot(wm != 0);
        while (nz) { // up to CR_BATCH non-empty words at a time, their loads issued together
            int gs[CR_BATCH], id[CR_BATCH];
            bool al[CR_BATCH];
            double x[CR_BATCH], y[CR_BATCH], z[CR_BATCH];
#pragma unroll
            for (int k = 0; k < CR_BATCH; ++k) {
                gs[k] = -1; id[k] = 0; al[k] = false; x[k] = y[k] = z[k] = 0.0;
                if (nz) {
                    const int l = __builtin_ctzll(nz);
                    nz &= nz - 1;
                    gs[k] = base + l * CR_WAVES;
                    const uint64_t word = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(wm >> 32), l) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wm, l);
                    const int pos = gs[k] * 64 + lane;
                    al[k] = ((word >> lane) & 1) && pos < s.Pn;
                    if (al[k]) {
                        id[k] = s.list ? (int)(s.list[pos] & 0x7fffffffu) : pos;
                        x[k] = s.pts[3 * (size_t)id[k]]; y[k] = s.pts[3 * (size_t)id[k] + 1]; z[k] = s.pts[3 * (size_t)id[k] + 2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CR_BATCH; ++k) {
                if (gs[k] < 0) break;
                bool alive = al[k];
                if (alive) {
                    const double dist = ell_dist2(Ci, d, x[k], y[k], z[k]); // squared: same order, same "<= 1"
                    if (MODE == KEEP_OUTSIDE) alive = 1 - sqrt(dist) > CR_EPS;
                    if (MODE == KEEP_INSIDE) alive = dist <= 1;
                    if (MODE == KEEP_BEHIND_PLANE) alive = cut_side(n, q, x[k], y[k], z[k]) < 0;
                    if (alive && before(dist, id[k], best.dist, best.idx)) best = Best{dist, id[k], x[k], y[k], z[k]};
                }
                const uint64_t o = __ballot(alive);
                if (lane == 0) out[gs[k]] = o;
            }
        }
    }
#ifdef FRP_CORRIDOR_PROFILE
    long long ts1 = wall_clock64();
    Best r_ = block_min(best, s_red, phase);
    if (threadIdx.x == 0) { g_prof[0] += ts1 - ts0; g_prof[1] += wall_clock64() - ts1; }
    return r_;
#else
    return block_min(best, s_red, phase);
#endif
}

// The same pass when the whole list fits the wave's REGISTER TILE: up to CR_TILE words per wave (CR_TILE * 256 points
// per workgroup), whose coordinates and cloud indices were loaded once after the first scan and stay in VGPRs for
// the 20-30 scans of the decomposition -- no memory traffic at all besides the mask words.
// d2 = the squared metric distance of the point in the FINAL ellipsoid: the hyperplane loop (decomp_base.h:63-83) cuts with a fixed
// ellipsoid, so the distances its rounds compare are computed once, by the KEEP_ALL scan that opens it
struct Tile { double x[CR_TILE], y[CR_TILE], z[CR_TILE], d2[CR_TILE]; int id[CR_TILE]; };

template <int MODE>
__device__ __forceinline__ Best scan_tile(Tile &t, int W, const uint64_t *in, uint64_t *out, const Uni &u, Best *s_red, int &phase,
                                          const double *pq = nullptr, const double *pn = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const M3 Ci = ld3(u.Ci);
    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
    double q[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (MODE == KEEP_BEHIND_PLANE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { q[k] = pq[k]; n[k] = pn[k]; } // the hyperplane, computed by every lane from the reduction's winner
    }
    uint64_t w[CR_TILE];
#pragma unroll
    for (int j = 0; j < CR_TILE; ++j) { const int g = wave + j * CR_WAVES; w[j] = g < W ? in[g] : 0; }
    Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < CR_TILE; ++j) {
        const int g = wave + j * CR_WAVES;
        if (g >= W) break;
        if (w[j] == 0) { if (out != in && lane == 0) out[g] = 0; continue; }
        bool alive = (w[j] >> lane) & 1;
        if (alive) {
            // (the rounds of the hyperplane loop compare the distances its opening KEEP_ALL scan stored: same values, no recomputation)
            const double dist = MODE == KEEP_BEHIND_PLANE ? t.d2[j] : ell_dist2(Ci, d, t.x[j], t.y[j], t.z[j]);
            if (MODE == KEEP_ALL) t.d2[j] = dist;
            if (MODE == KEEP_OUTSIDE) alive = 1 - sqrt(dist) > CR_EPS;
            if (MODE == KEEP_INSIDE) alive = dist <= 1;
            if (MODE == KEEP_BEHIND_PLANE) alive = cut_side(n, q, t.x[j], t.y[j], t.z[j]) < 0;
            if (alive && before(dist, t.id[j], best.dist, best.idx)) best = Best{dist, t.id[j], t.x[j], t.y[j], t.z[j]};
        }
        const uint64_t o = __ballot(alive);
        if (lane == 0) out[g] = o;
    }
    return block_min(best, s_red, phase);
}

// first scan of a decomposition: obs_ = cloud points inside the local box (decomp_base.h:33-38) -> m0, obs = those
// inside the seed ellipsoid -> m1 and m2.  Every point is read here, so the loads of CR_UNROLL word groups are
// issued before any of them is used.
__device__ __forceinline__ Best scan_cloud(const Scan &s, uint64_t *m0, uint64_t *m1, uint64_t *m2, uint32_t *list, Uni &u, bool has_box, const double *bbox, Best *s_red, int &phase)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const M3 Ci = ld3(u.Ci);
    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
    // The six planes of add_local_bbox have unit normals +-dir_h, +-dir, +-dir_v, so signed_dist(x) > epsilon_ for
    // any of them is a bound on the coordinates of x - p1 in that frame (12 + 4 registers instead of 36).
    double fr[3][3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = u.p1[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) fr[k][j] = u.frame[k][j];
    }
    const double bh = bbox[1] + CR_EPS, bd_lo = -bbox[0] - CR_EPS, bd_hi = u.len + bbox[0] + CR_EPS, bv = bbox[2] + CR_EPS;
    Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
    for (int g0 = wave; g0 < s.W; g0 += CR_WAVES * CR_UNROLL) {
        double x[CR_UNROLL], y[CR_UNROLL], z[CR_UNROLL];
#pragma unroll
        for (int k = 0; k < CR_UNROLL; ++k) {
            const int idx = (g0 + k * CR_WAVES) * 64 + lane;
            const size_t o3 = 3 * (size_t)(idx < s.Pn ? idx : 0);
            x[k] = s.Pn ? s.pts[o3] : 0.0; y[k] = s.Pn ? s.pts[o3 + 1] : 0.0; z[k] = s.Pn ? s.pts[o3 + 2] : 0.0;
        }
        uint64_t w0[CR_UNROLL];
        bool i1[CR_UNROLL];
        int total = 0;
#pragma unroll
        for (int k = 0; k < CR_UNROLL; ++k) {
            const int g = g0 + k * CR_WAVES, idx = g * 64 + lane;
            bool in0 = g < s.W && idx < s.Pn;
            i1[k] = false;
            if (has_box) { // Polyhedron::inside: rejected if signed_dist > epsilon_ (polyhedron.h:51-58)
                const double ex = x[k] - o[0], ey = y[k] - o[1], ez = z[k] - o[2];
                const double h = dot3(fr[0][0], fr[0][1], fr[0][2], ex, ey, ez), t = dot3(fr[1][0], fr[1][1], fr[1][2], ex, ey, ez),
                             v = dot3(fr[2][0], fr[2][1], fr[2][2], ex, ey, ez);
                in0 = in0 && !(h > bh) && !(-h > bh) && !(t > bd_hi) && !(t < bd_lo) && !(v > bv) && !(-v > bv);
            }
            if (in0) {
                const double dist = ell_dist2(Ci, d, x[k], y[k], z[k]);
                i1[k] = dist <= 1;
                if (i1[k] && dist < best.dist) best = Best{dist, idx, x[k], y[k], z[k]};
            }
            w0[k] = __ballot(in0);
            const uint64_t w1 = __ballot(i1[k]);
            if (lane == 0 && g < s.W) { m0[g] = w0[k]; m1[g] = w1; m2[g] = w1; }
            total += (int)__popcll(w0[k]);
        }
        if (total) { // append the in-box points to the dense list (any order: minima are tie-broken by cloud index);
                     // one LDS atomic per CR_UNROLL words
            int at = 0;
            if (lane == 0) at = atomicAdd(&u.count, total);
            at = __builtin_amdgcn_readfirstlane(at);
#pragma unroll
            for (int k = 0; k < CR_UNROLL; ++k) {
                const int mine = at + (int)__popcll(w0[k] & ((1ull << lane) - 1));
                if (((w0[k] >> lane) & 1) && mine < CR_LIST)
                    list[mine] = (uint32_t)((g0 + k * CR_WAVES) * 64 + lane) | (i1[k] ? 0x80000000u : 0u);
                at += (int)__popcll(w0[k]);
            }
        }
    }
    return block_min(best, s_red, phase);
}

// first scan of a decomposition when the cloud comes with a uniform grid (frp_nmpc_cloud_grid_build): only the cell
// rows that meet the axis-aligned hull of the local box are read -- each row (cells ix0..ix1 of one (iy, iz)) is one
// contiguous run of the cell-sorted points, CR_GROWS rows in flight per wave.  Produces the dense list only (no cloud
// masks); minima are tie-broken by the ORIGINAL cloud index, so the result is the same as scanning the whole cloud.
constexpr int CR_GROWS = 2;
__device__ __forceinline__ Best scan_grid(const frp_nmpc_corridor &c, uint32_t *list, Uni &u, Best *s_red, int &phase)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const M3 Ci = ld3(u.Ci);
    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
    double fr[3][3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = u.p1[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) fr[k][j] = u.frame[k][j];
    }
    const double bh = c.bbox[1] + CR_EPS, bd_lo = -c.bbox[0] - CR_EPS, bd_hi = u.len + c.bbox[0] + CR_EPS, bv = c.bbox[2] + CR_EPS;
    // axis-aligned hull of the box { o + h fr0 + t fr1 + v fr2 : |h| <= bh, bd_lo <= t <= bd_hi, |v| <= bv } in cells
    int lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ctr = o[k] + 0.5 * (bd_lo + bd_hi) * fr[1][k];
        const double half = bh * fabs(fr[0][k]) + 0.5 * (bd_hi - bd_lo) * fabs(fr[1][k]) + bv * fabs(fr[2][k]);
        const double a = floor((ctr - half - c.grid_origin[k]) / c.grid_cell), b = floor((ctr + half - c.grid_origin[k]) / c.grid_cell);
        const int n = c.grid_dims[k];
        lo[k] = a < 0 ? 0 : (a > n - 1 ? n - 1 : (int)a);   // points beyond the grid were binned into its border cells
        hi[k] = b < 0 ? 0 : (b > n - 1 ? n - 1 : (int)b);
    }
    const int ny = hi[1] - lo[1] + 1, rows = ny * (hi[2] - lo[2] + 1), nx = c.grid_dims[0];
    Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
    for (int r0 = wave; r0 < rows; r0 += CR_WAVES * CR_GROWS) {
        int beg[CR_GROWS], end[CR_GROWS], most = 0;
#pragma unroll
        for (int k = 0; k < CR_GROWS; ++k) {
            const int r = r0 + k * CR_WAVES;
            beg[k] = end[k] = 0;
            if (r < rows) {
                const size_t row = ((size_t)(lo[2] + r / ny) * c.grid_dims[1] + (lo[1] + r % ny)) * nx;
                beg[k] = c.grid_start[row + lo[0]];
                end[k] = c.grid_start[row + hi[0] + 1];
            }
            most = max(most, end[k] - beg[k]);
        }
        for (int off = 0; off < most; off += 64) { // usually one trip: a row of cells holds a few dozen points
            double x[CR_GROWS], y[CR_GROWS], z[CR_GROWS];
            int id[CR_GROWS];
#pragma unroll
            for (int k = 0; k < CR_GROWS; ++k) {
                const int p = beg[k] + off + lane;
                const bool ok = p < end[k];
                const size_t p3 = 3 * (size_t)(ok ? p : 0);
                x[k] = ok ? c.grid_points[p3] : 0.0; y[k] = ok ? c.grid_points[p3 + 1] : 0.0; z[k] = ok ? c.grid_points[p3 + 2] : 0.0;
                id[k] = ok ? c.grid_index[p] : -1;
            }
            uint64_t w0[CR_GROWS];
            bool i1[CR_GROWS];
            int total = 0;
#pragma unroll
            for (int k = 0; k < CR_GROWS; ++k) {
                bool in0 = id[k] >= 0;
                i1[k] = false;
                const double ex = x[k] - o[0], ey = y[k] - o[1], ez = z[k] - o[2];
                const double h = dot3(fr[0][0], fr[0][1], fr[0][2], ex, ey, ez), t = dot3(fr[1][0], fr[1][1], fr[1][2], ex, ey, ez),
                             v = dot3(fr[2][0], fr[2][1], fr[2][2], ex, ey, ez);
                in0 = in0 && !(h > bh) && !(-h > bh) && !(t > bd_hi) && !(t < bd_lo) && !(v > bv) && !(-v > bv);
                if (in0) {
                    const double dist = ell_dist2(Ci, d, x[k], y[k], z[k]);
                    i1[k] = dist <= 1;
                    if (i1[k] && before(dist, id[k], best.dist, best.idx)) best = Best{dist, id[k], x[k], y[k], z[k]};
                }
                w0[k] = __ballot(in0);
                total += (int)__popcll(w0[k]);
            }
            if (total) {
                int at = 0;
                if (lane == 0) at = atomicAdd(&u.count, total);
                at = __builtin_amdgcn_readfirstlane(at);
#pragma unroll
                for (int k = 0; k < CR_GROWS; ++k) {
                    const int mine = at + (int)__popcll(w0[k] & ((1ull << lane) - 1));
                    if (((w0[k] >> lane) & 1) && mine < CR_LIST) list[mine] = (uint32_t)id[k] | (i1[k] ? 0x80000000u : 0u);
                    at += (int)__popcll(w0[k]);
                }
            }
        }
    }
    return block_min(best, s_red, phase);
}

// LinearConstraint row of hyperplane (q, n) seen from the seed centre (polyhedron.h:98-118); thread 0 only
__device__ void emit_row(Uni &u, const double q[3], const double n_[3], int F, double *s_A, double *s_b, double *gA, double *gb)
{
    double n[3] = {n_[0], n_[1], n_[2]};
    double cc = q[0] * n[0] + q[1] * n[1] + q[2] * n[2];
    if (n[0] * u.mid[0] + n[1] * u.mid[1] + n[2] * u.mid[2] - cc > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; cc = -cc; }
    const int r = u.rows;
    if (r < F) {
        s_A[3 * r] = n[0]; s_A[3 * r + 1] = n[1]; s_A[3 * r + 2] = n[2]; s_b[r] = cc;
        gA[3 * r] = n[0]; gA[3 * r + 1] = n[1]; gA[3 * r + 2] = n[2]; gb[r] = cc;
    } else
        u.overflow = 1;
    u.rows = r + 1;
}

#ifdef FRP_CORRIDOR_PROFILE
#define CR_T0 long long t0_ = wall_clock64();
#define CR_ACC(v) { long long t1_ = wall_clock64(); v += t1_ - t0_; t0_ = t1_; }
#define CR_CNT(v) ++v;
#else
#define CR_CNT(v)
#define CR_T0
#define CR_ACC(v)
#endif

// GRID = true: the first scan of every decomposition goes through the uniform grid; a planner that meets a box with more
// than CR_LIST points marks itself (poly_index[b][0] = -1) and leaves, and the GRID = false kernel launched right behind
// with only_flagged = 1 redoes just those planners from the plain cloud.  Two kernels instead of one with both first
// scans inlined: the combined one needs 256 VGPRs + 100 spilled SGPRs and loses the second resident workgroup per CU.
template <bool GRID>
__global__ __launch_bounds__(CR_THREADS) __attribute__((amdgpu_waves_per_eu(FRP_CR_WPE, FRP_CR_WPE))) void corridor_kernel(frp_nmpc_corridor c, int only_flagged)
{
#ifdef FRP_CORRIDOR_PROFILE
    long long tp_check = 0, tp_init = 0, tp_cloud = 0, tp_lead = 0, tp_scan = 0, tp_emit = 0, tp_begin = wall_clock64();
    int np_scan = 0;
#endif
    extern __shared__ uint64_t s_mask[];
    __shared__ double s_A[FRP_CORRIDOR_MAX_F * 3], s_b[FRP_CORRIDOR_MAX_F];
    __shared__ Best s_red[2 * CR_WAVES];
    int phase = 0;
    __shared__ Uni u;
    const int b = blockIdx.x, tid = threadIdx.x;
    Scan sc;
    sc.Pn = c.cloud_count ? c.cloud_count[c.cloud_per_planner ? b : 0] : c.P;
    sc.Pn = sc.Pn < c.P ? sc.Pn : c.P;
    sc.W = (c.P + 63) / 64;
    sc.pts = c.cloud + (c.cloud_per_planner ? (size_t)b * c.P * 3 : 0);
    const int W_cloud = sc.W, P_cloud = sc.Pn;
    uint64_t *m0 = s_mask, *m1 = s_mask + sc.W, *m2 = s_mask + 2 * sc.W; // obs_, obs, working list
    uint32_t *list = reinterpret_cast<uint32_t *>(s_mask + 3 * sc.W);
    const double *ref = c.ref_pos + (size_t)b * c.N * 3, *yaw = c.ref_yaw + (size_t)b * c.N, *Eb = c.ellipsoid + (size_t)b * c.N * 9;
    const bool has_box = c.bbox[0] != 0.0 || c.bbox[1] != 0.0 || c.bbox[2] != 0.0;
    if (only_flagged && c.poly_index[(size_t)b * c.N] != -1) return; // (behind another kernel: only the planners it left flagged)
    int npoly = 0, rows = 0; // rows = stored rows of the last polytope (s_A / s_b)
    // Every round of the reference's while-loops removes at least the closest point, so a list is exhausted after at
    // most Pn rounds; the bound only matters for non-finite input, where the reference would spin forever.
    const int max_rounds = sc.Pn + 8;
    if (tid == 0) u.overflow = 0;

    for (int i = 0; i < c.N; ++i) {
        CR_T0
        // ---- does the stage's inflated tube ellipsoid fit the last polytope? (nmpc_solver.cpp:291-313) ----------
        if (npoly > 0) {
            int viol = 0;
            if (tid < rows) {
                const double a0 = s_A[3 * tid], a1 = s_A[3 * tid + 1], a2 = s_A[3 * tid + 2];
                const double *E = Eb + 9 * i;
                const double e0 = E[0] * a0 + E[1] * a1 + E[2] * a2, e1 = E[3] * a0 + E[4] * a1 + E[5] * a2, e2 = E[6] * a0 + E[7] * a1 + E[8] * a2;
                const double add = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
                viol = (a0 * ref[3 * i] + a1 * ref[3 * i + 1] + a2 * ref[3 * i + 2] - (s_b[tid] - c.inflation * add)) > 0;
            }
            if (!__syncthreads_or(viol)) {
                if (tid == 0) c.poly_index[(size_t)b * c.N + i] = npoly - 1;
                CR_ACC(tp_check)
                continue;
            }
        }
        CR_ACC(tp_check)
        // ---- new decomposition around the seed segment (nmpc_solver.cpp:315-329) -------------------------------
        if (tid == 0) {
            double sy, cy;
            sincos(yaw[i], &sy, &cy);
            const double p1[3] = {ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]};
            const double p2[3] = {p1[0] + c.seed_len * cy, p1[1] + c.seed_len * sy, p1[2]};
            const double dv[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            const double len = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) u.mid[k] = (p1[k] + p2[k]) / 2;
            u.len = len;
            if (has_box) { // local box planes (line_segment.h:47-85)
                const double dir[3] = {dv[0] / len, dv[1] / len, dv[2] / len};
                double dh[3] = {dir[1], -dir[0], 0.0};
                double hn = sqrt(dh[0] * dh[0] + dh[1] * dh[1]);
                if (hn == 0.0) { dh[0] = -1.0; dh[1] = 0.0; hn = 1.0; }
                dh[0] /= hn; dh[1] /= hn;
                const double dvv[3] = {dir[1] * dh[2] - dir[2] * dh[1], dir[2] * dh[0] - dir[0] * dh[2], dir[0] * dh[1] - dir[1] * dh[0]};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u.frame[0][k] = dh[k]; u.frame[1][k] = dir[k]; u.frame[2][k] = dvv[k]; u.p1[k] = p1[k];
                    u.box[0][k] = p1[k] + dh[k] * c.bbox[1];  u.box[6][k] = dh[k];
                    u.box[1][k] = p1[k] - dh[k] * c.bbox[1];  u.box[7][k] = -dh[k];
                    u.box[2][k] = p2[k] + dir[k] * c.bbox[0]; u.box[8][k] = dir[k];
                    u.box[3][k] = p1[k] - dir[k] * c.bbox[0]; u.box[9][k] = -dir[k];
                    u.box[4][k] = p1[k] + dvv[k] * c.bbox[2]; u.box[10][k] = dvv[k];
                    u.box[5][k] = p1[k] - dvv[k] * c.bbox[2]; u.box[11][k] = -dvv[k];
                }
            }
            // seed ellipsoid (line_segment.h:139-154)
            const double f = len / 2;
            double ax0 = f + c.offset_x, ax1 = f, ax2 = f, c00 = f + c.offset_x, cdd = f;
            if (ax0 > 0) { const double ratio = ax1 / ax0; ax0 *= ratio; ax1 *= ratio; ax2 *= ratio; c00 *= ratio; cdd *= ratio; }
            u.ax[0] = ax0; u.ax[1] = ax1; u.ax[2] = ax2;
            const double pitch = atan2(-dv[2], sqrt(dv[0] * dv[0] + dv[1] * dv[1])), yw = atan2(dv[1], dv[0]);
            const M3 Ri = mul(quat_to_rot(cos(yw / 2), 0, 0, sin(yw / 2)), quat_to_rot(cos(pitch / 2), 0, sin(pitch / 2), 0));
            u.count = 0;
            st3(u.Ri, Ri); st3(u.Rf, Ri);
            st3(u.Ci, inverse(rot_diag_rot(Ri, c00, cdd, cdd)));
        }
        __syncthreads();
        CR_ACC(tp_init)
        sc.list = nullptr; sc.W = W_cloud; sc.Pn = P_cloud;
        Best cp;
        if (GRID) {
            cp = scan_grid(c, list, u, s_red, phase);
            if (u.count > CR_LIST) { // more points in the box than the list holds: leave this planner to the plain-cloud kernel
                if (tid == 0) c.poly_index[(size_t)b * c.N] = -1;
                return;
            }
        } else
            cp = scan_cloud(sc, m0, m1, m2, list, u, has_box, c.bbox, s_red, phase);
        if (u.count <= CR_LIST) { // the usual case: from here on a position is an entry of the dense list
            sc.list = list; sc.Pn = u.count; sc.W = (u.count + 63) / 64;
            for (int g = tid >> 6; g < sc.W; g += CR_WAVES) {
                const int pos = g * 64 + (tid & 63);
                const bool valid = pos < sc.Pn;
                const uint64_t w0 = __ballot(valid), w1 = __ballot(valid && (list[valid ? pos : 0] >> 31));
                if ((tid & 63) == 0) { m0[g] = w0; m1[g] = w1; m2[g] = w1; }
            }
        }
        const bool tiled = u.count <= CR_TILE * CR_THREADS;
        Tile tile;
#pragma unroll
        for (int j = 0; j < CR_TILE; ++j) {
            const int pos = ((tid >> 6) + j * CR_WAVES) * 64 + (tid & 63);
            const bool valid = tiled && pos < sc.Pn;
            tile.id[j] = valid ? (int)(list[pos] & 0x7fffffffu) : 0;
            tile.x[j] = valid ? sc.pts[3 * (size_t)tile.id[j]] : 0.0;
            tile.y[j] = valid ? sc.pts[3 * (size_t)tile.id[j] + 1] : 0.0;
            tile.z[j] = valid ? sc.pts[3 * (size_t)tile.id[j] + 2] : 0.0;
            tile.d2[j] = 0.0;
        }
        CR_ACC(tp_cloud)
        // shrink the second axis until no obstacle is inside (line_segment.h:156-181)
        for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
            if (tid == 0) {
                const double pw[3] = {cp.x - u.mid[0], cp.y - u.mid[1], cp.z - u.mid[2]};
                const M3 Ri = ld3(u.Ri);
                double p[3];
                tmul(Ri, pw, p);
                const double roll = atan2(p[2], p[1]);
                const M3 Rf = mul(Ri, quat_to_rot(cos(roll / 2), sin(roll / 2), 0, 0));
                tmul(Rf, pw, p);
                if (p[0] < u.ax[0]) u.ax[1] = fabs(p[1]) / sqrt(1 - (p[0] / u.ax[0]) * (p[0] / u.ax[0]));
                st3(u.Rf, Rf);
                st3(u.Ci, inverse(rot_diag_rot(Rf, u.ax[0], u.ax[1], u.ax[1])));
            }
            __syncthreads();
            CR_ACC(tp_lead)
            cp = tiled ? scan_tile<KEEP_OUTSIDE>(tile, sc.W, m2, m2, u, s_red, phase) : scan<KEEP_OUTSIDE>(sc, m2, m2, u, s_red, phase);
            CR_ACC(tp_scan) CR_CNT(np_scan)
        }
        // third axis (line_segment.h:183-208)
        if (tid == 0) st3(u.Ci, inverse(rot_diag_rot(ld3(u.Rf), u.ax[0], u.ax[1], u.ax[2])));
        __syncthreads();
        CR_ACC(tp_lead)
        cp = tiled ? scan_tile<KEEP_INSIDE>(tile, sc.W, m1, m2, u, s_red, phase) : scan<KEEP_INSIDE>(sc, m1, m2, u, s_red, phase);
        CR_ACC(tp_scan) CR_CNT(np_scan)
        for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
            if (tid == 0) {
                const double pw[3] = {cp.x - u.mid[0], cp.y - u.mid[1], cp.z - u.mid[2]};
                const M3 Rf = ld3(u.Rf);
                double p[3];
                tmul(Rf, pw, p);
                const double dd = 1 - (p[0] / u.ax[0]) * (p[0] / u.ax[0]) - (p[1] / u.ax[1]) * (p[1] / u.ax[1]);
                if (dd > CR_EPS) u.ax[2] = fabs(p[2]) / sqrt(dd);
                st3(u.Ci, inverse(rot_diag_rot(Rf, u.ax[0], u.ax[1], u.ax[2])));
            }
            __syncthreads();
            CR_ACC(tp_lead)
            cp = tiled ? scan_tile<KEEP_OUTSIDE>(tile, sc.W, m2, m2, u, s_red, phase) : scan<KEEP_OUTSIDE>(sc, m2, m2, u, s_red, phase);
            CR_ACC(tp_scan) CR_CNT(np_scan)
        }
        // hyperplanes (decomp_base.h:63-83) + LinearConstraint rows (polyhedron.h:98-118)
        double *gA = c.poly_A + (((size_t)b * c.N + npoly) * c.F) * 3, *gb = c.poly_b + ((size_t)b * c.N + npoly) * c.F;
        if (tid == 0) {
            const M3 Ci = ld3(u.Ci);
            st3(u.CC, mul(Ci, transpose(Ci))); // C^-1 C^-T (ellipsoid.h:53-58)
            u.rows = 0;
        }
        CR_ACC(tp_lead)
        cp = tiled ? scan_tile<KEEP_ALL>(tile, sc.W, m0, m2, u, s_red, phase) : scan<KEEP_ALL>(sc, m0, m2, u, s_red, phase); // Ci is unchanged since the last barrier
        CR_ACC(tp_scan) CR_CNT(np_scan)
        for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
            // closest_hyperplane (ellipsoid.h:53-58) by every lane from the winner the reduction handed out: no
            // publish-through-LDS, hence no barrier between "pick" and "cut"
            const double q[3] = {cp.x, cp.y, cp.z};
            const double w[3] = {q[0] - u.mid[0], q[1] - u.mid[1], q[2] - u.mid[2]};
            double n[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) n[k] = u.CC[3 * k] * w[0] + u.CC[3 * k + 1] * w[1] + u.CC[3 * k + 2] * w[2];
            const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) n[k] /= nl;
            if (tid == 0) emit_row(u, q, n, c.F, s_A, s_b, gA, gb);
            CR_ACC(tp_lead)
            cp = tiled ? scan_tile<KEEP_BEHIND_PLANE>(tile, sc.W, m2, m2, u, s_red, phase, q, n) : scan<KEEP_BEHIND_PLANE>(sc, m2, m2, u, s_red, phase, q, n);
            CR_ACC(tp_scan) CR_CNT(np_scan)
        }
        if (tid == 0) {
            if (has_box)
                for (int k = 0; k < 6; ++k) emit_row(u, u.box[k], u.box[6 + k], c.F, s_A, s_b, gA, gb);
            c.poly_nfaces[(size_t)b * c.N + npoly] = u.rows;
            c.poly_index[(size_t)b * c.N + i] = npoly;
        }
        __syncthreads(); // rows of the new polytope visible to the containment check of the next stage
        CR_ACC(tp_emit)
        rows = u.rows < c.F ? u.rows : c.F;
        ++npoly;
    }
#ifdef FRP_CORRIDOR_PROFILE
    if (tid == 0 && (b == 0 || b == 1000)) {
        printf("scan body %lld reduce %lld (all WGs thread 0)\n", g_prof[0], g_prof[1]);
        printf("corridor wg %d: total %lld check %lld init %lld cloud %lld lead %lld scan %lld (%d scans) emit %lld [100 MHz ticks], %d polytopes\n", b,
               wall_clock64() - tp_begin, tp_check, tp_init, tp_cloud, tp_lead, tp_scan, np_scan, tp_emit, npoly);
    }
#endif
    if (tid == 0) {
        for (int k = npoly; k < c.N; ++k) c.poly_nfaces[(size_t)b * c.N + k] = 0;
        if (c.poly_count) c.poly_count[b] = u.overflow ? -npoly : npoly; // (overflow = a polytope was truncated, so npoly >= 1 and the sign is never lost)
    }
}


// ================================================================== one wavefront per planner (round 4), boxes of any size (round 5)
// A decomposition is a chain of ~20 dependent rounds (pick the closest point, cut, filter), each a few hundred instructions: with four
// wavefronts per planner a round pays a workgroup barrier, an LDS exchange of the per-wave minima and the latency of everything in
// between, and the CU holds 3 planners (12 waves at the 168-register budget).  Measured on the full-tick workload (4096 planners,
// 18 k points, ~1000 of them in a local box): per planner and tick 58 us in 42 scans, 33 us in the two first scans, 25 us between
// scans, 15 us in 20 containment checks -- all latency; two waves per planner (6 per CU) already ran 0.90 -> 0.76 ms.  This kernel
// gives a planner ONE wavefront, so nothing in a round crosses a wave:
//   * the points a round looks at live in the wave's registers, CW_TILE points per lane; the point sets (obs_, obs, the working set) are
//     one bit per point in three 32-bit registers PER LANE -- no LDS masks, no ballots to store them, empty tile rows are skipped;
//   * the closest point of a round is a DPP minimum and six v_readlane: no barrier, no LDS;
//   * the hyperplane loop (decomp_base.h:63-83) compares distances in a FIXED ellipsoid: they are computed once by its opening scan;
//   * the containment checks of the stages behind a new polytope (nmpc_solver.cpp:291-313) are evaluated together, lane = stage, the
//     rows read from LDS: one global round trip per polytope instead of one per stage.
// The wave-uniform 3x3 algebra stays on lane 0 behind LDS (struct Uni) exactly as in the four-wave kernel -- same instructions, same
// results: the kernels produce bit-identical polytopes (tests/test_gpu_parity.py::_check_corridor compares every grid launch -- this
// kernel -- with the plain-cloud launch of the workgroup kernel, array for array).
// Round 4 held the WHOLE box in the tile (20 rows = 1280 points at 256 registers, 2 waves per SIMD = 8 planners per CU: 0.90 -> 0.40 ms
// for the tick's corridor; boxes beyond the tile went back to the workgroup kernels: 4.13 / 11.6 ms on the 19 k / 62 k-point clouds of
// tests/tools/corridor_bench.py).  Round 5: a decomposition never needs its box all at once --
//   * find_ellipsoid (line_segment.h:136-211) only looks at the points inside the SEED ellipsoid: pass A streams the grid rows under the
//     box's hull, counts the in-box points and lists just those (more than the tile holds: the planner is flagged, poly_index[b][0] =
//     -1, for the workgroup kernels launched behind) -- and, betting that there are none, the first shell of the next step too;
//   * find_polyhedron (decomp_base.h:63-83) visits the in-box points in order of their distance in the final ellipsoid, and every cut
//     removes what lies behind it.  So the points are taken in SHELLS of that distance: pass B streams the hull again and lists the
//     points with T_lo <= d2 < T_hi that are in front of every plane cut so far (the planes sit in LDS; the test is the scan's own
//     expression, and a point is alive iff it is in front of ALL planes, whatever the order they are tried in); the tile runs the
//     reference's loop on them until none is left, and since every point outside the shell is farther than every point inside, the
//     closest alive point of the shell IS the closest alive point.  The first shell is sized for 3/8 of a tile (from the count the previous decomposition's shell
//     held; the cloud's mean density for a planner's first box); behind it nearly everything is already cut, so the next shell is tried unbounded and narrowed only if it overflows.
//   Same picks, same cuts, same rows (tests/test_gpu_parity.py::test_corridor_dense_clouds_boxes_beyond_the_register_tile).
// -- and with the box out of the registers the tile can be SMALL: a wavefront's rounds are a dependent chain (one wave per SIMD instead
// of two: the same 87 us per planner and tick), so what counts is wavefronts in flight.  Measured, every planner through this form
// (19 k cloud / 62 k cloud / the tick's corridor, ms; tools/dbg/corridor_allshell.sh): 20 rows at 2 waves per SIMD 1.84 / 3.16 / 0.447,
// 16 rows 1.74 / 2.99 / 0.427, 12 rows at 3 waves (108 spilled registers) 1.76 / 3.01 / 0.459, 8 rows at 4 waves 1.48 / 2.69 / 0.369 --
// with 9 KB of LDS per planner (64 cuts kept, a one-row packing buffer) so that sixteen planners fit a CU 1.43 / 2.61 / 0.339, and with
// the stream two words deep instead of eight (fewer registers in the passes) 1.23 / 2.16 / 0.293 (6 rows at 5 waves: 1.63 / 4.11 /
// 0.395); the rows of the cuts made after the loop, lane = cut, instead of by lane 0 inside every round 1.20 / 2.11 / 0.284; 7 rows and
// shells sized for 5/8 of a tile 1.18 / 2.08 / 0.280; pass A reading only the cells under its first shell (whose bound follows the previous
// decomposition's count) 1.05 / 1.40 / 0.270, and with that a first shell of 3/8 of a tile **0.97 / 1.46 / 0.254**
// (profiles/r05_corridor_knobs.txt).  That is the shipped configuration; the round-4 form (whole box, 20 rows) is gone: 1.89 / 3.22 /
// 0.385 with it in front.
// Needs the uniform grid and the local box (the production configuration).  What this kernel gives up on -- more than a tile of points
// inside the seed ellipsoid, more than CS_PLANES cuts, a shell it cannot narrow -- it flags for the workgroup kernels.
#ifndef FRP_CW_TILE
#define FRP_CW_TILE 7
#endif
#ifndef FRP_CW_WPE
#define FRP_CW_WPE 4
#endif
#ifndef FRP_CW_PACK   // experiment knob: 0 = the survivors of the cuts stay in the tile rows they were listed in
#define FRP_CW_PACK 1
#endif
#ifndef FRP_CW_D2   // experiment knob: 0 = recompute the hyperplane loop's distances every round (one register pair per tile row fewer)
#define FRP_CW_D2 1
#endif
constexpr int CW_TILE = FRP_CW_TILE, CW_CAP = 64 * CW_TILE;
static_assert(CW_TILE <= 32, "one bit per tile row in a 32-bit lane mask");
struct TileW { double x[CW_TILE], y[CW_TILE], z[CW_TILE], d2[FRP_CW_D2 ? CW_TILE : 1]; int id[CW_TILE]; };

#define CW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l)));
}
// the wave's minimum in (distance, cloud index) order, uniform in every lane; idx = INT_MAX when no lane had a point
__device__ __forceinline__ Best wave_best(const Best &mine)
{
    const double d = wave_min_f64(mine.dist);
    uint64_t own = __ballot(mine.dist == d && mine.idx != 0x7fffffff);
    if (own == 0) return Best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
    if (own & (own - 1)) { // equal distances: the smaller cloud index (the reference's first minimum on its order-preserving lists)
        const int i = wave_min_i32(mine.dist == d ? mine.idx : 0x7fffffff);
        own = __ballot(mine.dist == d && mine.idx == i);
    }
    const int l = __builtin_ctzll(own);
    Best r;
    r.dist = d; r.idx = __builtin_amdgcn_readlane(mine.idx, l);
    r.x = readlane_f64(mine.x, l); r.y = readlane_f64(mine.y, l); r.z = readlane_f64(mine.z, l);
    return r;
}

template <int MODE>
__device__ __forceinline__ Best scan_wave(TileW &t, int W, unsigned in, unsigned &out, const Uni &u, const double *pq = nullptr, const double *pn = nullptr,
                                          int *kept = nullptr)
{
    const M3 Ci = ld3(u.Ci);
    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
    double q[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (MODE == KEEP_BEHIND_PLANE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { q[k] = pq[k]; n[k] = pn[k]; }
    }
    Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
    unsigned o = 0;
    int nk = 0;
#pragma unroll
    for (int j = 0; j < CW_TILE; ++j) {
        if (j < W) {
            bool alive = (in >> j) & 1u;
            if (__ballot(alive) != 0) { // (wave-uniform: a tile row with no member is skipped)
                if (alive) {
                    const double dist = (MODE == KEEP_BEHIND_PLANE && FRP_CW_D2) ? t.d2[FRP_CW_D2 ? j : 0] : ell_dist2(Ci, d, t.x[j], t.y[j], t.z[j]);
                    if (MODE == KEEP_ALL && FRP_CW_D2) t.d2[FRP_CW_D2 ? j : 0] = dist;
                    if (MODE == KEEP_OUTSIDE) alive = 1 - sqrt(dist) > CR_EPS;
                    if (MODE == KEEP_INSIDE) alive = dist <= 1;
                    if (MODE == KEEP_BEHIND_PLANE) alive = cut_side(n, q, t.x[j], t.y[j], t.z[j]) < 0;
                    if (alive && before(dist, t.id[j], best.dist, best.idx)) best = Best{dist, t.id[j], t.x[j], t.y[j], t.z[j]};
                }
                o |= (alive ? 1u : 0u) << j;
                if (kept) nk += (int)__popcll(__ballot(alive));
            }
        }
    }
    out = o;
    if (kept) *kept = nk;
    return wave_best(best);
}

#ifndef FRP_CS_FILL // eighths of a tile a shell is sized for (3: since pass A reads only the cells under its shell a smaller one pays -- profiles/r05_corridor_knobs.txt)
#define FRP_CS_FILL 3
#endif
#ifndef FRP_CS_FMAX // the most a first shell's bound grows from one decomposition to the next
#define FRP_CS_FMAX 1.6
#endif
#ifndef FRP_CS_PLANES
#define FRP_CS_PLANES 64
#endif
constexpr int CS_PLANES = FRP_CS_PLANES; // cuts of one decomposition kept for the later shells (more: the planner is left to the workgroup kernels)
constexpr int CS_RETRIES = 48;

// the local box as the first scans test it (frame at p1, half widths with epsilon_), and its axis-aligned hull in grid cells
struct BoxFrame { double fr[3][3], o[3], bh, bd_lo, bd_hi, bv; };
__device__ __forceinline__ BoxFrame load_box(const Uni &u, const frp_nmpc_corridor &c)
{
    BoxFrame f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f.o[k] = u.p1[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) f.fr[k][j] = u.frame[k][j];
    }
    f.bh = c.bbox[1] + CR_EPS; f.bd_lo = -c.bbox[0] - CR_EPS; f.bd_hi = u.len + c.bbox[0] + CR_EPS; f.bv = c.bbox[2] + CR_EPS;
    return f;
}
__device__ __forceinline__ bool in_box(const BoxFrame &f, double x, double y, double z, int id)
{
    const double ex = x - f.o[0], ey = y - f.o[1], ez = z - f.o[2];
    const double h = dot3(f.fr[0][0], f.fr[0][1], f.fr[0][2], ex, ey, ez), tt = dot3(f.fr[1][0], f.fr[1][1], f.fr[1][2], ex, ey, ez),
                 v = dot3(f.fr[2][0], f.fr[2][1], f.fr[2][2], ex, ey, ez);
    return id >= 0 && !(h > f.bh) && !(-h > f.bh) && !(tt > f.bd_hi) && !(tt < f.bd_lo) && !(v > f.bv) && !(-v > f.bv);
}
__device__ __forceinline__ void box_hull(const BoxFrame &f, const frp_nmpc_corridor &c, int (&lo)[3], int (&hi)[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double ctr = f.o[k] + 0.5 * (f.bd_lo + f.bd_hi) * f.fr[1][k];
        const double half = f.bh * fabs(f.fr[0][k]) + 0.5 * (f.bd_hi - f.bd_lo) * fabs(f.fr[1][k]) + f.bv * fabs(f.fr[2][k]);
        const double a = floor((ctr - half - c.grid_origin[k]) / c.grid_cell), bb = floor((ctr + half - c.grid_origin[k]) / c.grid_cell);
        const int n = c.grid_dims[k];
        lo[k] = a < 0 ? 0 : (a > n - 1 ? n - 1 : (int)a);
        hi[k] = bb < 0 ? 0 : (bb > n - 1 ? n - 1 : (int)bb);
    }
}

// every point of the cell-sorted cloud under the hull [lo, hi] of a local box.  The grid rows under the hull (cells lo[0]..hi[0] of one
// (iy, iz): one contiguous run of the sorted points each) are laid end to end -- lane = row fetches its run, a wave scan gives the run
// offsets, a lane finds the run of its flat position by a 6-step search in LDS -- so every wavefront-wide load is full whatever the
// runs' lengths, CS_U of them are in flight, and the next batch is fetched before the current one is looked at (a pass is a chain of
// L2 round trips: 2500 .. 8000 candidates per pass, ~1.5x the in-box points).  f(batch, chunks) is called in uniform control flow with
// the first `chunks` 64-point words of the batch live (id < 0: no point in this lane) and returns false (uniformly) to stop the pass.
// s_row: 128 ints of LDS.
#ifndef FRP_CS_U
#define FRP_CS_U 2
#endif
constexpr int CS_U = FRP_CS_U;
struct HullBatch { double x[CS_U], y[CS_U], z[CS_U]; int id[CS_U]; };

// Before a row is read it is clipped: its cells form a box [x range] x [one cell in y] x [one cell in z] (border cells, which also hold
// the points outside the grid, count as unbounded on their outer side), and a point can only matter if it is on the keep side of every
// plane in s_cuts[0 .. ncuts) -- the six faces of the local box pushed out by a micrometre, then the cuts made so far -- so per plane the
// row keeps just the x range in which SOME point of its y-z cross-section is on the keep side (bounds rounded outward; the exact tests
// are still made per point).  The hull of a rotated box loses a third of its cells this way, the pass behind a finished shell four
// fifths: what is alive by then lies inside the polytope under construction.
template <class Fn>
__device__ __forceinline__ void stream_hull(const frp_nmpc_corridor &c, const int (&lo)[3], const int (&hi)[3], int *s_row, const double *s_cuts, int ncuts, Fn &&f)
{
    const int lane = threadIdx.x;
    const int ny = hi[1] - lo[1] + 1, rows = ny * (hi[2] - lo[2] + 1), nx = c.grid_dims[0];
    const double inf = __builtin_huge_val();
    for (int rb = 0; rb < rows; rb += 64) {
        int mbeg = 0, run = 0;
        if (rb + lane < rows) {
            const int r = rb + lane;
            const int iy = lo[1] + r % ny, iz = lo[2] + r / ny;
            const double ylo = iy == 0 ? -inf : c.grid_origin[1] + iy * c.grid_cell, yhi = iy == c.grid_dims[1] - 1 ? inf : c.grid_origin[1] + (iy + 1) * c.grid_cell;
            const double zlo = iz == 0 ? -inf : c.grid_origin[2] + iz * c.grid_cell, zhi = iz == c.grid_dims[2] - 1 ? inf : c.grid_origin[2] + (iz + 1) * c.grid_cell;
            double xlo = -inf, xhi = inf;
            for (int p = 0; p < ncuts; ++p) {
                const double *pl = s_cuts + 6 * p;
                const double n0 = pl[3], n1 = pl[4], n2 = pl[5];
                // the least n1 (y - q1) + n2 (z - q2) over the cross-section: a point (x, y, z) is kept only if n0 (x - q0) + that < 0
                const double ry = n1 > 0 ? n1 * (ylo - pl[1]) : (n1 < 0 ? n1 * (yhi - pl[1]) : 0.0), rz = n2 > 0 ? n2 * (zlo - pl[2]) : (n2 < 0 ? n2 * (zhi - pl[2]) : 0.0);
                const double rmin = ry + rz;
                if (n0 != 0.0) {
                    const double off = -rmin * __builtin_amdgcn_rcp(n0), bnd = pl[0] + off, mg = 1e-6 + 1e-6 * fabs(off);
                    if (n0 > 0) { if (bnd + mg < xhi) xhi = bnd + mg; } else { if (bnd - mg > xlo) xlo = bnd - mg; }
                } else if (rmin > 1e-6)
                    xlo = inf;
            }
            if (xlo <= xhi) {
                // the cells of [xlo, xhi] the way the grid bins a coordinate: clamped to the grid -- a bound outside the grid still meets the
                // border cell, which holds the points out there (tests: test_corridor_grid_smaller_than_the_cloud_and_other_edges) -- then to the hull
                const double a = floor((xlo - c.grid_origin[0]) / c.grid_cell), bb = floor((xhi - c.grid_origin[0]) / c.grid_cell);
                int ia = a < 0 ? 0 : (a > nx - 1 ? nx - 1 : (int)a), ib = bb < 0 ? 0 : (bb > nx - 1 ? nx - 1 : (int)bb);
                ia = ia > lo[0] ? ia : lo[0]; ib = ib < hi[0] ? ib : hi[0];
                if (ia <= ib) {
                    const size_t row = ((size_t)iz * c.grid_dims[1] + iy) * nx;
                    mbeg = c.grid_start[row + ia];
                    run = c.grid_start[row + ib + 1] - mbeg;
                }
            }
        }
        int inc = run;
#pragma unroll
        for (int dl = 1; dl < 64; dl <<= 1) { const int v = __shfl_up(inc, dl); if (lane >= dl) inc += v; }
        const int total = __builtin_amdgcn_readlane(inc, 63);
        CW_SYNC(); // (the previous block's searches are done)
        s_row[lane] = inc - run; s_row[64 + lane] = mbeg; // rows past the last one have an empty run: their offset is `total`, beyond every position
        CW_SYNC();
        if (total == 0) continue;
        // (no branch in here: positions past the end are clamped to the last point, so the CS_U searches advance in lock step -- one LDS
        // round trip per step, not per step and load -- and the loads of a batch are issued back to back)
        auto fetch = [&](int t0, HullBatch &B) {
            int t[CS_U], r[CS_U];
#pragma unroll
            for (int k = 0; k < CS_U; ++k) { const int tt = t0 + 64 * k + lane; t[k] = tt < total ? tt : total - 1; r[k] = 0; }
#pragma unroll
            for (int step = 32; step; step >>= 1) { // the last run that starts at or before t (never an empty one)
                int v[CS_U];
#pragma unroll
                for (int k = 0; k < CS_U; ++k) v[k] = s_row[r[k] + step];
#pragma unroll
                for (int k = 0; k < CS_U; ++k) r[k] += v[k] <= t[k] ? step : 0;
            }
            int p[CS_U];
#pragma unroll
            for (int k = 0; k < CS_U; ++k) p[k] = s_row[64 + r[k]] + (t[k] - s_row[r[k]]);
#pragma unroll
            for (int k = 0; k < CS_U; ++k) {
                const size_t p3 = 3 * (size_t)p[k];
                B.x[k] = c.grid_points[p3]; B.y[k] = c.grid_points[p3 + 1]; B.z[k] = c.grid_points[p3 + 2];
                B.id[k] = c.grid_index[p[k]];
            }
#pragma unroll
            for (int k = 0; k < CS_U; ++k) B.id[k] = t0 + 64 * k + lane < total ? B.id[k] : -1;
        };
        HullBatch cur, nxt;
        fetch(0, cur);
        for (int t0 = 0; t0 < total; t0 += 64 * CS_U) {
            const bool has_next = t0 + 64 * CS_U < total;
            if (has_next) fetch(t0 + 64 * CS_U, nxt);
            const int left = (total - t0 + 63) / 64;
            if (!f(cur, left < CS_U ? left : CS_U)) return;
            if (has_next) cur = nxt;
        }
    }
}

// The points a cut leaves alive are scattered over all the tile rows they were listed in (a thousand points: after five cuts a hundred
// are left, in seven rows of sixteen on average -- and a round costs its live ROWS).  Once no more than CW_PACK are alive they are moved
// to the first rows, through LDS (the list's space: it has been read by then), and the rounds behind visit two rows.  Which lane holds a
// point does not matter to any result: minima are taken in (distance, cloud index) order.
#ifndef FRP_CW_PACKN
#define FRP_CW_PACKN 64
#endif
constexpr int CW_PACK = FRP_CW_PACKN;
constexpr int CW_LIST = CW_CAP > CW_PACK * 9 ? CW_CAP : CW_PACK * 9; // entries of the in-box list; also the packing buffer (36 bytes per packed point)
static_assert(CW_PACK % 64 == 0, "whole tile rows");
__device__ __forceinline__ void pack_tile(TileW &t, unsigned &m, int &W, int alive, uint32_t *buf)
{
    const int lane = threadIdx.x;
    double *bx = reinterpret_cast<double *>(buf), *by = bx + CW_PACK, *bz = by + CW_PACK, *bd = bz + CW_PACK;
    int *bi = reinterpret_cast<int *>(bd + CW_PACK);
    const int c = __popc(m);
    int inc = c;
#pragma unroll
    for (int dl = 1; dl < 64; dl <<= 1) { const int v = __shfl_up(inc, dl); if (lane >= dl) inc += v; }
    int slot = inc - c;
#pragma unroll
    for (int j = 0; j < CW_TILE; ++j) {
        if (j < W && ((m >> j) & 1u)) {
            bx[slot] = t.x[j]; by[slot] = t.y[j]; bz[slot] = t.z[j]; bd[slot] = t.d2[FRP_CW_D2 ? j : 0]; bi[slot] = t.id[j];
            ++slot;
        }
    }
    CW_SYNC();
    unsigned o = 0;
#pragma unroll
    for (int j = 0; j < CW_PACK / 64; ++j) {
        const int pos = j * 64 + lane;
        const bool valid = pos < alive;
        t.x[j] = valid ? bx[pos] : 0.0; t.y[j] = valid ? by[pos] : 0.0; t.z[j] = valid ? bz[pos] : 0.0;
        t.d2[FRP_CW_D2 ? j : 0] = valid ? bd[pos] : 0.0; t.id[j] = valid ? bi[pos] : 0;
        o |= (valid ? 1u : 0u) << j;
    }
    CW_SYNC();
    m = o; W = CW_PACK / 64;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRP_CW_WPE, FRP_CW_WPE))) void corridor_wave_kernel(frp_nmpc_corridor c)
{
    __shared__ double s_A[FRP_CORRIDOR_MAX_F * 3], s_b[FRP_CORRIDOR_MAX_F];
    __shared__ uint32_t list[CW_LIST];
    __shared__ Uni u;
    __shared__ double s_pl[(6 + CS_PLANES) * 6]; // the local box's faces (pushed out, for the row clipping only), then the cuts of the running decomposition, (q, n) as the scans use them
    __shared__ int s_row[128];                         // stream_hull's run offsets
    __shared__ double s_seedCi[9];                     // C^-1 of the seed ellipsoid, to tell whether find_ellipsoid changed it
    __shared__ int s_same;
    double rho_prev = 0.0, T1_prev = 0.0; // the planner's previous decomposition (the next box is a little further along the path): cloud points per unit of
    int cnt_prev = -1;                    // volume around its seed, the bound of its first shell and the points that shell held (-1: no decomposition yet)
    const int b = blockIdx.x, lane = threadIdx.x;
#ifdef FRP_CORRIDOR_PROFILE
    long long tp_a = 0, tp_b = 0, tp_tile = 0, tp_shrink = 0, tp_rest = 0, tp_begin = wall_clock64();
    int np_b = 0, np_retry = 0, np_shell = 0, np_dec = 0, np_round = 0, np_listed = 0, np_box = 0;
    CR_T0
#endif
    const double *ref = c.ref_pos + (size_t)b * c.N * 3, *yaw = c.ref_yaw + (size_t)b * c.N, *Eb = c.ellipsoid + (size_t)b * c.N * 9;
    const int P_cloud = c.P;
    const int max_rounds = P_cloud + 8; // (see corridor_kernel: only non-finite input needs the bound)
    if (lane == 0) u.overflow = 0;
    int npoly = 0;
    int i = 0;
    while (i < c.N) {
        // ---- new decomposition around the seed segment of stage i (nmpc_solver.cpp:315-329): lane 0, as in corridor_kernel ----
        if (lane == 0) {
            double sy, cy;
            sincos(yaw[i], &sy, &cy);
            const double p1[3] = {ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]};
            const double p2[3] = {p1[0] + c.seed_len * cy, p1[1] + c.seed_len * sy, p1[2]};
            const double dv[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            const double len = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) u.mid[k] = (p1[k] + p2[k]) / 2;
            u.len = len;
            { // local box planes (line_segment.h:47-85)
                const double dir[3] = {dv[0] / len, dv[1] / len, dv[2] / len};
                double dh[3] = {dir[1], -dir[0], 0.0};
                double hn = sqrt(dh[0] * dh[0] + dh[1] * dh[1]);
                if (hn == 0.0) { dh[0] = -1.0; dh[1] = 0.0; hn = 1.0; }
                dh[0] /= hn; dh[1] /= hn;
                const double dvv[3] = {dir[1] * dh[2] - dir[2] * dh[1], dir[2] * dh[0] - dir[0] * dh[2], dir[0] * dh[1] - dir[1] * dh[0]};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u.frame[0][k] = dh[k]; u.frame[1][k] = dir[k]; u.frame[2][k] = dvv[k]; u.p1[k] = p1[k];
                    u.box[0][k] = p1[k] + dh[k] * c.bbox[1];  u.box[6][k] = dh[k];
                    u.box[1][k] = p1[k] - dh[k] * c.bbox[1];  u.box[7][k] = -dh[k];
                    u.box[2][k] = p2[k] + dir[k] * c.bbox[0]; u.box[8][k] = dir[k];
                    u.box[3][k] = p1[k] - dir[k] * c.bbox[0]; u.box[9][k] = -dir[k];
                    u.box[4][k] = p1[k] + dvv[k] * c.bbox[2]; u.box[10][k] = dvv[k];
                    u.box[5][k] = p1[k] - dvv[k] * c.bbox[2]; u.box[11][k] = -dvv[k];
                }
            }
            const double f = len / 2;
            double ax0 = f + c.offset_x, ax1 = f, ax2 = f, c00 = f + c.offset_x, cdd = f;
            if (ax0 > 0) { const double ratio = ax1 / ax0; ax0 *= ratio; ax1 *= ratio; ax2 *= ratio; c00 *= ratio; cdd *= ratio; }
            u.ax[0] = ax0; u.ax[1] = ax1; u.ax[2] = ax2;
            const double pitch = atan2(-dv[2], sqrt(dv[0] * dv[0] + dv[1] * dv[1])), yw = atan2(dv[1], dv[0]);
            const M3 Ri = mul(quat_to_rot(cos(yw / 2), 0, 0, sin(yw / 2)), quat_to_rot(cos(pitch / 2), 0, sin(pitch / 2), 0));
            st3(u.Ri, Ri); st3(u.Rf, Ri);
            st3(u.Ci, inverse(rot_diag_rot(Ri, c00, cdd, cdd)));
            st3(s_seedCi, ld3(u.Ci));
            for (int k = 0; k < 6; ++k)
                for (int j = 0; j < 3; ++j) { s_pl[6 * k + j] = u.box[k][j] + 1e-6 * u.box[6 + k][j]; s_pl[6 * k + 3 + j] = u.box[6 + k][j]; }
        }
        CW_SYNC();
        // ---- first scan through the grid: the in-box points -> list (cloud index | inside-the-seed-ellipsoid flag), the closest inside one
        int count = 0;
        Best cp;
        int hlo[3] = {0, 0, 0}, hhi[3] = {0, 0, 0}, nbox = 0; // the hull of the box in grid cells, the in-box points
#ifdef FRP_CORRIDOR_PROFILE
        CR_ACC(tp_rest)
#endif
        double T1 = -1.0; // bound of the shell pass A lists beside the points inside the seed ellipsoid (< 0: none)
        int rest1 = 0;    //        in-box points beyond it
        // pass A: count the in-box points, list those inside the seed ellipsoid (flag bit), find the closest of them -- and, betting that
        // none is inside (then find_ellipsoid leaves the seed ellipsoid as it is), list the first shell of the seed ellipsoid's metric too
        const M3 Ci = ld3(u.Ci);
        const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
        const BoxFrame bf = load_box(u, c);
        box_hull(bf, c, hlo, hhi);
        // FRP_CS_FILL eighths of a tile at the density the previous decomposition met -- the cloud's mean, for the planner's first box (any
        // value is correct; this one avoids retries)
        const double box_vol = 8.0 * c.bbox[1] * c.bbox[2] * (0.5 * u.len + c.bbox[0]), unit_seed = 4.1887902047863905 * u.ax[0] * u.ax[1] * u.ax[2];
        {
            const double target = (double)(CW_CAP * FRP_CS_FILL / 8);
            if (cnt_prev >= 0 && T1_prev > 1.0 && T1_prev < __builtin_huge_val()) {
                // the previous shell held cnt_prev points: scale its bound for the target as if the count grew with the shell's volume, by at
                // most 1.6 either way (a density from the shell's own volume would be fooled by the free space around the path)
                double f = cnt_prev > 0 ? cbrt(target / (double)cnt_prev) : 1.26;
                f = f * f; f = f < 0.6 ? 0.6 : (f > FRP_CS_FMAX ? FRP_CS_FMAX : f);
                T1 = T1_prev * f > 1.0 ? T1_prev * f : T1_prev;
            } else {
                const double rho = rho_prev > 0.0 ? rho_prev : (double)c.P / (c.grid_cell * c.grid_cell * c.grid_cell * c.grid_dims[0] * c.grid_dims[1] * c.grid_dims[2]);
                if (rho * box_vol > 0.9 * CW_CAP) {
                    const double r = cbrt(target / (rho * unit_seed)); // (rho unit_seed: points per unit of d2^(3/2))
                    if (r * r > 1.0 && r * r < __builtin_huge_val()) T1 = r * r;
                } else
                    T1 = __builtin_huge_val();
            }
        }
        // A bounded first shell is an ellipsoid around the seed, a fraction of the box: its axis-aligned bounds join the box faces in the row
        // clipping (cut slots 0..5, free until the first cut is made), so this pass reads the cells under the SHELL, not under the box
        // (the points inside the seed ellipsoid lie inside it too).  The in-box count is then the shell's neighbourhood's, and whether
        // anything lies beyond is not known: the pass behind the shell always runs.
        bool clipped = T1 > 1.0 && T1 < __builtin_huge_val();
        if (clipped) {
            if (lane == 0) {
                const M3 R = ld3(u.Ri);
                for (int k = 0; k < 3; ++k) {
                    double e2 = 0.0;
                    for (int j = 0; j < 3; ++j) e2 += (R.m[3 * k + j] * u.ax[j]) * (R.m[3 * k + j] * u.ax[j]);
                    const double ext = sqrt(T1 * e2) * (1.0 + 1e-9) + 1e-9;
                    for (int sgn = 0; sgn < 2; ++sgn) {
                        double *pl = s_pl + 36 + 6 * (2 * k + sgn);
                        for (int j = 0; j < 3; ++j) { pl[j] = u.mid[j] + (j == k ? (sgn ? -ext : ext) : 0.0); pl[3 + j] = j == k ? (sgn ? -1.0 : 1.0) : 0.0; }
                    }
                }
            }
            CW_SYNC();
        }
        for (;;) {
            Best best{1.7976931348623157e308, 0x7fffffff, 0.0, 0.0, 0.0};
            int n_in = 0;
            count = 0; nbox = 0; rest1 = 0;
            stream_hull(c, hlo, hhi, s_row, s_pl, clipped ? 12 : 6, [&](const HullBatch &B, int chunks) {
#pragma unroll
                for (int k = 0; k < CS_U; ++k) {
                    if (k >= chunks) break;
                    const double x = B.x[k], y = B.y[k], z = B.z[k];
                    const int id = B.id[k];
                    const bool in0 = in_box(bf, x, y, z, id);
                    bool i1 = false, s1 = false;
                    if (in0) {
                        const double dist = ell_dist2(Ci, d, x, y, z);
                        i1 = dist <= 1;
                        s1 = !i1 && dist < T1;
                        if (i1 && before(dist, id, best.dist, best.idx)) best = Best{dist, id, x, y, z};
                    }
                    nbox += (int)__popcll(__ballot(in0));
                    n_in += (int)__popcll(__ballot(i1));
                    rest1 += (int)__popcll(__ballot(in0 && !i1 && !s1));
                    const uint64_t w1 = __ballot(i1 || s1);
                    if (w1) {
                        const int mine = count + (int)__popcll(w1 & ((1ull << lane) - 1));
                        if ((i1 || s1) && mine < CW_CAP) list[mine] = (uint32_t)id | (i1 ? 0x80000000u : 0u);
                        count += (int)__popcll(w1);
                    }
                }
                return count <= CW_CAP;
            });
            if (count > CW_CAP && T1 > 0.0) { // the bet's shell overflowed the tile: the inside points alone, the whole box; the next box starts from half the bound
                T1_prev = T1 < __builtin_huge_val() ? 0.5 * T1 : 0.0; cnt_prev = T1_prev > 1.0 ? (int)(CW_CAP * FRP_CS_FILL / 8) : -1;
                T1 = -1.0; clipped = false; CW_SYNC();
                continue;
            }
            if (T1 > 0.0) { T1_prev = T1; cnt_prev = count - n_in; }
            // the density met (for a first shell in another metric, below): the shell's points over its volume, or the box's over the box's
            rho_prev = clipped ? (double)(count > 0 ? count : 1) / (T1 * sqrt(T1) * unit_seed) : (double)nbox / box_vol;
            if (clipped) rest1 = 1;
            if (n_in > 0) T1 = -1.0; // find_ellipsoid has work to do: the listed shell is not one of the final ellipsoid (its points stay out of m1)
            cp = wave_best(best);
            break;
        }
#ifdef FRP_CORRIDOR_PROFILE
        CR_ACC(tp_a) ++np_dec; np_box += nbox;
#endif
        if (count > CW_CAP) { // more points in the box than the register tile holds: the workgroup kernels behind take this planner
            if (lane == 0) c.poly_index[(size_t)b * c.N] = -1;
            return;
        }
        CW_SYNC();
        // ---- the register tile and the three point sets (one bit per tile row and lane)
        const int W = (count + 63) / 64;
        TileW tile;
        unsigned m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
        for (int j = 0; j < CW_TILE; ++j) {
            const int pos = j * 64 + lane;
            const bool valid = pos < count;
            const uint32_t e = valid ? list[pos] : 0u;
            const int idj = (int)(e & 0x7fffffffu);
            tile.id[j] = idj;
            tile.x[j] = valid ? c.cloud[3 * (size_t)idj] : 0.0;
            tile.y[j] = valid ? c.cloud[3 * (size_t)idj + 1] : 0.0;
            tile.z[j] = valid ? c.cloud[3 * (size_t)idj + 2] : 0.0;
            if (FRP_CW_D2) tile.d2[FRP_CW_D2 ? j : 0] = 0.0;
            m0 |= (valid ? 1u : 0u) << j;
            m1 |= ((valid && (e >> 31)) ? 1u : 0u) << j;
        }
        m2 = m1;
        // shrink the second axis until no obstacle is inside (line_segment.h:156-181)
        for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
            if (lane == 0) {
                const double pw[3] = {cp.x - u.mid[0], cp.y - u.mid[1], cp.z - u.mid[2]};
                const M3 Ri = ld3(u.Ri);
                double p[3];
                tmul(Ri, pw, p);
                const double roll = atan2(p[2], p[1]);
                const M3 Rf = mul(Ri, quat_to_rot(cos(roll / 2), sin(roll / 2), 0, 0));
                tmul(Rf, pw, p);
                if (p[0] < u.ax[0]) u.ax[1] = fabs(p[1]) / sqrt(1 - (p[0] / u.ax[0]) * (p[0] / u.ax[0]));
                st3(u.Rf, Rf);
                st3(u.Ci, inverse(rot_diag_rot(Rf, u.ax[0], u.ax[1], u.ax[1])));
            }
            CW_SYNC();
            cp = scan_wave<KEEP_OUTSIDE>(tile, W, m2, m2, u);
        }
        // third axis (line_segment.h:183-208)
        if (lane == 0) st3(u.Ci, inverse(rot_diag_rot(ld3(u.Rf), u.ax[0], u.ax[1], u.ax[2])));
        CW_SYNC();
        cp = scan_wave<KEEP_INSIDE>(tile, W, m1, m2, u);
        for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
            if (lane == 0) {
                const double pw[3] = {cp.x - u.mid[0], cp.y - u.mid[1], cp.z - u.mid[2]};
                const M3 Rf = ld3(u.Rf);
                double p[3];
                tmul(Rf, pw, p);
                const double dd = 1 - (p[0] / u.ax[0]) * (p[0] / u.ax[0]) - (p[1] / u.ax[1]) * (p[1] / u.ax[1]);
                if (dd > CR_EPS) u.ax[2] = fabs(p[2]) / sqrt(dd);
                st3(u.Ci, inverse(rot_diag_rot(Rf, u.ax[0], u.ax[1], u.ax[2])));
            }
            CW_SYNC();
            cp = scan_wave<KEEP_OUTSIDE>(tile, W, m2, m2, u);
        }
        // hyperplanes (decomp_base.h:63-83) + LinearConstraint rows (polyhedron.h:98-118)
        double *gA = c.poly_A + (((size_t)b * c.N + npoly) * c.F) * 3, *gb = c.poly_b + ((size_t)b * c.N + npoly) * c.F;
        if (lane == 0) {
            const M3 Ci = ld3(u.Ci);
            st3(u.CC, mul(Ci, transpose(Ci)));
            u.rows = 0;
        }
        CW_SYNC();
        { // the in-box points in shells of their distance in the final ellipsoid
#ifdef FRP_CORRIDOR_PROFILE
            CR_ACC(tp_shrink)
#endif
            const double inf = __builtin_huge_val();
            int npl = 0, tries = 0;
            double T_lo = -1.0, T_hi = inf;
            bool have_tile = false;
            if (T1 > 0.0) { // pass A's bet: is the final ellipsoid the seed ellipsoid, bit for bit?  Then its shell is the first one, already in the tile
                if (lane == 0) {
                    int same = 1;
                    for (int k = 0; k < 9; ++k) same &= u.Ci[k] == s_seedCi[k] ? 1 : 0;
                    s_same = same;
                }
                CW_SYNC();
                have_tile = s_same != 0;
            }
            if (!have_tile && rho_prev * box_vol > 0.9 * CW_CAP) { // first shell: the same fraction of a tile at the density pass A met, in the FINAL ellipsoid's metric
                const double per_unit = rho_prev * 4.1887902047863905 * u.ax[0] * u.ax[1] * u.ax[2]; // points per unit of d2^(3/2)
                const double r = cbrt((double)(CW_CAP * FRP_CS_FILL / 8) / per_unit);
                if (r * r > 1.0 && r * r < inf) T_hi = r * r;
            }
            // find_polyhedron's loop on the points of the tile (decomp_base.h:63-83); every cut is kept for the shells behind.  false: too many cuts
            auto cut_tile = [&](int Ws, unsigned s0) -> bool {
                unsigned s2 = 0;
                cp = scan_wave<KEEP_ALL>(tile, Ws, s0, s2, u);
                const double *hm = u.mid, *hC = u.CC; // (measured: the twelve values in scalar registers across the rounds instead -- no gain)
                for (int guard = 0; cp.idx != 0x7fffffff && guard < max_rounds; ++guard) {
                    const double q[3] = {cp.x, cp.y, cp.z};
                    const double w[3] = {q[0] - hm[0], q[1] - hm[1], q[2] - hm[2]};
                    double n[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) n[k] = hC[3 * k] * w[0] + hC[3 * k + 1] * w[1] + hC[3 * k + 2] * w[2];
                    const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) n[k] /= nl;
                    if (npl >= CS_PLANES) return false;
                    if (lane == 0) { // (the cut is kept; its ROW is made after the loop, off this chain)
#pragma unroll
                        for (int k = 0; k < 3; ++k) { s_pl[36 + 6 * npl + k] = q[k]; s_pl[36 + 6 * npl + 3 + k] = n[k]; }
                    }
                    ++npl;
                    if (FRP_CW_PACK && Ws > CW_PACK / 64) {
                        int kept;
                        cp = scan_wave<KEEP_BEHIND_PLANE>(tile, Ws, s2, s2, u, q, n, &kept);
                        if (kept <= CW_PACK) pack_tile(tile, s2, Ws, kept, list);
                    } else
                        cp = scan_wave<KEEP_BEHIND_PLANE>(tile, Ws, s2, s2, u, q, n);
                }
                CW_SYNC(); // the new cuts are visible to every lane; the list may be overwritten
                return true;
            };
            bool more = true;
            if (have_tile) { // (its own copy of the loop: inside the shell loop below the tile is dead while a pass streams)
                if (!cut_tile(W, m0)) { if (lane == 0) c.poly_index[(size_t)b * c.N] = -1; return; }
#ifdef FRP_CORRIDOR_PROFILE
                CR_ACC(tp_tile) np_round += npl; ++np_shell; np_listed += count;
#endif
                more = rest1 > 0 && T1 < inf;
                T_lo = T1; T_hi = inf;
            }
            while (more) {
                int cnt = 0, rest = 0;
                {
                    const M3 Ci = ld3(u.Ci);
                    const double d[3] = {u.mid[0], u.mid[1], u.mid[2]};
                    const BoxFrame bf = load_box(u, c);
                    stream_hull(c, hlo, hhi, s_row, s_pl, 6 + npl, [&](const HullBatch &B, int chunks) {
                        bool al[CS_U];
                        double dist[CS_U];
                        uint64_t any = 0;
#pragma unroll
                        for (int k = 0; k < CS_U; ++k) {
                            al[k] = k < chunks && in_box(bf, B.x[k], B.y[k], B.z[k], B.id[k]);
                            dist[k] = 0.0;
                            if (al[k]) { dist[k] = ell_dist2(Ci, d, B.x[k], B.y[k], B.z[k]); al[k] = dist[k] >= T_lo; }
                            any |= __ballot(al[k]);
                        }
                        // a point is alive iff it is in front of every cut so far (most are behind one of the first): a cut is fetched from
                        // LDS once for the whole batch
                        for (int p = 0; p < npl && any; ++p) {
                            double q[3], n[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) { q[j] = s_pl[36 + 6 * p + j]; n[j] = s_pl[36 + 6 * p + 3 + j]; }
                            any = 0;
#pragma unroll
                            for (int k = 0; k < CS_U; ++k) {
                                al[k] = al[k] && cut_side(n, q, B.x[k], B.y[k], B.z[k]) < 0;
                                any |= __ballot(al[k]);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < CS_U; ++k) {
                            const bool sh = al[k] && dist[k] < T_hi;
                            rest += (int)__popcll(__ballot(al[k] && !sh));
                            const uint64_t w = __ballot(sh);
                            if (w) {
                                const int mine = cnt + (int)__popcll(w & ((1ull << lane) - 1));
                                if (sh && mine < CW_CAP) list[mine] = (uint32_t)B.id[k];
                                cnt += (int)__popcll(w);
                            }
                        }
                        return cnt <= CW_CAP;
                    });
                }
#ifdef FRP_CORRIDOR_PROFILE
                CR_ACC(tp_b) ++np_b; if (cnt > CW_CAP) ++np_retry; else { ++np_shell; np_listed += cnt; }
#endif
                if (cnt > CW_CAP) { // more than a tile: narrow the shell and stream again
                    if (++tries > CS_RETRIES) { if (lane == 0) c.poly_index[(size_t)b * c.N] = -1; return; }
                    const double base = T_lo > 0.0 ? T_lo : 0.0;
                    T_hi = T_hi == inf ? (base > 1.0 ? base : 1.0) * 2.5 : base + (T_hi - base) * 0.4;
                    CW_SYNC();
                    continue;
                }
                CW_SYNC();
                unsigned s0 = 0;
#pragma unroll
                for (int j = 0; j < CW_TILE; ++j) {
                    const int pos = j * 64 + lane;
                    const bool valid = pos < cnt;
                    const int idj = valid ? (int)list[pos] : 0;
                    tile.id[j] = idj;
                    tile.x[j] = valid ? c.cloud[3 * (size_t)idj] : 0.0;
                    tile.y[j] = valid ? c.cloud[3 * (size_t)idj + 1] : 0.0;
                    tile.z[j] = valid ? c.cloud[3 * (size_t)idj + 2] : 0.0;
                    if (FRP_CW_D2) tile.d2[FRP_CW_D2 ? j : 0] = 0.0;
                    s0 |= (valid ? 1u : 0u) << j;
                }
                if (!cut_tile((cnt + 63) / 64, s0)) { if (lane == 0) c.poly_index[(size_t)b * c.N] = -1; return; }
#ifdef FRP_CORRIDOR_PROFILE
                CR_ACC(tp_tile) np_round += npl;
#endif
                more = rest > 0 && T_hi < inf;
                T_lo = T_hi; T_hi = inf;
            }
            // the LinearConstraint rows of the cuts (polyhedron.h:98-118), lane = cut, all at once: emit_row's own arithmetic, but not one
            // lane-0 detour (LDS counter, three dot products, eight stores) in every round of the loop above
            static_assert(CS_PLANES <= 64, "one lane per cut");
            if (lane < npl) {
                const double *pl = s_pl + 36 + 6 * lane;
                double n[3] = {pl[3], pl[4], pl[5]};
                double cc = pl[0] * n[0] + pl[1] * n[1] + pl[2] * n[2];
                if (n[0] * u.mid[0] + n[1] * u.mid[1] + n[2] * u.mid[2] - cc > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; cc = -cc; }
                if (lane < c.F) {
                    s_A[3 * lane] = n[0]; s_A[3 * lane + 1] = n[1]; s_A[3 * lane + 2] = n[2]; s_b[lane] = cc;
                    gA[3 * lane] = n[0]; gA[3 * lane + 1] = n[1]; gA[3 * lane + 2] = n[2]; gb[lane] = cc;
                }
            }
            if (lane == 0) { u.rows = npl; if (npl > c.F) u.overflow = 1; }
            CW_SYNC();
        }
        if (lane == 0) {
            for (int k = 0; k < 6; ++k) emit_row(u, u.box[k], u.box[6 + k], c.F, s_A, s_b, gA, gb);
            c.poly_nfaces[(size_t)b * c.N + npoly] = u.rows;
            c.poly_index[(size_t)b * c.N + i] = npoly;
        }
        CW_SYNC();
        const int rows = u.rows < c.F ? u.rows : c.F;
        // ---- which of the stages behind still fit this polytope (nmpc_solver.cpp:291-313)?  All of them at once: lane = (row group, stage) --
        // 64 / N lanes share a stage's rows (three at N = 20: a row costs a square root, thirty of them on twenty lanes were 6 us of every
        // decomposition) and a ballot puts the groups' verdicts together
        const int cgrp = 64 / c.N, cst = lane % c.N, csub = lane / c.N;
        bool viol = false;
        if (csub < cgrp && cst > i) {
            const double *E = Eb + 9 * cst;
            const double E0 = E[0], E1 = E[1], E2 = E[2], E3 = E[3], E4 = E[4], E5 = E[5], E6 = E[6], E7 = E[7], E8 = E[8];
            const double r0 = ref[3 * cst], r1 = ref[3 * cst + 1], r2 = ref[3 * cst + 2];
            for (int r = csub; r < rows; r += cgrp) {
                const double a0 = s_A[3 * r], a1 = s_A[3 * r + 1], a2 = s_A[3 * r + 2];
                const double e0 = E0 * a0 + E1 * a1 + E2 * a2, e1 = E3 * a0 + E4 * a1 + E5 * a2, e2 = E6 * a0 + E7 * a1 + E8 * a2;
                const double add = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
                viol = viol || (a0 * r0 + a1 * r1 + a2 * r2 - (s_b[r] - c.inflation * add)) > 0;
            }
        }
        uint64_t vm = __ballot(viol);
        for (int g = 1; g < cgrp; ++g) vm |= vm >> (g * c.N); // (bits 0 .. N-1: the stage's verdict over all its row groups; garbage above)
        if (c.N < 64) vm &= (1ull << c.N) - 1ull;
        const int next = vm ? (int)__builtin_ctzll(vm) : c.N; // the first stage whose inflated tube ellipsoid leaves the polytope
        if (lane > i && lane < next) c.poly_index[(size_t)b * c.N + lane] = npoly;
        ++npoly;
        i = next;
    }
#ifdef FRP_CORRIDOR_PROFILE
    if (lane == 0 && (b == 0 || b == 1000) && np_dec > 0)
        printf("wave %d: total %lld passA %lld passB %lld (%d passes, %d retries, %d shells) tile+rounds %lld shrink %lld rest %lld [100 MHz ticks]; %d decompositions, "
               "%d in-box points, %d listed, %d cuts (cumulative per shell)\n", b, wall_clock64() - tp_begin, tp_a, tp_b, np_b, np_retry, np_shell, tp_tile, tp_shrink, tp_rest,
               np_dec, np_box, np_listed, np_round);
#endif
    if (lane == 0) {
        for (int k = npoly; k < c.N; ++k) c.poly_nfaces[(size_t)b * c.N + k] = 0;
        if (c.poly_count) c.poly_count[b] = u.overflow ? -npoly : npoly;
    }
}

} // namespace frp

namespace frp {

__device__ __forceinline__ int grid_cell_of(const double *pt, const double *origin, double cell, const int *dims)
{
    int ix[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = floor((pt[k] - origin[k]) / cell);
        ix[k] = !(a > 0) ? 0 : (a > dims[k] - 1 ? dims[k] - 1 : (int)a); // NaN and points beyond the grid go to border cells
    }
    return (ix[2] * dims[1] + ix[1]) * dims[0] + ix[0];
}

struct GridArgs { const double *cloud; int P; double origin[3]; double cell; int dims[3]; double *points; int *index; int *start; int *cursor; };

__global__ void grid_zero_kernel(GridArgs g, int cells)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= cells; i += gridDim.x * blockDim.x) { g.start[i] = 0; if (i < cells) g.cursor[i] = 0; }
}
__global__ void grid_count_kernel(GridArgs g)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.P; i += gridDim.x * blockDim.x)
        atomicAdd(&g.start[grid_cell_of(g.cloud + 3 * (size_t)i, g.origin, g.cell, g.dims) + 1], 1);
}
// inclusive prefix sum of start[1..cells] in place, one workgroup walking the array in 1024-element chunks
__global__ __launch_bounds__(1024) void grid_scan_kernel(GridArgs g, int cells)
{
    __shared__ int s_part[16], s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 1; base <= cells; base += 1024) {
        const int i = base + tid;
        int v = i <= cells ? g.start[i] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off); if (lane >= off) v += t; }
        if (lane == 63) s_part[wave] = v;
        __syncthreads();
        int add = s_carry;
        for (int w = 0; w < wave; ++w) add += s_part[w];
        if (i <= cells) g.start[i] = v + add;
        __syncthreads();
        if (tid == 1023) s_carry = v + add;
        __syncthreads();
    }
}
__global__ void grid_scatter_kernel(GridArgs g)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.P; i += gridDim.x * blockDim.x) {
        const double *pt = g.cloud + 3 * (size_t)i;
        const int cell = grid_cell_of(pt, g.origin, g.cell, g.dims);
        const int at = g.start[cell] + atomicAdd(&g.cursor[cell], 1); // order inside a cell is irrelevant (ties go by cloud index)
        g.points[3 * (size_t)at] = pt[0]; g.points[3 * (size_t)at + 1] = pt[1]; g.points[3 * (size_t)at + 2] = pt[2];
        g.index[at] = i;
    }
}

} // namespace frp

extern "C" int frp_nmpc_cloud_grid_build(const double *cloud, int P, const double origin[3], double cell, const int dims[3],
                                         double *grid_points, int *grid_index, int *grid_start, int *scratch, void *stream)
{
    if (P < 0 || (P > 0 && !cloud) || !origin || !dims || !(cell > 0.0) || !grid_points || !grid_index || !grid_start || !scratch) return FRP_ERR_ARG;
    if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || (long long)dims[0] * dims[1] * dims[2] > FRP_CORRIDOR_MAX_CELLS) return FRP_ERR_ARG;
    const int cells = dims[0] * dims[1] * dims[2];
    frp::GridArgs g = {cloud, P, {origin[0], origin[1], origin[2]}, cell, {dims[0], dims[1], dims[2]}, grid_points, grid_index, grid_start, scratch};
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(frp::grid_zero_kernel, dim3(256), dim3(256), 0, st, g, cells);
    if (P > 0) hipLaunchKernelGGL(frp::grid_count_kernel, dim3(256), dim3(256), 0, st, g);
    hipLaunchKernelGGL(frp::grid_scan_kernel, dim3(1), dim3(1024), 0, st, g, cells);
    if (P > 0) hipLaunchKernelGGL(frp::grid_scatter_kernel, dim3(256), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}

extern "C" int frp_nmpc_corridor_batch(const frp_nmpc_corridor *p, void *stream)
{
    if (!p || p->B <= 0 || p->N < 1 || p->N > 64 || p->F < 6 || p->F > FRP_CORRIDOR_MAX_F || p->P < 0 || p->P > FRP_CORRIDOR_MAX_POINTS ||
        (p->P > 0 && !p->cloud) || !p->ref_pos || !p->ref_yaw || !p->ellipsoid || !p->poly_A || !p->poly_b || !p->poly_nfaces || !p->poly_index)
        return FRP_ERR_ARG;
    if (!(p->seed_len > 0.0) || !(p->inflation >= 0.0)) return FRP_ERR_ARG;
    if (p->grid_start && (!p->grid_points || !p->grid_index || !(p->grid_cell > 0.0) || p->grid_dims[0] < 1 || p->grid_dims[1] < 1 || p->grid_dims[2] < 1 ||
                          p->cloud_per_planner))
        return FRP_ERR_ARG;
    const size_t lds = (size_t)3 * ((p->P + 63) / 64) * sizeof(uint64_t) + frp::CR_LIST * sizeof(uint32_t);
    const bool has_box = p->bbox[0] != 0.0 || p->bbox[1] != 0.0 || p->bbox[2] != 0.0;
    const bool grid = p->grid_start && has_box && !p->cloud_count;
    // production configuration (shared cloud with a grid, local box, N <= 64 = one lane per stage): one wavefront per planner;
    // planners it flags (more than a tile of points inside a seed ellipsoid, more than CS_PLANES cuts) go to the workgroup kernel
    // through the grid, and what THAT one flags (more in-box points than its LDS list) to the plain-cloud kernel.
    // FRP_CORRIDOR_WAVE=0 (experiments): workgroup kernels only
    static const bool wave_off = [] { const char *e = getenv("FRP_CORRIDOR_WAVE"); return e && e[0] == '0'; }();
    const bool wave = grid && !wave_off;
    if (wave) hipLaunchKernelGGL(frp::corridor_wave_kernel, dim3((unsigned)p->B), dim3(64), 0, static_cast<hipStream_t>(stream), *p);
    if (grid) hipLaunchKernelGGL(frp::corridor_kernel<true>, dim3((unsigned)p->B), dim3(frp::CR_THREADS), lds, static_cast<hipStream_t>(stream), *p, wave ? 1 : 0);
    hipLaunchKernelGGL(frp::corridor_kernel<false>, dim3((unsigned)p->B), dim3(frp::CR_THREADS), lds, static_cast<hipStream_t>(stream), *p, grid ? 1 : 0);
    return hipGetLastError() == hipSuccess ? FRP_OK : FRP_ERR_HIP;
}
