// frp_kernels.hip -- gfx950 kernels around the NMPC solver (hand-written HIP, FP64) and the launcher.
//
//   * order_keys_kernel / order_bucket_kernel: launch order of the persistent solver workgroups (longest expected solve first);
//   * stage_eval_kernel: the batched model callback = FORCESNLPsolver_*_casadi2forces (casadi2forces.c:42-245) for B*N stage
//     points at once (HBM-bound; parity tool of SURVEY 8a-5, not on the solve path);
//   * launch_ipm: queue set-up + the solver kernel of frp_ipm_lds.hip, which replaces the reference's closed NLP solver
//     FORCESNLPsolver_{normal,final}_solve (FORCESNLPsolver_normal.h:323; forces_normal.cpp:139).
// (The round-1 single-wavefront solver with its HBM workspace lived here until round 3; it lost the same-box A/B by 34 %
// -- profiles/r02_ab_r01_vs_lds.json -- and no caller could reach it any more: the reference never passes more than 30
// corridor rows, forces_normal.cpp:114.)
#include <hip/hip_runtime.h>
#include <math.h>
#include "frp_model.hpp"
#include "../../include/frp_nmpc.h"
#include <vector>
#include "frp_kernels.h"
#include "frp_device.hpp"
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstdio>

namespace frp {


// ------------------------------------------------------------------ launch order: longest expected solve first
// The launch ends when its slowest problem does: at the BASELINE batch of 4096 the typical solve takes 5 interior-point
// iterations, one in a hundred takes 10..25, and a 25-iteration solve that starts in the second round of resident
// workgroups ends after everybody else has gone home.  Only the ORDER in which the persistent workgroups pull problems
// changes; every problem is solved exactly as before.
// Proxy for the work of a solve, from the caller's initial guess alone:
//     key = f(z0) * (1 + W_EQ * defect(z0)) * (1 + W_IN * violation(z0))
// f = objective (a plan that starts far from its reference takes more iterations), defect = max |x_{k+1} - rk2(z_k)|,
// |xinit - x_0| (an initial guess that is not a trajectory), violation = how far it leaves the corridor / the box.  The
// slow solves are the ones whose guess is infeasible AND expensive: of the 28 problems of configs[2] that need >= 10
// iterations, 25 are in the first 768 of this key (17 with the objective alone, round 2), and the oracle's iteration
// counts put through a list-scheduling model give makespan / ideal 1.11 instead of 1.16 (perfect knowledge: 1.07).
constexpr double ORDER_W_EQ = 40.0, ORDER_W_IN = 20.0;
// One wavefront holds as many problems as fit (lane = (problem, stage): three at the reference's N = 20); the per-problem
// sums / maxima are segmented shuffle reductions.
template <bool MAX>
__device__ __forceinline__ double segment_reduce(double v, int k, int N)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_down(v, off);
        if (k + off < N) v = MAX ? fmax(v, o) : v + o;
    }
    return v; // complete in the segment's first lane
}
__global__ __launch_bounds__(256) void order_keys_kernel(int B, int N, int M, int model, const int *__restrict__ models, const double *__restrict__ xinit,
                                                         const double *__restrict__ x0, const double *__restrict__ params,
                                                         const int *__restrict__ nfaces, double w_eq, double w_in, double *__restrict__ keys)
{
    const int per = 64 / N, lane = threadIdx.x & 63, seg = lane / N, k = lane - seg * N; // problems per wavefront; stage of this lane
    const int b = (blockIdx.x * 4 + (threadIdx.x >> 6)) * per + seg;
    const bool act = seg < per && b < B;
    const int np = NPRE + 4 * M;
    double c = 0.0, eq = 0.0, vi = 0.0;
    double zl[NZ];
    const size_t t = act ? (size_t)b * N + k : 0;
#pragma unroll
    for (int i = 0; i < NZ; i++) zl[i] = x0[t * NZ + i];
    const double *pk = params + t * np;
    if (act) {
        double p10[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
        c = stage_cost(zl, p10, stage_class(k, N), models ? models[b] : model, nullptr);
        // the box and the live corridor rows
#pragma unroll
        for (int i = 0; i < NZ; i++) vi = fmax(vi, fmax(lower_bound(i) - zl[i], zl[i] - upper_bound(i)));
        // (six rows per pass with clamped indices: the 24 loads of a pass are independent and in flight together)
        // (clamped to the rows the parameter block holds: a count beyond M is the solver kernel's PARAM_VALUE exit, and this
        // kernel runs before it -- it must not read past the stage's block on the way there)
        int nf = nfaces ? nfaces[t] : M;
        nf = nf < 0 ? 0 : (nf > M ? M : nf);
        for (int j0 = 0; j0 < nf; j0 += 6) {
            double ar[6][3], br[6];
#pragma unroll
            for (int jj = 0; jj < 6; jj++) {
                const int j = j0 + jj < nf ? j0 + jj : nf - 1;
                ar[jj][0] = pk[NPRE + 3 * j]; ar[jj][1] = pk[NPRE + 3 * j + 1]; ar[jj][2] = pk[NPRE + 3 * j + 2];
                br[jj] = pk[NPRE + 3 * M + j];
            }
#pragma unroll
            for (int jj = 0; jj < 6; jj++) vi = fmax(vi, ar[jj][0] * zl[8] + ar[jj][1] * zl[9] + ar[jj][2] * zl[10] - br[jj] - HU);
        }
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) eq = fmax(eq, fabs(xinit[(size_t)b * 9 + i] - zl[8 + i]));
        }
    }
    // dynamics defect against the next stage's [w; x] (lane + 1)
    double xn[9];
    rk2<false>(zl + 8, zl, pk + 3, xn, nullptr);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double nx = __shfl_down(zl[4 + i], 1);
        if (act && k < N - 1) eq = fmax(eq, fabs(zl[i] - nx));
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double nx = __shfl_down(zl[8 + i], 1);
        if (act && k < N - 1) eq = fmax(eq, fabs(xn[i] - nx));
    }
    c = segment_reduce<false>(c, k, N); eq = segment_reduce<true>(eq, k, N); vi = segment_reduce<true>(vi, k, N);
    if (act && k == 0) {
        const double key = c * (1.0 + w_eq * eq) * (1.0 + w_in * vi);
        keys[b] = (key == key && key < 1e300) ? key : 0.0;
    }
}
// The caller knows better: a receding-horizon tick re-solves yesterday's problem shifted by one stage, and the iteration
// count of the previous tick predicts this one's (frp_nmpc_batch.order_hint).  Unknown (<= 0) sorts with the typical solve.
__global__ __launch_bounds__(256) void order_hint_kernel(int B, const int *__restrict__ hint, double *__restrict__ keys)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) keys[b] = hint[b] > 0 ? (double)hint[b] : 4.0;
}
// order = the problems sorted by decreasing key, to bucket resolution (also zeroes the work-queue head): one workgroup, a 1024-bin counting sort on the
// (monotone) bit pattern of the non-negative keys -- i.e. on a log scale -- with the bins spread over the key range of
// this batch.  The order inside a bin is arbitrary (atomics); order[] is a permutation for any input.
__global__ __launch_bounds__(1024) void order_bucket_kernel(int B, const double *__restrict__ keys, int *__restrict__ order, int *__restrict__ counter, int *__restrict__ cu_slots, int head)
{
    // (the kernel is all latency: the keys of a batch of up to 4096 are read once and stay in registers, the extrema
    // go through one shuffle reduction per wavefront, and the prefix sum is a shuffle scan plus sixteen wave totals)
    __shared__ unsigned long long w_lo[16], w_hi[16];
    __shared__ int hist[1024], wtot[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) { counter[0] = head; counter[1] = 0; counter[2] = 0; } // queue head of the solve that follows on this stream (behind the `head` problems dealt out by counter[2]: KernelArgs::head_start); CUs its long solves have to themselves
    for (int i = t; i < CU_SLOT_ENTRIES; i += 1024) cu_slots[i] = 0;
    hist[t] = 0;
    constexpr int KR = 4; // keys per thread held in registers
    auto bits = [](double k) { return (unsigned long long)__double_as_longlong(k > 0.0 ? k : 0.0); };
    unsigned long long kr[KR];
#pragma unroll
    for (int j = 0; j < KR; j++) kr[j] = (t + j * 1024 < B) ? bits(keys[t + j * 1024]) : ~0ull;
    unsigned long long lo = ~0ull, hi = 0ull;
#pragma unroll
    for (int j = 0; j < KR; j++)
        if (t + j * 1024 < B) { lo = kr[j] < lo ? kr[j] : lo; hi = kr[j] > hi ? kr[j] : hi; }
    for (int i = t + KR * 1024; i < B; i += 1024) {
        const unsigned long long u = bits(keys[i]);
        lo = u < lo ? u : lo; hi = u > hi ? u : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
        lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
    }
    if (lane == 0) { w_lo[wv] = lo; w_hi[wv] = hi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; w++) { lo = w_lo[w] < lo ? w_lo[w] : lo; hi = w_hi[w] > hi ? w_hi[w] : hi; }
    const unsigned long long base = lo, span = hi - lo;
    const int shift = span < 1024ull ? 0 : 54 - __builtin_clzll(span); // smallest shift with (span >> shift) < 1024
    auto bin = [&](unsigned long long u) { return 1023 - (int)((u - base) >> shift); }; // bin 0 = largest keys
#pragma unroll
    for (int j = 0; j < KR; j++)
        if (t + j * 1024 < B) atomicAdd(&hist[bin(kr[j])], 1);
    for (int i = t + KR * 1024; i < B; i += 1024) atomicAdd(&hist[bin(bits(keys[i]))], 1);
    __syncthreads();
    { // exclusive prefix sum over the 1024 bins
        const int own = hist[t];
        int v = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(v, off);
            if (lane >= off) v += up;
        }
        if (lane == 63) wtot[wv] = v;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) before += w < wv ? wtot[w] : 0;
        hist[t] = before + v - own;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KR; j++)
        if (t + j * 1024 < B) order[atomicAdd(&hist[bin(kr[j])], 1)] = t + j * 1024;
    for (int i = t + KR * 1024; i < B; i += 1024) order[atomicAdd(&hist[bin(bits(keys[i]))], 1)] = i;
}

// ------------------------------------------------------------------ batched model callback
// One thread per (problem, stage): the reference's extfunc for B*N stage points (casadi2forces.c:42-245).  HBM-bound:
// 147 doubles in, 282 out per point, 221 of them the dense 13 x 17 Jacobian (column-major, ld 13, as the reference's
// sparse2fullcopy writes it).  A thread that stored its own Jacobian would touch 64 cache lines per store instruction, so
// the Jacobian leaves through LDS: every thread parks the 51 entries of its compact linearisation, then the wavefront
// writes the points' Jacobians one after the other as runs of 64 consecutive doubles (source of each dense entry: a
// compile-time table into the parked record, whose slots 51 / 52 / 53 hold 0, 1 and dt).
constexpr int SE_SLOTS = 55; // 51 + constants (0, 1, dt), odd stride
constexpr int SE_MAXM = 64;  // corridor rows staged per point
__host__ __device__ constexpr int jc_source(int e) // dense entry e = col * 13 + row  ->  slot of the parked record
{
    const int col = e / 13, row = e % 13;
    if (row >= 9) return col == row - 9 ? 52 : 51;
    const int bi = row / 3, ii = row % 3;
    if (col < 4) { // lin_B(row, col)
        if (col == 3) return bi == 0 ? 36 + ii : (bi == 1 ? 39 + ii : 51);
        if (bi == 1) return 42 + ii * 3 + col;
        if (bi == 2) return ii == col ? 53 : 51;
        return 51;
    }
    if (col < 8) return 51;
    const int j = col - 8, bj = j / 3, jj = j % 3; // lin_A(row, j)
    if (bi == 0) return bj == 0 ? (ii == jj ? 52 : 51) : (bj == 1 ? 0 : 9) + ii * 3 + jj;
    if (bi == 1) return bj == 0 ? 51 : (bj == 1 ? 18 : 27) + ii * 3 + jj;
    return (bj == 2 && ii == jj) ? 52 : 51;
}
struct JcTable {
    unsigned char v[256];
};
constexpr JcTable make_jc_table()
{
    JcTable t{};
    for (int e = 0; e < 256; e++) t.v[e] = (unsigned char)(e < 221 ? jc_source(e) : 51);
    return t;
}
__device__ const JcTable g_jc_table = make_jc_table();

__global__ __launch_bounds__(64) void stage_eval_kernel(int B, int N, int M, int model, const double *__restrict__ z,
                                                          const double *__restrict__ params, double *__restrict__ f,
                                                          double *__restrict__ gf, double *__restrict__ c,
                                                          double *__restrict__ Jc, double *__restrict__ h)
{
    __shared__ double s_lin[1][64 * SE_SLOTS]; // one wavefront per workgroup: 31 KB of LDS each, five per CU
    __shared__ double s_pos[64 * 3]; // one point's corridor rows [A (3 M) | b (M)]
    const size_t total = (size_t)B * N;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool act = t < total;
    const int k = act ? (int)(t % N) : 0;
    const int np = NPRE + 4 * M;
    const double *zk = z + (act ? t : 0) * NZ, *pk = params + (act ? t : 0) * np;
    double zl[NZ], p10[NPRE];
#pragma unroll
    for (int i = 0; i < NZ; i++) zl[i] = zk[i];
#pragma unroll
    for (int i = 0; i < NPRE; i++) p10[i] = pk[i];
    const int sc = stage_class(k, N);
    if (act && (f || gf)) {
        double g[NZ];
        const double cost = stage_cost(zl, p10, sc, model, g);
        if (f) f[t] = cost;
        if (gf) {
#pragma unroll
            for (int i = 0; i < NZ; i++) gf[t * NZ + i] = g[i];
        }
    }
    double *mine = s_lin[wv] + lane * SE_SLOTS;
    s_pos[lane * 3] = zl[8]; s_pos[lane * 3 + 1] = zl[9]; s_pos[lane * 3 + 2] = zl[10];
    if (c || Jc) {
        if (act && sc != STAGE_LAST) {
            Lin L;
            double xn[9];
            rk2<true>(zl + 8, zl, p10 + 3, xn, &L);
            if (c) {
#pragma unroll
                for (int i = 0; i < 9; i++) c[t * 13 + i] = xn[i];
#pragma unroll
                for (int i = 0; i < 4; i++) c[t * 13 + 9 + i] = zl[i];
            }
            const double *Lc = reinterpret_cast<const double *>(&L);
#pragma unroll
            for (int i = 0; i < 51; i++) mine[i] = Lc[i];
            mine[51] = 0.0; mine[52] = 1.0; mine[53] = DT;
        } else {
            if (act && c) for (int i = 0; i < 13; i++) c[t * 13 + i] = 0.0;
            for (int i = 0; i < 54; i++) mine[i] = 0.0; // the last stage has no dynamics: a zero Jacobian
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const size_t t0 = t - lane; // first point of this wavefront
    const int npts = t0 < total ? (int)((total - t0) < 64 ? (total - t0) : 64) : 0;
    if (Jc) {
        const int src0 = g_jc_table.v[lane], src1 = g_jc_table.v[64 + lane], src2 = g_jc_table.v[128 + lane], src3 = g_jc_table.v[192 + lane];
        for (int p = 0; p < npts; p++) {
            const double *rec = s_lin[wv] + p * SE_SLOTS;
            double *J = Jc + (t0 + p) * 221;
            J[lane] = rec[src0];
            J[64 + lane] = rec[src1];
            J[128 + lane] = rec[src2];
            if (lane < 221 - 192) J[192 + lane] = rec[src3];
        }
    }
    if (h && M > 0) {
        // corridor rows h = A pos - b, eight points at a time: the wavefront reads their [A | b] (4 M consecutive doubles each)
        // with coalesced loads -- sixteen in flight per lane -- into the LDS area the Jacobian pass has just vacated, then
        // writes the 8 M results, consecutive in memory, with coalesced stores
        if (M <= SE_MAXM) {
            constexpr int PB = 8;
            double *ab = s_lin[0]; // [PB][4 SE_MAXM]
            static_assert(PB * 4 * SE_MAXM <= 64 * SE_SLOTS, "staging area");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int p0 = 0; p0 < npts; p0 += PB) {
                const int nb = npts - p0 < PB ? npts - p0 : PB;
                double r0[PB], r1[PB], r2[PB], r3[PB];
#pragma unroll
                for (int q = 0; q < PB; q++) {
                    const double *src = params + (t0 + p0 + (q < nb ? q : 0)) * np + NPRE;
                    r0[q] = lane < 4 * M ? src[lane] : 0.0;
                    r1[q] = lane + 64 < 4 * M ? src[lane + 64] : 0.0;
                    r2[q] = lane + 128 < 4 * M ? src[lane + 128] : 0.0;
                    r3[q] = lane + 192 < 4 * M ? src[lane + 192] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < PB; q++) {
                    double *dst = ab + q * 4 * SE_MAXM;
                    dst[lane] = r0[q]; dst[lane + 64] = r1[q]; dst[lane + 128] = r2[q]; dst[lane + 192] = r3[q];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double *out = h + (t0 + p0) * M;
                for (int o = lane; o < nb * M; o += 64) {
                    const int q = o / M, j = o - q * M;
                    const double *a = ab + q * 4 * SE_MAXM, *ps = s_pos + (p0 + q) * 3;
                    out[o] = a[3 * j] * ps[0] + a[3 * j + 1] * ps[1] + a[3 * j + 2] * ps[2] - a[3 * M + j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (act) {
            const double *A = pk + NPRE, *bb = pk + NPRE + 3 * M;
            for (int j = 0; j < M; j++) h[t * M + j] = A[3 * j] * zl[8] + A[3 * j + 1] * zl[9] + A[3 * j + 2] * zl[10] - bb[j];
        }
    }
}

// ------------------------------------------------------------------ launchers
// workspace = [work-queue counter, 256 B][per-CU arrival counters, 8 KB][keys: B doubles][order: B ints][packed P of the resident
// workgroups of the Q4 solver variants: 20 x 92 doubles each -- 15 MB for the 1024 workgroups of a full chip, L2-resident]
// (the rest of the solver's per-iteration state lives in LDS and registers: nothing per problem in HBM)
constexpr int QUEUE_RESERVED = 32 + CU_SLOT_ENTRIES / 2; // doubles
static int resident_cap(int per_cu)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int cap = cus * per_cu;
    if (const char *e = getenv("FRP_RESIDENT_SLOTS")) { // tuning knob
        const int v = atoi(e);
        if (v > 0) cap = v;
    }
    return cap;
}
static bool q4_shape(int N, int MF) { return lds_q4_enabled() && N <= 20 && MF <= lds_q4_max_rows(); } // (frp_ipm_lds.hip: q4_covers also looks at the options)
static bool q30_shape(int N, int MF) { return lds_q30_enabled() && N > 20 && N <= 30 && MF <= 16; }
static size_t pws_doubles(int B, int N, int MF)
{
    if (q30_shape(N, MF)) { const int cap = resident_cap(3); return (size_t)(B <= cap ? B : cap) * lds_q30_pws_doubles_per_slot() + 16; }
    if (!q4_shape(N, MF)) return 0;
    const int cap = resident_cap(4);
    return (size_t)(B <= cap ? B : cap) * lds_q4_pws_doubles_per_slot() + 16; // (+ 128 bytes: the blocks start on a cache line)
}
size_t ws_bytes(int B, int N, int MF)
{
    return (QUEUE_RESERVED + (size_t)B + ((size_t)B + 1) / 2 + pws_doubles(B, N, MF)) * sizeof(double);
}

__global__ void reset_counter_kernel(int *counter, int *cu_slots)
{
    if (threadIdx.x == 0) { counter[0] = 0; counter[1] = 0; counter[2] = 0; }
    for (int i = threadIdx.x; i < CU_SLOT_ENTRIES; i += blockDim.x) cu_slots[i] = 0;
}

static int lds_resident_slots(int B, const KernelArgs &k)
{
    const int cap = resident_cap(lds_workgroups_per_cu(k));
    // every resident workgroup is used: the queue is pulled dynamically, so evening the slots over the "rounds" of the batch
    // (what the single-wave kernel does) only lowers the residency -- measured 1.58 vs 1.71 ms at B = 4096 (768 vs 683)
    return B <= cap ? B : cap;
}

struct OrderWeights {
    double w[2];
    OrderWeights() : w{ORDER_W_EQ, ORDER_W_IN}
    {
        if (const char *e = getenv("FRP_ORDER_W")) { // "w_eq,w_in": tuning knob
            double a = -1.0, b = -1.0;
            if (sscanf(e, "%lf,%lf", &a, &b) == 2 && a >= 0.0 && b >= 0.0) { w[0] = a; w[1] = b; }
        }
    }
};
static double order_weight(int i)
{
    static const OrderWeights ow; // (initialised once, thread-safe)
    return ow.w[i];
}

// ---- measurement hook: hipEvent pairs around the solver kernel of the launches between begin and end
namespace {
struct KernelTimer {
    bool on = false;
    int used = 0, stride = 1, seen = 0;
    std::vector<hipEvent_t> ev; // 2 per launch
    void clear()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear(); used = 0; seen = 0; stride = 1; on = false;
    }
} g_timer;
}
hipError_t kernel_timing_begin(int max_launches, int stride)
{
    if (max_launches <= 0 || stride <= 0 || g_timer.on) return hipErrorInvalidValue;
    g_timer.clear();
    g_timer.stride = stride;
    g_timer.ev.reserve(2 * (size_t)max_launches);
    for (int i = 0; i < 2 * max_launches; i++) {
        hipEvent_t e = nullptr;
        // (timing enabled, no system-scope fence: a plain event makes the preceding kernel's writes host-visible when it is
        // recorded -- a cache write-back between the kernels of a launch that the unobserved run does not have)
        const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
        if (rc != hipSuccess) { g_timer.clear(); return rc; }
        g_timer.ev.push_back(e);
    }
    g_timer.on = true;
    return hipSuccess;
}
hipError_t kernel_timing_end(float *avg_ms, int *launches)
{
    if (!g_timer.on || !avg_ms || !launches) return hipErrorInvalidValue;
    double total = 0.0;
    hipError_t rc = hipSuccess;
    for (int i = 0; i < g_timer.used && rc == hipSuccess; i++) {
        float ms = 0.f;
        rc = hipEventSynchronize(g_timer.ev[2 * i + 1]);
        if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, g_timer.ev[2 * i], g_timer.ev[2 * i + 1]);
        total += ms;
    }
    *launches = g_timer.used;
    *avg_ms = g_timer.used > 0 ? (float)(total / g_timer.used) : 0.f;
    g_timer.clear();
    return rc;
}

hipError_t launch_ipm(const KernelArgs &a, hipStream_t stream)
{
    if (!lds_kernel_supports(a.N, a.MF)) return hipErrorInvalidValue; // (fill_args rejects these before they get here)
    KernelArgs k = a;
    double *q = a.ws;
    k.pws = nullptr;
    if (q4_shape(a.N, a.MF) || q30_shape(a.N, a.MF)) {
        const uintptr_t p = reinterpret_cast<uintptr_t>(q + QUEUE_RESERVED + (size_t)a.B + ((size_t)a.B + 1) / 2);
        k.pws = reinterpret_cast<double *>((p + 127) & ~(uintptr_t)127);
    }
    int slots = lds_resident_slots(a.B, k);
    if (a.slot_reserve > 0) { // room for another kernel's workgroups beside the persistent ones (the pipelined host path's gather)
        int lim = resident_cap(lds_workgroups_per_cu(k)) - a.slot_reserve;
        if (lim < 64) lim = 64;
        if (slots > lim) slots = lim;
    }
    k.counter = reinterpret_cast<int *>(q);
    k.cu_slots = reinterpret_cast<int *>(q + 32);
    // Long solves get their CU to themselves (frp_ipm_lds.hip, Q4 variants): from iteration `iso_it` on the other workgroups of the CU
    // finish what they have and wait -- the launch ends with its longest solve, and a solve alone on a CU iterates a quarter faster.
    // At most `iso_cap` CUs at a time (a workload of long solves must not idle the chip).  FRP_ISO_IT=0 switches it off.
    static const int iso_env = [] { const char *e = getenv("FRP_ISO_IT"); return e ? atoi(e) : 12; }();
    k.iso_it = a.B > 1 ? iso_env : 0;
    k.iso_cap = resident_cap(1) / 16;
    k.order = nullptr;
    k.head_start = 0;
    if (a.B > slots) { // more problems than resident workgroups: order the queue, longest expected solve first
        k.self_reset = 0;
        double *keys = q + QUEUE_RESERVED;
        int *order = reinterpret_cast<int *>(keys + a.B);
        k.order = order;
        if (a.order_hint)
            hipLaunchKernelGGL(order_hint_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, stream, a.B, a.order_hint, keys);
        else
            hipLaunchKernelGGL(order_keys_kernel, dim3((unsigned)((a.B + 4 * (64 / a.N) - 1) / (4 * (64 / a.N)))), dim3(256), 0, stream, a.B, a.N, a.M, a.model,
                               a.models, a.xinit, a.x0, a.params, a.nfaces, order_weight(0), order_weight(1), keys);
        // The longest expected solves get a CU each FROM THE START (Q4 variants; FRP_HEAD_START=0 switches it off): the launch ends with its
        // longest solve -- configs[2]: one 25-iteration problem of 4096, 8th in this order --, a problem alone on a CU iterates a sixth faster
        // (76.6 k against 90.9 k cycles), and the isolation by iteration count (iso_it) only sees it at iteration 12.
        static const int head_env = [] { const char *e = getenv("FRP_HEAD_START"); return e ? atoi(e) : 0; }(); // (default off: measured 0.852 -> 0.861 / 0.868 ms at 8 / 16 -- the launch does not end with its longest solve, profiles/r06_head_start.txt)
        if (lds_workgroups_per_cu(k) == 4 && k.iso_it > 0 && head_env > 0) k.head_start = std::min(std::min(head_env, k.iso_cap), slots / 8);
        hipLaunchKernelGGL(order_bucket_kernel, dim3(1), dim3(1024), 0, stream, a.B, keys, order, k.counter, k.cu_slots, k.head_start);
    } else if (a.self_reset && a.B == 1) {
        k.cu_slots = nullptr; // (the caller zeroed the queue head once; the kernel puts it back)
    } else {
        k.self_reset = 0;
        // a one-thread kernel rather than hipMemsetAsync: as a node of a captured hipGraph the 4-byte memset was not
        // ordered before the solve on the graph's first launch (ROCm 7.2; tools/graph_tick.py), which left the queue
        // exhausted and the previous outputs in place
        hipLaunchKernelGGL(reset_counter_kernel, dim3(1), dim3(256), 0, stream, k.counter, k.cu_slots);
    }
    const bool timed = g_timer.on && (g_timer.seen++ % g_timer.stride) == 0 && 2 * (size_t)g_timer.used + 1 < g_timer.ev.size();
    if (timed) (void)hipEventRecord(g_timer.ev[2 * g_timer.used], stream);
    const hipError_t rc = launch_ipm_lds(k, slots, stream);
    if (timed) { (void)hipEventRecord(g_timer.ev[2 * g_timer.used + 1], stream); g_timer.used++; }
    return rc;
}

hipError_t launch_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *gf, double *c, double *Jc, double *h, hipStream_t stream)
{
    const size_t total = (size_t)B * N;
    const int blocks = (int)((total + 63) / 64);
    hipLaunchKernelGGL(stage_eval_kernel, dim3(blocks), dim3(64), 0, stream, B, N, M, model, z, params, f, gf, c, Jc, h);
    return hipGetLastError();
}

} // namespace frp

