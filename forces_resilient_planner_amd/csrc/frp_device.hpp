// frp_device.hpp -- gfx950 device helpers shared by the solver kernels: DPP cross-lane moves and reductions,
// wave-level LDS hand-off fences, FP64 MFMA tile products, the 4x4 pivot-block factorisations.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace frp {

// ------------------------------------------------------------------ wave helpers
// Cross-lane reductions on DPP (no LDS round trips): butterflies inside each 16-lane row with quad_perm / row_half_mirror
// / row_mirror, then the four row results are combined through v_readlane.  Every lane ends up with the result.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double lane_read(double v, int src) // wave-uniform src lane
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
struct OpSum { static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); } };
struct OpMin { static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };
template <class Op>
__device__ __forceinline__ double row16_reduce(double v) // all 16 lanes of a row get the row result
{
#ifdef FRP_SHFL_REDUCE // debugging aid: the same butterflies through ds_bpermute
    for (int o = 1; o < 16; o <<= 1) v = Op::f(v, __shfl_xor(v, o));
    return v;
#endif
    v = Op::f(v, dpp_move<0xB1>(v));  // quad_perm [1,0,3,2]
    v = Op::f(v, dpp_move<0x4E>(v));  // quad_perm [2,3,0,1]
    v = Op::f(v, dpp_move<0x141>(v)); // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    v = Op::f(v, dpp_move<0x140>(v)); // row_mirror: halves of the row
    return v;
}
template <class Op>
__device__ __forceinline__ double wave_reduce(double v)
{
    v = row16_reduce<Op>(v);
    return Op::f(Op::f(lane_read(v, 0), lane_read(v, 16)), Op::f(lane_read(v, 32), lane_read(v, 48)));
}
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<OpMax>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<OpMin>(v); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<OpSum>(v); }
// sum over the 16 lanes of one row group (lanes with equal lane >> 4)
__device__ __forceinline__ double row16_sum(double v) { return row16_reduce<OpSum>(v); }
__device__ __forceinline__ double lane_bcast(double v, int src) // wave-uniform src lane
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// One wavefront per workgroup.  LDS operations of one wave are executed in issue order, so lanes can hand
// data to each other through LDS with only a COMPILER ordering fence: WSYNC() emits no instruction and,
// unlike __syncthreads(), does not drain the vector-memory counter -- prefetched global loads and
// streamed stores stay in flight across it.  FULLSYNC() (= __syncthreads()) is used at phase boundaries
// where lanes exchange data through global memory.
#define WSYNC()                                                   \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
#define FULLSYNC() __syncthreads()

// 1 / x for normal, finite x (pivots, slacks, multipliers: all strictly positive and far from the denormal range):
// hardware reciprocal seed (relative error 4.6e-8 on gfx950, tools/ubench/rcp_f64.hip) + ONE third-order step
// r (1 + e + e^2), e = 1 - x r: residual e^3 ~ 1e-22, i.e. correctly rounded to ~1 ulp in 4 instructions instead of the
// 11 of the IEEE division expansion (no scaling / fix-up of denormal, infinite or NaN operands).
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// symmetric positive definite 4x4 inverse via LDL'; returns false if a pivot is not positive (results then undefined)
// (optionally also the factors: m = L^-1 (unit lower triangular, row-major 4x4) and dinv = diag(D)^-1, R = m' D^-1 m)
__device__ __forceinline__ bool spd4_inverse(const double *a /*row-major 4x4, lower part used*/, double *r /*16*/,
                                             double *mout = nullptr /*16*/, double *dinv = nullptr /*4*/)
{
    const double a00 = a[0], a11 = a[5], a21 = a[9], a22 = a[10];
    const double a31 = a[13], a32 = a[14], a33 = a[15];
    // The inputs are wave-uniform (v_readlane results, scalar registers) and a VALU instruction reads at most one scalar
    // operand: the first column enters every product below, so it is copied to vector registers ONCE -- left alone the
    // compiler copies an operand per instruction (12 copies per call instead of 3).
    double a10 = a[4], a20 = a[8], a30 = a[12];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(a10), "+v"(a20), "+v"(a30));
#endif
    const double d0 = a00;
    const double i0 = fast_rcp(d0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = a11 - l10 * a10;
    const double i1 = fast_rcp(d1);
    const double t21 = a21 - l20 * a10, t31 = a31 - l30 * a10;
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = a22 - l20 * a20 - l21 * t21;
    const double i2 = fast_rcp(d2);
    const double t32 = a32 - l30 * a20 - l31 * t21;
    const double l32 = t32 * i2;
    const double d3 = a33 - l30 * a30 - l31 * t31 - l32 * t32;
    const double i3 = fast_rcp(d3);
    // inverse of unit lower L: m = L^-1
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -l20 - l21 * m10;
    const double m31 = -l31 - l32 * m21;
    const double m30 = -l30 - l31 * m10 - l32 * m20;
    // R = m' D^-1 m
    const double r33 = i3;
    const double r32 = m32 * i3, r31 = m31 * i3, r30 = m30 * i3;
    const double r22 = i2 + m32 * r32;
    const double r21 = m21 * i2 + m32 * r31;
    const double r20 = m20 * i2 + m32 * r30;
    const double r11 = i1 + m21 * m21 * i2 + m31 * r31;
    const double r10 = m10 * i1 + m21 * m20 * i2 + m31 * r30;
    const double r00 = i0 + m10 * m10 * i1 + m20 * m20 * i2 + m30 * r30;
    r[0] = r00; r[1] = r10; r[2] = r20; r[3] = r30;
    r[4] = r10; r[5] = r11; r[6] = r21; r[7] = r31;
    r[8] = r20; r[9] = r21; r[10] = r22; r[11] = r32;
    r[12] = r30; r[13] = r31; r[14] = r32; r[15] = r33;
    if (mout) {
        mout[0] = 1.0; mout[1] = 0.0; mout[2] = 0.0; mout[3] = 0.0;
        mout[4] = m10; mout[5] = 1.0; mout[6] = 0.0; mout[7] = 0.0;
        mout[8] = m20; mout[9] = m21; mout[10] = 1.0; mout[11] = 0.0;
        mout[12] = m30; mout[13] = m31; mout[14] = m32; mout[15] = 1.0;
        dinv[0] = i0; dinv[1] = i1; dinv[2] = i2; dinv[3] = i3;
    }
    // branch-free: a non-positive (or NaN) pivot poisons the results, which the caller discards
    return (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
}

// LDL' factors of a symmetric positive definite 4x4 block (lower part of row-major a): m6 = the strictly lower
// triangle of m = L^-1 in the order (1,0) (2,0) (2,1) (3,0) (3,1) (3,2), dinv = 1 / diag(D).  Branch-free; returns
// false if a pivot is not positive (results then undefined).
__device__ __forceinline__ bool ldl4(const double *a, double *m6, double *dinv)
{
    const double a00 = a[0], a11 = a[5], a21 = a[9], a22 = a[10];
    const double a31 = a[13], a32 = a[14], a33 = a[15];
    double a10 = a[4], a20 = a[8], a30 = a[12]; // (one copy to vector registers each: see spd4_inverse)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(a10), "+v"(a20), "+v"(a30));
#endif
    const double d0 = a00;
    const double i0 = fast_rcp(d0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = a11 - l10 * a10;
    const double i1 = fast_rcp(d1);
    const double t21 = a21 - l20 * a10, t31 = a31 - l30 * a10;
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = a22 - l20 * a20 - l21 * t21;
    const double i2 = fast_rcp(d2);
    const double t32 = a32 - l30 * a20 - l31 * t21;
    const double l32 = t32 * i2;
    const double d3 = a33 - l30 * a30 - l31 * t31 - l32 * t32;
    double i3 = fast_rcp(d3);
    // (i3 is only read by the lanes of row group 3: left alone, the compiler sinks the reciprocal into an EXEC-masked region --
    // a dozen scalar instructions and a branch on the serial path of every stage to save four VALU instructions)
    asm volatile("" : "+v"(i3));
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -l20 - l21 * m10;
    const double m31 = -l31 - l32 * m21;
    const double m30 = -l30 - l31 * m10 - l32 * m20;
    m6[0] = m10; m6[1] = m20; m6[2] = m21; m6[3] = m30; m6[4] = m31; m6[5] = m32;
    dinv[0] = i0; dinv[1] = i1; dinv[2] = i2; dinv[3] = i3;
    return (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
}

// Explicit global address space: inside non-inlined device functions a plain double* is a GENERIC
// pointer and compiles to flat_load/flat_store, which also count on lgkmcnt and therefore make every
// LDS wait drain the in-flight prefetches.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double cgdouble;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------ 16x16 FP64 tiles in registers
// Tile X: lane l = 16 g + c holds x[r] = X[4r + g][c], r = 0..3 (the C/D layout of
// v_mfma_f64_16x16x4_f64; A operand of slice s = X'[.., 4s+g] i.e. again x[s], B operand = x[s]).
typedef double d4 __attribute__((ext_vector_type(4)));

// D = X' Y + C
__device__ __forceinline__ d4 mm_tn(const d4 x, const d4 y, d4 c)
{
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], c, 0, 0, 0);
    return c;
}
// D = X[0:4,:]' Y[0:4,:] + C  (only the first four rows of X and Y contribute)
__device__ __forceinline__ d4 mm_tn4(double x0, double y0, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, c, 0, 0, 0); }

// ---- mat-vec products on v_mfma_f64_4x4x4_4b (4 independent 4x4x4 blocks, 24 cycles instead of the 64 of the
// 16x16x4 instruction; layout probed in tools/ubench/mfma_f64_4x4.hip): for block b
//     A[i][k] in lane 16k + 4b + i,   B[k][j] in lane 16k + 4b + j,   D[i][j] in lane 16i + 4b + j.
// "V layout" of a 16-vector: lane l holds x[4 qI + qk] with qk = l >> 4, qI = (l >> 2) & 3 (replicated over qj = l & 3).
// y = A x + c with block b = output rows 4b..4b+3; in step m block b contracts columns 4((b+m)&3).. with the input
// rotated by m quads inside each 16-lane row (DPP row_ror), so that the output comes out in V layout again:
//     A-operand register m, lane l  <->  A[4 qI + qj][4 ((qI + m) & 3) + qk]      (gather tables TAB4_*)
template <int M>
__device__ __forceinline__ double quad_rot(double v) // result in quad q <- quad (q + M) & 3 of the same 16-lane row
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, 0x120 + (16 - 4 * M), 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), 0x120 + (16 - 4 * M), 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// Sum over the 64 lanes on the matrix pipe: with B = 1 the 4x4x4 product sums the four lanes 16 k + 4 b + i over k, a second
// one the four partial sums of a block over i, and two quad rotations add the four blocks -- 2 MFMAs + 6 VALU instructions
// instead of the 23 of the DPP / readlane reduction, and the MFMAs cost the SIMD's VALU nothing (the element-wise waves
// share their SIMDs with Riccati waves that are short of exactly that).  The result is in every lane.  Summation order:
// ((x_l + x_{l+16}) + x_{l+32}) + x_{l+48} per lane column, then rows of four, then blocks -- fixed, like the DPP order.
__device__ __forceinline__ double wave_sum_mx(double x)
{
    double d = mfma4(x, 1.0, 0.0);
    d = mfma4(d, 1.0, 0.0);
    d += quad_rot<1>(d);
    d += quad_rot<2>(d);
    return d;
}
__device__ __forceinline__ double matvec4(const d4 &A, double x, double c)
{
    const double x1 = quad_rot<1>(x), x2 = quad_rot<2>(x), x3 = quad_rot<3>(x);
    double d = mfma4(A[0], x, c);
    d = mfma4(A[1], x1, d);
    d = mfma4(A[2], x2, d);
    return mfma4(A[3], x3, d);
}

} // namespace frp
