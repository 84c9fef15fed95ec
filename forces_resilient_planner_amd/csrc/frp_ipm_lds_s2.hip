// frp_ipm_lds_s2.hip -- fifth translation unit of the LDS-resident solver: the small-launch variants.  A launch of at most two problems per CU (the drop-in
// FORCESNLPsolver_*_solve call is the case of ONE) has the registers of a whole SIMD for two wavefronts: the same kernels as frp_ipm_lds.hip's (20, 2) and (20, 5)
// variants, plain and twisted, compiled for two wavefronts per SIMD -- 256 registers instead of 168, no spills, no reloads on the chain -- together with their own
// copies of the sweeps (a function behind a call takes the tightest register budget among its callers: in the main translation unit that is 168).
// build.py compiles this unit with the sweeps INLINED (-DFRP_INLINE_FACTOR -DFRP_INLINE_SWEEPS: at 256 registers there is room, and the gather tables become loop invariants
// held in registers).  Drop-in call 0.1268 -> 0.1206 ms, 0.1147 -> 0.1080 ms with the twisted solve (profiles/r06_park.txt).  Contributes frp::launch_ipm_lds_s2.
#define FRP_LDS_S2_TU
#include "frp_ipm_lds.hip"
