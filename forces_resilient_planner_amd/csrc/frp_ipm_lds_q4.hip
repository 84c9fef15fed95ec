// frp_ipm_lds_q4.hip -- third translation unit of the LDS-resident solver: the FOUR-problems-per-CU variants.  Same sources as
// frp_ipm_lds.hip on the Q4 record layout (241 doubles per stage instead of 309: packed P_k in an L2-resident workspace in global
// memory, T' and p in the slots of the consumed Hessian) and on workgroups of three wavefronts (Riccati; model + a bound round;
// faces + the other bound rounds): 4 x 40 KB of LDS and 12 waves at 168 VGPRs per CU.  Contributes frp::launch_ipm_lds_q4.
#define FRP_LDS_Q4_TU
#include "frp_ipm_lds.hip"
