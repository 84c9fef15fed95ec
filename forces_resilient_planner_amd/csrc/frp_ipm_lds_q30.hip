// frp_ipm_lds_q30.hip -- fourth translation unit of the LDS-resident solver: horizons of up to 30 stages at THREE problems per CU (BASELINE configs[3]).
// Same sources as frp_ipm_lds.hip on the Q30 record layout (215 doubles per stage: packed P_k in the L2-resident workspace like Q4, T' and p in the slots of the
// consumed Hessian, P d aliased onto the p slots, no trig hand-over slots, the external force from the parameters) on four-wavefront workgroups with the
// lane == stage model phase: 3 x 52 KB of LDS and 12 waves at 168 VGPRs per CU.  Contributes frp::launch_ipm_lds_q30.
#define FRP_LDS_Q30_TU
#include "frp_ipm_lds.hip"
