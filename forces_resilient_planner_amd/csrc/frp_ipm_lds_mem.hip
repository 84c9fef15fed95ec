// frp_ipm_lds_mem.hip -- second translation unit of the LDS-resident solver: the kernel variants that re-read the corridor rows
// from the parameters (FREG = false: (20, 10), (32, 15), (64, 30)), compiled WITHOUT the code-generation flags of build.py's
// CODEGEN_FLAGS (see the end of frp_ipm_lds.hip).  Contributes frp::launch_ipm_lds_mem.
#define FRP_LDS_MEM_TU
#include "frp_ipm_lds.hip"
