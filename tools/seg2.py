"""Read the g_prof device counters (library built with -DFRP_PROFILE); one problem => per-solve totals."""
import sys, ctypes, numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import solver, workloads
w = workloads.config2(1)
z, fl, it, info = solver.solve_batch_host(w)
lib = solver.lib()
hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
sym = ctypes.c_void_p(); sz = ctypes.c_size_t()
# hipGetSymbolAddress needs the host-side symbol; use hipModule-less path: read via exported helper if present
get = getattr(lib, 'frp_debug_read_prof', None)
buf = (ctypes.c_longlong * 24)()
get(buf)
names = ["aff:sync0", "aff:body", "aff:(unused)", "aff:reduce", "aff:sync1", "-", "step:sync0", "step:body", "step:sync1", "-", "-", "-", "f0","f1","f2","f3","f4","f5","bv:looptop", "bv:X+G issue", "bv:E issue", "bv:stage next", "bv:E wait+pn+stores", "bv:vmcnt wait", "fwd:v1 prep", "fwd:mfma0+vmwait", "fwd:stage+D1", "fwd:du wait", "fwd:dz+D2 issue", "fwd:v wait"]
print("iterations", it[0])
for n, v in zip(names, buf):
    print(f"{n:12s} {v / max(it[0],1):10.0f} cycles/iter")
