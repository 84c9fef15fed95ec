import sys, numpy as np
sys.path.insert(0, '/root/repo')
from forces_resilient_planner_amd import solver, workloads
w = workloads.config2(4, seed=workloads.SEED0 + 3)
z, fl, it, info = solver.solve_batch_host(w, solver.default_options(maxit=0))
x0 = w["x0"]; p = w["params"]; N = 20
def stage_cost(zk, pk, first):
    wwp, win, wr, yr = pk[6], pk[7], pk[8], pk[9]
    c = wwp*((pk[0:3]-zk[8:11])**2).sum() + 12*wwp*(yr-zk[16])**2 + win*((zk[0:3]/(np.pi/2))**2).sum() + wr*((zk[0:4]-zk[4:8])**2).sum()
    if first: c += 10*win*(zk[4:7]**2).sum()
    return c
for b in range(2):
    ref = np.array([stage_cost(x0[b,k], p[b,k], k==0) for k in range(N)])
    got = z[b,:,16]
    print("b", b, "obj gpu", info[b,4], "sum lanes", got.sum(), "ref", ref.sum())
    print(" diff per stage", np.round(got-ref, 4))
