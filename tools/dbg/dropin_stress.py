"""The drop-in call 200 000 times in a row (no stream synchronisation between calls: the completion word is spun on): latency of the first and
the last thousand calls, resident memory before and after -- the runtime must not pile up anything per un-synchronised launch."""
import ctypes, os, sys, time, resource
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from forces_resilient_planner_amd import solver, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
w0 = workloads.config0()
p = solver.ForcesParams(); o = solver.ForcesOutput(); info = solver.ForcesInfo()
p.xinit[:] = w0["xinit"][0]; p.x0[:] = w0["x0"][0].ravel(); p.all_parameters[:] = w0["params"][0].ravel(); p.num_of_threads = 1
f = solver.lib().FORCESNLPsolver_normal_solve
def burst(k):
    t = time.perf_counter()
    for _ in range(k):
        fl = f(ctypes.byref(p), ctypes.byref(o), ctypes.byref(info), None, None)
        assert fl == 1
    return (time.perf_counter() - t) / k * 1e6
burst(100)
r0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
first = burst(1000)
burst(n - 2000)
last = burst(1000)
r1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"{n} calls: first thousand {first:.1f} us per call, last thousand {last:.1f} us, max resident set {r0 / 1024:.0f} -> {r1 / 1024:.0f} MB, pobj {info.pobj:.9f}")
