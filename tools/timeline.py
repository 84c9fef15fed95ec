"""Launch time line of the persistent-workgroup solver (needs the -DFRP_PROFILE library): when each solve starts and
ends on the 100 MHz wall clock, to see how much of the launch is the saturated phase and how much is the tail."""
import sys, numpy as np
sys.path.insert(0, '.')
from forces_resilient_planner_amd import solver, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workloads.config2(B, seed=workloads.SEED0 + 3)
for rep in range(2):
    z, fl, it, info = solver.solve_batch_host(w)
t0 = info[:, 6].min()
st = (info[:, 6] - t0) / 100.0; en = (info[:, 7] - t0) / 100.0   # microseconds
print(f"B {B}: span {en.max():.0f} us; mean its {it.mean():.2f}")
for q in (50, 90, 99, 99.9):
    print(f"  {q}% of the solves finished by {np.percentile(en, q):.0f} us")
active = lambda t: int(((st <= t) & (en > t)).sum())
print("  active solves over time:", " ".join(f"{t}us:{active(t)}" for t in range(0, int(en.max()) + 1, 250)))
last = np.argsort(-en)[:8]
for b in last:
    print(f"  problem {b}: start {st[b]:.0f} end {en[b]:.0f} its {it[b]} -> {1e3 * (en[b] - st[b]) / max(it[b], 1):.0f} ns/iter")
cyc = info[:, :6].sum(1) / np.maximum(it, 1)
print("  cycles/iter (clock64) mean", cyc.mean())
