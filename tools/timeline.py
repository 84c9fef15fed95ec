"""Launch time line of the persistent-workgroup solver (needs the -DFRP_PROFILE library): when each solve starts and
ends on the 100 MHz wall clock, to see how much of the launch is the saturated phase and how much is the tail."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forces_resilient_planner_amd import solver, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = workloads.config2(B, seed=workloads.SEED0 + 3)
import torch
ds = solver.DeviceSolver(B, w["N"], w["M"], int(w["nfaces"].max()), w["model"])
ds.upload(w)
for rep in range(2):
    ds.solve(); torch.cuda.synchronize()
it = ds.iters.cpu().numpy(); info = ds.info.cpu().numpy()
t0 = info[:, 6].min()
st = (info[:, 6] - t0) / 100.0; en = (info[:, 7] - t0) / 100.0   # microseconds
print(f"B {B}: span {en.max():.0f} us; mean its {it.mean():.2f}")
for q in (50, 90, 99, 99.9):
    print(f"  {q}% of the solves finished by {np.percentile(en, q):.0f} us")
active = lambda t: int(((st <= t) & (en > t)).sum())
print("  active solves over time:", " ".join(f"{t}us:{active(t)}" for t in range(0, int(en.max()) + 1, 100)))
last = np.argsort(-en)[:8]
for b in last:
    print(f"  problem {b}: start {st[b]:.0f} end {en[b]:.0f} its {it[b]} -> {1e3 * (en[b] - st[b]) / max(it[b], 1):.0f} ns/iter")
slots = 768 if os.environ.get("FRP_Q4") == "0" else 1024
idle = sum(max(0.0, en.max() - t) for t in sorted(en)[-slots:])  # slot-time between a slot's last solve and the end of the launch
print(f"  slot-time idle at the end of the launch: {idle / (slots * en.max()):.3f} of the launch; busy time per slot {((en - st).sum() / slots):.0f} us")
dur = (en - st) / np.maximum(it, 1)
print("  us per iteration by start time quartile:", [round(float(dur[(st >= a) & (st < b)].mean()), 1) for a, b in zip(np.percentile(st, [0, 25, 50, 75]), list(np.percentile(st, [25, 50, 75])) + [st.max() + 1])])
