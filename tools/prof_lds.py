"""Per-wave work / barrier-wait cycles of the LDS-resident kernel (library built with -DFRP_PROFILE, selected through FRP_LIB):
   tools/build_variant.sh prof -DFRP_PROFILE && FRP_LIB=forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py [B] [config]"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from forces_resilient_planner_amd import solver, workloads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = sys.argv[2] if len(sys.argv) > 2 else "2"
w = {"1": workloads.config1, "2": workloads.config2, "3": workloads.config3}[cfg](B)
twist = int(sys.argv[3]) if len(sys.argv) > 3 else 0
opt = solver.default_options(twist=twist)
buf = (ctypes.c_longlong * 96)()
# (the Q4 variants are a translation unit of their own with their own counters: FRP_Q4=0 in the environment keeps the launch on the three-per-CU variants)
q4 = os.environ.get("FRP_Q4", "1") != "0" and cfg in ("1", "2") and not twist and B > 768
q30 = os.environ.get("FRP_Q30", "1") != "0" and cfg == "3" and not twist and (B > 1792 or os.environ.get("FRP_Q30_MIN_B") == "0")  # the three-per-CU variant of round 6 (its own counters)
reader = solver.lib().frp_debug_read_prof_lds_q4 if q4 else (solver.lib().frp_debug_read_prof_lds_q30 if q30 else solver.lib().frp_debug_read_prof_lds)
solver.solve_batch_host(w, opt)
reader(buf)
z, fl, it, info = solver.solve_batch_host(w, opt)
reader(buf)
p = np.array(buf[:64]).reshape(4, 16); sg = np.array(buf[64:96])
its = p[0, 10]
names = ["eval->A", "predictor->C", "affine->D", "corrector->E", "stepA->F"]
print(f"B {B} cfg {cfg}: solves {B}, iterations {its} (mean {its / B:.2f}); cycles per iteration, per wave: work before the barrier | wait at it")
for wv, role in enumerate(["riccati", "model+f", "bounds"] if q4 else ["riccati", "model", "bounds", "faces"]):
    print(f"  wave {wv} {role:8s}: " + "  ".join(f"{names[i]} {p[wv, i] / its:7.0f}|{p[wv, 5 + i] / its:7.0f}" for i in range(5)) +
          f"   total {(p[wv, :10].sum()) / its:8.0f}   init/problem {p[wv, 11] / max(p[wv, 12], 1):7.0f}")
print("  factor sweep segments (cycles per iteration): " + "  ".join(f"{n} {sg[i] / its:6.0f}" for i, n in enumerate(["mfma X/G", "gather", "pivot", "tail mfma", "P update+stores", "loop"])))
print("  whole sweeps (cycles per iteration; forward and forward+y are cumulative with the sweep before them): " + "  ".join(f"{n} {sg[8 + i] / its:6.0f}" for i, n in enumerate(["factor", "+forward", "backvec", "+forward y"])))
print("  model phase segments (cycles per iteration): " + "  ".join(f"{n} {sg[16 + i] / its:6.0f}" for i, n in enumerate(["park y, dx0", "trig1+accel1+J1", "trig2+accel2", "J2 J1 products", "d + shifts", "gm"])))
if q4 and sg[16:21].sum() > 0:  # -DFRP_PROFILE_W1
    print("  model + corridor wave, evaluation phase (cycles per iteration, lane 0's clock; timer reads drain the LDS queue): " +
          "  ".join(f"{n} {sg[16 + i] / its:6.0f}" for i, n in enumerate(["model phase", "corridor rows + sums", "model: park y + trig + step", "model: d", "model: linearisation + M'y"])))
if q4:
    print("  Riccati wave on SIMD 0..3 / role = wave index (workgroups of both launches): " + " ".join(str(int(v)) for v in sg[24:29]))
if twist:
    tn = {22: "factor (half)", 23: "arrive", 6: "wait B (riccati)", 7: "wait B (model)", 24: "meet factor+solve (riccati)", 25: "meet factor+solve (model)",
          26: "forward (half)", 27: "backsub", 28: "backvec (half)", 29: "arrive vec", 12: "wait B' (riccati)", 13: "wait B' (model)",
          14: "meet solve (riccati)", 15: "meet solve (model)", 30: "forward (half)", 31: "backsub"}
    print(f"  twisted solve (m = {twist}), cycles per iteration: " + "  ".join(f"{n} {sg[i] / its:6.0f}" for i, n in tn.items()))
