"""Per-wave phase cycles (profile build) for 30 corridor rows per stage -- the tick's solver variant (20, 10, re-reading) -- next to the 6-row headline."""
import ctypes, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from forces_resilient_planner_amd import solver, workloads
from forces_resilient_planner_amd import layout as L
from forces_resilient_planner_amd.workloads import _random_states, _line_reference, _bbox_faces, _rot, EGO, _finish, init_mpc_output, SEED0
from forces_resilient_planner_amd.adapter import ForcesAdapter

def rows30(B, seed=SEED0 + 3, nx=24):
    N, M, model = L.N_REF, 30, L.MODEL_NORMAL
    rng = np.random.default_rng(seed)
    st = _random_states(rng, B)
    ref_pos, ref_yaw, heading = _line_reference(rng, st, N)
    f_ext = rng.uniform(-3, 3, (B, 3))
    A6, b6 = _bbox_faces(ref_pos, np.repeat(heading[:, None], N, 1))
    nrm = rng.normal(size=(B, N, nx, 3)); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    off = rng.uniform(1.0, 3.0, (B, N, nx))
    bx = (nrm * ref_pos[:, :, None, :]).sum(-1) + off
    A = np.concatenate([A6, nrm], 2); b = np.concatenate([b6, bx], 2)
    nf = np.full((B, N), 6 + nx, dtype=np.int32)
    R = _rot(st[:, 6:9]); E1 = R @ (EGO[None, :, None] * np.swapaxes(R, -1, -2)); E = np.repeat(E1[:, None], N, 1)
    return _finish(ForcesAdapter(B, model, N, M), init_mpc_output(st, N), f_ext, ref_pos, ref_yaw, E, A, b, nf, model)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, w in (("6 rows", workloads.config2(B)), ("30 rows", rows30(B)), ("15 rows", rows30(B, nx=9))):
    buf = (ctypes.c_longlong * 96)()
    rd = solver.lib().frp_debug_read_prof_lds_mem if int(w["nfaces"].max()) > 15 else solver.lib().frp_debug_read_prof_lds
    solver.solve_batch_host(w); rd(buf)
    z, fl, it, info = solver.solve_batch_host(w); rd(buf)
    p = np.array(buf[:64]).reshape(4, 16); its = p[0, 10]
    names = ["eval->A", "predictor->C", "affine->D", "corrector->E", "stepA->F"]
    print(f"{name}: converged {np.mean(fl == 1):.3f} mean it {its / B:.2f}")
    for wv, role in enumerate(["riccati", "model", "bounds", "faces"]):
        print(f"  wave {wv} {role:8s}: " + "  ".join(f"{names[i]} {p[wv, i] / its:7.0f}|{p[wv, 5 + i] / its:7.0f}" for i in range(5)) + f"   total {(p[wv, :10].sum()) / its:8.0f}")
