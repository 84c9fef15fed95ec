"""Which dense-cloud case differs between the grid launch (wave + shell kernels) and the plain-cloud launch (workgroup kernel)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_parity as T
from forces_resilient_planner_amd import solver
def run(name, cloud, ref, yaw, E, consts=None):
    a = solver.corridor_batch_host(cloud, ref, yaw, E, F=64, consts=consts)
    for cell in (0.5, 0.23):
        g = solver.corridor_batch_host(cloud, ref, yaw, E, F=64, consts=consts, grid_cell=cell)
        same = [bool(np.array_equal(x, y)) for x, y in zip(a, g)]
        print(name, cell, "pi/A/b/nf/cnt equal:", same, flush=True)
        if not all(same):
            pi, A, b, nf, cnt = a
            for p in range(ref.shape[0]):
                for k in range(abs(cnt[p])):
                    m = min(nf[p, k], 64)
                    if not (np.array_equal(A[p, k], g[1][p, k]) and np.array_equal(b[p, k], g[2][p, k])):
                        G = np.c_[g[1][p, k, :m], g[2][p, k, :m]]; O = np.c_[A[p, k, :m], b[p, k, :m]]
                        D = np.abs(G[:, None, :] - O[None, :, :]).max(axis=2)
                        print("  planner", p, "poly", k, "rows", m, "g rows", g[3][p, k], "set-match", sorted(D.argmin(axis=0)) == list(range(m)), "max min-dist", D.min(axis=0).max(),
                              "first differing row", int(np.argmax(np.any(G != O, axis=1))))
w = T._corridor_world
c = w(23, P=40000, B=2, tunnel=0.25); run("seed1.5", *c, consts=dict(seed_len=1.5, bbox=(2.0, 2.0, 1.0), inflation=1.1))
c = w(24, P=50000, B=2, grid=0.12); run("voxel", *c)
cloud, ref, yaw, E = w(25, P=6000, B=2)
rng = np.random.default_rng(25)
v = rng.normal(size=(3000, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
run("sphere", np.r_[cloud, ref[0, 0] + np.array([0.05, 0.0, 0.0]) + 0.9 * v], ref, yaw, E)
