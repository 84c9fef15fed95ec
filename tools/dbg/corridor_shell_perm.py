import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_parity as T
from forces_resilient_planner_amd import solver
cloud, ref, yaw, E = T._corridor_world(23, P=40000, B=2, tunnel=0.25)
consts = dict(seed_len=1.5, bbox=(2.0, 2.0, 1.0), inflation=1.1)
a = solver.corridor_batch_host(cloud, ref, yaw, E, F=64, consts=consts)
g = solver.corridor_batch_host(cloud, ref, yaw, E, F=64, consts=consts, grid_cell=0.5)
for p, k in ((0, 1), (0, 6), (1, 0)):
    m = a[3][p, k]
    G = np.c_[g[1][p, k, :m], g[2][p, k, :m]]; O = np.c_[a[1][p, k, :m], a[2][p, k, :m]]
    D = np.abs(G[:, None, :] - O[None, :, :]).max(axis=2)
    print(p, k, "row of the plain launch at which each row of the grid launch sits:", D.argmin(axis=1).tolist())
