"""Calibration workload for FETCH_SIZE / WRITE_SIZE in THIS code's access pattern (4 bytes per lane, coalesced planar
maps): one ef_op_transform_maps over a 2048x1536 vertex+normal map = 6 planar float reads + 6 planar float writes per
pixel, every byte touched exactly once: 75 497 472 B read, 75 497 472 B written.  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_calibrate.py      (and again with WRITE_SIZE)
and compare the k_transform_maps row with those byte counts (tools/pmc_traffic.sh does both)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticfusion_amd import api
cols, rows = 2048, 1536
rng = np.random.RandomState(0)
v = rng.uniform(0.5, 2.0, size=(3 * rows, cols)).astype(np.float32)
n = rng.uniform(-1, 1, size=(3 * rows, cols)).astype(np.float32)
R = np.eye(3, dtype=np.float32)
t = np.zeros(3, np.float32)
for _ in range(3):
    api.ops.transform_maps(v, n, R, t)
print("calibration bytes per launch: read", v.nbytes + n.nbytes, "written", v.nbytes + n.nbytes)
