"""Developer probe: per-phase s_memtime ticks of ns_tile_step.  Needs a library built with -DPDEGYM_TIMING:
    bash tools/build_variant.sh timing -DPDEGYM_TIMING && python tools/timing_probe_ns.py pdecontrolgym_amd/lib/ab/libtiming.so"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
wl = bench.WORKLOADS["ns2d_c4"](torch.device("cuda", 0), 1, S=K)
wl.prepare(20)
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
sc = wl.env.t["scratch"].cpu().numpy().view(np.uint32)[:, 2].reshape(wl.B, -1)[:, :5].astype(np.int64)
names = ["load+predictor", "bc + park u*, v* on chip + rhs", "jacobi", "store p + corrector + obs/reward", "block reduce + tail"]
for i, nm in enumerate(names):
    print(f"K={K} {nm:36s} med {np.median(sc[:, i]):9.0f}  min {sc[:, i].min():9d}  max {sc[:, i].max():9d} ticks")
print("total med", np.median(sc.sum(1)))
