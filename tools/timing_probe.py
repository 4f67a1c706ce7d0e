"""Developer probe: per-wave phase timestamps of the 1D step kernel (library built with -DPDEGYM_TIMING)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tim", "libpdegym_hip_timing.so")
import torch
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
wl = bench.Parabolic1D(torch.device("cuda", 0), 1, S=S)
wl.prepare(40)
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
ring = wl.env.t["ring"].cpu().numpy().view(np.uint32)[:, 116:122].astype(np.int64)
t0 = ring[:, 0] + (ring[:, 1] << 32)
t0 = t0 - t0.min()
load, loop, epi = ring[:, 2], ring[:, 3], ring[:, 4]
end = t0 + load + loop + epi
print(f"S={S} waves={len(t0)}  start: min {t0.min()} med {np.median(t0):.0f} max {t0.max()}   (s_memtime ticks)")
print(f"  load  med {np.median(load):.0f} max {load.max()}")
print(f"  loop  med {np.median(loop):.0f} max {loop.max()}")
print(f"  epilogue med {np.median(epi):.0f} max {epi.max()}")
print(f"  kernel span (first start -> last end): {end.max()} ticks; last start at {t0.max()}")
hist = np.histogram(t0, bins=8)
print("  start-time histogram:", hist[0].tolist(), [int(x) for x in hist[1]])
# per-XCD view (s_memtime bases differ between XCDs): cluster by start value
order = np.argsort(t0)
ts = t0[order]
gaps = np.nonzero(np.diff(ts) > 10_000_000)[0]
bounds = [0] + (gaps + 1).tolist() + [len(ts)]
for k in range(len(bounds) - 1):
    idx = order[bounds[k]:bounds[k + 1]]
    s = t0[idx] - t0[idx].min()
    e = s + load[idx] + loop[idx] + epi[idx]
    q = np.percentile(s, [0, 25, 50, 75, 100]).astype(int).tolist()
    print(f"  XCD-group {k}: waves {len(idx)} start pct[0,25,50,75,100]={q}  last end {e.max()}  first end {e.min()}")
