"""Developer probe: where do the 4096 waves of the C2 launch land (XCC / SE / CU / SIMD) and how long do they run."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tim", "libpdegym_hip_timing.so")
import torch
import bench
wl = bench.Parabolic1D(torch.device("cuda", 0), 1)
wl.prepare(40)
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
ring = wl.env.t["ring"].cpu().numpy().view(np.uint32)[:, 116:123].astype(np.int64)
load, loop, epi, hw, xcc = ring[:, 2], ring[:, 3], ring[:, 4], ring[:, 5], ring[:, 6] & 0xf
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key_cu = xcc * 10000 + se * 1000 + sh * 100 + cu
per_cu = collections.Counter(key_cu.tolist())
per_simd = collections.Counter((key_cu * 10 + simd).tolist())
print("distinct CUs used:", len(per_cu), " waves per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
print("distinct SIMDs used:", len(per_simd), " waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
tot = load + loop + epi
for k in sorted(set(per_simd.values())):
    sel = np.array([per_simd[int(x)] == k for x in (key_cu * 10 + simd)])
    print(f"  SIMDs holding {k} waves: wave time med {np.median(tot[sel]):.0f} max {tot[sel].max()} ticks ({sel.sum()} waves)")
