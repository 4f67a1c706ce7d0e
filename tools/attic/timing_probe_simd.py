"""Developer probe: per-SIMD view of the C2 launch -- start offsets and loop durations of the 4 waves sharing a SIMD."""
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
t0 = ring[:, 0] + (ring[:, 1] << 32)
load, loop, epi, hw, xcc = ring[:, 2], ring[:, 3], ring[:, 4], ring[:, 5], ring[:, 6] & 0xf
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
groups = collections.defaultdict(list)
for i, k in enumerate(key.tolist()):
    groups[k].append(i)
rows = []
for k, idx in groups.items():
    idx = np.array(idx)
    o = np.argsort(t0[idx])
    idx = idx[o]
    st = t0[idx] - t0[idx].min()
    en = st + load[idx] + loop[idx] + epi[idx]
    rows.append(np.concatenate([st, loop[idx], en]))
rows = np.array([r for r in rows if len(r) == 12])
print("SIMDs with 4 waves:", len(rows))
m = np.median(rows, axis=0).astype(int)
print("median start offsets (by start order):", m[0:4].tolist())
print("median loop ticks    (by start order):", m[4:8].tolist())
print("median end ticks     (by start order):", m[8:12].tolist())
print("SIMD busy span: median", int(np.median(rows[:, 8:12].max(axis=1))), "max", int(rows[:, 8:12].max()))
inst = np.arange(len(key))
print("block -> waves of one block land on SIMDs:", [int(simd[i]) for i in range(8)], "CUs:", [int(cu[i]) for i in range(8)])
# per-XCD spans
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    s = t0[sel] - t0[sel].min()
    e = s + load[sel] + loop[sel] + epi[sel]
    print(f"XCD {x}: waves {sel.sum()} first start 0, last start {s.max()}, last end {e.max()}")
