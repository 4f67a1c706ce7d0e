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
xcc = wl.env.t["ring"].cpu().numpy().view(np.uint32)[:, 122].astype(np.int64) & 0xf
hw = wl.env.t["ring"].cpu().numpy().view(np.uint32)[:, 121].astype(np.int64)
ghz = 2.1
for x in sorted(set(xcc.tolist())):
    idx = np.nonzero(xcc == x)[0]
    last = t0[idx].max()
    idx = idx[t0[idx] > last - 200_000]            # the waves of the last launch on this XCD
    s = t0[idx] - t0[idx].min()
    e = s + load[idx] + loop[idx] + epi[idx]
    q = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).astype(int).tolist()
    print(f"  XCD {x}: waves {len(idx)}  start {q(s)}  load {q(load[idx])}  loop {q(loop[idx])}  epi {q(epi[idx])}  end {q(e)}"
          f"  span {e.max() / ghz / 1e3:.2f} us")
    # per-SIMD wave counts on this XCD: HW_ID bits: wave_id[3:0] simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
    simd = (hw[idx] >> 4) & 3; cu = (hw[idx] >> 8) & 0xf; se = (hw[idx] >> 13) & 7
    key = (se * 16 + cu) * 4 + simd
    cnt = np.bincount(key)
    cnt = cnt[cnt > 0]
    print(f"          waves per SIMD: hist {np.bincount(cnt).tolist()}  (SIMDs used {len(cnt)})")
# the four waves of a few SIMDs: (start offset, load, loop, wave slot) in start order
x = 7
idx = np.nonzero(xcc == x)[0]
idx = idx[t0[idx] > t0[idx].max() - 200_000]
base = t0[idx].min()
simd = (hw[idx] >> 4) & 3; cu = (hw[idx] >> 8) & 0xf; se = (hw[idx] >> 13) & 7
key = (se * 16 + cu) * 4 + simd
for kk in sorted(set(key.tolist()))[:12]:
    sel = idx[key == kk]
    sel = sel[np.argsort(t0[sel])]
    print(f"  SIMD {kk:4d}: " + "  ".join(f"[inst {i} start {t0[i]-base} load {load[i]} loop {loop[i]} slot {hw[i]&0xf}]" for i in sel))

raw = wl.env.t["ring"].cpu().numpy().view(np.uint32).astype(np.int64)
r0, rd = raw[:, 123], raw[:, 124]
cyc = load + loop + epi
lastl = r0 > r0.max() - 5000          # 50 us window: the last launch (100 MHz ticks)
print(f"  realtime: waves in last launch {lastl.sum()}  start spread {(r0[lastl].max() - r0[lastl].min()) / 100:.2f} us"
      f"  span (first start -> last end) {((r0 + rd)[lastl].max() - r0[lastl].min()) / 100:.2f} us")
print(f"  clock (cycles / realtime) per wave: median {np.median(cyc[lastl] / (rd[lastl] / 100.0)) / 1e3:.3f} GHz")
st = (r0[lastl] - r0[lastl].min()) / 100.0
print("  start offsets us pct[0,10,50,90,100]:", np.percentile(st, [0, 10, 50, 90, 100]).round(2).tolist())
en = ((r0 + rd)[lastl] - r0[lastl].min()) / 100.0
print("  end offsets us pct[0,10,50,90,100]:", np.percentile(en, [0, 10, 50, 90, 100]).round(2).tolist())
karg = raw[:, 125]
print("  kernarg wait cycles pct[0,10,50,90,100]:", np.percentile(karg[lastl], [0, 10, 50, 90, 100]).astype(int).tolist(),
      " row-load wait (load - kernarg):", np.percentile((load - karg)[lastl], [0, 10, 50, 90, 100]).astype(int).tolist())
