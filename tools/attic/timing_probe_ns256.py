"""Phase timestamps of the fused 256 x 256 step (library built with -DPDEGYM_NS256_TIMING, tools/build_variant.sh):
per wave s_memtime ticks (100 MHz) for front / pressure load / sweeps / back, and the spread of start and end times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd import _native as N
N.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch, bench

B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
wl = bench.WORKLOADS["ns2d_c5"](torch.device("cuda", 0), 1, B=B)
wl.prepare(8)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
sc = wl.env.t["scratch"].view(torch.int32).reshape(B, -1)[:, :64].cpu().numpy().astype(np.int64).reshape(B, 8, 8)
d = sc[:, :, :4] & 0xffffffff
names = ["front", "p load", "sweeps", "store+back"]
tick_us = 0.01
for i, n in enumerate(names):
    print(f"{n:12s} mean {d[:, :, i].mean() * tick_us:8.1f} us   min {d[:, :, i].min() * tick_us:8.1f}   max {d[:, :, i].max() * tick_us:8.1f}")
t0 = sc[:, :, 4] & 0xffffffff
t1 = sc[:, :, 5] & 0xffffffff
base = t0.min()
print("start spread (us):", np.percentile((t0 - base) * tick_us, [0, 25, 50, 75, 100]).round(1))
print("end   spread (us):", np.percentile((t1 - base) * tick_us, [0, 25, 50, 75, 100]).round(1))
first = (t0[:, 0] - base) * tick_us < 50
print("first-generation workgroups:", int(first.sum()), " their total (us):", ((t1 - t0)[first].mean() * tick_us).round(1),
      " later ones:", ((t1 - t0)[~first].mean() * tick_us).round(1) if (~first).any() else None)
