"""C2 shape: step time against the number of sub-steps S (loop cost per sub-step = slope; intercept = load/ramp/epilogue)
and against the batch B.  Run on an MI355X: python tools/c2_slope.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for B in (1024, 2048, 4096, 8192):
    for S in (25, 50, 100, 200, 400):
        wl = bench.Parabolic1D(dev, 1, B=B, S=S)
        r = bench.run_workload(wl, 200, 20, 1, graph=True, repeats=3)
        print(f"B={B:6d} S={S:4d}  {r['step_ms_events']*1e3:8.2f} us/step", flush=True)
