"""Throughput of the mixed-precision (float64 beta / float64 control) parity mode next to the float32 kernels, C2 shape."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for b64 in (False, True):
    for B in (4096, 16384):
        wl = bench.Parabolic1D(torch.device("cuda", 0), 1, B=B)
        if b64:
            wl.beta = wl.beta.double()
        r = bench.run_workload(wl, 50, 5, 1, graph=True, repeats=3)
        print(f"beta {'float64' if b64 else 'float32'} B={B}: {r['step_ms_events']*1e3:.1f} us/step  {B/r['step_ms_events']/1e3:.1f} M env-steps/s", flush=True)
