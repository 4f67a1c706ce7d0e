#!/usr/bin/env python3
"""The SB3-facing path: PDEVecEnv.step(numpy actions) -> numpy observations / rewards / dones / infos, host round trip included
(what PPO("MlpPolicy", venv).learn() drives).  python tools/bench_vecenv_host.py [B]   (bench.py reports the same number as
also.vecenv_host)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for _ in range(2):
    r = bench.vecenv_host_rate(torch.device("cuda", 0), B=B)
    print(f"PDEVecEnv.step, B={B}, nx=256, S=100 (numpy in, numpy out): {r['us_per_step']:.0f} us per step = {r['value']:.3g} env-steps/s "
          f"({r['host_bytes_per_step'] / 1e6:.1f} MB across PCIe per step)")
