import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdecontrolgym_amd.batch_tumor import TumorBatch
from pdecontrolgym_amd import _native as N
xs = np.linspace(0, 200, 201); ic = 0.8e5*np.exp(-0.25*xs**2)
for B in (64, 1024, 4096, 16384):
    eng = TumorBatch(600, 1, 200, 1, 61.2, num_envs=B)
    for mode, name in ((N.TUMOR_RUN_GROWTH, "growth"), (N.TUMOR_RUN_TO_END, "to_end")):
        eng.reset(ic); eng.advance(mode); torch.cuda.synchronize()
        eng.reset(ic); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.advance(mode); torch.cuda.synchronize(); el = time.perf_counter() - t0
        days = int(eng.t["time_index"][0])
        print(f"B={B} {name}: {days} days in {el*1e3:.2f} ms = {el/days*1e6:.2f} us/day, {B*days/el:.3g} patient-days/s")
    eng.reset(ic); torch.cuda.synchronize()
    z = torch.zeros(B, dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    for _ in range(100): eng.step(z)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"B={B} daily launches: {el/100*1e6:.2f} us/day")
