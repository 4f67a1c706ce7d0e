"""A/B of the column-per-lane NavierStokes2D kernel: python tools/ab_ns_col.py [lib.so | -]  -- 21 x 21, K = 2000 timings at three batch sizes in both
dtypes, then a bitwise comparison with the workgroup kernel on 600 random instances (profiles/r04_ab_f64_division.txt, last entry)."""
import os, sys, time
sys.path.insert(0, '.')
from pdecontrolgym_amd import _native as N
lib = sys.argv[1] if len(sys.argv) > 1 else "-"      # an alternative build of the library (tools/build_variant.sh), or "-" for the in-tree one
if lib != '-': N.LIB_PATH = os.path.abspath(lib)
import numpy as np, torch
sys.path.insert(0, 'tools')
import bench_ns_example as be
for dtype in (torch.float64, torch.float32):
    for B in (3072, 8192, 32768):
        a = be.run(B, False, dtype=dtype)
        print(f"{lib:36s} 21x21 K=2000 {str(dtype)[6:]} B={B}: {a*1e3:.3f} ms/step", flush=True)
# correctness: column kernel vs workgroup kernel bitwise on random fields
from pdecontrolgym_amd.batch2d import NSBatch2D
n, K, B = 21, 57, 600
nt = 50
rng = np.random.default_rng(0)
outs = []
for no_col in (0, 1):
    be._dbg("DEBUG_NS_NO_COL", no_col)
    be._dbg("DEBUG_NS_COL_MIN_BATCH", 0)
    env = NSBatch2D(T=0.05, dt=1e-3, X=1, dx=1/(n-1), Y=1, dy=1/(n-1), boundary_condition=be.BC, U_ref=np.zeros((nt, n, n, 2)),
                    action_ref=2*np.ones(nt), gamma=0.1, maximum_pressure_iteration=K, num_envs=B, device="cuda", dtype=torch.float64)
    r = np.random.default_rng(1)
    env.reset(r.normal(size=(B, n, n)), r.normal(size=(B, n, n)), r.normal(size=(B, n, n)))
    a = torch.tensor(r.uniform(2, 4, (B, 1)), device="cuda")
    for _ in range(3):
        o = env.step(a)
    torch.cuda.synchronize()
    outs.append((o[0].cpu().numpy().copy(), env.t["p"].cpu().numpy().copy(), o[1].cpu().numpy().copy()))
print(lib, "bitwise vs workgroup kernel:", all(np.array_equal(x, y) for x, y in zip(outs[0][:2], outs[1][:2])), np.abs(outs[0][2]-outs[1][2]).max())
