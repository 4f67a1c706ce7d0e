import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pdecontrolgym_amd.batch2d import NSBatch2D
BC = {"upper": ["Controllable", "Dirchilet"], "lower": ["Dirchilet", "Dirchilet"], "left": ["Dirchilet", "Dirchilet"], "right": ["Dirchilet", "Dirchilet"]}
def run(nocol, K, n=21, B=5, dtype=torch.float64, steps=1):
    os.environ["PDEGYM_NS_NO_COL"] = "1" if nocol else "0"
    nt = 50
    env = NSBatch2D(T=0.05, dt=1e-3, X=1, dx=1/(n-1), Y=1, dy=1/(n-1), boundary_condition=BC, U_ref=np.zeros((nt, n, n, 2)),
                    action_ref=2*np.ones(nt), gamma=0.1, maximum_pressure_iteration=K, num_envs=B, device="cuda", dtype=dtype)
    rng = np.random.default_rng(0)
    u0 = rng.normal(size=(B, n, n)); v0 = rng.normal(size=(B, n, n)); p0 = rng.normal(size=(B, n, n))
    env.reset(u0, v0, p0)
    a = torch.tensor(rng.uniform(2, 4, (B, 1)), dtype=dtype, device="cuda")
    for _ in range(steps):
        obs, r, te = env.step(a)
    torch.cuda.synchronize()
    return obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy()
for K in (0, 1, 2, 5):
    a = run(False, K); b = run(True, K)
    print("K", K, "obs diff", np.abs(a[0]-b[0]).max(), "p diff", np.abs(a[1]-b[1]).max(), "reward diff", np.abs(a[2]-b[2]).max())
    if K == 2:
        d = np.abs(a[1]-b[1])[0]
        idx = np.argwhere(d > 0)
        print(len(idx), idx[:30].tolist())
        print("rows with diffs", sorted(set(idx[:,0].tolist())), "cols", sorted(set(idx[:,1].tolist())))
