"""Experiment (round 6): the SB3-facing face at the C2 shape with the step kernel writing the observations STRAIGHT into pinned host
memory (zero-copy, separate-state engine) against the shipped path (device observation + one pinned D2H copy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pdecontrolgym_amd import _native as N
from pdecontrolgym_amd.batch1d import PDEBatch1D, RewardSpec

B, nx, S = 4096, 256, 100
dx = 1.0 / nx; dt = 0.25 * dx * dx
kw = dict(T=1000 * S * dt, dt=dt, X=1, dx=dx, control_sample_rate=S * dt, control_type="Dirchilet", sensing_loc="full", sensing_type=None,
          normalize=True, max_control_value=20, limit_pde_state_size=True, max_state_value=1e10)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
init = (rng.uniform(1, 10, (B, 1)) * np.ones((1, nx + 1))).astype(np.float32)
beta = (50 * np.cos(8 * np.arccos(np.linspace(0, 1, nx + 1)))).astype(np.float32)

def run(zero_copy, steps=200):
    e = PDEBatch1D("parabolic", reward=RewardSpec(N.REWARD_TUNED1D, 1000 * S, -1e3, 3e2), num_envs=B, device=dev, state_in_obs=not zero_copy, **kw)
    e.reset(torch.tensor(init), torch.tensor(beta))
    e.enable_auto_reset(torch.tensor(init), keep_final_obs=True)
    a_pin = torch.zeros(B, dtype=torch.float32, pin_memory=True); a_dev = torch.zeros(B, device=dev)
    obs_pin = [torch.zeros(B, nx + 1, dtype=torch.float32, pin_memory=True) for _ in range(3)]
    pk_pin = torch.zeros(6 * B, dtype=torch.uint8, pin_memory=True)
    acts = rng.uniform(-1, 1, (steps + 10, B)).astype(np.float32)
    st = torch.cuda.current_stream(dev)
    def step(k):
        a_pin.numpy()[...] = acts[k]
        a_dev.copy_(a_pin, non_blocking=True)
        if zero_copy:
            o = obs_pin[k % 3]
            e.step(a_dev, out_obs=o, out_reward=pk_pin[:4 * B].view(torch.float32), out_terminated=pk_pin[4 * B:5 * B], out_truncated=pk_pin[5 * B:])
            st.synchronize()
            return o.numpy(), pk_pin.numpy()
        obs, r, te, tr = e.step(a_dev)
        o = obs_pin[k % 3]
        o.copy_(obs, non_blocking=True); pk_pin.copy_(e.host_pack, non_blocking=True)
        st.synchronize()
        return o.numpy(), pk_pin.numpy()
    for k in range(10): step(k)
    t0 = time.perf_counter()
    for k in range(10, 10 + steps): out = step(k)
    el = (time.perf_counter() - t0) / steps
    return el * 1e6, float(out[0].sum())

for rep in range(2):
    for zc in (False, True):
        us, chk = run(zc)
        print(f"zero_copy={zc}: {us:.1f} us per step of {B} envs  ({B / us * 1e6:.3g} env-steps/s)  checksum {chk:.6g}")
