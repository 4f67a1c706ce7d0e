#!/usr/bin/env python3
"""TrafficPDE1D (reference notebook grid, M = 51, control_freq = 2, float64): env-steps as separate launches vs
pdegym_traffic_rollout (one launch per T steps), with commands given ahead and with an MLP policy in the loop.

    python tools/bench_traffic_rollout.py [B] [T]
"""
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_control_gym  # noqa: E402
from pde_control_gym import DeviceRollout, FusedMLP  # noqa: E402
from pde_control_gym.src import TrafficARZReward  # noqa: E402
from pdecontrolgym_amd.batch_traffic import TrafficBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)


def timed_graph(body, rewind, reps=5):
    rewind()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
        rewind()
        with torch.cuda.graph(g, stream=s):
            body()
    torch.cuda.current_stream().wait_stream(s)
    ts = []
    for _ in range(reps + 1):
        rewind()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    return ts[len(ts) // 2]


res = {"config": f"TrafficPDE1D M=51 f64 control_freq=2 B={B} T={T}, hipGraph replay"}
# ---- commands given ahead -------------------------------------------------------------------------------------------------
gen = torch.Generator().manual_seed(0)
rs = torch.tensor([0.115, 0.12, 0.125], dtype=torch.float64)[torch.randint(0, 3, (B,), generator=gen)]
qs = rs * 40 * (1 - rs / 0.16)
acts = ((torch.rand(T, B, 1, generator=gen, dtype=torch.float64) * 0.4 + 0.8) * qs[None, :, None]).to(dev)
outs = {}
for form in ("step_launches", "one_launch"):
    env = TrafficBatch(240, 0.25, 500, 10, "outlet", 40, 0.16, 60, True, 2, num_envs=B, device=dev)
    env.set_action_bounds(qs)
    env.reset(rs)
    obs = torch.zeros(T + 1, B, 2 * env.M, dtype=torch.float64, device=dev)
    rew = torch.zeros(T, B, dtype=torch.float64, device=dev)
    dn = torch.zeros(T, B, dtype=torch.uint8, device=dev)
    tr = torch.zeros(T, B, dtype=torch.uint8, device=dev)
    snap = {k: env.t[k].clone() for k in ("r", "y", "time")}

    def rewind():
        for k, v in snap.items():
            env.t[k].copy_(v)

    def body():
        if form == "one_launch":
            env.rollout(obs, acts, rew, dn, tr)
        else:
            for t in range(T):
                o, r, d, c = env.step(acts[t])
                obs[t + 1].copy_(o), rew[t].copy_(r), dn[t].copy_(d), tr[t].copy_(c)

    el = timed_graph(body, rewind)
    res["open_loop_" + form] = {"us_per_env_step": el / T * 1e6, "env_steps_per_s": B * T / el}
    outs[form] = (obs.clone(), rew.clone())
res["open_loop_bitwise_equal"] = all(torch.equal(a, b) for a, b in zip(outs["step_launches"], outs["one_launch"]))
# ---- policy in the loop (DeviceRollout) -------------------------------------------------------------------------------------
BASE = dict(T=240, dt=0.25, X=500, dx=10, v_steady=10, ro_steady=0.12, v_max=40, ro_max=0.16, tau=60)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(102, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1)).to(dev)
with torch.no_grad():
    net[-1].bias.fill_(4.6)
for name, one in (("fused_mlp_plus_step_launch", False), ("one_launch", True)):
    random.seed(0)
    venv = pde_control_gym.make_vec("PDEControlGym-TrafficPDE1D", num_envs=B, reward_class=TrafficARZReward(), simulation_type="outlet",
                                    limit_pde_state_size=True, control_freq=2, **BASE)
    venv.reset_tensor()
    ro = DeviceRollout(venv, FusedMLP(net), T, action_low=3.0, action_high=6.0, one_launch=one)
    ro.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ro.run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 5
    res["policy_" + name] = {"us_per_env_step": el / T * 1e6, "env_steps_per_s": B * T / el}
print(json.dumps(res))
