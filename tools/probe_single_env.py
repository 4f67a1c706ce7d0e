"""Where the fixed cost of a batch-of-one env.step() goes (S = 1 transport shape): launch call, stream synchronisation, Python."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_single as bs
import pde_control_gym
from pde_control_gym.src.environments1d.base_env_1d import classify_control

env_id, grid = bs.SHAPES["transport_s1"]
p, n, beta = bs._params(env_id, grid)
env = pde_control_gym.make(env_id, device="cuda", record_history=False, **p).unwrapped
env.reset()
core = env._core
a = np.array([0.3], dtype=np.float32)
for _ in range(200):
    env.step(a)
N = 3000
def t(fn, n=N):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6
call = core._host_call()
st = torch.cuda.current_stream(core.device)
print("env.step                         %.2f us" % t(lambda: env.step(a)))
env.reset()
print("core.step_host                   %.2f us" % t(lambda: core.step_host(0.3, 0)))
env.reset()
print("prepared call + sync             %.2f us" % t(lambda: (call(), st.synchronize())))
env.reset()
def launch_only():
    call()
x = t(launch_only, 200); st.synchronize()
print("prepared call only (200 queued)  %.2f us" % x)
env.reset()
print("core._host_call() lookup         %.2f us" % t(lambda: core._host_call()))
print("classify_control                 %.2f us" % t(lambda: classify_control(a)))
print("stream lookup + sync (idle)      %.2f us" % t(lambda: torch.cuda.current_stream(core.device).synchronize()))
print("obs_to_user                      %.2f us" % t(lambda: env._obs_to_user()))
