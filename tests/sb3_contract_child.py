"""Child process of tests/test_sb3_contract.py: runs with tests/stubs on sys.path, so ``stable_baselines3`` and ``gymnasium`` resolve
to the contract stand-ins and pde_control_gym's batched faces inherit from them exactly as they would from the real packages.

Usage: python tests/sb3_contract_child.py double|hip          (prints CONTRACT-OK on success)

What it checks is what the reference's caller needs (examples/transportPDE/transport1Dppo.py:77-90: ``PPO("MlpPolicy", env)``;
examples/reactionDiffusionPDE/ParabolicPDEExample.ipynb:192 "Wrapping the env in a DummyVecEnv"):
  1. the type gate of BaseAlgorithm._wrap_env lets the batched environments through AS THEY ARE (no DummyVecEnv around them),
     rejects a look-alike that is not a VecEnv, and accepts the single environments as gymnasium.Env;
  2. VecEnv bookkeeping: constructor state, seed / set_options are consumed by reset, VecEnvWrapper stacks on top, unwrap walks;
  3. an on-policy collect_rollouts-shaped loop -- actions [n_envs, action_dim], step, info-buffer update from infos[i]["episode"],
     the time-limit bootstrap reads infos[i]["terminal_observation"] / ["TimeLimit.truncated"], rollout storage keeps what it was
     handed -- runs over episode ends, and every observation / reward / flag equals B single environments stepped one by one;
  4. the gymnasium.vector face is a gymnasium.vector.VectorEnv with batched spaces and same-step final observations.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "stubs"), ROOT]

import numpy as np  # noqa: E402


def main(kind):
    import gymnasium
    import stable_baselines3
    from stable_baselines3.common.base_class import WouldWrapInDummyVecEnv, _wrap_env
    from stable_baselines3.common.vec_env import VecEnv, VecEnvWrapper, VecMonitor, is_vecenv_wrapped, unwrap_vec_wrapper
    assert stable_baselines3.__version__.endswith("contract-stub") and gymnasium.__version__.endswith("contract-stub")

    import pde_control_gym
    from pde_control_gym import _compat
    from pde_control_gym.src import TransportPDE1D, TunedReward1D
    from pde_control_gym.vector_tumor import TumorVecEnv
    assert _compat.HAVE_SB3 and _compat.HAVE_GYMNASIUM and _compat.HAVE_GYMNASIUM_VECTOR
    assert issubclass(pde_control_gym.PDEVecEnv, VecEnv) and issubclass(TumorVecEnv, VecEnv)
    assert issubclass(pde_control_gym.GymnasiumVectorAdapter, gymnasium.vector.VectorEnv)
    assert issubclass(TransportPDE1D, gymnasium.Env)

    if kind == "double":
        from tests.fake_backend import FakeBackend
        bk = lambda: dict(device="cpu", backend=FakeBackend())  # noqa: E731
    else:
        import torch
        assert torch.cuda.is_available(), "hip run without a GPU"
        bk = lambda: dict(device="cuda")  # noqa: E731

    B, T, dt, nx = 5, 0.04, 1e-4, 100                     # control_sample_rate 0.01 -> 100 sub-steps, 4 env-steps per episode
    beta = (5 * np.cos(7.35 * np.arccos(np.linspace(0, 1, nx)))).astype(np.float32)
    ics = [np.linspace(1.0, 2.0 + k, nx).astype(np.float32) for k in range(B)]

    def params(init):
        return {"T": T, "dt": dt, "X": 1, "dx": 1e-2, "reward_class": TunedReward1D(int(round(T / dt)), -1e3, 3e2),
                "normalize": False, "sensing_loc": "full", "control_type": "Dirchilet", "sensing_type": None,
                "sensing_noise_func": lambda state: state, "limit_pde_state_size": True, "max_state_value": 1e10,
                "max_control_value": 20, "control_sample_rate": 0.01, "reset_init_condition_func": init,
                "reset_recirculation_func": lambda nx: beta}

    import itertools
    cyc = itertools.cycle(ics)                            # instance b restarts from the same row every episode (B draws per round)
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **bk(), **params(lambda nx: next(cyc)))

    # 1. the type gate ------------------------------------------------------------------------------------------------
    assert _wrap_env(venv) is venv

    class LookAlike:                                      # the round-3 face: right methods, wrong type
        num_envs, observation_space, action_space = B, venv.observation_space, venv.action_space
        reset, step_async, step_wait = venv.reset, venv.step_async, venv.step_wait
    try:
        _wrap_env(LookAlike())
        raise AssertionError("a duck-typed VecEnv passed the gate")
    except ValueError as e:
        assert "not a Gymnasium environment" in str(e)
    single = gymnasium.make("PDEControlGym-TransportPDE1D", **bk(), **params(lambda nx: ics[0]))
    try:
        _wrap_env(single)
        raise AssertionError
    except WouldWrapInDummyVecEnv:
        pass                                              # a gymnasium.Env: SB3 would wrap it into DummyVecEnv([lambda: env])

    # 2. VecEnv bookkeeping -----------------------------------------------------------------------------------------------
    assert venv.num_envs == B and len(venv.reset_infos) == B and venv.render_mode is None
    assert venv.metadata == {"render_modes": []}
    assert venv.seed(7) == [7 + i for i in range(B)] and venv._seeds[2] == 9
    venv.set_options({"k": 1})
    assert venv._options == [{"k": 1}] * B
    mon = VecMonitor(venv)
    assert isinstance(mon, VecEnvWrapper) and mon.unwrapped is venv and venv.unwrapped is venv
    assert is_vecenv_wrapped(mon, VecMonitor) and unwrap_vec_wrapper(mon, VecMonitor) is mon
    assert not is_vecenv_wrapped(venv, VecMonitor)
    assert mon.env_is_wrapped(VecMonitor) == [False] * B and mon.get_attr("render_mode", [0, 3]) == [None, None]
    assert mon.env_method("get_images", indices=1) == [[]]
    assert _wrap_env(mon) is mon
    obs = mon.reset()
    assert venv._seeds == [None] * B and venv._options == [{}] * B and venv.reset_infos == [{}] * B
    assert obs.shape == (B, nx) and obs.dtype == np.float32

    singles = []
    for b in range(B):
        e = TransportPDE1D(**bk(), **params(lambda nx, b=b: ics[b]))
        o, _ = e.reset()
        np.testing.assert_array_equal(o, obs[b])
        singles.append(e)

    # 3. collect_rollouts-shaped loop ----------------------------------------------------------------------------------------
    n_steps = 10                                          # episodes of 4 env-steps: two ends per instance inside the rollout
    rng = np.random.default_rng(0)
    w = rng.normal(0, 0.05, (nx, 1)).astype(np.float32)
    buf_obs = np.zeros((n_steps, B, nx), np.float32)
    buf_rew = np.zeros((n_steps, B), np.float32)
    buf_start = np.zeros((n_steps, B), bool)
    last_obs, last_starts = obs, np.ones(B, bool)
    ep_info_buffer, bootstraps, ends = [], 0, 0
    for t in range(n_steps):
        actions = np.tanh(last_obs @ w) + rng.normal(0, 0.3, (B, 1)).astype(np.float32)
        clipped = np.clip(actions, venv.action_space.low, venv.action_space.high)
        new_obs, rewards, dones, infos = mon.step(clipped)
        assert new_obs.shape == (B, nx) and new_obs.dtype == np.float32
        assert rewards.shape == (B,) and rewards.dtype == np.float32
        assert dones.shape == (B,) and dones.dtype == np.bool_ and isinstance(infos, (list, tuple)) and len(infos) == B
        for i, e in enumerate(singles):
            o1, r1, te, tr, _ = e.step(clipped[i])
            assert bool(dones[i]) == bool(te or tr)
            np.testing.assert_allclose(rewards[i], r1, rtol=1e-6)
            if dones[i]:
                np.testing.assert_array_equal(infos[i]["terminal_observation"], o1)
                assert infos[i]["TimeLimit.truncated"] == bool(tr and not te)
                o1, _ = e.reset()
            np.testing.assert_array_equal(new_obs[i], o1)
        for idx, info in enumerate(infos):                # _update_info_buffer
            if info.get("episode") is not None:
                ep_info_buffer.append(info["episode"])
            assert info.get("is_success") is None
        for idx, done in enumerate(dones):                # time-limit bootstrap of OnPolicyAlgorithm.collect_rollouts
            if done:
                ends += 1
                assert infos[idx].get("terminal_observation") is not None
                if infos[idx].get("TimeLimit.truncated", False):
                    bootstraps += 1
        buf_obs[t], buf_rew[t], buf_start[t] = last_obs, rewards, last_starts      # RolloutBuffer.add copies
        last_obs, last_starts = new_obs, dones
    assert ends == 2 * B and len(ep_info_buffer) == 2 * B and all(e["l"] == 4 for e in ep_info_buffer) and bootstraps == 0
    for t in range(1, n_steps):                           # results handed out earlier were not overwritten by later steps
        assert not np.array_equal(buf_obs[t], buf_obs[t - 1])
    keep = [mon.step(np.zeros((B, 1), np.float32))[0] for _ in range(5)]
    assert all(not np.shares_memory(keep[0], k) for k in keep[1:]) and not np.array_equal(keep[0], keep[4])
    mon.close()

    # 4. gymnasium.vector face ------------------------------------------------------------------------------------------------
    cyc2 = itertools.cycle(ics)
    # ... built by gymnasium's own constructor through the registered vector entry point
    g = gymnasium.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **bk(), **params(lambda nx: next(cyc2)))
    assert type(g) is pde_control_gym.GymnasiumVectorAdapter and g.num_envs == B
    assert isinstance(g, gymnasium.vector.VectorEnv) and g.unwrapped is g and not g.closed
    assert g.single_observation_space.shape == (nx,) and g.observation_space.shape == (B, nx)
    assert g.single_action_space.shape == (1,) and g.action_space.shape == (B, 1)
    assert g.metadata["autoreset_mode"] == gymnasium.vector.AutoresetMode.SAME_STEP
    o, info = g.reset(seed=3, options=None)
    assert o.shape == (B, nx) and info == {}
    for t in range(4):
        o, r, term, trunc, info = g.step(g.action_space.sample())
    assert term.all() and not trunc.any() and info["_final_obs"].all() and info["final_obs"][2].shape == (nx,)
    assert info["_final_info"].all() and info["final_info"][0] == {}
    np.testing.assert_array_equal(o[1], ics[1])           # same-step auto-reset: already the next episode's first observation
    g.close()
    g.close()
    assert g.closed
    print("CONTRACT-OK", kind)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "double")
