"""The reference's REAL callers against this package ("missing" 1 / "next" 4).

Neither gymnasium nor stable-baselines3 ships in the MI355X image, so every test here SKIPS today (``pytest.importorskip``); the
contract is otherwise tested against stand-ins in a child process (tests/test_sb3_contract.py, tests/stubs/README.md).  This file
is what runs the first time a box HAS the packages: the calls are the ones the reference's examples make --

  * ``gym.make(id, **parameters)`` + ``PPO("MlpPolicy", env).learn()``     examples/transportPDE/transport1Dppo.py:77-90
  * ``check_env`` (gymnasium's and SB3's)                                   the Env contract both trainers assume
  * ``gym.make_vec(id, num_envs=N, **parameters)``                          the registered vector entry point (gymnasium >= 1.0)
  * ``PPO("MlpPolicy", pde_control_gym.make_vec(...), n_steps=8).learn(64)``   the batched face handed to SB3 AS IT IS
  * the NavierStokes2D loop of examples/NavierStokes/NS2Dppo.py:29-66 through gym.make

-- on the CPU double (tests/fake_backend.py: the NumPy oracle behind the backend interface; host logic only) and, ``-m gpu``, on HIP.
The stand-ins of tests/stubs are never on sys.path in this process; a stub that leaked in would make the tests skip, not pass.
"""
import itertools

import numpy as np
import pytest

gymnasium = pytest.importorskip("gymnasium", reason="gymnasium is not installed in this image (tests/test_sb3_contract.py covers the stand-ins)")
if str(getattr(gymnasium, "__version__", "")).endswith("contract-stub"):      # tests/stubs leaked onto sys.path: not the real thing
    pytest.skip("the gymnasium on sys.path is the contract stand-in of tests/stubs", allow_module_level=True)

import pde_control_gym  # noqa: E402

from tests.five_ids import BETA, ICS, NX, _backend, _five_ids, _ns_params, _transport_params  # noqa: E402,F401


KINDS = ["double", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", KINDS)
def test_gym_make_builds_a_real_gymnasium_env_that_passes_check_env(kind):
    from gymnasium.utils.env_checker import check_env
    env = gymnasium.make("PDEControlGym-TransportPDE1D", **_backend(kind), **_transport_params(lambda nx: ICS[0]))
    assert isinstance(env.unwrapped, gymnasium.Env) and isinstance(env.unwrapped, pde_control_gym.src.TransportPDE1D)
    assert isinstance(env.observation_space, gymnasium.spaces.Box) and env.observation_space.shape == (NX,)
    assert isinstance(env.action_space, gymnasium.spaces.Box) and env.action_space.shape == (1,)
    check_env(env.unwrapped, skip_render_check=True)
    obs, info = env.reset(seed=0)
    assert obs.shape == (NX,) and obs.dtype == np.float32 and isinstance(info, dict)
    done, n = False, 0
    while not done:
        obs, rew, term, trunc, info = env.step(env.action_space.sample())
        assert obs.dtype == np.float32 and np.isscalar(float(rew)) and isinstance(info, dict)
        done, n = bool(term or trunc), n + 1
    assert n == 4
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_gym_make_vec_builds_the_batched_environment_through_the_vector_entry_point(kind):
    if not hasattr(gymnasium, "make_vec"):
        pytest.skip("this gymnasium has no make_vec (< 0.29)")
    B = 5
    cyc = itertools.cycle(ICS[:B])
    g = gymnasium.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_backend(kind), **_transport_params(lambda nx: next(cyc)))
    assert type(g) is pde_control_gym.GymnasiumVectorAdapter and isinstance(g, gymnasium.vector.VectorEnv) and g.num_envs == B
    assert g.single_observation_space.shape == (NX,) and g.observation_space.shape == (B, NX) and g.action_space.shape == (B, 1)
    obs, info = g.reset(seed=3)
    assert obs.shape == (B, NX) and obs.dtype == np.float32
    for _ in range(4):
        obs, rew, term, trunc, info = g.step(g.action_space.sample())
    assert rew.shape == (B,) and term.shape == (B,) and trunc.shape == (B,) and (term | trunc).all()
    np.testing.assert_array_equal(obs[1], ICS[1])          # same-step auto-reset (AutoresetMode.SAME_STEP): the next episode's first row
    g.close()


@pytest.mark.parametrize("kind", KINDS)
def test_ppo_learns_on_the_batched_face_as_it_is(kind):
    sb3 = pytest.importorskip("stable_baselines3", reason="stable-baselines3 is not installed in this image")
    from stable_baselines3.common.vec_env import VecEnv, VecMonitor
    B = 4
    cyc = itertools.cycle(ICS[:B])
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=B, **_backend(kind), **_transport_params(lambda nx: next(cyc)))
    assert isinstance(venv, VecEnv)                       # BaseAlgorithm._wrap_env takes it without a DummyVecEnv around it
    model = sb3.PPO("MlpPolicy", VecMonitor(venv), n_steps=8, batch_size=16, n_epochs=1, device="cpu", seed=0)
    assert model.get_env().num_envs == B and model.get_env().unwrapped is venv
    model.learn(total_timesteps=64)
    assert model.num_timesteps >= 64
    assert len(model.ep_info_buffer) > 0 and all(int(e["l"]) == 4 for e in model.ep_info_buffer)     # Monitor statistics of 4-step episodes
    obs = model.get_env().reset()
    act, _ = model.predict(obs, deterministic=True)
    assert act.shape == (B, 1)
    venv.close()


@pytest.mark.parametrize("kind", KINDS)
def test_ppo_learns_on_the_single_environment_like_the_reference_example(kind):
    """transport1Dppo.py:77-90 verbatim in shape: ``env = gym.make(id, **hyperbolicParameters); PPO("MlpPolicy", env, ...).learn()``."""
    sb3 = pytest.importorskip("stable_baselines3", reason="stable-baselines3 is not installed in this image")
    from stable_baselines3.common.env_checker import check_env as sb3_check_env
    env = gymnasium.make("PDEControlGym-TransportPDE1D", **_backend(kind), **_transport_params(lambda nx: ICS[2]))
    sb3_check_env(env.unwrapped, warn=True)
    model = sb3.PPO("MlpPolicy", env, n_steps=8, batch_size=8, n_epochs=1, device="cpu", seed=0)
    model.learn(total_timesteps=32)
    assert model.num_timesteps >= 32
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_navier_stokes_through_gym_make_runs_the_reference_loop(kind):
    """NS2Dppo.py:29-66: gym.make, reset, step until terminated, rewards collected."""
    env = gymnasium.make("PDEControlGym-NavierStokes2D", **_backend(kind), **_ns_params())
    obs, info = env.reset(seed=0)
    assert obs.shape == (16, 16, 2)
    total, done, n = 0.0, False, 0
    while not done and n < 50:
        obs, rew, term, trunc, info = env.step(np.array([3.0]))
        total, done, n = total + float(rew), bool(term or trunc), n + 1
    assert done and n == 5 and np.isfinite(total)          # T / dt = 6: nt = 7 frames, terminated when time_index reaches nt - 2
    env.close()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("which", range(5), ids=["transport", "parabolic", "navier_stokes", "traffic", "brain_tumor"])
def test_check_env_passes_on_every_registered_id(kind, which):
    """gymnasium's (and, when installed, SB3's) ``check_env`` on ALL five ids the reference registers, each built by
    ``gym.make`` from its own parameter dictionary."""
    from gymnasium.utils.env_checker import check_env
    env_id, params = _five_ids()[which]
    env = gymnasium.make(env_id, **_backend(kind), **params)
    assert isinstance(env.unwrapped, gymnasium.Env)
    check_env(env.unwrapped, skip_render_check=True)
    try:
        from stable_baselines3.common.env_checker import check_env as sb3_check_env
    except ImportError:
        sb3_check_env = None
    if sb3_check_env is not None:
        sb3_check_env(gymnasium.make(env_id, **_backend(kind), **params).unwrapped, warn=True)
    env.close()
