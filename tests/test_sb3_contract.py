"""The batched faces against the Stable-Baselines3 / gymnasium.vector type contracts.

Neither package is installed here or on the GPU boxes, so the check runs in a child process whose sys.path starts with
tests/stubs -- stand-ins that restate the packages' public contract (tests/stubs/README.md).  In THIS process the stand-ins are
never imported: the rest of the suite keeps exercising the no-gymnasium fallback of pde_control_gym/_compat.py.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(kind):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sb3_contract_child.py"), kind], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and f"CONTRACT-OK {kind}" in out, out[-4000:]


def test_batched_faces_pass_the_sb3_and_gymnasium_type_gates_cpu_double():
    _run("double")


@pytest.mark.gpu
def test_batched_faces_pass_the_sb3_and_gymnasium_type_gates_hip():
    _run("hip")


def test_fallback_bases_have_the_same_contract_without_the_packages():
    """Without SB3 / gymnasium the stand-in bases of _compat.py give the batched faces the same constructor state and helpers."""
    import numpy as np
    import pde_control_gym
    from pde_control_gym import _compat
    from tests.fake_backend import FakeBackend
    from tests.test_host_api import _transport_params
    assert "stable_baselines3" not in sys.modules or not _compat.HAVE_SB3
    venv = pde_control_gym.make_vec("PDEControlGym-TransportPDE1D", num_envs=3, device="cpu", backend=FakeBackend(),
                                    **_transport_params())
    assert isinstance(venv, _compat.VecEnv) and venv.reset_infos == [{}] * 3 and venv.render_mode is None
    assert venv.seed(5) == [5, 6, 7]
    venv.set_options([{"a": 1}, {}, {}])
    venv.reset()
    assert venv._seeds == [None] * 3 and venv._options == [{}] * 3
    assert venv.get_attr("num_envs", 1) == [3] and venv.env_is_wrapped(object) == [False] * 3
    g = pde_control_gym.GymnasiumVectorAdapter(venv)
    assert isinstance(g, _compat.VectorEnv) and g.observation_space.shape == (3, 100) and g.single_observation_space.shape == (100,)
    o, r, te, tr, info = g.step(np.zeros((3, 1), np.float32))
    assert o.shape == (3, 100) and not te.any()
