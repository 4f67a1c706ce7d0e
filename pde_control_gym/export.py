"""Trajectory export in the on-disk layouts the reference's examples read and write.

* ``target.npz`` -- keys ``u``, ``v``: float64 ``[nt, nx, ny]`` velocity histories (read by
  examples/NavierStokes/NS2Doptimization.py:33-35, NS2Dppo.py and NS2Dsac.py to build ``U_ref``);
* ``result/NS_optmization.npz`` -- keys ``U, V, desired_U, desired_V, actions`` (written at NS2Doptimization.py:118);
* 1D trajectories -- the examples keep ``env.u`` / lists of observations in memory for plotting
  (transport1Dbackstepping.py:96-106); ``save_trajectory_1d`` stores the same arrays.

Writers only: all numbers come from the environments (``env.U`` / ``env.u`` histories or a ``DeviceRollout``'s device
buffers); nothing is recomputed here.
"""
from __future__ import annotations

import numpy as np


def _ns_history(env_or_U):
    U = getattr(env_or_U, "U", env_or_U)
    U = np.asarray(U)
    if U.ndim != 4 or U.shape[-1] != 2:
        raise ValueError(f"expected a [nt, nx, ny, 2] velocity history, got shape {U.shape}")
    return U


def save_ns_target(path, env_or_U):
    """Write a ``target.npz``-compatible file (keys ``u``, ``v``) from ``env.U`` or an ``[nt, nx, ny, 2]`` array."""
    U = _ns_history(env_or_U)
    np.savez(path, u=U[:, :, :, 0], v=U[:, :, :, 1])


def load_ns_target(path):
    """``U_ref`` as the examples build it: ``np.stack([u, v], axis=-1)`` of a ``target.npz``-style file."""
    z = np.load(path)
    return np.stack([z["u"], z["v"]], axis=-1)


def save_ns_optimization(path, env_or_U, desired_U, desired_V, actions):
    """Write the result file of the adjoint-optimisation example (keys ``U, V, desired_U, desired_V, actions``)."""
    U = _ns_history(env_or_U)
    np.savez(path, U=U[:, :, :, 0], V=U[:, :, :, 1], desired_U=np.array(desired_U), desired_V=np.array(desired_V),
             actions=np.asarray(actions))


def save_trajectory_1d(path, env, actions=None, rewards=None):
    """Write ``u`` = the environment's trajectory ``[nt, n]`` (needs ``record_history=True`` for the GPU-resident 1D
    environments; BrainTumor1D and TrafficPDE1D keep theirs on the host) plus optional per-step arrays."""
    u = np.asarray(env.u)
    if u.ndim != 2 or u.shape[0] != env.nt:
        raise ValueError("the environment holds no full trajectory (construct it with record_history=True)")
    extra = {}
    if actions is not None:
        extra["actions"] = np.asarray(actions)
    if rewards is not None:
        extra["rewards"] = np.asarray(rewards)
    np.savez(path, u=u, time_index=np.int64(env.time_index), dt=np.float64(env.dt), dx=np.float64(env.dx), **extra)


def save_rollout(path, rollout):
    """Write a ``DeviceRollout``'s buffers (``obs [T+1,B,D]``, ``actions``, ``rewards``, ``terminated``, ``truncated``)
    after one device-to-host copy each."""
    np.savez(path, **{k: getattr(rollout, k).cpu().numpy() for k in ("obs", "actions", "rewards", "terminated", "truncated")})
