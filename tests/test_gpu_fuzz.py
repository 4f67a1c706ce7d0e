"""Seeded differential fuzz of the 1D step kernels against the oracle (tests/fuzz_1d.py holds the generator; the long run
is `python tests/fuzz_1d.py 600`).  Bit patterns of rows and observations (so -0.0 != +0.0), flags, time indices, rewards."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_1d_kernels_against_oracle(seed):
    import fuzz_1d
    rng = np.random.default_rng(seed)
    done = 0
    for k in range(400):
        d = fuzz_1d.one_case(rng, k)
        done += d is not None and not d.endswith(")")
    assert done >= 100


def test_negative_zero_boundary_command_is_kept():
    """A commanded boundary value of exactly -0.0 survives all sub-steps (the fast loop's frozen slot would turn it into +0.0)."""
    import torch
    from pdecontrolgym_amd.batch1d import PDEBatch1D
    for kind, n in (("transport", 64), ("parabolic", 65)):
        dx = 1.0 / 64
        dt = 0.25 * dx * dx if kind == "parabolic" else 0.5 * dx
        env = PDEBatch1D(kind, 200 * dt, dt, 1, dx, 20 * dt, num_envs=3, device="cuda")
        env.reset(torch.ones(3, n), torch.zeros(3, n))
        env.step(torch.tensor([-0.0, 0.0, 0.5]))
        last = env.u[:, -1].cpu().numpy().view(np.uint32)
        assert last[0] == 0x80000000 and last[1] == 0 and env.u[2, -1].item() == 0.5


@pytest.mark.parametrize("which,count", [("ns", 60), ("ns256", 14), ("traffic", 60), ("tumor", 40)])
def test_fuzz_other_kernels_against_oracle(which, count):
    """tests/fuzz_more.py: NS2D float64 bit-exact for random grids / BC combinations / sweep counts (and float32 tiled ==
    generic), traffic ARZ and brain-tumour (daily steps and the in-kernel growth run) for random parameter sets."""
    import fuzz_more
    fn = {"ns": fuzz_more.ns_case, "ns256": fuzz_more.ns256_case, "traffic": fuzz_more.traffic_case, "tumor": fuzz_more.tumor_case}[which]
    rng = np.random.default_rng(17)
    done = 0
    for k in range(count):
        d = fn(rng, k)
        done += d is not None and not d.endswith(")")
    assert done >= count // 2


@pytest.mark.parametrize("seed", [5, 6])
def test_fuzz_rollout_kernels_against_step_calls(seed):
    """tests/fuzz_rollout.py: pdegym_*_rollout / pdegym_traffic_rollout == T step calls, bit for bit, for random shapes, reward
    kinds, auto-reset pools and rollout lengths (the long run is `python tests/fuzz_rollout.py 600`)."""
    import fuzz_rollout
    rng = np.random.default_rng(seed)
    n = {"1d": 0, "traffic": 0}
    for _ in range(300):
        which = "traffic" if rng.random() < 0.3 else "1d"
        with np.errstate(all="ignore"):
            d = fuzz_rollout.case_traffic(rng) if which == "traffic" else fuzz_rollout.case_1d(rng)
        n[which] += d is not None
    assert n["1d"] >= 150 and n["traffic"] >= 50


def test_fuzz_policy_inside_rollout_against_two_launch_path():
    """tests/fuzz_policy_rollout.py: random networks (1 .. 4 layers, up to 256 units) inside the 1D rollout kernels against
    pdegym_mlp_forward + step calls; layers of more than 64 units (cooperative MFMA evaluation) bit for bit."""
    import fuzz_policy_rollout as fp
    rng = np.random.default_rng(4)
    wide = done = 0
    for k in range(250):
        d = fp.traffic_case(rng, k) if k % 5 == 4 else fp.one_case(rng, k)
        wide += d.endswith("[wide]")
        done += not d.endswith(")")
    assert done >= 200 and wide >= 60

