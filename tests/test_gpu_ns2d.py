"""GPU parity tests for the NS2D HIP kernels, through the C ABI.

Stated tolerances:
  * float64 build vs reference golden vectors / oracle: fields BIT-EXACT (same operation order, no FMA
    contraction, IEEE division); rewards rtol 1e-12 (summation order of the Frobenius norm);
  * float32 build vs the float64 oracle from identical state, ONE step: rtol 1e-5, atol 2e-6*max|field|
    (velocity), pressure atol 5e-5*max|p| (K Jacobi sweeps accumulate rounding); reward rtol 1e-4.
"""
import numpy as np
import pytest

from tests.cases import NS_BC, ns_bc_from_array

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dbg(key, value):
    """Test-only kernel dispatch override (pdegym_debug_set, include/pdegym.h); takes ints or the "0"/"1" strings looped over."""
    from pdecontrolgym_amd import _native as N
    N.load().pdegym_debug_set(getattr(N, key), int(value))


def _mk(kw, B, dtype):
    from pdecontrolgym_amd.batch2d import NSBatch2D
    return NSBatch2D(num_envs=B, device="cuda", dtype=dtype, **kw)


def test_ns_f64_reproduces_target_npz_frames(golden_ns):
    """examples/NavierStokes/target.npz (21x21, K=2000, 199 steps): bit-exact, every instance."""
    g = golden_ns["N1"]
    B = 3
    kw = dict(T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05, boundary_condition=NS_BC, U_ref=np.zeros((200, 21, 21, 2)),
              action_ref=2.0 * np.ones(1000), gamma=0.1)
    env = _mk(kw, B, torch.float64)
    env.reset(g.u0, g.v0, np.zeros((21, 21)))
    keep = set(int(k) for k in g.keep)
    for t in range(1, 200):
        obs, r, te = env.step(np.full(B, g.actions[t - 1]))
        if t in keep:
            o = obs.cpu().numpy()
            for b in (0, B - 1):
                np.testing.assert_array_equal(o[b, :, :, 0], g[f"u{t}"], err_msg=f"u frame {t}")
                np.testing.assert_array_equal(o[b, :, :, 1], g[f"v{t}"], err_msg=f"v frame {t}")
    assert te.cpu().numpy().all()
    np.testing.assert_array_equal(env.p.cpu().numpy()[1], g.p_final)


@pytest.mark.parametrize("case", ["N2_32", "N2_64", "N2_48"])
def test_ns_f64_mixed_bc_golden(golden_ns, case):
    g = golden_ns[case]
    kw = dict(T=int(g.nt) * float(g.dt), dt=float(g.dt), X=1, dx=float(g.dx), Y=1, dy=float(g.dx),
              boundary_condition=ns_bc_from_array(g.bc), U_ref=g.U_ref, action_ref=g.action_ref, gamma=0.1,
              maximum_pressure_iteration=50)
    env = _mk(kw, 2, torch.float64)
    env.reset(g.u0, g.v0, g.p0)
    for i, a in enumerate(g.actions):
        obs, r, te = env.step(np.full(2, a))
        np.testing.assert_array_equal(obs.cpu().numpy()[1], g.obs[i])
        np.testing.assert_array_equal(env.p.cpu().numpy()[1], g.p[i])
        np.testing.assert_allclose(r.cpu().numpy()[1], g.rewards[i], rtol=1e-12)


def test_ns_f64_reward_with_reference_trajectory(golden_ns):
    g, gb = golden_ns["N1"], golden_ns["N1b"]
    Uref = np.zeros((200, 21, 21, 2))
    for k in g.keep:
        Uref[int(k)] = 0.5 * np.stack([g[f"u{int(k)}"], g[f"v{int(k)}"]], -1)
    kw = dict(T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05, boundary_condition=NS_BC, U_ref=Uref,
              action_ref=2.0 * np.ones(1000), gamma=0.1)
    env = _mk(kw, 1, torch.float64)
    env.reset(g.u0, g.v0, np.zeros((21, 21)))
    for t in (1, 2):
        obs, r, te = env.step(np.array([g.actions[t - 1]]))
        np.testing.assert_allclose(r.cpu().numpy()[0], gb.rewards[t - 1], rtol=1e-12)


def test_ns_f64_c4_checksums(golden_ns):
    """BASELINE config 4 shape (128x128, K=50) against the reference's checksums and sampled points."""
    g = golden_ns["N3"]
    n, nt = 128, int(g.nt)
    kw = dict(T=nt * float(g.dt), dt=float(g.dt), X=1, dx=float(g.dx), Y=1, dy=float(g.dx), boundary_condition=NS_BC,
              U_ref=np.zeros((nt, n, n, 2)), action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=50)
    env = _mk(kw, 2, torch.float64)
    one = np.ones((n, n))
    env.reset(g.ic[0] * one, g.ic[1] * one, g.ic[2] * one)
    for i, a in enumerate(g.actions):
        obs, r, te = env.step(np.full(2, a))
        o = obs.cpu().numpy()[1]
        p = env.p.cpu().numpy()[1]
        sums = [np.linalg.norm(o[..., 0]), np.linalg.norm(o[..., 1]), np.linalg.norm(p), o.min(), o.max(), p.min(), p.max()]
        np.testing.assert_allclose(sums, g.sums[i], rtol=1e-13)
        pts = g.pts
        smp = np.stack([o[pts[:, 0], pts[:, 1], 0], o[pts[:, 0], pts[:, 1], 1], p[pts[:, 0], pts[:, 1]]], -1)
        np.testing.assert_array_equal(smp, g.samples[i])
        np.testing.assert_allclose(r.cpu().numpy()[1], g.rewards[i], rtol=1e-12)


def _random_case(n, B, K, seed, bc, action_dim=1):
    rng = np.random.default_rng(seed)
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = 8
    xs = np.linspace(0, 1, n)
    Xg, Yg = np.meshgrid(xs, xs)
    u0 = np.stack([np.sin(2 * np.pi * Xg * rng.uniform(0.5, 2)) * np.cos(np.pi * Yg) * rng.uniform(0.5, 2) + rng.uniform(-1, 1) for _ in range(B)])
    v0 = np.stack([np.cos(np.pi * Xg) * np.sin(2 * np.pi * Yg * rng.uniform(0.5, 2)) * rng.uniform(0.5, 2) + rng.uniform(-1, 1) for _ in range(B)])
    p0 = rng.uniform(-1, 1, (B, n, n))
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=bc, U_ref=rng.uniform(-1, 1, (nt, n, n, 2)),
              action_ref=rng.uniform(1, 3, nt), gamma=0.1, maximum_pressure_iteration=K)
    acts = rng.uniform(2, 4, (3, B, action_dim))
    return kw, u0, v0, p0, acts


BC_MIX = {"upper": ["Controllable", "Neumann"], "lower": ["Neumann", "Controllable"],
          "left": ["Neumann", "Dirchilet"], "right": ["Controllable", "Neumann"]}


@pytest.mark.parametrize("n,B,K,bc,adim", [(21, 5, 7, BC_MIX, 1), (40, 3, 50, NS_BC, 1), (128, 2, 50, BC_MIX, 1),
                                           (33, 2, 11, BC_MIX, 33), (8, 4, 3, BC_MIX, 1)])
def test_ns_f64_matches_oracle_random(n, B, K, bc, adim):
    from oracle import pde_oracle as po
    kw, u0, v0, p0, acts = _random_case(n, B, K, 100 + n, bc, adim)
    orc = po.NavierStokesOracle(**kw)
    env = _mk(dict(kw, action_dim=adim), B, torch.float64)
    orc.reset(u0, v0, p0)
    env.reset(u0, v0, p0)
    for a in acts:
        o_ref, r_ref, te_ref, _ = orc.step(a)
        obs, r, te = env.step(a)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
        np.testing.assert_array_equal(env.p.cpu().numpy(), orc.p)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-12)
        np.testing.assert_array_equal(te.cpu().numpy().astype(bool), te_ref)


@pytest.mark.parametrize("n,B,K,bc", [(21, 4, 50, NS_BC), (64, 3, 50, BC_MIX), (128, 4, 50, NS_BC), (128, 2, 50, BC_MIX),
                                      (100, 2, 20, BC_MIX), (48, 2, 200, NS_BC)])
def test_ns_f32_single_step_vs_f64_oracle(n, B, K, bc):
    """float32 throughput build: one step from identical state vs the float64 oracle (tolerances in the header)."""
    from oracle import pde_oracle as po
    kw, u0, v0, p0, acts = _random_case(n, B, K, 7 + n, bc)
    orc = po.NavierStokesOracle(**kw)
    env = _mk(kw, B, torch.float32)
    for a in acts:
        # restart both from the SAME float32-representable state each step (single-step comparison)
        u32, v32, p32 = (x.astype(np.float32) for x in (u0, v0, p0))
        orc.reset(u32.astype(np.float64), v32.astype(np.float64), p32.astype(np.float64))
        env.reset(u32, v32, p32)
        a32 = a.astype(np.float32)
        o_ref, r_ref, _, _ = orc.step(a32.astype(np.float64))
        obs, r, te = env.step(a32)
        o = obs.cpu().numpy().astype(np.float64)
        scale = np.abs(o_ref).max()
        np.testing.assert_allclose(o, o_ref, rtol=1e-5, atol=2e-6 * scale)
        pscale = np.abs(orc.p).max()
        np.testing.assert_allclose(env.p.cpu().numpy().astype(np.float64), orc.p, rtol=1e-4, atol=5e-5 * pscale)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-4)
        u0, v0, p0 = o_ref[..., 0], o_ref[..., 1], orc.p


def test_ns_f32_horizon_vs_f64_oracle_c4():
    """STATED float32 HORIZON (docs/HISTORY.md section 4): BASELINE config 4 (128x128, K=50, lid-driven boundary set of NS2Dppo.py:21-26),
    smooth and constant initial fields, random lid actions U(2,4): over a 100-step episode the float32 throughput kernels stay
    within 1e-5 * max|U| of the float64 oracle in the velocity field (measured 2e-6, tools/ns_f32_horizon.py) and within
    5e-5 * max|p| in the pressure (measured 1e-6 .. 7e-6); rewards rtol 1e-5.  The flow is dissipative (nu = 0.1), the Jacobi
    solve is a contraction: rounding errors do not accumulate, the bound is flat in the number of steps."""
    from oracle import pde_oracle as po
    n, B, K, steps = 128, 2, 50, 100
    rng = np.random.default_rng(5)
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = steps + 2
    xs = np.linspace(0, 1, n)
    Xg, Yg = np.meshgrid(xs, xs)
    u0 = np.stack([np.sin(2 * np.pi * Xg) * np.cos(np.pi * Yg) * 1.5 + 0.3, -3.7 * np.ones((n, n))]).astype(np.float32)
    v0 = np.stack([np.cos(np.pi * Xg) * np.sin(2 * np.pi * Yg) * 0.8 - 0.5, 2.2 * np.ones((n, n))]).astype(np.float32)
    p0 = np.zeros((B, n, n), dtype=np.float32)
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=NS_BC, U_ref=np.zeros((nt, n, n, 2)),
              action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K)
    orc = po.NavierStokesOracle(**kw)
    env = _mk(kw, B, torch.float32)
    orc.reset(u0.astype(np.float64), v0.astype(np.float64), p0.astype(np.float64))
    env.reset(u0, v0, p0)
    worst_u = worst_p = 0.0
    for t in range(steps):
        a = rng.uniform(2, 4, B).astype(np.float32)
        o_ref, r_ref, _, _ = orc.step(a.astype(np.float64))
        obs, r, te = env.step(a)
        if t % 10 == 9 or t < 3:
            o = obs.cpu().numpy().astype(np.float64)
            worst_u = max(worst_u, float(np.abs(o - o_ref).max() / np.abs(o_ref).max()))
            worst_p = max(worst_p, float(np.abs(env.p.cpu().numpy() - orc.p).max() / np.abs(orc.p).max()))
            np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-5)
    assert worst_u < 1e-5 and worst_p < 5e-5, (worst_u, worst_p)


@pytest.mark.parametrize("n,steps", [(64, 300), (128, 400), (256, 60)])
def test_ns_f32_long_horizon_vs_f64_kernels(n, steps):
    """Same bound over longer episodes and the other register-tiled grid sizes, against the float64 HIP kernels (which the
    tests above pin bit-exactly to the oracle): max|U32 - U64| <= 1e-5 * max|U64| at every checkpoint."""
    B, K = 4, 50
    rng = np.random.default_rng(n)
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = steps + 2
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=NS_BC, U_ref=np.zeros((nt, n, n, 2)),
              action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=K)
    e64, e32 = _mk(kw, B, torch.float64), _mk(kw, B, torch.float32)
    c = rng.uniform(-5, 5, (3, B, 1, 1)).astype(np.float32)
    ic = [np.broadcast_to(c[k], (B, n, n)).copy() for k in range(3)]
    e64.reset(*[x.astype(np.float64) for x in ic])
    e32.reset(*ic)
    for t in range(steps):
        a = rng.uniform(2, 4, B).astype(np.float32)
        o64, _, _ = e64.step(a.astype(np.float64))
        o32, _, _ = e32.step(a)
        if t % 20 == 19:
            err = float((o32.double() - o64).abs().max() / o64.abs().max())
            assert err < 1e-5, (t, err)


def test_solve_pressure_public_api_f64():
    """env.solve_pressure on arbitrary fields (examples/NavierStokes/NS2Doptimization.py:97)."""
    from oracle import pde_oracle as po
    kw, u0, v0, p0, _ = _random_case(21, 3, 60, 5, NS_BC)
    orc = po.NavierStokesOracle(**kw)
    env = _mk(kw, 3, torch.float64)
    out = env.solve_pressure(u0, v0, p0)
    np.testing.assert_array_equal(out.cpu().numpy(), orc.solve_pressure(u0, v0, p0))


def test_ns_masked_reset_and_properties_c4_size():
    """BASELINE config 4 size (128x128, K=50, B=512, float32): batch invariance + masked reset + finiteness."""
    B, n = 512, 128
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = 20
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=NS_BC, U_ref=np.zeros((nt, n, n, 2), dtype=np.float32),
              action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=50)
    env = _mk(kw, B, torch.float32)
    g = torch.Generator().manual_seed(3)
    c = (torch.rand(B // 2, 3, generator=g) * 10 - 5)
    c = torch.cat([c, c])
    one = torch.ones(1, n, n)
    u0, v0, p0 = (c[:, k].reshape(B, 1, 1) * one for k in range(3))
    env.reset(u0, v0, p0)
    for _ in range(3):
        a = torch.rand(B // 2, generator=g) * 2 + 2
        a = torch.cat([a, a])
        obs, r, te = env.step(a)
    o = obs.cpu()
    assert torch.isfinite(o).all()
    assert torch.equal(o[: B // 2], o[B // 2:])                       # duplicated instances agree bitwise
    assert torch.equal(o[:, -1, 1:-1, 0], a.reshape(B, 1).expand(B, n - 2).float())   # upper edge u = action
    assert (o[:, 0, :, :] == 0).all() and (o[:, :, 0, :] == 0).all()  # Dirichlet walls
    # reward == -0.5*||U||^2/n^2 - gamma/2*(a-2)^2 recomputed from obs
    rr = -0.5 * (o.double() ** 2).sum(dim=(1, 2, 3)) / n / n - 0.05 * (a.double() - 2) ** 2
    torch.testing.assert_close(r.cpu().double(), rr, rtol=1e-5, atol=1e-6)
    mask = torch.zeros(B, dtype=torch.uint8)
    mask[::3] = 1
    env.reset(u0, v0, p0, mask=mask)
    ti = env.time_index.cpu()
    assert torch.equal(ti, torch.where(mask.bool(), 0, 3).int())
    assert torch.equal(env.u.cpu()[::3], u0[::3].float())


@pytest.mark.parametrize("n", [128, 64, 256])
def test_ns_f32_tiled_kernel_equals_generic_kernel_bitwise(n):
    """The register-tiled float32 kernel and the generic float32 kernel evaluate the same expression tree:
    fields, pressure and observations must agree bit for bit over several steps (reward: summation order)."""
    import os
    kw, u0, v0, p0, acts = _random_case(n, 5, 50 if n != 256 else 37, 900 + n, BC_MIX)    # 256: an odd sweep count (the row rotation ends off the identity map)
    outs = []
    for force in ("0", "1"):
        _dbg("DEBUG_NS_GENERIC", force)
        try:
            env = _mk(kw, 5, torch.float32)
            env.reset(u0, v0, p0)
            res = []
            for a in acts:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_GENERIC", "0")
    for (o1, p1, r1), (o2, p2, r2) in zip(*outs):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_allclose(r1, r2, rtol=1e-5)


@pytest.mark.parametrize("K,interleaved,adim", [(50, True, 1), (7, True, 128), (1, False, 1), (0, True, 1), (33, False, 128)])
def test_ns_f64_tiled_kernel_equals_generic_kernel_and_oracle_bitwise(K, interleaved, adim):
    """The float64 register-tiled 128x128 kernel (the reference's precision on the fast path): fields, pressure and
    observations bit-identical to the generic float64 kernel AND to the NumPy oracle; even / odd / zero sweep counts, the
    interleaved (observation = state) and the separate-u,v state layouts, scalar and per-node boundary actions."""
    import os
    from oracle import pde_oracle as po
    from pdecontrolgym_amd.batch2d import NSBatch2D
    B = 3
    kw, u0, v0, p0, acts = _random_case(128, B, K, 4000 + K, BC_MIX, adim)
    kw = dict(kw, action_dim=adim)
    outs = []
    for force in ("0", "1"):
        _dbg("DEBUG_NS_GENERIC", force)
        try:
            env = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float64, interleaved_state=interleaved, **kw)
            env.reset(u0, v0, p0)
            res = []
            for a in acts:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy(), env.u.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_GENERIC", "0")
    orc = po.NavierStokesOracle(**{k: v for k, v in kw.items() if k != "action_dim"})
    orc.reset(u0, v0, p0)
    for a, (o1, p1, r1, u1), (o2, p2, r2, u2) in zip(acts, *outs):
        o_ref, r_ref, _, _ = orc.step(a)
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_array_equal(u1, u2)
        np.testing.assert_array_equal(o1, o_ref)
        np.testing.assert_array_equal(p1, orc.p)
        np.testing.assert_allclose(r1, r_ref, rtol=1e-12)
        np.testing.assert_allclose(r1, r2, rtol=1e-12)


@pytest.mark.parametrize("K,interleaved,adim", [(50, True, 1), (25, True, 256), (26, False, 1), (1, True, 1), (0, False, 1), (35, True, 1),
                                                (51, True, 1), (2, True, 256), (24, False, 256), (76, True, 1), (100, False, 1), (0, True, 1)])
def test_ns_f64_256_slab_passes_equal_generic_kernel_bitwise(K, interleaved, adim):
    """256 x 256 float64 (BASELINE config 5 at the reference's precision): the env-step runs as passes of <= 25 sweeps over
    three overlapping slabs per instance, the first pass opening with the predictor / rhs phase and the last one closing with
    corrector / observation / reward (pdegym_ns256_f64.hip).  Fields, pressure and observations must equal the
    workgroup-per-instance kernel bit for bit: one launch (K <= 25: front + sweeps + back in the same kernel), exactly 25,
    25 + 1, two full passes (BASELINE), 50 + 1 (three passes), four passes, odd and even counts per pass, none; both state
    layouts (the interleaved one ping-pongs the pressure field, the separate one brings it home in an even number of passes),
    scalar and per-node boundary actions."""
    from pdecontrolgym_amd.batch2d import NSBatch2D
    B = 2
    kw, u0, v0, p0, acts = _random_case(256, B, K, 5000 + K, BC_MIX, adim)
    kw = dict(kw, action_dim=adim)
    outs = []
    for force in ("0", "1"):
        _dbg("DEBUG_NS_GENERIC", force)
        try:
            env = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float64, interleaved_state=interleaved, **kw)
            env.reset(u0, v0, p0)
            res = []
            for a in acts[:2]:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), env.u.cpu().numpy().copy(), r.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_GENERIC", "0")
    for (o1, p1, u1, r1), (o2, p2, u2, r2) in zip(*outs):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_array_equal(u1, u2)
        np.testing.assert_allclose(r1, r2, rtol=1e-12)


@pytest.mark.parametrize("n,dtype,K", [(21, "float64", 200), (21, "float64", 7), (33, "float32", 50), (64, "float64", 31),
                                       (64, "float32", 50), (5, "float64", 12),
                                       # one LDS copy (grids of 4097..16384 cells): float64 128x128 is 128 KB
                                       (128, "float64", 31), (100, "float64", 8), (100, "float32", 21), (128, "float32", 12),
                                       (65, "float64", 9)])
def test_ns_lds_resident_jacobi_equals_global_memory_jacobi_bitwise(n, dtype, K):
    """Grids of <= 4096 cells keep p in LDS during the sweeps; same arithmetic as the global-memory loop -> identical
    fields, pressure (incl. the public solve_pressure) and rewards.  Odd and even K, both dtypes."""
    import os
    td = getattr(torch, dtype)
    kw, u0, v0, p0, acts = _random_case(n, 3, K, 77 + n + K, BC_MIX)
    outs = []
    for no_lds in ("0", "1"):
        _dbg("DEBUG_NS_NO_LDS_JACOBI", no_lds)
        _dbg("DEBUG_NS_GENERIC", "1")
        try:
            env = _mk(kw, 3, td)
            env.reset(u0, v0, p0)
            res = []
            for a in acts:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy()))
            res.append((env.solve_pressure(u0, v0, p0).cpu().numpy().copy(),))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_NO_LDS_JACOBI", "0")
            _dbg("DEBUG_NS_GENERIC", "0")
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            if x.ndim == 1:      # rewards: the block reduction order follows the workgroup size, which differs between the paths
                np.testing.assert_allclose(x, y, rtol=1e-5 if dtype == "float32" else 1e-13)
            else:
                np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("n,dtype", [(128, "float32"), (40, "float32"), (21, "float64"), (256, "float32")])
def test_ns_interleaved_state_equals_separate_fields(n, dtype):
    """interleaved_state=True (state lives in the double-buffered obs tensors) == separate u, v fields, bitwise."""
    from pdecontrolgym_amd.batch2d import NSBatch2D
    td = getattr(torch, dtype)
    kw, u0, v0, p0, acts = _random_case(n, 3, 30, 31 + n, BC_MIX)
    outs = []
    for inter in (True, False):
        env = NSBatch2D(num_envs=3, device="cuda", dtype=td, interleaved_state=inter, **kw)
        assert (env.t["u"] is None) == inter
        env.reset(u0, v0, p0)
        res = []
        for a in acts:
            obs, r, te = env.step(a)
            res.append((obs.cpu().numpy().copy(), env.u.cpu().numpy().copy(), env.v.cpu().numpy().copy(),
                        env.p.cpu().numpy().copy(), r.cpu().numpy().copy()))
        # masked reset keeps the other instances' state
        m = torch.tensor([1, 0, 1], dtype=torch.uint8)
        env.reset(u0 * 0 + 1.5, v0 * 0 - 0.5, p0 * 0, mask=m)
        obs, r, te = env.step(acts[0])
        res.append((obs.cpu().numpy().copy(), env.u.cpu().numpy().copy(), env.v.cpu().numpy().copy(), env.p.cpu().numpy().copy(),
                    r.cpu().numpy().copy()))
        outs.append(res)
    for x, y in zip(*outs):
        for a, b in zip(x, y):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,B,dtype", [(256, 512, "float32"), (128, 4096, "float32"), (256, 512, "float64"), (128, 4096, "float64")])
def test_ns_full_size_shards_properties_and_sampled_oracle(n, B, dtype):
    """BASELINE config 5's per-GPU shard (256x256, 512 instances) and the metric string's NS2D 128x128 at batch 4096, K = 50, at
    FULL size, in float32 and at the reference's own precision: duplicated instances give identical fields (batch invariance),
    everything is finite, the lid row carries the action, and sampled instances agree with the float64 oracle -- bit for bit in
    float64, within the stated float32 tolerance otherwise."""
    from oracle import pde_oracle as po
    td = getattr(torch, dtype)
    rng = np.random.default_rng(n + B)
    dx = 1.0 / (n - 1)
    dt = 0.2 * 0.5 * dx * dx / 0.1
    nt = 6
    # smooth random fields (constants excite only the lid) and a non-zero reference trajectory for the reward
    g1 = torch.linspace(0, 1, n, dtype=torch.float64)
    Yg, Xg = torch.meshgrid(g1, g1, indexing="ij")
    Uref = np.stack([np.stack([np.sin(np.pi * (Xg.numpy() + 0.1 * t)) * np.cos(np.pi * Yg.numpy()),
                               0.5 * np.cos(2 * np.pi * Xg.numpy()) * np.sin(np.pi * (Yg.numpy() - 0.05 * t))], -1) for t in range(nt)])
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dx, boundary_condition=NS_BC, U_ref=Uref.astype(dtype),
              action_ref=2.0 * np.ones(nt), gamma=0.1, maximum_pressure_iteration=50)
    env = _mk(kw, B, td)
    half = B // 2
    c = torch.as_tensor(rng.uniform(-1, 1, (half, 8)), dtype=torch.float64).reshape(half, 8, 1, 1)
    fx, fy = 0.5 + 1.5 * c[:, 6].abs(), 0.5 + 1.5 * c[:, 7].abs()
    fields = (3 * c[:, 0] * torch.sin(2 * np.pi * fx * Xg) * torch.cos(np.pi * Yg) + 2 * c[:, 1],
              3 * c[:, 2] * torch.cos(np.pi * Xg) * torch.sin(2 * np.pi * fy * Yg) + 2 * c[:, 3],
              c[:, 4] * torch.cos(np.pi * Xg) * torch.cos(np.pi * fy * Yg) + c[:, 5])
    ic = [torch.cat([f, f]).to(td).contiguous() for f in fields]
    del fields
    env.reset(*ic)
    acts = []
    for _ in range(2):
        a = torch.as_tensor(rng.uniform(2, 4, half), dtype=td)
        a = torch.cat([a, a])
        acts.append(a)
        obs, r, te = env.step(a)
    assert torch.isfinite(obs).all() and torch.isfinite(env.p).all()
    assert torch.equal(obs[:half], obs[half:]) and torch.equal(env.p[:half], env.p[half:]) and torch.equal(r[:half], r[half:])
    assert torch.equal(obs[:, -1, 1:-1, 0].cpu(), acts[-1].reshape(B, 1).expand(B, n - 2))      # "upper" edge: u = action
    sel = [0, half // 3, half - 1]
    orc = po.NavierStokesOracle(**dict(kw, U_ref=Uref.astype(dtype).astype(np.float64)))
    orc.reset(*[x[sel].double().numpy() for x in ic])
    for a in acts:
        o_ref, r_ref, _, _ = orc.step(a[sel].double().numpy())
    o = obs[sel].cpu().double().numpy()
    if dtype == "float64":
        np.testing.assert_array_equal(o, o_ref)
        np.testing.assert_array_equal(env.p[sel].cpu().numpy(), orc.p)
        np.testing.assert_allclose(r[sel].cpu().numpy(), r_ref, rtol=1e-12)
    else:
        np.testing.assert_allclose(o, o_ref, rtol=1e-5, atol=1e-5 * np.abs(o_ref).max())
        np.testing.assert_allclose(r[sel].cpu().numpy(), r_ref, rtol=1e-4)


def test_ns_c5_grid_256_parity():
    """BASELINE config 5 grid (256x256, K=50): float64 bit-exact vs the oracle, float32 within the stated tolerance."""
    from oracle import pde_oracle as po
    kw, u0, v0, p0, acts = _random_case(256, 2, 50, 256, NS_BC)
    orc = po.NavierStokesOracle(**kw)
    env = _mk(kw, 2, torch.float64)
    orc.reset(u0, v0, p0)
    env.reset(u0, v0, p0)
    for a in acts[:2]:
        o_ref, r_ref, _, _ = orc.step(a)
        obs, r, te = env.step(a)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
        np.testing.assert_array_equal(env.p.cpu().numpy(), orc.p)
        np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-12)
    env32 = _mk(kw, 2, torch.float32)
    u32, v32, p32 = (x.astype(np.float32) for x in (u0, v0, p0))
    orc.reset(u32.astype(np.float64), v32.astype(np.float64), p32.astype(np.float64))
    env32.reset(u32, v32, p32)
    a32 = acts[0].astype(np.float32)
    o_ref, r_ref, _, _ = orc.step(a32.astype(np.float64))
    obs, r, te = env32.step(a32)
    np.testing.assert_allclose(obs.cpu().numpy().astype(np.float64), o_ref, rtol=1e-5, atol=2e-6 * np.abs(o_ref).max())
    np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-4)


def _rect_case(ny, nx, B, K, seed, bc, action_dim=1):
    rng = np.random.default_rng(seed)
    dx, dy = 1.0 / (nx - 1), 1.0 / (ny - 1)
    dt = 0.2 * 0.5 * min(dx, dy) ** 2 / 0.1
    nt = 8
    Xg, Yg = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
    u0 = np.stack([np.sin(2 * np.pi * Xg * rng.uniform(0.5, 2)) * np.cos(np.pi * Yg) + rng.uniform(-1, 1) for _ in range(B)])
    v0 = np.stack([np.cos(np.pi * Xg) * np.sin(2 * np.pi * Yg * rng.uniform(0.5, 2)) + rng.uniform(-1, 1) for _ in range(B)])
    p0 = rng.uniform(-1, 1, (B, ny, nx))
    kw = dict(T=nt * dt, dt=dt, X=1, dx=dx, Y=1, dy=dy, boundary_condition=bc, U_ref=rng.uniform(-1, 1, (nt, ny, nx, 2)),
              action_ref=rng.uniform(1, 3, nt), gamma=0.1, maximum_pressure_iteration=K, action_dim=action_dim)
    acts = rng.uniform(2, 4, (3, B, action_dim))
    return kw, u0, v0, p0, acts


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("nx,B,K,interleaved,per_node", [(21, 1, 7, True, False), (21, 7, 2, True, True), (21, 64, 33, False, False),
                                                         (21, 4, 1, False, True), (3, 5, 4, True, False), (10, 13, 1, True, False),
                                                         (22, 3, 0, True, False), (32, 5, 9, False, False), (33, 2, 5, True, False),
                                                         (64, 3, 6, True, False)])     # per-node actions need a square grid
def test_ns_column_kernel_equals_workgroup_kernel_bitwise(dtype, nx, B, K, interleaved, per_node):
    """ns_col_step (21 rows, one lane per column, floor(64 / nx) instances per wave, no barriers) against ns_generic_step:
    identical velocity, pressure, flags; rewards to the rounding of a different summation order.  Ragged batches (idle lane
    groups), the first / later sweep forms (K = 0, 1, 2, more), mixed boundary conditions with scalar and per-node actions."""
    import os
    from pdecontrolgym_amd.batch2d import NSBatch2D
    td = getattr(torch, dtype)
    adim = 21 if per_node else 1
    kw, u0, v0, p0, acts = _rect_case(21, nx, B, K, 900 + nx + K, BC_MIX, adim)
    outs = []
    for col in (True, False):
        _dbg("DEBUG_NS_COL_MIN_BATCH", "0")
        _dbg("DEBUG_NS_NO_COL", "0" if col else "1")
        try:
            env = NSBatch2D(num_envs=B, device="cuda", dtype=td, interleaved_state=interleaved, **kw)
            env.reset(u0, v0, p0)
            res = []
            for a in acts:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), env.u.cpu().numpy().copy(),
                            env.time_index.cpu().numpy().copy(), te.cpu().numpy().copy(), r.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_COL_MIN_BATCH", -1)
            _dbg("DEBUG_NS_NO_COL", "0")
    for a, b in zip(*outs):
        for x, y in zip(a[:-1], b[:-1]):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_allclose(a[-1], b[-1], rtol=1e-5 if dtype == "float32" else 1e-12)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("ny,nx,B,K", [(11, 11, 7, 5), (16, 30, 5, 3), (26, 26, 4, 6), (31, 64, 2, 4), (31, 8, 9, 2), (8, 8, 11, 4),
                                       (32, 32, 3, 5)])
def test_ns_column_kernel_other_heights_bitwise(dtype, ny, nx, B, K):
    """The other instantiated grid heights (8, 11, 16, 26, 31, 32 rows) against the workgroup kernel."""
    import os
    from pdecontrolgym_amd.batch2d import NSBatch2D
    td = getattr(torch, dtype)
    kw, u0, v0, p0, acts = _rect_case(ny, nx, B, K, 700 + ny + nx, BC_MIX, 1)
    outs = []
    for col in (True, False):
        _dbg("DEBUG_NS_COL_MIN_BATCH", "0")
        _dbg("DEBUG_NS_NO_COL", "0" if col else "1")
        try:
            env = NSBatch2D(num_envs=B, device="cuda", dtype=td, **kw)
            env.reset(u0, v0, p0)
            res = []
            for a in acts:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_COL_MIN_BATCH", -1)
            _dbg("DEBUG_NS_NO_COL", "0")
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_allclose(a[2], b[2], rtol=1e-5 if dtype == "float32" else 1e-12)


def test_ns_column_kernel_reproduces_target_npz_and_oracle(golden_ns):
    """The reference's shipped example (21 x 21, K = 2000, float64) on the column kernel: the committed target.npz frames bit for
    bit (B = 5: two waves, the second with one live lane group), and a random mixed-boundary case against the oracle."""
    import os
    from oracle import pde_oracle as po
    _dbg("DEBUG_NS_COL_MIN_BATCH", "0")
    try:
        g = golden_ns["N1"]
        B = 5
        kw = dict(T=0.2, dt=1e-3, X=1, dx=0.05, Y=1, dy=0.05, boundary_condition=NS_BC, U_ref=np.zeros((200, 21, 21, 2)),
                  action_ref=2.0 * np.ones(1000), gamma=0.1)
        env = _mk(kw, B, torch.float64)
        env.reset(g.u0, g.v0, np.zeros((21, 21)))
        keep = set(int(k) for k in g.keep)
        for t in range(1, 200):
            obs, r, te = env.step(np.full(B, g.actions[t - 1]))
            if t in keep:
                o = obs.cpu().numpy()
                for b in (0, 2, B - 1):
                    np.testing.assert_array_equal(o[b, :, :, 0], g[f"u{t}"], err_msg=f"u frame {t}")
                    np.testing.assert_array_equal(o[b, :, :, 1], g[f"v{t}"], err_msg=f"v frame {t}")
        assert te.cpu().numpy().all()
        np.testing.assert_array_equal(env.p.cpu().numpy()[3], g.p_final)
        kw, u0, v0, p0, acts = _random_case(21, 5, 7, 121, BC_MIX, 1)
        orc = po.NavierStokesOracle(**kw)
        env = _mk(kw, 5, torch.float64)
        orc.reset(u0, v0, p0)
        env.reset(u0, v0, p0)
        for a in acts:
            o_ref, r_ref, te_ref, _ = orc.step(a)
            obs, r, te = env.step(a)
            np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
            np.testing.assert_array_equal(env.p.cpu().numpy(), orc.p)
            np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-12)
    finally:
        _dbg("DEBUG_NS_COL_MIN_BATCH", -1)


@pytest.mark.parametrize("B", [1030, 3100])
def test_ns_column_kernel_default_dispatch_large_batch(B):
    """Without any switch a float64 batch of >= 400 instances on a 21-row grid takes the column kernel (float32: any batch): up to
    1024 waves the one-wave-per-SIMD build without spills (B = 1030: 344 waves, the last one with one live lane group), beyond
    that the two-wave build (B = 3100: 1034 waves).  Both equal the workgroup kernel bit for bit."""
    import os
    from pdecontrolgym_amd.batch2d import NSBatch2D
    kw, u0, v0, p0, acts = _rect_case(21, 21, B, 5, 4321, BC_MIX, 1)
    outs = []
    for no_col in ("0", "1"):
        _dbg("DEBUG_NS_COL_MIN_BATCH", -1)
        _dbg("DEBUG_NS_NO_COL", no_col)
        try:
            env = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float64, **kw)
            env.reset(u0, v0, p0)
            res = []
            for a in acts[:2]:
                obs, r, te = env.step(a)
                res.append((obs.cpu().numpy().copy(), env.p.cpu().numpy().copy(), r.cpu().numpy().copy()))
            outs.append(res)
        finally:
            _dbg("DEBUG_NS_NO_COL", "0")
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_allclose(a[2], b[2], rtol=1e-12)


@pytest.mark.parametrize("n,dtype,K,B,T,adim", [(21, "float64", 40, 7, 9, 1), (21, "float32", 25, 64, 9, 1), (11, "float64", 7, 5, 6, 11),
                                                  (32, "float32", 12, 3, 7, 1), (16, "float64", 0, 4, 5, 1), (26, "float32", 3, 1, 4, 26)])
def test_ns_rollout_kernel_equals_step_calls_bitwise(n, dtype, K, B, T, adim):
    """pdegym_ns2d_rollout_* (round 4: T env-steps of the small-grid column kernel in ONE launch, commands given ahead -- the loop
    of examples/NavierStokes/NS2Dppo.py:52-66 / the forward sweep of NS2Doptimization.py) against T step calls: every observation
    slot, reward and flag, the pressure, the time index, the restart counters and the kept terminal observations agree bit for
    bit, across episode ends with the fused auto-reset (episodes of 4 steps, pools of 2B rows)."""
    from pdecontrolgym_amd.batch2d import NSBatch2D
    td = getattr(torch, dtype)
    kw, u0, v0, p0, _ = _random_case(n, B, K, 900 + n + K, BC_MIX, adim)
    kw = dict(kw, action_dim=adim, T=5 * kw["dt"])                  # nt = 5: an episode ends after 4 steps
    kw["U_ref"], kw["action_ref"] = kw["U_ref"][:5], kw["action_ref"][:5]
    rng = np.random.default_rng(n * 13 + K)
    pools = [rng.uniform(-1, 1, (2 * B, n, n)) for _ in range(3)]
    acts = torch.as_tensor(rng.uniform(2, 4, (T, B, adim)), dtype=td, device="cuda")
    outs = []
    # the step calls take the column kernel too (small float64 batches default to the workgroup kernel, whose FIELDS are the same
    # bits but whose reward sum runs in another order)
    _dbg("DEBUG_NS_COL_MIN_BATCH", "0")
    for mode in ("steps", "rollout"):
        env = NSBatch2D(num_envs=B, device="cuda", dtype=td, **kw)
        assert env.can_rollout()
        env.reset(u0, v0, p0)
        env.enable_auto_reset(*pools)
        obs = torch.zeros(T + 1, B, n, n, 2, dtype=td, device="cuda")
        obs[0].copy_(env.t["obs"])
        rew = torch.zeros(T, B, dtype=td, device="cuda")
        te = torch.zeros(T, B, dtype=torch.uint8, device="cuda")
        if mode == "steps":
            env.t["obs"] = obs[0]
            for t in range(T):
                env.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t])
        else:
            env.rollout(obs, acts, rew, te)
        outs.append([x.cpu().numpy().copy() for x in (obs, rew, te, env.p, env.t["time_index"], env.t["reset_count"], env.t["final_obs"])])
        nxt = env.step(acts[0])[0].cpu().numpy().copy()           # the engine carries on from slot T with ordinary step calls
        outs[-1].append(nxt)
    _dbg("DEBUG_NS_COL_MIN_BATCH", -1)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert outs[0][2].sum() > 0 and np.isfinite(outs[0][0]).all()
    big = NSBatch2D(num_envs=1, device="cuda", dtype=td, **dict(_random_case(64, 1, 3, 1, BC_MIX)[0]))
    assert not big.can_rollout()
    with pytest.raises(ValueError):
        big.rollout(None, None, None, None)


def test_ns_rollout_vs_default_dispatch_small_float64_batch():
    """with the DEFAULT dispatch a float64 batch below the column kernel's minimum (400 per 1024 SIMDs) steps on the
    workgroup kernel while the rollout always runs the column kernel: every FIELD, flag and counter is still the same bits, the
    rewards agree to rounding (another summation order), as include/pdegym.h states."""
    from pdecontrolgym_amd.batch2d import NSBatch2D
    n, K, B, T = 21, 30, 9, 6
    kw, u0, v0, p0, _ = _random_case(n, B, K, 7321, BC_MIX, 1)
    kw = dict(kw, action_dim=1)
    acts = torch.as_tensor(np.random.default_rng(5).uniform(2, 4, (T, B, 1)), dtype=torch.float64, device="cuda")
    _dbg("DEBUG_NS_COL_MIN_BATCH", -1)
    outs = []
    for mode in ("steps", "rollout"):
        env = NSBatch2D(num_envs=B, device="cuda", dtype=torch.float64, **kw)
        env.reset(u0, v0, p0)
        obs = torch.zeros(T + 1, B, n, n, 2, dtype=torch.float64, device="cuda")
        obs[0].copy_(env.t["obs"])
        rew = torch.zeros(T, B, dtype=torch.float64, device="cuda")
        te = torch.zeros(T, B, dtype=torch.uint8, device="cuda")
        if mode == "steps":
            env.t["obs"] = obs[0]
            for t in range(T):
                env.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=te[t])
        else:
            env.rollout(obs, acts, rew, te)
        outs.append([x.cpu().numpy().copy() for x in (obs, te, env.p, env.t["time_index"], rew)])
    for a, b in zip(outs[0][:4], outs[1][:4]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(outs[0][4], outs[1][4], rtol=1e-13)
