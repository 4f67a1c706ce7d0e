"""Oracle-on-tensors test double for the HIP backend (CPU tests of the HOST logic only).

Implements the backend interface of pdecontrolgym_amd/backend.py on CPU torch tensors by running the NumPy
oracle.  It lives under tests/ because only tests may touch the oracle; the product ships no CPU path.
"""
import numpy as np
import torch

from oracle import pde_oracle as po
from pdecontrolgym_amd import _native as N


class FakeBackend:
    name = "oracle-test-double"

    def __init__(self):
        self.core = None

    def bind(self, core):
        self.core = core
        return self

    # ---- 1D -------------------------------------------------------------------------------------
    def _orc1d(self, P):
        c = self.core
        if P.reward_kind == N.REWARD_TUNED1D:
            rw = po.TunedReward1DOracle(P.reward_nt, P.truncate_penalty, P.terminate_reward)
        elif P.reward_kind >= N.REWARD_NORM_L1:
            rw = po.NormRewardOracle(P.reward_nt, {2: "1", 3: "2", 4: "inf"}[P.reward_kind], P.truncate_penalty, P.terminate_reward,
                                     {N.HORIZON_TEMPORAL: "temporal", N.HORIZON_DIFFERENTIAL: "differential", N.HORIZON_T: "t-horizon"}[P.reward_horizon],
                                     P.reward_t_horizon or 5)
        else:
            rw = None
        cls = po.ParabolicOracle if c.kind == "parabolic" else (po.BurgersOracle if getattr(c, "flux", "linear") == "burgers" else po.TransportOracle)
        return cls(c.T, c.dt, c.X, c.dx, c.control_sample_rate, control_type=c.control_type, sensing_loc=c.sensing_loc,
                   sensing_type=c.sensing_type, normalize=c.normalize, max_control_value=c.max_control_value,
                   limit_pde_state_size=c.limit_pde_state_size, max_state_value=c.max_state_value, reward=rw,
                   keep_history=False)

    @staticmethod
    def _beta(T, B):
        b = T["beta"].numpy()
        return np.tile(b[None], (B, 1)) if b.ndim == 1 else b

    def _load(self, orc, T, B):
        orc.B = B
        orc.beta = self._beta(T, B)
        src = T["state_in"] if T.get("state_in") is not None else (T["u"] if T.get("u") is not None else T["obs"])
        orc.row = src.numpy().copy()                 # state_in: the previous observation is the state (include/pdegym.h)
        orc.prev_row = orc.row.copy()
        orc.time_index = T["time_index"].numpy().astype(np.int64)
        orc.bsum = T["bsum"].numpy().copy()
        orc.ring = T["ring"].numpy().copy()
        orc._thor = isinstance(orc.reward, po.NormRewardOracle) and orc.reward.horizon == "t-horizon"
        if orc._thor:        # include/pdegym.h PDEGYM_HORIZON_T: the ring holds the reward's own norms; the start row's is recorded by the call
            orc.kring = orc.ring.copy()
            orc.kring[np.arange(B), orc.time_index & 127] = orc.reward.row_norms(orc.row)
        if T.get("history") is not None:
            orc.keep_history = True
            orc.hist = T["history"].numpy()          # shares memory with the tensor: rows are written in place

    def _store(self, orc, T):
        if T.get("u") is not None:
            T["u"].copy_(torch.from_numpy(orc.row))
        T["time_index"].copy_(torch.from_numpy(orc.time_index.astype(np.int32)))
        T["bsum"].copy_(torch.from_numpy(orc.bsum))
        T["ring"].copy_(torch.from_numpy(orc.kring if getattr(orc, "_thor", False) else orc.ring))

    def step1d(self, kind, P, T, B):
        orc = self._orc1d(P)
        self._load(orc, T, B)
        with np.errstate(all="ignore"):
            obs, r, te, tr = orc.step(T["action"].numpy(), action_kind=("f32", "f64", "weak")[P.action_kind])
        T["obs"].copy_(torch.from_numpy(np.asarray(obs, dtype=np.float32).reshape(B, -1)))
        if r is not None:
            T["reward"].copy_(torch.from_numpy(r.astype(np.float32)))
        T["norm_now"].copy_(torch.from_numpy(orc.norm_now.astype(np.float32)))
        T["norm_back"].copy_(torch.from_numpy(orc.norm_back.astype(np.float32)))
        T["terminated"].copy_(torch.from_numpy(te.astype(np.uint8)))
        T["truncated"].copy_(torch.from_numpy(tr.astype(np.uint8)))
        self._store(orc, T)
        if T.get("reset_init") is not None:
            done = te | tr
            if done.any():
                if T.get("final_obs") is not None:
                    T["final_obs"][torch.from_numpy(done)] = T["obs"][torch.from_numpy(done)]
                # pool row of the k-th restart of instance b: (b + k*B) mod P  (include/pdegym.h)
                cnt = T["reset_count"].numpy().astype(np.int64) if T.get("reset_count") is not None else np.zeros(B, dtype=np.int64)
                rows = torch.from_numpy((np.arange(B) + cnt * B) % T["reset_init"].shape[0])
                dm = torch.from_numpy(done)
                if T.get("reset_beta") is not None:
                    T["beta"][dm] = T["reset_beta"][rows][dm]
                if T.get("reset_count") is not None:
                    T["reset_count"][dm] += 1
                self.reset1d(P, T, T["reset_init"][rows].contiguous(), torch.from_numpy(done.astype(np.uint8)), B, _keep_flags=True)

    def rollout1d(self, kind, P, T, obs, actions, rewards, terminated, truncated, B, policy=None, obs_noise=None, obs_seen=None):
        # the C ABI's contract: T step calls -- full-state sensing: the row read from slot t and written to slot t + 1; scalar
        # sensing: the state in T["u"], slot t + 1 receives the sensed value -- and the policy, when given, evaluated first on
        # slot t (+ obs_noise[t]; obs_seen[t] receives what it read)
        import ctypes as C
        full = P.sensing == N.SENSE_FULL
        for t in range(actions.shape[0]):
            if policy is not None:
                net = type(policy)()
                C.memmove(C.addressof(net), C.addressof(policy), C.sizeof(policy))
                if policy.noise:
                    net.noise = policy.noise + 4 * t * B * policy.noise_stride
                seen = obs[t] if obs_noise is None else obs[t] + obs_noise[t]
                if obs_seen is not None:
                    obs_seen[t].copy_(seen)
                self.mlp_forward(net, seen.contiguous(), actions[t].view(B, 1), B)
            S = dict(T)
            S.update(obs=obs[t + 1], action=actions[t], reward=rewards[t], terminated=terminated[t], truncated=truncated[t], history=None)
            if full:
                S.update(state_in=obs[t], u=None)
            else:
                S.update(state_in=None)
            self.step1d(kind, P, S, B)

    def reset1d(self, P, T, init, mask, B, _keep_flags=False):
        orc = self._orc1d(P)
        m = np.ones(B, dtype=bool) if mask is None else mask.numpy().astype(bool)
        init = init.numpy()
        fresh = self._orc1d(P)
        obs = fresh.reset(init, self._beta(T, B))
        self._load(orc, T, B)
        orc.row[m] = fresh.row[m]
        orc.time_index[m] = 0
        orc.bsum[m] = fresh.bsum[m]
        orc.ring[m, 0] = po._rownorm(fresh.row[m])
        if T.get("history") is not None:
            h = T["history"].numpy()
            h[m] = 0
            h[m, 0] = fresh.row[m]
        self._store(orc, T)
        o = np.asarray(obs, dtype=np.float32).reshape(B, -1)
        T["obs"][torch.from_numpy(m)] = torch.from_numpy(o[m])
        if not _keep_flags:
            T["terminated"][torch.from_numpy(m)] = 0
            T["truncated"][torch.from_numpy(m)] = 0

    def rownorm2(self, rows, out):
        out.copy_(torch.from_numpy(po._rownorm(rows.numpy())))

    # ---- NS2D -----------------------------------------------------------------------------------
    def _orc_ns(self, T):
        c = self.core
        np_dt = np.float64 if T["p"].dtype == torch.float64 else np.float32
        kw = dict(c.ctor)
        return po.NavierStokesOracle(U_ref=T["U_ref"].numpy(), action_ref=T["action_ref"].numpy(), dtype=np_dt, **kw)

    def ns2d_step(self, P, T, B):
        orc = self._orc_ns(T)
        if T.get("state_in") is not None:
            st = T["state_in"].numpy()
            orc.reset(st[..., 0], st[..., 1], T["p"].numpy())
        else:
            orc.reset(T["u"].numpy(), T["v"].numpy(), T["p"].numpy())
        orc.time_index = T["time_index"].numpy().astype(np.int64)
        obs, r, te, _ = orc.step(T["action"].numpy())
        for k, a in (("u", orc.u), ("v", orc.v), ("p_out" if T.get("p_out") is not None else "p", orc.p), ("obs", obs), ("reward", r)):
            if T.get(k) is not None:
                T[k].copy_(torch.from_numpy(np.ascontiguousarray(a)).to(T[k].dtype))
        T["time_index"].copy_(torch.from_numpy(orc.time_index.astype(np.int32)))
        T["terminated"].copy_(torch.from_numpy(te.astype(np.uint8)))
        if T.get("reset_u0") is not None and te.any():          # fused auto-reset: pool row (b + k*B) mod P
            dm = torch.from_numpy(te.astype(bool))
            cnt = T["reset_count"].numpy().astype(np.int64) if T.get("reset_count") is not None else np.zeros(B, dtype=np.int64)
            rows = torch.from_numpy((np.arange(B) + cnt * B) % T["reset_u0"].shape[0])
            if T.get("final_obs") is not None:
                T["final_obs"][dm] = T["obs"][dm]
            pk = "p_out" if T.get("p_out") is not None else "p"
            T[pk][dm] = T["reset_p0"][rows][dm]
            if T.get("u") is not None:
                T["u"][dm] = T["reset_u0"][rows][dm]
                T["v"][dm] = T["reset_v0"][rows][dm]
            T["obs"][dm] = torch.stack([T["reset_u0"][rows][dm], T["reset_v0"][rows][dm]], dim=-1)
            T["time_index"][dm] = 0
            if T.get("reset_count") is not None:
                T["reset_count"][dm] += 1

    def ns2d_rollout(self, P, T, obs, actions, rewards, terminated, B):
        # the C ABI's contract: T step calls with the state read from slot t and written to slot t + 1, the pressure in T["p"]
        for t in range(actions.shape[0]):
            S = dict(T)
            S.update(u=None, v=None, p_out=None, state_in=obs[t], obs=obs[t + 1], action=actions[t], reward=rewards[t], terminated=terminated[t])
            self.ns2d_step(P, S, B)

    def ns2d_reset(self, P, T, u0, v0, p0, mask, B):
        m = torch.ones(B, dtype=torch.bool) if mask is None else mask.bool()
        if T.get("u") is not None:
            T["u"][m] = u0[m]
            T["v"][m] = v0[m]
        T["p"][m] = p0[m]
        T["obs"][m] = torch.stack([u0[m], v0[m]], dim=-1)
        T["time_index"][m] = 0
        T["terminated"][m] = 0

    def ns2d_solve_pressure(self, P, u, v, p_in, p_out, scratch, B):
        orc = self._orc_ns({"u": u, "p": p_in, "U_ref": torch.zeros(1, self.core.ny, self.core.nx, 2), "action_ref": torch.zeros(1)})
        p_out.copy_(torch.from_numpy(orc.solve_pressure(u.numpy(), v.numpy(), p_in.numpy())))


    # ---- Traffic ARZ ----------------------------------------------------------------------------
    def _orc_traffic(self):
        c = self.core
        return po.TrafficOracle(c.T, c.dt, c.X, c.dx, c.simulation_type, c.vm, c.rm, c.tau, c.limit, c.control_freq)

    def traffic_step(self, P, T, B):
        orc = self._orc_traffic()
        orc.reset(T["rs"].numpy(), T["qs_clip"].numpy())
        orc.r, orc.y = T["r"].numpy().copy(), T["y"].numpy().copy()
        orc.time_index = T["time"].numpy().copy()
        a = T["action"].numpy()
        obs, r, d, t = orc.step(a if self.core.action_dim == 2 else a[:, :1])
        for k, v in (("r", orc.r), ("y", orc.y), ("time", orc.time_index), ("obs", obs), ("reward", r)):
            T[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
        T["done"].copy_(torch.from_numpy(d.astype(np.uint8)))
        T["truncated"].copy_(torch.from_numpy(t.astype(np.uint8)))
        if T.get("reset_rs") is not None:           # fused auto-reset (include/pdegym.h)
            fin = d | t
            if fin.any():
                m = torch.from_numpy(fin)
                if T.get("final_obs") is not None:
                    T["final_obs"][m] = T["obs"][m]
                cnt = T["reset_count"].numpy().astype(np.int64) if T.get("reset_count") is not None else np.zeros(B, dtype=np.int64)
                rows = torch.from_numpy((np.arange(B) + cnt * B) % T["reset_rs"].shape[0])
                T["rs"][m] = T["reset_rs"][rows][m]
                if T.get("reset_count") is not None:
                    T["reset_count"][m] += 1
                keep = (T["done"].clone(), T["truncated"].clone())
                self.traffic_reset(P, T, T["reset_profile"], m.to(torch.uint8), B)
                T["done"].copy_(keep[0])
                T["truncated"].copy_(keep[1])

    def traffic_rollout(self, P, T, obs, actions, rewards, done, truncated, B, policy=None):
        # the C ABI's contract: T step calls (and the policy, when given, evaluated on slot t first)
        import ctypes as C
        A = actions.shape[2]
        for t in range(actions.shape[0]):
            if policy is not None:
                net = type(policy)()
                C.memmove(C.addressof(net), C.addressof(policy), C.sizeof(policy))
                net.x_f64 = net.y_f64 = 1
                if policy.noise:
                    net.noise = policy.noise + 4 * t * B * policy.noise_stride
                self.mlp_forward(net, obs[t], actions[t], B)
            S = dict(T)
            S.update(action=actions[t], obs=obs[t + 1], reward=rewards[t], done=done[t], truncated=truncated[t])
            self.traffic_step(P, S, B)

    def traffic_reset(self, P, T, profile, mask, B):
        orc = self._orc_traffic()
        obs = orc.reset(T["rs"].numpy())
        m = torch.ones(B, dtype=torch.bool) if mask is None else mask.bool()
        T["r"][m] = torch.from_numpy(orc.r)[m]
        T["y"][m] = torch.from_numpy(orc.y)[m]
        T["obs"][m] = torch.from_numpy(obs)[m]
        T["time"][m] = 0
        T["done"][m] = 0
        T["truncated"][m] = 0

    # ---- Brain tumour ---------------------------------------------------------------------------
    def _orc_tumor(self, P):
        return po.BrainTumorOracle(self.core.T, self.core.dt, self.core.X, self.core.dx, P.total_dosage,
                                   P.thr_t1 / P.k, P.thr_t2 / P.k, P.dose_end, P.D, P.rho, P.alpha, P.alpha_beta_ratio, P.k,
                                   P.detect_radius, P.death_radius)

    def _tumor_day(self, P, T, B, part, control, use_kill):
        """One oracle day for the instances in ``part`` (bool [B]); everything else is left untouched."""
        orc = self._orc_tumor(P)
        orc.thr1, orc.thr2 = P.thr_t1, P.thr_t2
        orc.reset(T["u"].numpy(), T["t_benchmark"].numpy())
        orc.time_index = T["time_index"].numpy().astype(np.int64)
        orc.stage = T["stage"].numpy().astype(np.int64)
        orc.remaining = T["remaining"].numpy().copy()
        d = T["days"].numpy().astype(np.int64)
        orc.growthDays, orc.therapyDays, orc.postDays, orc.simulationDays, orc.cDeathDay = (d[:, i].copy() for i in range(5))
        live = orc.time_index < orc.nt - 1
        obs, r, te, tr = orc.step(control)
        out = np.stack([orc.T1, orc.radius_abs(orc.u, orc.thr2), orc.treat_r, orc.applied], axis=1)
        days = np.stack([orc.growthDays, orc.therapyDays, orc.postDays, orc.simulationDays, orc.cDeathDay], axis=1).astype(np.int32)
        m = torch.from_numpy(part)
        upd = torch.from_numpy(part & live)
        T["u"][upd] = torch.from_numpy(obs)[upd]
        T["time_index"][upd] = torch.from_numpy(orc.time_index.astype(np.int32))[upd]
        T["stage"][upd] = torch.from_numpy(orc.stage.astype(np.int32))[upd]
        T["remaining"][upd] = torch.from_numpy(orc.remaining)[upd]
        T["days"][upd] = torch.from_numpy(days)[upd]
        T["out"][upd] = torch.from_numpy(out)[upd]
        if T.get("history") is not None:
            for b_ in np.nonzero(part & live)[0]:
                T["history"][b_, int(orc.time_index[b_])] = torch.from_numpy(obs[b_])
                T["t1_log"][b_, int(orc.time_index[b_])] = float(orc.T1[b_] / self.core.dx)
        return m, upd, r, te, tr

    def tumor_step(self, P, T, B):
        part = np.ones(B, bool) if T.get("active") is None else T["active"].numpy().astype(bool)
        m, upd, r, te, tr = self._tumor_day(P, T, B, part, T["control"].numpy(), True)
        T["reward"][m] = torch.from_numpy(r)[m]
        T["terminated"][m] = torch.from_numpy(te.astype(np.uint8))[m]
        T["truncated"][m] = torch.from_numpy(tr.astype(np.uint8))[m]

    def tumor_advance(self, P, T, mode, max_days, B):
        if mode == 0:
            return self.tumor_step(P, T, B)
        act = np.ones(B, bool) if T.get("active") is None else T["active"].numpy().astype(bool)
        stage = T["stage"].numpy()
        live = T["time_index"].numpy() < self.core.nt - 1
        part = act & live & {1: stage == 0, 2: stage == 2, 3: np.ones(B, bool)}[mode]
        zero = np.zeros(B)
        for _ in range(max_days):
            if not part.any():
                break
            m, upd, r, te, tr = self._tumor_day(P, T, B, part, zero, False)
            T["reward"][m] = torch.from_numpy(r)[m]
            T["terminated"][m] = torch.from_numpy(te.astype(np.uint8))[m]
            T["truncated"][m] = torch.from_numpy(tr.astype(np.uint8))[m]
            done = te | tr
            part = part & ~done
            if mode == 1:
                part = part & (T["stage"].numpy() == 0)

    def tumor_reset(self, P, T, init, mask, B):
        m = torch.ones(B, dtype=torch.bool) if mask is None else mask.bool()
        src = init.expand(B, -1) if init.dim() == 1 else init
        T["u"][m] = src[m]
        T["time_index"][m] = 0
        T["stage"][m] = 0
        T["remaining"][m] = P.total_dosage
        T["days"][m] = torch.tensor([0, 0, 0, 0, -1], dtype=torch.int32)

    def mlp_forward(self, net, x, y, B):
        """CPU double of pdegym_mlp_forward: float32 accumulation over k in ascending order, bias last, like the kernel."""
        import ctypes
        import numpy as np
        import torch
        h = x[:B].detach().cpu().numpy().astype(np.float32)       # float64 observations are rounded on the way in
        for i in range(net.n_layers):
            L = net.layer[i]
            k4 = (L.in_dim + 3) // 4
            blk = np.ctypeslib.as_array(ctypes.cast(L.w, ctypes.POINTER(ctypes.c_float)), (k4, L.out_dim, 4))
            w = blk.transpose(1, 0, 2).reshape(L.out_dim, 4 * k4)[:, :L.in_dim]
            acc = np.zeros((h.shape[0], L.out_dim), np.float32)
            for k in range(L.in_dim):     # an fma is exact before its single rounding: emulate with a double product
                acc = (acc.astype(np.float64) + h[:, k:k + 1].astype(np.float64) * w[:, k][None, :].astype(np.float64)).astype(np.float32)
            if L.b:
                acc = acc + np.ctypeslib.as_array(ctypes.cast(L.b, ctypes.POINTER(ctypes.c_float)), (L.out_dim,))[None, :]
            h = np.tanh(acc) if L.act == 1 else (np.maximum(acc, 0) if L.act == 2 else acc)
            h = h.astype(np.float32)
        if net.noise:
            nz = np.ctypeslib.as_array(ctypes.cast(net.noise, ctypes.POINTER(ctypes.c_float)), (h.shape[0], int(net.noise_stride)))
            h = h + nz[:, :h.shape[1]]
        if net.clamp:
            h = np.clip(h, net.lo, net.hi)
        y[:B].copy_(torch.from_numpy(h).to(y.dtype))

