"""HIP backend: turns (params struct, dict of torch tensors) into C-ABI calls on libpdegym_hip.so.

The batched environment cores (batch1d.py / batch2d.py) talk to a backend object with this small
interface.  ``HipBackend`` is the only backend the product ships; it refuses CPU tensors.  (The CPU test
suite injects an oracle-backed double from tests/ to exercise host logic without a GPU.)
"""
from __future__ import annotations

import ctypes as C

from . import _native as N


def _on_device_of(key):
    """Decorator: run the C-ABI call with the device of tensor-dict entry ``key`` (or of the tensor argument at that position)
    current.  The kernels launch on the thread's current HIP device and the stream handle passed is that device's current
    stream, so an engine built on cuda:1 while cuda:0 is current must switch first."""
    def deco(fn):
        def wrapped(self, *args, **kw):
            import torch
            dev = None
            for a in args:
                if isinstance(a, dict) and key in a:
                    dev = a[key].device
                    break
                if torch.is_tensor(a):
                    dev = a.device
                    break
            if dev is None or dev.type != "cuda" or dev.index == torch.cuda.current_device():
                return fn(self, *args, **kw)          # already current (the common case): no device switch, no context manager
            with torch.cuda.device(dev):
                return fn(self, *args, **kw)
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        wrapped.device_guard_key = key          # (tests/test_host_api.py checks that every C-ABI entry point carries the guard)
        return wrapped
    return deco


class HipBackend:
    name = "hip-gfx950"

    def __init__(self):
        self.lib = N.load()

    # ---- 1D -------------------------------------------------------------------------------------
    @staticmethod
    def _bufs1d(T) -> N.Bufs1D:
        import torch
        b = N.Bufs1D()
        si = T.get("state_in")
        if si is not None:       # the rows live in the (double-buffered) observation tensors: read from the previous one
            b.state_in, b.u = N.dptr(si, torch.float32), None
        else:
            b.u = N.dptr(T["u"], torch.float32) if T.get("u") is not None else None
        b.beta = N.dptr(T["beta"])                 # float32, or float64 in the reference's mixed-precision mode (P.beta_f64)
        b.beta_stride = 0 if T["beta"].dim() == 1 else T["beta"].stride(0)
        b.action = N.dptr(T["action"])             # float32, or float64 when P.action_kind != ACTION_F32
        b.time_index = N.dptr(T["time_index"], torch.int32)
        b.bsum = N.dptr(T["bsum"], torch.float64)
        b.ring = N.dptr(T["ring"], torch.float32)
        b.obs = N.dptr(T["obs"], torch.float32)
        b.reward = N.dptr(T["reward"], torch.float32)
        b.norm_now = N.dptr(T["norm_now"], torch.float32)
        b.norm_back = N.dptr(T["norm_back"], torch.float32)
        b.terminated = N.dptr(T["terminated"], torch.uint8)
        b.truncated = N.dptr(T["truncated"], torch.uint8)
        h = T.get("history")
        b.history = N.dptr(h, torch.float32) if h is not None else None
        ri, fo = T.get("reset_init"), T.get("final_obs")
        b.reset_init = N.dptr(ri, torch.float32) if ri is not None else None
        b.final_obs = N.dptr(fo, torch.float32) if fo is not None else None
        rb, rc = T.get("reset_beta"), T.get("reset_count")
        b.reset_beta = N.dptr(rb, T["beta"].dtype) if rb is not None else None
        b.reset_count = N.dptr(rc, torch.int32) if rc is not None else None
        b.reset_pool_rows = int(ri.shape[0]) if ri is not None else 0
        if rb is not None and (T["beta"].dim() == 1 or rb.shape != ri.shape):
            raise N.NativeError("reset_beta needs a per-instance beta [B, n] and the shape of reset_init")
        return b

    @_on_device_of("bsum")        # (a state tensor: the outputs of a host-facing engine may live in pinned host memory)
    def step1d(self, kind: str, P: N.Params1D, T: dict, B: int):
        import torch
        fn = self.lib.pdegym_transport_step if kind == "transport" else self.lib.pdegym_parabolic_step
        if T["beta"].dtype != (torch.float64 if P.beta_f64 else torch.float32):
            raise N.NativeError(f"beta is {T['beta'].dtype} but params.beta_f64 = {P.beta_f64}")
        if T["action"].dtype != (torch.float32 if P.action_kind == N.ACTION_F32 else torch.float64):
            raise N.NativeError(f"action is {T['action'].dtype} but params.action_kind = {P.action_kind}")
        bufs = self._bufs1d(T)
        N.check(fn(C.byref(P), C.byref(bufs), B, N.current_stream_ptr(T["bsum"].device)), f"pdegym_{kind}_step")

    def prepare_step1d(self, kind: str, P: N.Params1D, T: dict, B: int):
        """The step call with its argument structures built ONCE: returns a zero-argument callable that launches
        pdegym_{kind}_step on the current stream of the state's device.  ``P`` is passed by reference (fields such as
        ``action_kind`` may change between calls); the tensors of ``T`` must stay alive and in place."""
        import torch
        fn = self.lib.pdegym_transport_step if kind == "transport" else self.lib.pdegym_parabolic_step
        if T["beta"].dtype != (torch.float64 if P.beta_f64 else torch.float32):
            raise N.NativeError(f"beta is {T['beta'].dtype} but params.beta_f64 = {P.beta_f64}")
        bufs = self._bufs1d(T)
        dev = T["bsum"].device                     # (outputs may live in pinned host memory: the state names the device)
        pP, pB, what = C.byref(P), C.byref(bufs), f"pdegym_{kind}_step"
        current_stream, current_device = torch.cuda.current_stream, torch.cuda.current_device

        def call(_keep=(bufs, T)):
            if current_device() != dev.index:
                with torch.cuda.device(dev):
                    rc = fn(pP, pB, B, current_stream(dev).cuda_stream)
            else:
                rc = fn(pP, pB, B, current_stream(dev).cuda_stream)
            if rc:
                N.check(rc, what)
        return call
    prepare_step1d.device_guard_key = "bsum"     # the guard is inside the prepared call (the state's device made current)

    @_on_device_of("obs")
    def rollout1d(self, kind: str, P: N.Params1D, T: dict, obs, actions, rewards, terminated, truncated, B: int, policy=None,
                  obs_noise=None, obs_seen=None):
        """T env-steps in one launch (pdegym_*_rollout): ``obs`` [T+1, B, obs_dim] (slot 0 = input rows with full-state sensing;
        with scalar sensing the state is ``T["u"]``, advanced in place), ``actions`` / ``rewards`` / ``terminated`` /
        ``truncated`` [T, B].  ``policy``: an ``N.Mlp`` descriptor evaluated inside the launch (``actions`` is then an output;
        its ``noise``, if set, is [T, B]); ``obs_noise`` / ``obs_seen`` [T, B, obs_dim]: the policy reads obs[t] + obs_noise[t],
        which obs_seen[t] receives."""
        import torch
        fn = self.lib.pdegym_transport_rollout if kind == "transport" else self.lib.pdegym_parabolic_rollout
        steps = int(actions.shape[0])
        full = P.sensing == N.SENSE_FULL
        od = P.n if full else 1
        if tuple(obs.shape) != (steps + 1, B, od) or not obs.is_contiguous():
            raise N.NativeError(f"rollout obs must be a contiguous [{steps + 1}, {B}, {od}] tensor, got {tuple(obs.shape)}")
        for name, x in (("actions", actions), ("rewards", rewards), ("terminated", terminated), ("truncated", truncated)):
            if tuple(x.shape) != (steps, B) or not x.is_contiguous():
                raise N.NativeError(f"rollout {name} must be a contiguous [{steps}, {B}] tensor, got {tuple(x.shape)}")
        for name, x in (("obs_noise", obs_noise), ("obs_seen", obs_seen)):
            if x is not None and (tuple(x.shape) != (steps, B, od) or not x.is_contiguous() or x.dtype != torch.float32):
                raise N.NativeError(f"rollout {name} must be a contiguous float32 [{steps}, {B}, {od}] tensor, got {tuple(x.shape)}")
        bufs = self._bufs1d({**T, "state_in": None, "u": None if full else T["u"], "history": None})
        ro = N.Rollout1D()
        ro.T = steps
        ro.obs, ro.actions, ro.rewards = N.dptr(obs, torch.float32), N.dptr(actions, torch.float32), N.dptr(rewards, torch.float32)
        ro.terminated, ro.truncated = N.dptr(terminated, torch.uint8), N.dptr(truncated, torch.uint8)
        ro.policy = C.addressof(policy) if policy is not None else None
        ro.obs_noise = N.dptr(obs_noise, torch.float32) if obs_noise is not None else None
        ro.obs_seen = N.dptr(obs_seen, torch.float32) if obs_seen is not None else None
        N.check(fn(C.byref(P), C.byref(bufs), C.byref(ro), B, N.current_stream_ptr(obs.device)), f"pdegym_{kind}_rollout")

    @_on_device_of("bsum")
    def reset1d(self, P: N.Params1D, T: dict, init, mask, B: int):
        import torch
        bufs = self._bufs1d(T)
        m = N.dptr(mask, torch.uint8) if mask is not None else None
        N.check(self.lib.pdegym_reset1d_masked(C.byref(P), C.byref(bufs), N.dptr(init, torch.float32), m, B,
                                               N.current_stream_ptr(T["bsum"].device)), "pdegym_reset1d_masked")

    @_on_device_of("u")
    def rownorm2(self, rows, out):
        import torch
        B, n = rows.shape
        N.check(self.lib.pdegym_rownorm2_f32(N.dptr(rows, torch.float32), N.dptr(out, torch.float32), n, B,
                                             N.current_stream_ptr(rows.device)), "pdegym_rownorm2_f32")

    # ---- NS2D -----------------------------------------------------------------------------------
    @staticmethod
    def _bufs_ns(T, dtype) -> N.BufsNS2D:
        import torch
        b = N.BufsNS2D()
        for k in ("p", "scratch", "action", "U_ref", "action_ref", "obs", "reward"):
            setattr(b, k, N.dptr(T[k], dtype))
        for k in ("u", "v", "state_in", "p_out", "reset_u0", "reset_v0", "reset_p0", "final_obs"):
            setattr(b, k, N.dptr(T[k], dtype) if T.get(k) is not None else None)
        b.reset_count = N.dptr(T["reset_count"], torch.int32) if T.get("reset_count") is not None else None
        b.reset_pool_rows = int(T["reset_u0"].shape[0]) if T.get("reset_u0") is not None else 0
        b.time_index = N.dptr(T["time_index"], torch.int32)
        b.terminated = N.dptr(T["terminated"], torch.uint8)
        b.nt_ref = int(min(T["U_ref"].shape[0], T["action_ref"].shape[0]))
        return b

    @staticmethod
    def _sfx(dtype):
        import torch
        return "f32" if dtype == torch.float32 else "f64"

    @_on_device_of("p")
    def ns2d_step(self, P: N.ParamsNS2D, T: dict, B: int):
        dtype = T["p"].dtype
        bufs = self._bufs_ns(T, dtype)
        fn = getattr(self.lib, "pdegym_ns2d_step_" + self._sfx(dtype))
        N.check(fn(C.byref(P), C.byref(bufs), B, N.current_stream_ptr(T["p"].device)), "pdegym_ns2d_step")

    def prepare_ns2d_step(self, P: N.ParamsNS2D, T: dict, B: int):
        """pdegym_ns2d_step_* with its argument structures built once (see prepare_step1d): a zero-argument callable."""
        import torch
        dtype = T["p"].dtype
        bufs = self._bufs_ns(T, dtype)
        fn = getattr(self.lib, "pdegym_ns2d_step_" + self._sfx(dtype))
        dev = T["p"].device
        pP, pB = C.byref(P), C.byref(bufs)
        current_stream, current_device = torch.cuda.current_stream, torch.cuda.current_device

        def call(_keep=(bufs, T)):
            if current_device() != dev.index:
                with torch.cuda.device(dev):
                    rc = fn(pP, pB, B, current_stream(dev).cuda_stream)
            else:
                rc = fn(pP, pB, B, current_stream(dev).cuda_stream)
            if rc:
                N.check(rc, "pdegym_ns2d_step")
        return call
    prepare_ns2d_step.device_guard_key = "p"

    @_on_device_of("p")
    def ns2d_rollout(self, P: N.ParamsNS2D, T: dict, obs, actions, rewards, terminated, B: int):
        """T env-steps in one launch (pdegym_ns2d_rollout_*, small grids): ``obs`` [T+1, B, ny, nx, 2] (slot 0 = the input
        state), ``actions`` [T, B, action_dim], ``rewards`` / ``terminated`` [T, B]."""
        import torch
        dtype = T["p"].dtype
        steps = int(actions.shape[0])
        want = {"obs": (steps + 1, B, P.ny, P.nx, 2), "actions": (steps, B, P.action_dim), "rewards": (steps, B), "terminated": (steps, B)}
        for name, x in (("obs", obs), ("actions", actions), ("rewards", rewards), ("terminated", terminated)):
            if tuple(x.shape) != want[name] or not x.is_contiguous():
                raise N.NativeError(f"rollout {name} must be a contiguous {list(want[name])} tensor, got {tuple(x.shape)}")
        bufs = self._bufs_ns({**T, "u": None, "v": None, "state_in": None, "p_out": None, "obs": obs[0], "action": actions[0],
                              "reward": rewards[0], "terminated": terminated[0]}, dtype)
        ro = N.RolloutNS2D()
        ro.T = steps
        ro.obs, ro.actions, ro.rewards = N.dptr(obs, dtype), N.dptr(actions, dtype), N.dptr(rewards, dtype)
        ro.terminated = N.dptr(terminated, torch.uint8)
        fn = getattr(self.lib, "pdegym_ns2d_rollout_" + self._sfx(dtype))
        N.check(fn(C.byref(P), C.byref(bufs), C.byref(ro), B, N.current_stream_ptr(T["p"].device)), "pdegym_ns2d_rollout")

    @_on_device_of("p")
    def ns2d_reset(self, P: N.ParamsNS2D, T: dict, u0, v0, p0, mask, B: int):
        import torch
        dtype = T["p"].dtype
        bufs = self._bufs_ns(T, dtype)
        fn = getattr(self.lib, "pdegym_ns2d_reset_masked_" + self._sfx(dtype))
        m = N.dptr(mask, torch.uint8) if mask is not None else None
        N.check(fn(C.byref(P), C.byref(bufs), N.dptr(u0, dtype), N.dptr(v0, dtype), N.dptr(p0, dtype), m, B,
                   N.current_stream_ptr(T["p"].device)), "pdegym_ns2d_reset_masked")

    @_on_device_of("p")
    def ns2d_solve_pressure(self, P: N.ParamsNS2D, u, v, p_in, p_out, scratch, B: int):
        dtype = u.dtype
        fn = getattr(self.lib, "pdegym_ns2d_solve_pressure_" + self._sfx(dtype))
        N.check(fn(C.byref(P), N.dptr(u, dtype), N.dptr(v, dtype), N.dptr(p_in, dtype), N.dptr(p_out, dtype),
                   N.dptr(scratch, dtype), B, N.current_stream_ptr(u.device)), "pdegym_ns2d_solve_pressure")


    # ---- Traffic ARZ ---------------------------------------------------------------------------
    @staticmethod
    def _bufs_traffic(T) -> N.BufsTraffic:
        import torch
        b = N.BufsTraffic()
        for k in ("r", "y", "action", "time", "rs", "qs_clip", "obs", "reward"):
            setattr(b, k, N.dptr(T[k], torch.float64))
        b.done = N.dptr(T["done"], torch.uint8)
        b.truncated = N.dptr(T["truncated"], torch.uint8)
        b.action_stride = int(T["action"].shape[1]) if T["action"].dim() == 2 else 1
        rr = T.get("reset_rs")
        if rr is not None:           # fused auto-reset (include/pdegym.h)
            b.reset_rs, b.reset_profile = N.dptr(rr, torch.float64), N.dptr(T["reset_profile"], torch.float64)
            b.reset_pool_rows = int(rr.shape[0])
            b.final_obs = N.dptr(T["final_obs"], torch.float64) if T.get("final_obs") is not None else None
            b.reset_count = N.dptr(T["reset_count"], torch.int32) if T.get("reset_count") is not None else None
        return b

    @_on_device_of("r")
    def traffic_step(self, P: N.ParamsTraffic, T: dict, B: int):
        bufs = self._bufs_traffic(T)
        N.check(self.lib.pdegym_traffic_step(C.byref(P), C.byref(bufs), B, N.current_stream_ptr(T["r"].device)),
                "pdegym_traffic_step")

    @_on_device_of("r")
    def traffic_rollout(self, P: N.ParamsTraffic, T: dict, obs, actions, rewards, done, truncated, B: int, policy=None):
        """T env-steps in one launch (pdegym_traffic_rollout): ``obs`` [T+1, B, 2M], ``actions`` [T, B, A] (A = 1 or 2; an output
        when ``policy`` -- an ``N.Mlp`` descriptor, its ``noise`` [T, B, A] -- is given), ``rewards`` / ``done`` / ``truncated`` [T, B]."""
        import torch
        steps = int(actions.shape[0])
        A = int(actions.shape[2])
        if tuple(obs.shape) != (steps + 1, B, 2 * P.M) or not obs.is_contiguous():
            raise N.NativeError(f"rollout obs must be a contiguous [{steps + 1}, {B}, {2 * P.M}] tensor, got {tuple(obs.shape)}")
        if tuple(actions.shape) != (steps, B, A) or A not in (1, 2) or not actions.is_contiguous():
            raise N.NativeError(f"rollout actions must be a contiguous [{steps}, {B}, 1 or 2] tensor, got {tuple(actions.shape)}")
        for name, x in (("rewards", rewards), ("done", done), ("truncated", truncated)):
            if tuple(x.shape) != (steps, B) or not x.is_contiguous():
                raise N.NativeError(f"rollout {name} must be a contiguous [{steps}, {B}] tensor, got {tuple(x.shape)}")
        bufs = self._bufs_traffic({**T, "action": actions[0]})
        ro = N.RolloutTraffic()
        ro.T = steps
        ro.obs, ro.actions, ro.rewards = N.dptr(obs, torch.float64), N.dptr(actions, torch.float64), N.dptr(rewards, torch.float64)
        ro.done, ro.truncated = N.dptr(done, torch.uint8), N.dptr(truncated, torch.uint8)
        ro.policy = C.addressof(policy) if policy is not None else None
        N.check(self.lib.pdegym_traffic_rollout(C.byref(P), C.byref(bufs), C.byref(ro), B, N.current_stream_ptr(obs.device)),
                "pdegym_traffic_rollout")

    @_on_device_of("r")
    def traffic_reset(self, P: N.ParamsTraffic, T: dict, profile, mask, B: int):
        import torch
        bufs = self._bufs_traffic(T)
        m = N.dptr(mask, torch.uint8) if mask is not None else None
        N.check(self.lib.pdegym_traffic_reset_masked(C.byref(P), C.byref(bufs), N.dptr(profile, torch.float64), m, B,
                                                     N.current_stream_ptr(T["r"].device)), "pdegym_traffic_reset_masked")

    # ---- Brain tumour --------------------------------------------------------------------------
    @staticmethod
    def _bufs_tumor(T) -> N.BufsTumor:
        import torch
        b = N.BufsTumor()
        for k in ("u", "xscale", "control", "remaining", "t_benchmark", "reward", "out"):
            setattr(b, k, N.dptr(T[k], torch.float64))
        b.kill = N.dptr(T.get("kill"), torch.float64) if T.get("kill") is not None else None
        for k in ("time_index", "stage", "days"):
            setattr(b, k, N.dptr(T[k], torch.int32))
        b.terminated = N.dptr(T["terminated"], torch.uint8)
        b.truncated = N.dptr(T["truncated"], torch.uint8)
        b.active = N.dptr(T.get("active"), torch.uint8) if T.get("active") is not None else None
        b.history = N.dptr(T.get("history"), torch.float64) if T.get("history") is not None else None
        b.t1_log = N.dptr(T.get("t1_log"), torch.float64) if T.get("t1_log") is not None else None
        return b

    @_on_device_of("u")
    def tumor_advance(self, P: N.ParamsTumor, T: dict, mode: int, max_days: int, B: int):
        bufs = self._bufs_tumor(T)
        N.check(self.lib.pdegym_tumor_advance(C.byref(P), C.byref(bufs), mode, max_days, B, N.current_stream_ptr(T["u"].device)),
                "pdegym_tumor_advance")

    @_on_device_of("u")
    def tumor_step(self, P: N.ParamsTumor, T: dict, B: int):
        bufs = self._bufs_tumor(T)
        N.check(self.lib.pdegym_tumor_step(C.byref(P), C.byref(bufs), B, N.current_stream_ptr(T["u"].device)),
                "pdegym_tumor_step")

    @_on_device_of("u")
    def tumor_reset(self, P: N.ParamsTumor, T: dict, init, mask, B: int):
        import torch
        bufs = self._bufs_tumor(T)
        m = N.dptr(mask, torch.uint8) if mask is not None else None
        stride = 0 if init.dim() == 1 else init.shape[-1]
        N.check(self.lib.pdegym_tumor_reset_masked(C.byref(P), C.byref(bufs), N.dptr(init, torch.float64), stride, m, B,
                                                   N.current_stream_ptr(T["u"].device)), "pdegym_tumor_reset_masked")

    # ---- policy network ----------------------------------------------------------------------------
    @_on_device_of("x")
    def mlp_forward(self, net: N.Mlp, x, y, B: int):
        """y[b] = net(x[b]) for b < B: x [B, in_dim], y [B, out_dim] rows (row stride = stride(0)); float32, or float64 with
        net.x_f64 / net.y_f64 set (the caller sets them from the tensors' dtypes)."""
        import torch
        for t_, name, f64 in ((x, "x", net.x_f64), (y, "y", net.y_f64)):
            want = torch.float64 if f64 else torch.float32
            if not t_.is_cuda or t_.dtype != want or t_.dim() != 2 or t_.stride(1) != 1:
                raise N.NativeError(f"mlp_forward: {name} must be a {want} HIP tensor [B, width] with unit inner stride")
        N.check(self.lib.pdegym_mlp_forward(C.byref(net), x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), B,
                                            N.current_stream_ptr(x.device)), "pdegym_mlp_forward")


_default = None


def default_backend():
    """The process-wide HIP backend (loads libpdegym_hip.so; raises if it is missing)."""
    global _default
    if _default is None:
        _default = HipBackend()
    return _default
