"""On-device rollout loop: policy inference and environment stepping never leave the GPU.

The reference trains with SB3, which steps ONE environment per Python call (``DummyVecEnv(n=1)``,
examples/transportPDE/transport1Dppo.py:77-90).  With thousands of instances per launch the per-step host work
(Python, launch latency) becomes the bottleneck at small sub-step counts, so ``DeviceRollout`` records the whole
T-step rollout -- policy forward pass (any torch module), action clamp, environment step with fused auto-reset,
writes into the [T, B, ...] rollout buffers -- into ONE hipGraph and replays it: zero host work per step.
"""
from __future__ import annotations


class DeviceRollout:
    """``policy``: callable mapping an observation tensor [B, obs_dim] to actions [B] (or [B, 1]) on the same device -- any
    torch module, or a ``pdecontrolgym_amd.FusedMLP`` (Linear/Tanh/ReLU stack evaluated, clamped and stored in ONE launch).
    Buffers: ``obs[T+1, B, D]``, ``actions[T, B]`` (``[T, B, action_dim]`` for Navier-Stokes and two-command traffic), ``rewards[T, B]``, ``terminated[T, B]``, ``truncated[T, B]``; with
    ``action_noise=True`` also ``action_noise[T, B]`` (float32), added to the policy output of step t before the clamp.
    ``one_launch``: see the constructor (1D engines with full-state sensing + a small ``FusedMLP``: the rollout is ONE kernel)."""

    def __init__(self, venv, policy, n_steps: int, use_graph: bool = True, action_low: float = -1.0, action_high: float = 1.0,
                 action_noise: bool = False, one_launch=None, sensing_noise: bool = False):
        import torch
        kind = getattr(venv, "kind", "tumor")
        self.venv, self.policy, self.T = venv, policy, int(n_steps)
        self.lo, self.hi = float(action_low), float(action_high)
        core = venv.core
        # the transport / reaction-diffusion and Navier-Stokes engines write straight into the rollout buffers; the other
        # engines (traffic, brain tumour: several launches and device-side masks per step) go through step_tensor and one
        # copy per output
        self._ns = kind == "ns2d"
        self._direct = hasattr(core, "obs_dim") or self._ns
        cur = core.t["obs"] if "obs" in core.t else core.t["u"]      # the tumour engine's observation IS its live row
        B, dev, dt = core.num_envs, core.device, cur.dtype
        oshape = tuple(cur.shape[1:])                                # (D,) for the 1D engines, (ny, nx, 2) for Navier-Stokes
        self.obs = torch.zeros((self.T + 1, B) + oshape, dtype=dt, device=dev)
        adim = int(getattr(core, "action_dim", 1))      # Navier-Stokes: its action_dim; traffic 'both': inlet and outlet command
        self.actions = torch.zeros((self.T, B) + ((adim,) if (self._ns or adim > 1) else ()), dtype=dt, device=dev)
        self.rewards = torch.zeros(self.T, B, dtype=dt, device=dev)
        self.terminated = torch.zeros(self.T, B, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(self.T, B, dtype=torch.uint8, device=dev)
        # exploration noise of a stochastic policy: slot t is added to the policy output of step t before the clamp.  The
        # caller fills the buffer in place before each run() (e.g. ``ro.action_noise.normal_().mul_(std)``): a replayed graph
        # reads the new draws.
        self.action_noise = torch.zeros_like(self.actions, dtype=torch.float32) if action_noise else None
        # transport / reaction-diffusion (any control / sensing combination), or traffic, and a FusedMLP of <= 256-unit layers: the WHOLE rollout is one
        # kernel launch (pdegym_*_rollout with the policy inside: no kernel boundary between env-steps, none between policy and step).
        # one_launch=None: whenever it applies; True: required; False: T x (policy launch + step launch) as for the others.
        self._traffic = kind == "traffic"
        fits = bool(hasattr(core, "policy_fits_rollout") and core.policy_fits_rollout(policy)
                    and dt == (torch.float64 if self._traffic else torch.float32))
        if one_launch and not fits:
            raise ValueError("one_launch=True needs a transport / reaction-diffusion engine whose state has one home (any control / "
                             "sensing combination, no history, float32 operands) or a traffic engine of <= 64 nodes, and a FusedMLP "
                             "that fits the rollout kernel (layers of <= 256 units, see policy_fits_rollout)")
        self.one_launch = fits if one_launch is None else bool(one_launch)
        # device-side sensing noise (PDEVecEnv(sensing_noise_tensor_func=...)): the policy reads obs_seen[t] = f(obs[t]) while
        # obs[t] -- the plant state with full-state sensing -- stays clean; the call is part of the captured graph (torch's
        # random generators are graph-safe: every replay draws new numbers).  The one-launch kernel has no such hook.
        self._noise_f = getattr(venv, "sensing_noise_tensor_func", None)
        # sensing_noise=True: the same hook as ADDITIVE noise the caller draws ahead (fill ``ro.sensing_noise`` [T + 1, B, ...] in
        # place before each run(), e.g. ``ro.sensing_noise.normal_().mul_(sigma)``): obs_seen[t] = obs[t] + sensing_noise[t].
        # This form also runs inside the one-launch rollout kernels (pdegym_rollout1d.obs_noise / obs_seen).
        self.sensing_noise = None
        if sensing_noise:
            if self._noise_f is not None:
                raise ValueError("sensing_noise=True and sensing_noise_tensor_func are two forms of the same hook: use one")
            self.sensing_noise = torch.zeros_like(self.obs)
            self._noise_f = None
        self.obs_seen = torch.zeros_like(self.obs) if (self._noise_f is not None or sensing_noise) else None
        if self._noise_f is not None:
            if one_launch:
                raise ValueError("one_launch=True cannot apply sensing_noise_tensor_func (the policy runs inside the step kernel); "
                                 "sensing_noise=True (pre-drawn additive noise) can")
            self.one_launch = False
        if self.sensing_noise is not None and self.one_launch and self._traffic:
            self.one_launch = False          # (the traffic rollout kernel has no obs_noise input)
        self.use_graph = bool(use_graph) and dev.type == "cuda"
        self._graph = None
        self._state_buf = None     # observation tensor the captured graph leaves the end state in (see _rebind_state)

    def _body(self):
        import torch
        core = self.venv.core
        if self.one_launch:
            acts, nz = self.actions, self.action_noise
            if self._traffic and acts.dim() == 2:       # the traffic kernel takes [T, B, action_dim]
                acts, nz = acts.unsqueeze(2), (nz.unsqueeze(2) if nz is not None else None)
            extra = {}
            if self.sensing_noise is not None:
                extra = dict(obs_noise=self.sensing_noise[:self.T], obs_seen=self.obs_seen[:self.T])
            core.rollout(self.obs, acts, self.rewards, self.terminated, self.truncated, policy=self.policy,
                         clamp=(self.lo, self.hi), noise=nz, **extra)
            if self.sensing_noise is not None:
                torch.add(self.obs[self.T], self.sensing_noise[self.T], out=self.obs_seen[self.T])
            return
        own = {k: core.t[k] for k in ("reward", "terminated", "truncated", "obs") if k in core.t} if self._direct else {}
        pingpong = self._ns and getattr(core, "_p_pingpong", False)
        if pingpong:        # keep the pressure in ONE tensor while the steps are baked into a graph (the C side copies it home)
            core._p_pingpong, saved_p_out = False, core.t["p_out"]
            core.t["p_out"] = None
        try:
            self._steps(core, torch)
            if self._direct:
                # The engine's own observation buffer must end up holding slot T for EVERY engine that wrote its observations
                # straight into the rollout buffers: the next run() (and any reader of the engine's current observation) starts
                # from it.  Where the observation IS the state (Navier-Stokes; the 1D engines with full-state sensing) this copy
                # is also the state hand-over; engines with a separate state (state_in_obs=False, scalar sensing, history
                # recording) advanced ``u`` in place and only their observation would otherwise be stale (advisor finding r3).
                own["obs"].copy_(self.obs[self.T])
        finally:
            if pingpong:
                core._p_pingpong, core.t["p_out"] = True, saved_p_out
            # the step kernel was pointed at slot t of the rollout buffers; hand the engine its own output tensors back so
            # that a later plain env.step() cannot overwrite rewards[T-1] / terminated[T-1] / truncated[T-1]
            # (graph-safe: only Python references change)
            for k, v in own.items():
                core.t[k] = v
            if getattr(core, "state_in_obs", False):
                core.t["u"] = core.t["obs"]

    def _rebind_state(self, core):
        """The captured graph has the ADDRESS of the engine's observation tensor baked in (it copies slot T there, and where the
        observation is the state that copy is the state hand-over).  The engines double-buffer their observations, so a plain
        step() / reset between two run() calls leaves ``core.t["obs"]`` naming the other buffer: after a replay the engine is
        pointed back at the buffer the graph wrote, otherwise the next step would restart from the pre-rollout state."""
        buf = self._state_buf
        if buf is None or core.t.get("obs") is buf:
            return
        for i, o in enumerate(getattr(core, "_obs", ())):
            if o is buf:
                core._flip = i
        core.t["obs"] = buf
        if getattr(core, "state_in_obs", False):
            core.t["u"] = buf

    def _steps(self, core, torch):
        if self._ns or getattr(core, "state_in_obs", False):
            # slot 0 of the rollout buffer is the input state of the first step (not the engine's own buffer, which a graph
            # warm-up run leaves in its end state)
            core.t["obs"] = self.obs[0]
        fused = hasattr(self.policy, "forward_into") and self.obs.dtype in (torch.float32, torch.float64)
        seen = self.obs if self.obs_seen is None else self.obs_seen
        for t in range(self.T):
            nz = self.action_noise[t] if self.action_noise is not None else None
            if self.obs_seen is not None:
                with torch.no_grad():
                    self._see(t, torch)
            if fused:       # pdecontrolgym_amd.FusedMLP: forward pass (+ noise) + action clamp in one launch, written into slot t
                self.policy.forward_into(seen[t], self.actions[t], clamp=(self.lo, self.hi), noise=nz)
            else:
                with torch.no_grad():
                    a = self.policy(seen[t]).reshape(self.actions[t].shape)
                    if nz is not None:
                        a = a + nz.to(a.dtype)
                    a = a.clamp(self.lo, self.hi)
                self.actions[t].copy_(a)
            if self._ns:
                # Navier-Stokes: the observation IS the state, so slot t of the rollout buffer is also the next step's input
                core.step(self.actions[t], out_obs=self.obs[t + 1], out_reward=self.rewards[t], out_terminated=self.terminated[t])
            elif self._direct:
                # the step kernel writes observation / reward / flags straight into slot t of the rollout buffers
                core.step(self.actions[t], out_obs=self.obs[t + 1], out_reward=self.rewards[t],
                          out_terminated=self.terminated[t], out_truncated=self.truncated[t])
            else:
                o, r, te, tr = self.venv.step_tensor(self.actions[t])
                self.obs[t + 1].copy_(o)
                self.rewards[t].copy_(r)
                self.terminated[t].copy_(te)
                self.truncated[t].copy_(tr)
        if self.obs_seen is not None:
            with torch.no_grad():
                self._see(self.T, torch)

    def _see(self, t, torch):
        """obs_seen[t]: what the policy reads of observation t (sensing_noise_tensor_func, or obs + the pre-drawn noise)."""
        if self.sensing_noise is not None:
            torch.add(self.obs[t], self.sensing_noise[t], out=self.obs_seen[t])
        else:
            self.obs_seen[t].copy_(self._noise_f(self.obs[t]))

    def run(self, first_obs=None):
        """Roll T steps from ``first_obs`` (default: the environment's current observation). Returns self."""
        import torch
        core = self.venv.core
        if hasattr(self.policy, "refresh"):      # FusedMLP: pick up optimizer updates before the graph is (re)played
            self.policy.refresh()
        self.obs[0].copy_((core.t["obs"] if self._direct or "obs" in core.t else core.t["u"]) if first_obs is None else first_obs)
        if not self.use_graph:
            self._body()
            return self
        if self._graph is None:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            # everything a step mutates in place (the fused auto-reset advances reset_count and may redraw beta rows)
            if self._ns:
                keys = [k for k in ("u", "v", "p", "p_out", "time_index", "reset_count") if torch.is_tensor(core.t.get(k))]
            elif self._direct:
                keys = [k for k in ("u", "time_index", "bsum", "ring", "reset_count", "beta") if torch.is_tensor(core.t.get(k))]
            else:
                keys = [k for k, v in core.t.items() if torch.is_tensor(v)]
            snapshot = {k: core.t[k].clone() for k in keys}
            extra = {k: getattr(self.venv, k).clone() for k in ("_consecutive", "treatment_calls", "soft_constraint_violations")
                     if torch.is_tensor(getattr(self.venv, k, None))}
            self._state_buf = core.t["obs"] if "obs" in core.t else None
            with torch.cuda.stream(side):
                self._body()                               # warm-up on the side stream (allocator, lazy init)
                for k, v in snapshot.items():
                    if core.t[k].shape == v.shape:         # (an input slot such as the traffic engine's "action" may have been rebound)
                        core.t[k].copy_(v)                 # ... then rewind the environment state
                for k, v in extra.items():
                    getattr(self.venv, k).copy_(v)
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, stream=side):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            for k, v in snapshot.items():
                if core.t[k].shape == v.shape:
                    core.t[k].copy_(v)
            for k, v in extra.items():
                getattr(self.venv, k).copy_(v)
        self._graph.replay()
        self._rebind_state(core)
        return self
