"""Fused forward pass of a small MLP policy on the device (C ABI: pdegym_mlp_forward, csrc/pdegym_mlp.hip).

The reference's RL controllers are SB3 ``MlpPolicy`` networks -- Linear/Tanh stacks, two hidden layers of 64 units by
default (examples/transportPDE/transport1Dppo.py:88-90) -- evaluated once per ``env.step``.  ``FusedMLP`` wraps such a
``torch.nn.Sequential`` (or any iterable of ``Linear`` / ``Tanh`` / ``ReLU`` / ``Identity`` / ``Flatten`` modules) and
evaluates it in ONE kernel launch.  The kernel reads the weights in a blocked transpose ``[ceil(in/4), out, 4]`` (one
16-byte load per lane = four inputs of its neuron, contiguous per wave), so the wrapper keeps that copy next to the module's
parameters: ``refresh()`` re-transposes the layers
whose parameters changed (checked through the tensors' version counters; called automatically by ``__call__`` /
``forward_into`` outside graph capture and by ``DeviceRollout.run`` before it replays its hipGraph).  The copies keep their
addresses, so a captured graph sees refreshed weights.
"""
from __future__ import annotations

from . import _native as N


class FusedMLP:
    """``policy = FusedMLP(torch_module)``; ``policy(obs)`` -> ``[B, out_dim]`` float32 actions on the same device.

    ``clamp=(lo, hi)`` fuses the action-box clamp into the launch.  ``forward_into(obs, out)`` writes into a caller tensor
    (``pde_control_gym.DeviceRollout`` points it at slot t of its action buffer).  Inference only (no autograd graph).
    """

    def __init__(self, module, clamp=None, backend=None):
        import torch
        layers = []            # [weight, bias, act]
        mods = list(module) if not isinstance(module, torch.nn.Linear) else [module]
        for m in mods:
            if isinstance(m, torch.nn.Linear):
                layers.append([m.weight, m.bias, N.MLP_IDENTITY])
            elif isinstance(m, (torch.nn.Tanh, torch.nn.ReLU)):
                if not layers or layers[-1][2] != N.MLP_IDENTITY:
                    raise ValueError("an activation must follow a Linear layer (one activation per layer)")
                layers[-1][2] = N.MLP_TANH if isinstance(m, torch.nn.Tanh) else N.MLP_RELU
            elif isinstance(m, (torch.nn.Identity, torch.nn.Flatten)):
                continue
            else:
                raise ValueError(f"FusedMLP supports Linear / Tanh / ReLU / Identity / Flatten stacks, got {type(m).__name__}")
        if not 1 <= len(layers) <= N.MLP_MAX_LAYERS:
            raise ValueError(f"FusedMLP takes 1..{N.MLP_MAX_LAYERS} Linear layers, got {len(layers)}")
        for i, (w, b, _) in enumerate(layers):
            if w.dtype != torch.float32 or not w.is_contiguous() or (b is not None and (b.dtype != torch.float32 or not b.is_contiguous())):
                raise ValueError("FusedMLP needs contiguous float32 parameters")
            if w.shape[0] > N.MLP_MAX_WIDTH:
                raise ValueError(f"layer {i} is wider than {N.MLP_MAX_WIDTH} units")
            if i and w.shape[1] != layers[i - 1][0].shape[0]:
                raise ValueError(f"layer {i} takes {w.shape[1]} inputs but layer {i - 1} produces {layers[i - 1][0].shape[0]}")
        if layers[0][0].shape[1] > N.MLP_MAX_INPUT:
            raise ValueError(f"observation rows wider than {N.MLP_MAX_INPUT} are not supported")
        if clamp is not None and not float(clamp[0]) <= float(clamp[1]):
            raise ValueError("clamp must be (lo, hi) with lo <= hi")
        self.module, self.layers = module, layers
        # blocked transpose [ceil(in/4), out, 4] (include/pdegym.h); the buffers keep their addresses for captured graphs
        self._wt = [torch.zeros((w.shape[1] + 3) // 4, w.shape[0], 4, dtype=torch.float32, device=w.device) for w, _, _ in layers]
        self._seen = [None] * len(layers)
        self.refresh(force=True)
        self.in_dim, self.out_dim = int(layers[0][0].shape[1]), int(layers[-1][0].shape[0])
        self.clamp = None if clamp is None else (float(clamp[0]), float(clamp[1]))
        self.device = layers[0][0].device
        if backend is None:
            from .backend import default_backend
            backend = default_backend()
        self.backend = backend

    @classmethod
    def from_sb3(cls, policy, clamp=None, backend=None):
        """The deterministic actor of a Stable-Baselines3 policy as a ``FusedMLP`` (duck-typed, SB3 itself is not imported):
        ``ActorCriticPolicy`` (PPO / A2C ``MlpPolicy``): ``mlp_extractor.policy_net`` followed by ``action_net`` -- the mean of
        the Gaussian, which SB3 clips to the action box (pass ``clamp``); ``SACPolicy``: ``actor.latent_pi`` + ``actor.mu`` with
        the tanh squashing of ``actor.forward(deterministic=True)``; ``TD3Policy``: ``actor.mu`` alone (a Sequential that already
        ends in Tanh -- its actor has no ``latent_pi``).  The wrapper shares the parameters
        with ``policy`` (no copy), so ``model.learn()`` steps are picked up by ``refresh()``.  Observations must already be
        flat vectors (``FlattenExtractor``), which is what ``MlpPolicy`` uses on these environments."""
        import torch
        if hasattr(policy, "mlp_extractor") and hasattr(policy, "action_net"):
            mods = list(policy.mlp_extractor.policy_net) + [policy.action_net]
        elif hasattr(policy, "actor") and hasattr(policy.actor, "mu"):
            head = policy.actor.mu
            lat = getattr(policy.actor, "latent_pi", None)
            mods = (list(lat) if lat is not None else []) + (list(head) if isinstance(head, torch.nn.Sequential) else [head])
            if not isinstance(mods[-1], torch.nn.Tanh):
                mods.append(torch.nn.Tanh())          # SAC squashes the mean; TD3's mu already ends in Tanh
        else:
            raise ValueError("expected an SB3 ActorCriticPolicy (mlp_extractor + action_net) or SAC/TD3 policy (actor.mu, with actor.latent_pi in front for SAC)")
        return cls(torch.nn.Sequential(*mods), clamp=clamp, backend=backend)

    def refresh(self, force: bool = False):
        """Bring the transposed weight copies up to date with the module's parameters (in-place updates such as optimizer
        steps bump a tensor's version counter).  Not callable while a hipGraph is being captured."""
        import torch
        with torch.no_grad():
            for i, (w, _, _) in enumerate(self.layers):
                if force or w._version != self._seen[i]:
                    out_dim, in_dim = w.shape
                    k4 = (in_dim + 3) // 4
                    padded = torch.nn.functional.pad(w.detach(), (0, 4 * k4 - in_dim))      # [out, 4 k4], zeros past in_dim
                    self._wt[i].permute(1, 0, 2).copy_(padded.view(out_dim, k4, 4))        # write through the [out, k4, 4] view
                    self._seen[i] = w._version
        return self

    def _net(self, clamp, x_f64=False, y_f64=False) -> N.Mlp:
        net = N.Mlp()
        net.x_f64, net.y_f64 = int(x_f64), int(y_f64)
        net.n_layers = len(self.layers)
        net.clamp = 0 if clamp is None else 1
        net.lo, net.hi = (0.0, 0.0) if clamp is None else clamp
        for i, (w, b, act) in enumerate(self.layers):
            L = net.layer[i]
            L.w, L.b = self._wt[i].data_ptr(), (b.data_ptr() if b is not None else None)
            L.in_dim, L.out_dim, L.act = int(w.shape[1]), int(w.shape[0]), act
        return net

    def forward_into(self, obs, out, clamp="default", noise=None):
        """out[b, :] = net(obs[b, :]); ``obs`` [B, in_dim] (any trailing shape that flattens to in_dim), ``out`` [B, out_dim]
        or [B] when out_dim == 1, on the parameters' device.  float64 observations (traffic, tumour, float64 Navier-Stokes)
        are rounded to float32 as they are read and a float64 ``out`` receives the widened float32 result -- the casts SB3
        makes around its float32 policy.  ``noise`` (float32, same shape as the action): added to the network output before
        the clamp -- the caller's pre-scaled exploration noise of a Gaussian policy.  Returns ``out``."""
        B = obs.shape[0]
        x = obs.reshape(B, -1)
        if x.shape[1] != self.in_dim:
            raise ValueError(f"observation rows have {x.shape[1]} entries, the network takes {self.in_dim}")
        y = out.reshape(B, self.out_dim)
        if y.data_ptr() != out.data_ptr():
            raise ValueError("out must be viewable as [B, out_dim] without a copy")
        import torch
        if not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.refresh()
        for t_, name in ((x, "obs"), (y, "out")):
            if t_.dtype not in (torch.float32, torch.float64):
                raise N.NativeError(f"FusedMLP: {name} must be float32 or float64, got {t_.dtype}")
        net = self._net(self.clamp if clamp == "default" else clamp, x.dtype == torch.float64, y.dtype == torch.float64)
        if noise is not None:       # exploration noise, added before the clamp (float32 [B, out_dim] or [B] when out_dim == 1)
            nz = noise.reshape(B, self.out_dim)
            if nz.dtype != torch.float32 or nz.device != x.device or nz.stride(1) != 1 or nz.data_ptr() != noise.data_ptr():
                raise ValueError("noise must be a float32 tensor on the observations' device, viewable as [B, out_dim]")
            net.noise, net.noise_stride = nz.data_ptr(), nz.stride(0)
        self.backend.mlp_forward(net, x, y, B)
        return out

    def __call__(self, obs):
        import torch
        out = torch.empty(obs.shape[0], self.out_dim, dtype=obs.dtype, device=obs.device)
        return self.forward_into(obs, out)
