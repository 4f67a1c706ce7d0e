"""Checkpoint / resume of a batched environment (``state_dict()`` / ``load_state_dict()``).

The reference's training scripts checkpoint every N steps (examples/transportPDE/transport1Dppo.py:80-86 saves the model; its
single environment is rebuilt from its parameter dictionary).  A batch of thousands of instances in mid-episode is not
rebuildable that way, so every engine can hand out and take back its device state:

    sd = venv.state_dict()            # torch tensors (clones) + a few Python scalars: ``torch.save(sd, path)`` works
    ...
    venv.load_state_dict(sd)          # same construction parameters required; copies IN PLACE (captured hipGraphs stay valid)

What is state: everything in the engine's tensor dictionary that a step reads and that is not a constructor constant or a
per-call input -- live rows / fields, plant parameters, time indices, the reward's running sums, the fused auto-reset pools and
their restart counters, the last outputs.  What is not: user callbacks and their random generators (Python objects of the
caller), the contents of ``DeviceRollout`` buffers.
"""
from __future__ import annotations

# 1: rounds 3-4 (meta without the reward / sensing keys of the 1D engine, no "format" entry); 2: round 5 on
CHECKPOINT_FORMAT = 2

# per-call inputs, scratch and constructor constants: not part of a checkpoint
_SKIP = {"action", "state_in", "scratch", "U_ref", "action_ref", "xscale", "active", "reset_profile", "p_out", "control", "kill"}


class EngineCheckpoint:
    """Mixin of the batched engines (``self.t``: name -> tensor | None, ``self.num_envs``)."""

    def _checkpoint_meta(self):
        return {"engine": type(self).__name__, "num_envs": int(self.num_envs)}

    def state_dict(self):
        import torch
        sd = {"format": CHECKPOINT_FORMAT, "meta": self._checkpoint_meta(), "tensors": {}}
        for k, v in self.t.items():
            if torch.is_tensor(v) and k not in _SKIP:
                sd["tensors"][k] = v.detach().clone()
        return sd

    def load_state_dict(self, sd):
        import torch
        meta = self._checkpoint_meta()
        fmt = int(sd.get("format", 1))
        if fmt > CHECKPOINT_FORMAT:
            raise ValueError(f"checkpoint format {fmt} is newer than this library's ({CHECKPOINT_FORMAT}): upgrade the library")
        saved = sd.get("meta") or {}
        # every key the checkpoint recorded must agree; keys this library has added since (format 1 checkpoints of the 1D engine
        # know nothing of the reward / sensing configuration) are absent from an older checkpoint and are not held against it
        wrong = {k: (saved[k], meta.get(k)) for k in saved if saved[k] != meta.get(k)}
        if wrong or saved.get("engine") != meta.get("engine"):
            raise ValueError(f"checkpoint was written by {saved}, this engine is {meta} (mismatch: {wrong})")
        for k, v in sd["tensors"].items():
            cur = self.t.get(k)
            if torch.is_tensor(cur) and cur.shape == v.shape and cur.dtype == v.dtype:
                cur.copy_(v)                               # in place: addresses baked into captured graphs stay valid
            else:                                          # a pool / optional tensor this engine has not allocated (yet)
                self.t[k] = v.to(self.device).clone()
        self._after_load(sd)

    def _after_load(self, sd):
        pass
