"""BrainTumorReward (reference rewards/brain_tumor_reward.py:5-73).

Two rewards share one entry point: at the end of an episode the survival gain ``time_index - t_benchmark`` over the
untreated baseline, and during therapy a toxicity penalty ``-50 * clip((AD - dmaxsafe(TR)) / (TD - dmaxsafe(TR)), 0, 1)**(1/3)``
with ``dmaxsafe(TR) = 116 * TR**-0.685`` (TR treatment radius, AD applied dose, TD total dose).  Until ``t_benchmark`` is
set every reward is 0.  ``BrainTumor1D.step`` evaluates this definition on the scalars the step kernel returns; the
batched engine evaluates the same expression inside the kernel (pdecontrolgym_amd/csrc/pdegym_tumor.hip).
"""
from typing import Optional

import numpy as np

from pde_control_gym.src.rewards.base_reward import BaseReward


class BrainTumorReward(BaseReward):
    """Dispatches on what the caller knows: no baseline yet -> 0; episode over -> survival gain; otherwise (a treatment
    day) -> toxicity penalty of the dose just applied.  Same values as the reference's single ``reward`` body."""

    LAMBDA_TOXIC = 50            # weight of the toxicity penalty
    SAFE_DOSE_COEFF, SAFE_DOSE_EXPONENT = 116, -0.685

    @classmethod
    def dmaxsafe(cls, treatment_radius):
        """Largest dose (Gy) considered safe for a treated region of this radius (mm)."""
        return cls.SAFE_DOSE_COEFF * (treatment_radius ** cls.SAFE_DOSE_EXPONENT)

    @staticmethod
    def survival_gain(time_index, t_benchmark):
        return time_index - t_benchmark

    @classmethod
    def toxicity(cls, treatment_radius, applied_dosage, total_dosage):
        """Fraction in [0, 1] of the way from the safe dose to the whole prescription, cube-rooted."""
        safe = cls.dmaxsafe(treatment_radius)
        excess = (applied_dosage - safe) / (total_dosage - safe)
        return min(max(excess, 0.0), 1.0) ** (1 / 3)

    def reward(self, uVec: np.ndarray = None, time_index: int = None, terminate: Optional[bool] = None,
               truncate: Optional[bool] = None, action: Optional[float] = None, verbose=True, **kwargs):
        """Keywords: ``t_benchmark`` always; ``treatment_radius``, ``applied_dosage``, ``total_dosage`` on treatment days."""
        t_benchmark = kwargs["t_benchmark"]
        if t_benchmark is None:
            value, why = 0, "no t_benchmark yet (run TherapyWrapper.benchmark() first)"
        elif terminate or truncate:
            value, why = self.survival_gain(time_index, t_benchmark), f"survival gain {time_index} - {t_benchmark}"
        else:
            tox = self.toxicity(kwargs["treatment_radius"], kwargs["applied_dosage"], kwargs["total_dosage"])
            value = - self.LAMBDA_TOXIC * tox
            why = (f"toxicity {tox:.4f} (radius {kwargs['treatment_radius']}, dose {kwargs['applied_dosage']}, "
                   f"safe {self.dmaxsafe(kwargs['treatment_radius'])})")
        if verbose:
            print(f"[brain tumour reward] {value}: {why}")
        return value
