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
    LAMBDA_TOXIC = 50

    @staticmethod
    def dmaxsafe(treatment_radius):
        return 116 * (treatment_radius ** -0.685)

    def reward(self, uVec: np.ndarray = None, time_index: int = None, terminate: Optional[bool] = None,
               truncate: Optional[bool] = None, action: Optional[float] = None, verbose=True, **kwargs):
        t_benchmark = kwargs["t_benchmark"]
        if t_benchmark is None:
            if verbose:
                print("Warning: t_benchmark is not yet set -> returned reward of 0\n")
            return 0
        if terminate or truncate:
            if verbose:
                print(f"Reward Class: time_index - t_benchmark = {time_index} - {t_benchmark}")
            return time_index - t_benchmark
        treatment_radius, applied_dosage = kwargs["treatment_radius"], kwargs["applied_dosage"]
        total_dosage = kwargs["total_dosage"]
        maxsafe = self.dmaxsafe(treatment_radius)
        ratio = (applied_dosage - maxsafe) / (total_dosage - maxsafe)
        r_toxic = (min(max(ratio, 0.0), 1.0)) ** (1 / 3)
        if verbose:
            print(f"Reward Class: - l_t*r_toxic = {- self.LAMBDA_TOXIC * r_toxic}")
            print(f"\tParams: treatment_radius={treatment_radius} applied_dosage={applied_dosage} "
                  f"dmaxsafe(treatment_radius)={maxsafe}")
        return - self.LAMBDA_TOXIC * r_toxic
