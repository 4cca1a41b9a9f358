"""HAPPO trainer (sequential per-agent updates weighted by the other agents' probability ratios).

Surface of the reference's onpolicy/algorithms/happo/happo_trainer.py (HAPPO :9, cal_value_loss :51,
ppo_update :89, train :170).  It is the MAPPO update with four differences, each restated here as the
reference has it:
  * minibatches are 13-tuples; the clipped surrogate is multiplied by ``factor_batch`` (:137-141);
  * the importance weight is the product over the action dimensions (:131);
  * the value normaliser is never fed by the trainer: under ValueNorm the statistics stay at their
    initial values (normalize() does not update, :60-63); under ``--use_popart`` it is the stand-alone
    popart_hatrpo.PopArt whose normalize() updates on every call -- twice per value loss (:62-63);
  * advantages subtract de-normalised value predictions only under ``--use_popart`` (:180-183).
"""
import torch

from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
from onpolicy.algorithms.utils.popart_hatrpo import PopArt
from onpolicy.utils.valuenorm import ValueNorm


class HAPPO(R_MAPPO):
    _use_factor = True
    _updates_normalizer = False

    def _make_value_normalizer(self):
        if self._use_popart:
            return PopArt(1, device=self.device)
        if self._use_valuenorm:
            return ValueNorm(1, device=self.device)
        return None

    def _denormalize_advantages(self):
        return bool(self._use_popart)

    def _fused_loss_allowed(self):
        return not self._use_popart       # the self-updating normaliser has side effects per normalize() call

    def _value_targets(self, return_batch, update_normalizer):
        if self._use_popart or self._use_valuenorm:
            if self._use_popart and self.dp.active:
                raise NotImplementedError("HAPPO's self-updating PopArt has no data-parallel form")
            # two calls on purpose: popart_hatrpo.PopArt moves its statistics on each of them
            return self.value_normalizer.normalize(return_batch), self.value_normalizer.normalize(return_batch)
        return return_batch, return_batch

    def _ratio(self, action_log_probs, old_action_log_probs):
        return torch.prod(torch.exp(action_log_probs - old_action_log_probs), dim=-1, keepdim=True)
