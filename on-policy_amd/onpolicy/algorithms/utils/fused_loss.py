"""Fused PPO loss (K7, ``mappo_ppo_loss_f32``): value and gradient of the clipped-surrogate loss for a
Discrete action head in one pass over a minibatch span, instead of ~100 elementwise / reduction launches
(reference r_mappo.py:52-89, :119-153).  The trainer feeds the gradients it returns straight into
``torch.autograd.backward`` at the head's logits and the critic's values.

Used when the policy has a single Categorical head and the tensors are on a HIP device; everything else
(MultiDiscrete / continuous / mixed heads, CPU) takes the framework path in ``R_MAPPO.ppo_update``.
``MAPPO_FUSED_LOSS=0`` disables it.
"""
import os

import torch

from onpolicy import _native


def enabled():
    return os.environ.get("MAPPO_FUSED_LOSS", "1") != "0"


def supported(policy, device):
    act = getattr(getattr(policy, "actor", None), "act", None)
    return (enabled() and torch.device(device).type == "cuda" and act is not None
            and getattr(act, "action_type", None) == "Discrete" and hasattr(policy, "evaluate_logits"))


def _flat(x):
    return None if x is None else x.detach().reshape(-1).contiguous()


def ppo_loss(logits, available_actions, actions, old_logp, adv, active, factor, values, value_preds, returns,
             norm, inv_denoms, sums, *, clip, huber_delta, entropy_coef, value_loss_coef, use_huber,
             use_clipped_value_loss, policy_active_masks, value_active_masks, need_actor=True):
    """-> (dlogits or None, dvalues).  ``sums`` (float64[4], device) accumulates
    {sum w_p * (-surrogate), sum w_p * entropy, sum w_v * value_loss, sum ratio}."""
    rows, na = logits.shape
    lg = logits.detach().contiguous()
    dlogits = torch.empty_like(lg) if need_actor else None
    v = _flat(values)
    dvalues = torch.empty_like(v)
    avail = None if available_actions is None else available_actions.detach().contiguous()
    keep = [lg, avail, _flat(actions), _flat(old_logp), _flat(adv), _flat(active), _flat(factor), v,
            _flat(value_preds), _flat(returns), norm, inv_denoms]
    p = _native.ptr
    a = _native.PPOLoss(*[p(t) for t in keep], p(dlogits), p(dvalues), p(sums), rows, na, float(clip),
                        float(huber_delta), float(entropy_coef), float(value_loss_coef),
                        (_native.LOSS_HUBER if use_huber else 0)
                        | (_native.LOSS_CLIPPED_VALUE if use_clipped_value_loss else 0)
                        | (_native.LOSS_POLICY_ACTIVE_MASKS if policy_active_masks else 0)
                        | (_native.LOSS_VALUE_ACTIVE_MASKS if value_active_masks else 0))
    _native.check(_native.lib().mappo_ppo_loss_f32(a, _native.stream_of(lg.device)), "mappo_ppo_loss_f32")
    return dlogits, dvalues.view_as(values)


def sample_supported(logits):
    """K14 takes this head's rollout sampling: float32 HIP logits of <= 64 actions, device-sampling mode, no autograd."""
    from . import distributions
    return (os.environ.get("MAPPO_FUSED_SAMPLE", "1") != "0" and torch.is_tensor(logits) and logits.is_cuda
            and logits.dtype == torch.float32 and logits.dim() == 2 and 0 < logits.shape[1] <= 64
            and not torch.is_grad_enabled() and distributions.SAMPLING_RNG == "device")


def sample_categorical(logits, available_actions=None):
    """(actions [rows, 1] int64, log-probs [rows, 1]) of one draw per row from softmax(masked logits)
    (``mappo_categorical_sample``, K14): what ``Categorical.forward`` -> ``FixedCategorical.sample`` / ``log_probs`` give
    (reference distributions.py:14-28, :55-68) as two launches -- the Exponential(1) noise from torch's generator and the
    kernel -- instead of ~15."""
    rows, na = logits.shape
    lg = logits.detach().contiguous()
    avail = None if available_actions is None else available_actions.detach().to(torch.float32).contiguous()
    noise = torch.empty_like(lg).exponential_(1.0)
    actions = torch.empty((rows, 1), dtype=torch.int64, device=lg.device)
    logp = torch.empty((rows, 1), dtype=torch.float32, device=lg.device)
    p = _native.ptr
    _native.check(_native.lib().mappo_categorical_sample(p(lg), p(avail), p(noise), p(actions), p(logp), rows, na,
                                                         _native.stream_of(lg.device)), "mappo_categorical_sample")
    return actions, logp
