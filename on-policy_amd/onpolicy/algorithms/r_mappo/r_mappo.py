"""MAPPO trainer: advantage normalisation, ppo_epoch x num_mini_batch clipped-surrogate updates.

Drop-in for the reference's ``R_MAPPO`` (onpolicy/algorithms/r_mappo/r_mappo.py: R_MAPPO :8,
cal_value_loss :52, ppo_update :91, train :171, prep_training :226, prep_rollout :230) with the
same constructor, methods, return values and ``train_info`` keys.  What differs is where the data
lives and when the host looks at it:

  * minibatches arrive as device tensors from the HBM buffer's fused gather kernels, so there is no
    ``torch.from_numpy(...).to(device)`` per field per minibatch (reference r_mappo.py:113-117);
  * the advantage statistics come from the moments the GAE kernel already accumulated, and the
    normalisation is folded into the gathers (reference r_mappo.py:179-187 makes ~6 full passes);
  * the six logged scalars are accumulated on the device and read back once per ``train()``
    instead of three ``.item()`` syncs per minibatch (reference r_mappo.py:212-214);
  * data-parallel training (one process per GPU, rollout threads sharded over ranks): gradients of
    actor and critic are summed with ONE RCCL all-reduce per update over a flat bucket that the
    parameters' ``.grad`` tensors are views of, preceded by one tiny all-reduce of the batch
    statistics that must be global (loss denominators, ValueNorm moments).  Every rank then
    applies the identical clipped Adam step -- the result equals the single-GPU update on the
    union of the ranks' minibatches up to float32 summation order.

The forward / backward maths of the actor and critic stays in PyTorch.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from onpolicy.utils.util import get_gard_norm, huber_loss, mse_loss
from onpolicy.utils.valuenorm import ValueNorm
from onpolicy.algorithms.utils.util import check
from onpolicy.utils import dist as mdist
from onpolicy.algorithms.utils import fused_loss
from onpolicy.algorithms.utils.fused_mlp import RowSource


class R_MAPPO():
    """
    :param args: (argparse.Namespace) flags (onpolicy/config.py).
    :param policy: (R_MAPPOPolicy) policy to update.
    :param device: (torch.device) device the networks live on.
    """

    def __init__(self, args, policy, device=torch.device("cpu")):
        self.device = device
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.policy = policy

        self.clip_param = args.clip_param
        self.ppo_epoch = args.ppo_epoch
        self.num_mini_batch = args.num_mini_batch
        self.data_chunk_length = args.data_chunk_length
        self.value_loss_coef = args.value_loss_coef
        self.entropy_coef = args.entropy_coef
        self.max_grad_norm = args.max_grad_norm
        self.huber_delta = args.huber_delta

        self._use_recurrent_policy = args.use_recurrent_policy
        self._use_naive_recurrent = args.use_naive_recurrent_policy
        self._use_max_grad_norm = args.use_max_grad_norm
        self._use_clipped_value_loss = args.use_clipped_value_loss
        self._use_huber_loss = args.use_huber_loss
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_value_active_masks = args.use_value_active_masks
        self._use_policy_active_masks = args.use_policy_active_masks

        self.value_normalizer = self._make_value_normalizer()

        # data parallelism over rollout threads; world size 1 unless torch.distributed is up
        self.dp = mdist.DataParallel(self.policy.actor, self.policy.critic, device)
        # set by train() while it feeds ppo_update with row-standardised observations
        self._obs_standardized = False
        # ppo_update as a captured HIP graph (update_graph.py), built on first use
        self._update_graph = None
        # Discrete head on a HIP device: loss + gradient in one kernel (K7) instead of the framework ops
        self._fused_loss = fused_loss.supported(self.policy, device) and self._fused_loss_allowed()

    # ------------------------------------------------------------------ what HAPPO overrides
    _use_factor = False          # minibatches may carry a 13th element (HAPPO factor); MAPPO ignores it
    _updates_normalizer = True   # ppo_update feeds the returns to the value normaliser (r_mappo.py:65)

    def _make_value_normalizer(self):
        assert (self._use_popart and self._use_valuenorm) == False, (
            "self._use_popart and self._use_valuenorm can not be set True simultaneously")
        if self._use_popart:
            return self.policy.critic.v_out
        if self._use_valuenorm:
            return ValueNorm(1, device=self.device)
        return None

    def _fused_loss_allowed(self):
        return True

    def _denormalize_advantages(self):
        """Whether the advantages are returns - D(value_preds) (r_mappo.py:179-182) or the raw difference."""
        return self._use_popart or self._use_valuenorm

    def _value_targets(self, return_batch, update_normalizer):
        """(target of the clipped error, target of the plain error), reference r_mappo.py:64-70."""
        if self._use_popart or self._use_valuenorm:
            if update_normalizer:
                self._normalizer_update(return_batch)
            target = self.value_normalizer.normalize(return_batch)
        else:
            target = return_batch
        return target, target

    def _ratio(self, action_log_probs, old_action_log_probs):
        return torch.exp(action_log_probs - old_action_log_probs)          # r_mappo.py:129

    # ------------------------------------------------------------------ losses
    def _normalizer_update(self, return_batch, moments=None):
        """ValueNorm / PopArt EMA update (reference r_mappo.py:65) from GLOBAL batch moments."""
        if moments is not None:
            self.value_normalizer.update(return_batch, batch_moments=moments)
            return
        if not self.dp.active:
            self.value_normalizer.update(return_batch)
            return
        x = return_batch.detach()
        # (torch.full, not torch.tensor(..., device=): a host scalar uploaded with a blocking copy drains the stream)
        stats = torch.stack([x.sum(0).reshape(()).double(), (x ** 2).sum(0).reshape(()).double(),
                             torch.full((), float(x.shape[0]), dtype=torch.float64, device=x.device)])
        self.dp.all_reduce(stats)
        mean = (stats[0] / stats[2]).float().reshape(1)
        mean_sq = (stats[1] / stats[2]).float().reshape(1)
        self.value_normalizer.update(return_batch, batch_moments=(mean, mean_sq))

    def cal_value_loss(self, values, value_preds_batch, return_batch, active_masks_batch):
        """Clipped (huber | mse) value loss against normalised returns
        (reference r_mappo.py:52-89); updates the value normaliser as a side effect."""
        return self._value_loss(values, value_preds_batch, return_batch, active_masks_batch, True)

    def _value_loss(self, values, value_preds_batch, return_batch, active_masks_batch, update_normalizer):
        value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param,
                                                                                    self.clip_param)
        target_clipped, target_original = self._value_targets(return_batch, update_normalizer)
        error_clipped = target_clipped - value_pred_clipped
        error_original = target_original - values

        if self._use_huber_loss:
            value_loss_clipped = huber_loss(error_clipped, self.huber_delta)
            value_loss_original = huber_loss(error_original, self.huber_delta)
        else:
            value_loss_clipped = mse_loss(error_clipped)
            value_loss_original = mse_loss(error_original)

        if self._use_clipped_value_loss:
            value_loss = torch.max(value_loss_original, value_loss_clipped)
        else:
            value_loss = value_loss_original

        if self._use_value_active_masks:
            value_loss = (value_loss * active_masks_batch).sum() / active_masks_batch.sum()
        else:
            value_loss = value_loss.mean()
        return value_loss

    # ------------------------------------------------------------------ one minibatch
    # PyTorch-ROCm's LayerNorm (and other row-wise kernels) index with 32 bits: at the north-star
    # size one minibatch is 13.1 M rows x 384 features = 5.0e9 elements and the kernels silently
    # wrap / fault.  Minibatches above this many elements per tensor are therefore evaluated in
    # row spans; gradients accumulate, so the update is the same full-minibatch update.
    MAX_TENSOR_ELEMENTS = 1 << 30

    def _row_spans(self, sample):
        """-> (spans, chunk_len).  Feed-forward minibatches (chunk_len None) are cut into row spans
        [lo, hi); recurrent ones ([L * mb, ...] sequence fields with row l * mb + j and [mb, ...] RNN states)
        into spans of whole chunks j in [lo, hi), every span keeping all L steps of its chunks."""
        rows = sample[10].shape[0] if sample[10] is not None else sample[5].shape[0]
        # a RowSource is never materialised by the fused trunk: what the networks write per row are 64-wide activations
        widest = max((64 if isinstance(t, RowSource) else int(np.prod(t.shape[1:])))
                     for t in (sample[0], sample[1]) if t is not None)
        cap = max(1, self.MAX_TENSOR_ELEMENTS // max(1, widest))
        recurrent = sample[2] is not None and sample[2].shape[0] != rows
        if rows <= cap:
            return [(0, rows)], None
        if recurrent:
            mb = sample[2].shape[0]
            chunk_len = rows // mb
            units, cap = mb, max(1, cap // chunk_len)
        else:
            chunk_len, units = None, rows
        n = -(-units // cap)
        step = -(-units // n)
        return [(lo, min(units, lo + step)) for lo in range(0, units, step)], chunk_len

    def ppo_update(self, sample, update_actor=True, _front_only=False, _scales=None):
        """One actor step and one critic step on a minibatch (reference r_mappo.py:91-169).
        -> (value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights).
        (``_front_only`` / ``_scales``: update_graph.py captures everything up to the gradients as one graph and hands the
        scalar prologue in as a static tensor.)"""
        share_obs_batch, obs_batch, rnn_states_batch, rnn_states_critic_batch, actions_batch, \
            value_preds_batch, return_batch, masks_batch, active_masks_batch, old_action_log_probs_batch, \
            adv_targ, available_actions_batch = sample[:12]
        factor_batch = check(sample[12]).to(**self.tpdv) if (self._use_factor and len(sample) > 12) else None

        old_action_log_probs_batch = check(old_action_log_probs_batch).to(**self.tpdv)
        adv_targ = check(adv_targ).to(**self.tpdv)
        value_preds_batch = check(value_preds_batch).to(**self.tpdv)
        return_batch = check(return_batch).to(**self.tpdv)
        active_masks_batch = check(active_masks_batch).to(**self.tpdv)

        spans, chunk_len = self._row_spans(sample)
        rows = adv_targ.shape[0]
        # In a data-parallel job each rank's loss is a mean over ITS minibatch; weighting it by
        # (local denominator / global denominator) makes the all-reduced gradient the gradient of
        # the global-batch mean.  Weights are exactly 1 for world size 1.
        normalized = (self._use_popart or self._use_valuenorm) and self._updates_normalizer
        # The normaliser is fed once per minibatch, AFTER the first forward pass and before the loss -- the
        # reference's order (r_mappo.py:120 evaluate_actions, then :65 update inside cal_value_loss).  It matters
        # under PopArt, whose update rescales the value head the forward pass runs through.  ``pending`` is what
        # the update will be fed: None = nothing to do, () = the local minibatch, (mean, mean_sq) = global moments.
        # (A minibatch cut into several row spans -- MAX_TENSOR_ELEMENTS, unfused routes only -- evaluates span 1 before the
        # update and the others after it; under PopArt, whose update rescales v_out, that deviates from a single pass by
        # the one EMA step (beta = 0.99999).  The fused trunk route only cuts above 2^30 / 64 = 16.7 M rows, _row_spans.)
        pending = () if normalized else None
        fused = self._fused_loss and adv_targ.is_cuda and actions_batch is not None
        # fused loss on a HIP device: denominators, their reciprocals and the returns' batch moments in three launches
        if _scales is not None:
            scales = _scales
        else:
            scales = self.dp.minibatch_scales(active_masks_batch, return_batch, self._use_policy_active_masks,
                                              self._use_value_active_masks) if fused else None
        w_actor = w_critic = 1.0
        if scales is not None:
            if normalized:
                pending = (scales[6:7], scales[7:8])
        elif self.dp.active:    # one small collective for the loss denominators and the normaliser moments
            w_actor, w_critic, moments = self.dp.minibatch_stats(
                active_masks_batch, return_batch, self._use_policy_active_masks, self._use_value_active_masks)
            if normalized:
                pending = moments

        self.dp.zero_grad(self.policy.actor_optimizer, self.policy.critic_optimizer)

        def feed_normalizer():
            nonlocal pending
            if pending is not None:
                self._normalizer_update(return_batch, pending or None)
                pending = None

        single = len(spans) == 1

        n_chunks = rows // chunk_len if chunk_len else None

        def cut(x, lo, hi):
            """The part of a minibatch tensor that belongs to span [lo, hi)."""
            if x is None or single:
                return x
            if isinstance(x, RowSource):
                return x.rows_slice(lo, hi)
            if chunk_len is None or x.shape[0] == n_chunks:        # row spans / per-chunk RNN states
                return x[lo:hi]
            tail = x.shape[1:]                                      # [L * mb, ...] with row l * mb + j
            return x.reshape(chunk_len, n_chunks, *tail)[:, lo:hi].reshape(chunk_len * (hi - lo), *tail)

        value_loss = policy_loss = dist_entropy = None
        ratios = []
        if fused:
            value_loss, policy_loss, dist_entropy, imp_weights = self._fused_spans(
                spans, cut, (share_obs_batch, obs_batch, rnn_states_batch, rnn_states_critic_batch, actions_batch,
                             value_preds_batch, return_batch, masks_batch, active_masks_batch,
                             old_action_log_probs_batch, adv_targ, available_actions_batch, factor_batch),
                w_actor, w_critic, feed_normalizer, update_actor, scales)
        for lo, hi in ([] if fused else spans):
            am = cut(active_masks_batch, lo, hi)
            values, action_log_probs, entropy = self.policy.evaluate_actions(
                cut(share_obs_batch, lo, hi), cut(obs_batch, lo, hi), cut(rnn_states_batch, lo, hi),
                cut(rnn_states_critic_batch, lo, hi), cut(actions_batch, lo, hi), cut(masks_batch, lo, hi),
                cut(available_actions_batch, lo, hi), am, **self._eval_kwargs())

            # clipped surrogate (r_mappo.py:129-139)
            adv_span = cut(adv_targ, lo, hi)
            imp_weights = self._ratio(action_log_probs, cut(old_action_log_probs_batch, lo, hi))
            surr1 = imp_weights * adv_span
            surr2 = torch.clamp(imp_weights, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_span
            surr = torch.min(surr1, surr2)
            if factor_batch is not None:
                surr = cut(factor_batch, lo, hi) * surr                            # happo_trainer.py:137-141
            per_sample = -torch.sum(surr, dim=-1, keepdim=True)
            if self._use_policy_active_masks:
                p_loss = (per_sample * am).sum() / am.sum()
            else:
                p_loss = per_sample.mean()
            feed_normalizer()
            v_loss = self._value_loss(values, cut(value_preds_batch, lo, hi), cut(return_batch, lo, hi), am,
                                      update_normalizer=False)

            # span weights: this span's share of the minibatch denominators (exactly 1 for one span)
            if len(spans) == 1:
                sw_actor = sw_critic = 1.0
            else:
                frac_rows = float(am.shape[0]) / rows
                frac_active = am.sum() / active_masks_batch.sum()
                sw_actor = frac_active if self._use_policy_active_masks else frac_rows
                sw_critic = frac_active if self._use_value_active_masks else frac_rows

            if update_actor:
                ((p_loss - entropy * self.entropy_coef) * (sw_actor * w_actor)).backward()
            (v_loss * self.value_loss_coef * (sw_critic * w_critic)).backward()

            if len(spans) == 1:
                value_loss, policy_loss, dist_entropy = v_loss, p_loss, entropy
            else:
                acc = lambda tot, x, w: x.detach() * w if tot is None else tot + x.detach() * w
                value_loss = acc(value_loss, v_loss, sw_critic)
                policy_loss = acc(policy_loss, p_loss, sw_actor)
                dist_entropy = acc(dist_entropy, entropy, sw_actor)
            ratios.append(imp_weights.detach() if len(spans) > 1 else imp_weights)
            del values, action_log_probs, entropy, surr1, surr2, surr, per_sample, p_loss, v_loss

        if not fused:
            imp_weights = ratios[0] if len(ratios) == 1 else torch.cat(ratios, 0)
        if _front_only:         # (update_graph.py: the gradient exchange and the optimiser steps follow separately)
            return value_loss, policy_loss, dist_entropy, imp_weights

        self.dp.all_reduce_grads()  # no-op for world size 1
        actor_grad_norm, critic_grad_norm = self._update_back(update_actor)
        return value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights

    def _update_back(self, update_actor, lr_devices=(None, None)):
        """Clipping + optimiser step of both networks on the (all-reduced) gradients -> (actor, critic) gradient norms."""
        actor_grad_norm = self._clip_and_step(self.policy.actor, self.policy.actor_optimizer,
                                              update_actor or not self.dp.active, lr_devices[0])
        critic_grad_norm = self._clip_and_step(self.policy.critic, self.policy.critic_optimizer, True, lr_devices[1])
        return actor_grad_norm, critic_grad_norm

    def _clip_and_step(self, net, optimizer, step, lr_device=None):
        """Clip the network's gradients to ``max_grad_norm`` (or only measure them) and take the optimiser step
        (reference r_mappo.py:146-153, :160-167) -> the gradient norm before clipping.  On a HIP device the whole thing is
        K13 (two launches instead of ~8); otherwise, or when a parameter has no gradient, the PyTorch calls."""
        from onpolicy.algorithms.utils import fused_optim
        params = [p for p in net.parameters() if p.requires_grad]
        if step and params and params[0].is_cuda and fused_optim.supported(optimizer, params):
            return fused_optim.clip_and_step(optimizer, params, self.max_grad_norm if self._use_max_grad_norm else None,
                                             lr_device)
        assert lr_device is None, "a captured update needs the fused clip + Adam kernels"
        if self._use_max_grad_norm:
            norm = nn.utils.clip_grad_norm_(net.parameters(), self.max_grad_norm)
        else:
            norm = get_gard_norm(net.parameters())
        if step:
            optimizer.step()
        return norm

    def _fused_spans(self, spans, cut, tensors, w_actor, w_critic, feed_normalizer, update_actor, scales=None):
        """The span loop of ppo_update through the fused loss kernel (K7): per span one forward to the
        head's logits / the critic's values, one ``mappo_ppo_loss_f32`` launch that evaluates the loss and
        its gradient, one backward from those gradients.  -> (value_loss, policy_loss, dist_entropy,
        mean ratio) as device scalars."""
        share_obs, obs, rnn_a, rnn_c, actions, value_preds, returns, masks, active, old_logp, adv, avail, factor = tensors
        dev, rows = adv.device, adv.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        actions = check(actions).to(**f32)
        avail = None if avail is None else check(avail).to(**f32)
        # Nothing below may touch the host: a Python scalar turned into a device tensor (torch.as_tensor(1.0, device=...))
        # is a blocking copy that drains the stream once per update -- the ~40 small launches up to the first trunk
        # kernel then run at launch latency instead of from a filled queue (0.5 ms per update at a 512-thread shard).
        if scales is not None:      # DataParallel.minibatch_scales: [inv (2) | scale of the four sums (4) | moments (2)]
            inv, scale4 = scales[0:2], scales[2:6]
        else:
            n_rows = torch.full((), float(rows), **f32)
            masked = self._use_policy_active_masks or self._use_value_active_masks
            active_total = active.sum() if masked else n_rows
            # [1 / policy denominator, 1 / value denominator, 1 / rows] of THIS rank's minibatch
            inv3 = 1.0 / torch.stack([active_total if self._use_policy_active_masks else n_rows,
                                      active_total if self._use_value_active_masks else n_rows, n_rows])
            inv_local = inv3[:2]
            # data-parallel weights are local / global denominators, so this is 1 / the GLOBAL denominators
            if torch.is_tensor(w_actor) or torch.is_tensor(w_critic):
                inv = (inv_local * torch.stack([torch.as_tensor(w_actor, **f32).reshape(()),
                                                torch.as_tensor(w_critic, **f32).reshape(())])).contiguous()
            elif w_actor == 1.0 and w_critic == 1.0:
                inv = inv_local
            else:
                inv = torch.stack([inv_local[0] * float(w_actor), inv_local[1] * float(w_critic)])
            scale4 = torch.stack([inv3[0], inv3[0], inv3[1], inv3[2]])
        sums = torch.zeros(4, dtype=torch.float64, device=dev)
        normalized = self._use_popart or self._use_valuenorm
        for lo, hi in spans:
            values, logits = self.policy.evaluate_logits(
                cut(share_obs, lo, hi), cut(obs, lo, hi), cut(rnn_a, lo, hi), cut(rnn_c, lo, hi), cut(masks, lo, hi),
                **self._eval_kwargs())
            feed_normalizer()           # reference order: forward, normaliser update, loss (r_mappo.py:120-66)
            norm = self.value_normalizer.denorm_scalars().to(**f32).contiguous() if normalized else None
            dlogits, dvalues = fused_loss.ppo_loss(
                logits, cut(avail, lo, hi), cut(actions, lo, hi), cut(old_logp, lo, hi), cut(adv, lo, hi),
                cut(active, lo, hi), cut(factor, lo, hi), values, cut(value_preds, lo, hi), cut(returns, lo, hi), norm, inv,
                sums, clip=self.clip_param, huber_delta=self.huber_delta, entropy_coef=self.entropy_coef,
                value_loss_coef=self.value_loss_coef, use_huber=self._use_huber_loss,
                use_clipped_value_loss=self._use_clipped_value_loss,
                policy_active_masks=self._use_policy_active_masks, value_active_masks=self._use_value_active_masks)
            if update_actor:
                torch.autograd.backward([logits, values], [dlogits, dvalues])
            else:
                values.backward(dvalues)
            del values, logits, dlogits, dvalues
        # sums = [policy loss, entropy, value loss, ratio] numerators -> local means, one launch
        means = sums.float() * scale4
        return means[2], means[0], means[1], means[3]

    def _fused_trunks(self, fold):
        """Both networks' trunks qualify for the fused kernels (K9) and expect the kind of rows the sampler would hand
        them (standardised iff the input LayerNorm is folded)."""
        from onpolicy.algorithms.utils import fused_mlp
        nets = (getattr(self.policy, "actor", None), getattr(self.policy, "critic", None))
        if not fused_mlp.enabled() or torch.device(self.device).type != "cuda" or any(n is None for n in nets):
            return False
        for net in nets:
            base = getattr(net, "base", None)
            if base is None or not fused_mlp.trunk_supported(base) or bool(base._use_feature_normalization) != bool(fold):
                return False
        return True

    def _eval_kwargs(self):
        return {"obs_standardized": True} if self._obs_standardized else {}

    # ------------------------------------------------------------------ one update phase
    def _advantages(self, buffer):
        """Normalised advantages for the samplers (reference r_mappo.py:179-187)."""
        if hasattr(buffer, "normalized_advantages"):
            reduce_fn = self.dp.all_reduce if self.dp.active else None
            if self._denormalize_advantages() == bool(self._use_popart or self._use_valuenorm):
                return buffer.normalized_advantages(self.value_normalizer, all_reduce=reduce_fn)
            return buffer.normalized_advantages(self.value_normalizer, all_reduce=reduce_fn,
                                                denormalize=self._denormalize_advantages())
        # Foreign buffers that hold host arrays in the reference's format (e.g. the reference's own
        # SharedReplayBuffer): same arithmetic in torch on this trainer's device.
        returns = torch.as_tensor(np.asarray(buffer.returns[:-1]), dtype=torch.float32)
        value_preds = torch.as_tensor(np.asarray(buffer.value_preds[:-1]), dtype=torch.float32)
        active = torch.as_tensor(np.asarray(buffer.active_masks[:-1]), dtype=torch.float32)
        if self._denormalize_advantages():
            value_preds = self.value_normalizer.denormalize(value_preds.to(self.device)).cpu()
        adv = returns - value_preds
        on = active != 0.0
        sums = torch.stack([adv[on].double().sum(), (adv[on].double() ** 2).sum(),
                            on.sum().double()]).to(self.device)
        if self.dp.active:
            self.dp.all_reduce(sums)
        sums = sums.cpu()
        mean = sums[0] / sums[2]
        std = torch.sqrt(torch.clamp(sums[1] / sums[2] - mean ** 2, min=0.0))
        return ((adv - mean.float()) / (std.float() + 1e-5)).numpy()

    def train(self, buffer, update_actor=True):
        """ppo_epoch passes over the buffer in num_mini_batch minibatches (reference r_mappo.py:171-224).
        -> dict with value_loss, policy_loss, dist_entropy, actor_grad_norm, critic_grad_norm, ratio
        (means over the updates; global-batch values in a data-parallel job)."""
        from onpolicy.utils import gemm_tuning
        with gemm_tuning.tuning():      # the update's GEMM shapes repeat every iteration: worth benchmarking once
            return self._train(buffer, update_actor)

    def _train(self, buffer, update_actor):
        advantages = self._advantages(buffer)
        # let the sampler do the parameter-free half of the input LayerNorm while it copies the rows
        # (only for observation widths the standardising gather has kernels for: wider rows are gathered as
        # they are and go through the policy's own feature_norm)
        fold = bool(getattr(buffer, "supports_standardized_obs", False)) and \
            getattr(buffer, "can_standardize_obs", lambda: True)() and \
            hasattr(self.policy, "can_fold_input_norm") and self.policy.can_fold_input_norm()
        gen_kwargs = {"standardize_obs": True} if fold else {}
        # ... and, where both trunks run through the fused hidden-64 kernels, not copy the rows at all
        if getattr(buffer, "supports_lazy_obs", False) and self._fused_trunks(fold):
            gen_kwargs["lazy_obs"] = True

        if hasattr(buffer, "plan_epochs"):
            buffer.plan_epochs(self.ppo_epoch)      # (a no-op of the device sampler; kept for foreign buffers)

        keys = ('value_loss', 'policy_loss', 'dist_entropy', 'actor_grad_norm', 'critic_grad_norm', 'ratio')
        totals = torch.zeros(len(keys), dtype=torch.float32, device=self.device)
        if self._update_graph is not None:
            self._update_graph.begin_train()    # (captures against parameters / optimiser state that were rebound are dropped)

        for _ in range(self.ppo_epoch):
            if self._use_recurrent_policy:
                data_generator = buffer.recurrent_generator(advantages, self.num_mini_batch,
                                                            self.data_chunk_length, **gen_kwargs)
            elif self._use_naive_recurrent:
                data_generator = buffer.naive_recurrent_generator(advantages, self.num_mini_batch, **gen_kwargs)
            else:
                data_generator = buffer.feed_forward_generator(advantages, self.num_mini_batch, **gen_kwargs)

            for sample in self._with_prologue_ahead(data_generator):
                self._obs_standardized = fold
                try:
                    value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights \
                        = self._run_update(sample, update_actor)
                finally:
                    self._obs_standardized = False
                with torch.no_grad():
                    ratio = imp_weights.detach()
                    ratio = ratio.reshape(()) if ratio.numel() == 1 else ratio.mean()     # (the fused loss returns the mean)
                    totals += torch.stack([
                        value_loss.detach().reshape(()), policy_loss.detach().reshape(()),
                        dist_entropy.detach().reshape(()),
                        torch.as_tensor(actor_grad_norm, **self.tpdv).reshape(()),
                        torch.as_tensor(critic_grad_norm, **self.tpdv).reshape(()),
                        ratio])

        self.dp.drop_scales()
        num_updates = self.ppo_epoch * self.num_mini_batch
        totals = self.dp.average_info(totals / num_updates)
        values = totals.tolist()  # the only device->host sync of the update phase
        return dict(zip(keys, values))

    def _run_update(self, sample, update_actor):
        """``ppo_update`` -- replayed from a captured HIP graph where the update qualifies (update_graph.py: ~10 launches per
        update instead of ~100 on the recurrent route), eagerly otherwise.  Subclasses that override ``ppo_update`` stay eager."""
        if type(self).ppo_update is R_MAPPO.ppo_update:
            if self._update_graph is None:
                from onpolicy.algorithms.r_mappo.update_graph import UpdateGraph
                self._update_graph = UpdateGraph(self)
            out = self._update_graph.run(sample, update_actor)
            if out is not None:
                return out
        return self.ppo_update(sample, update_actor)

    def _with_prologue_ahead(self, generator):
        """Data-parallel jobs with several minibatches per epoch: minibatch i + 1 is drawn BEFORE update i runs and its
        scalar prologue (local sums + the 32-byte all-reduce, ``DataParallel.begin_scales``) is put in flight, so the
        collective travels under update i's kernels instead of sitting in front of update i + 1's loss.  (One minibatch
        per epoch needs none of this: the whole-batch tuple is the same object in every epoch and its prologue is
        computed once per train().)  Costs one extra minibatch of sampler output alive at a time."""
        if not (self.dp.active and self._fused_loss and self.num_mini_batch > 1) or \
                os.environ.get("MAPPO_PROLOGUE_AHEAD", "1") == "0":
            yield from generator
            return
        it = iter(generator)
        cur = next(it, None)
        while cur is not None:
            nxt = next(it, None)
            if nxt is not None and torch.is_tensor(nxt[8]) and torch.is_tensor(nxt[6]) and nxt[8].is_cuda:
                self.dp.begin_scales(nxt[8], nxt[6], self._use_policy_active_masks, self._use_value_active_masks)
            yield cur
            cur = nxt

    def prep_training(self):
        self.policy.actor.train()
        self.policy.critic.train()

    def prep_rollout(self):
        self.policy.actor.eval()
        self.policy.critic.eval()
