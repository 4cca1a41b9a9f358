"""``MATTrainer``: PPO update of the Multi-Agent Transformer -- interface and arithmetic of the reference's
onpolicy/algorithms/mat/mat_trainer.py (cal_value_loss :50-88, ppo_update :90-152, train :154-203): one network, one
optimiser, one joint loss; minibatches come from ``feed_forward_generator_transformer`` (whole env steps, agents
kept together) and the advantages are the buffer's own GAE accumulator (the mat branches of ``compute_returns``).

On the HBM buffer the advantage normalisation uses the moments the GAE launch already produced
(``normalized_advantages``) and the standardisation is folded into the sampler's gather; the logged scalars stay on
the device until the end of ``train`` (one host sync per update phase instead of six per minibatch)."""
import numpy as np
import torch
import torch.nn as nn

from onpolicy.algorithms.utils.util import check
from onpolicy.utils import dist as mdist
from onpolicy.utils.util import get_gard_norm, huber_loss, mse_loss
from onpolicy.utils.valuenorm import ValueNorm


_HYPER = ("clip_param", "ppo_epoch", "num_mini_batch", "data_chunk_length", "value_loss_coef", "entropy_coef",
          "max_grad_norm", "huber_delta", "dec_actor")
_SWITCHES = {"_use_recurrent_policy": "use_recurrent_policy", "_use_naive_recurrent": "use_naive_recurrent_policy",
             "_use_max_grad_norm": "use_max_grad_norm", "_use_clipped_value_loss": "use_clipped_value_loss",
             "_use_huber_loss": "use_huber_loss", "_use_valuenorm": "use_valuenorm",
             "_use_value_active_masks": "use_value_active_masks", "_use_policy_active_masks": "use_policy_active_masks"}


def _masked_mean(x, weights, use_weights):
    return (x * weights).sum() / weights.sum() if use_weights else x.mean()


class MATTrainer(object):
    def __init__(self, args, policy, num_agents, device=torch.device("cpu")):
        self.policy, self.num_agents, self.device = policy, num_agents, device
        self.tpdv = dict(dtype=torch.float32, device=device)
        for name in _HYPER:
            setattr(self, name, getattr(args, name))
        for attr, flag in _SWITCHES.items():
            setattr(self, attr, getattr(args, flag))
        self.value_normalizer = ValueNorm(1, device=self.device) if self._use_valuenorm else None
        # data parallel over rollout threads (no-op for one process): the transformer's gradients live in one flat
        # bucket that is all-reduced once per minibatch, next to one small collective for the loss denominators and
        # the ValueNorm moments -- the same scheme as R_MAPPO (onpolicy/utils/dist.py)
        self.dp = mdist.DataParallel(policy.transformer, nn.Module(), device)

    def cal_value_loss(self, values, value_preds_batch, return_batch, active_masks_batch, batch_moments=None):
        """Clipped value loss against the (normalised) returns; the normaliser is updated with this minibatch first
        (``batch_moments``: the all-reduced moments of the GLOBAL minibatch in a data-parallel job)."""
        if self._use_valuenorm:
            self.value_normalizer.update(return_batch, batch_moments=batch_moments)
            target = self.value_normalizer.normalize(return_batch)
        else:
            target = return_batch
        per_sample = (lambda e: huber_loss(e, self.huber_delta)) if self._use_huber_loss else mse_loss
        loss = per_sample(target - values)
        if self._use_clipped_value_loss:
            near_old = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param, self.clip_param)
            loss = torch.max(loss, per_sample(target - near_old))
        return _masked_mean(loss, active_masks_batch, self._use_value_active_masks)

    def _surrogate(self, action_log_probs, old_action_log_probs, adv_targ, active_masks):
        ratio = torch.exp(action_log_probs - old_action_log_probs)
        clipped = torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        objective = torch.min(ratio * adv_targ, clipped * adv_targ).sum(dim=-1, keepdim=True)
        return _masked_mean(-objective, active_masks, self._use_policy_active_masks), ratio

    def ppo_update(self, sample):
        """One minibatch: joint loss = surrogate - entropy bonus + value loss, one backward, one Adam step.
        -> (value_loss, grad_norm, policy_loss, dist_entropy, grad_norm, importance weights), the reference's tuple."""
        share_obs, obs, rnn_a, rnn_c, actions, value_preds, returns, masks, active, old_logp, adv, avail = sample[:12]
        old_logp, adv, value_preds, returns, active = (check(x).to(**self.tpdv)
                                                       for x in (old_logp, adv, value_preds, returns, active))
        # each rank's loss terms are means over ITS rows; weighting them by local / global denominators makes the
        # summed gradient the gradient of the global-batch means (weights are exactly 1 for one process)
        w_actor, w_critic, moments = 1.0, 1.0, None
        if self.dp.active:
            w_actor, w_critic, moments = self.dp.minibatch_stats(active, returns, self._use_policy_active_masks,
                                                                 self._use_value_active_masks)
        values, action_log_probs, dist_entropy = self.policy.evaluate_actions(
            share_obs, obs, rnn_a, rnn_c, actions, masks, avail, active)
        policy_loss, imp_weights = self._surrogate(action_log_probs, old_logp, adv, active)
        value_loss = self.cal_value_loss(values, value_preds, returns, active, moments)
        loss = (policy_loss - dist_entropy * self.entropy_coef) * w_actor + value_loss * self.value_loss_coef * w_critic
        self._last_weights = (w_actor, w_critic)

        self.dp.zero_grad(self.policy.optimizer)
        loss.backward()
        self.dp.all_reduce_grads()
        params = self.policy.transformer.parameters()
        grad_norm = nn.utils.clip_grad_norm_(params, self.max_grad_norm) if self._use_max_grad_norm \
            else get_gard_norm(params)
        self.policy.optimizer.step()
        return value_loss, grad_norm, policy_loss, dist_entropy, grad_norm, imp_weights

    def _advantages(self, buffer):
        """(advantages - mean) / (std + 1e-5) over the active entries (mat_trainer.py:160-164)."""
        if hasattr(buffer, "normalized_advantages"):
            return buffer.normalized_advantages(self.value_normalizer,
                                                all_reduce=self.dp.all_reduce if self.dp.active else None)
        adv = np.asarray(buffer.advantages)
        if self.dp.active:         # host buffers in a multi-rank job: global moments from three all-reduced sums
            on = np.asarray(buffer.active_masks[:-1]) != 0.0
            sums = torch.tensor([adv[on].astype(np.float64).sum(), (adv[on].astype(np.float64) ** 2).sum(),
                                 float(on.sum())], dtype=torch.float64, device=self.device)
            self.dp.all_reduce(sums)
            mean = float(sums[0] / sums[2])
            std = float(torch.sqrt(torch.clamp(sums[1] / sums[2] - mean ** 2, min=0.0)))
            return (adv - np.float32(mean)) / (np.float32(std) + 1e-5)
        masked = adv.copy()
        masked[np.asarray(buffer.active_masks[:-1]) == 0.0] = np.nan
        return (adv - np.nanmean(masked)) / (np.nanstd(masked) + 1e-5)

    def train(self, buffer):
        advantages = self._advantages(buffer)
        keys = ('value_loss', 'policy_loss', 'dist_entropy', 'actor_grad_norm', 'critic_grad_norm', 'ratio')
        totals = torch.zeros(len(keys), dtype=torch.float32, device=self.device)
        for _ in range(self.ppo_epoch):
            for sample in buffer.feed_forward_generator_transformer(advantages, self.num_mini_batch):
                value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights = \
                    self.ppo_update(sample)
                with torch.no_grad():
                    # rank means become global means when weighted by local / global denominators and SUMMED over
                    # ranks; average_info below divides by the world size, hence the factor
                    w_actor, w_critic = (torch.as_tensor(w, **self.tpdv).reshape(()) * self.dp.world_size
                                         if self.dp.active else 1.0 for w in self._last_weights)
                    totals += torch.stack([
                        value_loss.detach().reshape(()) * w_critic, policy_loss.detach().reshape(()) * w_actor,
                        dist_entropy.detach().reshape(()) * w_actor,
                        torch.as_tensor(actor_grad_norm, **self.tpdv).reshape(()),
                        torch.as_tensor(critic_grad_norm, **self.tpdv).reshape(()),
                        imp_weights.detach().mean().reshape(())])
        totals = self.dp.average_info(totals / (self.ppo_epoch * self.num_mini_batch))
        return dict(zip(keys, totals.tolist()))

    def prep_training(self):
        self.policy.train()

    def prep_rollout(self):
        self.policy.eval()
