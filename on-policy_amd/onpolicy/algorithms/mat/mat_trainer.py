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
from onpolicy.utils.util import get_gard_norm, huber_loss, mse_loss
from onpolicy.utils.valuenorm import ValueNorm


class MATTrainer(object):
    def __init__(self, args, policy, num_agents, device=torch.device("cpu")):
        self.device = device
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.policy = policy
        self.num_agents = num_agents
        for name in ("clip_param", "ppo_epoch", "num_mini_batch", "data_chunk_length", "value_loss_coef",
                     "entropy_coef", "max_grad_norm", "huber_delta", "dec_actor"):
            setattr(self, name, getattr(args, name))
        self._use_recurrent_policy = args.use_recurrent_policy
        self._use_naive_recurrent = args.use_naive_recurrent_policy
        self._use_max_grad_norm = args.use_max_grad_norm
        self._use_clipped_value_loss = args.use_clipped_value_loss
        self._use_huber_loss = args.use_huber_loss
        self._use_valuenorm = args.use_valuenorm
        self._use_value_active_masks = args.use_value_active_masks
        self._use_policy_active_masks = args.use_policy_active_masks
        self.value_normalizer = ValueNorm(1, device=self.device) if self._use_valuenorm else None

    def cal_value_loss(self, values, value_preds_batch, return_batch, active_masks_batch):
        value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param, self.clip_param)
        if self._use_valuenorm:
            self.value_normalizer.update(return_batch)
            target = self.value_normalizer.normalize(return_batch)
        else:
            target = return_batch
        error_clipped, error_original = target - value_pred_clipped, target - values
        loss_of = (lambda e: huber_loss(e, self.huber_delta)) if self._use_huber_loss else mse_loss
        value_loss = loss_of(error_original)
        if self._use_clipped_value_loss:
            value_loss = torch.max(value_loss, loss_of(error_clipped))
        if self._use_value_active_masks:
            return (value_loss * active_masks_batch).sum() / active_masks_batch.sum()
        return value_loss.mean()

    def ppo_update(self, sample):
        share_obs_batch, obs_batch, rnn_states_batch, rnn_states_critic_batch, actions_batch, value_preds_batch, \
            return_batch, masks_batch, active_masks_batch, old_action_log_probs_batch, adv_targ, \
            available_actions_batch = sample[:12]
        old_action_log_probs_batch = check(old_action_log_probs_batch).to(**self.tpdv)
        adv_targ = check(adv_targ).to(**self.tpdv)
        value_preds_batch = check(value_preds_batch).to(**self.tpdv)
        return_batch = check(return_batch).to(**self.tpdv)
        active_masks_batch = check(active_masks_batch).to(**self.tpdv)

        values, action_log_probs, dist_entropy = self.policy.evaluate_actions(
            share_obs_batch, obs_batch, rnn_states_batch, rnn_states_critic_batch, actions_batch, masks_batch,
            available_actions_batch, active_masks_batch)
        imp_weights = torch.exp(action_log_probs - old_action_log_probs_batch)
        surr1 = imp_weights * adv_targ
        surr2 = torch.clamp(imp_weights, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_targ
        surrogate = -torch.sum(torch.min(surr1, surr2), dim=-1, keepdim=True)
        if self._use_policy_active_masks:
            policy_loss = (surrogate * active_masks_batch).sum() / active_masks_batch.sum()
        else:
            policy_loss = surrogate.mean()
        value_loss = self.cal_value_loss(values, value_preds_batch, return_batch, active_masks_batch)
        loss = policy_loss - dist_entropy * self.entropy_coef + value_loss * self.value_loss_coef

        self.policy.optimizer.zero_grad()
        loss.backward()
        if self._use_max_grad_norm:
            grad_norm = nn.utils.clip_grad_norm_(self.policy.transformer.parameters(), self.max_grad_norm)
        else:
            grad_norm = get_gard_norm(self.policy.transformer.parameters())
        self.policy.optimizer.step()
        return value_loss, grad_norm, policy_loss, dist_entropy, grad_norm, imp_weights

    def _advantages(self, buffer):
        """(advantages - mean) / (std + 1e-5) over the active entries (mat_trainer.py:160-164)."""
        if hasattr(buffer, "normalized_advantages"):
            return buffer.normalized_advantages(self.value_normalizer)
        adv = np.asarray(buffer.advantages)
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
                    totals += torch.stack([
                        value_loss.detach().reshape(()), policy_loss.detach().reshape(()),
                        dist_entropy.detach().reshape(()), torch.as_tensor(actor_grad_norm, **self.tpdv).reshape(()),
                        torch.as_tensor(critic_grad_norm, **self.tpdv).reshape(()),
                        imp_weights.detach().mean().reshape(())])
        return dict(zip(keys, (totals / (self.ppo_epoch * self.num_mini_batch)).tolist()))

    def prep_training(self):
        self.policy.train()

    def prep_rollout(self):
        self.policy.eval()
