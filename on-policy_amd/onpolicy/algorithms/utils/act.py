"""Action head: maps actor features to a distribution for the env's action space, samples /
evaluates actions.  Interface of the reference's onpolicy/algorithms/utils/act.py (ACTLayer :5,
forward :44, get_probs :91, evaluate_actions :115); the space is recognised by class name.
"""
import torch
import torch.nn as nn

from .distributions import Bernoulli, Categorical, DiagGaussian, FixedCategorical


def _masked_mean(x, active_masks, squeeze=True):
    """(x * m).sum() / m.sum() with m = active_masks ([B,1]); plain mean when m is None."""
    if active_masks is None:
        return x.mean()
    m = active_masks.squeeze(-1) if squeeze else active_masks
    return (x * m).sum() / active_masks.sum()


class ACTLayer(nn.Module):
    def __init__(self, action_space, inputs_dim, use_orthogonal, gain, args=None):
        super(ACTLayer, self).__init__()
        self.mixed_action = False
        self.multi_discrete = False
        self.mujoco_box = False
        self.action_type = kind = action_space.__class__.__name__
        if kind == "Discrete":
            self.action_out = Categorical(inputs_dim, action_space.n, use_orthogonal, gain)
        elif kind == "Box":
            self.mujoco_box = True
            self.action_out = DiagGaussian(inputs_dim, action_space.shape[0], use_orthogonal, gain)
        elif kind == "MultiBinary":
            self.action_out = Bernoulli(inputs_dim, action_space.shape[0], use_orthogonal, gain)
        elif kind == "MultiDiscrete":
            self.multi_discrete = True
            dims = action_space.high - action_space.low + 1
            self.action_outs = nn.ModuleList(
                [Categorical(inputs_dim, int(d), use_orthogonal, gain) for d in dims])
        else:  # [Box, Discrete]
            self.mixed_action = True
            self.action_outs = nn.ModuleList([
                DiagGaussian(inputs_dim, action_space[0].shape[0], use_orthogonal, gain),
                Categorical(inputs_dim, action_space[1].n, use_orthogonal, gain)])

    def _single(self, x, available_actions):
        if self.mujoco_box or self.action_type == "MultiBinary":
            return self.action_out(x)
        return self.action_out(x, available_actions)

    def forward(self, x, available_actions=None, deterministic=False):
        if self.mixed_action or self.multi_discrete:
            actions, log_probs = [], []
            for head in self.action_outs:
                dist = head(x)
                a = dist.mode() if deterministic else dist.sample()
                log_probs.append(dist.log_probs(a))
                actions.append(a.float() if self.mixed_action else a)
            actions = torch.cat(actions, -1)
            log_probs = torch.cat(log_probs, -1)
            if self.mixed_action:
                log_probs = log_probs.sum(-1, keepdim=True)
            return actions, log_probs
        if self.action_type == "Discrete" and not deterministic and not torch.is_grad_enabled() and x.is_cuda:
            return self.from_logits(self.action_out.linear(x), available_actions, deterministic)
        dist = self._single(x, available_actions)
        actions = dist.mode() if deterministic else dist.sample()
        return actions, dist.log_probs(actions)

    def from_logits(self, logits, available_actions=None, deterministic=False):
        """``forward`` for a Discrete head whose Linear was already evaluated (inside the fused trunk launch): masking,
        sampling and log-probabilities as ``Categorical.forward`` + ``forward`` do them (reference act.py:44-60,
        distributions.py:55-68)."""
        assert self.action_type == "Discrete"
        from . import fused_loss
        if not deterministic and fused_loss.sample_supported(logits):
            return fused_loss.sample_categorical(logits, available_actions)      # K14: masking + sample + log-prob
        if available_actions is not None:
            logits = torch.where(available_actions == 0, torch.full_like(logits, -1e10), logits)
        dist = FixedCategorical(logits=logits)
        actions = dist.mode() if deterministic else dist.sample()
        return actions, dist.log_probs(actions)

    def get_probs(self, x, available_actions=None):
        if self.mixed_action or self.multi_discrete:
            return torch.cat([head(x).probs for head in self.action_outs], -1)
        return self._single(x, available_actions).probs

    def evaluate_actions_trpo(self, x, action, available_actions=None, active_masks=None):
        """What the trust-region trainer needs besides log-probs and entropy (reference act.py:180-235):
        -> (action_log_probs, dist_entropy, mean, std, normalised logits).  ``mean`` / ``std`` carry gradients for
        Box heads only; categorical heads report theirs as NaN like ``torch.distributions.Categorical`` and are
        compared through their normalised logits instead (``None`` for Box heads)."""
        if self.mixed_action:
            raise NotImplementedError("the trust-region update is not defined for the mixed action head")
        if self.multi_discrete:
            log_probs, ents, means, stds, logits = [], [], [], [], []
            for head, act in zip(self.action_outs, torch.transpose(action, 0, 1)):
                dist = head(x)
                log_probs.append(dist.log_probs(act))
                ents.append(_masked_mean(dist.entropy(), active_masks))
                means.append(dist.mean)
                stds.append(dist.stddev)
                logits.append(dist.logits)
            # the reference averages the per-head entropies through a fresh tensor, i.e. without gradient
            return (torch.cat(log_probs, -1), torch.stack(ents).detach().mean(), torch.cat(means, -1),
                    torch.cat(stds, -1), torch.cat(logits, -1))
        dist = self._single(x, available_actions)
        categorical = isinstance(dist, FixedCategorical)
        ent = dist.entropy()
        if active_masks is None:
            entropy = ent.mean()
        elif categorical:
            entropy = _masked_mean(ent, active_masks)
        else:
            # the reference multiplies the [B] entropies by the [B, 1] masks (act.py:231), a [B, B] outer product
            # whose normalised sum is just the sum of the entropies; logged only -- computed here without the product
            entropy = ent.sum()
        return dist.log_probs(action), entropy, dist.mean, dist.stddev, (dist.logits if categorical else None)

    def evaluate_actions(self, x, action, available_actions=None, active_masks=None):
        if self.mixed_action:
            cont, disc = action.split((2, 1), -1)
            log_probs, ents = [], []
            for head, act in zip(self.action_outs, (cont, disc.long())):
                dist = head(x)
                log_probs.append(dist.log_probs(act))
                ent = dist.entropy()
                if active_masks is not None:
                    same_rank = ent.dim() == active_masks.dim()
                    ents.append(_masked_mean(ent, active_masks, squeeze=not same_rank))
                else:
                    ents.append(ent.mean())
            action_log_probs = torch.cat(log_probs, -1).sum(-1, keepdim=True)
            dist_entropy = ents[0] / 2.0 + ents[1] / 0.98
        elif self.multi_discrete:
            log_probs, ents = [], []
            for head, act in zip(self.action_outs, torch.transpose(action, 0, 1)):
                dist = head(x)
                log_probs.append(dist.log_probs(act))
                ents.append(_masked_mean(dist.entropy(), active_masks))
            action_log_probs = torch.cat(log_probs, -1)
            dist_entropy = sum(ents) / len(ents)
        else:
            dist = self._single(x, available_actions)
            action_log_probs = dist.log_probs(action)
            dist_entropy = _masked_mean(dist.entropy(), active_masks)
        return action_log_probs, dist_entropy
