"""Action distributions and their linear heads.

Public names follow the reference's onpolicy/algorithms/utils/distributions.py (FixedCategorical :14,
FixedNormal :32, FixedBernoulli :44, Categorical :55, DiagGaussian :71, Bernoulli :94, AddBias :106)
because ACTLayer, checkpoints (``linear``, ``fc_mean``, ``logstd._bias``) and user code refer to them.

The categorical is written directly on tensors instead of subclassing torch.distributions: the
formulas are the ones torch.distributions.Categorical evaluates (normalised logits =
x - logsumexp(x); probs = softmax; entropy = -sum(clamp(logits) * probs); sample = multinomial),
minus the argument validation and lazy-property machinery that dominate at small batch sizes.
Unavailable actions are masked with ``torch.where`` rather than the reference's boolean-index
assignment (distributions.py:67), which on a GPU costs a device->host sync per call.
"""
import torch
import torch.nn as nn

from .tall_linear import TallLinear
from .util import init


# 'device': torch.multinomial on the tensor's own device (fast path).  'host': draw the noise on
# the CPU generator exactly as torch.multinomial does there (one Exponential(1) per probability,
# argmax of p / q) and upload it, so that a GPU run reproduces the reference's CPU action stream
# for identical probabilities (integer-sampling parity mode).
SAMPLING_RNG = "device"


def set_sampling_rng(mode):
    global SAMPLING_RNG
    assert mode in ("device", "host")
    SAMPLING_RNG = mode


class FixedCategorical(object):
    def __init__(self, logits):
        self.logits = logits - logits.logsumexp(dim=-1, keepdim=True)
        self._probs = None

    @property
    def probs(self):
        if self._probs is None:
            self._probs = torch.softmax(self.logits, dim=-1)
        return self._probs

    def sample(self):
        p = self.probs
        flat = p.reshape(-1, p.size(-1))
        if SAMPLING_RNG == "host" and flat.is_cuda:
            q = torch.empty(flat.shape, dtype=flat.dtype).exponential_(1)      # CPU generator
            return (flat / q.to(flat.device)).argmax(-1, keepdim=True).reshape(p.shape[:-1] + (1,))
        return torch.multinomial(flat, 1, True).reshape(p.shape[:-1] + (1,))

    def log_prob(self, value):
        idx = value.long().unsqueeze(-1)
        return self.logits.gather(-1, idx).squeeze(-1)

    def log_probs(self, actions):
        return self.log_prob(actions.squeeze(-1)).view(actions.size(0), -1).sum(-1).unsqueeze(-1)

    def entropy(self):
        logits = torch.clamp(self.logits, min=torch.finfo(self.logits.dtype).min)
        return -(logits * self.probs).sum(-1)

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)

    # used by the TRPO-style evaluators of the reference
    @property
    def mean(self):
        return torch.full(self.logits.shape[:-1], float('nan'), device=self.logits.device)

    stddev = mean


class FixedNormal(torch.distributions.Normal):
    def log_probs(self, actions):
        return super().log_prob(actions).sum(-1, keepdim=True)

    def entropy(self):
        return super().entropy().sum(-1)

    def mode(self):
        return self.mean


class FixedBernoulli(torch.distributions.Bernoulli):
    def log_probs(self, actions):
        return super().log_prob(actions).view(actions.size(0), -1).sum(-1).unsqueeze(-1)

    def entropy(self):
        return super().entropy().sum(-1)

    def mode(self):
        return torch.gt(self.probs, 0.5).float()


def _head(num_inputs, num_outputs, use_orthogonal, gain):
    w_init = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    return init(TallLinear(num_inputs, num_outputs), w_init, lambda b: nn.init.constant_(b, 0), gain)


class Categorical(nn.Module):
    def __init__(self, num_inputs, num_outputs, use_orthogonal=True, gain=0.01):
        super(Categorical, self).__init__()
        self.linear = _head(num_inputs, num_outputs, use_orthogonal, gain)

    def forward(self, x, available_actions=None):
        x = self.linear(x)
        if available_actions is not None:
            x = torch.where(available_actions == 0, torch.full_like(x, -1e10), x)
        return FixedCategorical(logits=x)


class DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs, use_orthogonal=True, gain=0.01):
        super(DiagGaussian, self).__init__()
        self.fc_mean = _head(num_inputs, num_outputs, use_orthogonal, gain)
        self.logstd = AddBias(torch.zeros(num_outputs))

    def forward(self, x):
        action_mean = self.fc_mean(x)
        action_logstd = self.logstd(torch.zeros_like(action_mean))
        return FixedNormal(action_mean, action_logstd.exp())


class Bernoulli(nn.Module):
    def __init__(self, num_inputs, num_outputs, use_orthogonal=True, gain=0.01):
        super(Bernoulli, self).__init__()
        self.linear = _head(num_inputs, num_outputs, use_orthogonal, gain)

    def forward(self, x):
        return FixedBernoulli(logits=self.linear(x))


class AddBias(nn.Module):
    def __init__(self, bias):
        super(AddBias, self).__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1, 1)
        return x + self._bias.t().view(*shape)
