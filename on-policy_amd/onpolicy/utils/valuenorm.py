"""Running, debiased mean / variance normaliser for value targets (default ``use_valuenorm``).

Same maths and public surface as the reference's onpolicy/utils/valuenorm.py (ValueNorm :8,
running_mean_var :32, update :39, normalize :57, denormalize :68) with three device-side
differences:
  * the statistics are registered buffers that live on the trainer's device;
  * ``denormalize`` / ``normalize`` return a tensor on that device when given a tensor (the
    reference always returns numpy from denormalize, valuenorm.py:77, forcing a D2H copy per
    call); numpy in still gives numpy out;
  * ``denorm_scalars()`` hands the GAE kernel (sigma, mu) as a 2-float device tensor without a
    host sync, and ``update`` can take batch moments that were all-reduced across GPUs.
"""
import numpy as np
import torch
import torch.nn as nn


class ValueNorm(nn.Module):
    def __init__(self, input_shape, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device=torch.device("cpu")):
        super(ValueNorm, self).__init__()
        self.input_shape = input_shape
        self.norm_axes = norm_axes
        self.epsilon = epsilon
        self.beta = beta
        self.per_element_update = per_element_update
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.register_buffer("running_mean", torch.zeros(input_shape, **self.tpdv))
        self.register_buffer("running_mean_sq", torch.zeros(input_shape, **self.tpdv))
        self.register_buffer("debiasing_term", torch.tensor(0.0, **self.tpdv))

    def reset_parameters(self):
        self.running_mean.zero_()
        self.running_mean_sq.zero_()
        self.debiasing_term.zero_()

    def running_mean_var(self):
        debias = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / debias
        mean_sq = self.running_mean_sq / debias
        var = (mean_sq - mean ** 2).clamp(min=1e-2)
        return mean, var

    def denorm_scalars(self):
        """float32 device tensor [sigma, mu] for mappo_gae_f32 (input_shape == 1)."""
        mean, var = self.running_mean_var()
        return torch.stack([torch.sqrt(var).reshape(()), mean.reshape(())])

    def _as_tensor(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(**self.tpdv)

    @torch.no_grad()
    def update(self, input_vector, batch_moments=None):
        """EMA update from a batch.  ``batch_moments = (mean, mean_sq)`` overrides the local batch
        statistics (data-parallel training passes the all-reduced global moments so that every
        rank applies the identical update)."""
        if batch_moments is None:
            x = self._as_tensor(input_vector)
            axes = tuple(range(self.norm_axes))
            batch_mean = x.mean(dim=axes)
            batch_sq_mean = (x ** 2).mean(dim=axes)
        else:
            batch_mean, batch_sq_mean = batch_moments
        if self.per_element_update:
            batch_size = np.prod(input_vector.size()[:self.norm_axes])
            weight = self.beta ** batch_size
        else:
            weight = self.beta
        self.running_mean.mul_(weight).add_(batch_mean * (1.0 - weight))
        self.running_mean_sq.mul_(weight).add_(batch_sq_mean * (1.0 - weight))
        self.debiasing_term.mul_(weight).add_(1.0 * (1.0 - weight))

    def normalize(self, input_vector):
        x = self._as_tensor(input_vector)
        mean, var = self.running_mean_var()
        lead = (None,) * self.norm_axes
        return (x - mean[lead]) / torch.sqrt(var)[lead]

    def denormalize(self, input_vector):
        """x * sqrt(var) + mean.  ndarray -> ndarray (reference behaviour); tensor -> tensor."""
        as_numpy = isinstance(input_vector, np.ndarray)
        x = self._as_tensor(input_vector)
        mean, var = self.running_mean_var()
        lead = (None,) * self.norm_axes
        out = x * torch.sqrt(var)[lead] + mean[lead]
        return out.cpu().numpy() if as_numpy else out
