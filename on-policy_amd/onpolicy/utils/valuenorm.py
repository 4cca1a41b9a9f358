"""Running, debiased mean / variance normaliser for value targets (default ``use_valuenorm``).

Public surface of the reference's onpolicy/utils/valuenorm.py (ValueNorm :8, running_mean_var :32, update :39,
normalize :57, denormalize :68); the arithmetic lives in ``running_moments.DebiasedMoments``.  Device-side
differences: the statistics are buffers on the trainer's device; tensors in give tensors out (the reference always
returns numpy from ``denormalize``, valuenorm.py:77 -- a D2H copy per call; numpy in still gives numpy out);
``denorm_scalars()`` feeds the GAE kernel; ``update`` can take all-reduced batch moments.
"""
import numpy as np
import torch
import torch.nn as nn

from onpolicy.utils.running_moments import DebiasedMoments


class ValueNorm(DebiasedMoments, nn.Module):
    def __init__(self, input_shape, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device=torch.device("cpu")):
        nn.Module.__init__(self)
        self.input_shape, self.norm_axes = input_shape, norm_axes
        self.beta, self.epsilon, self.per_element_update = beta, epsilon, per_element_update
        self.tpdv = dict(dtype=torch.float32, device=device)
        for name, shape in (("running_mean", input_shape), ("running_mean_sq", input_shape), ("debiasing_term", ())):
            self.register_buffer(name, torch.zeros(shape, **self.tpdv))

    def reset_parameters(self):
        self.zero_moments()

    def running_mean_var(self):
        return self._mean_var()

    def update(self, input_vector, batch_moments=None):
        """EMA step from a batch, or from ``batch_moments = (mean, mean_sq)`` -- data-parallel training passes the
        all-reduced global moments so that every rank applies the identical update."""
        weight = self.beta
        if self.per_element_update:          # one decay per sample instead of one per batch (valuenorm.py:47-49)
            weight = self.beta ** np.prod(input_vector.size()[:self.norm_axes])
        self._fold_in(input_vector, batch_moments, weight)
