"""PopArt output layer (``--use_popart``): a Linear head whose targets are normalised by running
statistics, with the weights rescaled on every statistics update so the de-normalised output is
preserved.  Surface of the reference's onpolicy/algorithms/utils/popart.py (PopArt :7, update :49,
debiased_mean_var :72, normalize :78, denormalize :88).

Differences: statistics are buffers; ``update`` rescales ``weight`` / ``bias`` in place (the
reference assigns plain tensors to registered Parameters, popart.py:64,69-70, which raises
TypeError on CPU); tensors in give tensors out of ``denormalize``; ``denorm_scalars`` feeds the GAE
kernel.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from onpolicy.utils.running_moments import DebiasedMoments


class PopArt(DebiasedMoments, nn.Module):
    _first_moment, _second_moment = "mean", "mean_sq"

    def __init__(self, input_shape, output_shape, norm_axes=1, beta=0.99999, epsilon=1e-5,
                 device=torch.device("cpu")):
        nn.Module.__init__(self)
        self.beta, self.epsilon, self.norm_axes = beta, epsilon, norm_axes
        self.input_shape, self.output_shape = input_shape, output_shape
        self.tpdv = dict(dtype=torch.float32, device=device)
        # parameters are drawn on the host (CPU generator, like every other layer here) and moved
        # afterwards, so a seed gives the same initial head whatever the device
        self.weight = nn.Parameter(torch.empty(output_shape, input_shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(output_shape, dtype=torch.float32))
        self.register_buffer("stddev", torch.ones(output_shape, dtype=torch.float32))
        for name, shape in (("mean", output_shape), ("mean_sq", output_shape), ("debiasing_term", ())):
            self.register_buffer(name, torch.zeros(shape, dtype=torch.float32))
        self.reset_parameters()
        self.to(device)

    def reset_parameters(self):
        """nn.Linear's default initialisation of the head, zero statistics."""
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(nn.init._calculate_fan_in_and_fan_out(self.weight)[0])
        nn.init.uniform_(self.bias, -bound, bound)
        self.zero_moments()

    def forward(self, input_vector):
        return F.linear(self._as_tensor(input_vector), self.weight, self.bias)

    def debiased_mean_var(self):
        return self._mean_var()

    @torch.no_grad()
    def update(self, input_vector, batch_moments=None):
        """Move the statistics, then rescale the head so that its de-normalised output is unchanged:
        w' = w sigma / sigma',  b' = (sigma b + mu - mu') / sigma'."""
        old_mean, old_var = self._mean_var()
        old_sigma = torch.sqrt(old_var)
        self._fold_in(input_vector, batch_moments, self.beta)
        self.stddev.copy_((self.mean_sq - self.mean ** 2).sqrt().clamp(min=1e-4))
        new_mean, new_var = self._mean_var()
        new_sigma = torch.sqrt(new_var)
        # Rebind .data instead of writing in place: the forward pass of the current minibatch has
        # already saved the old weights for its backward (update() runs between forward and
        # backward, r_mappo.py:65), and those saved tensors must stay untouched.
        self.weight.data = self.weight.data * (old_sigma / new_sigma).unsqueeze(-1)
        self.bias.data = (old_sigma * self.bias.data + old_mean - new_mean) / new_sigma
