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

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class PopArt(nn.Module):
    def __init__(self, input_shape, output_shape, norm_axes=1, beta=0.99999, epsilon=1e-5,
                 device=torch.device("cpu")):
        super(PopArt, self).__init__()
        self.beta = beta
        self.epsilon = epsilon
        self.norm_axes = norm_axes
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.input_shape = input_shape
        self.output_shape = output_shape
        # parameters are drawn on the host (CPU generator, like every other layer here) and moved
        # afterwards, so a seed gives the same initial head whatever the device
        f32 = dict(dtype=torch.float32)
        self.weight = nn.Parameter(torch.empty(output_shape, input_shape, **f32))
        self.bias = nn.Parameter(torch.empty(output_shape, **f32))
        self.register_buffer("stddev", torch.ones(output_shape, **f32))
        self.register_buffer("mean", torch.zeros(output_shape, **f32))
        self.register_buffer("mean_sq", torch.zeros(output_shape, **f32))
        self.register_buffer("debiasing_term", torch.tensor(0.0, **f32))
        self.reset_parameters()
        self.to(device)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(self.bias, -bound, bound)
        self.mean.zero_()
        self.mean_sq.zero_()
        self.debiasing_term.zero_()

    def _as_tensor(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(dtype=torch.float32, device=self.weight.device)   # follows the module when it is moved

    def forward(self, input_vector):
        return F.linear(self._as_tensor(input_vector), self.weight, self.bias)

    @torch.no_grad()
    def update(self, input_vector, batch_moments=None):
        old_mean, old_var = self.debiased_mean_var()
        old_stddev = torch.sqrt(old_var)
        if batch_moments is None:
            x = self._as_tensor(input_vector)
            axes = tuple(range(self.norm_axes))
            batch_mean = x.mean(dim=axes)
            batch_sq_mean = (x ** 2).mean(dim=axes)
        else:
            batch_mean, batch_sq_mean = batch_moments
        self.mean.mul_(self.beta).add_(batch_mean * (1.0 - self.beta))
        self.mean_sq.mul_(self.beta).add_(batch_sq_mean * (1.0 - self.beta))
        self.debiasing_term.mul_(self.beta).add_(1.0 * (1.0 - self.beta))
        self.stddev.copy_((self.mean_sq - self.mean ** 2).sqrt().clamp(min=1e-4))
        new_mean, new_var = self.debiased_mean_var()
        new_stddev = torch.sqrt(new_var)
        # Rebind .data instead of writing in place: the forward pass of the current minibatch has
        # already saved the old weights for its backward (update() runs between forward and
        # backward, r_mappo.py:65), and those saved tensors must stay untouched.
        self.weight.data = self.weight.data * (old_stddev / new_stddev).unsqueeze(-1)
        self.bias.data = (old_stddev * self.bias.data + old_mean - new_mean) / new_stddev

    def debiased_mean_var(self):
        debias = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.mean / debias
        mean_sq = self.mean_sq / debias
        var = (mean_sq - mean ** 2).clamp(min=1e-2)
        return mean, var

    def denorm_scalars(self):
        mean, var = self.debiased_mean_var()
        return torch.stack([torch.sqrt(var).reshape(()), mean.reshape(())])

    def normalize(self, input_vector):
        x = self._as_tensor(input_vector)
        mean, var = self.debiased_mean_var()
        lead = (None,) * self.norm_axes
        return (x - mean[lead]) / torch.sqrt(var)[lead]

    def denormalize(self, input_vector):
        as_numpy = isinstance(input_vector, np.ndarray)
        x = self._as_tensor(input_vector)
        mean, var = self.debiased_mean_var()
        lead = (None,) * self.norm_axes
        out = x * torch.sqrt(var)[lead] + mean[lead]
        return out.cpu().numpy() if as_numpy else out
