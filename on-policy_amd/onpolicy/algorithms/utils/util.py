"""Module-construction helpers (reference onpolicy/algorithms/utils/util.py: init :7,
get_clones :13, check :16)."""
from copy import deepcopy

import numpy as np
import torch
from torch import nn


def init(module, weight_init, bias_init, gain=1):
    """Re-initialise a freshly constructed layer in place and return it.  The layer's own
    constructor has already consumed its default-init random draws, which keeps the global RNG
    stream -- and therefore every weight under a given seed -- identical to the reference."""
    with torch.no_grad():
        weight_init(module.weight, gain=gain)
        bias = getattr(module, "bias", None)
        if bias is not None:
            bias_init(bias)
    return module


def get_clones(module, N):
    """N independent deep copies as a ModuleList."""
    return nn.ModuleList(deepcopy(module) for _ in range(N))


def check(value):
    """numpy array -> tensor sharing its memory; everything else (tensors, None) passes through."""
    if isinstance(value, np.ndarray):
        return torch.from_numpy(value)
    return value
