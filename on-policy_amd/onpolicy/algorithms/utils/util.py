"""Module-construction helpers (reference onpolicy/algorithms/utils/util.py: init :7,
get_clones :13, check :16)."""
import copy

import numpy as np
import torch
import torch.nn as nn


def init(module, weight_init, bias_init, gain=1):
    """Re-initialise a freshly constructed layer in place and return it.  The layer's own
    constructor has already consumed its default-init random draws, which keeps the global RNG
    stream -- and therefore every weight under a given seed -- identical to the reference."""
    weight_init(module.weight.data, gain=gain)
    if module.bias is not None:
        bias_init(module.bias.data)
    return module


def get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def check(value):
    return torch.from_numpy(value) if isinstance(value, np.ndarray) else value
