"""Loss / schedule / space-shape helpers shared by the buffer and the trainer.

Same names and semantics as the reference's onpolicy/utils/util.py (check :6, get_gard_norm :9,
update_linear_schedule :17, huber_loss :23, mse_loss :28, get_shape_from_obs_space :31,
get_shape_from_act_space :40) -- callers in the reference scripts import them by these names.
"""
import math

import numpy as np
import torch


def check(value):
    """ndarray -> tensor (zero copy); tensors pass through untouched, which is what lets device
    tensors from the HBM buffer flow into the networks without a host round trip."""
    return torch.from_numpy(value) if isinstance(value, np.ndarray) else value


def get_gard_norm(params):
    """Global L2 norm of the gradients (name kept from the reference, typo included)."""
    total = 0.0
    for p in params:
        if p.grad is not None:
            total = total + p.grad.norm() ** 2
    # tensor in, tensor out: no host sync per minibatch (the trainer reads its logged scalars once per train())
    return torch.sqrt(total) if torch.is_tensor(total) else math.sqrt(total)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """lr = initial_lr * (1 - epoch / total)."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for group in optimizer.param_groups:
        group['lr'] = lr


def huber_loss(e, d):
    """0.5 e^2 inside |e| <= d, d (|e| - d/2) outside -- written with the reference's two
    indicator masks so the float32 result is the same expression tree."""
    abs_e = abs(e)
    inside = (abs_e <= d).float()
    outside = (abs_e > d).float()
    return inside * e ** 2 / 2 + outside * d * (abs_e - d / 2)


def mse_loss(e):
    return e ** 2 / 2


def _space_kind(space):
    return space.__class__.__name__


def get_shape_from_obs_space(obs_space):
    """Spaces are recognised by class NAME ('Box' / 'list'), so gym, gymnasium or duck-typed
    stand-ins all work."""
    kind = _space_kind(obs_space)
    if kind == 'Box':
        return obs_space.shape
    if kind == 'list':
        return obs_space
    raise NotImplementedError("observation space %s" % kind)


def get_shape_from_act_space(act_space):
    kind = _space_kind(act_space)
    if kind == 'Discrete':
        return 1
    if kind == 'MultiDiscrete':
        return act_space.shape
    if kind in ('Box', 'MultiBinary'):
        return act_space.shape[0]
    return act_space[0].shape[0] + 1  # mixed [Box, Discrete] spaces ("agar")


def tile_images(img_nhwc):
    """N images -> one near-square mosaic (rendering helper of the vector envs)."""
    imgs = np.asarray(img_nhwc)
    n, h, w, c = imgs.shape
    rows = int(np.ceil(np.sqrt(n)))
    cols = int(np.ceil(float(n) / rows))
    pad = np.zeros((rows * cols - n, h, w, c), dtype=imgs.dtype)
    grid = np.concatenate([imgs, pad], 0).reshape(rows, cols, h, w, c)
    return grid.transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)
