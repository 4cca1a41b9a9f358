"""Image trunk for rank-3 observations [C, W, H] with pixel values 0..255:
``obs / 255 -> conv(k x k, hidden/2 filters) -> act -> flatten -> linear(hidden) -> act -> linear(hidden) -> act``.

Module paths (``cnn.cnn.0`` convolution, ``cnn.cnn.3`` / ``cnn.cnn.5`` linears) and the order in which the layers
are created match the reference's onpolicy/algorithms/utils/cnn.py (CNNLayer :12, CNNBase :46), so checkpoints load
and a seed gives the same weights.  Not on the benchmarked path (none of BASELINE.json's configs has image
observations); pinned by tests/golden/space_cases.npz.
"""
import torch.nn as nn

from .util import init

_PIXEL_RANGE = 255.0


def _conv_output_cells(width, height, kernel_size, stride):
    """Spatial size after one unpadded convolution, as the reference sizes its first linear layer (cnn.py:34)."""
    return (width - kernel_size + stride) * (height - kernel_size + stride)


class CNNLayer(nn.Module):
    def __init__(self, obs_shape, hidden_size, use_orthogonal, use_ReLU, kernel_size=3, stride=1):
        super(CNNLayer, self).__init__()
        nonlinearity = 'relu' if use_ReLU else 'tanh'
        weight_init = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_

        def prepared(layer):
            return init(layer, weight_init, lambda bias: nn.init.constant_(bias, 0),
                        gain=nn.init.calculate_gain(nonlinearity))

        def activation():
            return nn.ReLU() if use_ReLU else nn.Tanh()

        channels, width, height = obs_shape[:3]
        filters = hidden_size // 2
        stack = [prepared(nn.Conv2d(channels, filters, kernel_size=kernel_size, stride=stride)), activation(),
                 nn.Flatten(start_dim=1)]
        fan_in = filters * _conv_output_cells(width, height, kernel_size, stride)
        for _ in range(2):
            stack += [prepared(nn.Linear(fan_in, hidden_size)), activation()]
            fan_in = hidden_size
        self.cnn = nn.Sequential(*stack)

    def forward(self, x):
        return self.cnn(x / _PIXEL_RANGE)


class CNNBase(nn.Module):
    """What the actor / critic instantiate for image observations: ``CNNBase(args, obs_shape)``."""

    def __init__(self, args, obs_shape):
        super(CNNBase, self).__init__()
        self.hidden_size = args.hidden_size
        self._use_orthogonal, self._use_ReLU = args.use_orthogonal, args.use_ReLU
        self.cnn = CNNLayer(obs_shape, args.hidden_size, args.use_orthogonal, args.use_ReLU)

    def forward(self, x):
        return self.cnn(x)
