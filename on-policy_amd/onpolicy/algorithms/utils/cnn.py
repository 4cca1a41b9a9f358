"""Image trunk (obs of rank 3): x/255 -> Conv2d -> act -> flatten -> Linear -> act -> Linear -> act.
Names / construction order follow the reference's onpolicy/algorithms/utils/cnn.py (CNNLayer :12,
CNNBase :46) for checkpoint and seed compatibility.  Not on the benchmarked path."""
import torch.nn as nn

from .util import init


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class CNNLayer(nn.Module):
    def __init__(self, obs_shape, hidden_size, use_orthogonal, use_ReLU, kernel_size=3, stride=1):
        super(CNNLayer, self).__init__()
        act = nn.ReLU() if use_ReLU else nn.Tanh()
        w_init = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
        gain = nn.init.calculate_gain('relu' if use_ReLU else 'tanh')

        def make(m):
            return init(m, w_init, lambda b: nn.init.constant_(b, 0), gain=gain)

        channels, width, height = obs_shape[0], obs_shape[1], obs_shape[2]
        conv_out = hidden_size // 2 * (width - kernel_size + stride) * (height - kernel_size + stride)
        self.cnn = nn.Sequential(
            make(nn.Conv2d(in_channels=channels, out_channels=hidden_size // 2, kernel_size=kernel_size,
                           stride=stride)),
            act, Flatten(),
            make(nn.Linear(conv_out, hidden_size)), act,
            make(nn.Linear(hidden_size, hidden_size)), act)

    def forward(self, x):
        return self.cnn(x / 255.0)


class CNNBase(nn.Module):
    def __init__(self, args, obs_shape):
        super(CNNBase, self).__init__()
        self._use_orthogonal = args.use_orthogonal
        self._use_ReLU = args.use_ReLU
        self.hidden_size = args.hidden_size
        self.cnn = CNNLayer(obs_shape, self.hidden_size, self._use_orthogonal, self._use_ReLU)

    def forward(self, x):
        return self.cnn(x)
