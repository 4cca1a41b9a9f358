"""Feature trunk of actor and critic: [LayerNorm] -> (Linear -> act -> LayerNorm) x (1 + layer_N).

Parameter names (``feature_norm``, ``mlp.fc1.{0,2}``, ``mlp.fc2.<i>.{0,2}``) and the order in which
layers are constructed match the reference's onpolicy/algorithms/utils/mlp.py (MLPLayer :6,
MLPBase :33), so reference checkpoints load and a given seed yields the same weights.
"""
import torch.nn as nn

from . import fused_mlp
from .fused_norm import FusedLayerNorm, DenseBlock
from .tall_linear import TallLinear, tall_linear, linear512_norm_ok, linear512_relu_norm
from .util import init


def _weight_init(use_orthogonal):
    return nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_


def _zero_bias(b):
    return nn.init.constant_(b, 0)


def _block(in_dim, out_dim, use_orthogonal, use_ReLU, act):
    gain = nn.init.calculate_gain('relu' if use_ReLU else 'tanh')
    linear = init(TallLinear(in_dim, out_dim), _weight_init(use_orthogonal), _zero_bias, gain=gain)
    return DenseBlock(linear, act, FusedLayerNorm(out_dim))


class MLPLayer(nn.Module):
    def __init__(self, input_dim, hidden_size, layer_N, use_orthogonal, use_ReLU):
        super(MLPLayer, self).__init__()
        self._layer_N = layer_N
        act = nn.ReLU() if use_ReLU else nn.Tanh()
        self.fc1 = _block(input_dim, hidden_size, use_orthogonal, use_ReLU, act)
        self.fc2 = nn.ModuleList(
            [_block(hidden_size, hidden_size, use_orthogonal, use_ReLU, act) for _ in range(layer_N)])

    def forward(self, x):
        x = self.fc1(x)
        for layer in self.fc2:
            x = layer(x)
        return x


class MLPBase(nn.Module):
    def __init__(self, args, obs_shape, cat_self=True, attn_internal=False):
        super(MLPBase, self).__init__()
        self._use_feature_normalization = args.use_feature_normalization
        self._use_orthogonal = args.use_orthogonal
        self._use_ReLU = args.use_ReLU
        self._stacked_frames = args.stacked_frames
        self._layer_N = args.layer_N
        self.hidden_size = args.hidden_size
        # arithmetic of K9's matrix products: --matrix_arithmetic (None = the process default), a per-network choice
        self.matrix_arithmetic = getattr(args, "matrix_arithmetic", None)
        obs_dim = obs_shape[0]
        if self._use_feature_normalization:
            self.feature_norm = FusedLayerNorm(obs_dim)
        self.mlp = MLPLayer(obs_dim, self.hidden_size, self._layer_N, self._use_orthogonal, self._use_ReLU)
        if self.matrix_arithmetic is not None:       # (the Linear layers carry the choice for K15, hidden size 512)
            fused_mlp.set_matrix_arithmetic(self.mlp, self.matrix_arithmetic)

    def fuses(self, x):
        """Whether ``forward(x)`` takes the fused-kernel route for this input."""
        return isinstance(x, fused_mlp.RowSource) and fused_mlp.trunk_supported(self) \
            and x.standardized == bool(self._use_feature_normalization) and x.shape[1] >= 4

    def can_fold_input_norm(self):
        """True if ``forward(x, standardized=True)`` is available: the input LayerNorm exists and its
        eps is the one the standardising gather uses."""
        return bool(self._use_feature_normalization) and abs(self.feature_norm.eps - 1e-5) < 1e-12

    def forward(self, x, standardized=False, head=None):
        """``x`` may be a ``fused_mlp.RowSource`` (rows of the rollout buffer named by a sampler minibatch, not yet
        gathered): the whole trunk -- and ``head``, an output Linear, when given -- then runs through the fused hidden-64
        kernels (K9) if this trunk qualifies; otherwise the rows are gathered and evaluated as below.  ``head`` is only
        applied on the fused route (callers check ``fuses(x)``).

        ``standardized=True``: ``x`` already holds (x - mean) / sqrt(var + eps) per row (the sampler
        computed it while gathering).  The LayerNorm's affine half is then folded into the first
        Linear, LN(x) W^T + b = xhat (W * gamma)^T + (b + W beta): the same function of the same
        parameters (autograd reaches gamma / beta through the two tiny products), without ever
        materialising the normalised [rows, D] input or its gradient."""
        if isinstance(x, fused_mlp.RowSource):
            if self.fuses(x):
                return fused_mlp.trunk_forward(self, x, head)
            standardized = x.standardized
            x = x.materialize()
        if not self._use_feature_normalization:
            return self.mlp(x)
        if not standardized:
            return self.mlp(self.feature_norm(x))
        first = self.mlp.fc1
        linear, act, norm = first[0], first[1], first[2]
        w_eff = linear.weight * self.feature_norm.weight
        b_eff = linear.bias + linear.weight @ self.feature_norm.bias
        if isinstance(norm, FusedLayerNorm) and norm._fusable(x) \
                and linear512_norm_ok(x, w_eff, b_eff, act, norm, self.matrix_arithmetic):
            h = linear512_relu_norm(x, w_eff, b_eff, norm)      # hidden 512: bias, ReLU and LayerNorm in K15's epilogue
        elif isinstance(norm, FusedLayerNorm) and norm.fuses_bias(x, act, linear.out_features):
            h = norm.forward_act(tall_linear(x, w_eff, None, self.matrix_arithmetic), act, pre_bias=b_eff)   # bias add in the LN kernel
        else:
            h = norm.forward_act(tall_linear(x, w_eff, b_eff, self.matrix_arithmetic), act)
        for layer in self.mlp.fc2:
            h = layer(h)
        return h
