"""LayerNorm module backed by the streaming HIP kernels of libmappo_hip.so (mappo_layernorm_fwd /
_bwd, on-policy_amd/csrc/mappo_norm.hip) for HIP float32 tensors.

Same parameters, state_dict keys and maths as ``nn.LayerNorm`` over the last dimension (the only
way the reference uses it: onpolicy/algorithms/utils/mlp.py:17-22,47-53, rnn.py:22,79).  Why it
exists: for the narrow rows of these networks (D = 48 .. 512) PyTorch-ROCm's generic LayerNorm
kernels run at ~0.5 TB/s and were 54 % of the north-star update; these kernels stream at HBM speed.
Autograd still drives it (a torch.autograd.Function), CPU tensors and unsupported widths go through
``F.layer_norm`` unchanged.  Set MAPPO_FUSED_LAYERNORM=0 to use PyTorch's kernels everywhere.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_ENABLED = os.environ.get("MAPPO_FUSED_LAYERNORM", "1") != "0"


def supported(x, D):
    if not (_ENABLED and x.is_cuda and x.dtype == torch.float32):
        return False
    return (D % 4 == 0 and D <= 2048) or D <= 1536


ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2


class _LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(act(x + pre_bias)) over the last dimension; ``x`` is saved as the pre-activation and
    the backward recomputes act(.), so the activation output is never materialised.  ``pre_bias`` (optional)
    is the bias of the Linear that produced ``x``: adding it here lets the GEMM run without a bias, and its
    gradient -- the column sums of dx -- falls out of the backward kernel instead of a separate reduction
    over the [rows, D] gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act=ACT_NONE, pre_bias=None):
        from onpolicy import _native
        lib = _native.lib()
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        M = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        w = weight.contiguous()
        b = bias.contiguous()
        pre = None if pre_bias is None else pre_bias.contiguous()
        if M > 0:
            _native.check(lib.mappo_bias_act_layernorm_fwd(x2.data_ptr(), _native.ptr(pre), w.data_ptr(), b.data_ptr(),
                                                           y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D,
                                                           float(eps), int(act), _native.stream_of(x.device)),
                          "mappo_bias_act_layernorm_fwd")
        ctx.save_for_backward(x2, w, mean, rstd, pre)
        ctx.x_shape = x.shape
        ctx.act = int(act)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        from onpolicy import _native
        lib = _native.lib()
        x2, w, mean, rstd, pre = ctx.saved_tensors
        M, D = x2.shape
        dy2 = dy.contiguous().view(M, D)
        want_pre = pre is not None and ctx.needs_input_grad[5]
        dx = torch.empty_like(x2) if (ctx.needs_input_grad[0] or want_pre) else None
        dw = torch.empty(D, dtype=torch.float32, device=x2.device)
        db = torch.empty(D, dtype=torch.float32, device=x2.device)
        dpre = torch.empty(D, dtype=torch.float32, device=x2.device) if want_pre else None
        partials = torch.empty((3 if want_pre else 2) * lib.mappo_layernorm_max_blocks() * D, dtype=torch.float32,
                               device=x2.device)
        _native.check(lib.mappo_bias_act_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), _native.ptr(pre), mean.data_ptr(),
                                                       rstd.data_ptr(), w.data_ptr(), _native.ptr(dx), dw.data_ptr(),
                                                       db.data_ptr(), _native.ptr(dpre), partials.data_ptr(), M, D,
                                                       ctx.act, _native.stream_of(x2.device)),
                      "mappo_bias_act_layernorm_bwd")
        return (dx.view(ctx.x_shape) if ctx.needs_input_grad[0] else None), dw, db, None, None, dpre


class FusedLayerNorm(nn.LayerNorm):
    """nn.LayerNorm(D) whose forward / backward run the HIP streaming kernels on the GPU."""

    def _fusable(self, x):
        return len(self.normalized_shape) == 1 and self.elementwise_affine and self.bias is not None \
            and x.numel() > 0 and supported(x, self.normalized_shape[-1])

    def forward(self, x):
        if self._fusable(x):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps, ACT_NONE)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)

    @staticmethod
    def _kind(act_module):
        return ACT_TANH if isinstance(act_module, nn.Tanh) else ACT_RELU if isinstance(act_module, nn.ReLU) else None

    def forward_act(self, z, act_module, pre_bias=None):
        """LayerNorm(act(z + pre_bias)) with the bias add and the activation fused into the kernels when
        possible."""
        kind = self._kind(act_module)
        if kind is not None and self._fusable(z):
            return _LayerNormFn.apply(z, self.weight, self.bias, self.eps, kind, pre_bias)
        if pre_bias is not None:
            z = z + pre_bias
        return self.forward(act_module(z))

    def fuses_bias(self, x, act_module, out_dim):
        """Whether forward_act can absorb the preceding Linear's bias for rows like ``x``."""
        return self._kind(act_module) is not None and len(self.normalized_shape) == 1 and self.elementwise_affine \
            and self.bias is not None and out_dim <= 1024 and x.dim() == 2 and x.numel() > 0 \
            and supported(x, out_dim) and torch.is_grad_enabled()


class DenseBlock(nn.Sequential):
    """``Sequential(Linear, act, LayerNorm)`` -- same children indices / state_dict keys as the
    reference's blocks (mlp.py:17-22) -- evaluated as Linear followed by ONE fused act+LayerNorm pass."""

    def forward(self, x):
        linear, act, norm = self[0], self[1], self[2]
        if isinstance(norm, FusedLayerNorm):
            from .tall_linear import tall_linear, linear512_norm_ok, linear512_relu_norm
            arith = getattr(linear, "matrix_arithmetic", None)
            if norm._fusable(x) and linear512_norm_ok(x, linear.weight, linear.bias, act, norm, arith):
                return linear512_relu_norm(x, linear.weight, linear.bias, norm)     # hidden 512: ONE K15 launch (epilogue)
            if linear.bias is not None and norm.fuses_bias(x, act, linear.out_features):
                # GEMM without bias; the bias add and its gradient ride along in the act+LayerNorm kernels
                return norm.forward_act(tall_linear(x, linear.weight, None, getattr(linear, "matrix_arithmetic", None)), act,
                                        pre_bias=linear.bias)
            return norm.forward_act(linear(x), act)
        return norm(act(linear(x)))
