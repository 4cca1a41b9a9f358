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
    """y = LayerNorm(act(x)) over the last dimension; ``x`` is saved as the pre-activation and the
    backward recomputes act(x), so the activation output is never materialised."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act=ACT_NONE):
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
        if M > 0:
            _native.check(lib.mappo_act_layernorm_fwd(x2.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                      mean.data_ptr(), rstd.data_ptr(), M, D, float(eps),
                                                      int(act), _native.stream_of(x.device)),
                          "mappo_act_layernorm_fwd")
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.x_shape = x.shape
        ctx.act = int(act)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        from onpolicy import _native
        lib = _native.lib()
        x2, w, mean, rstd = ctx.saved_tensors
        M, D = x2.shape
        dy2 = dy.contiguous().view(M, D)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dw = torch.empty(D, dtype=torch.float32, device=x2.device)
        db = torch.empty(D, dtype=torch.float32, device=x2.device)
        partials = torch.empty(2 * lib.mappo_layernorm_max_blocks() * D, dtype=torch.float32, device=x2.device)
        _native.check(lib.mappo_act_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  w.data_ptr(), None if dx is None else dx.data_ptr(),
                                                  dw.data_ptr(), db.data_ptr(), partials.data_ptr(), M, D,
                                                  ctx.act, _native.stream_of(x2.device)),
                      "mappo_act_layernorm_bwd")
        return (None if dx is None else dx.view(ctx.x_shape)), dw, db, None, None


class FusedLayerNorm(nn.LayerNorm):
    """nn.LayerNorm(D) whose forward / backward run the HIP streaming kernels on the GPU."""

    def _fusable(self, x):
        return len(self.normalized_shape) == 1 and self.elementwise_affine and self.bias is not None \
            and x.numel() > 0 and supported(x, self.normalized_shape[-1])

    def forward(self, x):
        if self._fusable(x):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps, ACT_NONE)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)

    def forward_act(self, z, act_module):
        """LayerNorm(act(z)) with the activation fused into the kernels when possible."""
        kind = ACT_TANH if isinstance(act_module, nn.Tanh) else ACT_RELU if isinstance(act_module, nn.ReLU) else None
        if kind is not None and self._fusable(z):
            return _LayerNormFn.apply(z, self.weight, self.bias, self.eps, kind)
        return self.forward(act_module(z))


class DenseBlock(nn.Sequential):
    """``Sequential(Linear, act, LayerNorm)`` -- same children indices / state_dict keys as the
    reference's blocks (mlp.py:17-22) -- evaluated as Linear followed by ONE fused act+LayerNorm pass."""

    def forward(self, x):
        linear, act, norm = self[0], self[1], self[2]
        if isinstance(norm, FusedLayerNorm):
            return norm.forward_act(linear(x), act)
        return norm(act(linear(x)))
