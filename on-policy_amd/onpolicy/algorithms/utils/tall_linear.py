"""``nn.Linear`` whose weight / bias gradients are computed with a split-K batched GEMM when the
input is a tall matrix (10^5 .. 10^7 rows by a few dozen features -- every Linear of the MAPPO
networks during the update, reference onpolicy/algorithms/utils/mlp.py:17-22, act.py / distributions.py
heads, r_actor_critic.py v_out).

Why: ``dW = dY^T @ X`` reduces over the row dimension.  For [2.6 M, 64]^T x [2.6 M, 64] the BLAS
library launches a handful of workgroups that each walk millions of rows (3.3 ms, 0.4 TB/s on an
MI355X; profiles/r01_bench_ns_kernel_stats.csv), and ``dY.sum(0)`` for the bias is as slow.  Cutting
the rows into S slabs turns it into a batched GEMM [S, N, R] x [S, R, K] with thousands of
independent tiles followed by a small sum over S -- same maths, float32 summation order aside.
Everything is ordinary PyTorch (bmm / sum); parameters, init and state_dict are those of nn.Linear.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_ROWS = 1 << 16      # below this the library GEMM is fine
_SLAB_ROWS = 4096        # rows per slab (R)


def splitk_weight_grad(dy, x):
    """dy^T @ x for tall [M, N], [M, K] matrices as a batched GEMM over row slabs + a sum over slabs."""
    M, N = dy.shape
    K = x.shape[1]
    S = M // _SLAB_ROWS
    main = S * _SLAB_ROWS
    if S == 0:
        return dy.t() @ x
    dw = torch.bmm(dy[:main].view(S, _SLAB_ROWS, N).transpose(1, 2), x[:main].view(S, _SLAB_ROWS, K)).sum(0)
    if main < M:
        dw = dw + dy[main:].t() @ x[main:]
    return dw


def column_sums(dy):
    """dy.sum(0) for a tall [M, N] matrix in two stages (slab sums, then over slabs)."""
    M, N = dy.shape
    S = M // _SLAB_ROWS
    main = S * _SLAB_ROWS
    if S == 0:
        return dy.sum(0)
    db = dy[:main].unflatten(0, (S, _SLAB_ROWS)).sum(1).sum(0)      # also for column slices (strided rows)
    if main < M:
        db = db + dy[main:].sum(0)
    return db


class _TallLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        # [S, N, R] x [S, R, K] -> [S, N, K] -> sum over slabs
        dw = splitk_weight_grad(dy, x) if ctx.needs_input_grad[1] else None
        db = column_sums(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def tall_linear(x, weight, bias):
    """F.linear with the split-K backward when ``x`` is a tall HIP matrix."""
    if x.dim() == 2 and x.is_cuda and x.shape[0] >= _MIN_ROWS and torch.is_grad_enabled() \
            and x.is_contiguous():
        return _TallLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class TallLinear(nn.Linear):
    def forward(self, x):
        return tall_linear(x, self.weight, self.bias)
