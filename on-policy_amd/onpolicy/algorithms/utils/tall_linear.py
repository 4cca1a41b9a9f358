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


class _Linear512Fn(torch.autograd.Function):
    """``F.linear`` for 512 output features on a tall HIP matrix through K15 (``mappo_linear512_*``, csrc/mappo_lin_impl.h): the
    Linear layers of the hidden-512 trunks (reference onpolicy/algorithms/utils/mlp.py:17-22 at --hidden_size 512:
    scripts/train_hanabi_forward.sh:15-17) in six-term bf16 arithmetic -- float32 in and out, every product from six bf16 x bf16
    terms of the operands' exact three-way splits on the bf16 matrix cores, float32 accumulation (include/mappo_hip.h
    MAPPO_ARITH_SIX_TERM).  Forward ``x W^T (+ b)``, weight gradient ``dy^T x`` (row ranges summed in a fixed order), input
    gradient ``dy W`` of a 512 -> 512 layer through the forward kernel on the planes of ``W^T``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        dev = x.device
        stream = _native.stream_of(dev)
        rows, ldx = x.shape
        w = weight.detach().contiguous()
        K = int(w.shape[1])
        planes = torch.empty(lib.mappo_linear512_planes_floats(K), dtype=torch.float32, device=dev)
        _native.check(lib.mappo_linear512_prepare(p(w), K, K, 0, p(planes), stream), "mappo_linear512_prepare")
        y = torch.empty((rows, 512), dtype=torch.float32, device=dev)
        b = None if bias is None else bias.detach().contiguous()
        if b is not None and b.data_ptr() % 16:     # (a bias that is a view at an odd offset: the kernel reads 16-byte vectors)
            b = b.clone()
        from . import fused_mlp         # (bench.py: an event pair + the algorithmic FLOPs / bytes of the launch while profiling)
        with fused_mlp._Timed("mappo_linear512_forward", 2.0 * rows * K * 512, 4.0 * rows * (ldx + 512)):
            _native.check(lib.mappo_linear512_forward(p(x), rows, K, int(ldx), p(planes), p(b), p(y), stream),
                          "mappo_linear512_forward")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dw = _linear512_grads(ctx.needs_input_grad[0], ctx.needs_input_grad[1], dy, x, w)
        db = column_sums(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class _Linear512NormFn(torch.autograd.Function):
    """``LayerNorm(relu(x W^T + b))`` of a hidden-512 block (reference onpolicy/algorithms/utils/mlp.py:17-22:
    ``Sequential(Linear, ReLU, LayerNorm)``) with the bias add, the activation and the LayerNorm in the EPILOGUE of K15's forward
    (``mappo_linear512_forward_norm``): a workgroup of that kernel holds all 512 features of its 128 rows in registers, so the
    row statistics cost two cross-lane exchanges and the [rows, 512] pre-activation is written once and never read back by a
    LayerNorm launch of its own (K6's forward read it: one of the two streams of that launch).  The backward is the unfused
    pair: K6's backward on the saved pre-activation (bias included: ``pre_bias`` is a zero vector there, its gradient is the
    Linear's bias gradient), then K15's input gradient and weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, ln_weight, ln_bias, eps):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        dev = x.device
        stream = _native.stream_of(dev)
        rows, ldx = x.shape
        w = weight.detach().contiguous()
        K = int(w.shape[1])
        planes = torch.empty(lib.mappo_linear512_planes_floats(K), dtype=torch.float32, device=dev)
        _native.check(lib.mappo_linear512_prepare(p(w), K, K, 0, p(planes), stream), "mappo_linear512_prepare")
        b = _aligned16(bias.detach().contiguous())
        g = _aligned16(ln_weight.detach().contiguous())
        be = _aligned16(ln_bias.detach().contiguous())
        z = torch.empty((rows, 512), dtype=torch.float32, device=dev)
        y = torch.empty((rows, 512), dtype=torch.float32, device=dev)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty(rows, dtype=torch.float32, device=dev)
        from . import fused_mlp
        with fused_mlp._Timed("mappo_linear512_forward", 2.0 * rows * K * 512, 4.0 * rows * (ldx + 1024 + 2)):
            _native.check(lib.mappo_linear512_forward_norm(p(x), rows, K, int(ldx), p(planes), p(b), p(g), p(be), float(eps),
                                                           _ACT_RELU, p(z), p(y), p(mean), p(rstd), stream),
                          "mappo_linear512_forward_norm")
        ctx.save_for_backward(x, w, z, mean, rstd, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        x, w, z, mean, rstd, g = ctx.saved_tensors
        dev = x.device
        stream = _native.stream_of(dev)
        rows = x.shape[0]
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        dg = torch.empty(512, dtype=torch.float32, device=dev)
        dbe = torch.empty(512, dtype=torch.float32, device=dev)
        db = torch.empty(512, dtype=torch.float32, device=dev)
        zero = torch.zeros(512, dtype=torch.float32, device=dev)      # (the bias is already inside z)
        partials = torch.empty(3 * lib.mappo_layernorm_max_blocks() * 512, dtype=torch.float32, device=dev)
        _native.check(lib.mappo_bias_act_layernorm_bwd(p(dy), p(z), p(zero), p(mean), p(rstd), p(g), p(dz), p(dg), p(dbe), p(db),
                                                       p(partials), rows, 512, _ACT_RELU, stream),
                      "mappo_bias_act_layernorm_bwd")
        dx, dw = _linear512_grads(ctx.needs_input_grad[0], ctx.needs_input_grad[1], dz, x, w)
        return dx, dw, db, dg, dbe, None


_ACT_RELU = 2       # (fused_norm.ACT_RELU, include/mappo_hip.h)


def _aligned16(t):
    """The kernels read these vectors in 16-byte pieces: a view at an odd offset is copied."""
    return t.clone() if t.data_ptr() % 16 else t


def _linear512_grads(want_dx, want_dw, dy, x, w):
    """Input and weight gradient of ``y = x W^T`` through K15 (dX = dY W: the forward kernel on the planes of W^T for a
    512 -> 512 layer, the library GEMM otherwise)."""
    from onpolicy import _native
    lib, p = _native.lib(), _native.ptr
    dev = x.device
    stream = _native.stream_of(dev)
    rows, ldx = x.shape
    K = int(w.shape[1])
    dx = dw = None
    if want_dx:
        if K == 512 and ldx == 512:     # dX = dY W: the forward kernel on the planes of W^T
            planes = torch.empty(lib.mappo_linear512_planes_floats(512), dtype=torch.float32, device=dev)
            _native.check(lib.mappo_linear512_prepare(p(w), 512, 512, 1, p(planes), stream), "mappo_linear512_prepare")
            dx = torch.empty((rows, 512), dtype=torch.float32, device=dev)
            _native.check(lib.mappo_linear512_forward(p(dy), rows, 512, 512, p(planes), None, p(dx), stream),
                          "mappo_linear512_forward")
        else:
            dx = dy @ w
            if ldx != K:
                dx = F.pad(dx, (0, ldx - K))
    if want_dw:
        dw = torch.empty((512, K), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.mappo_linear512_wgrad_workspace_floats(K), dtype=torch.float32, device=dev)
        from . import fused_mlp
        with fused_mlp._Timed("mappo_linear512_wgrad", 2.0 * rows * K * 512, 4.0 * rows * (ldx + 512)):
            _native.check(lib.mappo_linear512_wgrad(p(dy), p(x), rows, K, int(ldx), p(dw), p(ws), stream),
                          "mappo_linear512_wgrad")
    return dx, dw


def linear512_ok(x, weight, arith=None):
    """Whether ``tall_linear`` sends this product through K15: 512 output features, a tall contiguous float32 HIP matrix whose
    rows hold exactly the weight's columns -- or those columns zero-padded to whole float4s, the one wider layout this
    repo makes (``fused_mlp.standardize_rows``); any other width mismatch goes to ``F.linear``, which raises -- the six-term arithmetic selected
    (``arith``: _native.ARITH_* or None = the process default, MAPPO_MATRIX_ARITHMETIC) and MAPPO_LINEAR512 not 0.
    (MAPPO_LINEAR512_MIN_ROWS: tests send small fixtures through the kernels; below 65 536 rows the library GEMM is as good.)"""
    import os
    from onpolicy import _native
    if os.environ.get("MAPPO_LINEAR512", "1") == "0" or not (torch.is_tensor(x) and x.is_cuda):
        return False
    a = _native.default_arith() if arith is None else _native.arith_code(arith)
    return a == _native.ARITH_SIX_TERM and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() \
        and x.shape[0] >= int(os.environ.get("MAPPO_LINEAR512_MIN_ROWS", _MIN_ROWS)) and weight.dim() == 2 and weight.shape[0] == 512 and weight.dtype == torch.float32 \
        and weight.shape[1] >= 4 and x.shape[1] in (weight.shape[1], (weight.shape[1] + 3) // 4 * 4) and weight.is_cuda


def tall_linear(x, weight, bias, arith=None):
    """F.linear with the split-K backward when ``x`` is a tall HIP matrix; 512-wide layers in six-term arithmetic through
    K15 (``_Linear512Fn``)."""
    if linear512_ok(x, weight, arith):
        return _Linear512Fn.apply(x, weight, bias)
    if x.dim() == 2 and x.shape[1] == (weight.shape[1] + 3) // 4 * 4 != weight.shape[1]:
        x = x[:, :weight.shape[1]]      # the zero padding of a standardised copy (shared_buffer._whole_batch_views): a strided view
    if x.dim() == 2 and x.is_cuda and x.shape[0] >= _MIN_ROWS and torch.is_grad_enabled() \
            and x.is_contiguous():
        return _TallLinearFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def linear512_relu_norm(x, weight, bias, norm):
    """``norm(relu(x W^T + b))`` in ONE K15 launch (``_Linear512NormFn``); callers check ``linear512_norm_ok`` first."""
    return _Linear512NormFn.apply(x, weight, bias, norm.weight, norm.bias, float(norm.eps))


def linear512_norm_ok(x, weight, bias, act_module, norm, arith=None):
    """Whether a block ``Sequential(Linear, act, LayerNorm)`` takes the fused epilogue: K15 takes the product, the activation is
    ReLU (Hanabi's, the reference's default ``--use_ReLU``), the LayerNorm is an affine one over the 512 features, gradients
    are on (the rollout's evaluations keep the two-launch route: nothing to save there) and MAPPO_LINEAR512_NORM=1.
    OPT-IN: on the MI355X the block form is 3.5 % SLOWER on the Hanabi-shaped step (profiles/r06_ab_lin512_block_epilogue.json:
    the epilogue's LayerNorm weight / bias loads queue behind the tile's own stores in the in-order vector-memory counter)."""
    import os
    return os.environ.get("MAPPO_LINEAR512_NORM", "0") == "1" and bias is not None and isinstance(act_module, nn.ReLU) \
        and isinstance(norm, nn.LayerNorm) and tuple(norm.normalized_shape) == (512,) and norm.elementwise_affine \
        and norm.bias is not None and torch.is_grad_enabled() and linear512_ok(x, weight, arith)


class TallLinear(nn.Linear):
    # arithmetic of the layer's product where K15 takes it (_native.ARITH_* or None = the process default)
    matrix_arithmetic = None

    def forward(self, x):
        return tall_linear(x, self.weight, self.bias, self.matrix_arithmetic)
