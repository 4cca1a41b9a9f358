"""-m gpu: K15 (mappo_linear512_*: the Linear layers of the hidden-512 trunks of BASELINE configs[4] in six-term bf16 arithmetic;
reference onpolicy/algorithms/utils/mlp.py:17-22 at --hidden_size 512, scripts/train_hanabi_forward.sh:15-17) on the MI355X:
forward / weight gradient / input gradient against float64 with the bound of tests/six_term_harness.py, Hanabi's unaligned
widths, ragged row counts, the autograd route ``tall_linear`` takes, determinism; then a hidden-512 MLPBase against its
float64 twin and against the library-GEMM route.  The reference-generated end-to-end case at these shapes is
tests/test_gpu_cfg_shapes.py::cfg5_shape (which runs through these kernels: asserted there)."""
import copy

import numpy as np
import pytest
import torch

from helpers import make_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
U = 2.0 ** -24


def _bound(S, K):
    return (16.0 + K / 6.0) * U * S


@pytest.mark.parametrize("rows,K,ldx", [(128 * 300 + 37, 1285, 1285), (128 * 300 + 37, 1288, 1288), (70000, 512, 512),
                                        (128 * 257, 1385, 1388), (66000, 40, 40)])
def test_forward_and_both_gradients_vs_float64(rows, K, ldx, monkeypatch):
    from onpolicy.algorithms.utils.tall_linear import _Linear512Fn, linear512_ok
    monkeypatch.setenv("MAPPO_LINEAR512_MIN_ROWS", "1")         # (tall_linear itself switches over at 65 536 rows)
    g = torch.Generator(device=DEV).manual_seed(rows + K)
    x = torch.zeros(rows, ldx, device=DEV)
    x[:, :K] = torch.randn(rows, K, device=DEV, generator=g) * 1.5 + 0.3
    w = (torch.randn(512, K, device=DEV, generator=g) * 0.1).requires_grad_()
    b = torch.randn(512, device=DEV, generator=g).requires_grad_()
    need_dx = K == 512
    x.requires_grad_(need_dx)
    assert linear512_ok(x, w)
    y = _Linear512Fn.apply(x, w, b)
    dy = torch.randn(rows, 512, device=DEV, generator=g) / rows ** 0.5
    y.backward(dy)
    x64, w64, b64, d64 = x.detach().double()[:, :K], w.detach().double(), b.detach().double(), dy.double()
    ref = x64 @ w64.t() + b64
    S = x64.abs() @ w64.abs().t() + b64.abs()
    assert bool(torch.isfinite(y).all())
    assert float(((y.detach().double() - ref).abs() / _bound(S, K)).max()) <= 1.0
    dw_ref = d64.t() @ x64
    S = d64.abs().t() @ x64.abs()
    err = float(((w.grad.double() - dw_ref).abs() / ((16.0 + rows / 6.0) * U * S + 1e-300)).max())
    assert err <= 1.0, err
    torch.testing.assert_close(b.grad.double(), d64.sum(0), rtol=1e-4, atol=1e-5)
    if need_dx:
        dx_ref = d64 @ w64
        S = d64.abs() @ w64.abs()
        assert float(((x.grad.double() - dx_ref).abs() / _bound(S, 512)).max()) <= 1.0


def test_deterministic_and_ragged_rows():
    from onpolicy.algorithms.utils.tall_linear import _Linear512Fn
    g = torch.Generator(device=DEV).manual_seed(3)
    rows = 128 * 600 + 5
    x = torch.randn(rows, 1285, device=DEV, generator=g)
    w = (torch.randn(512, 1285, device=DEV, generator=g) * 0.1).requires_grad_()
    dy = torch.randn(rows, 512, device=DEV, generator=g)
    outs = []
    for _ in range(2):
        w.grad = None
        y = _Linear512Fn.apply(x, w, None)
        y.backward(dy)
        outs.append((y.detach().clone(), w.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the rows past the last full tile were written, nothing beyond them touched
    torch.testing.assert_close(outs[0][0][-5:].double(), x[-5:].double() @ w.detach().double().t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,K,ldx", [(128 * 300 + 37, 1285, 1288), (70000, 512, 512), (128 * 257 + 1, 1385, 1385)])
def test_block_epilogue_vs_the_two_launch_route(rows, K, ldx, monkeypatch):
    """mappo_linear512_forward_norm (Linear + ReLU + LayerNorm of a reference block, mlp.py:17-22, in K15's forward) against
    K15's plain forward followed by K6: the pre-activation bit for bit, the block's output, its statistics and every gradient
    to rounding (the two routes differ in where the Linear's bias is added and in the order of the row sums).  The block form
    is opt-in (MAPPO_LINEAR512_NORM=1): measured 3.5 % slower on the Hanabi-shaped step, profiles/r06_ab_lin512_block_epilogue.json."""
    from onpolicy import _native
    from onpolicy.algorithms.utils.fused_norm import FusedLayerNorm, ACT_RELU, _LayerNormFn
    from onpolicy.algorithms.utils.tall_linear import _Linear512Fn, _Linear512NormFn
    monkeypatch.setenv("MAPPO_LINEAR512_MIN_ROWS", "1")
    g = torch.Generator(device=DEV).manual_seed(rows + K)
    x0 = torch.zeros(rows, ldx, device=DEV)
    x0[:, :K] = torch.randn(rows, K, device=DEV, generator=g) * 1.2 + 0.1
    w0 = torch.randn(512, K, device=DEV, generator=g) * (1.0 / K ** 0.5)
    b0 = torch.randn(512, device=DEV, generator=g) * 0.3
    norm = FusedLayerNorm(512).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(512, device=DEV, generator=g))
        norm.bias.copy_(0.2 * torch.randn(512, device=DEV, generator=g))
    dy = torch.randn(rows, 512, device=DEV, generator=g) / rows ** 0.5
    out = {}
    for route in ("block", "two"):
        x = x0.clone().requires_grad_(K == 512)
        w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
        norm.zero_grad()
        _native.count_calls(True)
        try:
            if route == "block":
                y = _Linear512NormFn.apply(x, w, b, norm.weight, norm.bias, norm.eps)
            else:
                y = _LayerNormFn.apply(_Linear512Fn.apply(x, w, None), norm.weight, norm.bias, norm.eps, ACT_RELU, b)
            y.backward(dy)
            torch.cuda.synchronize()
            calls = _native.calls()
        finally:
            _native.count_calls(False)
        assert calls.get("mappo_linear512_forward_norm", 0) == (1 if route == "block" else 0), calls
        assert calls.get("mappo_bias_act_layernorm_fwd", 0) == (0 if route == "block" else 1), calls
        out[route] = dict(y=y.detach(), dw=w.grad, db=b.grad, dg=norm.weight.grad.clone(), dbe=norm.bias.grad.clone(),
                          dx=x.grad if K == 512 else None)
    # against float64 on a slice of rows
    sl = slice(0, 4096)
    z64 = x0[sl, :K].double() @ w0.double().t() + b0.double()
    ref = torch.nn.functional.layer_norm(torch.relu(z64), (512,), norm.weight.double(), norm.bias.double(), norm.eps)
    ref = ref.detach()
    e_blk = float((out["block"]["y"][sl].double() - ref).abs().max())
    e_two = float((out["two"]["y"][sl].double() - ref).abs().max())
    print("\n[block epilogue rows %d K %d] y vs float64: block %.2e, two launches %.2e" % (rows, K, e_blk, e_two))
    assert e_blk <= max(2.0 * e_two, 5e-6)
    for k in ("y", "dw", "db", "dg", "dbe", "dx"):
        a, c = out["block"][k], out["two"][k]
        if a is None:
            continue
        a, c = a.detach().double(), c.detach().double()
        err = float((a - c).abs().max() / c.abs().max())
        fro = float((a - c).norm() / c.norm())
        # (a pre-activation within rounding of zero -- a few dozen of the 2e7 here -- has its unit on in one route and off in
        # the other: the entries of the gradients that row and unit reach move by that row's contribution, ~1 / sqrt(rows)
        # of the largest entry; everything else agrees to rounding, which the norm shows)
        print("   %-3s max %.2e  norm %.2e" % (k, err, fro))
        assert (err <= 1e-5) if k == "y" else (fro <= 2e-3 and err <= 5e-2), (k, err, fro)
    # rows past the last full tile: written; determinism
    x = x0.clone()
    y1 = _Linear512NormFn.apply(x, w0, b0, norm.weight, norm.bias, norm.eps)
    assert torch.equal(y1, out["block"]["y"]) and bool(torch.isfinite(y1).all())


@pytest.mark.parametrize("relu", [True, False], ids=["relu", "tanh"])
def test_hidden512_trunk_vs_float64_and_vs_the_library_route(monkeypatch, relu):
    """MLPBase at Hanabi's shapes (obs 1285, hidden 512, layer_N 2, input LayerNorm folded into the first Linear on
    standardised rows): outputs and every parameter gradient against the float64 module, through K15 (the default) and through
    the library GEMMs (--matrix_arithmetic f32_mfma); K15 must have carried every 512-wide product."""
    from onpolicy import _native
    from onpolicy.algorithms.utils.mlp import MLPBase
    rows, D = 70000, 1285
    g = torch.Generator().manual_seed(7)
    xs = torch.randn(rows, D, generator=g) * 1.3 + 0.2
    dy = torch.randn(rows, 512, generator=g) / rows ** 0.5
    res = {}
    for arith in ("six_term", "f32_mfma"):
        args = make_args(hidden_size=512, layer_N=2, use_ReLU=relu, matrix_arithmetic=arith)
        torch.manual_seed(5)
        base = MLPBase(args, (D,))
        ref = copy.deepcopy(base).double()
        base = base.to(DEV)
        x = xs.to(DEV)
        xhat = torch.nn.functional.layer_norm(x, (D,))          # what the standardising gather hands the trunk
        _native.count_calls(True)
        try:
            y = base(xhat, standardized=True)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            calls = _native.calls()
        finally:
            _native.count_calls(False)
        if arith == "six_term":
            assert calls.get("mappo_linear512_forward", 0) == 3 + 2 and calls.get("mappo_linear512_wgrad", 0) == 3, calls
            assert calls.get("mappo_linear512_forward_norm", 0) == 0, calls     # (opt-in: MAPPO_LINEAR512_NORM=1)
        else:
            assert calls.get("mappo_linear512_forward", 0) == 0
        y_ref = ref(xs.double())
        y_ref.backward(dy.double())
        errs = {"y": float((y.detach().cpu().double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())}
        for (name, p), q in zip(base.named_parameters(), ref.parameters()):
            errs[name] = float((p.grad.cpu().double() - q.grad).abs().max() / (q.grad.abs().max() + 1e-30))
        res[arith] = errs
    print("\n[hidden-512 trunk, %s] max error / largest entry vs float64: %s" % ("relu" if relu else "tanh", res))
    for k in res["six_term"]:
        # (a three-block ReLU / LayerNorm trunk in float32 sits 1e-3 from float64 on its input LayerNorm's gradient -- under
        # either arithmetic; what matters is that the six-term route is no further away than the library route)
        # (ReLU: a pre-activation within rounding of zero flips its unit on or off, whatever the arithmetic -- both routes sit
        # ~1e-2 from float64 there; Tanh is smooth)
        lim = 5e-2 if relu else 5e-3
        assert res["six_term"][k] < lim and res["f32_mfma"][k] < lim, (k, res["six_term"][k], res["f32_mfma"][k])
        assert res["six_term"][k] <= 4.0 * res["f32_mfma"][k] + 2e-6, (k, res["six_term"][k], res["f32_mfma"][k])
