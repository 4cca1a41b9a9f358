"""-m gpu: K12, the GRU of the recurrent policies over a whole chunk in one launch per direction
(mappo_gru_seq_forward / _backward), through RNNLayer and autograd on the device against a float64 restatement of the
reference's RNNLayer (onpolicy/algorithms/utils/rnn.py:7-80).  The reference-generated fixtures that run through it are
tests/test_gpu_trainer_h64.py::test_fused_trunk_update_vs_reference[h64_gru*]."""
import numpy as np
import pytest
import torch

from test_gru_kernels_emulated import reference

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["six_term", "f32_mfma"])
def _arithmetic(request, monkeypatch):
    """Every test of this file under both arithmetic forms of K12's projections (include/mappo_hip.h MAPPO_ARITH_*: the
    six-term bf16 form -- the default -- and the float32 MFMA); the layers built here carry no choice of their own, so the
    process default MAPPO_MATRIX_ARITHMETIC decides."""
    monkeypatch.setenv("MAPPO_MATRIX_ARITHMETIC", request.param)
    yield


def _layer(dev, seed):
    from onpolicy.algorithms.utils.rnn import RNNLayer
    torch.manual_seed(seed)
    layer = RNNLayer(64, 64, 1, True)
    with torch.no_grad():
        for p in layer.parameters():        # biases and LayerNorm parameters away from their 0 / 1 defaults
            p.add_(0.1 * torch.randn_like(p))
    return layer.to(dev)


@pytest.mark.parametrize("L,B", [(1, 200), (10, 96), (10, 32 * 1024 + 17), (4, 33)])
def test_chunk_kernel_layer_vs_float64(L, B):
    dev = torch.device("cuda", 0)
    layer = _layer(dev, L + B)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(L * B, 64, generator=g)
    h0 = torch.randn(B, 1, 64, generator=g)
    masks = (torch.rand(L * B, 1, generator=g) > 0.1).float()
    dy = torch.randn(L * B, 64, generator=g)
    xd, hd = x.to(dev).requires_grad_(), h0.to(dev).requires_grad_()
    assert layer._chunk_kernel_ok(xd)
    y, h_last = layer(xd, hd, masks.to(dev))
    (y * dy.to(dev)).sum().backward()

    P = {"w_ih": layer.rnn.weight_ih_l0, "w_hh": layer.rnn.weight_hh_l0, "b_ih": layer.rnn.bias_ih_l0,
         "b_hh": layer.rnn.bias_hh_l0, "ln_g": layer.norm.weight, "ln_b": layer.norm.bias}
    tp = {k: v.detach().cpu().double().requires_grad_() for k, v in P.items()}
    tx, th = x.double().requires_grad_(), h0[:, 0].double().requires_grad_()
    y_ref, h_ref = reference(tp, tx, th, masks[:, 0].double(), L, B)
    (y_ref * dy.double()).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), y_ref.detach(), rtol=0, atol=2e-5 * float(y_ref.abs().max()))
    torch.testing.assert_close(h_last[:, 0].detach().cpu().double(), h_ref.detach(), rtol=0, atol=2e-5)
    pairs = [("dx", xd.grad, tx.grad), ("dh0", hd.grad[:, 0], th.grad)] + [(k, P[k].grad, tp[k].grad) for k in P]
    for name, got, ref in pairs:
        torch.testing.assert_close(got.cpu().double(), ref, rtol=0, atol=1e-4 * float(ref.abs().max()) + 1e-9, msg=name)


def test_chunk_kernel_is_deterministic():
    dev = torch.device("cuda", 0)
    layer = _layer(dev, 5)
    L, B = 10, 50000
    g = torch.Generator().manual_seed(4)
    x, h0 = torch.randn(L * B, 64, generator=g).to(dev), torch.randn(B, 1, 64, generator=g).to(dev)
    masks = (torch.rand(L * B, 1, generator=g) > 0.05).float().to(dev)
    outs = []
    for _ in range(2):
        layer.zero_grad()
        xd = x.clone().requires_grad_()
        y, _ = layer(xd, h0, masks)
        y.square().sum().backward()
        outs.append([y.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("out,L,B", [(1, 10, 96), (5, 10, 2000), (18, 10, 777), (6, 1, 200), (2, 4, 33)])
def test_output_linear_inside_the_chunk_kernels(out, L, B):
    """RNNLayer.forward(..., head=Linear(64, out)): the head's output comes out of the forward launch and the backward
    launch forms the gradient at the layer's output from the gradient at the head's output; the head's own gradients
    (split-K GEMM on the kept features, column sums) and every other gradient against the float64 restatement."""
    dev = torch.device("cuda", 0)
    layer = _layer(dev, out + L + B)
    torch.manual_seed(out)
    head = torch.nn.Linear(64, out).to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(L * B, 64, generator=g)
    h0 = torch.randn(B, 1, 64, generator=g)
    masks = (torch.rand(L * B, 1, generator=g) > 0.1).float()
    dlog = torch.randn(L * B, out, generator=g)
    xd, hd = x.to(dev).requires_grad_(), h0.to(dev).requires_grad_()
    assert layer.head_ok(xd, head)
    logits, h_last = layer(xd, hd, masks.to(dev), head=head)
    assert logits.shape == (L * B, out)
    ((logits * dlog.to(dev)).sum() + 0.5 * h_last.sum()).backward()

    P = {"w_ih": layer.rnn.weight_ih_l0, "w_hh": layer.rnn.weight_hh_l0, "b_ih": layer.rnn.bias_ih_l0,
         "b_hh": layer.rnn.bias_hh_l0, "ln_g": layer.norm.weight, "ln_b": layer.norm.bias, "head_w": head.weight,
         "head_b": head.bias}
    tp = {k: v.detach().cpu().double().requires_grad_() for k, v in P.items()}
    tx, th = x.double().requires_grad_(), h0[:, 0].double().requires_grad_()
    y_ref, h_ref = reference(tp, tx, th, masks[:, 0].double(), L, B)
    logits_ref = y_ref @ tp["head_w"].t() + tp["head_b"]
    ((logits_ref * dlog.double()).sum() + 0.5 * h_ref.sum()).backward()
    torch.testing.assert_close(logits.detach().cpu().double(), logits_ref.detach(), rtol=0,
                               atol=3e-5 * float(logits_ref.abs().max()))
    pairs = [("dx", xd.grad, tx.grad), ("dh0", hd.grad[:, 0], th.grad)] + [(k, P[k].grad, tp[k].grad) for k in P]
    for name, got, ref in pairs:
        torch.testing.assert_close(got.cpu().double(), ref, rtol=0, atol=1e-4 * float(ref.abs().max()) + 1e-9, msg=name)
    # a head the kernels do not take is refused, not silently evaluated some other way
    wide = torch.nn.Linear(64, 19).to(dev)
    assert not layer.head_ok(xd, wide)
    with pytest.raises(ValueError):
        layer(xd, hd, masks.to(dev), head=wide)


@pytest.mark.parametrize("rows", [1, 16 * 5 + 3, 128000, 16 * 70000 + 9])
def test_weight_gradient_kernel_vs_float64_and_the_library_route(rows, monkeypatch):
    """mappo_gru_weight_grads (round 5: dW_ih and dW_hh of K12 in ONE six-term launch instead of three library GEMMs) against
    float64 with the six-term bound, at sizes from one row to a million (every workgroup count, ragged last tile); and through
    RNNLayer: the kernel's gradients agree with the library route's (MAPPO_GRU_WEIGHT_GRAD_KERNEL=0) to float32 noise."""
    from onpolicy import _native
    from test_gru_kernels_emulated import weight_grads_case
    dev = torch.device("cuda", 0)
    for scale in (None, 12.0):
        dw, ref, mag = weight_grads_case(_native.lib(), rows, rows % 1000 + (0 if scale is None else 1), scale, device=dev)
        bound = (16 + rows / 6) * 2.0 ** -24 * mag + 1e-30
        worst = float(((dw - ref).abs() / bound).max())
        assert torch.isfinite(dw).all() and worst <= 1.0, (rows, scale, worst)


def test_weight_gradient_kernel_carries_the_six_term_backward(monkeypatch):
    from onpolicy import _native
    import onpolicy.algorithms.utils.rnn as rnn_mod
    dev = torch.device("cuda", 0)
    L, B = 10, 4000
    g = torch.Generator().manual_seed(11)
    x = torch.randn(L * B, 64, generator=g).to(dev)
    h0 = torch.randn(B, 1, 64, generator=g).to(dev)
    masks = (torch.rand(L * B, 1, generator=g) > 0.1).float().to(dev)
    dy = torch.randn(L * B, 64, generator=g).to(dev)
    grads = {}
    for use in (True, False):
        monkeypatch.setattr(rnn_mod, "_WEIGHT_GRAD_KERNEL", use)
        layer = _layer(dev, 5)
        _native.count_calls(True)
        y, _ = layer(x.clone().requires_grad_(), h0.clone().requires_grad_(), masks)
        (y * dy).sum().backward()
        calls = _native.calls()
        _native.count_calls(False)
        six = _native.default_arith() == _native.ARITH_SIX_TERM
        assert (calls.get("mappo_gru_weight_grads", 0) == 1) == (use and six), calls
        grads[use] = [layer.rnn.weight_ih_l0.grad.clone(), layer.rnn.weight_hh_l0.grad.clone()]
    for a, b in zip(grads[True], grads[False]):
        torch.testing.assert_close(a, b, rtol=0, atol=2e-5 * float(b.abs().max()))
