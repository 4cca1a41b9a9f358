"""-m gpu: the fused hidden-64 trunk (K9: mappo_mlp_forward / mappo_mlp_backward / mappo_row_stats through the C ABI)
against float64 copies of the reference's modules (MLPBase + output Linear, onpolicy/algorithms/utils/mlp.py:6-58) on
the same parameters and the same sampler rows.  Tolerance: float32 products accumulated in a different order than the
float64 reference, judged relative to the largest magnitude of each tensor (stated per assert)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import make_args

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _modules(din, layer_N, relu, out, feature_norm=True, seed=0):
    from onpolicy.algorithms.utils.mlp import MLPBase
    args = make_args(hidden_size=64, layer_N=layer_N, use_ReLU=relu, use_feature_normalization=feature_norm)
    torch.manual_seed(seed)
    base = MLPBase(args, (din,))
    head = nn.Linear(64, out) if out else None
    with torch.no_grad():       # LayerNorm affine parameters away from (1, 0) so that their gradients are exercised
        for p in base.parameters():
            if p.dim() == 1:
                p.add_(0.2 * torch.randn_like(p))
    return base, head


def _compare(din, layer_N, relu, out, rows, src_rows, chunk=None, feature_norm=True, rtol=3e-5):
    from onpolicy.algorithms.utils import fused_mlp
    base, head = _modules(din, layer_N, relu, out, feature_norm)
    ref_base, ref_head = copy.deepcopy(base).double(), (copy.deepcopy(head).double() if head else None)
    base, head = base.to(DEV), (head.to(DEV) if head else None)
    assert fused_mlp.trunk_supported(base)
    g = torch.Generator().manual_seed(din + rows)
    src = (torch.randn(src_rows, din, generator=g) * 1.5 + 0.7)
    if chunk is None:
        idx = torch.randperm(src_rows, generator=g)[:rows]
    else:
        L = chunk[0]
        idx = torch.randperm(src_rows // L, generator=g)[:rows // L]
    src_d = src.to(DEV)
    xin = fused_mlp.standardize_rows(src_d) if feature_norm else src_d      # once per train() in the trainer
    rs = fused_mlp.RowSource(xin, idx.to(DEV), chunk, standardized=feature_norm, width=din)     # (xin may be zero-padded)
    y = fused_mlp.trunk_forward(base, rs, head)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.to(DEV))
    # float64 reference on the rows the RowSource names
    x = src.double()[rs.source_rows().cpu()]
    feats = ref_base(x)
    y_ref = ref_head(feats) if ref_head else feats
    y_ref.backward(dy.double())
    tol = lambda ref: dict(rtol=0, atol=rtol * float(ref.detach().abs().max()) + 1e-7)
    torch.testing.assert_close(y.detach().cpu().double(), y_ref.detach(), **tol(y_ref))
    pairs = list(zip(base.named_parameters(), ref_base.parameters()))
    if head:
        pairs += list(zip(head.named_parameters(), ref_head.parameters()))
    for (name, p), q in pairs:
        assert p.grad is not None, name
        # gradients are sums over `rows` float32 terms: judged against the largest entry of the tensor
        torch.testing.assert_close(p.grad.cpu().double(), q.grad, rtol=0,
                                   atol=10 * rtol * float(q.grad.abs().max()) + 1e-6, msg=lambda m: name + ": " + m)


@pytest.fixture(autouse=True, params=[("six_term", 0), ("six_term", 128), ("six_term", 256), ("f32_mfma", 0)],
                ids=["six_term", "six_term_fwd4_every_width", "six_term_separate_dw1", "f32_mfma"])
def _arithmetic(request, monkeypatch):
    """Every test of this file under both arithmetic forms of the matrix products (include/mappo_hip.h MAPPO_ARITH_*; the
    modules built here carry no choice of their own, so the process default MAPPO_MATRIX_ARITHMETIC decides): the six-term
    bf16 form as shipped (the default), the same with tuning bit 128 (version 4 of the forward -- first layer in six-term
    form, one wave per SIMD -- also for inputs narrower than 128 floats), the same with tuning bit 256 (aligned inputs of at most
    64 columns keep the separate first-layer weight-gradient kernel instead of the chain's fused form), and the float32 MFMA.  Shapes without a six-term
    kernel run the float32 kernels under every parameter."""
    from onpolicy import _native
    name, flags = request.param
    monkeypatch.setenv("MAPPO_MATRIX_ARITHMETIC", name)
    old = _native.lib().mappo_mlp_set_flags(flags)
    assert old >= 0
    yield
    _native.lib().mappo_mlp_set_flags(old)


CASES = [
    (48, 1, False, 5, 70, 200),            # north-star actor (tanh, Discrete(5)), one partial tile
    (384, 1, False, 1, 5000, 9000),        # north-star critic width, many tiles per workgroup loop
    (30, 1, True, 1, 128, 128),            # ReLU, exactly one tile
    (19, 0, False, 3, 37, 64),             # single Linear block, odd (unaligned) rows
    (435, 2, False, 18, 300, 700),         # SMAC widths: odd din > 384 (two k slabs), layer_N = 2, 18 actions
    (150, 1, False, 0, 1000, 4000),        # trunk only (features for the GRU)
    (200, 1, False, 2, 16 * 700 + 3, 20000),   # direct-to-LDS weight gradient with waves of two and of one k tile, ragged end
    (28, 1, True, 2, 64 * 300 + 17, 30000),    # one k tile: row-split weight gradient, several tiles per wave
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_trunk_vs_float64_modules(case):
    _compare(*case)


@pytest.mark.parametrize("din", [384, 400, 48, 30])
def test_persistent_loops_in_steady_state(din):
    """With the grid capped at 2 workgroups every workgroup walks dozens of tiles: the software pipelines of all K9 kernels
    (register chunk buffers of the forward, the direct-to-LDS slot rings and their counted waits in the weight-gradient
    kernels: direct (384, two k slabs at 400), row-split (48), loader version (30)) run through their steady state with
    the values checked, which the uncapped cases -- one or two tiles per workgroup -- do not exercise."""
    from onpolicy import _native
    _native.lib().mappo_mlp_set_grid_cap(2)
    try:
        _compare(din, 1, False, 3, 128 * 24 + 16 * 3 + 5, 9000)
    finally:
        _native.lib().mappo_mlp_set_grid_cap(0)


def test_trunk_without_input_layernorm():
    _compare(40, 1, False, 4, 450, 450, feature_norm=False)


def test_trunk_on_recurrent_chunk_rows():
    T, N, A, L = 7, 30, 2, 3           # T % L != 0: chunks straddle trajectories (shared_buffer.py:554-566)
    _compare(24, 1, False, 0, L * 100, T * N * A, chunk=(L, T, N, A))


@pytest.mark.parametrize("din,out", [(384, 1), (48, 5)])
def test_trunk_at_scale_vs_float64(din, out):
    """One million rows (every workgroup of every K9 kernel deep in its persistent loop -- 30 to 250 tiles each) against
    the float64 modules on the device: outputs and every gradient."""
    from onpolicy.algorithms.utils import fused_mlp
    base, head = _modules(din, 1, False, out, seed=5)
    ref_base, ref_head = copy.deepcopy(base).double().to(DEV), copy.deepcopy(head).double().to(DEV)
    base, head = base.to(DEV), head.to(DEV)
    rows, src_rows = (1 << 20) + 77, (1 << 20) + 5000
    g = torch.Generator(device=DEV).manual_seed(din)
    src = torch.randn(src_rows, din, device=DEV, generator=g) * 1.5 + 0.7
    idx = torch.randperm(src_rows, device=DEV, generator=g)[:rows]
    rs = fused_mlp.RowSource(fused_mlp.standardize_rows(src), idx, standardized=True, width=din)
    dy = torch.randn(rows, out, device=DEV, generator=g) / rows ** 0.5
    y = fused_mlp.trunk_forward(base, rs, head)
    y.backward(dy)
    y_ref = ref_head(ref_base(src.double()[idx]))
    y_ref.backward(dy.double())
    torch.testing.assert_close(y.detach().double(), y_ref.detach(), rtol=0, atol=1e-5 * float(y_ref.abs().max()))
    for (name, p), q in list(zip(base.named_parameters(), ref_base.parameters())) + \
            list(zip(head.named_parameters(), ref_head.parameters())):
        # sums of a million float32 terms against float64, judged against the tensor's largest entry (measured: 1e-6)
        torch.testing.assert_close(p.grad.double(), q.grad, rtol=0, atol=2e-5 * float(q.grad.abs().max()) + 1e-9,
                                   msg=lambda m: name + ": " + m)


def test_trunk_at_scale_is_deterministic_and_finite():
    """2.6 M rows through every workgroup's persistent loop: finite, identical run to run (fixed reduction order), and
    the row-sum gradient of the output bias equals the column sums of dy."""
    from onpolicy.algorithms.utils import fused_mlp
    base, head = _modules(384, 1, False, 1, seed=3)
    base, head = base.to(DEV), head.to(DEV)
    rows = 1 << 21
    src = torch.randn(rows + 1000, 384, device=DEV)
    rs = fused_mlp.RowSource(fused_mlp.standardize_rows(src), torch.randperm(rows + 1000, device=DEV)[:rows],
                             standardized=True)
    dy = torch.randn(rows, 1, device=DEV)
    outs = []
    for _ in range(2):
        for p in list(base.parameters()) + list(head.parameters()):
            p.grad = None
        y = fused_mlp.trunk_forward(base, rs, head)
        y.backward(dy)
        outs.append((y.detach().clone(), [p.grad.clone() for p in base.parameters()], head.bias.grad.clone()))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    torch.testing.assert_close(outs[0][2], dy.double().sum(0).float(), rtol=1e-4, atol=1e-2)


def test_row_source_materialize_equals_eager_gather():
    from onpolicy.algorithms.utils import fused_mlp
    src = torch.randn(500, 54, device=DEV) * 3 + 1
    xhat = fused_mlp.standardize_rows(src)          # 54 -> rows of 56 floats, two zero columns
    assert xhat.shape == (500, 56) and not xhat[:, 54:].any()
    assert torch.equal(xhat[:, :54], fused_mlp.standardize_rows(src, pad=False))
    torch.testing.assert_close(xhat[:, :54], torch.nn.functional.layer_norm(src, (54,)), rtol=1e-4, atol=1e-5)
    idx = torch.randperm(500, device=DEV)[:123]
    rs = fused_mlp.RowSource(xhat, idx, standardized=True, width=54)
    assert rs.shape == (123, 54) and torch.equal(rs.materialize(), xhat[idx][:, :54])
    assert rs[10:20].rows == 10 and torch.equal(rs[10:20].idx, idx[10:20])
    assert rs.table().dtype == torch.int32 and torch.equal(rs.table()[:123].long(), idx)
