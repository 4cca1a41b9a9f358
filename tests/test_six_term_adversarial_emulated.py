"""Worst-case numerics of K9's matrix products under both arithmetic forms (include/mappo_hip.h MAPPO_ARITH_*), on the host SIMT
emulator (same kernel source, bf16 conversions and MFMA accumulation restated in C++): magnitudes spread over 38 decades,
cancelling dot products, subnormal operands, non-finite and out-of-bf16-range operands -- against float64, bounds relative to
sum |x| |w| (tests/six_term_harness.py).  tests/test_gpu_six_term_adversarial.py repeats every case on the MI355X."""
import ctypes
import os

import numpy as np
import pytest

import mlp_reference as R
import six_term_harness as H

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ (host build of the emulator) not found")

SIX, F32 = 0, 1


@pytest.fixture(scope="module")
def be():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt"))
    import build
    return H.HostBackend(R.bind(ctypes.CDLL(build.build())))


@pytest.mark.parametrize("arith", [SIX, F32], ids=["six_term", "f32_mfma"])
@pytest.mark.parametrize("gen", sorted(H.GENERATORS))
@pytest.mark.parametrize("din", [384, 48])
def test_first_layer_products(be, arith, gen, din):
    """din 384: version 4 of the forward (first layer in six-term form); din 48: version 3 (first layer float32 MFMA in both
    forms -- the same bound must hold, which also calibrates it)."""
    worst, in_u = H.first_layer_errors(be, arith, gen, 96, din, seed=din + len(gen))
    print("\n[first layer %s din %d arith %d] worst error = %.2f of the bound, %.1f u sum|x||w|" % (gen, din, arith, worst, in_u))
    assert worst <= 1.0


@pytest.mark.parametrize("arith", [SIX, F32], ids=["six_term", "f32_mfma"])
@pytest.mark.parametrize("gen", ["wide_gamma", "cancelling"])
@pytest.mark.parametrize("din", [384, 48])
def test_hidden_layer_products(be, arith, gen, din):
    """The 64 x 64 hidden product: version 4 (weight planes in registers) at din 384, version 3 (planes in LDS) at din 48."""
    worst, in_u = H.hidden_layer_errors(be, arith, gen, 96, din, seed=din + len(gen))
    print("\n[hidden layer %s din %d arith %d] worst error = %.2f of the bound, %.1f u sum|n||w|" % (gen, din, arith, worst, in_u))
    assert worst <= 1.0


@pytest.mark.parametrize("arith", [SIX, F32], ids=["six_term", "f32_mfma"])
@pytest.mark.parametrize("gen", ["wide_rows", "cancelling"])
def test_first_layer_weight_gradient_products(be, arith, gen):
    """The direct first-layer weight-gradient kernel (din 384): the contraction runs over the rows."""
    worst, in_u = H.weight_gradient_errors(be, arith, gen, 16 * 14, 384, seed=len(gen))
    print("\n[dW1 %s arith %d] worst error = %.2f of the bound, %.1f u sum|dz||x|" % (gen, arith, worst, in_u))
    assert worst <= 1.0


@pytest.mark.parametrize("din", [384, 48])
def test_non_finite_and_out_of_range_operands(be, din):
    """The documented contract (include/mappo_hip.h MAPPO_ARITH_SIX_TERM): an operand that is +-inf, NaN or beyond the bf16
    range turns the outputs it reaches into NaN under the six-term form -- a superset of where the float32 form is non-finite
    (whose Tanh saturates a +-inf pre-activation to +-1: finite outputs) -- and never touches another row; the largest finite
    bf16 value is an ordinary operand."""
    six, y6, clean6, bad = H.non_finite_rows(be, SIX, din, 160, seed=din)
    f32, y32, clean32, _ = H.non_finite_rows(be, F32, din, 160, seed=din)
    others = np.setdiff1d(np.arange(160), bad)
    np.testing.assert_array_equal(y6[others], clean6[others])          # no other row changed by a single bit
    np.testing.assert_array_equal(y32[others], clean32[others])
    assert np.isfinite(clean6).all() and np.isfinite(clean32).all()
    inf_p, inf_m, nan_r, huge, bf16max = bad
    assert np.isnan(y32[nan_r]).all() and np.isnan(y6[nan_r]).all()     # NaN in -> NaN out, both forms
    for r in (inf_p, inf_m, huge):
        assert np.isfinite(y32[r]).all(), "float32 MFMA + Tanh: +-inf / 3.4e38 saturate to finite outputs"
    if din == 384:      # first layer in six-term form: inf - inf in the split
        for r in (inf_p, inf_m, huge):
            assert np.isnan(y6[r]).all()
    else:               # version 3: the first layer is float32 MFMA under both forms, the hidden layer sees bounded values
        for r in (inf_p, inf_m, huge):
            np.testing.assert_allclose(y6[r], y32[r], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y6[bf16max], y32[bf16max], rtol=1e-4, atol=1e-5)
    assert np.isfinite(y6[bf16max]).all()


@pytest.mark.parametrize("arith", [SIX, F32], ids=["six_term", "f32_mfma"])
def test_non_finite_rows_make_the_gradient_norm_non_finite(be, arith):
    """What the update relies on: with a poisoned row in the minibatch the first-layer weight gradient is non-finite under
    either form (float32: dz1 = 0 behind the saturated Tanh times inf = NaN), so the gradient norm is non-finite and
    mappo_clip_adam's contract (a non-finite norm poisons every gradient, tests/test_gpu_optim.py) applies identically."""
    net, y, clean, bad = H.non_finite_rows(be, arith, 384, 160, seed=9)
    grads, dz1 = net.backward(np.random.default_rng(1).standard_normal((160, 1)).astype(np.float32))
    w1g = grads[:64 * 384]
    assert not np.isfinite(w1g).all()
    assert not np.isfinite(np.sqrt((grads.astype(np.float64) ** 2).sum()))
