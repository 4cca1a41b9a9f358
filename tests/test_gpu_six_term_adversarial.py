"""-m gpu: worst-case numerics of the K9 / K12 matrix products on the MI355X under both arithmetic forms (include/mappo_hip.h
MAPPO_ARITH_SIX_TERM -- the default -- and MAPPO_ARITH_F32_MFMA) against float64: magnitudes spread over 36 decades, cancelling
dot products, subnormal operands, +-inf / NaN / beyond-bf16-range operands, and the update's poison contract under a non-finite
observation.  Bounds are relative to sum |x| |w| (tests/six_term_harness.py); the host-emulator twin of every kernel-level case
is tests/test_six_term_adversarial_emulated.py.  VERDICT r4 "Next round" 1(b)."""
import ctypes

import numpy as np
import pytest
import torch

import mlp_reference as R
import six_term_harness as H
from helpers import Box, Discrete, make_args

pytestmark = pytest.mark.gpu

SIX, F32 = 0, 1
ARITH = pytest.mark.parametrize("arith", [SIX, F32], ids=["six_term", "f32_mfma"])


@pytest.fixture(scope="module")
def be():
    from onpolicy import _native
    _native.lib()
    return H.DeviceBackend(R.bind(ctypes.CDLL(_native.LIB_PATH)), torch.device("cuda", 0))


@ARITH
@pytest.mark.parametrize("gen", sorted(H.GENERATORS))
@pytest.mark.parametrize("din", [384, 48])
def test_first_layer_products(be, arith, gen, din):
    worst, in_u = H.first_layer_errors(be, arith, gen, 128 * 40 + 19, din, seed=din + len(gen))
    print("\n[first layer %s din %d arith %d] worst error = %.2f of the bound, %.1f u sum|x||w|" % (gen, din, arith, worst, in_u))
    assert worst <= 1.0


@ARITH
@pytest.mark.parametrize("gen", ["wide_gamma", "cancelling"])
@pytest.mark.parametrize("din", [384, 48])
def test_hidden_layer_products(be, arith, gen, din):
    worst, in_u = H.hidden_layer_errors(be, arith, gen, 128 * 40 + 19, din, seed=din + len(gen))
    print("\n[hidden layer %s din %d arith %d] worst error = %.2f of the bound, %.1f u sum|n||w|" % (gen, din, arith, worst, in_u))
    assert worst <= 1.0


@ARITH
@pytest.mark.parametrize("gen", ["wide_rows", "cancelling"])
def test_first_layer_weight_gradient_products(be, arith, gen):
    worst, in_u = H.weight_gradient_errors(be, arith, gen, 16 * 700, 384, seed=len(gen))
    print("\n[dW1 %s arith %d] worst error = %.2f of the bound, %.1f u sum|dz||x|" % (gen, arith, worst, in_u))
    assert worst <= 1.0


@pytest.mark.parametrize("din", [384, 48])
def test_non_finite_and_out_of_range_operands(be, din):
    """include/mappo_hip.h MAPPO_ARITH_SIX_TERM: +-inf / NaN / |x| >= 3.3962e38 operands -> NaN in every output they reach (a
    superset of where the float32 form is non-finite: its Tanh saturates an infinite pre-activation to +-1), no other row
    touched, the largest finite bf16 value an ordinary operand."""
    six, y6, clean6, bad = H.non_finite_rows(be, SIX, din, 128 * 3 + 5, seed=din)
    f32, y32, clean32, _ = H.non_finite_rows(be, F32, din, 128 * 3 + 5, seed=din)
    others = np.setdiff1d(np.arange(len(y6)), bad)
    np.testing.assert_array_equal(y6[others], clean6[others])
    np.testing.assert_array_equal(y32[others], clean32[others])
    inf_p, inf_m, nan_r, huge, bf16max = bad
    assert np.isnan(y32[nan_r]).all() and np.isnan(y6[nan_r]).all()
    for r in (inf_p, inf_m, huge):
        assert np.isfinite(y32[r]).all()
        if din == 384:
            assert np.isnan(y6[r]).all()
        else:
            np.testing.assert_allclose(y6[r], y32[r], rtol=1e-4, atol=1e-5)
    assert np.isfinite(y6[bf16max]).all()
    np.testing.assert_allclose(y6[bf16max], y32[bf16max], rtol=1e-4, atol=1e-5)
    # and the gradient norm of a minibatch with such rows is non-finite under either form
    for net in (six, f32):
        grads, _ = net.backward(np.random.default_rng(1).standard_normal((len(y6), 1)).astype(np.float32))
        assert not np.isfinite(np.sqrt((grads.astype(np.float64) ** 2).sum()))


@pytest.mark.parametrize("arith", ["six_term", "f32_mfma"])
@pytest.mark.parametrize("bad", [float("inf"), float("nan"), 3.4e38])
def test_update_poison_contract_under_a_non_finite_observation(arith, bad):
    """End to end: one bad entry in ``share_obs`` of a critic WITHOUT input LayerNorm (the value reaches K9 as it is).  The
    critic's gradient norm is non-finite under either arithmetic form, mappo_clip_adam poisons every critic gradient like
    clip_grad_norm_ does (reference r_mappo.py:146-167), Adam carries it into every critic parameter; the actor, whose
    observations are clean, stays finite."""
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.utils import fused_mlp
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    dev = torch.device("cuda", 0)
    T, N, A, Do, Ds, na = 8, 16, 3, 48, 384, 5
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=1,
                     num_mini_batch=1, use_feature_normalization=False, matrix_arithmetic=arith)
    spaces = Box((Do,)), Box((Ds,)), Discrete(na)
    torch.manual_seed(3)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, A, *spaces, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    for name in ("share_obs", "obs", "rewards"):
        getattr(buf, name).normal_(generator=g)
    buf.value_preds[:-1].normal_(generator=g)
    buf.actions.copy_(torch.randint(0, na, buf.actions.shape, generator=g, device=dev).float())
    buf.action_log_probs.fill_(-float(np.log(na)))
    buf.share_obs[2, 5, 1, 17] = bad
    buf.compute_returns(torch.zeros(N, A, 1), trainer.value_normalizer)
    trainer.prep_training()
    fused_mlp.profile(True)
    try:
        info = trainer.train(buf)
        torch.cuda.synchronize()
        launches = fused_mlp.profile_times()
    finally:
        fused_mlp.profile(False)
    assert launches["mappo_mlp_forward"][0] == 2 and launches["mappo_mlp_backward"][0] == 2      # K9 carried both networks
    assert np.isfinite(info["actor_grad_norm"])
    for name, p in policy.actor.named_parameters():
        assert bool(torch.isfinite(p).all()), name
    if bad == 3.4e38 and arith == "f32_mfma":
        # the documented difference: a finite value beyond the bf16 range is an ordinary float32 operand -- the Tanh saturates,
        # its gradient is exactly zero, nothing is poisoned; under the six-term form it splits into inf - inf = NaN
        assert np.isfinite(info["critic_grad_norm"])
        assert all(bool(torch.isfinite(p).all()) for p in policy.critic.parameters())
        return
    assert not np.isfinite(info["critic_grad_norm"])
    for name, p in policy.critic.named_parameters():
        assert bool(torch.isnan(p).all()), name


def _gru_errors(monkeypatch, seed, decades):
    """max |K12 - float64| / largest entry of every output and gradient of one RNNLayer call, under both arithmetic forms."""
    from onpolicy.algorithms.utils.rnn import RNNLayer
    from test_gru_kernels_emulated import reference
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    layer = RNNLayer(64, 64, 1, True)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for name in ("weight_ih_l0", "weight_hh_l0"):
            w = getattr(layer.rnn, name)
            w.mul_(10.0 ** (decades * torch.rand(w.shape, generator=g) - decades / 2))
        for p in (layer.rnn.bias_ih_l0, layer.rnn.bias_hh_l0, layer.norm.weight, layer.norm.bias):
            p.add_(0.1 * torch.randn(p.shape, generator=g))
    layer = layer.to(dev)
    L, B = 10, 32 * 40 + 7
    x = torch.randn(L * B, 64, generator=g) * 10.0 ** (8 * torch.rand(L * B, 1, generator=g) - 6)
    h0 = torch.randn(B, 1, 64, generator=g)
    masks = (torch.rand(L * B, 1, generator=g) > 0.1).float()
    dy = torch.randn(L * B, 64, generator=g)
    P = {"w_ih": layer.rnn.weight_ih_l0, "w_hh": layer.rnn.weight_hh_l0, "b_ih": layer.rnn.bias_ih_l0,
         "b_hh": layer.rnn.bias_hh_l0, "ln_g": layer.norm.weight, "ln_b": layer.norm.bias}
    tp = {k: v.detach().cpu().double().requires_grad_() for k, v in P.items()}
    tx, th = x.double().requires_grad_(), h0[:, 0].double().requires_grad_()
    y_ref, h_ref = reference(tp, tx, th, masks[:, 0].double(), L, B)
    (y_ref * dy.double()).sum().backward()
    ref = {"y": y_ref.detach(), "h_last": h_ref.detach(), "dx": tx.grad, "dh0": th.grad}
    ref.update({k: tp[k].grad for k in P})
    err = {}
    for arith in ("six_term", "f32_mfma"):
        monkeypatch.setenv("MAPPO_MATRIX_ARITHMETIC", arith)
        for p in layer.parameters():
            p.grad = None
        xd, hd = x.to(dev).requires_grad_(), h0.to(dev).requires_grad_()
        y, h_last = layer(xd, hd, masks.to(dev))
        (y * dy.to(dev)).sum().backward()
        got = {"y": y.detach(), "h_last": h_last[:, 0].detach(), "dx": xd.grad, "dh0": hd.grad[:, 0]}
        got.update({k: P[k].grad for k in P})
        assert all(bool(torch.isfinite(v).all()) for v in got.values())
        err[arith] = {k: float((got[k].cpu().double() - ref[k]).abs().max() / ref[k].abs().max()) for k in ref}
    return err


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_gru_chunk_kernels_on_wide_magnitudes(monkeypatch, seed):
    """K12 with the trunk features and the weights far from N(0, 1): rows of x at 1e-6 .. 1e2 of their usual size, W_ih / W_hh
    entries spread over two decades (saturated and near-linear gates side by side).  Both forms stay within 2e-4 of float64
    (relative to the largest entry) and the six-term form is no further from it than a small multiple of the float32 MFMA.
    Measured over 16 seeds x 2 host thread counts (tools/r05/k12_wide_stats.py, MI355X): six-term <= 2.2e-5, float32 MFMA
    <= 5.0e-5, worst per-quantity ratio 3.0, median 1.4."""
    err = _gru_errors(monkeypatch, seed, 2.0)
    print("\n[K12 wide magnitudes, seed %d] max error / largest entry vs float64:" % seed, err)
    for k in err["six_term"]:
        assert err["f32_mfma"][k] < 2e-4 and err["six_term"][k] < 2e-4, (k, err)
        assert err["six_term"][k] <= 5.0 * err["f32_mfma"][k] + 5e-6, (k, err["six_term"][k], err["f32_mfma"][k])


def test_gru_chunk_kernels_where_the_recurrence_amplifies_rounding(monkeypatch):
    """The same with the weights spread over FOUR decades: gates with slopes up to 25 over ten steps, then a LayerNorm, amplify
    any float32 rounding by three orders of magnitude, so float64 is matched only loosely by EITHER form and which of the two
    lands closer depends on the instance (the same seeds gave ratios between 0.8 and 5.3 when only the host's thread count --
    hence the last bit of the orthogonal initialisation -- changed; 32 instances: both forms up to 4e-2).  Asserted: everything
    finite, both forms within 0.15 of float64's largest entry."""
    err = _gru_errors(monkeypatch, 2, 4.0)
    print("\n[K12 amplified rounding] max error / largest entry vs float64:", err)
    for k in err["six_term"]:
        assert err["f32_mfma"][k] < 0.15 and err["six_term"][k] < 0.15, (k, err)
