"""-m gpu: the MAT hooks of the HBM buffer -- the transformer branches of
compute_returns through mappo_gae_mat_f32 and feed_forward_generator_transformer -- against fixtures
produced by the reference's SharedReplayBuffer with algorithm_name "mat" / "mat_dec".  Bit-exact."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from test_oracle_mat import FIELDS, BUF_FIELDS, mat_returns_cases

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _vn(n):
    from onpolicy.utils.valuenorm import ValueNorm
    vn = ValueNorm(1, device=DEV)
    vn.running_mean.fill_(float(n[0]))
    vn.running_mean_sq.fill_(float(n[1]))
    vn.debiasing_term.fill_(float(n[2]))
    return vn


def _buffer(args, A, Do=3, Ds=4, na=5):
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    return SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(na), device=DEV)


def test_mat_compute_returns_vs_reference(gold):
    for z, m, key in mat_returns_cases(gold):
        args = make_args(episode_length=m["T"], n_rollout_threads=m["N"], use_gae=m["use_gae"],
                         use_valuenorm=m["use_valuenorm"], use_proper_time_limits=m["use_proper_time_limits"],
                         algorithm_name=m["algo"])
        buf = _buffer(args, m["A"])
        for name in ("rewards", "masks", "bad_masks", "active_masks"):
            getattr(buf, name).copy_(torch.from_numpy(z[key + name]))
        buf.value_preds.copy_(torch.from_numpy(z[key + "value_preds_in"]))
        vn = _vn(z[key + "norm"]) if (key + "norm") in z else None
        buf.compute_returns(z[key + "next_value"], vn)
        np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"], err_msg=str(m))
        np.testing.assert_array_equal(buf.value_preds[-1].cpu().numpy(), z[key + "next_value"]
                                      if m["use_gae"] else z[key + "value_preds_in"][-1])
        if m["use_gae"] and not m["use_proper_time_limits"]:
            # the transformer branches store the GAE accumulator; its masked moments feed the trainer
            adv = z[key + "advantages"]
            np.testing.assert_array_equal(buf.advantages.cpu().numpy(), adv, err_msg=str(m))
            handle = buf.normalized_advantages(vn)
            np.testing.assert_array_equal(handle.raw.cpu().numpy(), adv)
            on = z[key + "active_masks"][:-1] != 0
            mean, std = [float(x) for x in handle.stats.cpu().numpy()]
            a64 = adv[on].astype(np.float64)
            assert mean == pytest.approx(a64.mean(), rel=1e-6, abs=1e-7)
            assert std == pytest.approx(a64.std(), rel=1e-6, abs=1e-7)


def test_mat_gae_large_and_argument_errors():
    """A north-star-shaped slice (many columns, 8 agents) against the oracle, plus the entry point's checks."""
    from oracle import oracle
    from onpolicy import _native
    T, N, A = 50, 4096, 8
    rng = np.random.default_rng(8)
    r = rng.standard_normal((T, N, A, 1)).astype(np.float32)
    v = rng.standard_normal((T + 1, N, A, 1)).astype(np.float32)
    nv = rng.standard_normal((N, A, 1)).astype(np.float32)
    m = (rng.random((T + 1, N, A, 1)) < 0.95).astype(np.float32)
    for denorm in (False, True):
        args = make_args(episode_length=T, n_rollout_threads=N, algorithm_name="mat", use_valuenorm=denorm)
        buf = _buffer(args, A)
        buf.rewards.copy_(torch.from_numpy(r)); buf.value_preds.copy_(torch.from_numpy(v))
        buf.masks.copy_(torch.from_numpy(m))
        vn = _vn([0.3e-4, 2.9e-4, 3.0e-5]) if denorm else None
        buf.compute_returns(nv, vn)
        sigma, mu = ([float(x) for x in vn.denorm_scalars().cpu().numpy()] if denorm else (1.0, 0.0))
        ret, _, adv = oracle.compute_returns_mat(r, v, nv, m, num_agents=A, sigma=sigma, mu=mu, denorm=denorm)
        np.testing.assert_array_equal(buf.returns[:-1].cpu().numpy(), ret[:-1])
        np.testing.assert_array_equal(buf.advantages.cpu().numpy(), adv)
    lib = _native.lib()
    p = _native.ptr
    b = buf
    call = lambda C, agents, flags, adv: lib.mappo_gae_mat_f32(
        p(b.rewards), p(b.value_preds), p(b._dev(nv).reshape(-1)), p(b.masks), p(b.returns), None, adv, None, None,
        T, C, agents, 0.99, 0.95, flags, None)
    assert call(N * A, A, 0, None) == -1                   # MAPPO_E_NULL: advantages are required
    assert call(N * A, 7, 0, p(b.advantages)) < 0          # C not a multiple of num_agents
    assert call(N * A, A, 1, p(b.advantages)) < 0          # only MAPPO_GAE_DENORM is a valid flag
    assert call(N * A, A, 4, p(b.advantages)) < 0          # DENORM without scalars


@pytest.mark.parametrize("recurrent", [True, False])
@pytest.mark.parametrize("case", ["tf1", "tf2", "tf7"])
def test_transformer_generator_vs_reference(gold, case, recurrent):
    z = gold.npz("mat_cases")
    meta = [m for m in gold.meta("mat_cases")["generators"] if m.get("case") == case][0]
    sh = z["mgen_buf_share_obs"].shape
    args = make_args(episode_length=sh[0] - 1, n_rollout_threads=sh[1], hidden_size=z["mgen_buf_rnn_states"].shape[-1],
                     algorithm_name="mat", sampler_rng="host", use_recurrent_policy=recurrent)
    buf = _buffer(args, sh[2], Do=z["mgen_buf_obs"].shape[-1], Ds=sh[-1], na=z["mgen_buf_available_actions"].shape[-1])
    for name in BUF_FIELDS:
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z["mgen_buf_" + name]))
    torch.manual_seed(4)
    batches = list(buf.feed_forward_generator_transformer(z["mgen_buf_advantages"], meta["num_mini_batch"]))
    assert len(batches) == meta["n_batches"]
    for bi, sample in enumerate(batches):
        assert len(sample) == 12
        for fname, t in zip(FIELDS, sample):
            exp = z["mgen_%s_b%d_%s" % (case, bi, fname)]
            assert t.is_cuda and tuple(t.shape) == exp.shape, (fname, tuple(t.shape), exp.shape)
            if fname.startswith("rnn_states") and not recurrent:
                assert float(t.abs().sum()) == 0.0       # a feed-forward buffer stores no RNN state
            else:
                np.testing.assert_array_equal(t.cpu().numpy(), exp, err_msg="%s b%d %s" % (case, bi, fname))
