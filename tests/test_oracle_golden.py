"""The plain-C / numpy oracle against fixtures produced by the reference itself
(oracle/make_golden.py).  Bit-exact for everything except the two advantage moments, which
numpy accumulates in float32 pairwise sums (tolerance stated below)."""
import numpy as np
import pytest
import torch

from oracle import oracle


def _flags(m):
    return dict(use_gae=m["use_gae"], use_proper_time_limits=m["use_proper_time_limits"],
                denorm=m["use_valuenorm"])


def _scalars(z, key, m):
    if not m["use_valuenorm"]:
        return 1.0, 0.0
    n = z[key + "norm"]
    return oracle.normalizer_scalars(n[0], n[1], n[2])


def test_kats(gold):
    """SURVEY.md section 8c known-answer vectors A-E."""
    z = gold.npz("kat_returns")
    r = np.array([1, 2, 3, -1], dtype=np.float32).reshape(4, 1)
    v = np.array([0.5, 0.4, 0.3, 0.25, 0.0], dtype=np.float32).reshape(5, 1)
    m = np.array([1, 1, 0, 1, 1], dtype=np.float32).reshape(5, 1)
    b = np.array([1, 1, 1, 1, 0], dtype=np.float32).reshape(5, 1)
    nv = np.array([0.2], dtype=np.float32)
    expect = {"A": [2.88298, 2.0, 2.0793593, -0.98020005, 0], "C": [2.9008002, 2.0, 2.2580938, -0.80200005, 0],
              "D": [2.9008002, 2.0, 3.2475, 0.25, 0], "E": [2.98, 2.0, 2.2060199, -0.802, 0.2]}
    kw = {"A": dict(denorm=True), "B": dict(denorm=True), "C": {}, "D": dict(use_proper_time_limits=True),
          "E": dict(use_gae=False)}
    for name in "ABCDE":
        sigma, mu = 1.0, 0.0
        if name in "AB":
            n = z["kat_%s_norm" % name]
            sigma, mu = oracle.normalizer_scalars(n[0], n[1], n[2])
        ret, _ = oracle.compute_returns(r, v, nv, m, b, sigma=sigma, mu=mu, **kw[name])
        np.testing.assert_array_equal(ret[:, 0], z["kat_%s_returns" % name])
        if name in expect:
            np.testing.assert_allclose(ret[:, 0], np.array(expect[name], dtype=np.float32), rtol=1e-6)


def test_compute_returns_matrix(gold):
    z = gold.npz("returns_cases")
    for m in gold.meta("returns_cases"):
        key = "ret%03d_" % m["id"]
        sigma, mu = _scalars(z, key, m)
        ret, v = oracle.compute_returns(z[key + "rewards"], z[key + "value_preds_in"], z[key + "next_value"],
                                        z[key + "masks"], z[key + "bad_masks"], sigma=sigma, mu=mu,
                                        **_flags(m))
        np.testing.assert_array_equal(ret, z[key + "returns"], err_msg=str(m))
        np.testing.assert_array_equal(v, z[key + "value_preds_out"], err_msg=str(m))
        adv = oracle.advantages(ret, v, sigma=sigma, mu=mu, denorm=m["use_valuenorm"])
        np.testing.assert_array_equal(adv, z[key + "advantages"], err_msg=str(m))
        mean, std, cnt = oracle.adv_moments(adv, z[key + "active_masks"][:-1])
        gm, gs = z[key + "adv_mean_std"]
        # numpy nanmean/nanstd: float32 pairwise accumulation -> a few ulp of float32
        assert abs(mean - gm) <= 2e-6 * max(1.0, abs(gm)) + 1e-7, m
        if np.isfinite(gs):
            assert abs(std - gs) <= 2e-6 * max(1.0, abs(gs)) + 1e-7, m
        # normalisation itself is bit-exact given the reference's float32 mean / std
        normed = oracle.adv_normalize(adv, gm, gs)
        np.testing.assert_array_equal(normed, z[key + "advantages_normed"], err_msg=str(m))


FIELDS = ["share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
          "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions"]
BUF_NAME = dict(old_action_log_probs="action_log_probs", adv_targ="advantages")


def _oracle_buffer(z):
    class NS(object):
        pass
    sh = z["gen_buf_share_obs"].shape
    args = NS()
    args.episode_length, args.n_rollout_threads = sh[0] - 1, sh[1]
    args.hidden_size, args.recurrent_N = z["gen_buf_rnn_states"].shape[-1], 1
    args.gamma, args.gae_lambda = 0.99, 0.95
    args.use_gae, args.use_popart, args.use_valuenorm, args.use_proper_time_limits = True, False, True, False

    class Box(object):
        def __init__(self, shape):
            self.shape = shape

    class Discrete(object):
        def __init__(self, n):
            self.n = n
    buf = oracle.OracleBuffer(args, sh[2], Box((z["gen_buf_obs"].shape[-1],)), Box((sh[-1],)),
                              Discrete(z["gen_buf_available_actions"].shape[-1]))
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
                 "masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        getattr(buf, name)[...] = z["gen_buf_" + name]
    return buf


@pytest.mark.parametrize("case,call", [
    ("ff2", lambda b, a: b.feed_forward_generator(a, 2)),
    ("ff7", lambda b, a: b.feed_forward_generator(a, 7)),
    ("rec_L5", lambda b, a: b.recurrent_generator(a, 2, 5)),
    ("rec_L4", lambda b, a: b.recurrent_generator(a, 3, 4)),
    ("naive3", lambda b, a: b.naive_recurrent_generator(a, 3)),
])
def test_generators(gold, case, call):
    z = gold.npz("generator_cases")
    buf = _oracle_buffer(z)
    adv = z["gen_buf_advantages"]
    torch.manual_seed(5)  # same CPU generator state the fixture was drawn under
    batches = list(call(buf, adv))
    n = [m for m in gold.meta("generator_cases") if m.get("case") == case][0]["n_batches"]
    assert len(batches) == n
    for bi, sample in enumerate(batches):
        assert len(sample) == 12
        for fname, arr in zip(FIELDS, sample):
            np.testing.assert_array_equal(arr, z["gen_%s_b%d_%s" % (case, bi, fname)],
                                          err_msg="%s batch %d field %s" % (case, bi, fname))
