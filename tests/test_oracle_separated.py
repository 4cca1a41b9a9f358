"""The oracle's SeparatedReplayBuffer restatement against fixtures produced by the reference's own
class (oracle/make_golden_separated.py).  Bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle

FIELDS = ["share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
          "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions", "factor"]
BUF_FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
              "masks", "active_masks", "action_log_probs", "available_actions", "rewards")
CASES = [
    ("ff3", lambda b, a: b.feed_forward_generator(a, 3)),
    ("ff7", lambda b, a: b.feed_forward_generator(a, 7)),
    ("rec_L5", lambda b, a: b.recurrent_generator(a, 2, 5)),
    ("rec_L4", lambda b, a: b.recurrent_generator(a, 3, 4)),
    ("naive3", lambda b, a: b.naive_recurrent_generator(a, 3)),
]


class _Args(object):
    def __init__(self, **kw):
        self.gamma, self.gae_lambda, self.recurrent_N, self.hidden_size = 0.99, 0.95, 1, 8
        self.use_gae, self.use_popart, self.use_valuenorm, self.use_proper_time_limits = True, False, True, False
        self.__dict__.update(kw)


class Box(object):
    def __init__(self, shape):
        self.shape = shape


class Discrete(object):
    def __init__(self, n):
        self.n = n


class _Norm(object):
    """Value normaliser stand-in with the reference's attribute names (valuenorm.py:20-24)."""

    def __init__(self, n):
        self.running_mean, self.running_mean_sq, self.debiasing_term = [np.float32(x) for x in n]


def separated_returns_cases(gold):
    z = gold.npz("separated_cases")
    for m in gold.meta("separated_cases")["returns"]:
        yield z, m, "sret%03d_" % m["id"]


def test_separated_compute_returns(gold):
    n = 0
    for z, m, key in separated_returns_cases(gold):
        args = _Args(episode_length=m["T"], n_rollout_threads=m["N"], use_gae=m["use_gae"],
                     use_popart=m["use_popart"], use_valuenorm=m["use_valuenorm"],
                     use_proper_time_limits=m["use_proper_time_limits"])
        buf = oracle.OracleSeparatedBuffer(args, Box((3,)), Box((4,)), Discrete(5))
        for name in ("rewards", "masks", "bad_masks", "active_masks"):
            getattr(buf, name)[...] = z[key + name]
        buf.value_preds[...] = z[key + "value_preds_in"]
        vn = _Norm(z[key + "norm"]) if (key + "norm") in z else None
        buf.compute_returns(z[key + "next_value"], vn)
        np.testing.assert_array_equal(buf.returns, z[key + "returns"], err_msg=str(m))
        n += 1
    assert n >= 16


def oracle_separated_buffer(z):
    sh = z["sgen_buf_share_obs"].shape
    args = _Args(episode_length=sh[0] - 1, n_rollout_threads=sh[1], hidden_size=z["sgen_buf_rnn_states"].shape[-1])
    buf = oracle.OracleSeparatedBuffer(args, Box((z["sgen_buf_obs"].shape[-1],)), Box((sh[-1],)),
                                       Discrete(z["sgen_buf_available_actions"].shape[-1]))
    for name in BUF_FIELDS:
        getattr(buf, name)[...] = z["sgen_buf_" + name]
    return buf


@pytest.mark.parametrize("with_factor", [False, True])
@pytest.mark.parametrize("case,call", CASES)
def test_separated_generators(gold, case, call, with_factor):
    z = gold.npz("separated_cases")
    buf = oracle_separated_buffer(z)
    if with_factor:
        buf.update_factor(z["sgen_buf_factor"])
        case += "_factor"
    torch.manual_seed(9)
    batches = list(call(buf, z["sgen_buf_advantages"]))
    n = [m for m in gold.meta("separated_cases")["generators"] if m.get("case") == case][0]["n_batches"]
    assert len(batches) == n
    for bi, sample in enumerate(batches):
        assert len(sample) == (13 if with_factor else 12)
        for fname, arr in zip(FIELDS, sample):
            np.testing.assert_array_equal(arr, z["sgen_%s_b%d_%s" % (case, bi, fname)],
                                          err_msg="%s batch %d field %s" % (case, bi, fname))
