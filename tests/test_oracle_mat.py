"""The oracle's restatement of the MAT hooks (transformer branches of compute_returns and
feed_forward_generator_transformer) against fixtures produced by the reference's SharedReplayBuffer with
algorithm_name "mat" / "mat_dec" (oracle/make_golden_mat.py).  Bit-exact, including numpy's pairwise
float32 mean over the agent axis."""
import numpy as np
import pytest
import torch

from oracle import oracle
from test_oracle_separated import _Args, Box, Discrete, _Norm

FIELDS = ["share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
          "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions"]
BUF_FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
              "masks", "active_masks", "action_log_probs", "available_actions", "rewards")


def mat_returns_cases(gold):
    z = gold.npz("mat_cases")
    for m in gold.meta("mat_cases")["returns"]:
        yield z, m, "mat%03d_" % m["id"]


def test_mat_compute_returns(gold):
    n_mean = 0
    for z, m, key in mat_returns_cases(gold):
        args = _Args(episode_length=m["T"], n_rollout_threads=m["N"], use_gae=m["use_gae"],
                     use_valuenorm=m["use_valuenorm"], use_proper_time_limits=m["use_proper_time_limits"],
                     algorithm_name=m["algo"])
        buf = oracle.OracleBuffer(args, m["A"], Box((3,)), Box((4,)), Discrete(5))
        for name in ("rewards", "masks", "bad_masks", "active_masks"):
            getattr(buf, name)[...] = z[key + name]
        buf.value_preds[...] = z[key + "value_preds_in"]
        vn = _Norm(z[key + "norm"]) if (key + "norm") in z else None
        buf.compute_returns(z[key + "next_value"], vn)
        np.testing.assert_array_equal(buf.returns, z[key + "returns"], err_msg=str(m))
        np.testing.assert_array_equal(buf.advantages, z[key + "advantages"], err_msg=str(m))
        n_mean += int(m["use_gae"] and not m["use_proper_time_limits"] and not m["use_valuenorm"])
    assert n_mean >= 5      # the agent-mean branch at A = 2, 3, 8, 9, 17


@pytest.mark.parametrize("case", ["tf1", "tf2", "tf7"])
def test_transformer_generator(gold, case):
    z = gold.npz("mat_cases")
    meta = [m for m in gold.meta("mat_cases")["generators"] if m.get("case") == case][0]
    sh = z["mgen_buf_share_obs"].shape
    args = _Args(episode_length=sh[0] - 1, n_rollout_threads=sh[1], hidden_size=z["mgen_buf_rnn_states"].shape[-1],
                 algorithm_name="mat")
    buf = oracle.OracleBuffer(args, sh[2], Box((z["mgen_buf_obs"].shape[-1],)), Box((sh[-1],)),
                              Discrete(z["mgen_buf_available_actions"].shape[-1]))
    for name in BUF_FIELDS:
        getattr(buf, name)[...] = z["mgen_buf_" + name]
    torch.manual_seed(4)
    batches = list(buf.feed_forward_generator_transformer(z["mgen_buf_advantages"], meta["num_mini_batch"]))
    assert len(batches) == meta["n_batches"]
    for bi, sample in enumerate(batches):
        for fname, arr in zip(FIELDS, sample):
            np.testing.assert_array_equal(arr, z["mgen_%s_b%d_%s" % (case, bi, fname)],
                                          err_msg="%s batch %d field %s" % (case, bi, fname))
