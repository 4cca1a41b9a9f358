"""Continuous (Box) and MultiDiscrete action heads, and image observations (CNNBase), through policy + trainer against fixtures produced by the
reference's R_MAPPOPolicy / R_MAPPO on the same seeds (oracle/make_golden_spaces.py): identical initial
parameters, evaluate_actions outputs, train_info and final parameters within float32 tolerance."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from oracle import oracle
from test_misc_cpu import _MultiDiscrete

from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO

BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "rewards")


DISCRETE = {"cnn": 4, "naive_gru": 5, "gru2_xavier": 5}      # cases with a Discrete head (+ availability masks)


def _space(cname):
    if cname in DISCRETE:
        return Discrete(DISCRETE[cname])
    return Box((3,)) if cname == "box" else _MultiDiscrete([3, 4])       # sub-action ranges [0, 2] and [0, 3]


def build_space_case(gold, cname, device=torch.device("cpu")):
    meta = gold.meta("space_cases")[cname]
    args = make_args(episode_length=meta["T"], n_rollout_threads=meta["N"], **meta["args"])
    spaces = (Box((meta["Do"],)), Box((meta["Ds"],)), _space(cname)) if cname != "cnn" else \
        (Box((3, 9, 9)), Box((3, 9, 9)), _space(cname))
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=device)
    trainer = R_MAPPO(args, policy, device=device)
    return meta, args, spaces, policy, trainer


def check_final(z, key, meta, info, policy, rel=2e-4, atol=2e-5):
    for k, v in meta["train_info"].items():
        assert info[k] == pytest.approx(v, rel=rel, abs=2e-6), (k, info[k], v)
    for prefix, module in ((key + "final_actor.", policy.actor), (key + "final_critic.", policy.critic)):
        sd = module.state_dict()
        keys = [k[len(prefix):] for k in z.files if k.startswith(prefix)]
        assert sorted(keys) == sorted(sd.keys())
        for k in keys:
            np.testing.assert_allclose(sd[k].cpu().numpy(), z[prefix + k], rtol=1e-4, atol=atol, err_msg=prefix + k)


@pytest.mark.parametrize("cname", ["box", "multidiscrete", "cnn", "naive_gru", "gru2_xavier"])
def test_other_action_heads_match_reference(gold, cname):
    z = gold.npz("space_cases")
    key = "spc_%s_" % cname
    meta, args, spaces, policy, trainer = build_space_case(gold, cname)
    for prefix, module in ((key + "init_actor.", policy.actor), (key + "init_critic.", policy.critic)):
        sd = module.state_dict()
        assert sorted(k[len(prefix):] for k in z.files if k.startswith(prefix)) == sorted(sd.keys())
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), z[prefix + k], err_msg=prefix + k)
    buf = oracle.OracleBuffer(args, meta["A"], *spaces)
    assert buf.actions.shape[-1] == meta["act_width"] and (buf.available_actions is None) == (cname not in DISCRETE)
    for name in BUF + (("available_actions",) if cname in DISCRETE else ()):
        getattr(buf, name)[...] = z[key + "buf_" + name]
    B = meta["N"] * meta["A"]
    flat = lambda x: x[0].reshape(B, *x.shape[3:])
    trainer.prep_rollout()
    with torch.no_grad():
        values, logp, ent = policy.evaluate_actions(flat(buf.share_obs), flat(buf.obs), flat(buf.rnn_states),
                                                    flat(buf.rnn_states_critic), flat(buf.actions), flat(buf.masks),
                                                    None if cname not in DISCRETE else flat(buf.available_actions),
                                                    flat(buf.active_masks))
    tol = dict(rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(values.numpy(), z[key + "eval_values"], **tol)
    np.testing.assert_allclose(logp.numpy(), z[key + "eval_logp"], **tol)
    np.testing.assert_allclose(float(ent), float(z[key + "eval_entropy"]), **tol)
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    check_final(z, key, meta, info, policy)
