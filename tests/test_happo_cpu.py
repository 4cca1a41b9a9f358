"""HAPPO trainer (product, device-agnostic PyTorch) against fixtures produced by the reference's HAPPO
on its SeparatedReplayBuffer with a factor (oracle/make_golden_happo.py).  The minibatch source here is
the host OracleSeparatedBuffer (test infrastructure); the device buffer is checked in the -m gpu tests."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from oracle import oracle

from onpolicy.algorithms.happo.happo_trainer import HAPPO
from onpolicy.algorithms.happo.policy import HAPPO_Policy

CASES = ["mlp", "mlp_popart", "mlp_nonorm", "gru"]
BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "available_actions", "rewards")


def build_happo(gold, cname, device=torch.device("cpu")):
    meta = gold.meta("happo_cases")[cname]
    spec = meta["spec"]
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], **spec["args"])
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = HAPPO_Policy(args, *spaces, device=device)
    trainer = HAPPO(args, policy, device=device)
    return meta, spec, args, spaces, policy, trainer


def check_happo_result(z, key, meta, info, policy, trainer, rel=2e-4, atol=2e-5):
    ref_info = meta["train_info"]
    assert set(info) == set(ref_info)
    for k in ref_info:
        assert info[k] == pytest.approx(ref_info[k], rel=rel, abs=2e-6), (k, info[k], ref_info[k])
    for prefix, module in ((key + "final_actor.", policy.actor), (key + "final_critic.", policy.critic)):
        sd = module.state_dict()
        keys = [k[len(prefix):] for k in z.files if k.startswith(prefix)]
        assert sorted(keys) == sorted(sd.keys())
        for k in keys:
            np.testing.assert_allclose(sd[k].cpu().numpy(), z[prefix + k], rtol=1e-4, atol=atol, err_msg=prefix + k)
    if trainer.value_normalizer is not None:
        vn = trainer.value_normalizer
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        np.testing.assert_allclose(got, z[key + "final_norm"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cname", CASES)
def test_happo_train_matches_reference(gold, cname):
    z = gold.npz("happo_cases")
    key = "hap_%s_" % cname
    meta, spec, args, spaces, policy, trainer = build_happo(gold, cname)
    for prefix, module in ((key + "init_actor.", policy.actor), (key + "init_critic.", policy.critic)):
        for k, v in module.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z[prefix + k], err_msg=prefix + k)
    buf = oracle.OracleSeparatedBuffer(args, *spaces)
    for name in BUF:
        getattr(buf, name)[...] = z[key + "buf_" + name]
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns, z[key + "returns"])
    buf.update_factor(z[key + "factor"])
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    check_happo_result(z, key, meta, info, policy, trainer)


def test_happo_factor_scales_the_policy_gradient(gold):
    """factor == c everywhere scales the surrogate by c: with entropy_coef 0 and gradient clipping off the
    first actor gradient is c times the factor-1 gradient (happo_trainer.py:137-141)."""
    z = gold.npz("happo_cases")
    key = "hap_mlp_"
    grads = []
    for c in (1.0, 3.0):
        meta, spec, args, spaces, policy, trainer = build_happo(gold, "mlp")
        trainer.entropy_coef, trainer._use_max_grad_norm = 0.0, False
        buf = oracle.OracleSeparatedBuffer(args, *spaces)
        for name in BUF:
            getattr(buf, name)[...] = z[key + "buf_" + name]
        buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
        buf.update_factor(np.full((spec["T"], spec["N"], 1), c, dtype=np.float32))
        torch.manual_seed(3)
        sample = next(iter(buf.feed_forward_generator(np.ones((spec["T"], spec["N"], 1), np.float32), 1)))
        trainer.prep_training()
        trainer.ppo_update(sample)
        grads.append(torch.cat([p.grad.reshape(-1) for p in policy.actor.parameters() if p.grad is not None]))
    np.testing.assert_allclose(grads[1].numpy(), 3.0 * grads[0].numpy(), rtol=1e-4, atol=1e-7)
