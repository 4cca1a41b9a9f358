"""The PyTorch part of the path (networks + trainer on the host oracle buffer) against the reference-generated end-to-end
fixtures at the layer shapes of BASELINE.json configs[3] (SMAC MMM2) and configs[4] (Hanabi-Full, hidden 512 x 2):
tests/golden/trainer_cfg_cases.npz (oracle/make_golden_trainer.py: CASES_CFG; tests/cfg_shapes.py).  The device twin is
tests/test_gpu_cfg_shapes.py."""
import contextlib

import numpy as np
import pytest
import torch

import cfg_shapes as C
from oracle import oracle
from oracle.k10_partition import RandpermAsK10


@pytest.mark.parametrize("cname", C.CASES)
def test_train_matches_reference_at_baseline_config_shapes(gold, cname):
    z, key, meta, spec, args, spaces, policy, trainer = C.build(gold, cname)
    C.start_from_reference_weights(policy, z, key, rtol=1e-5, atol=1e-6)
    arrays, nv = C.inputs(spec, z, key)
    np.testing.assert_array_equal(nv, z[key + "next_value"])
    buf = oracle.OracleBuffer(args, spec["A"], *spaces)
    for name, arr in arrays.items():
        getattr(buf, name)[...] = arr

    # one rollout-side and one update-side forward on step 0 (what the generator recorded before training)
    B = spec["N"] * spec["A"]
    flat = lambda name: arrays[name][0].reshape(B, *arrays[name].shape[3:]) if name in arrays else \
        np.zeros((B,) + getattr(buf, name).shape[3:], np.float32)
    trainer.prep_rollout()
    torch.manual_seed(11)
    with torch.no_grad():
        values, actions, logp, _, _ = policy.get_actions(flat("share_obs"), flat("obs"), flat("rnn_states"),
                                                         flat("rnn_states_critic"), flat("masks"), flat("available_actions"))
        np.testing.assert_array_equal(actions.numpy(), z[key + "act_actions"])      # integer sampling: identical
        tol = dict(rtol=5e-5, atol=5e-6)
        np.testing.assert_allclose(values.numpy(), z[key + "act_values"], **tol)
        np.testing.assert_allclose(logp.numpy(), z[key + "act_logp"], **tol)
        ev_values, ev_logp, ev_ent = policy.evaluate_actions(
            flat("share_obs"), flat("obs"), flat("rnn_states"), flat("rnn_states_critic"), flat("actions"), flat("masks"),
            flat("available_actions"), flat("active_masks"))
        np.testing.assert_allclose(ev_values.numpy(), z[key + "eval_values"], **tol)
        np.testing.assert_allclose(ev_logp.numpy(), z[key + "eval_logp"], **tol)
        np.testing.assert_allclose(float(ev_ent), float(z[key + "eval_entropy"]), **tol)

    buf.compute_returns(nv, trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns, z[key + "returns"])
    trainer.prep_training()
    torch.manual_seed(21)
    sampler = RandpermAsK10(spec["args"]["num_mini_batch"]) if spec["k10"] else contextlib.nullcontext()
    with sampler as rec:
        info = trainer.train(buf)
    if spec["k10"]:
        assert len(rec.calls) == meta["n_perms"]
        for i, p in enumerate(rec.calls):
            np.testing.assert_array_equal(p, z[key + "perm%d" % i])
    for k, ref in meta["train_info"].items():
        assert info[k] == pytest.approx(ref, rel=3e-4, abs=2e-6), (k, info[k], ref)
    C.check_weights(z, key + "final_actor.", policy.actor, rtol=1e-4, atol=2e-5)
    C.check_weights(z, key + "final_critic.", policy.critic, rtol=1e-4, atol=2e-5)
    C.check_grads(z, key + "last_grad_actor.", policy.actor, rel=3e-4)
    C.check_grads(z, key + "last_grad_critic.", policy.critic, rel=3e-4)
    vn = trainer.value_normalizer
    got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
    np.testing.assert_allclose(got, z[key + "final_norm"], rtol=1e-5, atol=1e-9)
