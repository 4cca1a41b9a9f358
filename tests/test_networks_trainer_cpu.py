"""Product networks + trainer (the PyTorch part of the path, device-agnostic) against fixtures
produced by the reference's R_MAPPOPolicy / R_MAPPO on the same seeds (oracle/make_golden_trainer.py).
The minibatch source here is the host OracleBuffer (test infrastructure); the device buffer is
checked in the -m gpu tests."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from oracle import oracle

from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO

CASES = ["mlp", "mlp_relu", "gru", "mlp_nonorm"]
# hidden 64 (oracle/make_golden_trainer.py: CASES_H64): the width of every shipped MPE / SMAC configuration, the one the
# fused trunk kernels take on the device (tests/test_gpu_trainer_h64.py); here the same fixtures pin the torch modules
CASES_H64 = ["h64_ns", "h64_relu2", "h64_nofeat", "h64_odd", "h64_gru", "h64_gru_straddle"]
# the device-sampler route's fixtures (CASES_DEV: the reference fed K10's partition, oracle/k10_partition.py)
CASES_DEV = ["dev_relu2", "dev_tail", "dev_gru"]
ALL_CASES = CASES + CASES_H64


def _file(cname):
    if cname.startswith("dev_"):
        return "trainer_dev_cases"
    return "trainer_h64_cases" if cname.startswith("h64_") else "trainer_cases"


def _build(gold, cname):
    meta = gold.meta(_file(cname))[cname]
    spec = meta["spec"]
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], **spec["args"])
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces)
    trainer = R_MAPPO(args, policy)
    return meta, spec, args, spaces, policy, trainer


def _check_sd(z, prefix, module, rtol=0.0, atol=0.0):
    sd = module.state_dict()
    keys = [k[len(prefix):] for k in z.files if k.startswith(prefix)]
    assert sorted(keys) == sorted(sd.keys()), (sorted(keys), sorted(sd.keys()))
    for k in keys:
        np.testing.assert_allclose(sd[k].numpy(), z[prefix + k], rtol=rtol, atol=atol, err_msg=prefix + k)


@pytest.mark.parametrize("cname", ALL_CASES)
def test_init_matches_reference_seed(gold, cname):
    """Same seed => bit-identical initial weights and identical state_dict keys (checkpoint compat)."""
    z = gold.npz(_file(cname))
    _, _, _, _, policy, _ = _build(gold, cname)
    # (the 64-wide cases: orthogonal_ runs a multi-threaded LAPACK QR on [384, 64] matrices whose last bits depend on how
    # the host's BLAS happened to split the work -- seen once under load; the small ones are bit-identical)
    tol = dict(rtol=1e-5, atol=1e-6) if cname in CASES_H64 else {}
    _check_sd(z, "trn_%s_init_actor." % cname, policy.actor, **tol)
    _check_sd(z, "trn_%s_init_critic." % cname, policy.critic, **tol)


@pytest.mark.parametrize("cname", ALL_CASES)
def test_forward_matches_reference(gold, cname):
    z = gold.npz(_file(cname))
    key = "trn_%s_" % cname
    _, spec, args, _, policy, trainer = _build(gold, cname)
    B = spec["N"] * spec["A"]
    flat = lambda name: z[key + "buf_" + name][0].reshape(B, *z[key + "buf_" + name].shape[3:])
    trainer.prep_rollout()
    torch.manual_seed(11)
    with torch.no_grad():
        values, actions, logp, h_a, h_c = policy.get_actions(
            flat("share_obs"), flat("obs"), flat("rnn_states"), flat("rnn_states_critic"), flat("masks"),
            flat("available_actions"))
        # integer sampling: identical actions under the same CPU generator state
        np.testing.assert_array_equal(actions.numpy(), z[key + "act_actions"])
        tol = dict(rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(values.numpy(), z[key + "act_values"], **tol)
        np.testing.assert_allclose(logp.numpy(), z[key + "act_logp"], **tol)
        np.testing.assert_allclose(h_a.numpy(), z[key + "act_h_actor"], **tol)
        np.testing.assert_allclose(h_c.numpy(), z[key + "act_h_critic"], **tol)
        ev_values, ev_logp, ev_ent = policy.evaluate_actions(
            flat("share_obs"), flat("obs"), flat("rnn_states"), flat("rnn_states_critic"), flat("actions"),
            flat("masks"), flat("available_actions"), flat("active_masks"))
        np.testing.assert_allclose(ev_values.numpy(), z[key + "eval_values"], **tol)
        np.testing.assert_allclose(ev_logp.numpy(), z[key + "eval_logp"], **tol)
        np.testing.assert_allclose(float(ev_ent), float(z[key + "eval_entropy"]), **tol)


@pytest.mark.parametrize("cname", ALL_CASES + CASES_DEV)
def test_train_matches_reference(gold, cname):
    """compute_returns + R_MAPPO.train on the same seeds: same permutations, train_info and final
    parameters as the reference within float32 tolerance.  (``dev_*``: the permutation is K10's partition, drawn by the
    numpy restatement from the same generator state -- the recorded index lists must come out again.)"""
    import contextlib
    from oracle.k10_partition import RandpermAsK10
    z = gold.npz(_file(cname))
    key = "trn_%s_" % cname
    meta, spec, args, spaces, policy, trainer = _build(gold, cname)
    buf = oracle.OracleBuffer(args, spec["A"], *spaces)
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks",
                 "bad_masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        getattr(buf, name)[...] = z[key + "buf_" + name]
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns, z[key + "returns"])

    trainer.prep_training()
    torch.manual_seed(21)
    sampler = RandpermAsK10(spec["args"]["num_mini_batch"]) if cname in CASES_DEV else contextlib.nullcontext()
    with sampler as rec:
        info = trainer.train(buf)
    if cname in CASES_DEV:
        assert len(rec.calls) == meta["n_perms"]
        for i, p in enumerate(rec.calls):
            np.testing.assert_array_equal(p, z[key + "perm%d" % i])
    ref_info = meta["train_info"]
    assert set(info) == set(ref_info)
    for k in ref_info:
        assert info[k] == pytest.approx(ref_info[k], rel=2e-4, abs=2e-6), (k, info[k], ref_info[k])
    # Adam's first steps are ~lr-sized and sign-like, so weights move by ~lr regardless of tiny
    # gradient differences: compare with an absolute tolerance well below lr (5e-4 .. 7e-4)
    _check_sd(z, key + "final_actor.", policy.actor, rtol=1e-4, atol=2e-5)
    _check_sd(z, key + "final_critic.", policy.critic, rtol=1e-4, atol=2e-5)
    if trainer.value_normalizer is not None:
        vn = trainer.value_normalizer
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        np.testing.assert_allclose(got, z[key + "final_norm"], rtol=1e-5, atol=1e-9)
    if cname in CASES_H64 + CASES_DEV:      # what the last ppo_update left in .grad (clipped), relative to each tensor's largest entry
        for net, pre in ((policy.actor, "last_grad_actor."), (policy.critic, "last_grad_critic.")):
            for k, p in net.named_parameters():
                ref = z[key + pre + k]
                np.testing.assert_allclose(p.grad.numpy(), ref, rtol=0, atol=2e-4 * max(1e-12, np.abs(ref).max()),
                                           err_msg=pre + k)


@pytest.mark.parametrize("cname", ["mlp", "gru"])
def test_row_span_microbatching_is_equivalent(gold, cname):
    """Minibatches too large for PyTorch-ROCm's 32-bit row kernels are evaluated in spans with gradient
    accumulation -- row spans for feed-forward minibatches, spans of whole chunks (all L steps of a chunk
    stay together) for recurrent ones; that must give the same update as one pass (float32 summation
    order aside)."""
    z = gold.npz("trainer_cases")
    key = "trn_%s_" % cname
    results = []
    for cap in (1 << 30, 200):     # 200 elements / 11 features -> spans of 18 rows (3 chunks of 5 steps)
        meta, spec, args, spaces, policy, trainer = _build(gold, cname)
        trainer.MAX_TENSOR_ELEMENTS = cap
        buf = oracle.OracleBuffer(args, spec["A"], *spaces)
        for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks",
                     "bad_masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
            getattr(buf, name)[...] = z[key + "buf_" + name]
        buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(21)
        info = trainer.train(buf)
        results.append((info, {k: v.clone() for k, v in policy.actor.state_dict().items()},
                        {k: v.clone() for k, v in policy.critic.state_dict().items()}))
    (i0, a0, c0), (i1, a1, c1) = results
    for k in i0:
        assert i1[k] == pytest.approx(i0[k], rel=1e-4, abs=1e-6), k
    for k in a0:
        np.testing.assert_allclose(a1[k].numpy(), a0[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    for k in c0:
        np.testing.assert_allclose(c1[k].numpy(), c0[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


class _StandardizingOracleBuffer(oracle.OracleBuffer):
    """Host buffer whose samplers hand out row-standardised observations, like the device buffer's
    standardize_obs=True mode (test infrastructure)."""
    supports_standardized_obs = True

    @staticmethod
    def _std(x):
        x64 = x.astype(np.float64)
        mu = x64.mean(-1, keepdims=True)
        var = x64.var(-1, keepdims=True)
        return ((x64 - mu) / np.sqrt(var + 1e-5)).astype(np.float32)

    def _wrap(self, gen, standardize_obs):
        for sample in gen:
            if standardize_obs:
                sample = (self._std(sample[0]), self._std(sample[1])) + tuple(sample[2:])
            yield sample

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, standardize_obs=False):
        return self._wrap(super().feed_forward_generator(advantages, num_mini_batch, mini_batch_size), standardize_obs)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, standardize_obs=False):
        return self._wrap(super().recurrent_generator(advantages, num_mini_batch, data_chunk_length), standardize_obs)


@pytest.mark.parametrize("cname", ["mlp", "gru"])
def test_folded_input_layernorm_is_equivalent(gold, cname):
    """Sampler-side row standardisation + the LayerNorm affine folded into the first Linear gives the
    same update as LayerNorm inside the network (float32 rounding aside)."""
    z = gold.npz("trainer_cases")
    key = "trn_%s_" % cname
    results = []
    for cls in (oracle.OracleBuffer, _StandardizingOracleBuffer):
        meta, spec, args, spaces, policy, trainer = _build(gold, cname)
        buf = cls(args, spec["A"], *spaces)
        for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks",
                     "bad_masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
            getattr(buf, name)[...] = z[key + "buf_" + name]
        buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
        assert policy.can_fold_input_norm()
        trainer.prep_training()
        torch.manual_seed(21)
        info = trainer.train(buf)
        results.append((info, {k: v.clone() for k, v in policy.actor.state_dict().items()},
                        {k: v.clone() for k, v in policy.critic.state_dict().items()}))
    (i0, a0, c0), (i1, a1, c1) = results
    for k in i0:
        assert i1[k] == pytest.approx(i0[k], rel=2e-4, abs=2e-6), k
    for sd0, sd1 in ((a0, a1), (c0, c1)):
        for k in sd0:
            np.testing.assert_allclose(sd1[k].numpy(), sd0[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
