"""HATRPO trainer (product, device-agnostic PyTorch) against fixtures produced by the reference's HATRPO on its
SeparatedReplayBuffer with a factor (oracle/make_golden_hatrpo.py): critic Adam step, conjugate-gradient direction,
backtracking line search -- final parameters, normaliser statistics and the seven logged scalars of one train()."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from oracle import oracle

from onpolicy.algorithms.hatrpo import hatrpo_trainer as ht
from onpolicy.algorithms.hatrpo.hatrpo_trainer import HATRPO
from onpolicy.algorithms.hatrpo.policy import HATRPO_Policy

CASES = ["mlp", "mlp_popart", "mlp_nonorm", "gru", "rejected"]
BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "available_actions", "rewards")


def _build(gold, cname):
    meta = gold.meta("hatrpo_cases")[cname]
    spec = meta["spec"]
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], **spec["args"])
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["act"][1])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = HATRPO_Policy(args, *spaces)
    return meta, spec, args, spaces, policy, HATRPO(args, policy)


@pytest.mark.parametrize("cname", CASES)
def test_hatrpo_train_matches_reference(gold, cname, capsys):
    z = gold.npz("hatrpo_cases")
    key = "hat_%s_" % cname
    meta, spec, args, spaces, policy, trainer = _build(gold, cname)
    for prefix, module in ((key + "init_actor.", policy.actor), (key + "init_critic.", policy.critic)):
        for k, v in module.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), z[prefix + k], err_msg=prefix + k)
    buf = oracle.OracleSeparatedBuffer(args, *spaces)
    for name in BUF:
        getattr(buf, name)[...] = z[key + "buf_" + name]
    # the 6-tuple of HATRPO_Policy.evaluate_actions
    flat = lambda a: a.reshape(-1, *a.shape[2:])      # noqa: E731
    with torch.no_grad():
        ev = policy.evaluate_actions(flat(buf.share_obs[:2]), flat(buf.obs[:2]), flat(buf.rnn_states[0:1]),
                                     flat(buf.rnn_states_critic[0:1]), flat(buf.actions[:2]), flat(buf.masks[:2]),
                                     flat(buf.available_actions[:2]), torch.from_numpy(flat(buf.active_masks[:2])))
    assert len(ev) == 6
    for name, t in zip(("values", "logp", "entropy", "mean", "std", "logits"), ev):
        np.testing.assert_allclose(t.numpy(), z[key + "eval_" + name], rtol=1e-5, atol=1e-6, err_msg=name)
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns, z[key + "returns"])
    buf.update_factor(z[key + "factor"])
    trainer.prep_training()
    before = {k: v.clone() for k, v in policy.actor.state_dict().items()}
    torch.manual_seed(21)
    info = trainer.train(buf)
    ref_info = meta["train_info"]
    assert set(info) == set(ref_info)
    for k in ref_info:
        assert info[k] == pytest.approx(ref_info[k], rel=2e-3, abs=2e-5), (k, info[k], ref_info[k])
    for prefix, module in ((key + "final_actor.", policy.actor), (key + "final_critic.", policy.critic)):
        for k, v in module.state_dict().items():
            np.testing.assert_allclose(v.numpy(), z[prefix + k], rtol=2e-3, atol=1e-4, err_msg=prefix + k)
    if trainer.value_normalizer is not None:
        vn = trainer.value_normalizer
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        np.testing.assert_allclose(got, z[key + "final_norm"], rtol=1e-5, atol=1e-9)
    if cname == "rejected":       # no backtracking step accepted: the actor is exactly where it started
        assert "does not impove" in capsys.readouterr().out
        for k, v in policy.actor.state_dict().items():
            assert torch.equal(v, before[k]), k
    else:
        assert meta["actor_moved"] > 1e-3


def test_fisher_operator_is_the_hessian_of_the_kl_plus_damping():
    """fvp(v) from the single retained graph == Hessian-vector product of mean KL(pi_0 || pi_theta) at theta_0 computed
    independently (finite differences of the KL gradient) + 0.1 v; conjugate gradients solve F x = b with it."""
    args = make_args(algorithm_name="hatrpo", hidden_size=8, layer_N=1)
    torch.manual_seed(0)
    policy = HATRPO_Policy(args, Box((5,)), Box((6,)), Discrete(4))
    trainer = HATRPO(args, policy)
    actor = policy.actor.double()
    params = list(actor.parameters())
    obs = torch.randn(40, 5, dtype=torch.float64)
    actor.tpdv = dict(dtype=torch.float64, device=torch.device("cpu"))
    act = torch.randint(0, 4, (40, 1)).double()

    def dist_at():
        logp, ent, mean, std, logits = actor.evaluate_actions(obs, torch.zeros(40, 1, 8, dtype=torch.float64), act,
                                                              torch.ones(40, 1, dtype=torch.float64))
        return ht._Dist(mean, std, logits)
    here = dist_at()
    fvp = trainer._fisher_operator(here, params)
    v = torch.randn(sum(p.numel() for p in params), dtype=torch.float64)
    old = here.detach()
    theta = torch.nn.utils.parameters_to_vector(params).detach().clone()

    def kl_grad(at):
        torch.nn.utils.vector_to_parameters(at, params)
        kl = ht.kl_divergence(dist_at(), old).mean()
        return ht._flat(torch.autograd.grad(kl, params, allow_unused=True), params)
    eps = 1e-5
    hv = (kl_grad(theta + eps * v) - kl_grad(theta - eps * v)) / (2 * eps)
    torch.nn.utils.vector_to_parameters(theta, params)
    torch.testing.assert_close(fvp(v), hv + 0.1 * v, rtol=1e-5, atol=1e-7)
    b = torch.randn_like(v)
    x = trainer.conjugate_gradient(fvp, b, nsteps=200, residual_tol=1e-24)
    torch.testing.assert_close(fvp(x), b, rtol=1e-6, atol=1e-8)


def test_gaussian_kl_matches_torch_distributions():
    """The closed form used for Box heads (the reference's own Box path raises before reaching it, see
    oracle/make_golden_hatrpo.py) against torch.distributions.kl_divergence, summed over action dimensions."""
    torch.manual_seed(1)
    mu0, mu1 = torch.randn(9, 3), torch.randn(9, 3)
    s0, s1 = torch.rand(9, 3) + 0.2, torch.rand(9, 3) + 0.2
    got = ht.kl_divergence(ht._Dist(mu1, s1, None), ht._Dist(mu0, s0, None))
    exp = torch.distributions.kl_divergence(torch.distributions.Normal(mu0, s0),
                                            torch.distributions.Normal(mu1, s1)).sum(1, keepdim=True)
    torch.testing.assert_close(got, exp, rtol=1e-5, atol=1e-6)


def test_box_actions_take_a_trust_region_step():
    args = make_args(algorithm_name="hatrpo", hidden_size=8, layer_N=1, num_mini_batch=1, episode_length=6,
                     n_rollout_threads=4)
    spaces = Box((5,)), Box((6,)), Box((2,))
    torch.manual_seed(0)
    policy = HATRPO_Policy(args, *spaces)
    trainer = HATRPO(args, policy)
    buf = oracle.OracleSeparatedBuffer(args, *spaces)
    rng = np.random.default_rng(0)
    for name in ("share_obs", "obs", "rewards", "actions"):
        getattr(buf, name)[...] = rng.standard_normal(getattr(buf, name).shape).astype(np.float32)
    buf.action_log_probs[...] = -1.5
    buf.compute_returns(np.zeros((4, 1), np.float32), trainer.value_normalizer)
    buf.update_factor(np.ones((6, 4, 1), np.float32))
    before = torch.nn.utils.parameters_to_vector(policy.actor.parameters()).clone()
    info = trainer.train(buf)
    assert all(np.isfinite(v) for v in info.values()) and 0 < info["kl"] < args.kl_threshold
    assert not torch.equal(torch.nn.utils.parameters_to_vector(policy.actor.parameters()), before)
