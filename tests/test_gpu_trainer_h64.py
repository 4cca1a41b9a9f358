"""-m gpu: the update route the bench times -- lazy-obs sampler -> row table -> standardised copy -> K9 forward -> K7 loss
-> K9 backward -> Adam -- against what the REFERENCE's R_MAPPO.train produced at hidden size 64
(tests/golden/trainer_h64_cases.npz, written by oracle/make_golden_trainer.py: CASES_H64 from the reference's
algorithms/utils/mlp.py:6-58, act.py:44-60, r_actor_critic.py:147-175 driven by r_mappo.py:91-169).

Every case asserts that the fused trunk kernels were actually launched (forward AND backward), so a silent fall-back to
the PyTorch modules cannot pass.  Tolerances: tests/parity.py, about three times the worst deviation measured on the MI355X --
losses 7e-6 relative, weights 3e-5 absolute (an Adam step moves a weight by ~lr = 5e-4 .. 7e-4 whatever the gradient's
size -- also when the gradient is noise: 4 % of a step), the gradients the last update left in ``.grad`` to 1.5e-4 of each
tensor's largest entry."""
import numpy as np
import pytest

import parity
import torch

from helpers import Box, Discrete, assert_k9_carried_the_updates, make_args

pytestmark = pytest.mark.gpu

CASES = ["h64_ns", "h64_relu2", "h64_nofeat", "h64_odd", "h64_gru", "h64_gru_straddle"]
FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
          "active_masks", "action_log_probs", "available_actions", "rewards")


def _setup(gold, cname, dev, **extra):
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    z = gold.npz("trainer_h64_cases")
    meta = gold.meta("trainer_h64_cases")[cname]
    spec = meta["spec"]
    kw = dict(spec["args"])
    kw.update(extra)
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], **kw)
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, spec["A"], *spaces, device=dev)
    key = "trn_%s_" % cname
    for name in FIELDS:
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    return z, key, meta, spec, policy, trainer, buf


def _launches():
    from onpolicy.algorithms.utils import fused_mlp
    t = fused_mlp.profile_times()
    return t.get("mappo_mlp_forward", (0,))[0], t.get("mappo_mlp_backward", (0,))[0]


@pytest.mark.parametrize("graph", ["1", "0"], ids=["update_graph", "eager"])
@pytest.mark.parametrize("cname", CASES)
def test_fused_trunk_update_vs_reference(gold, cname, graph, monkeypatch, margins):
    """compute_returns + R_MAPPO.train at hidden 64 with the reference's permutations (sampler_rng=host, same CPU seed) -- with
    ppo_update replayed from a captured HIP graph after its first occurrence (the default) and all-eager."""
    from onpolicy.algorithms.utils import fused_mlp
    monkeypatch.setenv("MAPPO_UPDATE_GRAPH", graph)
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, cname, dev, sampler_rng="host")
    for net, pre in ((policy.actor, "init_actor."), (policy.critic, "init_critic.")):
        for k, v in net.state_dict().items():       # host LAPACK QR of orthogonal_: last-bit differences across boxes
            np.testing.assert_allclose(v.cpu().numpy(), z[key + pre + k], rtol=1e-4, atol=5e-6)
            v.copy_(torch.from_numpy(z[key + pre + k]))     # ... so start the update from the reference's exact weights
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"])
    trainer.prep_training()
    torch.manual_seed(21)
    fused_mlp.profile(True)
    try:
        info = trainer.train(buf)
        torch.cuda.synchronize()
        n_fwd, n_bwd = _launches()
    finally:
        fused_mlp.profile(False)
    # the route under test: one K9 forward + one K9 backward launch per network and update
    updates = spec["args"]["ppo_epoch"] * spec["args"]["num_mini_batch"]
    assert_k9_carried_the_updates(trainer, n_fwd, n_bwd, updates)
    buf.after_update()

    worst = parity.compare_update(z, key, meta, policy, trainer, info)
    margins("trainer_h64/%s/%s" % (cname, "graph" if graph == "1" else "eager"), worst)
    top = parity.top3(worst)
    print("\n[%s] K9 launches fwd %d bwd %d; largest relative errors: %s" % (cname, n_fwd, n_bwd, top))


@pytest.mark.parametrize("cname", ["h64_ns", "h64_gru"])
def test_forward_on_the_fused_route_vs_reference(gold, cname):
    """evaluate_actions through K9 (lazy rows of buffer step 0) against the reference's evaluate_actions outputs."""
    from onpolicy.algorithms.utils import fused_mlp
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, cname, dev, sampler_rng="host")
    for net, pre in ((policy.actor, "init_actor."), (policy.critic, "init_critic.")):
        for k, v in net.state_dict().items():
            v.copy_(torch.from_numpy(z[key + pre + k]))
    B = spec["N"] * spec["A"]
    trainer.prep_rollout()
    idx = torch.arange(B, device=dev)
    fold = policy.can_fold_input_norm()
    share = fused_mlp.RowSource(buf._obs_rows("share_obs", fold), idx, None, standardized=fold, width=spec["Ds"])
    obs = fused_mlp.RowSource(buf._obs_rows("obs", fold), idx, None, standardized=fold, width=spec["Do"])
    flat = lambda name: getattr(buf, name)[0].reshape(B, *getattr(buf, name).shape[3:])
    fused_mlp.profile(True)
    try:
        with torch.no_grad():
            values, logp, ent = policy.evaluate_actions(
                share, obs, flat("rnn_states"), flat("rnn_states_critic"), flat("actions"), flat("masks"),
                flat("available_actions"), flat("active_masks"), **({"obs_standardized": True} if fold else {}))
        torch.cuda.synchronize()
        n_fwd, _ = _launches()
    finally:
        fused_mlp.profile(False)
    assert n_fwd == 2, n_fwd
    tol = dict(rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(values.cpu().numpy(), z[key + "eval_values"], **tol)
    np.testing.assert_allclose(logp.cpu().numpy(), z[key + "eval_logp"], **tol)
    np.testing.assert_allclose(float(ent), float(z[key + "eval_entropy"]), **tol)
