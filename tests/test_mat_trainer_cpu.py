"""The Multi-Agent Transformer (onpolicy/algorithms/mat/*) against the reference's classes
(fixtures: oracle/make_golden_mat_trainer.py): identical initial weights under a seed, identical sampled and
deterministic actions, log-probs / values / entropies to float tolerance, and one MATTrainer.train() on the same
rollout (oracle buffer with the mat branches of compute_returns) ending in the reference's parameters."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, load_into, make_args
from oracle import oracle

from onpolicy.algorithms.mat.algorithm.transformer_policy import TransformerPolicy
from onpolicy.algorithms.mat.mat_trainer import MATTrainer

T, N, Do, Ds = 5, 4, 7, 9
INFO_KEYS = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")


class _BoundedBox(Box):
    def __init__(self, shape):
        Box.__init__(self, shape)
        self.high = np.ones(shape, dtype=np.float32)


_BoundedBox.__name__ = "Box"


def _parse(extra):
    out = {}
    for kv in filter(None, extra.split(",")):
        k, v = kv.split("=")
        out[k] = {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    return out


def _cases(gold):
    z = gold.npz("mat_trainer_cases")
    return z, [c.split("|") for c in z["cases"]]


def _build(z, name, algo, extra):
    A, k, is_box = (int(v) for v in z["mt_%s_spec" % name])
    args = make_args(episode_length=T, n_rollout_threads=N, algorithm_name=algo, ppo_epoch=2, num_mini_batch=2,
                     **_parse(extra))
    act_space = _BoundedBox((k,)) if is_box else Discrete(k)
    torch.manual_seed(11)
    np.random.seed(11)
    policy = TransformerPolicy(args, Box((Do,)), Box((Ds,)), act_space, A)
    return args, act_space, A, policy


def test_case_list(gold):
    z, cases = _cases(gold)
    assert [c[0] for c in cases] == ["discrete", "discrete_deep", "encode_state", "mat_dec", "dec_actor_own", "continuous"]


@pytest.mark.parametrize("idx", range(6))
def test_policy_and_trainer_match_reference(gold, idx):
    z, cases = _cases(gold)
    name, algo, extra = cases[idx]
    key = "mt_%s_" % name
    args, act_space, A, policy = _build(z, name, algo, extra)

    # same seed -> same weights, same state-dict layout (checkpoints are interchangeable)
    state = policy.transformer.state_dict()
    ref_keys = sorted(k[len(key) + 5:] for k in z.files if k.startswith(key + "init_"))
    assert sorted(state) == ref_keys
    for pname, p in state.items():
        np.testing.assert_array_equal(p.numpy(), z[key + "init_" + pname], err_msg=pname)

    buf = oracle.OracleBuffer(args, A, Box((Do,)), Box((Ds,)), act_space)
    load_into(buf, {f[len(key) + 4:]: z[f] for f in z.files if f.startswith(key + "buf_")})
    rows = lambda a: np.concatenate(a)          # noqa: E731
    avail = rows(buf.available_actions[0]) if buf.available_actions is not None else None
    policy.eval()
    for mode, det in (("sample", False), ("det", True)):
        torch.manual_seed(23)
        with torch.no_grad():
            values, actions, logp, rs, rc = policy.get_actions(rows(buf.share_obs[0]), rows(buf.obs[0]),
                                                               rows(buf.rnn_states[0]), rows(buf.rnn_states_critic[0]),
                                                               rows(buf.masks[0]), avail, det)
        if buf.available_actions is not None:
            np.testing.assert_array_equal(actions.numpy(), z[key + mode + "_actions"])
            assert actions.dtype == torch.int64
        else:
            np.testing.assert_allclose(actions.numpy(), z[key + mode + "_actions"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(logp.numpy(), z[key + mode + "_logp"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(values.numpy(), z[key + mode + "_values"], rtol=1e-5, atol=1e-6)
        assert rs.shape == buf.rnn_states[0].reshape(-1, *buf.rnn_states.shape[3:]).shape
    with torch.no_grad():
        got = policy.get_values(rows(buf.share_obs[0]), rows(buf.obs[0]), rows(buf.rnn_states_critic[0]),
                                rows(buf.masks[0]))
        np.testing.assert_allclose(got.numpy(), z[key + "get_values"], rtol=1e-5, atol=1e-6)
        ev = policy.evaluate_actions(rows(buf.share_obs[1]), rows(buf.obs[1]), None, None, rows(buf.actions[1]), None,
                                     rows(buf.available_actions[1]) if avail is not None else None,
                                     torch.from_numpy(rows(buf.active_masks[1])))
    for g, n in zip(ev, ("eval_values", "eval_logp", "eval_entropy")):
        np.testing.assert_allclose(g.numpy(), z[key + n], rtol=1e-4, atol=1e-5, err_msg=n)

    trainer = MATTrainer(args, policy, A)
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns, z[key + "returns"])
    np.testing.assert_array_equal(buf.advantages, z[key + "advantages"])
    torch.manual_seed(31)
    info = trainer.train(buf)
    np.testing.assert_allclose([info[k] for k in INFO_KEYS], z[key + "info"], rtol=2e-3, atol=2e-4)
    for pname, p in policy.transformer.state_dict().items():
        np.testing.assert_allclose(p.numpy(), z[key + "final_" + pname], rtol=2e-3, atol=2e-4, err_msg=pname)
    if trainer.value_normalizer is not None:
        vn = trainer.value_normalizer
        np.testing.assert_allclose([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)],
                                   z[key + "norm"], rtol=1e-5)


def test_incremental_decoding_equals_whole_sequence_passes():
    """Acting with cached keys / values gives the rows the reference's per-agent full decoder passes give: a plain
    callable decoder (no ``begin``) takes the reference's route through the same action functions."""
    from onpolicy.algorithms.utils import transformer_act
    args = make_args(algorithm_name="mat", n_block=2, n_embd=16, n_head=2)
    A, na = 5, 6
    torch.manual_seed(0)
    policy = TransformerPolicy(args, Box((Do,)), Box((Ds,)), Discrete(na), A)
    net = policy.transformer.eval()
    obs = torch.randn(9, A, Do)
    avail = (torch.rand(9, A, na) < 0.6).float()
    avail[..., 0] = 1
    with torch.no_grad():
        _, _, rep = net._encode(obs)
        whole = lambda *a: net.decoder(*a)      # noqa: E731  -- hides ``begin``: whole-sequence pass per agent
        for det in (True, False):
            torch.manual_seed(5)
            a1, l1 = transformer_act.discrete_autoregreesive_act(net.decoder, rep, obs, 9, A, na, net.tpdv, avail, det)
            torch.manual_seed(5)
            a2, l2 = transformer_act.discrete_autoregreesive_act(whole, rep, obs, 9, A, na, net.tpdv, avail, det)
            assert torch.equal(a1, a2)
            torch.testing.assert_close(l1, l2, rtol=1e-5, atol=1e-6)
        # teacher forcing on the sampled actions reproduces their log-probs
        logp, _ = transformer_act.discrete_parallel_act(net.decoder, rep, obs, a1, 9, A, na, net.tpdv, avail)
        torch.testing.assert_close(logp, l1, rtol=1e-5, atol=1e-6)


def test_checkpoint_round_trip(tmp_path):
    args = make_args(algorithm_name="mat_dec", dec_actor=True, share_actor=True, n_embd=8)
    torch.manual_seed(0)
    a = TransformerPolicy(args, Box((Do,)), Box((Ds,)), Discrete(4), 3)
    a.save(str(tmp_path), 7)
    torch.manual_seed(1)
    b = TransformerPolicy(args, Box((Do,)), Box((Ds,)), Discrete(4), 3)
    b.restore(str(tmp_path / "transformer_7.pt"))
    for (k1, v1), (k2, v2) in zip(a.transformer.state_dict().items(), b.transformer.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.parametrize("algo", ["mat", "mat_dec"])
def test_smac_runner_full_loop_with_the_transformer(monkeypatch, tmp_path, algo):
    """SMACRunner.run() with --algorithm_name mat / mat_dec on the fake SMAC env (host buffer stand-in): rollouts,
    updates, logging, eval, transformer_<episode>.pt checkpoints and restoring one through --model_dir."""
    import json
    import os
    import fake_envs
    import onpolicy.runner.shared.base_runner as base
    from host_buffer import HostSharedBuffer
    from onpolicy.runner.shared.smac_runner import SMACRunner
    from onpolicy.scripts.train import _launch
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    T, N, A, Do, Ds, na = 6, 3, 4, 7, 9, 6
    args = make_args(env_name="StarCraft2", algorithm_name=algo, episode_length=T, n_rollout_threads=N,
                     num_env_steps=3 * T * N, n_embd=16, n_head=2, ppo_epoch=2, num_mini_batch=1, use_wandb=False,
                     use_eval=True, n_eval_rollout_threads=2, eval_episodes=2, eval_interval=1, log_interval=1,
                     save_interval=1)
    _launch.apply_algorithm_flags(args, ("mat", "mat_dec"))
    assert args.dec_actor == args.share_actor == (algo == "mat_dec")
    args.map_name = "fake"

    def config(run_dir):
        return {"all_args": args, "envs": fake_envs.FakeSMACVecEnv(N, A, Do, Ds, na),
                "eval_envs": fake_envs.FakeSMACVecEnv(2, A, Do, Ds, na, seed=3), "num_agents": A,
                "device": torch.device("cpu"), "run_dir": run_dir}
    torch.manual_seed(1)
    runner = SMACRunner(config(tmp_path / "a"))
    assert type(runner.trainer).__name__ == "MATTrainer" and type(runner.policy).__name__ == "TransformerPolicy"
    runner.run()
    tags = {json.loads(l)["tag"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))}
    assert {"value_loss", "policy_loss", "ratio", "eval_win_rate"} <= tags
    ckpt = os.path.join(runner.save_dir, "transformer_0.pt")
    assert os.path.exists(ckpt)
    args.model_dir = ckpt
    torch.manual_seed(2)
    again = SMACRunner(config(tmp_path / "b"))
    for (k1, v1), (k2, v2) in zip(runner.policy.transformer.state_dict().items(),
                                  again.policy.transformer.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_separated_runner_refuses_the_transformer(tmp_path):
    import fake_envs
    from onpolicy.runner.separated.mpe_runner import MPERunner
    args = make_args(env_name="MPE", algorithm_name="mat", share_policy=False, use_wandb=False)
    args.scenario_name = "fake"
    with pytest.raises(NotImplementedError):
        MPERunner({"all_args": args, "envs": fake_envs.FakeMPEVecEnv(2, 3, 6, 5), "eval_envs": None, "num_agents": 3,
                   "device": torch.device("cpu"), "run_dir": tmp_path})
