"""Host logic of the shared runners (collect / insert / compute / train / the turn-based Hanabi loop) against
buffers and parameters produced by the REFERENCE's runners on the same deterministic fake envs and seeds
(oracle/make_golden_runners.py).  The HBM buffer is replaced by tests/host_buffer.py (oracle-backed) so that this
runs without a GPU; the -m gpu runner tests cover the device buffer."""
import os

import numpy as np
import pytest
import torch

import fake_envs
from helpers import make_args
from host_buffer import HostSharedBuffer

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
          "rewards", "masks", "bad_masks", "active_masks", "returns", "available_actions")
TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture
def host_buffer(monkeypatch):
    import onpolicy.runner.shared.base_runner as base
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)


def _config(args, envs, A, tmp_path, eval_envs=None):
    return {"all_args": args, "envs": envs, "eval_envs": eval_envs, "num_agents": A, "device": torch.device("cpu"),
            "run_dir": tmp_path}


def _check_eval(runner, step, expected):
    """runner.eval(step) must log what the reference runner logged (tag, value) for the same policy and eval envs."""
    import json
    path = os.path.join(runner.log_dir, "scalars.jsonl")
    before = len(open(path).read().splitlines()) if os.path.exists(path) else 0
    runner.eval(step)
    got = [json.loads(l) for l in open(path).read().splitlines()[before:]]
    assert [g["tag"] for g in got] == [e[0] for e in expected]
    for g, (tag, vals, at) in zip(got, expected):
        assert g["step"] == at
        for k, v in vals.items():
            assert g[k] == pytest.approx(v, rel=2e-4, abs=2e-6), (tag, g[k], v)


def _check_params(z, prefix, policy, exact=False):
    nets = (("transformer.", policy.transformer),) if hasattr(policy, "transformer") else \
        (("actor.", policy.actor), ("critic.", policy.critic))
    for net, mod in nets:
        for k, v in mod.state_dict().items():
            ref = z[prefix + net + k]
            if exact:
                np.testing.assert_array_equal(v.numpy(), ref, err_msg=prefix + net + k)
            else:
                np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=3e-5, err_msg=prefix + net + k)


def _check_buffer(z, prefix, buf):
    for name in FIELDS:
        got = getattr(buf, name).numpy()
        ref = z[prefix + name]
        if name == "actions":
            np.testing.assert_array_equal(got, ref, err_msg=name)          # integer sampling parity
        else:
            np.testing.assert_allclose(got, ref, err_msg=name, **TOL)


@pytest.mark.parametrize("cname", ["mpe_mlp", "mpe_rnn", "smac_rnn", "smac_mat", "smac_mat_dec"])
def test_rollout_and_update_match_reference_runner(gold, host_buffer, tmp_path, cname):
    from onpolicy.runner.shared.mpe_runner import MPERunner
    from onpolicy.runner.shared.smac_runner import SMACRunner
    z, meta = gold.npz("runner_cases"), gold.meta("runner_cases")[cname]
    sp = meta["spec"]
    T, N, A = sp["T"], sp["N"], sp["A"]
    smac = sp["env"] == "StarCraft2"
    args = make_args(env_name=sp["env"], episode_length=T, n_rollout_threads=N, num_env_steps=T * N, use_wandb=False,
                     use_eval=True, n_eval_rollout_threads=2, eval_episodes=4, **sp["args"])
    args.scenario_name = args.map_name = "fake"
    envs = fake_envs.FakeSMACVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"]) if smac \
        else fake_envs.FakeMPEVecEnv(N, A, sp["Do"], sp["na"])
    eval_envs = fake_envs.FakeSMACVecEnv(2, A, sp["Do"], sp["Ds"], sp["na"], seed=3) if smac \
        else fake_envs.FakeMPEVecEnv(2, A, sp["Do"], sp["na"], seed=3)
    torch.manual_seed(1)
    np.random.seed(1)
    runner = (SMACRunner if smac else MPERunner)(_config(args, envs, A, tmp_path, eval_envs))
    key = "run_%s_" % cname
    _check_params(z, key + "init_", runner.policy, exact=True)
    torch.manual_seed(5)
    runner.warmup()
    for step in range(T):
        res = runner.collect(step)
        if smac:
            values, actions, action_log_probs, rnn_states, rnn_states_critic = res
            obs, share_obs, rewards, dones, infos, avail = envs.step(actions.numpy())
            runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, action_log_probs, rnn_states,
                           rnn_states_critic))
        else:
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = res
            obs, rewards, dones, infos = envs.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))
    runner.compute()
    _check_buffer(z, key + "rollout_", runner.buffer)
    torch.manual_seed(9)
    info = runner.train()
    for k, v in meta["train_info"].items():
        assert info[k] == pytest.approx(v, rel=5e-4, abs=5e-6), (k, info[k], v)
    _check_buffer(z, key + "after_", runner.buffer)
    _check_params(z, key + "final_", runner.policy)
    _check_eval(runner, 777, meta["eval_logged"])


def test_hanabi_turn_loop_matches_reference_runner(gold, host_buffer, tmp_path):
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner
    z, meta = gold.npz("runner_cases"), gold.meta("runner_cases")["hanabi"]
    sp = meta["spec"]
    T, N, A = sp["T"], sp["N"], sp["A"]
    args = make_args(env_name="Hanabi", episode_length=T, n_rollout_threads=N, num_env_steps=4 * T * N, hidden_size=16,
                     ppo_epoch=2, num_mini_batch=1, algorithm_name="mappo", use_linear_lr_decay=True, log_interval=1000, save_interval=1000,
                     use_wandb=False)
    args.hanabi_name = "fake"
    args.n_eval_rollout_threads = 3
    envs = fake_envs.FakeChooseVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    runner = HanabiRunner(_config(args, envs, A, tmp_path, fake_envs.FakeChooseVecEnv(3, A, sp["Do"], sp["Ds"], sp["na"],
                                                                                     seed=4)))
    _check_params(z, "run_hanabi_init_", runner.policy, exact=True)
    torch.manual_seed(5)
    runner.run()
    assert runner.true_total_num_steps == meta["true_total_num_steps"]
    assert envs.steps == meta["env_steps"] and envs.games == meta["games"]
    _check_buffer(z, "run_hanabi_after_", runner.buffer)
    _check_params(z, "run_hanabi_final_", runner.policy)
    _check_eval(runner, 888, meta["eval_logged"])


@pytest.fixture
def host_separated_buffer(monkeypatch):
    from host_buffer import HostSeparatedBuffer
    import onpolicy.runner.separated.base_runner as base
    monkeypatch.setattr(base, "SeparatedReplayBuffer", HostSeparatedBuffer)


@pytest.mark.parametrize("cname", ["sep_mpe_mlp", "sep_smac_happo", "sep_smac_hatrpo"])
def test_separated_runners_match_reference(gold, host_separated_buffer, tmp_path, cname):
    """Per-agent policies / trainers / buffers, random update order, factor bookkeeping
    (reference runner/separated/base_runner.py:135-183) with MAPPO and with HAPPO trainers."""
    from onpolicy.runner.separated.mpe_runner import MPERunner
    from onpolicy.runner.separated.smac_runner import SMACRunner
    z, meta = gold.npz("runner_cases"), gold.meta("runner_cases")[cname]
    sp = meta["spec"]
    T, N, A = sp["T"], sp["N"], sp["A"]
    smac = sp["env"] == "StarCraft2"
    args = make_args(env_name=sp["env"], episode_length=T, n_rollout_threads=N, num_env_steps=T * N, use_wandb=False,
                     use_eval=True, n_eval_rollout_threads=2, eval_episodes=4, **sp["args"])
    args.scenario_name = args.map_name = "fake"
    envs = fake_envs.FakeSMACVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"]) if smac \
        else fake_envs.FakeMPEVecEnv(N, A, sp["Do"], sp["na"])
    eval_envs = fake_envs.FakeSMACVecEnv(2, A, sp["Do"], sp["Ds"], sp["na"], seed=3) if smac \
        else fake_envs.FakeMPEVecEnv(2, A, sp["Do"], sp["na"], seed=3)
    torch.manual_seed(1)
    np.random.seed(1)
    runner = (SMACRunner if smac else MPERunner)(_config(args, envs, A, tmp_path, eval_envs))
    key = "run_%s_" % cname
    for a in range(A):
        _check_params(z, key + "init%d_" % a, runner.policy[a], exact=True)
    torch.manual_seed(5)
    runner.warmup()
    for step in range(T):
        res = runner.collect(step)
        if smac:
            values, actions, action_log_probs, rnn_states, rnn_states_critic = res
            actions_env = np.stack([x.numpy() for x in actions], axis=1)
            obs, share_obs, rewards, dones, infos, avail = envs.step(actions_env)
            runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, action_log_probs, rnn_states,
                           rnn_states_critic))
        else:
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = res
            obs, rewards, dones, infos = envs.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))
    runner.compute()
    for a in range(A):
        _check_buffer(z, key + "rollout%d_" % a, runner.buffer[a])
    torch.manual_seed(9)
    infos_train = runner.train()
    for a in range(A):
        for k, v in meta["train_info"][a].items():
            assert infos_train[a][k] == pytest.approx(v, rel=5e-4, abs=5e-6), (a, k, infos_train[a][k], v)
        np.testing.assert_allclose(runner.buffer[a].factor.numpy(), z[key + "factor%d" % a], rtol=2e-4, atol=2e-6)
        _check_params(z, key + "final%d_" % a, runner.policy[a])
    if meta["eval_logged"] is not None:       # None: the reference's own eval failed on this case (see the generator)
        _check_eval(runner, 555, meta["eval_logged"])
    else:
        runner.eval(555)
