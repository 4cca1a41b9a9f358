"""Host logic of the shared runners (collect / insert / compute / train / the turn-based Hanabi loop) against
buffers and parameters produced by the REFERENCE's runners on the same deterministic fake envs and seeds
(oracle/make_golden_runners.py).  The HBM buffer is replaced by tests/host_buffer.py (oracle-backed) so that this
runs without a GPU; the -m gpu runner tests cover the device buffer."""
import numpy as np
import pytest
import torch

import fake_envs
from helpers import make_args
from host_buffer import HostSharedBuffer

import runner_replay
from runner_replay import check_buffer as _check_buffer, check_eval as _check_eval, check_params as _check_params


def _config(args, envs, A, tmp_path, eval_envs=None):
    return runner_replay.config(args, envs, A, tmp_path, eval_envs)


@pytest.fixture
def host_buffer(monkeypatch):
    import onpolicy.runner.shared.base_runner as base
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)


@pytest.mark.parametrize("cname", ["mpe_mlp", "mpe_rnn", "smac_rnn", "smac_mat", "smac_mat_dec", "mpe_mlp_h64", "smac_rnn_h64"])
def test_rollout_and_update_match_reference_runner(gold, host_buffer, tmp_path, cname):
    runner_replay.replay_shared_case(gold, tmp_path, cname)


def test_hanabi_turn_loop_matches_reference_runner(gold, host_buffer, tmp_path):
    runner_replay.replay_hanabi_case(gold, tmp_path)


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
