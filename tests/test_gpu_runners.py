"""-m gpu: the runner classes (the callers on either side of the hot path) driving the device
buffer with deterministic fake envs: what the envs emitted must be what the buffer holds, in the
reference's row conventions (observation-like fields at step+1, action-like at step), and
training must run through compute / train / after_update / save."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import make_args
from fake_envs import FakeMPEVecEnv, FakeSMACVecEnv

pytestmark = pytest.mark.gpu


def _config(args, envs, A, tmp_path):
    return {"all_args": args, "envs": envs, "eval_envs": None, "num_agents": A,
            "device": torch.device("cuda", 0), "run_dir": tmp_path}


@pytest.mark.parametrize("recurrent", [False, True])
def test_mpe_runner_rollout_and_update(tmp_path, recurrent):
    from onpolicy.runner.shared.mpe_runner import MPERunner
    T, N, A, Do, na = 8, 4, 3, 6, 5
    args = make_args(env_name="MPE", episode_length=T, n_rollout_threads=N, num_env_steps=2 * T * N,
                     hidden_size=16, ppo_epoch=2, num_mini_batch=2, use_recurrent_policy=recurrent,
                     algorithm_name="rmappo" if recurrent else "mappo", data_chunk_length=4, log_interval=1,
                     use_wandb=False)
    args.scenario_name = "fake_spread"
    envs = FakeMPEVecEnv(N, A, Do, na)
    torch.manual_seed(1)
    runner = MPERunner(_config(args, envs, A, tmp_path))
    runner.warmup()
    b = runner.buffer
    np.testing.assert_array_equal(b.obs[0].cpu().numpy(), envs.log[0]["obs"])
    np.testing.assert_array_equal(b.share_obs[0, :, 1].cpu().numpy(), envs.log[0]["obs"].reshape(N, -1))
    for step in range(T):
        out = runner.collect(step)
        values, actions = out[0], out[1]
        assert values.is_cuda and actions.is_cuda
        obs, rewards, dones, infos = envs.step(out[5])
        runner.insert((obs, rewards, dones, infos) + tuple(out[:5]))
        rec = envs.log[-1]
        np.testing.assert_array_equal(b.obs[step + 1].cpu().numpy(), rec["obs"])
        np.testing.assert_array_equal(b.rewards[step].cpu().numpy(), rec["rewards"])
        np.testing.assert_array_equal(b.actions[step, :, :, 0].cpu().numpy(), rec["actions"])
        np.testing.assert_array_equal(b.masks[step + 1, :, :, 0].cpu().numpy(), 1.0 - rec["dones"])
        np.testing.assert_array_equal(b.value_preds[step].cpu().numpy(), values.cpu().numpy())
        if recurrent:   # finished agents restart from a zero state
            assert float(b.rnn_states[step + 1][torch.as_tensor(rec["dones"])].abs().sum()) == 0.0
    assert b.step == 0
    runner.compute()
    assert float(b.returns[:-1].abs().sum()) > 0
    info = runner.train()
    assert all(np.isfinite(v) for v in info.values())
    np.testing.assert_array_equal(b.obs[0].cpu().numpy(), envs.log[-1]["obs"])     # after_update
    runner.save()
    assert os.path.exists(os.path.join(runner.save_dir, "actor.pt"))
    # the whole loop, as the train script drives it
    runner.run()
    lines = open(os.path.join(runner.log_dir, "scalars.jsonl")).read().strip().splitlines()
    assert any(json.loads(l)["tag"] == "value_loss" for l in lines)
    runner.writter.export_scalars_to_json(os.path.join(runner.log_dir, "summary.json"))


def test_smac_runner_masks(tmp_path):
    from onpolicy.runner.shared.smac_runner import SMACRunner
    T, N, A, Do, Ds, na = 8, 3, 4, 7, 9, 6
    args = make_args(env_name="StarCraft2", episode_length=T, n_rollout_threads=N, num_env_steps=2 * T * N,
                     hidden_size=16, ppo_epoch=1, num_mini_batch=1, use_recurrent_policy=True,
                     algorithm_name="rmappo", data_chunk_length=4, log_interval=1, use_wandb=False,
                     use_proper_time_limits=True)
    args.map_name = "fake"
    envs = FakeSMACVecEnv(N, A, Do, Ds, na)
    torch.manual_seed(1)
    runner = SMACRunner(_config(args, envs, A, tmp_path))
    runner.warmup()
    b = runner.buffer
    np.testing.assert_array_equal(b.available_actions[0].cpu().numpy(), envs.log[0]["available_actions"])
    for step in range(T):
        out = runner.collect(step)
        obs, share_obs, rewards, dones, infos, avail = envs.step(out[1].cpu().numpy())
        runner.insert((obs, share_obs, rewards, dones, infos, avail) + tuple(out))
        rec = envs.log[-1]
        dones_env = rec["dones"].all(1)
        exp_masks = np.ones((N, A)); exp_masks[dones_env] = 0
        exp_active = np.ones((N, A)); exp_active[rec["dones"]] = 0; exp_active[dones_env] = 1
        exp_bad = np.array([[0.0 if i[a]["bad_transition"] else 1.0 for a in range(A)] for i in rec["infos"]])
        np.testing.assert_array_equal(b.masks[step + 1, :, :, 0].cpu().numpy(), exp_masks)
        np.testing.assert_array_equal(b.active_masks[step + 1, :, :, 0].cpu().numpy(), exp_active)
        np.testing.assert_array_equal(b.bad_masks[step + 1, :, :, 0].cpu().numpy(), exp_bad)
        np.testing.assert_array_equal(b.share_obs[step + 1].cpu().numpy(), rec["share_obs"])
        np.testing.assert_array_equal(b.available_actions[step + 1].cpu().numpy(), rec["available_actions"])
    runner.compute()
    info = runner.train()
    assert all(np.isfinite(v) for v in info.values())
    runner.run()


def test_hanabi_runner_turn_based(tmp_path):
    """Turn-based runner on a fake choose-env: the loop (collect per player, chooseinsert, reward
    shift, compute, train, chooseafter_update, selective resets) runs, respects availability and
    counts only real moves."""
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner
    from fake_envs import FakeChooseVecEnv
    T, N, A, Do, Ds, na = 6, 5, 3, 9, 12, 7
    args = make_args(env_name="Hanabi", episode_length=T, n_rollout_threads=N, num_env_steps=4 * T * N,
                     hidden_size=16, ppo_epoch=2, num_mini_batch=1, algorithm_name="mappo", log_interval=1,
                     use_wandb=False)
    args.hanabi_name = "fake"
    envs = FakeChooseVecEnv(N, A, Do, Ds, na)
    torch.manual_seed(1)
    runner = HanabiRunner(_config(args, envs, A, tmp_path))
    runner.run()
    assert runner.true_total_num_steps == envs.steps > 0
    assert envs.games > 0 and len(runner.scores) >= 0
    b = runner.buffer
    assert torch.isfinite(b.returns).all() and torch.isfinite(b.rewards).all()
    # masks / active masks only ever hold 0 or 1
    for name in ("masks", "active_masks"):
        v = getattr(b, name)
        assert bool(((v == 0) | (v == 1)).all())
    lines = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    assert any(r["tag"] == "value_loss" for r in lines) and any(r["tag"] == "average_score" for r in lines)


def test_naive_recurrent_trainer_on_device():
    """--use_naive_recurrent_policy: whole-trajectory minibatches (chunk gather with L = T) through
    the trainer on the device buffer."""
    from helpers import Box, Discrete, fill_buffer_arrays, buffer_shapes, load_into
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A, Do, Ds, na = 8, 6, 2, 5, 9, 4
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=16, ppo_epoch=2, num_mini_batch=3,
                     use_naive_recurrent_policy=True, use_recurrent_policy=False, algorithm_name="rmappo")
    dev = torch.device("cuda", 0)
    spaces = Box((Do,)), Box((Ds,)), Discrete(na)
    torch.manual_seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, A, *spaces, device=dev)
    assert buf.rnn_states.stride()[0] != 0          # recurrent => real state storage
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, na, 16), np.random.default_rng(0), na=na)
    av = arrays["available_actions"][:-1]
    arrays["actions"] = (np.random.default_rng(1).random(av.shape) * av).argmax(-1)[..., None].astype(np.float32)
    load_into(buf, arrays)
    buf.compute_returns(arrays["next_value"], trainer.value_normalizer)
    trainer.prep_training()
    info = trainer.train(buf)
    assert all(np.isfinite(v) for v in info.values()), info


@pytest.mark.parametrize("recurrent", [False, True])
def test_football_runner(tmp_path, recurrent):
    from onpolicy.runner.shared.football_runner import FootballRunner
    from fake_envs import FakeFootballVecEnv
    T, N, A, Do, na = 6, 4, 3, 9, 7
    args = make_args(env_name="Football", episode_length=T, n_rollout_threads=N, num_env_steps=3 * T * N,
                     hidden_size=16, ppo_epoch=2, num_mini_batch=2, use_recurrent_policy=recurrent,
                     algorithm_name="rmappo" if recurrent else "mappo", data_chunk_length=3, log_interval=T * N,
                     save_interval=T * N, eval_interval=T * N, use_eval=True, n_eval_rollout_threads=2,
                     eval_episodes=3, use_wandb=False)
    envs = FakeFootballVecEnv(N, A, Do, na)
    torch.manual_seed(1)
    runner = FootballRunner({"all_args": args, "envs": envs, "eval_envs": FakeFootballVecEnv(2, A, Do, na, seed=5),
                             "num_agents": A, "device": torch.device("cuda", 0), "run_dir": tmp_path})
    runner.warmup()
    b = runner.buffer
    np.testing.assert_array_equal(b.obs[0].cpu().numpy(), envs.log[0]["obs"])
    np.testing.assert_array_equal(b.share_obs[0].cpu().numpy(), envs.log[0]["obs"])
    for step in range(T):
        out = runner.collect(step)
        obs, rewards, dones, infos = envs.step(out[5])
        runner.insert((obs, rewards, dones, infos) + tuple(out[:5]))
        rec = envs.log[-1]
        np.testing.assert_array_equal(b.obs[step + 1].cpu().numpy(), rec["obs"])
        np.testing.assert_array_equal(b.share_obs[step + 1].cpu().numpy(), rec["obs"])
        np.testing.assert_array_equal(b.actions[step, :, :, 0].cpu().numpy(), rec["actions"])
        np.testing.assert_array_equal(b.masks[step + 1, :, :, 0].cpu().numpy(), 1.0 - rec["dones"])
        if recurrent:
            assert float(b.rnn_states[step + 1][torch.as_tensor(rec["dones"])].abs().sum()) == 0.0
    assert len(runner.env_infos["goal"]) >= 2 and set(runner.env_infos) == {"goal", "win_rate", "steps"}
    runner.compute()
    info = runner.train()
    assert all(np.isfinite(v) for v in info.values())
    runner.run()
    lines = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    tags = {r["tag"] for r in lines}
    assert {"value_loss", "goal", "win_rate", "eval_goal", "eval_win_rate", "eval_step"} <= tags


@pytest.mark.parametrize("kind", ["Box", "MultiDiscrete", "MultiBinary"])
def test_other_action_spaces_on_the_device_buffer(kind):
    """Non-Discrete action heads (continuous, multi-discrete, multi-binary; reference utils/act.py:10-42,
    utils/util.py:40-52 for the stored action width): rollout actions go into the HBM buffer with the right
    width, there is no availability mask, and the update takes the framework loss path (the fused loss is for
    Discrete heads) through compute_returns / the samplers / train."""
    from helpers import Box
    from test_misc_cpu import _MultiDiscrete, _MultiBinary
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    space, width = {"Box": (Box((3,)), 3), "MultiDiscrete": (_MultiDiscrete([3, 4]), 2),
                    "MultiBinary": (_MultiBinary(4), 4)}[kind]
    T, N, A, Do, Ds = 6, 5, 2, 7, 14
    dev = torch.device("cuda", 0)
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=16, ppo_epoch=2, num_mini_batch=2)
    torch.manual_seed(3)
    policy = R_MAPPOPolicy(args, Box((Do,)), Box((Ds,)), space, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    assert not trainer._fused_loss
    buf = SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), space, device=dev)
    assert buf.available_actions is None and tuple(buf.actions.shape) == (T, N, A, width)
    rng = np.random.default_rng(0)
    buf.obs[0] = torch.as_tensor(rng.standard_normal((N, A, Do)), dtype=torch.float32)
    buf.share_obs[0] = torch.as_tensor(rng.standard_normal((N, A, Ds)), dtype=torch.float32)
    trainer.prep_rollout()
    flat = lambda x: x.reshape(N * A, *x.shape[2:])
    for step in range(T):
        with torch.no_grad():
            v, a, lp, ha, hc = policy.get_actions(flat(buf.share_obs[step]), flat(buf.obs[step]),
                                                  flat(buf.rnn_states[step]), flat(buf.rnn_states_critic[step]),
                                                  flat(buf.masks[step]))
        per_env = lambda x: x.reshape(N, A, *x.shape[1:])
        buf.insert(rng.standard_normal((N, A, Ds)).astype(np.float32), rng.standard_normal((N, A, Do)).astype(np.float32),
                   per_env(ha), per_env(hc), per_env(a), per_env(lp), per_env(v),
                   rng.standard_normal((N, A, 1)).astype(np.float32), np.ones((N, A, 1), np.float32))
    with torch.no_grad():
        nv = policy.get_values(flat(buf.share_obs[-1]), flat(buf.rnn_states_critic[-1]), flat(buf.masks[-1]))
    buf.compute_returns(nv.reshape(N, A, 1), trainer.value_normalizer)
    trainer.prep_training()
    info = trainer.train(buf)
    assert all(np.isfinite(v) for v in info.values()), info
    sample = next(iter(buf.feed_forward_generator(None, 2)))
    assert sample[11] is None and tuple(sample[4].shape) == (T * N * A // 2, width)


# ------------------------------------------------------------------ a12: the reference's runners, replayed on the HBM buffer
@pytest.mark.parametrize("cname", ["mpe_mlp", "mpe_rnn", "smac_rnn"])
def test_rollout_and_update_vs_reference_runner(gold, tmp_path, cname):
    """tests/test_runners_cpu.py's comparison with the REFERENCE's runners (runner_cases.npz: rollout buffer, update,
    buffer after the update, parameters, eval log) with the real thing in place of the host stand-in: policy on the
    GPU, rollout buffer in HBM, returns / samplers / loss through the HIP kernels.  ``--sampler_rng host`` draws the
    action noise and the minibatch permutations on the CPU generator like the reference, so the sampled actions must
    be identical; floats to the tolerances of the CPU test (2e-4)."""
    import runner_replay
    runner = runner_replay.replay_shared_case(gold, tmp_path, cname, device=torch.device("cuda", 0), init_exact=False,
                                              sampler_rng="host")
    assert runner.buffer.obs.is_cuda and type(runner.buffer).__name__ == "SharedReplayBuffer"
    assert next(runner.policy.actor.parameters()).is_cuda


@pytest.mark.parametrize("cname", ["mpe_mlp_h64", "smac_rnn_h64"])
def test_hidden64_rollout_and_update_vs_reference_runner(gold, tmp_path, cname):
    """The same replay at hidden size 64, where the device update runs through the fused trunk (K9) and, for the recurrent
    SMAC case, the GRU chunk kernels (K12); the test asserts that those kernels were launched."""
    import runner_replay
    from onpolicy.algorithms.utils import fused_mlp
    fused_mlp.profile(True)
    try:
        runner = runner_replay.replay_shared_case(gold, tmp_path, cname, device=torch.device("cuda", 0), init_exact=False,
                                                  sampler_rng="host")
        torch.cuda.synchronize()
        launches = fused_mlp.profile_times()
    finally:
        fused_mlp.profile(False)
    assert launches.get("mappo_mlp_forward", (0,))[0] > 0 and launches.get("mappo_mlp_backward", (0,))[0] > 0, launches
    assert runner.buffer.obs.is_cuda
    if cname == "smac_rnn_h64":
        assert runner.policy.actor.rnn._chunk_kernel_ok(torch.zeros(4, 64, device="cuda"))


def test_hanabi_turn_loop_vs_reference_runner(gold, tmp_path):
    """The reference's whole turn-based Hanabi loop (chooseinsert / reward shift / chooseafter_update, four episodes with
    lr decay) replayed on the HBM buffer."""
    import runner_replay
    runner = runner_replay.replay_hanabi_case(gold, tmp_path, device=torch.device("cuda", 0), init_exact=False,
                                              sampler_rng="host")
    assert runner.buffer.obs.is_cuda
