"""The built-in simple_spread (vectorised numpy, onpolicy/envs/mpe/simple_spread.py) against trajectories of the
reference's own particle environment (core.py physics + environment.py + scenarios/simple_spread.py, fixtures from
oracle/make_golden_mpe.py): same initial state + same actions => same observations, rewards, dones and positions."""
import numpy as np
import pytest

from onpolicy.envs.mpe.simple_spread import VecSimpleSpread


@pytest.mark.parametrize("case", [0, 1, 2])
def test_simple_spread_matches_reference_trajectories(gold, case):
    z = gold.npz("mpe_spread_cases")
    key = "mpe%d_" % case
    A, L, T = [int(x) for x in z[key + "dims"]]
    env = VecSimpleSpread(2, num_agents=A, num_landmarks=L, episode_length=T, seed=0, auto_reset=False)
    assert env.observation_space[0].shape == tuple(z[key + "obs_dim"])
    assert env.share_observation_space[0].shape == tuple(z[key + "share_obs_dim"])
    env.reset()
    # world 0 replays the reference trajectory, world 1 stays random (worlds must not interact)
    env.pos[0], env.vel[0], env.landmarks[0] = z[key + "pos0"], z[key + "vel0"], z[key + "landmarks"]
    np.testing.assert_allclose(env._obs()[0], z[key + "obs0"], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(99)
    for t in range(T):
        act = np.stack([z[key + "actions"][t], np.eye(5)[rng.integers(0, 5, A)]])
        obs, rew, done, info = env.step(act)
        np.testing.assert_allclose(env.pos[0], z[key + "pos"][t], rtol=1e-9, atol=1e-9, err_msg="t=%d" % t)
        np.testing.assert_allclose(obs[0], z[key + "obs"][t], rtol=2e-6, atol=2e-6, err_msg="t=%d" % t)
        np.testing.assert_allclose(rew[0], z[key + "rewards"][t].reshape(A, 1), rtol=1e-6, atol=1e-6, err_msg="t=%d" % t)
        np.testing.assert_array_equal(done[0], z[key + "dones"][t])


@pytest.mark.parametrize("case", [0, 1, 2])
def test_torch_simple_spread_matches_reference_trajectories(gold, case):
    """TorchSimpleSpread (worlds as tensors, action indices in) on the same reference trajectories."""
    import torch
    from onpolicy.envs.mpe.simple_spread import TorchSimpleSpread
    z = gold.npz("mpe_spread_cases")
    key = "mpe%d_" % case
    A, L, T = [int(x) for x in z[key + "dims"]]
    env = TorchSimpleSpread(2, num_agents=A, num_landmarks=L, episode_length=T, seed=0, auto_reset=False)
    assert env.device_resident and env.observation_space[0].shape == tuple(z[key + "obs_dim"])
    env.reset()
    for name, src in (("pos", "pos0"), ("vel", "vel0"), ("landmarks", "landmarks")):
        getattr(env, name)[0] = torch.from_numpy(np.asarray(z[key + src], dtype=np.float64))
    np.testing.assert_allclose(env._obs()[0].numpy(), z[key + "obs0"], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(99)
    for t in range(T):
        idx = np.stack([np.argmax(z[key + "actions"][t], -1), rng.integers(0, 5, A)])[..., None]
        obs, rew, done, info = env.step(torch.from_numpy(idx))
        assert obs.dtype == torch.float32 and rew.shape == (2, A, 1) and done.dtype == torch.bool
        np.testing.assert_allclose(env.pos[0].numpy(), z[key + "pos"][t], rtol=1e-9, atol=1e-9, err_msg="t=%d" % t)
        np.testing.assert_allclose(obs[0].numpy(), z[key + "obs"][t], rtol=2e-6, atol=2e-6, err_msg="t=%d" % t)
        np.testing.assert_allclose(rew[0].numpy(), z[key + "rewards"][t].reshape(A, 1), rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(done[0].numpy(), z[key + "dones"][t])
        assert len(info) == 2 and set(info[0][0]) == {"individual_reward"}


def test_torch_and_numpy_worlds_agree_including_auto_reset():
    """Same physics in both implementations (one-hot and index actions); finished worlds restart in place."""
    import torch
    from onpolicy.envs.mpe.simple_spread import TorchSimpleSpread
    n, A, L, T = 6, 4, 3, 5
    a = VecSimpleSpread(n, A, L, T, seed=0)
    b = TorchSimpleSpread(n, A, L, T, seed=0)
    a.reset()
    b.reset()
    rng = np.random.default_rng(0)
    for episode in range(2):
        b.pos, b.vel, b.landmarks = (torch.from_numpy(x.copy()) for x in (a.pos, a.vel, a.landmarks))
        for t in range(T):
            idx = rng.integers(0, 5, (n, A, 1))
            o1, r1, d1, i1 = a.step(np.eye(5)[idx[..., 0]])
            o2, r2, d2, i2 = b.step(np.eye(5)[idx[..., 0]] if t % 2 else torch.from_numpy(idx))
            np.testing.assert_allclose(r2.numpy(), r1, rtol=1e-6, atol=1e-5)
            np.testing.assert_array_equal(d2.numpy(), d1)
            assert i2[1][2]["individual_reward"] == pytest.approx(i1[1][2]["individual_reward"])
            if t + 1 < T:
                np.testing.assert_allclose(o2.numpy(), o1, rtol=1e-6, atol=1e-6)
        assert bool(d2.all()) and int(b.t.sum()) == 0 and float(b.vel.abs().sum()) == 0.0     # restarted
        assert float(b.pos.abs().max()) <= 1.0 and float(b.landmarks.abs().max()) <= 1.0


@pytest.mark.parametrize("algo", ["mappo", "rmappo"])
def test_train_script_with_device_resident_worlds(monkeypatch, tmp_path, algo):
    """train_mpe --use_device_env end to end (host buffer stand-in, "device" = CPU tensors): the runner hands the
    policy's action indices to the env as a tensor and inserts the tensors that come back; learning signals are logged
    as with the host env."""
    import json
    import os
    import torch
    import onpolicy.runner.shared.base_runner as base
    from host_buffer import HostSharedBuffer
    from onpolicy.scripts.train import _launch, train_mpe
    threads = torch.get_num_threads()

    def device_of(all_args):
        torch.set_num_threads(all_args.n_training_threads)
        return torch.device("cpu")
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    monkeypatch.setattr(_launch, "device_of", device_of)
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    argv = ["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3", "--num_landmarks", "3",
            "--algorithm_name", algo, "--n_rollout_threads", "4", "--episode_length", "10", "--num_env_steps", "120",
            "--ppo_epoch", "2", "--num_mini_batch", "1", "--data_chunk_length", "5", "--hidden_size", "16",
            "--use_wandb", "--log_interval", "1", "--n_training_threads", "1", "--use_eval", "--eval_interval", "2",
            "--n_eval_rollout_threads", "2"]
    try:
        runner = train_mpe.main(argv + ["--use_device_env"])
        assert type(runner.envs).__name__ == "TorchSimpleSpread" and type(runner.eval_envs).__name__ == "VecSimpleSpread"
        tags = {json.loads(l)["tag"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))}
        assert {"value_loss", "average_episode_rewards", "agent0/individual_rewards",
                "eval_average_episode_rewards"} <= tags
        assert torch.isfinite(runner.buffer.rewards).all() and float(runner.buffer.masks.min()) == 0.0
        with pytest.raises(NotImplementedError, match="use_device_env"):
            train_mpe.main(argv + ["--use_device_env", "--share_policy"])        # store_false: separated runner
    finally:
        torch.set_num_threads(threads)


def test_rasterised_frames_show_the_entities():
    from onpolicy.envs.mpe.MPE_env import SimpleSpreadEnv
    env = SimpleSpreadEnv(num_agents=2, num_landmarks=1, episode_length=5, seed=0)
    env.reset()
    env._world.pos[0] = [[0.0, 0.0], [1.0, -1.0]]
    env._world.landmarks[0] = [[-1.0, 1.0]]
    (frame,) = env.render("rgb_array", size=300)
    assert frame.shape == (300, 300, 3) and frame.dtype == np.uint8
    px = lambda x, y: tuple(int(v) for v in frame[int((1.5 - y) / 3 * 300), int((x + 1.5) / 3 * 300)])   # noqa: E731
    assert px(0.0, 0.0) == (89, 89, 217) and px(1.0, -1.0) == (89, 89, 217)      # agents
    assert px(-1.0, 1.0) == (64, 64, 64)                                         # landmark
    assert px(-1.0, -1.0) == (255, 255, 255) and px(0.0, 0.3) == (255, 255, 255)  # background, outside the agent disc
    with pytest.raises(NotImplementedError, match="rgb_array"):
        env.render("human")


@pytest.mark.parametrize("shared", [True, False])
def test_render_script_records_a_saved_policy(monkeypatch, tmp_path, shared):
    """scripts/render/render_mpe.py: checkpoints of a short training run, replayed deterministically with frames
    recorded (render.npz here: imageio is not installed)."""
    import os
    import torch
    import onpolicy.runner.shared.base_runner as base
    import onpolicy.runner.separated.base_runner as sep_base
    from host_buffer import HostSharedBuffer
    from onpolicy.scripts.render import render_mpe
    from onpolicy.scripts.train import _launch, train_mpe
    threads = torch.get_num_threads()

    def device_of(all_args):
        torch.set_num_threads(all_args.n_training_threads)
        return torch.device("cpu")
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    monkeypatch.setattr(_launch, "device_of", device_of)
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    common = ["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "2", "--num_landmarks", "2",
              "--algorithm_name", "mappo", "--episode_length", "6", "--hidden_size", "16", "--use_wandb",
              "--n_training_threads", "1"] + ([] if shared else ["--share_policy"])
    try:
        if shared:
            trained = train_mpe.main(common + ["--n_rollout_threads", "2", "--num_env_steps", "24", "--ppo_epoch", "1"])
        else:       # the separated runner needs per-agent HBM buffers to train; checkpoints of fresh policies do here
            from oracle import oracle
            monkeypatch.setattr(sep_base, "SeparatedReplayBuffer",
                                lambda a, o, s, act, device=None: oracle.OracleSeparatedBuffer(a, o, s, act))
            from onpolicy.runner.separated.mpe_runner import MPERunner
            from onpolicy.envs.mpe.simple_spread import VecSimpleSpread
            from onpolicy.config import get_config
            args = train_mpe.parse_args(common + ["--n_rollout_threads", "2"], get_config())
            _launch.apply_algorithm_flags(args, ("mappo",))
            trained = MPERunner({"all_args": args, "envs": VecSimpleSpread(2, 2, 2, 6), "eval_envs": None,
                                 "num_agents": 2, "device": torch.device("cpu"), "run_dir": tmp_path / "sep"})
            trained.save()
        argv = common + ["--n_rollout_threads", "1", "--use_render", "--save_gifs", "--render_episodes", "2",
                         "--ifi", "0.0", "--model_dir", str(trained.save_dir)]
        with pytest.raises(AssertionError, match="1 env"):
            render_mpe.main(common + ["--n_rollout_threads", "2", "--use_render", "--model_dir", str(trained.save_dir)])
        with pytest.raises(AssertionError, match="model_dir"):
            render_mpe.main(common + ["--n_rollout_threads", "1", "--use_render"])
        runner = render_mpe.main(argv)
        frames = np.load(os.path.join(runner.gif_dir, "render.npz"))["frames"]
        assert frames.shape == (2 * (1 + 6), 350, 350, 3) and frames.dtype == np.uint8
        assert (frames[0] != frames[3]).any()                       # the agents move
    finally:
        torch.set_num_threads(threads)
