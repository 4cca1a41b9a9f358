"""-m gpu: BASELINE.json configs[0] end to end -- MPE simple_spread, 3 agents, shared policy --
through the train script, the MPE runner, the HBM buffer and the HIP kernels."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scalars(log_dir, tag):
    out = []
    for line in open(os.path.join(log_dir, "scalars.jsonl")):
        rec = json.loads(line)
        if rec["tag"] == tag:
            out.append(rec[tag])
    return out


def test_config0_shapes_run(tmp_path, monkeypatch):
    """3 agents, 8 rollout threads, episode length 25 (the reference's CPU-runnable case)."""
    from onpolicy.scripts.train import train_mpe
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path))
    runner = train_mpe.main(["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3",
                             "--num_landmarks", "3", "--n_rollout_threads", "8", "--episode_length", "25",
                             "--num_env_steps", "1000", "--ppo_epoch", "10", "--use_ReLU", "--gain", "0.01",
                             "--lr", "7e-4", "--critic_lr", "7e-4", "--use_wandb", "--log_interval", "1",
                             "--algorithm_name", "mappo"])
    assert runner.buffer.obs.shape == (26, 8, 3, 18) and runner.buffer.share_obs.shape == (26, 8, 3, 54)
    vl = _scalars(runner.log_dir, "value_loss")
    assert len(vl) == 5 and all(np.isfinite(vl))
    assert os.path.exists(os.path.join(runner.log_dir, "summary.json"))
    assert os.path.exists(os.path.join(runner.save_dir, "actor.pt"))


def test_simple_spread_learns(tmp_path, monkeypatch):
    """A short run must improve the average episode reward (recurrent policy, 128 threads)."""
    from onpolicy.scripts.train import train_mpe
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path))
    runner = train_mpe.main(["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3",
                             "--num_landmarks", "3", "--n_rollout_threads", "128", "--episode_length", "25",
                             "--num_env_steps", str(128 * 25 * 60), "--ppo_epoch", "10", "--use_ReLU",
                             "--gain", "0.01", "--lr", "7e-4", "--critic_lr", "7e-4", "--use_wandb",
                             "--log_interval", "1", "--algorithm_name", "rmappo", "--seed", "1"])
    r = _scalars(runner.log_dir, "average_episode_rewards")
    assert len(r) == 60 and all(np.isfinite(r))
    first, last = np.mean(r[:5]), np.mean(r[-5:])
    print("average episode rewards: first 5 = %.2f, last 5 = %.2f" % (first, last))
    assert last > first + 0.02 * abs(first), (first, last)


@pytest.mark.parametrize("algo", ["mappo", "rmappo"])
def test_train_mpe_with_device_resident_worlds(tmp_path, monkeypatch, algo):
    """Row f1: the training worlds are tensors on the policy's device (TorchSimpleSpread, pinned to the reference's
    trajectories on CPU tensors in tests/test_mpe_env_cpu.py); collect -> env step -> insert never leaves the GPU."""
    from onpolicy.scripts.train import train_mpe
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    runner = train_mpe.main(["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3",
                             "--num_landmarks", "3", "--algorithm_name", algo, "--n_rollout_threads", "16",
                             "--episode_length", "10", "--num_env_steps", "480", "--ppo_epoch", "2", "--num_mini_batch", "1",
                             "--data_chunk_length", "5", "--hidden_size", "16", "--use_wandb", "--log_interval", "1",
                             "--n_training_threads", "1", "--use_device_env"])
    assert type(runner.envs).__name__ == "TorchSimpleSpread" and runner.envs.pos.is_cuda
    tags = {json.loads(l)["tag"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))}
    assert {"value_loss", "average_episode_rewards", "agent0/individual_rewards"} <= tags
    assert torch.isfinite(runner.buffer.rewards).all() and float(runner.buffer.masks.min()) == 0.0


@pytest.mark.parametrize("algo", ["mappo", "rmappo"])
def test_device_worlds_with_the_fused_hidden64_kernels(tmp_path, monkeypatch, algo):
    """BASELINE.json configs[2] in small: simple_spread (3 agents) on device-resident worlds (K11) at the shipped hidden
    size 64, so that the update runs through the fused trunk (K9) and, for rmappo, the GRU chunk kernels (K12; they also
    serve the rollout's single steps) -- the combination tools/cfg3_end_to_end.py measures at 4096 threads x 400 steps.
    Three iterations must reduce nothing to NaN and must have launched the kernels; mappo's average reward must not
    collapse (a wrong action / observation hand-over between K11 and K9 shows up as garbage rewards)."""
    from onpolicy.algorithms.utils import fused_mlp
    from onpolicy.scripts.train import train_mpe
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    N, T = 256, 25
    fused_mlp.profile(True)
    try:
        runner = train_mpe.main(["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3",
                                 "--num_landmarks", "3", "--algorithm_name", algo, "--n_rollout_threads", str(N),
                                 "--episode_length", str(T), "--num_env_steps", str(3 * N * T), "--ppo_epoch", "4",
                                 "--num_mini_batch", "1", "--data_chunk_length", "5", "--hidden_size", "64", "--use_ReLU",
                                 "--gain", "0.01", "--lr", "7e-4", "--critic_lr", "7e-4", "--use_wandb", "--log_interval",
                                 "1", "--n_training_threads", "1", "--use_device_env"])
        torch.cuda.synchronize()
        launches = fused_mlp.profile_times()
    finally:
        fused_mlp.profile(False)
    assert type(runner.envs).__name__ == "TorchSimpleSpread" and runner.envs.pos.is_cuda
    # 3 iterations x 4 epochs x 2 networks in the update: launched eagerly (event pairs) or replayed from the update graph
    from helpers import graph_replays
    replays = graph_replays(runner.trainer)
    assert 0 < replays <= 3 * 4 - 1, replays
    assert launches.get("mappo_mlp_backward", (0,))[0] == (3 * 4 - replays) * 2, (launches, replays)
    assert launches.get("mappo_mlp_forward", (0,))[0] >= (3 * 4 - replays) * 2, (launches, replays)
    if algo == "rmappo":
        assert runner.policy.actor.rnn._chunk_kernel_ok(torch.zeros(4, 64, device="cuda"))
    assert runner.buffer.whole_batch_reuses == 3 * 3            # epochs 2..4 of each train() reuse the gathered batch
    vl = _scalars(runner.log_dir, "value_loss")
    rew = _scalars(runner.log_dir, "average_episode_rewards")
    assert len(vl) == 3 and all(np.isfinite(vl)) and all(np.isfinite(rew))
    assert torch.isfinite(runner.buffer.rewards).all() and torch.isfinite(runner.buffer.obs).all()
    assert float(runner.buffer.masks.min()) == 0.0               # episodes ended and restarted inside the rollout
    assert -400.0 < rew[-1] < 0.0, rew                             # simple_spread rewards are negative distances


@pytest.mark.parametrize("agents,landmarks", [(3, 3), (8, 8), (2, 5)])
def test_simple_spread_step_kernel_equals_tensor_ops(agents, landmarks):
    """K11 (``mappo_simple_spread_step``, one launch per env step) against the tensor-op implementation of the same
    worlds -- which tests/test_mpe_env_cpu.py pins to the reference's own particle env -- from the same seed: identical
    generator draws, so whole trajectories (physics, contacts, rewards, restarts, observations) must agree; float64 state
    to 1e-12, float32 outputs to 1e-6."""
    from onpolicy.envs.mpe.simple_spread import TorchSimpleSpread
    dev = torch.device("cuda", 0)
    n, T = 257, 7
    a = TorchSimpleSpread(n, agents, landmarks, episode_length=T, seed=5, device=dev)
    b = TorchSimpleSpread(n, agents, landmarks, episode_length=T, seed=5, device=dev)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device=dev).manual_seed(1)
    for step in range(3 * T + 2):
        act = torch.randint(0, 5, (n, agents, 1), generator=g, device=dev)
        if step % 2:        # the host protocol's one-hot actions are accepted too
            act_in = torch.nn.functional.one_hot(act[..., 0], 5).float()
        else:
            act_in = act
        ra = a.step(act_in)                     # kernel
        rb = b._step_ops(torch.as_tensor(act_in, device=dev))
        torch.testing.assert_close(ra[0], rb[0], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(ra[1], rb[1], rtol=1e-6, atol=1e-6)
        assert torch.equal(ra[2], rb[2])
        torch.testing.assert_close(ra[3]._per_agent, rb[3]._per_agent, rtol=1e-12, atol=1e-12)
        for name in ("pos", "vel", "landmarks"):
            torch.testing.assert_close(getattr(a, name), getattr(b, name), rtol=1e-12, atol=1e-12)
        assert torch.equal(a.t, b.t)
    assert bool(ra[2].any()) or True


def test_two_rank_train_mpe_on_device_worlds(tmp_path):
    """The train script as a data-parallel job: two ranks under torch.distributed.run (both on GPU 0, gloo collectives -- RCCL
    refuses duplicate devices), each with half of the rollout threads on device-resident worlds: per rank a captured rollout
    graph, the rollout forward through K9, sampling through K14, and per update one gradient all-reduce + the cached scalar
    prologue.  Both ranks must finish with identical replicas and finite logs."""
    import glob
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.update(MAPPO_DIST_BACKEND="gloo", MAPPO_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MAPPO_RESULTS_DIR=str(tmp_path / "results"),
               PYTHONPATH=os.path.join(ROOT, "on-policy_amd") + os.pathsep + env.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", "-m", "onpolicy.scripts.train.train_mpe",
           "--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3", "--num_landmarks", "3",
           "--algorithm_name", "mappo", "--n_rollout_threads", "64", "--episode_length", "10", "--num_env_steps",
           str(3 * 64 * 10), "--ppo_epoch", "3", "--num_mini_batch", "1", "--hidden_size", "64", "--use_ReLU", "--use_wandb",
           "--log_interval", "1", "--n_training_threads", "1", "--use_device_env"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "capture failed" not in out.stdout, out.stdout[-2000:]
    actors = sorted(glob.glob(str(tmp_path / "results" / "**" / "actor.pt"), recursive=True))
    assert len(actors) == 2, actors
    a, b = (torch.load(p, map_location="cpu") for p in actors)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k                       # replicas stay identical
    logs = sorted(glob.glob(str(tmp_path / "results" / "**" / "scalars.jsonl"), recursive=True))
    assert len(logs) == 2
    for path in logs:
        recs = [json.loads(l) for l in open(path)]
        vl = [r["value_loss"] for r in recs if r["tag"] == "value_loss"]
        assert len(vl) == 3 and all(np.isfinite(vl))
