#!/usr/bin/env python
"""GPU-box check of the Hanabi rollout loop (BASELINE.json configs[4] per-GPU shapes: Hanabi-Full, 5 players, 1024
tables per GPU, episode_length 100): trains a few episodes through ``train_hanabi_forward`` and prints where an
episode's wall-clock goes -- collect (batched stepper + policy + turn bookkeeping on the device), buffer insert,
compute + train.

    python tools/hanabi_rollout_check.py [--tables 1024] [--players 5] [--episodes 3]
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=1024)
    ap.add_argument("--players", type=int, default=5)
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--episode_length", type=int, default=100)
    ap.add_argument("--game", default="Hanabi-Full")
    opt = ap.parse_args()
    os.environ.setdefault("MAPPO_RESULTS_DIR", tempfile.mkdtemp())
    import torch
    from onpolicy.runner.shared import hanabi_runner_forward as hr
    from onpolicy.scripts.train import train_hanabi_forward

    acc = {"collect": 0.0, "train": 0.0, "insert": 0.0}

    def timed(cls, name, key, sync=True):
        fn = getattr(cls, name)

        def wrapper(self, *a, **k):
            t0 = time.perf_counter()
            out = fn(self, *a, **k)
            if sync:
                torch.cuda.synchronize()
            acc[key] += time.perf_counter() - t0
            return out
        setattr(cls, name, wrapper)
    timed(hr.HanabiRunner, "collect", "collect")
    timed(hr.HanabiRunner, "train", "train")
    timed(hr.HanabiRunner, "compute", "train")
    steps = opt.episodes * opt.episode_length * opt.tables
    t0 = time.time()
    runner = train_hanabi_forward.main([
        "--env_name", "Hanabi", "--hanabi_name", opt.game, "--num_agents", str(opt.players), "--algorithm_name", "mappo",
        "--n_rollout_threads", str(opt.tables), "--episode_length", str(opt.episode_length), "--num_env_steps", str(steps),
        "--ppo_epoch", "15", "--hidden_size", "512", "--layer_N", "2", "--use_wandb", "--log_interval", "1000",
        "--save_interval", "1000", "--use_ReLU"])
    dt = time.time() - t0
    print("%s, %d players, %d tables, %d episodes of %d buffer steps: %.2f s total; collect %.2f s (%.2f ms per buffer "
          "step, %d real moves -> %.0f moves/s), compute + train %.2f s, rest %.2f s"
          % (opt.game, opt.players, opt.tables, opt.episodes, opt.episode_length, dt, acc["collect"],
             1e3 * acc["collect"] / (opt.episodes * opt.episode_length), runner.true_total_num_steps,
             runner.true_total_num_steps / max(acc["collect"], 1e-9), acc["train"], dt - acc["collect"] - acc["train"]))
