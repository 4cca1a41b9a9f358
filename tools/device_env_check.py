#!/usr/bin/env python
"""GPU-box check of the device-resident rollout loop (TorchSimpleSpread + shared MPE runner + HBM buffer): trains
simple_spread for a few episodes with --use_device_env and with the host env, prints FPS of both.

    python tools/device_env_check.py [--threads 4096] [--episodes 3]
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
    ap.add_argument("--threads", type=int, default=4096)
    ap.add_argument("--episodes", type=int, default=3)
    ap.add_argument("--episode_length", type=int, default=25)
    opt = ap.parse_args()
    os.environ.setdefault("MAPPO_RESULTS_DIR", tempfile.mkdtemp())
    from onpolicy.scripts.train import train_mpe
    steps = opt.episodes * opt.episode_length * opt.threads
    argv = ["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3", "--num_landmarks", "3",
            "--algorithm_name", "mappo", "--n_rollout_threads", str(opt.threads), "--episode_length",
            str(opt.episode_length), "--num_env_steps", str(steps), "--ppo_epoch", "10", "--use_ReLU", "--use_wandb",
            "--log_interval", "1000", "--save_interval", "1000"]
    import torch
    for extra in (["--use_device_env"], []):
        t0 = time.time()
        runner = train_mpe.main(argv + extra)
        dt = time.time() - t0
        # the rollout loop alone (collect -> env step -> insert), after everything is warm: what row f1 is about
        n_steps = 4 * opt.episode_length
        torch.cuda.synchronize()
        r0 = time.time()
        for i in range(n_steps):
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = runner.collect(i % opt.episode_length)
            obs, rewards, dones, infos = runner.envs.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))
        torch.cuda.synchronize()
        rdt = time.time() - r0
        print("%-18s %s: whole run %.2f s (%.0f env-steps/s incl. start-up and updates), mean reward %.3f; rollout loop "
              "alone %.3f ms per step = %.0f env-steps/s" % (
                  " ".join(extra) or "host env", type(runner.envs).__name__, dt, steps / dt,
                  float(runner.buffer.rewards.mean()), 1e3 * rdt / n_steps, opt.threads * n_steps / rdt))
