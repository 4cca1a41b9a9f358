#!/usr/bin/env python
"""BASELINE.json configs[2] end to end on one MI355X: MPE simple_spread, 3 agents, n_rollout_threads=4096,
episode_length=400, ppo_epoch=10 -- rollout (policy forward -> K11 env step on the device -> K2 insert) AND update
(compute_returns + R_MAPPO.train through K9) through the unmodified train script / runner (reference
onpolicy/scripts/train/train_mpe.py, runner/shared/mpe_runner.py:16-79), worlds resident on the device
(--use_device_env).

    python tools/cfg3_end_to_end.py [--threads 4096] [--episode_length 400] [--iterations 3] [--out file.json]

Prints one JSON line: env-steps/s of a whole iteration (rollout + update) in steady state, and the two phases apart.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=4096)
    ap.add_argument("--episode_length", type=int, default=400)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--algorithm_name", default="mappo")
    ap.add_argument("--out", default=None)
    opt = ap.parse_args()
    os.environ.setdefault("MAPPO_RESULTS_DIR", tempfile.mkdtemp())
    import torch
    from onpolicy.scripts.train import train_mpe
    T, N = opt.episode_length, opt.threads
    argv = ["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3", "--num_landmarks", "3",
            "--algorithm_name", opt.algorithm_name, "--n_rollout_threads", str(N), "--episode_length", str(T),
            "--num_env_steps", str(T * N), "--ppo_epoch", "10", "--num_mini_batch", "1", "--use_ReLU", "--gain", "0.01",
            "--lr", "7e-4", "--critic_lr", "7e-4", "--use_wandb", "--log_interval", "1000", "--save_interval", "1000",
            "--use_device_env"]
    t0 = time.time()
    runner = train_mpe.main(argv)           # one whole iteration: builds everything, warms allocator and kernels
    torch.cuda.synchronize()
    first = time.time() - t0
    from onpolicy.algorithms.utils import fused_mlp
    fused_mlp.profile(True)

    def rollout_eager():
        for step in range(T):
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = runner.collect(step)
            obs, rewards, dones, infos = runner.envs.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))

    def rollout_graph():
        runner.trainer.prep_rollout()
        runner.rollout_graph.begin_episode()
        for step in range(T):
            runner.rollout_graph.step()

    def measure(rollout):
        roll, upd = [], []
        for _ in range(opt.iterations):
            torch.cuda.synchronize()
            a = time.perf_counter()
            rollout()
            torch.cuda.synchronize()
            b = time.perf_counter()
            runner.compute()
            info = runner.train()
            torch.cuda.synchronize()
            c = time.perf_counter()
            roll.append(b - a)
            upd.append(c - b)
        return sum(roll) / len(roll), sum(upd) / len(upd), info

    graphed = getattr(runner, "rollout_graph", None) is not None
    er, eu, info = measure(rollout_eager)
    if graphed:
        r, u, info = measure(rollout_graph)
    else:
        r, u = er, eu
    mt = fused_mlp.profile_times()
    fused_mlp.profile(False)
    out = {"config": "BASELINE.json configs[2]: MPE simple_spread, 3 agents, n_rollout_threads=%d, episode_length=%d, "
                     "ppo_epoch=10, %s, 1 x MI355X, worlds on the device (K11)" % (N, T, opt.algorithm_name),
           "rollout": "one captured HIP graph launch + one slab write per step (runner/shared/rollout_graph.py)" if graphed
                      else "eager (collect -> envs.step -> insert, ~30 launches per step)",
           "env_steps_per_s_rollout_plus_update": round(T * N / (r + u), 1),
           "eager_loop_for_comparison": {"env_steps_per_s_rollout_plus_update": round(T * N / (er + eu), 1),
                                         "rollout_s": round(er, 4), "rollout_ms_per_env_step": round(1e3 * er / T, 4),
                                         "update_s": round(eu, 4)},
           "rollout_s": round(r, 4), "rollout_ms_per_env_step": round(1e3 * r / T, 4),
           "rollout_env_steps_per_s": round(T * N / r, 1),
           "update_s": round(u, 4), "update_env_steps_per_s": round(T * N / u, 1),
           "iterations": opt.iterations, "first_iteration_incl_startup_s": round(first, 2),
           "fused_trunk_launches": {k: v[0] for k, v in mt.items()},
           "mean_reward_last_rollout": float(runner.buffer.rewards.mean()),
           "train_info": {k: round(float(v), 6) for k, v in info.items()}}
    line = json.dumps(out)
    print(line)
    if opt.out:
        os.makedirs(os.path.dirname(os.path.abspath(opt.out)), exist_ok=True)
        with open(opt.out, "w") as f:
            f.write(line + "\n")
