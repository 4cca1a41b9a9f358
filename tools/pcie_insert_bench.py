#!/usr/bin/env python
"""GPU-box measurement of the caller side of the boundary: ``SharedReplayBuffer.insert`` fed with HOST arrays (what a
host-side VecEnv hands over every rollout step; reference onpolicy/runner/shared/mpe_runner.py:126-139 ->
shared_buffer.py:90-123), at the north-star shapes (57 MB per step, share_obs 50 MB).  Compares the pinned single-copy
staging path with one pageable ``.to(device)`` per field, and reports what a 400-step rollout adds to an iteration.

    python tools/pcie_insert_bench.py [--threads 4096] [--steps 40]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(opt):
    import numpy as np
    import torch
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A, Do, Ds, na = 400, opt.threads, 8, 48, 384, 5
    dev = torch.device("cuda", 0)
    args = make_args(episode_length=T, n_rollout_threads=N)
    buf = SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(na), device=dev)
    rng = np.random.default_rng(0)
    f = lambda *s: rng.standard_normal(s, dtype=np.float32)
    step = dict(share_obs=f(N, A, Ds), obs=f(N, A, Do), rnn_states_actor=np.zeros((N, A, 1, 64), np.float32),
                rnn_states_critic=np.zeros((N, A, 1, 64), np.float32), actions=f(N, A, 1), action_log_probs=f(N, A, 1),
                value_preds=f(N, A, 1), rewards=f(N, A, 1), masks=np.ones((N, A, 1), np.float32))
    nbytes = sum(v.nbytes for k, v in step.items() if not k.startswith("rnn"))
    for _ in range(3):
        buf.insert(**step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call = 0.0
    for _ in range(opt.steps):
        c0 = time.perf_counter()
        buf.insert(**step)
        call += time.perf_counter() - c0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ok = bool(torch.equal(buf.share_obs[buf.step if buf.step else T].cpu(), torch.from_numpy(step["share_obs"])))
    return {"pinned": os.environ.get("MAPPO_PINNED_INSERT", "1") != "0", "bytes_per_step": nbytes,
            "host_ms_per_insert_call": round(1e3 * call / opt.steps, 3),
            "ms_per_step_until_data_in_hbm": round(1e3 * wall / opt.steps, 3),
            "GB_per_s": round(nbytes * opt.steps / wall / 1e9, 1), "s_per_400_step_rollout": round(400 * wall / opt.steps, 3),
            "data_verified": ok}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--one", action="store_true")
    opt = ap.parse_args()
    if opt.one:
        print("RESULT " + json.dumps(run(opt)))
        sys.exit(0)
    res = []
    for pinned in ("1", "0"):
        env = dict(os.environ, MAPPO_PINNED_INSERT=pinned)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", "--threads", str(opt.threads),
                              "--steps", str(opt.steps)], capture_output=True, text=True, env=env, check=True)
        res.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    print(json.dumps({"workload": "insert() of one north-star rollout step from host arrays", "results": res}))
