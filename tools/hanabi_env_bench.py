#!/usr/bin/env python
"""Steps/s of the batched Hanabi stepper (uniformly random legal play), optionally next to the reference's env
(needs /root/reference, `make -C oracle ref`; the reference side runs in a subprocess because it registers its own
``onpolicy`` package).

    python tools/hanabi_env_bench.py [--tables 1024] [--players 5] [--steps 200] [--reference]
"""
import argparse
import os
import subprocess
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ours(opt):
    sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
    from onpolicy.envs.hanabi.batch import HanabiBatchVecEnv
    args = types.SimpleNamespace(hanabi_name=opt.game, num_agents=opt.players, use_obs_instead_of_state=opt.all_obs)
    vec = HanabiBatchVecEnv(args, [1 + 1000 * i for i in range(opt.tables)])
    rng = np.random.default_rng(0)
    obs, share, avail = vec.reset()
    t0 = time.perf_counter()
    moves = 0
    for _ in range(opt.steps):
        # one uniformly random legal move per table: argmax of noise over the legal entries
        a = np.argmax(rng.random(avail.shape, dtype=np.float32) * avail, axis=1)
        obs, share, rewards, dones, infos, avail = vec.step(a[:, None])
        moves += opt.tables
        done = np.array([d is True for d in dones])
        if done.any():
            o2, s2, a2 = vec.reset(done)
            avail = np.where(done[:, None], a2, avail)
    dt = time.perf_counter() - t0
    print("batched stepper : %d tables x %d steps, %.0f env steps/s (%.2f ms per batched step, obs %d, share %d)"
          % (opt.tables, opt.steps, moves / dt, 1e3 * dt / opt.steps, obs.shape[1], share.shape[1]))


def reference(opt):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_golden_hanabi as mg
    HanabiEnv = mg.load_reference_env()
    args = types.SimpleNamespace(hanabi_name=opt.game, num_agents=opt.players, use_obs_instead_of_state=opt.all_obs)
    env = HanabiEnv(args, 1)
    rng = np.random.default_rng(0)
    obs, share, avail = env.reset()
    n = max(50, min(2000, opt.steps * 4))
    t0 = time.perf_counter()
    for _ in range(n):
        a = rng.choice(np.nonzero(avail)[0])
        obs, share, rewards, done, info, avail = env.step([a])
        if done:
            obs, share, avail = env.reset()
    dt = time.perf_counter() - t0
    print("reference env   : 1 env x %d steps, %.0f env steps/s per process" % (n, n / dt))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="Hanabi-Full")
    ap.add_argument("--players", type=int, default=5)
    ap.add_argument("--tables", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--all-obs", action="store_true")
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--_reference_child", action="store_true", help=argparse.SUPPRESS)
    opt = ap.parse_args()
    if opt._reference_child:
        reference(opt)
    else:
        ours(opt)
        if opt.reference:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--_reference_child"] +
                           [a for a in sys.argv[1:] if a != "--reference"], check=True)
