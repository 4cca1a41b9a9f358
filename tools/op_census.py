#!/usr/bin/env python
"""GPU-box aid: which Python lines of the update launch the small kernels?  Runs the bench workload (default: the north
star at 512 threads = one rank's shard of an 8-GPU job), profiles ONE step with torch.profiler and prints device-kernel
launches grouped by the innermost onpolicy/ source line.

    python tools/op_census.py [--threads 512] [--workload ns]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--workload", default="ns")
    opt = ap.parse_args()
    import bench
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    wl = dict(bench.WORKLOADS[opt.workload])
    wl["N"] = opt.threads
    dev = torch.device("cuda", 0)
    args = bench.make_args(wl, wl["N"], ["--sampler_rng", "device"])
    spaces = bench.Box((wl["Do"],)), bench.Box((wl["Ds"],)), bench.Discrete(wl["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, wl["A"], *spaces, device=dev)
    next_value = bench.fill_synthetic(buf, wl, seed=1234)
    trainer.prep_training()

    def step():
        buf.compute_returns(next_value, trainer.value_normalizer)
        trainer.train(buf)
        buf.after_update()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by_line = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.device_type is not None and str(ev.device_type).endswith("CPU") and ev.name.startswith("aten::") and \
                ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") or \
                (ev.name.startswith("aten::") and ev.cpu_parent is None):
            where = "?"
            for fr in (ev.stack or []):
                if "onpolicy/" in fr:
                    where = fr.split("onpolicy/")[-1]
                    break
            by_line[where] += 1
            by_op[ev.name] += 1
    print("top-level aten ops per step: %d" % sum(by_op.values()))
    for k, v in by_line.most_common(45):
        print("%5d  %s" % (v, k))
    print("--- by op")
    for k, v in by_op.most_common(25):
        print("%5d  %s" % (v, k))


if __name__ == "__main__":
    main()
