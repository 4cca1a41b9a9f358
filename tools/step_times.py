#!/usr/bin/env python
"""Per-step wall times of one bench.py workload (every step fenced by a device synchronisation): is a workload's step time
one number or a mixture?  (Round 6: config 3's line is bimodal between processes, 47.5 vs 53.7 ms.)

    python tools/step_times.py --workload cfg3 --steps 40 [--mem-stats]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package on sys.path)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--threads", type=int, default=None)
    opt = ap.parse_args()
    wl = dict(bench.WORKLOADS[opt.workload])
    if opt.threads:
        wl["N"] = opt.threads
    dev = torch.device("cuda", 0)
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    args = bench.make_args(wl, wl["N"], ["--sampler_rng", "device"])
    spaces = bench.Box((wl["Do"],)), bench.Box((wl["Ds"],)), bench.Discrete(wl["na"])
    torch.manual_seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, wl["A"], *spaces, device=dev)
    nv = bench.fill_synthetic(buf, wl, seed=1234)
    trainer.prep_training()
    times, allocs = [], []
    for i in range(opt.steps):
        torch.cuda.synchronize(dev)
        a0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
        t0 = time.perf_counter()
        buf.compute_returns(nv, trainer.value_normalizer)
        trainer.train(buf)
        buf.after_update()
        torch.cuda.synchronize(dev)
        times.append(round(1e3 * (time.perf_counter() - t0), 3))
        allocs.append(torch.cuda.memory_stats(dev)["num_device_alloc"] - a0)
    print(json.dumps({"workload": opt.workload, "ms": times, "device_allocs_per_step": allocs,
                      "reserved_GB": round(torch.cuda.memory_reserved(dev) / 1e9, 2)}))


if __name__ == "__main__":
    main()
