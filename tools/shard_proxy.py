#!/usr/bin/env python
"""Strong-scaling proxy on ONE GPU (no multi-GPU node was available to the builder): the north-star step at the thread
counts one rank owns in a 2 / 4 / 8-GPU job, with the collectives of the data-parallel path switched on
(MAPPO_FORCE_DIST=1: a single rank goes through the same RCCL calls -- gradient bucket all-reduce, statistics
all-reduces -- so their launch cost is in the step; what a single rank cannot show is the xGMI transfer itself).

    python tools/shard_proxy.py [--out profiles/r03_shard_proxy.json]

Predicted speed-up at W GPUs = t(N) / t(N / W): per-rank work shrinks W-fold, the collectives stay.  The JSON also
carries the all-reduce's own device time per step, and how many milliseconds of inter-GPU latency per step the
prediction can absorb before the speed-up drops below the north star's 6x.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(threads, forced, workload):
    env = dict(os.environ)
    if forced:
        env["MAPPO_FORCE_DIST"] = "1"
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        env.setdefault("MASTER_PORT", "29577")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--threads", str(threads),
           "--steps", "5", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError("bench.py printed no JSON line:\n" + out.stdout[-2000:] + out.stderr[-2000:])
    return json.loads(line[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_shard_proxy.json"))
    ap.add_argument("--workload", default="ns")
    opt = ap.parse_args()
    full = None
    runs = []
    for threads in (4096, 2048, 1024, 512):
        rec = bench(threads, True, opt.workload)
        runs.append({"threads": threads, "ranks_this_stands_for": 4096 // threads, "ms_per_step": rec["ms_per_step"],
                     "env_steps_per_s_of_this_shard": rec["value"], "grad_allreduce": rec["grad_allreduce"],
                     "rccl_ranks": rec["rccl_ranks"], "roofline_frac": (rec.get("roofline") or {}).get("frac"),
                     "roofline_backward_frac": (rec.get("roofline_mlp_backward") or {}).get("frac"),
                     "roofline_gae_frac": (rec.get("roofline_gae") or {}).get("frac")})
        if threads == 4096:
            full = rec["ms_per_step"]
        print(runs[-1], flush=True)
    plain = bench(4096, False, opt.workload)
    pred = []
    for r in runs[1:]:
        w = r["ranks_this_stands_for"]
        speed = full / r["ms_per_step"]
        # inter-GPU time per step that would still leave 6x at 8 GPUs (or the same fraction, 0.75 W, at fewer)
        slack = full / (0.75 * w) - r["ms_per_step"]
        pred.append({"gpus": w, "predicted_speedup": round(speed, 2), "efficiency": round(speed / w, 3),
                     "ms_per_step_budget_left_for_xgmi_at_0.75_efficiency": round(slack, 3)})
    doc = {"what": "single-GPU shard proxy of the strong-scaling curve (north star, T=400 A=8; N = 4096 / W rollout threads "
                   "per rank), collectives issued through RCCL with one rank (MAPPO_FORCE_DIST=1)",
           "ms_per_step_without_collectives_N4096": plain["ms_per_step"], "runs": runs, "prediction": pred,
           "caveat": "xGMI transfer time of the 151 KB gradient bucket and of the two scalar all-reduces per update is not "
                     "in these numbers (one rank moves nothing); 20 updates per step"}
    os.makedirs(os.path.dirname(os.path.abspath(opt.out)), exist_ok=True)
    with open(opt.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["prediction"]))


if __name__ == "__main__":
    main()
