#!/usr/bin/env python
"""Strong-scaling proxy on ONE GPU (no multi-GPU node was available to the builder): the north-star step at the thread
counts one rank owns in a 2 / 4 / 8-GPU job, with the collectives of the data-parallel path switched on
(MAPPO_FORCE_DIST=1: a single rank goes through the same RCCL calls -- gradient bucket all-reduce, statistics
all-reduces -- so their launch cost is in the step; what a single rank cannot show is the xGMI transfer itself).

    python tools/shard_proxy.py [--workloads ns,smac,hanabi] [--out profiles/r04_shard_proxy.json]

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


def bench(threads, forced, workload, steps):
    env = dict(os.environ)
    if forced:
        env["MAPPO_FORCE_DIST"] = "1"
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        env.setdefault("MASTER_PORT", "29577")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--threads", str(threads),
           "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-f32-mfma"]
    # (--no-f32-mfma: the proxy needs the default arithmetic only.  Round 6: the float32 sibling steps of the 1024-thread
    # Hanabi shard sent TunableOp into minutes of tuning for GEMM shapes that are not in the shipped table -- the one-rank
    # RCCL watchdog then aborted the run after 600 s: profiles/r06_shard_proxy.json "incident".)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise RuntimeError("bench.py printed no JSON line:\n" + out.stdout[-2000:] + out.stderr[-2000:])
    return json.loads(line[-1])


# workload -> (global n_rollout_threads, rank counts to stand in for, timed steps, what BASELINE.json says about it)
JOBS = {
    "ns": (4096, (1, 2, 4, 8), 5, "north star, T=400 A=8, strong scaling 1/2/4/8 GPUs"),
    "smac": (512, (1, 8), 5, "BASELINE.json configs[3]: SMAC MMM2, 512 threads sharded 64 per GPU x 8, GRU chunk 10 "
                             "(scripts/train_smac_scripts/train_smac_MMM2.sh)"),
    "hanabi": (8192, (1, 8), 2, "BASELINE.json configs[4]: Hanabi-Full 5p, 8192 threads = 1024 per GPU x 8, 9.8 MB "
                                "gradient bucket x 15 updates per step (scripts/train_hanabi_forward.sh:16-17)"),
    "cfg2": (1024, (1, 8), 5, "BASELINE.json configs[1] shapes"),
    "cfg3": (4096, (1, 8), 5, "BASELINE.json configs[2] shapes"),
}


def run_job(workload):
    n_full, worlds, steps, what = JOBS[workload]
    full = None
    runs = []
    for w in worlds:
        threads = n_full // w
        rec = bench(threads, True, workload, steps)
        sc = rec.get("scalar_allreduce") or {}
        runs.append({"threads": threads, "ranks_this_stands_for": w, "ms_per_step": rec["ms_per_step"],
                     "env_steps_per_s_of_this_shard": rec["value"], "grad_allreduce": rec["grad_allreduce"],
                     "scalar_allreduce_per_step": sc.get("per_step"),
                     "rccl_ranks": rec["rccl_ranks"], "roofline_frac": (rec.get("roofline") or {}).get("frac"),
                     "roofline_backward_frac": (rec.get("roofline_mlp_backward") or {}).get("frac"),
                     "roofline_gae_frac": (rec.get("roofline_gae") or {}).get("frac"),
                     "roofline_gae_back_to_back_frac": ((rec.get("roofline_gae") or {}).get("back_to_back") or {}).get("frac")})
        if w == 1:
            full = rec["ms_per_step"]
        print(runs[-1], flush=True)
    plain = bench(n_full, False, workload, steps)
    pred = []
    for r in runs[1:]:
        w = r["ranks_this_stands_for"]
        speed = full / r["ms_per_step"]
        # inter-GPU time per step that would still leave 6x at 8 GPUs (or the same fraction, 0.75 W, at fewer)
        slack = full / (0.75 * w) - r["ms_per_step"]
        n_coll = (r["grad_allreduce"]["per_step"] or 0) + (r["scalar_allreduce_per_step"] or 0) + 1   # + advantage moments
        pred.append({"gpus": w, "predicted_speedup": round(speed, 2), "efficiency": round(speed / w, 3),
                     "ms_per_step_budget_left_for_xgmi_at_0.75_efficiency": round(slack, 3),
                     "collectives_per_step": n_coll,
                     "latency_budget_per_collective_us_at_0.75_efficiency": round(1e3 * slack / max(1, n_coll), 1)})
    return {"workload": workload, "what": what, "global_threads": n_full,
            "ms_per_step_without_collectives_full_N": plain["ms_per_step"], "runs": runs, "prediction": pred}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_shard_proxy.json"))
    ap.add_argument("--workloads", default="ns,smac,hanabi")
    opt = ap.parse_args()
    jobs = [run_job(w) for w in opt.workloads.split(",")]
    doc = {"what": "single-GPU shard proxy of the data-parallel jobs: the step at the thread count ONE rank owns in a W-GPU "
                   "job, every collective of the path issued through RCCL by one rank (MAPPO_FORCE_DIST=1)",
           "jobs": jobs,
           "caveat": "one rank moves nothing over xGMI: the transfer time of the gradient bucket and of the scalar "
                     "all-reduces is NOT in these numbers; `latency_budget_per_collective_us` is what each collective may "
                     "cost before the job falls below 0.75 W (6x at 8 GPUs)"}
    os.makedirs(os.path.dirname(os.path.abspath(opt.out)), exist_ok=True)
    with open(opt.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps([{j["workload"]: j["prediction"]} for j in jobs]))


if __name__ == "__main__":
    main()
