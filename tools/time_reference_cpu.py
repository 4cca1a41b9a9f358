#!/usr/bin/env python
"""Times the REFERENCE's own CPU path -- ``SharedReplayBuffer.compute_returns`` (/root/reference/onpolicy/utils/
shared_buffer.py:179) + ``R_MAPPO.train`` (algorithms/r_mappo/r_mappo.py:171) -- on bench.py's synthetic workloads,
imported in place through oracle/ref_import.py (the reference is never copied).  Test / measurement infrastructure:
it only runs where /root/reference is mounted (the build container; the GPU box has no reference), so its output is
committed as profiles/r02_cpu_reference.json and bench.py attaches that record next to its live CPU leg.

    python tools/time_reference_cpu.py [--workloads ns ns_rnn cfg2] [--out profiles/r02_cpu_reference.json]

Two thread settings per workload, as SURVEY.md section 8d asks: ``torch.set_num_threads(1)`` (what the shipped
scripts run: --n_training_threads 1, train_mpe_spread.sh:16) and ``os.cpu_count()``.  The full north-star batch
(28 GiB buffer + 30 GiB of gather copies) does not fit this container, so N (n_rollout_threads) is reduced and
everything else (T, A, dims, ppo_epoch, num_mini_batch, hyper-parameters) kept: env-steps/s = T * N / wall-clock is
a per-sample rate and carries over to the full N (the reference's cost is linear in the number of samples).
"""
import argparse
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# workload -> reduced n_rollout_threads for the timing (the GPU bench runs the full N of bench.WORKLOADS)
SAMPLE_N = {"ns": 64, "ns_rnn": 16, "cfg2": 128, "cfg3": 128, "smac": 8, "hanabi": 16}
# --port-vs-reference repeats every cell: smaller samples (the per-sample rate is what is compared)
PAIR_SAMPLE_N = {"ns": 16, "ns_rnn": 8, "cfg2": 64, "cfg3": 64, "smac": 4, "hanabi": 8}


def time_one(name, threads):
    """Runs in a fresh process (the reference registers itself as ``onpolicy``)."""
    import importlib.util
    import numpy as np
    import torch
    from oracle import ref_import
    ref = ref_import.load_reference()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    # bench.py puts the product's package on sys.path but imports it lazily: only its workload table is used here
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    wl = bench.WORKLOADS[name]
    n = SAMPLE_N[name]
    torch.set_num_threads(threads)
    argv = ["--episode_length", str(wl["T"]), "--n_rollout_threads", str(n)] + wl["flags"]
    args = ref.get_config().parse_known_args(argv)[0]
    args.use_recurrent_policy = bool(wl["recurrent"])          # train_mpe.py:68-80 of the reference
    args.use_naive_recurrent_policy = False
    spaces = ref.Box((wl["Do"],)), ref.Box((wl["Ds"],)), ref.Discrete(wl["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = ref.R_MAPPOPolicy(args, *spaces, device=torch.device("cpu"))
    trainer = ref.R_MAPPO(args, policy, device=torch.device("cpu"))
    buf = ref.SharedReplayBuffer(args, wl["A"], *spaces)
    rng = np.random.default_rng(0)
    f32 = np.float32
    for field in ("share_obs", "obs", "rewards", "rnn_states", "rnn_states_critic"):
        getattr(buf, field)[...] = rng.standard_normal(getattr(buf, field).shape, dtype=f32)
    buf.value_preds[:-1] = rng.standard_normal(buf.value_preds[:-1].shape, dtype=f32)
    buf.actions[...] = rng.integers(0, wl["na"], buf.actions.shape).astype(f32)
    buf.action_log_probs[...] = -np.log(wl["na"])
    buf.masks[...] = (rng.random(buf.masks.shape) >= 1.0 / 25).astype(f32)
    nv = rng.standard_normal(buf.value_preds.shape[1:], dtype=f32)
    trainer.prep_training()
    t0 = time.perf_counter()
    buf.compute_returns(nv, trainer.value_normalizer)
    t1 = time.perf_counter()
    info = trainer.train(buf)
    buf.after_update()
    t2 = time.perf_counter()
    return {"workload": name, "label": wl["label"], "T": wl["T"], "n_rollout_threads_timed": n,
            "n_rollout_threads_full": wl["N"], "agents": wl["A"], "torch_threads": threads,
            "compute_returns_s": round(t1 - t0, 4), "train_s": round(t2 - t1, 3),
            "env_steps_per_s": round(wl["T"] * n / (t2 - t0), 1),
            "value_loss": float(info["value_loss"])}


def time_port(name, threads):
    """The PORT that bench.py's live ``cpu_baseline`` leg runs on the GPU box's host (oracle buffer: C restatement of
    compute_returns + numpy gathers; this repo's R_MAPPO / networks on CPU tensors), here at the SAME n_rollout_threads and
    thread count as the reference run above and on the same machine: port / reference ties the two numbers together."""
    import importlib.util
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracle import oracle
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    wl = bench.WORKLOADS[name]
    n = SAMPLE_N[name]
    torch.set_num_threads(threads)
    args = bench.make_args(wl, n)
    spaces = bench.Box((wl["Do"],)), bench.Box((wl["Ds"],)), bench.Discrete(wl["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces)
    trainer = R_MAPPO(args, policy)
    buf = oracle.OracleBuffer(args, wl["A"], *spaces)
    rng = np.random.default_rng(0)
    f32 = np.float32
    for field in ("share_obs", "obs", "rewards", "rnn_states", "rnn_states_critic"):
        getattr(buf, field)[...] = rng.standard_normal(getattr(buf, field).shape, dtype=f32)
    buf.value_preds[:-1] = rng.standard_normal(buf.value_preds[:-1].shape, dtype=f32)
    buf.actions[...] = rng.integers(0, wl["na"], buf.actions.shape).astype(f32)
    buf.action_log_probs[...] = -np.log(wl["na"])
    buf.masks[...] = (rng.random(buf.masks.shape) >= 1.0 / 25).astype(f32)
    nv = rng.standard_normal(buf.value_preds.shape[1:], dtype=f32)
    trainer.prep_training()
    t0 = time.perf_counter()
    buf.compute_returns(nv, trainer.value_normalizer)
    t1 = time.perf_counter()
    info = trainer.train(buf)
    buf.after_update()
    t2 = time.perf_counter()
    return {"workload": name, "kind": "port", "T": wl["T"], "n_rollout_threads_timed": n, "torch_threads": threads,
            "compute_returns_s": round(t1 - t0, 4), "train_s": round(t2 - t1, 3),
            "env_steps_per_s": round(wl["T"] * n / (t2 - t0), 1), "value_loss": float(info["value_loss"])}


def port_vs_reference(workloads, out_path, repeats=3):
    """Reference and port back to back on this machine, same N, 1 thread and all cores -> out_path.  The build container
    is a shared microVM whose speed drifts by integer factors over minutes (round 6: the same reference run took 17 s and
    50 s an hour apart), so every (workload, threads) cell is ``repeats`` ALTERNATING reference / port pairs and the
    record keeps the pair with the median ratio (+ all ratios): the ratio of two runs that are seconds apart is what the
    machine can measure, absolute rates are not."""
    runs = []
    for name in workloads:
        for threads in (1, os.cpu_count() or 1):
            pairs = []
            for _ in range(repeats):
                pair = {}
                for mode in ("--one", "--one-port"):
                    out = subprocess.run([sys.executable, os.path.abspath(__file__), mode, name, str(threads), "--pair-sample"],
                                         capture_output=True, text=True, check=True)
                    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
                    pair["port" if mode == "--one-port" else "reference"] = rec
                    print(rec, flush=True)
                pair["port_over_reference"] = round(pair["port"]["env_steps_per_s"] / pair["reference"]["env_steps_per_s"], 3)
                pairs.append(pair)
            pairs.sort(key=lambda p: p["port_over_reference"])
            best = pairs[len(pairs) // 2]
            best["all_port_over_reference"] = [p["port_over_reference"] for p in pairs]
            runs.append(best)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    doc = {"what": "bench.py's CPU port (oracle buffer + this repo's trainer on CPU tensors) and the reference's own "
                   "compute_returns + R_MAPPO.train, same machine, same n_rollout_threads, same thread count; per cell the "
                   "median-ratio pair of %d alternating reference / port pairs" % repeats,
           "commit": commit,
           "host": {"cpu": cpu, "logical_cores": os.cpu_count(), "where": "build container"}, "runs": runs}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--port-vs-reference", metavar="OUT", help="time the port next to the reference, write OUT")
    ap.add_argument("--one-port", nargs=2, metavar=("WORKLOAD", "THREADS"), help=argparse.SUPPRESS)
    ap.add_argument("--workloads", nargs="+", default=["ns", "ns_rnn", "cfg2"])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference.json"))
    ap.add_argument("--one", nargs=2, metavar=("WORKLOAD", "THREADS"), help=argparse.SUPPRESS)
    ap.add_argument("--pair-sample", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--repeats", type=int, default=3)
    opt = ap.parse_args()
    if opt.pair_sample:
        SAMPLE_N.update(PAIR_SAMPLE_N)
    if opt.one:
        print("RESULT " + json.dumps(time_one(opt.one[0], int(opt.one[1]))))
        sys.exit(0)
    if opt.one_port:
        print("RESULT " + json.dumps(time_port(opt.one_port[0], int(opt.one_port[1]))))
        sys.exit(0)
    if opt.port_vs_reference:
        port_vs_reference(opt.workloads, opt.port_vs_reference, opt.repeats)
        sys.exit(0)
    import torch
    runs = []
    for name in opt.workloads:
        for threads in (1, os.cpu_count() or 1):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, str(threads)],
                                 capture_output=True, text=True, check=True)
            rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
            print(rec, flush=True)
            runs.append(rec)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    doc = {"what": "reference marlbenchmark/on-policy CPU path: SharedReplayBuffer.compute_returns + R_MAPPO.train "
                   "(onpolicy/utils/shared_buffer.py:179, onpolicy/algorithms/r_mappo/r_mappo.py:171), imported in place",
           "host": {"cpu": cpu, "logical_cores": os.cpu_count(), "machine": platform.machine(),
                    "where": "build container (the GPU box has no /root/reference)"},
           "torch": torch.__version__, "metric": "env-steps/s = T * n_rollout_threads / (compute_returns + train)",
           "note": "n_rollout_threads reduced to fit host memory / a bounded run; the rate is per sample and carries "
                   "over to the full N",
           "runs": runs}
    with open(opt.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", opt.out)
