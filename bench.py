#!/usr/bin/env python
"""Benchmark of the MAPPO hot path: env-steps/sec through GAE + ppo_update.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|cfg2|cfg3|ns_rnn|smac|hanabi] [--no-cpu-baseline]
                    [--matrix-arithmetic six_term|f32_mfma] [--no-f32-mfma] [--no-workloads]

The default command line (north star, one GPU) also carries `workloads`: the other BASELINE.json configs run for three
steps each by child processes of this script after the timed region (OTHER_WORKLOADS below; outside `value`).

One "step" = one pass of the hot path over one synthetic rollout that is already resident in HBM:
``buffer.compute_returns`` (HIP GAE scan) + ``R_MAPPO.train`` (ppo_epoch x num_mini_batch fused
gathers + PyTorch fwd/bwd + Adam) + ``buffer.after_update``.  Metric (BASELINE.json):
env-steps/sec = T * N / wall-clock, N = GLOBAL number of rollout threads.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the global N rollout threads are
sharded N / world per rank (strong scaling, fixed total work); per update one RCCL all-reduce of the
flat actor+critic gradient bucket plus two tiny statistic all-reduces.

Rank 0 prints ONE JSON line.  ``roofline`` describes the dominant kernel of the step, the fused trunk's forward launch
(K9; hidden 512: K15), timed with events on the launch stream inside the timed region and set against both of its roofs
-- algorithmic float32 FLOPs against the peak of the MFMA instruction the products are formed with (f32 MFMA, or bf16
MFMA / 6 terms under the six-term arithmetic) and algorithmic bytes against the HBM peak; ``bound`` names the nearer one; ``roofline_mlp_backward`` the same for the backward call, ``roofline_gae`` the GAE
scan BASELINE.json's north star names (HBM bound; algorithmic bytes from SURVEY.md section 8d), ``roofline_gather`` the
fused minibatch gather.  ``traffic`` fields quote the committed rocprofv3 PMC passes (``traffic_source`` names the file
each one came from).  ``cpu_baseline`` is the CPU port of the same path (oracle buffer + the same PyTorch trainer on
host cores) on a bounded sample, next to the recorded timings of the reference itself and the port / reference factor
measured on one machine.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL's buffer registration across ranks needs this (no-op if exported)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "on-policy_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32 MFMA peak (v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md); no xf32 / tf32 on gfx950
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md; no sparsity)
ARITHMETIC_TEXT = {
    "six_term": "f32 products from six bf16xbf16 terms of exact 3-way splits, f32 accumulate (K9 / K12 / K15 matrix products; "
                "MAPPO_ARITH_SIX_TERM); everything else f32",
    "f32_mfma": "f32 MFMA (v_mfma_f32_32x32x2_f32; MAPPO_ARITH_F32_MFMA)",
}

WORKLOADS = {
    # north star: simple_spread generalised to 8 agents, flags of train_mpe_spread.sh
    "ns": dict(T=400, N=4096, A=8, Do=48, Ds=384, na=5, cpu_sample_N=128,
               flags=["--algorithm_name", "mappo", "--hidden_size", "64", "--layer_N", "1", "--use_ReLU",
                      "--ppo_epoch", "10", "--num_mini_batch", "1", "--lr", "7e-4", "--critic_lr", "7e-4",
                      "--gain", "0.01"],
               recurrent=False,
               label="synthetic MPE simple_spread x8 agents, T=400 N=4096 A=8, mappo MLP h64, ppo_epoch=10, 1 minibatch"),
    # the north-star shapes with the recurrent policy ("run both mappo and rmappo")
    "ns_rnn": dict(T=400, N=4096, A=8, Do=48, Ds=384, na=5, cpu_sample_N=32,
                   flags=["--algorithm_name", "rmappo", "--hidden_size", "64", "--layer_N", "1", "--use_ReLU",
                          "--ppo_epoch", "10", "--num_mini_batch", "1", "--data_chunk_length", "10", "--lr", "7e-4",
                          "--critic_lr", "7e-4", "--gain", "0.01"],
                   recurrent=True,
                   label="synthetic MPE simple_spread x8 agents, T=400 N=4096 A=8, rmappo GRU h64 chunk 10, "
                         "ppo_epoch=10, 1 minibatch"),
    # BASELINE.json configs[1]
    "cfg2": dict(T=200, N=1024, A=5, Do=30, Ds=150, na=5, cpu_sample_N=64,
                 flags=["--algorithm_name", "mappo", "--hidden_size", "64", "--layer_N", "1", "--use_ReLU",
                        "--ppo_epoch", "10", "--num_mini_batch", "1", "--lr", "7e-4", "--critic_lr", "7e-4"],
                 recurrent=False,
                 label="synthetic T=200 N=1024 A=5, mappo MLP h64, ppo_epoch=10"),
    # BASELINE.json configs[2]: MPE simple_spread itself (3 agents, 3 landmarks: Do = 4 + 2*3 + 4*2 = 18, Ds = 3 * 18,
    # reference envs/mpe/scenarios/simple_spread.py:86-103) at N=4096, T=400, flags of train_mpe_spread.sh:14-17
    "cfg3": dict(T=400, N=4096, A=3, Do=18, Ds=54, na=5, cpu_sample_N=128,
                 flags=["--algorithm_name", "mappo", "--hidden_size", "64", "--layer_N", "1", "--use_ReLU",
                        "--ppo_epoch", "10", "--num_mini_batch", "1", "--lr", "7e-4", "--critic_lr", "7e-4",
                        "--gain", "0.01"],
                 recurrent=False,
                 label="synthetic MPE simple_spread 3 agents, T=400 N=4096 A=3, mappo MLP h64, ppo_epoch=10, 1 minibatch"),
    # BASELINE.json configs[4] shapes (Hanabi-Full, 5 players), feed-forward, hidden 512 x 2 layers
    "hanabi": dict(T=100, N=8192, A=5, Do=1285, Ds=1385, na=48, cpu_sample_N=16,
                   flags=["--algorithm_name", "mappo", "--hidden_size", "512", "--layer_N", "2",
                          "--ppo_epoch", "15", "--num_mini_batch", "1", "--lr", "7e-4", "--critic_lr", "1e-3",
                          "--gain", "0.01"],       # (train_hanabi_forward.sh:15-17 passes no --use_ReLU: ReLU blocks)
                   recurrent=False,
                   label="synthetic Hanabi-Full 5p shapes T=100 N=8192 A=5, mappo MLP h512 x2 (ReLU), ppo_epoch=15"),
    # BASELINE.json configs[3] shapes (SMAC MMM2), recurrent policy, chunk 10
    "smac": dict(T=400, N=512, A=10, Do=370, Ds=435, na=18, cpu_sample_N=8,
                 flags=["--algorithm_name", "rmappo", "--hidden_size", "64", "--layer_N", "1",
                        "--ppo_epoch", "5", "--num_mini_batch", "2", "--data_chunk_length", "10",
                        "--gain", "1"],
                 recurrent=True,
                 label="synthetic SMAC MMM2 shapes T=400 N=512 A=10, rmappo GRU h64 chunk 10, ppo_epoch=5, 2 minibatches"),
}


def csrc_digest():
    """sha256[:16] over the kernel sources (on-policy_amd/csrc/*.hip, *.h, *.cc, Makefile, sorted by name): what ties a
    committed rocprofv3 record to the code it was taken on -- computable on the GPU box, where there is no .git."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "on-policy_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cc")) or fn == "Makefile":
            h.update(fn.encode())
            with open(os.path.join(d, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


class Box(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class Discrete(object):
    def __init__(self, n):
        self.n = n


def make_args(wl, n_threads, extra=()):
    from onpolicy.config import get_config
    argv = ["--episode_length", str(wl["T"]), "--n_rollout_threads", str(n_threads)] + wl["flags"] + list(extra)
    args = get_config().parse_known_args(argv)[0]
    # scripts/train/train_mpe.py:68-80 of the reference: algorithm name decides the recurrent flags
    args.use_recurrent_policy = bool(wl["recurrent"])
    args.use_naive_recurrent_policy = False
    return args


def fill_synthetic(buf, wl, seed):
    """Random-obs / random-reward trajectory with the distributions of SURVEY.md section 8d, generated
    on the device (data: synthetic)."""
    g = torch.Generator(device=buf.device)
    g.manual_seed(seed)
    for name in ("share_obs", "obs", "rewards"):
        getattr(buf, name).normal_(generator=g)
    buf.value_preds[:-1].normal_(generator=g)
    na = wl["na"]
    buf.actions.copy_(torch.randint(0, na, buf.actions.shape, generator=g, device=buf.device).float())
    buf.action_log_probs.fill_(-float(np.log(na)))
    buf.masks.copy_((torch.rand(buf.masks.shape, generator=g, device=buf.device) >= 1.0 / 25).float())
    if wl["recurrent"]:
        buf.rnn_states.normal_(generator=g)
        buf.rnn_states_critic.normal_(generator=g)
        buf.active_masks.copy_((torch.rand(buf.masks.shape, generator=g, device=buf.device) < 0.9).float())
        av = (torch.rand(buf.available_actions.shape, generator=g, device=buf.device) < 0.7).float()
        av[..., 0] = 1.0
        buf.available_actions.copy_(av)
    nv = torch.empty(buf.value_preds.shape[1:], device=buf.device).normal_(generator=g)
    return nv


def cpu_baseline(wl):
    """The same path on host cores: oracle buffer (C restatement of the reference's compute_returns
    and numpy gathers) + the same PyTorch trainer on CPU tensors, on a bounded sample (fewer rollout
    threads, identical T / A / dims / hyper-parameters)."""
    from oracle import oracle
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    # PyTorch's CPU ops stop scaling (and then slow down) beyond a few dozen threads at these
    # batch sizes; 32 is the best case measured on the 256-core host of the MI355X box
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    n = wl["cpu_sample_N"]
    args = make_args(wl, n)
    spaces = Box((wl["Do"],)), Box((wl["Ds"],)), Discrete(wl["na"])
    torch.manual_seed(1)
    policy = R_MAPPOPolicy(args, *spaces)
    trainer = R_MAPPO(args, policy)
    buf = oracle.OracleBuffer(args, wl["A"], *spaces)
    rng = np.random.default_rng(0)
    f32 = np.float32
    for name in ("share_obs", "obs", "rewards", "rnn_states", "rnn_states_critic"):
        getattr(buf, name)[...] = rng.standard_normal(getattr(buf, name).shape, dtype=f32)
    buf.value_preds[:-1] = rng.standard_normal(buf.value_preds[:-1].shape, dtype=f32)
    buf.actions[...] = rng.integers(0, wl["na"], buf.actions.shape).astype(f32)
    buf.action_log_probs[...] = -np.log(wl["na"])
    buf.masks[...] = (rng.random(buf.masks.shape) >= 1.0 / 25).astype(f32)
    nv = rng.standard_normal(buf.value_preds.shape[1:], dtype=f32)
    trainer.prep_training()
    t0 = time.perf_counter()
    buf.compute_returns(nv, trainer.value_normalizer)
    t1 = time.perf_counter()
    trainer.train(buf)
    buf.after_update()
    t2 = time.perf_counter()
    total = t2 - t0
    return {"value": wl["T"] * n / total, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "1 iteration at n_rollout_threads=%d (same T=%d, A=%d, dims, ppo_epoch, minibatches): "
                      "%.3f s compute_returns + %.2f s train" % (n, wl["T"], wl["A"], t1 - t0, t2 - t1)}


def reference_recorded(workload):
    """The REFERENCE's own CPU path (its SharedReplayBuffer.compute_returns + R_MAPPO.train, imported in place) timed
    next to the port by tools/time_reference_cpu.py --port-vs-reference where /root/reference is mounted -- the GPU box
    has no reference, so the newest committed record (profiles/r0N_cpu_port_vs_reference.json: reference and port back
    to back on one machine, same N, same thread counts) is quoted, never re-measured here."""
    doc = fn = None
    for fn in ("r06_cpu_port_vs_reference.json", "r03_cpu_port_vs_reference.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                doc = json.load(f)
            if any(r["reference"]["workload"] == workload for r in doc["runs"]):
                break
            doc = None
        except Exception:
            doc = None
    if doc is None:
        return None
    pairs = [r for r in doc["runs"] if r["reference"]["workload"] == workload]
    return {"kind": "reference", "source": "profiles/%s (tools/time_reference_cpu.py --port-vs-reference)" % fn,
            "commit": doc.get("commit"),
            "port_vs_reference_same_machine": [
                {"torch_threads": r["reference"]["torch_threads"],
                 "reference_env_steps_per_s": r["reference"]["env_steps_per_s"],
                 "port_env_steps_per_s": r["port"]["env_steps_per_s"],
                 "port_over_reference": r["port_over_reference"]} for r in pairs],
            "host": doc["host"]["cpu"] + ", %d logical cores, build container" % doc["host"]["logical_cores"],
            "n_rollout_threads_timed": pairs[0]["reference"]["n_rollout_threads_timed"],
            "note": "same T / agents / dims / ppo_epoch as the GPU run, fewer rollout threads (host memory); env-steps/s "
                    "is a per-sample rate; the live `port` figure divided by port_over_reference ~ the reference on the "
                    "GPU box's host",
            "runs": [{"torch_threads": r["reference"]["torch_threads"], "env_steps_per_s": r["reference"]["env_steps_per_s"],
                      "compute_returns_s": r["reference"]["compute_returns_s"], "train_s": r["reference"]["train_s"]}
                     for r in pairs]}


# ---- the other BASELINE configs on the driver's line ----------------------------------------------------------------
# (VERDICT r5 "next" #1.)  `python bench.py` (north star, one GPU) appends `workloads`: every other BASELINE.json config
# -- and the 64-thread shard one rank of configs[3]'s 8-GPU job owns -- run for a few steps by THIS command line, each
# in a fresh process of this very script (`--workload X`), AFTER the timed region and outside `value`.  An entry is
# the child's own line cut down to the figures a reader compares: value / ms_per_step, arithmetic, the dominant
# launch against its roof, the GAE launch in situ, the CPU leg on a bounded sample.
OTHER_WORKLOADS = (
    # name, bench.py arguments, (steps, warmup) -- short steps are timed over more of them (three 14 ms steps scatter by 25 %) --, what
    ("cfg2", ["--workload", "cfg2"], (30, 5), "BASELINE.json configs[1]"),
    ("cfg3", ["--workload", "cfg3"], (10, 2), "BASELINE.json configs[2] (update phase; rollout + update: tools/cfg3_end_to_end.py)"),
    ("ns_rnn", ["--workload", "ns_rnn", "--cpu-sample-threads", "16"], (3, 1),
     "north-star shapes, recurrent policy (SURVEY 8d: mappo and rmappo)"),
    ("smac", ["--workload", "smac"], (10, 2), "BASELINE.json configs[3] shapes, all 512 threads on one GPU"),
    ("smac_shard64", ["--workload", "smac", "--threads", "64", "--no-cpu-baseline"], (40, 5),
     "BASELINE.json configs[3]: the 64 threads one of 8 ranks owns (single-GPU proxy of the per-rank step, no xGMI time)"),
    ("hanabi", ["--workload", "hanabi", "--cpu-sample-threads", "8"], (3, 1),
     "BASELINE.json configs[4] shapes, all 8192 threads on one GPU"),
)


def compact_line(o, what):
    r, g, cb = o.get("roofline") or {}, o.get("roofline_gae") or {}, o.get("cpu_baseline")
    out = {"what": what, "workload": o["config"]["workload"], "n_rollout_threads": o["config"]["n_rollout_threads"],
           "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
           "warmup": o["warmup"], "dtype": o["dtype"], "arithmetic": o["arithmetic"].split(" (")[0],
           "update_graph_replays_per_step": o.get("update_graph_replays_per_step"),
           "hbm_peak_bytes": (o.get("hbm_peak_bytes_per_rank") or [None])[0],
           "two_streams": o.get("two_streams"),
           "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_ms",
                                              "share_of_step", "timed_in")} if r else None,
           "roofline_gae": {"frac": g.get("frac"), "launch_ms": g.get("launch_ms"),
                            "event_pair_around_launch_frac": (g.get("event_pair_around_launch") or {}).get("frac"),
                            "back_to_back_frac": (g.get("back_to_back") or {}).get("frac")} if g else None}
    if cb is not None:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        if cb.get("reference_equivalent"):
            out["cpu_baseline"]["reference_equivalent"] = {
                k: cb["reference_equivalent"].get(k) for k in ("value", "port_over_reference", "gpu_over_reference_equivalent")}
    else:
        out["cpu_baseline"] = None
    return out


def other_workloads(opt, budget_s):
    """Run OTHER_WORKLOADS as child processes of this script (the GPU is free: the caller dropped its tensors) ->
    {name: compact entry | {"error": ...}}.  `budget_s` bounds the whole leg: a child that would start after it is
    recorded as skipped, a child gets what is left (+ a floor) as its timeout."""
    import subprocess
    t_start = time.perf_counter()
    res = {}
    only = [w for w in os.environ.get("MAPPO_BENCH_WORKLOADS", "").split(",") if w]      # (tests: a subset)
    for name, argv, (k_steps, k_warm), what in OTHER_WORKLOADS:
        if only and name not in only:
            continue
        left = budget_s - (time.perf_counter() - t_start)
        if left < 5:
            res[name] = {"what": what, "skipped": "the leg's time budget (%d s) was spent" % budget_s}
            continue
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + [
            "--steps", str(k_steps), "--warmup", str(k_warm), "--no-f32-mfma", "--no-workloads", "--sampler-rng", opt.sampler_rng,
            "--matrix-arithmetic", opt.matrix_arithmetic] + (["--no-gemm-tuning"] if opt.no_gemm_tuning else [])
        if opt.no_cpu_baseline and "--no-cpu-baseline" not in cmd:
            cmd.append("--no-cpu-baseline")
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=max(60.0, left))
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                res[name] = {"what": what, "error": "exit code %d: %s" % (p.returncode, (p.stderr or "")[-300:])}
                continue
            res[name] = compact_line(json.loads(lines[-1]), what)
            res[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        except subprocess.TimeoutExpired:
            res[name] = {"what": what, "error": "timed out after %.0f s" % max(60.0, left)}
        except Exception as exc:
            res[name] = {"what": what, "error": "%s: %s" % (type(exc).__name__, exc)}
    # the shard runs the same per-sample CPU work as the full config: quote that leg instead of timing it twice
    if res.get("smac_shard64") and res.get("smac") and "cpu_baseline" in res["smac_shard64"]:
        res["smac_shard64"]["cpu_baseline"] = res["smac"].get("cpu_baseline")
    return res


def self_launch(n_gpus):
    """Re-run this command line under ``torch.distributed.run`` with one rank per GPU (what the driver's own
    multi-GPU launch line does) and return its exit code.  The ranks inherit stdout, so rank 0's JSON line is
    this process's output."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--threads", type=int, default=None, help="override the global n_rollout_threads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-threads", type=int, default=None,
                    help="n_rollout_threads of the CPU leg's bounded sample (default: the workload's cpu_sample_N)")
    ap.add_argument("--no-workloads", action="store_true",
                    help="north-star run on one GPU: do not append the other BASELINE configs (`workloads`) to the line")
    ap.add_argument("--matrix-arithmetic", default="six_term", choices=["six_term", "f32_mfma"],
                    help="arithmetic of the K9 / K12 matrix products in the TIMED region (include/mappo_hip.h MAPPO_ARITH_*)")
    ap.add_argument("--no-f32-mfma", "--no-six-term", dest="no_other_arithmetic", action="store_true",
                    help="skip the extra steps (after the timed region) under the other arithmetic form")
    ap.add_argument("--sampler-rng", default="device", choices=["device", "host"])
    ap.add_argument("--gae-scan", action="store_true",
                    help="narrow buffers take the time-parallel GAE scan (--gae_scan: ~1e-6 relative instead of bit-exact)")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave GEMM kernel selection to the library heuristic (onpolicy/utils/gemm_tuning.py)")
    opt = ap.parse_args()

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL) and relay rank 0's line
        sys.exit(self_launch(opt.gpus))

    wl = dict(WORKLOADS[opt.workload])
    if opt.threads:
        wl["N"] = opt.threads
    if opt.cpu_sample_threads:
        wl["cpu_sample_N"] = opt.cpu_sample_threads
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == opt.gpus, "--gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (opt.gpus, world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MAPPO_SINGLE_DEVICE", "0") == "1":
        local_rank = 0          # all ranks on one GPU (test mode, with MAPPO_DIST_BACKEND=gloo)
    assert torch.cuda.is_available(), "bench.py measures the HIP path and needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from onpolicy.utils import dist as mdist
    if world > 1 or os.environ.get("MAPPO_FORCE_DIST", "0") == "1":
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # no version banner on stdout
        mdist.init_from_env(dev)
    if world > 1 and os.environ.get("MAPPO_DIST_BACKEND", "nccl") == "nccl":
        # a multi-GPU line must have been carried by RCCL on `--gpus` ranks -- anything else is not the measurement asked for
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == opt.gpus, \
            "bench.py --gpus %d: expected %d RCCL ranks, got backend=%s world=%s" % (
                opt.gpus, opt.gpus, dist.get_backend() if dist.is_initialized() else None,
                dist.get_world_size() if dist.is_initialized() else None)
    lo, hi = mdist.shard_threads(wl["N"], rank, world)
    n_local = hi - lo

    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy

    args = make_args(wl, n_local, ["--sampler_rng", opt.sampler_rng, "--matrix_arithmetic", opt.matrix_arithmetic] +
                     (["--gae_scan"] if opt.gae_scan else []))
    spaces = Box((wl["Do"],)), Box((wl["Ds"],)), Discrete(wl["na"])
    torch.manual_seed(1)          # identical replicas on every rank (init draws come from the CPU stream)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, wl["A"], *spaces, device=dev)
    next_value = fill_synthetic(buf, wl, seed=1234 + rank)
    trainer.prep_training()
    from onpolicy.utils import gemm_tuning
    tuned = (not opt.no_gemm_tuning) and gemm_tuning.enable()

    def step():
        buf.compute_returns(next_value, trainer.value_normalizer)
        info = trainer.train(buf)
        buf.after_update()
        return info

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if tuned:
        step()      # untimed: any GEMM shape without a stored winner is tuned here, never in the timed region
    for _ in range(opt.warmup):
        step()
    buf.profile_kernels(True)
    from onpolicy.algorithms.utils import fused_mlp
    fused_mlp.profile(True)
    trainer.dp.time_collectives(True)
    scalar0, reused0 = trainer.dp.scalar_collectives, trainer.dp.scales_reused
    replays0 = getattr(getattr(trainer, "_update_graph", None), "replays", 0)
    # Python's cyclic collector: a full collection walks every object alive (modules, fixtures of the process, ...) and fell
    # into the timed region of about one in ten to twenty steps -- ~60 ms, invisible on the north star's 4 s, +12 % on a
    # ten-step config-3 line (47.5 against 53.7 ms between processes; tools/step_times.py).  Everything alive now is
    # moved to the permanent generation: the collector stays on, its passes only look at what the steps themselves create.
    import gc
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        info = step()
    fence()
    elapsed = time.perf_counter() - t0
    kt = buf.kernel_times()
    mt = fused_mlp.profile_times()
    n_coll, coll_ms, coll_bytes = trainer.dp.collective_times()
    trainer.dp.time_collectives(False)
    # (counted here: the steps that follow -- K9 launch timing, the other arithmetic form -- issue collectives of their own)
    scalar_n, reused_n = trainer.dp.scalar_collectives - scalar0, trainer.dp.scales_reused - reused0
    # Small minibatches replay ppo_update from captured HIP graphs (algorithms/r_mappo/update_graph.py): launches inside a graph
    # carry no event pairs, so the K9 launch timings of the roofline objects are then taken from ONE extra step after the timed
    # region with the graphs switched off (same kernels, same shapes); `value` / `ms_per_step` stay the graphed steps'.
    ug = getattr(trainer, "_update_graph", None)
    graph_replays = 0 if ug is None else ug.replays - replays0
    k9_timed_in = "the timed region"
    # ... and small evaluations (<= 2^20 rows: graph-replayed minibatches, the row spans of a hidden-512 update) run actor and
    # critic on two streams (R_MAPPOPolicy.evaluate_logits): a launch that shares the chip with the other network's cannot be
    # set against a roofline.  In both cases the launches are timed one after the other in the extra step.
    two_streams = bool(getattr(policy, "_side_streams", None))
    if (ug is not None and ug.replays > 0 and not mt) or two_streams:
        if ug is not None:
            ug.off = True
        two = os.environ.get("MAPPO_TWO_STREAM_UPDATE")
        os.environ["MAPPO_TWO_STREAM_UPDATE"] = "0"
        try:
            fused_mlp.profile(True)
            step()
            torch.cuda.synchronize(dev)
            mt = fused_mlp.profile_times()
        finally:
            if ug is not None:
                ug.off = False
            if two is None:
                os.environ.pop("MAPPO_TWO_STREAM_UPDATE", None)
            else:
                os.environ["MAPPO_TWO_STREAM_UPDATE"] = two
        k9_timed_in = "one eager one-stream step after the timed region (the timed steps %s)" % " and ".join(
            ([] if not graph_replays else ["replay ppo_update from HIP graphs"]) +
            ([] if not two_streams else ["run actor and critic launches on two streams"]))
    # the GAE launch once more, outside the timed region, back to back (no update phase in between: caches and TLBs as the
    # previous launch left them) -- reported next to the in-situ figure as roofline_gae.back_to_back
    buf.profile_kernels(False)
    reps = 10
    buf.compute_returns(next_value, trainer.value_normalizer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(250000)
    e0.record()
    for _ in range(reps):
        buf.compute_returns(next_value, trainer.value_normalizer)
    e1.record()
    torch.cuda.synchronize(dev)
    gae_b2b_ms = e0.elapsed_time(e1) / reps
    from onpolicy import _native
    gae_variant, buf_gae_exact = _native.lib().mappo_gae_last_variant(), buf._gae_exact
    # Outside the contract's timed region, next to `value`: the same step under the OTHER arithmetic form of the K9 / K12 matrix
    # products (a per-policy choice carried by every call: policy.set_matrix_arithmetic).  The default -- and `value` -- is the
    # six-term form (float32 products from six bf16 x bf16 terms of the operands' exact three-way splits, float32 accumulate:
    # VERDICT r4's ruling, conditions in DESIGN.md section 2); `f32_mfma` is the float32 matrix instruction, measured in this
    # same run.
    other = None
    other_name = "f32_mfma" if opt.matrix_arithmetic == "six_term" else "six_term"
    if not opt.no_other_arithmetic and args.hidden_size in (64, 512):
        policy.set_matrix_arithmetic(other_name)
        # (no tuning of GEMM shapes the shipped table does not hold in this leg: under f32_mfma a hidden-512 shard with an
        # unseen row count sent TunableOp into minutes of benchmarking -- unseen shapes take the library's heuristic pick here)
        retune = bool(tuned) and gemm_tuning._TUNE_NEW_SHAPES
        if retune:
            gemm_tuning._TUNE_NEW_SHAPES = False       # (R_MAPPO.train's `with gemm_tuning.tuning()` follows this switch)
            torch.cuda.tunable.tuning_enable(False)
        try:
            k6 = max(1, min(opt.steps, 5))
            step()
            fence()
            t6 = time.perf_counter()
            for _ in range(k6):
                step()
            fence()
            e6 = time.perf_counter() - t6
        finally:
            policy.set_matrix_arithmetic(opt.matrix_arithmetic)
            if retune:
                gemm_tuning._TUNE_NEW_SHAPES = True
        if world > 1:
            t = torch.tensor([e6], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e6 = float(t.item())
        other = {"steps": k6, "ms_per_step": round(1e3 * e6 / k6, 3),
                 "value": round(wl["T"] * wl["N"] * k6 / e6, 1), "unit": "env-steps/s",
                 "note": "same run, measured after the timed region with policy.set_matrix_arithmetic(%r)" % other_name}
    # peak HBM held by the caching allocator on every rank (buffer + standardised copies + saved activations of an update)
    peak_mem = [int(torch.cuda.max_memory_allocated(dev))]
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64, device=dev)
        t[rank] = peak_mem[0]
        dist.all_reduce(t)
        peak_mem = [int(v) for v in t.tolist()]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / opt.steps
        value = wl["T"] * wl["N"] * opt.steps / elapsed

        here = csrc_digest()

        def pmc_traffic(name, nbytes):
            """(HBM bytes per launch, provenance) from the committed rocprofv3 PMC passes (profiles/r0N_pmc_summary.json, newest
            first: 2 x FETCH_SIZE + WRITE_SIZE, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950) -- only quoted when
            the profiled launch had the same algorithmic byte count.  Provenance: the file, the commit and the digest of the
            kernel sources the passes ran on (tools/summarize_round.py writes them), and whether that digest is the one of
            the sources this run was built from."""
            for fn in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json",
                       "r02_pmc_summary.json", "r01_pmc_summary.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", fn)) as f:
                        doc = json.load(f)
                    rec = doc.get(name)
                    if rec and abs(rec["algorithmic_bytes"] - nbytes) <= 0.01 * nbytes:
                        meta = doc.get("_provenance", {})
                        return rec["hbm_bytes"], {
                            "traffic_source": "committed rocprofv3 PMC passes (profiles/%s), not this run" % fn,
                            "traffic_commit": meta.get("commit"), "traffic_csrc_digest": meta.get("csrc_digest"),
                            "traffic_kernels": rec.get("kernels_seen"),
                            "traffic_matches_this_build": (meta.get("csrc_digest") == here) if meta.get("csrc_digest") else None}
                except Exception:
                    pass
            return None, {"traffic_source": None}

        def roof(name):
            if name not in kt:
                return None
            launches, ms, nbytes = kt[name]
            achieved = nbytes / (ms * 1e-3) / 1e9
            traffic, prov = pmc_traffic(name, nbytes)
            return dict({"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_ms": round(ms, 5), "launches": launches, "algorithmic_bytes": int(nbytes)}, **prov)

        def roof_mfma(name, what):
            """K9 / K15 launches (the fused trunk, the 512-wide Linear layers): the launch against BOTH of its roofs, and
            `bound` / `achieved` / `peak` / `frac` are those of the roof it sits closer to.
              * matrix cores: algorithmic float32 FLOPs of the launches / their time on the launch stream, against the peak of
                the instruction the products are formed with -- the dense f32 MFMA peak under `f32_mfma`; under `six_term`
                every algorithmic FLOP is six bf16 MFMA FLOPs, so the float32-equivalent peak is the dense bf16 peak / 6
                (416.7 TFLOP/s: above the f32 MFMA peak, which is the point of the form);
              * HBM: algorithmic bytes of the launches / the same time, against the spec peak.
            `frac_of_f32_mfma_peak` keeps the figure earlier rounds quoted (float32-equivalent rate / 157.3; it may exceed 1
            under `six_term` -- K15 does)."""
            if name not in mt:
                return None
            launches, ms, flops, nbytes = mt[name]
            tf = flops / launches / (ms * 1e-3) / 1e12
            gbs = nbytes / launches / (ms * 1e-3) / 1e9
            traffic, prov = pmc_traffic(name, nbytes / launches)
            six = opt.matrix_arithmetic == "six_term"
            mfma_peak = MFMA_BF16_PEAK_TFLOPS / 6 if six else MFMA_F32_PEAK_TFLOPS
            roofs = {"mfma": {"achieved": round(tf, 1), "peak": round(mfma_peak, 1),
                              "unit": "TFLOP/s" + (" (float32-equivalent; peak = dense bf16 MFMA / 6 terms)" if six else ""),
                              "frac": round(tf / round(mfma_peak, 1), 4)},
                     "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4)}}
            bound = "hbm" if roofs["hbm"]["frac"] > roofs["mfma"]["frac"] else "mfma"
            return {"kernel": what, "bound": bound, "achieved": roofs[bound]["achieved"], "peak": roofs[bound]["peak"],
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": roofs[bound]["frac"],
                    "traffic": traffic, **prov, "roofs": roofs,
                    "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                    "launch_ms": round(ms, 4), "launches": launches, "flop_per_launch": int(flops / launches),
                    "algorithmic_bytes": int(nbytes / launches),
                    "share_of_step": round(launches * ms / (opt.steps if k9_timed_in == "the timed region" else 1) / ms_per_step, 3),
                    "timed_in": k9_timed_in}

        out = {
            # BASELINE.json's metric (quoted on the north star); other workloads name their own shape
            "metric": "env-steps/sec through GAE+ppo_update, %d threads×%d agents×%d steps" % (wl["N"], wl["A"], wl["T"]),
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": opt.steps,
            "warmup": opt.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # how the float32 matrix products of K9 / K12 were formed in the timed region (inputs, outputs, accumulation and
            # every other operation are float32 either way)
            "arithmetic": ARITHMETIC_TEXT[opt.matrix_arithmetic] if (args.hidden_size == 64 or (
                args.hidden_size == 512 and opt.matrix_arithmetic == "six_term" and "mappo_linear512_forward" in mt)) else
                          "f32 (library float32 GEMMs + K6 / K7)",
            "hbm_peak_bytes_per_rank": peak_mem,
            "csrc_digest": here,        # of the kernel sources this run was built from (cf. roofline.traffic_csrc_digest)
            # updates of the timed region that were replays of a captured HIP graph (0: every update ran eagerly)
            "update_graph_replays_per_step": graph_replays / max(1, opt.steps),
            # captures that failed (and were finished / re-run eagerly, update_graph.py) since the trainer was built: 0 expected
            "update_graph_capture_failures": 0 if ug is None else ug.capture_failures,
            # whether the timed steps ran actor and critic on two streams (evaluations of <= 2^20 rows; rooflines then come from
            # one extra one-stream step: `timed_in`)
            "two_streams": two_streams,
            "config": {"workload": wl["label"], "T": wl["T"], "n_rollout_threads": wl["N"],
                       "threads_per_gpu": n_local, "agents": wl["A"], "obs_dim": wl["Do"],
                       "share_obs_dim": wl["Ds"], "actions": wl["na"], "ppo_epoch": args.ppo_epoch,
                       "num_mini_batch": args.num_mini_batch, "gemm_tuning": bool(tuned), "sampler_rng": opt.sampler_rng,
                       "parallelism": "dp%d over rollout threads" % world},
            # RCCL: ranks in the job and the gradient all-reduce (one flat actor+critic bucket per update) on rank 0
            "rccl_ranks": world if dist.is_initialized() and dist.get_backend() == "nccl" else 0,
            "grad_allreduce": {"per_step": n_coll // max(1, opt.steps), "bucket_bytes": coll_bytes,
                               "ms_per_step": round(coll_ms / max(1, opt.steps), 4)},
            # the scalar prologue (loss denominators + ValueNorm moments, 32 bytes): once per train() when the whole-batch
            # tuple is reused, otherwise once per update and issued one update ahead (DataParallel.begin_scales)
            "scalar_allreduce": {"per_step": scalar_n / max(1, opt.steps),
                                 "updates_served_from_cache_per_step": reused_n / max(1, opt.steps)},
            # the dominant kernel of the step: the fused trunk's forward launch (mlp_fwd4_kernel / mlp_fwd3_kernel; actor and
            # critic launches averaged, as rocprofv3 --stats averages them), against the nearer of its two roofs
            # (hidden 512: K15's forward launches -- first layer and hidden layers averaged, as rocprofv3 --stats averages them)
            "roofline": roof_mfma("mappo_mlp_forward", "mlp_fwd4_kernel / mlp_fwd3_kernel / mlp_fwd_kernel (mappo_mlp_forward)")
            or roof_mfma("mappo_linear512_forward", "lin_fwd_kernel (mappo_linear512_forward: K15, forward and input gradient)")
            or roof("mappo_gae_f32"),
            "roofline_linear512_wgrad": roof_mfma("mappo_linear512_wgrad", "lin_wgrad_kernel + lin_reduce_kernel (mappo_linear512_wgrad: K15)"),
            "roofline_mlp_backward": roof_mfma("mappo_mlp_backward",
                                               "mlp_bwd_kernel + mlp_dw1_{direct,rows}_kernel + reduce / finish (mappo_mlp_backward)"),
            # the kernel BASELINE.json's north star names (>= 70 % of HBM in the GAE scan), HBM bound
            "roofline_gae": roof("mappo_gae_f32"),
            # the sampler's traffic: the fused gather -- or, for one feed-forward minibatch per epoch on the trainer's route, only
            # the normalised advantages (the other fields are views of the buffer: SharedReplayBuffer._whole_batch_views)
            "roofline_gather": roof("mappo_gather_chunks" if wl["recurrent"] else "mappo_gather_rows") or roof("mappo_adv_normalize"),
            "train_info": {k: round(float(v), 6) for k, v in info.items()},
        }
        if other is not None:
            out[other_name] = other
        g = out["roofline_gae"]
        if g is not None:
            # `frac` / `launch_ms` are the in-situ figures (first launch of a step, right behind the previous step's update);
            # back to back the same launch finds part of its lines in the Infinity Cache and warm TLBs
            nbytes = g["algorithmic_bytes"]
            g["in_situ"] = {"launch_ms": g["launch_ms"], "frac": g["frac"]}
            # The same in-situ launches by the KERNEL'S OWN begin / end timestamps (HIP events attached to the dispatch:
            # hipExtLaunchKernelGGL through mappo_gae_time_next_launch) -- the duration rocprofv3 --kernel-trace reports.  The
            # event pair recorded around the launch also times two packets of the command processor (5-6 us on this ~50 us
            # kernel: profiles/r06_gae_in_situ_timing.json sets both against the trace of one run).  The profiled steps alternate
            # between the two methods (SharedReplayBuffer.compute_returns: both on one launch would lengthen what the pair
            # sees).  `frac` / `achieved` / `launch_ms` of this object are the dispatch-level figures when the hook took
            # launches; the figures of the event pair -- what every earlier round quoted -- stay under `event_pair_around_launch`.
            if "mappo_gae_f32/dispatch" in kt:
                dl, dms, _ = kt["mappo_gae_f32/dispatch"]
                g["event_pair_around_launch"] = {"launch_ms": g["launch_ms"], "achieved": g["achieved"], "frac": g["frac"],
                                                 "launches": g["launches"]}
                g["launch_ms"], g["launches"] = round(dms, 5), dl
                g["achieved"] = round(nbytes / (dms * 1e-3) / 1e9, 1)
                g["frac"] = round(nbytes / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                g["timing"] = "kernel begin / end timestamps (HIP events attached to the dispatch), in situ"
                g["in_situ"] = {"launch_ms": g["launch_ms"], "frac": g["frac"], "event_pair_around_launch": g["event_pair_around_launch"]["frac"]}
            else:
                g["timing"] = "HIP event pair recorded around the launch, in situ"
            # which kernel ran (mappo_gae_last_variant; >= 70: the time-parallel scan, only with --gae-scan) and its contract
            g["variant"] = int(gae_variant)
            g["bit_exact"] = bool(buf_gae_exact)
            g["back_to_back"] = {"launch_ms": round(gae_b2b_ms, 5), "launches": reps,
                                 "achieved": round(nbytes / (gae_b2b_ms * 1e-3) / 1e9, 1),
                                 "frac": round(nbytes / (gae_b2b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and not opt.no_cpu_baseline:
            cb = out["cpu_baseline"] = cpu_baseline(wl)
            ref = reference_recorded(opt.workload)
            factor = None
            if ref is not None and ref.get("port_vs_reference_same_machine"):
                f = [r["port_over_reference"] for r in ref["port_vs_reference_same_machine"]]
                factor = sum(f) / len(f)
            cb["sample"] += "; GPU / CPU-port ratio on env-steps/s = %.0fx" % (value / cb["value"])
            if factor:
                # the port is not the reference: on one machine it runs at `factor` x the reference's speed
                # (ref["source"]), so the reference on THIS host would do about value / factor
                cb["reference_equivalent"] = {
                    "value": round(cb["value"] / factor, 1), "unit": "env-steps/s", "port_over_reference": round(factor, 3),
                    "gpu_over_reference_equivalent": round(value / (cb["value"] / factor), 1),
                    "source": ref["source"],
                    "note": "port rate / (port / reference measured back to back on one machine); the figure to hold "
                            "against north_star's >= 10x, not the port's"}
                cb["sample"] += " (port), %.0fx against the reference-equivalent rate %.0f env-steps/s (port / %.2f)" % (
                    value / (cb["value"] / factor), cb["value"] / factor, factor)
            if ref is not None:
                cb["reference_recorded"] = ref
    else:
        out = None
    if out is not None and world == 1 and opt.workload == "ns" and not opt.threads and not opt.no_workloads:
        # the other BASELINE configs, by this same command line (outside `value`): hand the GPU over first
        del step, buf, trainer, policy, next_value, ug, info, kt, mt
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["workloads"] = other_workloads(opt, float(os.environ.get("MAPPO_BENCH_WORKLOADS_BUDGET_S", "240")))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL writes a version banner to the C stdio stream; push that out first so that the JSON
        # line is the LAST line this process prints
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
