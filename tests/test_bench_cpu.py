"""bench.py pieces that can run without a GPU: the workload table and the `cpu_baseline` leg (oracle buffer + the
product's trainer on host cores) on a shrunken copy of every workload; the timed HIP path itself is covered by the
-m gpu bench test."""
import importlib.util
import os

import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_workload_table(bench):
    assert set(bench.WORKLOADS) == {"ns", "ns_rnn", "cfg2", "cfg3", "smac", "hanabi"}
    c3 = bench.WORKLOADS["cfg3"]            # BASELINE.json configs[2]: simple_spread itself at N=4096, T=400
    assert (c3["T"], c3["N"], c3["A"], c3["Do"], c3["Ds"]) == (400, 4096, 3, 18, 54)
    ns = bench.WORKLOADS["ns"]
    assert (ns["T"], ns["N"], ns["A"], ns["Do"], ns["Ds"], ns["na"]) == (400, 4096, 8, 48, 384, 5)     # BASELINE north star
    args = bench.make_args(ns, 512)
    assert args.n_rollout_threads == 512 and args.ppo_epoch == 10 and args.use_ReLU is False             # --use_ReLU => Tanh
    assert bench.make_args(bench.WORKLOADS["smac"], 8).use_recurrent_policy is True


@pytest.mark.parametrize("name", ["ns", "ns_rnn", "smac"])
def test_cpu_baseline_leg_runs(bench, name):
    threads = torch.get_num_threads()
    wl = dict(bench.WORKLOADS[name], T=20, cpu_sample_N=2)
    wl["flags"] = [f if f != "10" or wl["flags"][i - 1] != "--ppo_epoch" else "2" for i, f in enumerate(wl["flags"])]
    try:
        out = bench.cpu_baseline(wl)
    finally:
        torch.set_num_threads(threads)
    assert out["kind"] == "port" and out["unit"] == "env-steps/s" and out["value"] > 0 and out["cores"] >= 1
    assert "n_rollout_threads=2" in out["sample"]


def test_plain_gpus_flag_launches_one_rank_per_gpu(bench, monkeypatch):
    """`python bench.py --gpus N` without a launcher around it starts N ranks itself through torch.distributed.run
    (loop-back rendezvous) and hands the original flags on; under a launcher the world size must match --gpus."""
    import subprocess
    import sys
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--threads", "64"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--threads", "64"] and cmd[-7].endswith("bench.py")
    # a launcher that started a different number of ranks than --gpus asks for is an error, not a silent N = 1 run
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(AssertionError, match="--gpus 4"):
        bench.main()
