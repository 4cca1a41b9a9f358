"""-m gpu: bench.py's one-line JSON contract at a small size, and the data-parallel code path on RCCL
with a single rank (MAPPO_FORCE_DIST=1: flat gradient bucket, statistic all-reduces, barrier) --
the 8-GPU run itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra_env=None, args=()):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--threads", "64", "--steps", "1",
                          "--warmup", "1", "--no-cpu-baseline", "--sampler-rng", "host"] + list(args),
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_contract_and_rccl_single_rank():
    plain = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in plain, key
    assert plain["unit"] == "env-steps/s" and plain["n_gpus"] == 1 and plain["data"] == "synthetic"
    assert plain["config"]["workload"] and "model" not in plain["config"]
    r = plain["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    forced = _run({"MAPPO_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577", "RANK": "0",
                   "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    # same seeds, same host permutations: the RCCL path must reproduce the plain update
    for k, v in plain["train_info"].items():
        assert forced["train_info"][k] == pytest.approx(v, rel=2e-3, abs=1e-5), k
