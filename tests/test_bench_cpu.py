"""bench.py's host-side pieces that need no GPU: the `workloads` leg's compaction of a child's JSON line, the table of child
command lines, the provenance digest of the kernel sources, and the CPU-reference record the line quotes."""
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_other_workloads_cover_every_baseline_config():
    b = _bench()
    names = [w[0] for w in b.OTHER_WORKLOADS]
    assert names == ["cfg2", "cfg3", "ns_rnn", "smac", "smac_shard64", "hanabi"]
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5            # configs[0] is the CPU-runnable plumbing case; [1..4] = cfg2, cfg3, smac, hanabi
    for name, argv, (steps, warmup), what in b.OTHER_WORKLOADS:
        assert argv[0] == "--workload" and argv[1] in b.WORKLOADS and steps >= 3 and warmup >= 1 and what
        # every child runs long enough to be timed: >= ~0.4 s of steps at the sizes measured in round 6
    assert dict((w[0], w[2]) for w in b.OTHER_WORKLOADS)["smac_shard64"][0] >= 30


def test_compact_line_keeps_what_a_reader_compares():
    b = _bench()
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_lines.json")))
    for name in ("cfg2", "cfg3", "ns_rnn", "smac", "smac_shard64", "hanabi"):
        c = b.compact_line(committed[name], "what")
        assert c["value"] == committed[name]["value"] and c["ms_per_step"] == committed[name]["ms_per_step"]
        assert c["dtype"] == "f32" and 0 < c["roofline"]["frac"] < 1 and c["roofline"]["bound"] in ("hbm", "mfma")
        assert 0 < c["roofline_gae"]["frac"] < 1 and c["roofline"]["timed_in"]
        assert (c["cpu_baseline"] is None) == (committed[name].get("cpu_baseline") is None)
    # a child without a GAE / roofline object (e.g. a failed profile) does not break the parent's line
    bare = {"config": {"workload": "w", "n_rollout_threads": 1}, "value": 1.0, "unit": "env-steps/s", "ms_per_step": 1.0,
            "steps": 1, "warmup": 0, "dtype": "f32", "arithmetic": "f32 (library)"}
    c = b.compact_line(bare, "what")
    assert c["roofline"] is None and c["roofline_gae"] is None and c["cpu_baseline"] is None


def test_committed_line_carries_its_workloads_and_provenance():
    ns = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_lines.json")))["ns"]
    assert set(ns["workloads"]) == {"cfg2", "cfg3", "ns_rnn", "smac", "smac_shard64", "hanabi"}
    alone = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_lines.json")))
    for name, e in ns["workloads"].items():     # in-line and own-command-line figures of one call agree (+- 3 %)
        assert abs(e["ms_per_step"] / alone[name]["ms_per_step"] - 1) < 0.03, name
    r = ns["roofline"]
    assert r["timed_in"] == "the timed region" and ns["two_streams"] is False
    assert r["traffic_commit"] and r["traffic_csrc_digest"] and r["traffic_matches_this_build"] is True
    b = _bench()
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_summary.json")))
    # the committed PMC passes ran on the kernel sources of this tree
    assert pmc["_provenance"]["csrc_digest"] == b.csrc_digest() == ns["csrc_digest"]
    rec = b.reference_recorded("ns")
    assert rec["source"].startswith("profiles/r06_cpu_port_vs_reference.json") and len(rec["port_vs_reference_same_machine"]) == 2
