#!/usr/bin/env python
"""Turns gpurun_out/<round>/ (tools/profile_r02.sh / tools/profile_r03.sh on the MI355X box) into the committed evidence
under profiles/ (python tools/summarize_round.py r03): kernel-statistics CSVs, the bench lines of every workload and
profiles/<round>_pmc_summary.json -- HBM bytes per launch of
the dominant kernels from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports half the bytes of wide coalesced reads: doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))


def counters(name):
    f = glob.glob(os.path.join(SRC, "pmc_" + name, "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(1024.0 * float(r["Counter_Value"]))      # the counters are in KB
    return acc


def mean_of(acc, key):
    vals = [v for k, vs in acc.items() if key in k for v in vs]
    return (sum(vals) / len(vals) if vals else 0.0), len(vals)


def seen(acc, *keys):
    """Kernel names (template arguments kept, argument lists cut) of the dispatches an entry was computed from."""
    return sorted({k.split("(")[0].strip() for k in acc if any(q in k for q in keys)})


def provenance():
    """What the passes ran on: the commit the call was made from (MAPPO_COMMIT, baked into the gpurun command line --
    the GPU box has no .git; gpurun_out/<round>/commit.txt) and the digest of the kernel sources AS THEY WERE ON THE BOX
    (csrc_digest.txt, bench.csrc_digest()).  bench.py prints both next to `traffic` and compares the digest with its own."""
    out = {}
    for key, fn in (("commit", "commit.txt"), ("csrc_digest", "csrc_digest.txt")):
        try:
            out[key] = open(os.path.join(SRC, fn)).read().strip() or None
        except OSError:
            out[key] = None
    return out


def main():
    from onpolicy.algorithms.utils.fused_mlp import _work
    fetch, write = counters("FETCH_SIZE"), counters("WRITE_SIZE")
    rows = 400 * 4096 * 8
    out = {}
    # forward launch: actor (48 -> 64 -> 64 -> 5) and critic (384 -> 64 -> 64 -> 1) launches averaged, like bench.py and
    # rocprofv3 --stats average them
    alg_f = (_work(rows, 48, 2, 5, False)[1] + _work(rows, 384, 2, 1, False)[1]) / 2
    f, n = mean_of(fetch, "mlp_fwd")          # mlp_fwd_kernel (actor) and, from round 4, mlp_fwd3_kernel (critic)
    w, _ = mean_of(write, "mlp_fwd")
    out["mappo_mlp_forward"] = {
        "algorithmic_bytes": alg_f, "fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes": 2 * f + w,
        "dispatches_averaged": n,
        "kernels_seen": seen(fetch, "mlp_fwd"),
        "note": "north-star bench.py step (separate --pmc passes with --kernel-trace only, tools/profile_r02.sh); actor "
                "and critic launches averaged; FETCH_SIZE doubled (gfx950 counts 64 B per 128 B request of a wide "
                "coalesced read, MI355X_MICROARCH.md)"}
    alg_b = (_work(rows, 48, 2, 5, True, True)[1] + _work(rows, 384, 2, 1, True)[1]) / 2
    # the small kernels of a backward call: two reductions (round 2), + the finish kernel (early round 3), one tail kernel now
    small = ("mlp_tail_kernel",) if any("mlp_tail" in k for k in fetch) else \
        ("mlp_reduce_kernel", "mlp_reduce_kernel") + (("mlp_finish_kernel",) if any("mlp_finish" in k for k in fetch) else ())
    # per mappo_mlp_backward call (= per tail launch): every launch of the call's kernels summed, divided by the number of
    # calls -- from round 5 the actor's call has no separate first-layer kernel (its dW1 is accumulated in the chain's launch),
    # so a sum of per-kernel means would no longer be a call
    def per_call(acc):
        keys = ("mlp_bwd_kernel", "mlp_dw1_") + tuple(set(small))
        total = sum(v for k, vs in acc.items() if any(q in k for q in keys) for v in vs)
        calls = sum(len(vs) for k, vs in acc.items() if small[0] in k) / (2 if small[0] == "mlp_reduce_kernel" else 1)
        return total / max(1, calls), int(calls)
    fb, n_calls = per_call(fetch)
    wb, _ = per_call(write)
    out["mappo_mlp_backward"] = {
        "algorithmic_bytes": alg_b, "fetch_size_bytes_raw": fb, "write_size_bytes": wb, "hbm_bytes": 2 * fb + wb,
        "calls_averaged": n_calls,
        "kernels_seen": seen(fetch, "mlp_bwd_kernel", "mlp_dw1_", *set(small)),
        "note": "one mappo_mlp_backward call = chain kernel (+ first-layer weight-gradient kernel for inputs wider than 64) + "
                "the tail kernel; actor and critic calls averaged, six-term arithmetic only (--no-f32-mfma)"}
    f, n = mean_of(fetch, "gae_")
    w, _ = mean_of(write, "gae_")
    out["mappo_gae_f32"] = {"algorithmic_bytes": 24 * rows, "fetch_size_bytes_raw": f, "write_size_bytes": w,
                            "hbm_bytes": 2 * f + w, "dispatches_averaged": n, "kernels_seen": seen(fetch, "gae_"),
                            "note": "north-star size, fused advantages epilogue (24 B / element)"}
    # the record gather of the north-star step (once per train(): the whole-batch tuple): 64-byte records read through the
    # index list + the gathered columns written; algorithmic bytes as SharedReplayBuffer._gather counts them
    f, n = mean_of(fetch, "gather_records_kernel")
    w, _ = mean_of(write, "gather_records_kernel")
    if n:
        rec_widths = 1 + 1 + 1 + 1 + 1 + 1 + 1 + 5      # actions, value_preds, returns, masks, active_masks, logp, adv, avail
        out["mappo_gather_rows"] = {"algorithmic_bytes": 2 * 4 * rec_widths * rows + 8 * rows, "fetch_size_bytes_raw": f,
                                    "write_size_bytes": w, "hbm_bytes": 2 * f + w, "dispatches_averaged": n,
                                    "kernels_seen": seen(fetch, "gather_records_kernel"),
                                    "note": "until round 4 the packed records were 64 bytes (16 floats, 12 of them payload: "
                                            "fetched bytes exceeded the algorithmic count by the padding); round 5 packs "
                                            "dense 48-byte records for the device sampler's ascending walks"}
    prov = provenance()
    with open(os.path.join(DST, TAG + "_pmc_summary.json"), "w") as fh:
        json.dump(dict(out, _provenance=prov), fh, indent=1)
    print("provenance:", prov)
    for k, v in out.items():
        print(k, "algorithmic %.3f GB, HBM %.3f GB (%.2fx)" % (v["algorithmic_bytes"] / 1e9, v["hbm_bytes"] / 1e9,
                                                              v["hbm_bytes"] / v["algorithmic_bytes"]))
    for w in ("ns", "cfg2", "cfg3", "smac", "ns_rnn", "hanabi"):
        for f in glob.glob(os.path.join(SRC, "prof_" + w, "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(DST, "%s_bench_%s_kernel_stats.csv" % (TAG, w)))
    lines = {}
    for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
        name = os.path.basename(f)[6:-5]
        try:
            lines[name] = json.loads(open(f).read())
        except ValueError:
            print("skipped (no JSON line):", f)
    with open(os.path.join(DST, TAG + "_bench_lines.json"), "w") as fh:
        json.dump(lines, fh, indent=1)
    print("bench lines:", {k: (round(v["value"]), v["ms_per_step"]) for k, v in lines.items()})


if __name__ == "__main__":
    main()
