#!/usr/bin/env python
"""Which CDNA4 instructions the kernels actually compile to: per source, counts of the instruction families that
the design relies on -- LDS-DMA loads (global_load_lds_dwordx4, gfx950's 128-bit direct-to-LDS load, the GAE ring),
f32 MFMA (the GRU step's hidden projection), 128-bit global loads / stores, non-temporal accesses, LDS reads, DPP /
cross-lane ops.  From ``hipcc -S --cuda-device-only`` (cross-compiles, no GPU needed).

    python tools/isa_summary.py [--write]      # --write refreshes profiles/isa_summary.json
"""
import argparse
import json
import os
import re
import subprocess
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "on-policy_amd", "csrc")
SNAPSHOT = os.path.join(ROOT, "profiles", "isa_summary.json")
SOURCES = ("mappo_gae.hip", "mappo_copy.hip", "mappo_norm.hip", "mappo_loss.hip", "mappo_rnn.hip", "mappo_mlp.hip",
           "mappo_perm.hip", "mappo_env.hip", "mappo_optim.hip")
FAMILIES = {
    "lds_dma_128bit": r"^\s*global_load_lds_dwordx4\b",
    "lds_dma_other": r"^\s*(global_load_lds_(?!dwordx4)\w+|buffer_load_\w+ .*\blds\b)",
    "mfma_f32": r"^\s*v_mfma_f32_\w+",
    "global_load_128bit": r"^\s*global_load_dwordx4\b",
    "global_store_128bit": r"^\s*global_store_dwordx4\b",
    "non_temporal": r"^\s*global_(load|store)_\w+ .*\bnt\b",
    "lds_read": r"^\s*ds_read\w*",
    "lds_write": r"^\s*ds_write\w*",
    "dpp_or_permute": r"^\s*(v_\w+_dpp\b|ds_bpermute_b32|ds_swizzle_b32|v_permlane\w+|v_readlane_b32)|\b(row_shr|row_bcast|quad_perm)\b",
    "global_atomic": r"^\s*global_atomic_\w+",
    "scratch_access": r"^\s*scratch_(load|store)\w*",
}


def assembly(source):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(CSRC, source), "-o", "-"]
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def summarise(source):
    text = assembly(source)
    counts = Counter()
    for line in text.splitlines():
        for name, pattern in FAMILIES.items():
            if re.search(pattern, line):
                counts[name] += 1
    mfma = Counter(re.findall(r"^\s*(v_mfma_\w+)", text, flags=re.M))
    out = {name: counts.get(name, 0) for name in FAMILIES}
    out["kernels"] = len(re.findall(r"^\s*\.amdhsa_kernel\b", text, flags=re.M))
    out["mfma_kinds"] = dict(mfma)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    opt = ap.parse_args()
    table = {src: summarise(src) for src in SOURCES}
    for src, row in table.items():
        print(src, {k: v for k, v in row.items() if v})
    if opt.write:
        with open(SNAPSHOT, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
        print("wrote", SNAPSHOT)
