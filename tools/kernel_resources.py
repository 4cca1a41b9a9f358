#!/usr/bin/env python
"""Register / LDS / scratch budget of every gfx950 kernel, from the compiler's own resource analysis
(``hipcc -Rpass-analysis=kernel-resource-usage``; cross-compiles, no GPU needed).

    python tools/kernel_resources.py                 # print the table
    python tools/kernel_resources.py --write         # refresh profiles/kernel_resources.json
"""
import argparse
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "on-policy_amd", "csrc")
SNAPSHOT = os.path.join(ROOT, "profiles", "kernel_resources.json")
SOURCES = ("mappo_gae.hip", "mappo_copy.hip", "mappo_norm.hip", "mappo_loss.hip", "mappo_rnn.hip", "mappo_mlp.hip",
           "mappo_perm.hip", "mappo_env.hip", "mappo_optim.hip")
FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes",
          "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_bytes"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True, check=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", n) for n in out]


def analyse(source):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(CSRC, source), "-o", os.devnull]
    text = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in body:
            key, value = body.rsplit(":", 1)
            if key.strip() in FIELDS:
                cur[FIELDS[key.strip()]] = int(value)
    for k, name in zip(kernels, demangle([k.pop("mangled") for k in kernels])):
        k["kernel"] = name
    return kernels


def collect():
    table = {}
    for src in SOURCES:
        for k in analyse(src):
            name = k.pop("kernel")
            table["%s :: %s" % (src, name)] = k
    return table


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    opt = ap.parse_args()
    table = collect()
    for name, k in table.items():
        print("%-110s vgpr %3d agpr %3d scratch %3d occ %d lds %6d" % (name[:110], k["vgprs"], k.get("agprs", 0),
                                                                      k["scratch_bytes"], k["occupancy"], k["lds_bytes"]))
    if opt.write:
        with open(SNAPSHOT, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
        print("wrote", SNAPSHOT, len(table), "kernels")
