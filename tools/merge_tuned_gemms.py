#!/usr/bin/env python
"""Merges TunableOp result files (gpurun_out/gemm_tuning/tunableop_results*.csv from tools/tune_gemms.sh) into the shipped
on-policy_amd/onpolicy/tuned_gemms_gfx950.csv: validator lines must agree, (op, shape) entries of the new files win."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = os.path.join(ROOT, "on-policy_amd", "onpolicy", "tuned_gemms_gfx950.csv")


def read(path):
    validators, entries = [], {}
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        if line.startswith("Validator,"):
            validators.append(line)
        else:
            op, shape = line.split(",")[:2]
            entries[(op, shape)] = line
    return validators, entries


def main():
    new_files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "gemm_tuning", "tunableop_results*.csv")))
    validators, entries = read(SHIPPED)
    before = len(entries)
    for f in new_files:
        v, e = read(f)
        if v != validators:
            raise SystemExit("%s was made by other library versions:\n%s\n%s" % (f, v, validators))
        entries.update(e)
    with open(SHIPPED, "w") as fh:
        fh.write("\n".join(validators + [entries[k] for k in sorted(entries)]) + "\n")
    print("%d -> %d entries" % (before, len(entries)))


if __name__ == "__main__":
    main()
