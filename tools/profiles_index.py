#!/usr/bin/env python
"""Writes profiles/INDEX.md: one line per evidence file (round, kind, what it holds), from the files' own "what" fields and
the tables of profiles/README.md (the long form).  Run after adding a file:  python tools/profiles_index.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
EXTRA = {
    "isa_summary.json": "CDNA4 instructions per kernel source (tools/isa_summary.py --write; tests/test_kernel_resources_cpu.py)",
    "kernel_resources.json": "registers / LDS / scratch / occupancy of every kernel instance from the compiler "
                             "(tools/kernel_resources.py --write; tests/test_kernel_resources_cpu.py)",
    "README.md": "the long form: per round, how each file was produced and what it showed",
    "INDEX.md": "this table",
}


def first_sentence(text, n=230):
    text = " ".join(str(text).split())
    return text if len(text) <= n else text[:n].rsplit(" ", 1)[0] + " ..."


def main():
    readme = open(os.path.join(PROF, "README.md")).read()
    table = {}
    for m in re.finditer(r"^\|\s*`([^`|]+)`[^|]*\|\s*(.+?)\s*\|\s*$", readme, re.M):
        table.setdefault(m.group(1), m.group(2))
    rows = []
    for name in sorted(os.listdir(PROF)):
        path = os.path.join(PROF, name)
        if os.path.isdir(path):
            continue
        what = EXTRA.get(name)
        if what is None and name.endswith(".json"):
            try:
                doc = json.load(open(path))
                for k in ("what", "note", "description"):
                    if isinstance(doc, dict) and isinstance(doc.get(k), str):
                        what = doc[k]
                        break
            except Exception:
                pass
        if what is None:
            what = table.get(name)
        if what is None and name.endswith("kernel_stats.csv") or (what is None and "kernel_stats" in name):
            what = "rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average ns) of the bench workload in the name"
        rnd = re.match(r"r(\d\d)_", name)
        kind = "kernel stats" if "kernel_stats" in name else "PMC / SQ counters" if "pmc" in name else \
            "A / B record" if "_ab_" in name else "bench lines" if "bench" in name else \
            "probe" if "probe" in name else "shard proxy" if "shard" in name else "snapshot" if rnd is None else "record"
        rows.append((name, rnd.group(1).lstrip("0") if rnd else "-", kind, first_sentence(what or "(see README.md)")))
    with open(os.path.join(PROF, "INDEX.md"), "w") as f:
        f.write("# profiles/ -- index\n\nEvidence copied from `gpurun_out/` (scratch) or produced by `tools/`; one line per file, "
                "newest round last within a kind.  Long form: `README.md`.  Regenerate: `python tools/profiles_index.py`.\n\n")
        f.write("| file | round | kind | what |\n|---|---|---|---|\n")
        for name, rnd, kind, what in sorted(rows, key=lambda r: (r[2], r[1].zfill(2), r[0])):
            f.write("| `%s` | %s | %s | %s |\n" % (name, rnd, kind, what.replace("|", "/")))
    print("wrote profiles/INDEX.md (%d files)" % len(rows))


if __name__ == "__main__":
    main()
