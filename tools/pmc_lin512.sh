#!/bin/bash
# GPU box: K15 alone (tools/bench_lin512.py) -- event timings against the library GEMMs, then SQ / GRBM counter passes over the
# same command (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/lin512
mkdir -p $OUT
cd $REPO
timeout 300 python tools/bench_lin512.py > $OUT/bench_lin512.json 2> $OUT/bench_lin512.err
cat $OUT/bench_lin512.json
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_a -o lin -- python $REPO/tools/bench_lin512.py --reps 2 --no-library > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -f csv -d $OUT/pmc_b -o lin -- python $REPO/tools/bench_lin512.py --reps 2 --no-library > $OUT/pmc_b.log 2>&1
find $OUT -name "*.db" -delete
python - <<'PY'
import collections, csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/lin512/"
for tag in ("pmc_a", "pmc_b"):
    tr = glob.glob(out + tag + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(out + tag + "/**/*counter_collection.csv", recursive=True)
    if not tr or not cc:
        print(tag, "no output", open(out + tag + ".log").read()[-400:]); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0]))}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        if "lin::lin_" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0][-40:]
            acc[name][r["Counter_Name"]].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0)))
    for name, cs in acc.items():
        print(tag, name, {k: ("%.4g" % (sum(v for v, _ in vs) / len(vs)), "%.3f ms" % (sum(d for _, d in vs) / len(vs) / 1e6)) for k, vs in cs.items()})
PY
