#!/bin/bash
# GPU box: SQ / GRBM counters (one --pmc pass with --kernel-trace only each) of the K9 forward in its three structures at the
# critic's shape -- is a fuller matrix pipe paid for with clock (power) or not reached at all?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/pmc_mlp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
for f in 16 24 4; do
  MAPPO_MLP_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/f$f -o v -- python $REPO/tools/bench_mlp.py --sequential --reps 5 --din 384 > $OUT/f$f.log 2>&1
  find $OUT/f$f -name "*.db" -delete
done
cd $REPO
python - <<'PY'
import csv, collections, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/pmc_mlp/"
for f in (16, 24, 4):
    d = out + "f%d/" % f
    tr = glob.glob(d + "*kernel_trace.csv"); cc = glob.glob(d + "*counter_collection.csv")
    if not tr or not cc:
        print("flags", f, "no output", open(out + "f%d.log" % f).read()[-300:]); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0]))}
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(cc[0])):
        if "mlp_fwd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in acc.items():
        n = len(disp)
        mean = lambda name: sum(x.get(name, 0.0) for x in disp.values()) / n
        ns = sum(dur[i] for i in disp if i in dur) / n
        cyc = mean("GRBM_GUI_ACTIVE") / 8
        mf = mean("SQ_VALU_MFMA_BUSY_CYCLES")
        print("flags", f, k, "n", n, "dur %.3f ms" % (ns / 1e6), "clock %.3f GHz" % (cyc / ns), "mfma busy %.3f" % (mf / (1024 * cyc)),
              "of nominal %.3f" % (mf / (1024 * 2.4 * ns)), "wait/wave %.3f" % (mean("SQ_WAIT_INST_ANY") / max(1.0, mean("SQ_WAVE_CYCLES"))),
              "valu insts %.3g" % mean("SQ_INSTS_VALU"))
PY
