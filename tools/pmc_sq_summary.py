#!/usr/bin/env python
"""Summary of tools/pmc_sq_pass.sh (one rocprofv3 --pmc pass of SQ / GRBM counters over the north-star bench step) ->
profiles/r03_pmc_sq_summary.json: per K9 kernel the shader clock the chip ran at (GRBM_GUI_ACTIVE, summed over the 8 XCDs,
against the dispatch's duration) and the share of that time the matrix pipe of an average SIMD was busy
(SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16, summed over the 1024 SIMDs).
(The keys are name fragments: a generic fragment such as "gru_seq_fwd_kernel" also collects the instances a more specific key names.)"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = sys.argv[1] if len(sys.argv) > 1 else "ns"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", TAG, "pmc_sq_" + W)
if W == "ns" and not os.path.isdir(SRC):
    SRC = os.path.join(ROOT, "gpurun_out", TAG, "pmc_sq")
KERNELS = {"mlp_fwd4_kernel": "K9 forward, version 4 (six-term, round 5 default): the critic's 384 wide input, both layers on the bf16 pipe",
           "mlp_fwd3_kernel<2, 1, 2, true>": "K9 forward, version 3 with the hidden layer in six-term form (round 5 default): the actor's 48 wide input",
           "mlp_dw1_direct_kernel<3, 4, true>": "K9 first-layer weight gradient, critic, six-term, one workgroup per CU (tuning bit 32)",
           "mlp_dw1_direct_kernel<3, 2, true>": "K9 first-layer weight gradient, critic, six-term, two workgroups per CU (round 5 default)",
           "mlp_dw1_direct_kernel<3, 4>": "K9 first-layer weight gradient, critic, float32 MFMA",
           "mlp_bwd_kernel<2, 1, 0, true, true>": "K9 backward chain + first-layer weight gradient in one launch, action head, six-term (round 5 default)",
           "mlp_bwd_kernel<2, 1, 0, true, false>": "K9 backward chain, action head, six-term, separate first-layer kernel (tuning bit 256)",
           "mlp_bwd_kernel<2, 1, 1, true": "K9 backward chain, value head, six-term (round 5 default)",
           "mlp_bwd_kernel<2, 1, 0, false": "K9 backward chain, action head, float32 MFMA",
           "mlp_bwd_kernel<2, 1, 1, false": "K9 backward chain, value head, float32 MFMA",
           "gru_seq_fwd_kernel<true>": "K12 forward, six-term (round 5 default)",
           "gru_seq_bwd_kernel<3, 5, true, true>": "K12 backward, actor, all six blocks as planes (round 5 default)",
           "gru_seq_bwd_kernel<1, 1, true, true>": "K12 backward, critic, all six blocks as planes (round 5 default)",
           "gru_seq_fwd_kernel": "K12 forward", "gru_seq_bwd_kernel<3, 5>": "K12 backward, actor (5-wide head inside)",
           "gru_seq_bwd_kernel<1, 1>": "K12 backward, critic (v_out inside)",
           "mlp_fwd_kernel": "K9 forward, loader / compute kernel (unaligned widths; until the middle of round 4 the narrow actor "
                             "inputs)",
           "mlp_fwd3_kernel<2, 1, 4>": "K9 forward, version 3, the critic's 384 wide input (full last chunk)",
           "mlp_fwd3_kernel<2, 1, 2>": "K9 forward, version 3, the actor's 48 wide input (last chunk: 2 groups of 8 columns)",
           "mlp_dw1_direct_kernel": "K9 first-layer weight gradient, critic",
           "mlp_dw1_rows_kernel": "K9 first-layer weight gradient, actor",
           "mlp_bwd_kernel<2, 1, 0>": "K9 backward chain, action head",
           "mlp_bwd_kernel<2, 1, 1>": "K9 backward chain, value head"}
XCDS, SIMDS = 8, 1024


def main():
    dur = {}
    for r in csv.DictReader(open(os.path.join(SRC, W + "_kernel_trace.csv"))):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(os.path.join(SRC, W + "_counter_collection.csv"))):
        for frag in KERNELS:
            if frag in r["Kernel_Name"]:
                acc[frag][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {"what": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY "
                   "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python bench.py --workload %s --steps 1 --warmup 1 "
                   "(tools/pmc_sq_pass.sh); " % W +
                   "means over the dispatches of the run",
           "kernels": {}}
    for frag, disp in acc.items():
        n = len(disp)
        mean = lambda name: sum(d.get(name, 0.0) for d in disp.values()) / n      # noqa: E731
        ns = sum(dur[i] for i in disp if i in dur) / n
        cycles = mean("GRBM_GUI_ACTIVE") / XCDS
        mfma = mean("SQ_VALU_MFMA_BUSY_CYCLES")
        out["kernels"][frag] = {
            "what": KERNELS[frag], "dispatches": n, "duration_ms": round(ns / 1e6, 4),
            "shader_clock_ghz": round(cycles / ns, 3),
            "mfma_busy_share_of_an_average_simd": round(mfma / (SIMDS * cycles), 3),
            "mfma_busy_cycles_per_simd": round(mfma / SIMDS),
            "busy_share_at_the_nominal_2.4_ghz": round(mfma / (SIMDS * 2.4 * ns), 3),
            "sq_wave_quad_cycles": mean("SQ_WAVE_CYCLES"), "sq_wait_inst_any": mean("SQ_WAIT_INST_ANY"),
            "sq_wait_any": mean("SQ_WAIT_ANY"), "sq_active_inst_any": mean("SQ_ACTIVE_INST_ANY")}
    dst = os.path.join(ROOT, "profiles", TAG + "_pmc_sq_summary.json" if W == "ns" else TAG + "_pmc_sq_summary_%s.json" % W)
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, v["duration_ms"], "ms", v["shader_clock_ghz"], "GHz", "MFMA busy", v["mfma_busy_share_of_an_average_simd"],
              "of nominal", v["busy_share_at_the_nominal_2.4_ghz"])


if __name__ == "__main__":
    sys.exit(main())
