"""Compile-time guard on the kernels' register / LDS budgets (no GPU needed: hipcc cross-compiles and reports each
kernel's resource usage).  profiles/kernel_resources.json is the committed snapshot (tools/kernel_resources.py
--write); here: no kernel may spill to scratch memory, the hot kernels keep the occupancy they were tuned for, and the
quick-to-compile sources still produce exactly the snapshot (all eight with MAPPO_CHECK_ALL_KERNEL_RESOURCES=1)."""
import importlib.util
import json
import os
import shutil

import pytest

from conftest import ROOT

SNAPSHOT = os.path.join(ROOT, "profiles", "kernel_resources.json")


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_snapshot_has_no_spills_and_expected_occupancy():
    table = json.load(open(SNAPSHOT))
    assert len(table) > 300 and {k.split(" :: ")[0] for k in table} == {
        "mappo_gae.hip", "mappo_copy.hip", "mappo_norm.hip", "mappo_loss.hip", "mappo_rnn.hip", "mappo_mlp.hip",
        "mappo_perm.hip", "mappo_env.hip", "mappo_optim.hip"}
    # no spills, except the identity-activation forward trunk (<= 32 bytes: loop-invariant addresses reloaded once per tile).
    # (Round 5: the 64-wide form of the time-parallel GAE scan -- a tuning variant that was never selected automatically and
    # spilled 68-196 bytes per lane with time limits -- is no longer built.)
    # (Round 6: K15's forward no longer spills -- 148 / 156 bytes per lane in round 5 -- its accumulators start from the bias
    # and the tile epilogue is 64 stores straight from the accumulator registers.)
    spills = {k: v["scratch_bytes"] for k, v in table.items() if v["scratch_bytes"]}
    assert all("mlp_fwd_kernel<0," in k and b <= 32 for k, b in spills.items()), spills
    # K15: both GEMM kernels one wave per SIMD with their 256 accumulators in the AGPR half; the forward in 8 instances (rows
    # aligned or not x bias or not x two / four feature tiles per MFMA group), the default (two tiles) within 128 vector registers;
    # the weight gradient in 4 (rows aligned or not x the X tile's bf16 planes shared through LDS or split by every wave)
    k15 = {k: v for k, v in table.items() if "lin::lin_fwd_kernel" in k or "lin::lin_wgrad_kernel" in k}
    # (+ 2 forward instances with the block's ReLU + LayerNorm in the epilogue, round 6: the accumulators are read value by value
    # through prim::acc_get -- ordinary vector arithmetic on them moved all 256 into vector registers and spilled 116-208 bytes)
    assert len(k15) == 14 and all(v["agprs"] == 256 and v["occupancy"] == 1 and v["scratch_bytes"] == 0 for v in k15.values())
    assert all(v["vgprs"] <= 128 for k, v in k15.items() if "lin_fwd_kernel" in k and ", 2, " in k)
    assert sum(1 for k in k15 if "lin_fwd_kernel" in k and k.rstrip(")").endswith("true>(lin::FwdArgs")) == 2, sorted(k15)
    # round 4: the version-3 forward -- 15 instances (layers x activation x groups of 8 columns in a row's last chunk), two
    # waves per SIMD (<= 256 registers), no scratch
    f3 = {k: v for k, v in table.items() if "mlp_fwd3_kernel<" in k}
    # (+ 9 two-layer instances with the hidden layer in the six-term bf16 form: what MAPPO_ARITH_SIX_TERM runs for the
    # two-layer shapes version 4 does not take; device-verified in round 5)
    assert len(f3) == 24 and all(v["scratch_bytes"] == 0 and v["occupancy"] == 2 for v in f3.values())
    # version 4 (both layers on the bf16 matrix pipe, MAPPO_ARITH_SIX_TERM): one wave per SIMD with the hidden layer's weight
    # planes in registers (up to 512), no scratch in its 9 instances (round 5: the form with a float32 hidden layer is gone)
    f4 = {k: v for k, v in table.items() if "mlp_fwd4_kernel<" in k}
    assert len(f4) == 9 and all(v["scratch_bytes"] == 0 and v["occupancy"] == 1 for v in f4.values())
    pick = lambda frag: [v for k, v in table.items() if frag in k]      # noqa: E731
    # K9: the forward trunk fits two workgroups of eight waves on a CU (<= 128 registers), the direct-to-LDS weight
    # gradient and the backward chain run one wave per SIMD with their accumulators in the AGPR half of the file
    assert all(v["occupancy"] == 4 and v["vgprs"] <= 128 for v in pick("mlp_fwd_kernel"))
    assert all(v["agprs"] >= 32 for v in pick("mlp_dw1_direct_kernel") + pick("mlp_dw1_rows_kernel"))
    # round 3: the backward chain -- one wave per SIMD by design (measured: a second wave only stretches every phase),
    # no scratch in any of its 15 instances (+ 6 six-term bf16 instances of the two-layer chain, round 4, + 6 that also
    # accumulate the first-layer weight gradient of narrow inputs, round 5: 256 + 241 registers at most); the GRU chunk kernels
    # likewise one workgroup per CU (104 KB of weights in LDS)
    assert len(pick("mlp_bwd_kernel")) == 27 and all(v["occupancy"] == 1 and v["scratch_bytes"] == 0 for v in pick("mlp_bwd_kernel"))
    assert all(v["occupancy"] == 1 for v in pick("gru_seq_fwd_kernel") + pick("gru_seq_bwd_kernel"))
    # ... its nine backward instances: no head / head of <= 2, 6, 18 outputs x the head's gradient sums on or off, x 2: float32
    # MFMA, and the six-term form with all six blocks of the transposed weights as bf16 planes (8 instances; the ninth -- the
    # widest head-sum instance -- would spill and keeps the four-block form of round 4)
    assert len(pick("gru_seq_bwd_kernel")) == 18 and all(v["scratch_bytes"] == 0 for v in pick("gru_seq_bwd_kernel"))
    assert all(v["occupancy"] >= 7 for v in pick("ppo_loss_kernel"))
    # round 5: K12's weight gradients in one six-term launch -- two workgroups per CU (72 KB of LDS each), no scratch
    (wg,) = pick("gru_wgrad_kernel")
    assert wg["occupancy"] == 2 and wg["scratch_bytes"] == 0
    # ... like the six-term direct first-layer weight-gradient kernel in its default form
    assert all(v["occupancy"] >= 2 and v["scratch_bytes"] == 0 for k, v in table.items() if "mlp_dw1_direct_kernel<" in k and ", 2, true>" in k)
    assert all(v["occupancy"] >= 6 for v in pick("gru_fwd_kernel")) and all(v["occupancy"] == 8 for v in pick("gru_bwd_kernel"))
    (step,) = pick("gru_step_fwd_kernel")                                # one workgroup per CU by design: W_hh in LDS
    assert step["lds_bytes"] == 49152 and step["occupancy"] == 1
    # the H = 64 LayerNorm instances of the benchmarked MLP (D = 64: <1|4, 16, 1, ...>) run at full occupancy
    assert all(v["occupancy"] == 8 for k, v in table.items() if "ln_fwd_kernel<4, 16, 1" in k or "ln_bwd_kernel<4, 16, 1" in k)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_sources_still_compile_to_the_snapshot():
    tool = _tool()
    table = json.load(open(SNAPSHOT))
    sources = tool.SOURCES if os.environ.get("MAPPO_CHECK_ALL_KERNEL_RESOURCES") == "1" else \
        ("mappo_copy.hip", "mappo_loss.hip", "mappo_rnn.hip", "mappo_mlp.hip", "mappo_perm.hip", "mappo_optim.hip")     # (gae, env: ~100 s each)
    for src in sources:
        fresh = {"%s :: %s" % (src, k.pop("kernel")): k for k in tool.analyse(src)}
        committed = {k: v for k, v in table.items() if k.startswith(src + " :: ")}
        assert fresh == committed, "%s: resource usage changed -- rerun tools/kernel_resources.py --write and review" % src


def test_isa_snapshot_shows_the_cdna4_instructions_the_design_relies_on():
    """profiles/isa_summary.json (tools/isa_summary.py --write): the GAE scan stages its tiles with gfx950's 128-bit
    direct-to-LDS loads, the one-kernel GRU step runs its hidden projection on the f32 MFMA, streaming kernels use
    128-bit and non-temporal accesses, and nothing touches scratch memory."""
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_summary.json")))
    assert isa["mappo_gae.hip"]["lds_dma_128bit"] > 1000 and isa["mappo_gae.hip"]["dpp_or_permute"] > 0
    assert isa["mappo_rnn.hip"]["mfma_kinds"] == {"v_mfma_f32_32x32x2_f32": isa["mappo_rnn.hip"]["mfma_f32"]}
    # the buffer path is HBM-bound (no MFMA); the fused trunk (K9) is the MFMA path and fetches its weight-gradient
    # operands with direct-to-LDS loads
    assert all(isa[s]["mfma_f32"] == 0 for s in isa if s not in ("mappo_rnn.hip", "mappo_mlp.hip"))
    # (plus the bf16 matrix instruction of the six-term kernels -- float32 products from six bf16 terms: the default
    # arithmetic since round 5; the float32-MFMA instances stay in the library for --matrix_arithmetic f32_mfma and for the
    # shapes without a six-term kernel)
    kinds = isa["mappo_mlp.hip"]["mfma_kinds"]
    assert set(kinds) == {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_bf16"}
    assert sum(kinds.values()) == isa["mappo_mlp.hip"]["mfma_f32"] and kinds["v_mfma_f32_32x32x2_f32"] > 12000
    assert isa["mappo_mlp.hip"]["lds_dma_128bit"] > 50
    for src in ("mappo_gae.hip", "mappo_copy.hip", "mappo_norm.hip"):
        assert isa[src]["global_load_128bit"] > 0 and isa[src]["non_temporal"] > 0, src
    assert all(row["scratch_access"] == 0 for src, row in isa.items() if src not in ("mappo_gae.hip", "mappo_mlp.hip"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_gru_step_still_compiles_to_mfma():
    spec = importlib.util.spec_from_file_location("isa_summary", os.path.join(ROOT, "tools", "isa_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.summarise("mappo_rnn.hip")
    assert fresh == json.load(open(os.path.join(ROOT, "profiles", "isa_summary.json")))["mappo_rnn.hip"]
