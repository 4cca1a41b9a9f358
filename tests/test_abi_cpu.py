"""CPU-side checks of the C ABI: the shared library loads and exports every symbol that
include/mappo_hip.h declares, the ctypes table covers the header, and argument validation rejects
bad calls before anything is enqueued (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from onpolicy import _native

HEADER = os.path.join(ROOT, "include", "mappo_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mappo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 13
    lib = _native.lib()
    for n in names:
        assert hasattr(lib, n), "libmappo_hip.so does not export %s" % n
        assert n in _native.SIGNATURES, "ctypes table misses %s" % n
    assert sorted(_native.SIGNATURES) == names
    assert lib.mappo_abi_version() == 2
    assert b"gfx950" in lib.mappo_build_info()


def test_struct_layouts_match_header():
    # struct mappo_field: 2 pointers + 4 int32 = 32 bytes; struct mappo_slab: 2 pointers + int64 = 24
    assert ctypes.sizeof(_native.Field) == 32
    assert ctypes.sizeof(_native.Slab) == 24
    # struct mappo_record_field: 2 pointers + 4 int32; struct mappo_ppo_loss: 15 pointers, int64, int, 4 floats,
    # unsigned -- field order as in the header
    assert ctypes.sizeof(_native.RecordField) == 32
    assert ctypes.sizeof(_native.PPOLoss) == 15 * 8 + 8 + 4 + 4 * 4 + 4
    src = open(HEADER).read()
    body = src[src.index("typedef struct mappo_ppo_loss {"):src.index("} mappo_ppo_loss_t;")]
    declared = re.findall(r"\b(\w+);", body)
    assert declared == [name for name, _ in _native.PPOLoss._fields_]


def test_new_entry_points_validate_arguments():
    lib = _native.lib()
    assert lib.mappo_gae_mat_f32(None, None, None, None, None, None, None, None, None, 4, 4, 2, 0.99, 0.95, 0, None) == -1
    assert lib.mappo_ppo_loss_f32(None, None) == -1
    loss = _native.PPOLoss()                       # all-NULL struct: inv_denoms missing
    assert lib.mappo_ppo_loss_f32(ctypes.byref(loss), None) == -1
    assert lib.mappo_gru_cell_fwd(None, None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_gru_cell_bwd(None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_gru_step_fwd(None, None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_bias_act_layernorm_fwd(None, None, None, None, None, None, None, 4, 64, 1e-5, 1, None) == -1
    assert lib.mappo_bias_act_layernorm_bwd(None, None, None, None, None, None, None, None, None, None, None, 4, 64, 1,
                                            None) == -1


def test_argument_validation_without_device():
    lib = _native.lib()
    assert lib.mappo_gae_f32(None, None, None, None, None, None, None, None, None, None, 4, 4,
                             0.99, 0.95, 1, None) == -1
    assert lib.mappo_adv_reduce(None, 1, None, None) == -1
    assert lib.mappo_gather_rows(None, 1, None, 1, None, None) == -1
    assert lib.mappo_slab_copy(None, 1, None) == -1
    assert lib.mappo_gae_partial_rows(0) == 0
    assert lib.mappo_gae_partial_rows(32768) == 2048
    assert lib.mappo_error_string(-2).decode().startswith("a size")
    assert lib.mappo_error_string(0) == b"ok"


def test_buffer_refuses_cpu():
    """The product path has no CPU fallback: it must fail loudly."""
    import torch
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    args = make_args(episode_length=4, n_rollout_threads=2)
    with pytest.raises(RuntimeError, match="HIP"):
        SharedReplayBuffer(args, 2, Box((3,)), Box((6,)), Discrete(5), device=torch.device("cpu"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP"):
            SharedReplayBuffer(args, 2, Box((3,)), Box((6,)), Discrete(5))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.lib()


def test_header_is_plain_c():
    """include/mappo_hip.h must be consumable from C (the cgo / JNI / ctypes side of an integration) and C++."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    for lang, std in (("c", "c99"), ("c++", "c++17")):
        out = subprocess.run(["gcc", "-fsyntax-only", "-x", lang, "-std=" + std, "-Wall", "-Werror", HEADER],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_default_arithmetic_is_the_six_term_form_and_a_per_call_field():
    """Round 5 (VERDICT r4's ruling): the matrix products of K9 / K12 default to float32 products from six bf16 x bf16 terms;
    the choice is the ``arith`` field of the two structs (0 = what a zero-initialised struct selects), not process state:
    no option bit of mappo_mlp_set_flags selects arithmetic any more.  The field sits in what was padding, so the struct
    layouts of ABI version 1 are unchanged."""
    import subprocess
    import sys
    src = open(HEADER).read()
    assert re.search(r"#define MAPPO_ARITH_SIX_TERM 0\b", src) and re.search(r"#define MAPPO_ARITH_F32_MFMA 1\b", src)
    assert _native.MLP().arith == _native.ARITH_SIX_TERM == 0 and _native.GRUSeq().arith == 0
    assert _native.MLP.arith.offset == _native.MLP.ln_eps.offset + 4 and _native.MLP.w1.offset == _native.MLP.arith.offset + 4
    assert _native.GRUSeq.arith.offset == _native.GRUSeq.L.offset + 4 and _native.GRUSeq.mb.offset == _native.GRUSeq.arith.offset + 4
    assert ctypes.sizeof(_native.MLP) == 248 and ctypes.sizeof(_native.GRUSeq) == 248        # as in ABI version 1
    env = {k: v for k, v in os.environ.items() if k not in ("MAPPO_MLP_FLAGS", "MAPPO_MATRIX_ARITHMETIC")}
    code = ("import sys; sys.path.insert(0, %r); from onpolicy import _native; from onpolicy.config import get_config; "
            "print(_native.default_arith(), get_config().parse_known_args([])[0].matrix_arithmetic, "
            "_native.lib().mappo_mlp_set_flags(0), _native.lib().mappo_mlp_set_flags(1024))" % os.path.join(ROOT, "on-policy_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.strip().splitlines()[-1] == "0 None 0 -1"
    for name in ("mappo_mlp", "mappo_gru_seq"):
        body = src[src.index("typedef struct %s {" % name):src.index("} %s_t;" % name)]
        assert re.search(r"int32_t arith;", body), name


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 13
    lib = _native.lib()
    for n in names:
        assert hasattr(lib, n), "libmappo_hip.so does not export %s" % n
        assert n in _native.SIGNATURES, "ctypes table misses %s" % n
    assert sorted(_native.SIGNATURES) == names
    assert lib.mappo_abi_version() == 2
    assert b"gfx950" in lib.mappo_build_info()


def test_struct_layouts_match_header():
    # struct mappo_field: 2 pointers + 4 int32 = 32 bytes; struct mappo_slab: 2 pointers + int64 = 24
    assert ctypes.sizeof(_native.Field) == 32
    assert ctypes.sizeof(_native.Slab) == 24
    # struct mappo_record_field: 2 pointers + 4 int32; struct mappo_ppo_loss: 15 pointers, int64, int, 4 floats,
    # unsigned -- field order as in the header
    assert ctypes.sizeof(_native.RecordField) == 32
    assert ctypes.sizeof(_native.PPOLoss) == 15 * 8 + 8 + 4 + 4 * 4 + 4
    src = open(HEADER).read()
    body = src[src.index("typedef struct mappo_ppo_loss {"):src.index("} mappo_ppo_loss_t;")]
    declared = re.findall(r"\b(\w+);", body)
    assert declared == [name for name, _ in _native.PPOLoss._fields_]


def test_new_entry_points_validate_arguments():
    lib = _native.lib()
    assert lib.mappo_gae_mat_f32(None, None, None, None, None, None, None, None, None, 4, 4, 2, 0.99, 0.95, 0, None) == -1
    assert lib.mappo_ppo_loss_f32(None, None) == -1
    loss = _native.PPOLoss()                       # all-NULL struct: inv_denoms missing
    assert lib.mappo_ppo_loss_f32(ctypes.byref(loss), None) == -1
    assert lib.mappo_gru_cell_fwd(None, None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_gru_cell_bwd(None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_gru_step_fwd(None, None, None, None, None, None, None, None, None, 4, 64, None) == -1
    assert lib.mappo_bias_act_layernorm_fwd(None, None, None, None, None, None, None, 4, 64, 1e-5, 1, None) == -1
    assert lib.mappo_bias_act_layernorm_bwd(None, None, None, None, None, None, None, None, None, None, None, 4, 64, 1,
                                            None) == -1


def test_argument_validation_without_device():
    lib = _native.lib()
    assert lib.mappo_gae_f32(None, None, None, None, None, None, None, None, None, None, 4, 4,
                             0.99, 0.95, 1, None) == -1
    assert lib.mappo_adv_reduce(None, 1, None, None) == -1
    assert lib.mappo_gather_rows(None, 1, None, 1, None, None) == -1
    assert lib.mappo_slab_copy(None, 1, None) == -1
    assert lib.mappo_gae_partial_rows(0) == 0
    assert lib.mappo_gae_partial_rows(32768) == 2048
    assert lib.mappo_error_string(-2).decode().startswith("a size")
    assert lib.mappo_error_string(0) == b"ok"


def test_buffer_refuses_cpu():
    """The product path has no CPU fallback: it must fail loudly."""
    import torch
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    args = make_args(episode_length=4, n_rollout_threads=2)
    with pytest.raises(RuntimeError, match="HIP"):
        SharedReplayBuffer(args, 2, Box((3,)), Box((6,)), Discrete(5), device=torch.device("cpu"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP"):
            SharedReplayBuffer(args, 2, Box((3,)), Box((6,)), Discrete(5))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.lib()


def test_header_is_plain_c():
    """include/mappo_hip.h must be consumable from C (the cgo / JNI / ctypes side of an integration) and C++."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    for lang, std in (("c", "c99"), ("c++", "c++17")):
        out = subprocess.run(["gcc", "-fsyntax-only", "-x", lang, "-std=" + std, "-Wall", "-Werror", HEADER],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_no_tuning_bit_is_set_out_of_the_box():
    """Out of the box no option bit of the K9 launchers is set (they are tuning / A-B hooks; arithmetic is the per-call `arith`
    field, see test_default_arithmetic_is_the_six_term_form_and_a_per_call_field).  (A fresh interpreter: the flags of this process may have been set by a
    test.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "MAPPO_MLP_FLAGS"}
    code = ("import sys; sys.path.insert(0, %r); from onpolicy import _native; "
            "print(_native.lib().mappo_mlp_set_flags(0))" % os.path.join(ROOT, "on-policy_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.strip().splitlines()[-1] == "0"
