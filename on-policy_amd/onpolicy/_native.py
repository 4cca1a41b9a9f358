"""ctypes binding of libmappo_hip.so (C ABI: include/mappo_hip.h).

There is deliberately no fallback: if the library is missing or a call fails this raises.  The
device path is the product; a CPU stand-in would silently invalidate every parity and
performance claim (the CPU oracle lives under oracle/ and is test infrastructure only).
"""
import ctypes
import os

import torch  # imported first so that the HIP runtime torch ships is the one this library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAPPO_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libmappo_hip.so"))

MAX_FIELDS = 16
MAX_MINIBATCHES = 64        # MAPPO_PERM_MAX_MINIBATCHES
GAE_USE_GAE, GAE_PROPER_TIME_LIMITS, GAE_DENORM, GAE_EXACT = 1, 2, 4, 8
# MAPPO_ARITH_*: how K9 / K12 form their float32 matrix products (the per-call ``arith`` field of MLP / GRUSeq)
ARITH_SIX_TERM, ARITH_F32_MFMA = 0, 1
ARITHMETICS = {"six_term": ARITH_SIX_TERM, "f32_mfma": ARITH_F32_MFMA}
ABI_VERSION = 2


def arith_code(name):
    """``--matrix_arithmetic`` name (or an ARITH_* code) -> the code the structs carry."""
    if isinstance(name, int):
        if name not in ARITHMETICS.values():
            raise ValueError("unknown matrix arithmetic code %r" % (name,))
        return name
    try:
        return ARITHMETICS[name]
    except KeyError:
        raise ValueError("unknown matrix arithmetic %r (one of %s)" % (name, sorted(ARITHMETICS)))


def default_arith():
    """What a module without an explicit choice uses: MAPPO_MATRIX_ARITHMETIC (A / B tooling), else the six-term form."""
    return arith_code(os.environ.get("MAPPO_MATRIX_ARITHMETIC", "six_term"))

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int


class Field(ctypes.Structure):
    """struct mappo_field (include/mappo_hip.h)."""
    _fields_ = [("src", _vp), ("dst", _vp), ("width", ctypes.c_int32), ("first_only", ctypes.c_int32),
                ("normalize", ctypes.c_int32), ("standardize", ctypes.c_int32)]


class RecordField(ctypes.Structure):
    """struct mappo_record_field (include/mappo_hip.h)."""
    _fields_ = [("src", _vp), ("dst", _vp), ("width", ctypes.c_int32), ("offset", ctypes.c_int32),
                ("normalize", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Slab(ctypes.Structure):
    """struct mappo_slab (include/mappo_hip.h)."""
    _fields_ = [("src", _vp), ("dst", _vp), ("count", _i64)]


MAX_STD_SLABS = 4       # MAPPO_MAX_STD_SLABS


class StdSlab(ctypes.Structure):
    """struct mappo_std_slab (include/mappo_hip.h)."""
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i64), ("D", _int), ("ld", _int), ("eps", ctypes.c_float)]


class PPOLoss(ctypes.Structure):
    """struct mappo_ppo_loss (include/mappo_hip.h)."""
    _fields_ = [(n, _vp) for n in ("logits", "available", "actions", "old_logp", "adv", "active", "factor", "values",
                                   "value_preds", "returns", "norm", "inv_denoms", "dlogits", "dvalues", "sums")] + \
               [("rows", _i64), ("n_actions", ctypes.c_int), ("clip", ctypes.c_float),
                ("huber_delta", ctypes.c_float), ("entropy_coef", ctypes.c_float),
                ("value_loss_coef", ctypes.c_float), ("flags", ctypes.c_uint)]


class MLP(ctypes.Structure):
    """struct mappo_mlp (include/mappo_hip.h): the fused hidden-64 trunk."""
    _fields_ = [("src", _vp), ("row_tab", _vp), ("rows", _i64),
                ("din", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("act", ctypes.c_int32), ("out", ctypes.c_int32),
                ("ln_eps", ctypes.c_float), ("arith", ctypes.c_int32), ("w1", _vp), ("bias", _vp * 3), ("ln_g", _vp * 3), ("ln_b", _vp * 3),
                ("w2", _vp * 2), ("wh", _vp), ("bh", _vp), ("y", _vp), ("z", _vp * 3), ("ln_stats", _vp * 3), ("dy", _vp), ("dz1", _vp),
                ("workspace", _vp), ("grads", _vp)]


class GRUSeq(ctypes.Structure):
    """struct mappo_gru_seq (include/mappo_hip.h): the GRU of a recurrent policy over a whole chunk."""
    _fields_ = [(n, _vp) for n in ("x", "h0", "masks", "w_ih", "w_hh", "b_ih", "b_hh", "ln_g", "ln_b")] + \
               [("ln_eps", ctypes.c_float), ("H", ctypes.c_int32), ("L", ctypes.c_int32), ("arith", ctypes.c_int32), ("mb", _i64)] + \
               [(n, _vp) for n in ("y", "h_last", "gates", "hm", "stats", "dy", "dx", "dgi", "dq", "dh0", "dh_last",
                                   "ln_grads", "workspace", "head_w", "head_b")] + \
               [("head_out", ctypes.c_int32), ("logits", _vp), ("dlogits", _vp), ("head_sums", ctypes.c_int32)]


ADAM_MAX_TENSORS = 64


class Adam(ctypes.Structure):
    """struct mappo_adam (include/mappo_hip.h): gradient clipping + Adam over all tensors of one network."""
    _fields_ = [(n, _vp * ADAM_MAX_TENSORS) for n in ("param", "grad", "exp_avg", "exp_avg_sq", "step")] + \
               [("numel", _i64 * ADAM_MAX_TENSORS), ("n", ctypes.c_int32)] + \
               [(n, ctypes.c_double) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "max_grad_norm")] + \
               [("grad_norm", _vp), ("workspace", _vp), ("lr_device", _vp)]


LOSS_HUBER, LOSS_CLIPPED_VALUE, LOSS_POLICY_ACTIVE_MASKS, LOSS_VALUE_ACTIVE_MASKS = 1, 2, 4, 8

# symbol -> (restype, argtypes); must list every function include/mappo_hip.h declares
SIGNATURES = {
    "mappo_gae_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64,
                             ctypes.c_double, ctypes.c_double, ctypes.c_uint, _vp]),
    "mappo_gae_mat_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _int,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_uint, _vp]),
    "mappo_gae_partial_rows": (_i64, [_i64]),
    "mappo_gae_set_variant": (_int, [_int]),
    "mappo_gae_last_variant": (_int, []),
    "mappo_gae_time_next_launch": (_int, []),
    "mappo_gae_timed_launch_ms": (_int, [_int, ctypes.POINTER(ctypes.c_float)]),
    "mappo_advantages_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]),
    "mappo_adv_reduce": (_int, [_vp, _i64, _vp, _vp]),
    "mappo_adv_stats": (_int, [_vp, _vp, _vp]),
    "mappo_adv_normalize": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "mappo_gather_rows": (_int, [ctypes.POINTER(Field), _int, _vp, _i64, _vp, _vp]),
    "mappo_gather_chunks": (_int, [ctypes.POINTER(Field), _int, _vp, _i64, _int, _int, _i64, _int, _vp, _vp]),
    "mappo_pack_records": (_int, [ctypes.POINTER(RecordField), _int, _vp, _int, _i64, _vp]),
    "mappo_gather_records": (_int, [_vp, _int, ctypes.POINTER(RecordField), _int, _vp, _i64, _int, _int, _i64,
                                    _int, _vp, _vp]),
    "mappo_gather_set_variant": (_int, [_int]),
    "mappo_slab_copy": (_int, [ctypes.POINTER(Slab), _int, _vp]),
    "mappo_slab_copy_std": (_int, [ctypes.POINTER(Slab), _int, ctypes.POINTER(StdSlab), _int, _vp]),
    "mappo_layernorm_max_blocks": (_int, []),
    "mappo_act_layernorm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, ctypes.c_float, _int, _vp]),
    "mappo_act_layernorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp]),
    "mappo_layernorm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, ctypes.c_float, _vp]),
    "mappo_layernorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "mappo_bias_act_layernorm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, ctypes.c_float, _int, _vp]),
    "mappo_bias_act_layernorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int,
                                            _vp]),
    "mappo_gru_cell_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "mappo_gru_cell_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "mappo_gru_step_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "mappo_gru_seq_gates_floats": (_i64, [_int, _i64]),
    "mappo_gru_seq_stats_floats": (_i64, [_int, _i64]),
    "mappo_gru_seq_workspace_floats": (_i64, []),
    "mappo_gru_seq_forward": (_int, [ctypes.POINTER(GRUSeq), _vp]),
    "mappo_gru_seq_backward": (_int, [ctypes.POINTER(GRUSeq), _vp]),
    "mappo_gru_weight_grads_workspace_floats": (_i64, []),
    "mappo_gru_weight_grads": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "mappo_fold_input_norm_forward": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "mappo_fold_input_norm_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "mappo_minibatch_sums_workspace_doubles": (_i64, []),
    "mappo_minibatch_sums": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "mappo_minibatch_scales": (_int, [_vp, _vp, _int, _int, _vp, _vp]),
    "mappo_valuenorm_workspace_doubles": (_i64, []),
    "mappo_valuenorm_update": (_int, [_vp, _i64, _vp, ctypes.c_double, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mappo_adam_workspace_floats": (_i64, []),
    "mappo_clip_adam": (_int, [ctypes.POINTER(Adam), _vp]),
    "mappo_ppo_loss_f32": (_int, [ctypes.POINTER(PPOLoss), _vp]),
    "mappo_categorical_sample": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _vp]),
    "mappo_minibatch_workspace_ints": (_i64, [_i64, _int]),
    "mappo_minibatch_indices": (_int, [_i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "mappo_mlp_forward": (_int, [ctypes.POINTER(MLP), _vp]),
    "mappo_mlp_backward": (_int, [ctypes.POINTER(MLP), _vp]),
    "mappo_mlp_grad_floats": (_i64, [_int, _int, _int]),
    "mappo_mlp_workspace_floats": (_i64, [_int, _int, _int]),
    "mappo_mlp_row_table_ints": (_i64, [_i64]),
    "mappo_mlp_set_grid_cap": (_int, [_int]),
    "mappo_mlp_set_debug": (_int, [_vp]),
    "mappo_mlp_set_flags": (_int, [_int]),
    "mappo_mlp_row_table": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp]),
    "mappo_standardize_rows": (_int, [_vp, _i64, _int, ctypes.c_float, _vp, _vp]),
    "mappo_standardize_rows_ld": (_int, [_vp, _i64, _int, ctypes.c_float, _vp, _int, _vp]),
    "mappo_linear512_planes_floats": (_i64, [_int]),
    "mappo_linear512_prepare": (_int, [_vp, _int, _int, _int, _vp, _vp]),
    "mappo_linear512_forward": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "mappo_linear512_forward_norm": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _vp, ctypes.c_float, _int, _vp, _vp, _vp, _vp,
                                           _vp]),
    "mappo_linear512_wgrad_workspace_floats": (_i64, [_int]),
    "mappo_linear512_wgrad": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "mappo_simple_spread_step": (_int, [_vp] * 11 + [_i64, _int, _int, _int, _int, _vp]),
    "mappo_abi_version": (_int, []),
    "mappo_build_info": (ctypes.c_char_p, []),
    "mappo_error_string": (ctypes.c_char_p, [_int]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raises NativeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libmappo_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C on-policy_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("MAPPO_GAE_VARIANT"):      # tuning hook (tools/): kernel variant + option bits
            L.mappo_gae_set_variant(int(os.environ["MAPPO_GAE_VARIANT"]))
        if L.mappo_abi_version() != ABI_VERSION:
            raise NativeError("libmappo_hip.so ABI version %d, expected %d" % (L.mappo_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


# test instrumentation: how often each entry point was called (``count_calls(True)`` starts a fresh count, ``calls()`` reads
# it) -- lets a device test assert WHICH kernels carried an update (no silent framework fall-back can pass)
_CALLS = None


def count_calls(on=True):
    global _CALLS
    _CALLS = {} if on else None


def calls():
    return dict(_CALLS or {})


def check(code, what):
    if _CALLS is not None:
        _CALLS[what] = _CALLS.get(what, 0) + 1
    if code != 0:
        msg = lib().mappo_error_string(code).decode()
        raise NativeError("%s failed: %s (code %d)" % (what, msg, code))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_of(device):
    return torch.cuda.current_stream(device).cuda_stream
