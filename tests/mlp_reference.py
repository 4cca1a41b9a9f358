"""TEST INFRASTRUCTURE: the fused trunk (K9, include/mappo_hip.h mappo_mlp_*) restated in float64 torch ops -- the
reference's modules applied to the same parameters (onpolicy/algorithms/utils/mlp.py:6-58 with the input LayerNorm's
affine half folded into the first Linear, plus the output Linear) -- and a ctypes mirror of ``mappo_mlp_t`` that works
for the product library (device pointers) and for the host-emulated build (tests/simt, host pointers)."""
import ctypes

import numpy as np
import torch

_vp, _i64, _i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32


class MLP(ctypes.Structure):
    _fields_ = [("src", _vp), ("row_tab", _vp), ("rows", _i64),
                ("din", _i32), ("n_layers", _i32), ("act", _i32), ("out", _i32), ("ln_eps", ctypes.c_float),
                ("arith", _i32), ("w1", _vp), ("bias", _vp * 3), ("ln_g", _vp * 3), ("ln_b", _vp * 3), ("w2", _vp * 2),
                ("wh", _vp), ("bh", _vp), ("y", _vp), ("z", _vp * 3), ("ln_stats", _vp * 3), ("dy", _vp), ("dz1", _vp),
                ("workspace", _vp), ("grads", _vp)]


def bind(lib):
    lib.mappo_mlp_forward.restype = ctypes.c_int
    lib.mappo_mlp_forward.argtypes = [ctypes.POINTER(MLP), _vp]
    lib.mappo_mlp_backward.restype = ctypes.c_int
    lib.mappo_mlp_backward.argtypes = [ctypes.POINTER(MLP), _vp]
    lib.mappo_mlp_grad_floats.restype = _i64
    lib.mappo_mlp_grad_floats.argtypes = [ctypes.c_int] * 3
    lib.mappo_mlp_workspace_floats.restype = _i64
    lib.mappo_mlp_workspace_floats.argtypes = [ctypes.c_int] * 3
    lib.mappo_mlp_row_table_ints.restype = _i64
    lib.mappo_mlp_row_table_ints.argtypes = [_i64]
    lib.mappo_mlp_row_table.restype = ctypes.c_int
    lib.mappo_mlp_row_table.argtypes = [_vp, _i64, _i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]
    lib.mappo_standardize_rows.restype = ctypes.c_int
    lib.mappo_standardize_rows.argtypes = [_vp, _i64, ctypes.c_int, ctypes.c_float, _vp, _vp]
    lib.mappo_standardize_rows_ld.restype = ctypes.c_int
    lib.mappo_standardize_rows_ld.argtypes = [_vp, _i64, ctypes.c_int, ctypes.c_float, _vp, ctypes.c_int, _vp]
    lib.mappo_mlp_set_grid_cap.restype = ctypes.c_int
    lib.mappo_mlp_set_grid_cap.argtypes = [ctypes.c_int]
    lib.mappo_mlp_set_flags.restype = ctypes.c_int
    lib.mappo_mlp_set_flags.argtypes = [ctypes.c_int]
    return lib


def random_net(rng, din, n_layers, out, scale=0.3):
    f = lambda *s: (scale * rng.standard_normal(s)).astype(np.float32)
    p = {"w1": f(64, din), "wh": f(max(out, 1), 64)[:out], "bh": f(max(out, 1))[:out]}
    for l in range(n_layers):
        p["bias%d" % l] = f(64)
        p["ln_g%d" % l] = (1.0 + f(64)).astype(np.float32)
        p["ln_b%d" % l] = f(64)
        if l > 0:
            p["w2_%d" % (l - 1)] = f(64, 64)
    return p


def rows_of_fragments(z, rows):
    """The saved activations come back in the kernels' own order (include/mappo_hip.h: element (row r, feature
    32 t + 8 q + 4 h + e) at float (r // 32) * 2048 + (4 t + q) * 256 + (32 h + r % 32) * 4 + e) -> [rows, 64]."""
    flat = np.asarray(z).reshape(-1)
    r = np.arange(rows)[:, None]
    f = np.arange(64)[None, :]
    t, q, h, e = f // 32, (f % 32) // 8, (f % 8) // 4, f % 4
    return flat[(r // 32) * 2048 + (4 * t + q) * 256 + (32 * h + r % 32) * 4 + e]


def source_rows(idx, rows, chunk_len=0, mb=0, T=0, N=0, A=0):
    """Source row of every launch row (shared_buffer.py:379-396 rows mode, :554-604 chunk mode)."""
    if idx is None:
        return np.arange(rows)
    if chunk_len <= 0:
        return np.asarray(idx[:rows])
    r = np.arange(rows)
    l, j = r // mb, r % mb
    f = np.asarray(idx)[j] * chunk_len + l
    n, rem = f // (A * T), f % (A * T)
    a, t = rem // T, rem % T
    return (t * N + n) * A + a


def standardize_ref(src, eps):
    x = torch.as_tensor(src, dtype=torch.float64)
    return (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + eps)


def forward_ref(params, src, srows, standardize, n_layers, act, out, eps=1e-5, in_eps=1e-5):
    """-> (y, [z_l]) in float64; ``params``: dict of float64 tensors (requires_grad for the backward)."""
    x = torch.as_tensor(src, dtype=torch.float64)[torch.as_tensor(srows)]
    if standardize:
        x = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + in_eps)
    fn = {0: lambda v: v, 1: torch.tanh, 2: torch.relu}[act]
    h, zs = x, []
    for l in range(n_layers):
        w = params["w1"] if l == 0 else params["w2_%d" % (l - 1)]
        z = h @ w.T + params["bias%d" % l]
        zs.append(z)
        h = torch.nn.functional.layer_norm(fn(z), (64,), params["ln_g%d" % l], params["ln_b%d" % l], eps)
    y = h if out == 0 else h @ params["wh"].T + params["bh"]
    return y, zs


def flat_grads(grads, din, n_layers, out):
    """dict name -> tensor in the library's flat gradient order."""
    parts = [grads["w1"].reshape(-1)]
    for l in range(n_layers):
        parts += [grads["bias%d" % l], grads["ln_g%d" % l], grads["ln_b%d" % l]]
    for l in range(1, n_layers):
        parts.append(grads["w2_%d" % (l - 1)].reshape(-1))
    if out > 0:
        parts += [grads["wh"].reshape(-1), grads["bh"]]
    return torch.cat(parts)
