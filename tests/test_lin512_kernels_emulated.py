"""K15 (csrc/mappo_lin_impl.h: tall Linear layers with 512 outputs in six-term bf16 arithmetic -- the GEMMs of the hidden-512
trunks of BASELINE configs[4]; reference onpolicy/algorithms/utils/mlp.py:6-58 at --hidden_size 512) on the host SIMT
emulator against float64: forward with and without bias, the transposed planes of the input gradient, the weight gradient
with ragged row counts, widths that are not multiples of 16 / 128, workgroups that loop over several tiles.  The same source
is compiled for gfx950; tests/test_gpu_lin512.py repeats the comparison on the device."""
import ctypes
import os

import numpy as np
import pytest

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ (host build of the emulator) not found")

U = 2.0 ** -24


@pytest.fixture(scope="module")
def emu():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt"))
    import build
    from onpolicy import _native
    lib = ctypes.CDLL(build.build())
    for name in ("mappo_linear512_planes_floats", "mappo_linear512_prepare", "mappo_linear512_forward", "mappo_linear512_forward_norm",
                 "mappo_linear512_wgrad_workspace_floats", "mappo_linear512_wgrad", "mappo_mlp_set_grid_cap"):
        res, args = _native.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def _aligned(shape, dtype=np.float32):
    """16-byte aligned array (the kernels' direct-to-LDS loads move 16-byte pieces)."""
    n = int(np.prod(shape))
    raw = np.empty(n + 4, dtype)
    off = (-raw.ctypes.data // 4) % 4
    return raw[off:off + n].reshape(shape)


def _planes(emu, w, K, ldw, transposed):
    planes = _aligned((emu.mappo_linear512_planes_floats(K),))
    planes[:] = np.nan
    assert emu.mappo_linear512_prepare(_ptr(w), K, ldw, transposed, _ptr(planes), None) == 0
    return planes


@pytest.mark.parametrize("rows,K,ldx,bias,cap", [(128 * 2 + 37, 40, 40, True, 0), (70, 1285, 1288, True, 0),
                                                 (128 * 5, 512, 512, False, 2), (33, 16, 16, True, 0), (200, 100, 104, False, 1),
                                                 (150, 1285, 1285, True, 0), (40, 21, 21, False, 0)])
def test_forward_vs_float64(emu, rows, K, ldx, bias, cap):
    rng = np.random.default_rng(rows + K)
    x = _aligned((rows, ldx))
    x[:] = 0.0
    x[:, :K] = rng.standard_normal((rows, K)) * 1.5 + 0.3
    w = (rng.standard_normal((512, K)) * 0.2).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32) if bias else None
    planes = _planes(emu, w, K, K, 0)
    y = _aligned((rows, 512))
    y[:] = np.nan
    emu.mappo_mlp_set_grid_cap(cap)
    try:
        assert emu.mappo_linear512_forward(_ptr(x), rows, K, ldx, _ptr(planes), _ptr(b), _ptr(y), None) == 0
    finally:
        emu.mappo_mlp_set_grid_cap(0)
    x64, w64 = x[:, :K].astype(np.float64), w.astype(np.float64)
    ref = x64 @ w64.T + (b.astype(np.float64) if bias else 0.0)
    S = np.abs(x64) @ np.abs(w64).T + (np.abs(b.astype(np.float64)) if bias else 0.0)
    assert np.isfinite(y).all()
    err = np.abs(y - ref) / ((16.0 + K / 6.0) * U * S)
    print("\n[K15 forward rows %d K %d] worst error = %.2f of the bound" % (rows, K, err.max()))
    assert err.max() <= 1.0


@pytest.mark.parametrize("rows,K,ldx,cap", [(128 + 37, 40, 40, 0), (70, 1285, 1288, 0), (128 * 3, 512, 512, 1), (33, 21, 21, 0)])
def test_forward_with_the_block_epilogue_vs_float64(emu, rows, K, ldx, cap):
    """mappo_linear512_forward_norm: y = x W^T + b as the plain forward gives it (bit for bit: the epilogue only reads the
    accumulators), yn = LayerNorm(relu(y)), mean / rstd the statistics K6's backward expects (reference mlp.py:17-22)."""
    rng = np.random.default_rng(rows * 7 + K)
    x = _aligned((rows, ldx))
    x[:] = 0.0
    x[:, :K] = rng.standard_normal((rows, K)) * 1.5 + 0.3
    w = (rng.standard_normal((512, K)) * 0.2).astype(np.float32)
    b, gamma, beta = (_aligned((512,)) for _ in range(3))
    b[:] = rng.standard_normal(512)
    gamma[:] = 1.0 + 0.3 * rng.standard_normal(512)
    beta[:] = 0.2 * rng.standard_normal(512)
    planes = _planes(emu, w, K, K, 0)
    y, yn, y0 = (_aligned((rows, 512)) for _ in range(3))
    mean, rstd = np.full(rows, np.nan, np.float32), np.full(rows, np.nan, np.float32)
    for a in (y, yn, y0):
        a[:] = np.nan
    eps = 1e-5
    emu.mappo_mlp_set_grid_cap(cap)
    try:
        assert emu.mappo_linear512_forward(_ptr(x), rows, K, ldx, _ptr(planes), _ptr(b), _ptr(y0), None) == 0
        assert emu.mappo_linear512_forward_norm(_ptr(x), rows, K, ldx, _ptr(planes), _ptr(b), _ptr(gamma), _ptr(beta), eps, 2,
                                                _ptr(y), _ptr(yn), _ptr(mean), _ptr(rstd), None) == 0
    finally:
        emu.mappo_mlp_set_grid_cap(0)
    assert np.array_equal(y, y0)
    a64 = np.maximum(y.astype(np.float64), 0.0)             # the LayerNorm of the pre-activation the kernel itself produced
    mu = a64.mean(1)
    var = a64.var(1)
    ref = (a64 - mu[:, None]) / np.sqrt(var + eps)[:, None] * gamma.astype(np.float64) + beta.astype(np.float64)
    assert np.isfinite(yn).all() and np.isfinite(mean).all() and np.isfinite(rstd).all()
    assert np.abs(mean - mu).max() <= 4e-7 * np.abs(a64).max()
    assert np.abs(rstd * np.sqrt(var + eps) - 1.0).max() <= 2e-6
    err = np.abs(yn - ref).max()
    print("\n[K15 forward + block epilogue rows %d K %d] worst LayerNorm error %.2e" % (rows, K, err))
    assert err <= 4e-6 * max(1.0, np.abs(ref).max())


def test_block_epilogue_argument_checks(emu):
    x = _aligned((4, 8))
    args = (_ptr(x), 4, 8, 8, _ptr(x), _ptr(x), _ptr(x), _ptr(x), 1e-5)
    assert emu.mappo_linear512_forward_norm(*args, 2, _ptr(x), _ptr(x), _ptr(x), None, None) == -1     # rstd missing
    assert emu.mappo_linear512_forward_norm(*args, 1, _ptr(x), _ptr(x), _ptr(x), _ptr(x), None) != 0    # only relu


def test_transposed_planes_give_the_input_gradient(emu):
    """dX = dY W for a 512 -> 512 Linear: the forward kernel on the planes of W^T."""
    rng = np.random.default_rng(5)
    rows = 150
    w = (rng.standard_normal((512, 512)) * 0.1).astype(np.float32)          # y = x W^T, W [out 512, in 512]
    dy = _aligned((rows, 512))
    dy[:] = rng.standard_normal((rows, 512))
    planes = _planes(emu, w, 512, 512, 1)       # element (k = out feature of the forward, "feature" = input column)
    dx = _aligned((rows, 512))
    dx[:] = np.nan
    assert emu.mappo_linear512_forward(_ptr(dy), rows, 512, 512, _ptr(planes), None, _ptr(dx), None) == 0
    ref = dy.astype(np.float64) @ w.astype(np.float64)
    S = np.abs(dy.astype(np.float64)) @ np.abs(w.astype(np.float64))
    assert (np.abs(dx - ref) <= (16.0 + 512 / 6.0) * U * S).all()


@pytest.mark.parametrize("rows,K,ldx,cap", [(16 * 9 + 5, 40, 40, 0), (16 * 3, 1285, 1288, 0), (16 * 40 + 1, 512, 512, 2),
                                            (7, 130, 132, 0), (16 * 21, 128, 128, 1), (16 * 4 + 3, 1285, 1285, 0), (50, 21, 21, 0)])
def test_weight_gradient_vs_float64(emu, rows, K, ldx, cap):
    rng = np.random.default_rng(rows * 3 + K)
    x = _aligned((rows, ldx))
    x[:] = 0.0
    x[:, :K] = rng.standard_normal((rows, K)) * 1.5 + 0.3
    dy = _aligned((rows, 512))
    dy[:] = rng.standard_normal((rows, 512))
    dw = np.full((512, K), np.nan, np.float32)
    ws = np.full(emu.mappo_linear512_wgrad_workspace_floats(K), np.nan, np.float32)
    emu.mappo_mlp_set_grid_cap(cap)
    try:
        assert emu.mappo_linear512_wgrad(_ptr(dy), _ptr(x), rows, K, ldx, _ptr(dw), _ptr(ws), None) == 0
    finally:
        emu.mappo_mlp_set_grid_cap(0)
    d64, x64 = dy.astype(np.float64), x[:, :K].astype(np.float64)
    ref = d64.T @ x64
    S = np.abs(d64).T @ np.abs(x64)
    assert np.isfinite(dw).all()
    err = np.abs(dw - ref) / ((16.0 + rows / 6.0) * U * S + 1e-300)
    print("\n[K15 wgrad rows %d K %d] worst error = %.2f of the bound" % (rows, K, err.max()))
    assert err.max() <= 1.0


def test_argument_checks(emu):
    x = _aligned((4, 8))
    assert emu.mappo_linear512_forward(None, 4, 8, 8, None, None, None, None) == -1
    assert emu.mappo_linear512_forward(_ptr(x), 4, 8, 6, _ptr(x), None, _ptr(x), None) == -2       # ldx < K
    assert emu.mappo_linear512_wgrad(_ptr(x), _ptr(x), 0, 8, 8, _ptr(x), _ptr(x), None) == -2
