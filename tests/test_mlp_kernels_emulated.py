"""The fused hidden-64 trunk kernels (K9: csrc/mappo_mlp_impl.h -- gather + standardise + MFMA layer chain forward,
backward chain, first-layer weight gradient) executed on the host SIMT emulator (tests/simt) and compared with the
float64 torch restatement of the reference's modules (tests/mlp_reference.py).  This checks what cannot be seen without
running the code: MFMA fragment layouts, the slot <-> feature permutation between layers, LDS staging between barriers,
index mapping of both samplers, tile tails.  The same source is compiled for gfx950 into libmappo_hip.so; the -m gpu
tests (tests/test_gpu_mlp.py) repeat these comparisons on the device."""
import ctypes
import os
import shutil

import numpy as np
import pytest
import torch

import mlp_reference as R

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ (host build of the emulator) not found")


@pytest.fixture(scope="module")
def emu_lib():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt"))
    import build
    return R.bind(ctypes.CDLL(build.build()))


# Every test runs under seven (tuning flags, arithmetic) settings (``arith`` = the per-call field of mappo_mlp_t):
#   fwd3 / fwd_loaders / dw1_two_per_cu -- float32 MFMA: version 3 of the forward (operands straight from global memory,
#     resident first-layer weights; every aligned width up to 448 with two or three layers; a row's last chunk runs only the
#     groups of 8 columns that hold data), tuning bit 4 = the loader / compute kernel that serves every other shape, tuning bit
#     32 = the two-slot form of the direct-to-LDS first-layer weight-gradient kernel (two workgroups per CU);
#   six_term -- the shipped default: version 4 of the forward (both layers as six bf16 x bf16 terms per float32 product on
#     the bf16 matrix pipe) for two-layer trunks with aligned inputs 128 .. 384 floats wide, version 3 with its hidden layer
#     in six-term form for the other two-layer shapes, the backward chain of two-layer trunks and the direct first-layer
#     weight-gradient kernel in the same form -- for aligned inputs of at most 64 columns that gradient is accumulated inside
#     the chain's launch (DW1); every other shape falls through to the float32 kernels;
#   six_term_fwd4_every_width -- the same with tuning bit 128: version 4 also for inputs narrower than 128 floats;
#   six_term_separate_dw1 -- the same with tuning bit 256: narrow inputs keep the separate first-layer weight-gradient kernel;
#   six_term_dw1_one_per_cu -- the same with tuning bit 32: the six-term direct first-layer weight-gradient kernel in its
#     four-slot form, one workgroup per CU (the default is two slots per wave, two workgroups per CU).
@pytest.fixture(params=[(0, 1), (4, 1), (32, 1), (0, 0), (128, 0), (256, 0), (32, 0)],
                ids=["fwd3", "fwd_loaders", "dw1_two_per_cu", "six_term", "six_term_fwd4_every_width", "six_term_separate_dw1",
                     "six_term_dw1_one_per_cu"])
def emu(emu_lib, request):
    flags, arith = request.param
    old = emu_lib.mappo_mlp_set_flags(flags)
    assert old >= 0
    emu_lib._arith = arith
    yield emu_lib
    emu_lib._arith = 0
    emu_lib.mappo_mlp_set_flags(old)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _run(emu, rng, din, n_layers, act, out, rows, src_rows, standardize=True, chunk=None, backward=True):
    src = (rng.standard_normal((src_rows, din)) * 1.5 + 0.7).astype(np.float32)
    p = R.random_net(rng, din, n_layers, out)
    kw = dict(chunk_len=0, mb=0, T=0, N=0, A=0)
    if chunk is None:
        idx = rng.permutation(src_rows)[:rows].astype(np.int64)
    else:
        L, T, N, A = chunk
        assert src_rows == T * N * A and rows % L == 0
        mb = rows // L
        idx = rng.permutation(src_rows // L)[:mb].astype(np.int64)
        kw = dict(chunk_len=L, mb=mb, T=T, N=N, A=A)
    xin = src
    if standardize:      # the kernels read a standardised copy of the source matrix (made once per train())
        xin = np.full_like(src, np.nan)
        assert emu.mappo_standardize_rows(_ptr(src), src_rows, din, 1e-5, _ptr(xin), None) == 0
        np.testing.assert_allclose(xin, R.standardize_ref(src, 1e-5).numpy(), rtol=2e-5, atol=2e-6)
    y = np.full((rows, out if out else 64), np.nan, np.float32)
    z = [np.full((emu.mappo_mlp_row_table_ints(rows), 64), np.nan, np.float32) for _ in range(n_layers)]
    tab = np.full(emu.mappo_mlp_row_table_ints(rows), -1, np.int32)
    R128 = (rows + 127) // 128 * 128
    assert tab.size == R128
    assert emu.mappo_mlp_row_table(_ptr(idx), rows, kw["mb"], kw["chunk_len"], kw["T"], kw["N"], kw["A"], _ptr(tab),
                                   None) == 0
    srows = R.source_rows(idx, rows, **kw)
    np.testing.assert_array_equal(tab[:rows], srows)
    np.testing.assert_array_equal(tab[rows:], np.full(R128 - rows, srows[-1]))     # padding = last row
    m = R.MLP(src=_ptr(xin), row_tab=_ptr(tab), rows=rows, din=din,
              n_layers=n_layers, act=act, out=out, ln_eps=1e-5, arith=getattr(emu, "_arith", 0), w1=_ptr(p["w1"]), wh=_ptr(p["wh"]) if out else None,
              bh=_ptr(p["bh"]) if out else None, y=_ptr(y))
    st = [np.full((emu.mappo_mlp_row_table_ints(rows), 2), np.nan, np.float32) for _ in range(n_layers)]
    for l in range(n_layers):
        m.bias[l], m.ln_g[l], m.ln_b[l] = _ptr(p["bias%d" % l]), _ptr(p["ln_g%d" % l]), _ptr(p["ln_b%d" % l])
        m.z[l], m.ln_stats[l] = _ptr(z[l]), _ptr(st[l])
        if l > 0:
            m.w2[l - 1] = _ptr(p["w2_%d" % (l - 1)])
    assert emu.mappo_mlp_forward(ctypes.byref(m), None) == 0
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    y_ref, z_ref = R.forward_ref(tp, src, srows, standardize, n_layers, act, out)
    np.testing.assert_allclose(y, y_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    fn = {0: lambda v: v, 1: torch.tanh, 2: torch.relu}[act]
    for l in range(n_layers):       # saved for the backward: normalised activation and its row statistics
        a_ref = fn(z_ref[l].detach())
        mean, var = a_ref.mean(1, keepdim=True), a_ref.var(1, unbiased=False, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        np.testing.assert_allclose(R.rows_of_fragments(z[l], rows), ((a_ref - mean) * rstd).numpy(), rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(st[l][:rows], torch.cat([mean, rstd], 1).numpy(), rtol=2e-4, atol=2e-5)
    if not backward:
        return
    dy = rng.standard_normal(y.shape).astype(np.float32)
    n_g = emu.mappo_mlp_grad_floats(din, n_layers, out)
    grads = np.full(n_g, np.nan, np.float32)
    ws = np.full(emu.mappo_mlp_workspace_floats(din, n_layers, out), np.nan, np.float32)
    dz1 = np.full((emu.mappo_mlp_row_table_ints(rows), 64), np.nan, np.float32)     # padded to the 128-row tile
    m.dy, m.dz1, m.workspace, m.grads = _ptr(dy), _ptr(dz1), _ptr(ws), _ptr(grads)
    assert emu.mappo_mlp_backward(ctypes.byref(m), None) == 0
    (y_ref * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    g_ref = R.flat_grads({k: v.grad for k, v in tp.items()}, din, n_layers, out).numpy()
    assert g_ref.shape == grads.shape
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(grads, g_ref, rtol=2e-4, atol=2e-5 * scale)


# (din, n_layers, act, out, rows, src_rows): tails in every dimension -- rows not a multiple of the 128-row tile / the
# 32-row wave tile, din not a multiple of the 32-wide chunk / of 4, several workgroups per launch
CASES = [
    (48, 2, 1, 5, 70, 200),        # north-star actor shapes (tanh, Discrete(5))
    (30, 2, 2, 1, 128, 128),       # cfg2 actor width, ReLU, value head, exactly one tile
    (19, 1, 1, 3, 37, 64),         # single layer, odd width (unaligned rows)
    (130, 3, 1, 0, 150, 300),      # layer_N = 2, trunk only (features for the GRU), three chunks with a tail
    (70, 2, 1, 2, 128 * 5 + 9, 900),   # several tiles and two chunks: the loaders' in-flight chunks cross tile boundaries
    (388, 1, 2, 1, 45, 64),        # din just above 384: one 512-column slab, waves with four and with three k tiles
    (436, 2, 1, 1, 45, 64),        # SMAC's critic input width (padded): one 512-column slab
    (152, 2, 1, 1, 16 * 11 + 3, 300),  # config 2's critic input (padded): five k tiles, every wave owns all of them
    (192, 1, 2, 2, 16 * 9, 200),       # six k tiles in the row-split kernel
    (96, 2, 1, 4, 16 * 13 + 7, 256),   # three
    (768, 1, 2, 1, 40, 64),        # two full 384-column slabs in the first-layer weight gradient
    (900, 1, 1, 2, 40, 64),        # two 512-column slabs, the second partly filled
    (28, 1, 2, 2, 16 * 9 + 1, 300),   # one k tile (row-split weight-gradient kernel), a tile with a single live row
    (200, 2, 1, 1, 16 * 7 + 5, 200),  # direct-to-LDS weight-gradient kernel: waves with two and with one k tile
    (40, 2, 2, 7, 32 * 9 + 3, 400),   # a head wider than the backward chain keeps in registers (sums through LDS), odd width
    (64, 1, 1, 18, 100, 128),         # SMAC-sized action head on a single layer
    (36, 3, 1, 6, 32 * 5, 200),       # three layers with a head
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_rows_mode_vs_float64_reference(emu, case):
    din, L, act, out, rows, src_rows = case
    _run(emu, np.random.default_rng(din * 7 + rows), din, L, act, out, rows, src_rows)


@pytest.mark.parametrize("cap", [1, 2])
@pytest.mark.parametrize("din,tiles", [(48, 7), (48, 6), (48, 5), (100, 2), (100, 3), (130, 3), (200, 4)])
def test_workgroups_looping_over_many_tiles(emu, cap, din, tiles):
    """Persistent loops: with the grid capped at 1 or 2 workgroups a workgroup processes several tiles, so the loaders'
    in-flight chunk loads cross tile boundaries and the pipelines run through their steady state and every tail length
    (number of pipeline iterations modulo the prefetch depth of 3)."""
    emu.mappo_mlp_set_grid_cap(cap)
    try:
        rows = 128 * tiles - 37
        _run(emu, np.random.default_rng(din + tiles), din, 2, 1, 3, rows, rows + 50)
    finally:
        emu.mappo_mlp_set_grid_cap(0)


@pytest.mark.parametrize("D", [30, 19, 150, 600])
def test_standardised_copy_padded_to_16_bytes(emu, D):
    """mappo_standardize_rows_ld: the rows of the copy are ld = D rounded up to 4 floats apart, the padding is zero, the
    data columns equal the unpadded call bit for bit."""
    rng = np.random.default_rng(D)
    rows, ld = 37, (D + 3) // 4 * 4
    src = (rng.standard_normal((rows, D)) * 2 + 0.5).astype(np.float32)
    plain = np.full((rows, D), np.nan, np.float32)
    padded = np.full((rows, ld), np.nan, np.float32)
    assert emu.mappo_standardize_rows(_ptr(src), rows, D, 1e-5, _ptr(plain), None) == 0
    assert emu.mappo_standardize_rows_ld(_ptr(src), rows, D, 1e-5, _ptr(padded), ld, None) == 0
    np.testing.assert_array_equal(padded[:, :D], plain)
    np.testing.assert_array_equal(padded[:, D:], 0.0)
    np.testing.assert_allclose(plain, R.standardize_ref(torch.tensor(src, dtype=torch.float64), 1e-5).numpy(), rtol=2e-5, atol=2e-6)


def test_unstandardised_input_and_identity_rows(emu):
    rng = np.random.default_rng(3)
    _run(emu, rng, 40, 2, 1, 4, 45, 45, standardize=False)


def test_chunk_mode_rows(emu):
    """recurrent_generator's rows: row l * mb + j <- element idx[j] * L + l of the (n, a, t)-ordered sequence, with
    T % L != 0 (chunks that straddle trajectories)."""
    T, N, A, L = 7, 3, 2, 3
    _run(emu, np.random.default_rng(5), 24, 2, 1, 0, L * 10, T * N * A, chunk=(L, T, N, A))


# widths of the version-4 forward: 64-column chunks in a ring of three (chunk counts 1 .. 6: every rotation of the ring),
# last chunks of 1 .. 4 k = 16 steps, the 384-wide critic input that fills the LDS with weight planes
@pytest.mark.parametrize("din,act,out", [(384, 1, 1), (128, 2, 5), (64, 1, 2), (256, 1, 0), (320, 2, 3), (20, 1, 5),
                                         (56, 1, 1), (212, 1, 4)])
def test_version4_widths(emu, din, act, out):
    emu.mappo_mlp_set_grid_cap(1)
    try:
        _run(emu, np.random.default_rng(din), din, 2, act, out, 128 * 2 + 45, 400)
    finally:
        emu.mappo_mlp_set_grid_cap(0)


@pytest.mark.parametrize("din,act,out", [(384, 1, 1), (48, 1, 5), (152, 2, 0), (20, 1, 3)])
def test_version4_hidden_layer_in_six_term_form(emu_lib, din, act, out):
    """Version 4 of the forward with its hidden layer on the bf16 matrix pipe too (weight planes in registers, folded bias
    from LDS): what MAPPO_ARITH_SIX_TERM selects (tuning bit 128: every aligned width).  Device-verified in round 5."""
    old = emu_lib.mappo_mlp_set_flags(128)
    emu_lib.mappo_mlp_set_grid_cap(1)
    emu_lib._arith = 0
    try:
        _run(emu_lib, np.random.default_rng(din + 1), din, 2, act, out, 128 * 2 + 45, 400)
    finally:
        emu_lib.mappo_mlp_set_grid_cap(0)
        emu_lib.mappo_mlp_set_flags(old)


@pytest.mark.parametrize("din,act,out", [(48, 1, 5), (384, 1, 1), (152, 2, 0), (20, 1, 3), (436, 1, 1)])
def test_version3_hidden_layer_in_six_term_form(emu_lib, din, act, out):
    """The version-3 forward (two waves per SIMD; the narrow actor inputs, widths above 384) with its hidden layer on the bf16
    matrix pipe (weight planes in LDS): what MAPPO_ARITH_SIX_TERM selects for two-layer trunks version 4 does not take (tuning
    bit 4 would keep the loader / compute kernel; 384 and 152 go to version 4 here).  Device-verified in round 5."""
    emu_lib.mappo_mlp_set_grid_cap(1)
    emu_lib._arith = 0
    try:
        _run(emu_lib, np.random.default_rng(din + 2), din, 2, act, out, 128 * 2 + 45, 400)
    finally:
        emu_lib.mappo_mlp_set_grid_cap(0)


@pytest.mark.parametrize("flags", [0, 256], ids=["fused", "separate_kernel"])
@pytest.mark.parametrize("din,act,out", [(20, 1, 1), (32, 2, 0), (64, 1, 5), (4, 1, 2), (36, 2, 1), (48, 1, 12), (60, 0, 0)])
def test_first_layer_weight_gradient_inside_the_chain(emu_lib, flags, din, act, out):
    """Round 5 (the bytes diet that fits): six-term two-layer trunks with aligned inputs of at most 64 columns
    accumulate dW1 inside the chain's launch (xhat tiles through the row table by direct-to-LDS loads, dz1 never written);
    tuning bit 256 keeps the separate kernel.  One and two k tiles, every head form (value head in registers, narrow heads in
    registers, wide heads through LDS, no head), gathered rows, several tiles per wave and two workgroups with a ragged end."""
    old = emu_lib.mappo_mlp_set_flags(flags)
    emu_lib._arith = 0
    try:
        for cap, rows in ((1, 32 * 9 + 5), (2, 128 * 3 + 33)):
            emu_lib.mappo_mlp_set_grid_cap(cap)
            _run(emu_lib, np.random.default_rng(din * 3 + out + rows), din, 2, act, out, rows, rows + 77)
    finally:
        emu_lib.mappo_mlp_set_grid_cap(0)
        emu_lib.mappo_mlp_set_flags(old)


def test_tuning_flags_reject_unknown_bits(emu_lib):
    """mappo_mlp_set_flags knows bits 1, 2, 4, 8, 32, 128, 256; anything else (e.g. the arithmetic bits of ABI version 1) is refused
    and changes nothing.  An unknown ``arith`` value is an argument error."""
    old = emu_lib.mappo_mlp_set_flags(4)
    try:
        for bad in (16, 64, 512, 1024, 2048, 4096, 8192, 4 | 64):
            assert emu_lib.mappo_mlp_set_flags(bad) == -1
            assert emu_lib.mappo_mlp_set_flags(4) == 4
    finally:
        emu_lib.mappo_mlp_set_flags(old)
    m = R.MLP(arith=7)
    assert emu_lib.mappo_mlp_forward(ctypes.byref(m), None) != 0


def test_first_layer_slab_split(emu):
    """Widths above 384 split the weight-gradient kernel over k slabs (blockIdx.y); critic width of the north star."""
    _run(emu, np.random.default_rng(11), 435, 2, 1, 1, 40, 50)


def test_three_way_bf16_split_is_exact(emu_lib):
    """The split the six-term kernels apply to every operand (csrc/mappo_mlp_impl.h: split3): x = p1 + p2 + p3 EXACTLY for
    normal float32 values (8 + 8 + 8 significand bits), every part a bf16 value, |p2| <= 2^-8 |p1|, |p3| <= 2^-16 |p1| --
    so the six kept products miss less than 2^-24 of x y."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(4096) * 10.0 ** rng.uniform(-20, 20, 4096),
                        [0.0, -0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 16777215.0]]).astype(np.float32)
    x = np.resize(x, (x.size + 7) // 8 * 8)
    p = [np.full(x.shape, np.nan, np.float32) for _ in range(3)]
    emu_lib.simt_split3.restype = None
    emu_lib.simt_split3(ctypes.c_void_p(x.ctypes.data), ctypes.c_longlong(x.size), *[ctypes.c_void_p(q.ctypes.data) for q in p])
    total = p[0].astype(np.float64) + p[1].astype(np.float64) + p[2].astype(np.float64)
    np.testing.assert_array_equal(total, x.astype(np.float64))
    for q in p:         # bf16 values: the low 16 bits of the float32 pattern are zero
        assert not (q.view(np.uint32) & 0xFFFF).any()
    nz = p[0] != 0
    assert (np.abs(p[1][nz]) <= np.abs(p[0][nz]) * 2.0 ** -8).all() and (np.abs(p[2][nz]) <= np.abs(p[0][nz]) * 2.0 ** -16).all()
