"""The fused GRU chunk kernels (K12: csrc/mappo_gru_impl.h -- both projections of every step on the MFMA, gates, mask
resets, output LayerNorm, truncated BPTT inside one launch) executed on the host SIMT emulator (tests/simt) and compared
with a float64 torch restatement of the reference's RNNLayer (onpolicy/algorithms/utils/rnn.py:7-80: nn.GRU cell with
the state multiplied by the mask before every step, LayerNorm on the outputs).  The same source is compiled for gfx950
into libmappo_hip.so; tests/test_gpu_gru_seq.py repeats the comparison on the device."""
import ctypes
import os

import numpy as np
import pytest
import torch

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ (host build of the emulator) not found")


@pytest.fixture(scope="module")
def emu_lib():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt"))
    import build
    from onpolicy import _native
    lib = ctypes.CDLL(build.build())
    for name in ("mappo_gru_seq_forward", "mappo_gru_seq_backward", "mappo_gru_seq_gates_floats",
                 "mappo_gru_seq_stats_floats", "mappo_gru_seq_workspace_floats", "mappo_mlp_set_grid_cap",
                 "mappo_mlp_set_flags", "mappo_gru_weight_grads", "mappo_gru_weight_grads_workspace_floats"):
        res, args = _native.SIGNATURES[name]
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


# Both arithmetic forms of the projections (field ``arith`` of mappo_gru_seq_t): the six-term bf16 form (weights as bf16
# planes in LDS; the backward with all six blocks of its transposed products as planes wherever that instance is built)
# and the float32 MFMA
@pytest.fixture(params=[1, 0], ids=["f32_mfma", "six_term"])
def emu(emu_lib, request):
    emu_lib._arith = request.param
    yield emu_lib
    emu_lib._arith = 0


def reference(p, x, h0, masks, L, mb):
    """float64: x [L * mb, 64], h0 [mb, 64], masks [L * mb] -> y [L * mb, 64], h_last (rnn.py:24-80 + PyTorch's GRU cell)."""
    h = h0
    outs = []
    for l in range(L):
        hm = h * masks[l * mb:(l + 1) * mb, None]
        gi = x[l * mb:(l + 1) * mb] @ p["w_ih"].t() + p["b_ih"]
        gh = hm @ p["w_hh"].t() + p["b_hh"]
        i_r, i_z, i_n = gi.chunk(3, -1)
        h_r, h_z, h_n = gh.chunk(3, -1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = n + z * (hm - n)
        outs.append(h)
    hs = torch.cat(outs, 0)
    y = torch.nn.functional.layer_norm(hs, (64,), p["ln_g"], p["ln_b"], 1e-5)
    return y, h


def run(emu, L, mb, seed, grid_cap=0, head_out=0):
    from onpolicy import _native
    rng = np.random.default_rng(seed)
    f32 = np.float32
    P = {"w_ih": rng.standard_normal((192, 64)) * 0.2, "w_hh": rng.standard_normal((192, 64)) * 0.2,
         "b_ih": rng.standard_normal(192) * 0.1, "b_hh": rng.standard_normal(192) * 0.1,
         "ln_g": 1.0 + 0.2 * rng.standard_normal(64), "ln_b": 0.1 * rng.standard_normal(64)}
    P = {k: v.astype(f32) for k, v in P.items()}
    x = rng.standard_normal((L * mb, 64)).astype(f32)
    h0 = rng.standard_normal((mb, 64)).astype(f32)
    masks = (rng.random(L * mb) > 0.2).astype(f32)
    dy = rng.standard_normal((L * mb, head_out or 64)).astype(f32)     # with a head: the gradient at its output
    hw = (rng.standard_normal((max(head_out, 1), 64)) * 0.3).astype(f32)
    hb = (rng.standard_normal(max(head_out, 1)) * 0.3).astype(f32)
    logits = np.full((L * mb, max(head_out, 1)), np.nan, f32)
    dhl = rng.standard_normal((mb, 64)).astype(f32)
    nan = lambda *shape: np.full(shape, np.nan, f32)
    y, h_last = nan(L * mb, 64), nan(mb, 64)
    gates = nan(emu.mappo_gru_seq_gates_floats(L, mb))
    stats = nan(emu.mappo_gru_seq_stats_floats(L, mb))
    hm = nan(L * mb, 64)
    dx, dgi, dq, dh0 = nan(L * mb, 64), nan(L * mb, 192), nan(L * mb, 64), nan(mb, 64)
    ln_grads, ws = nan(800), nan(emu.mappo_gru_seq_workspace_floats())
    ptr = lambda a: a.ctypes.data
    m = _native.GRUSeq(x=ptr(x), h0=ptr(h0), masks=ptr(masks), w_ih=ptr(P["w_ih"]), w_hh=ptr(P["w_hh"]), b_ih=ptr(P["b_ih"]),
                       b_hh=ptr(P["b_hh"]), ln_g=ptr(P["ln_g"]), ln_b=ptr(P["ln_b"]), ln_eps=1e-5, H=64, L=L, mb=mb,
                       arith=getattr(emu, "_arith", 0), y=ptr(y), h_last=ptr(h_last), gates=ptr(gates), hm=ptr(hm), stats=ptr(stats), dy=ptr(dy),
                       dx=ptr(dx), dgi=ptr(dgi), dq=ptr(dq), dh0=ptr(dh0), dh_last=ptr(dhl), ln_grads=ptr(ln_grads),
                       workspace=ptr(ws))
    if head_out:
        m.head_w, m.head_b, m.head_out, m.logits, m.dlogits, m.dy = ptr(hw), ptr(hb), head_out, ptr(logits), ptr(dy), None
        m.head_sums = int(head_out <= 6)
    emu.mappo_mlp_set_grid_cap(grid_cap)
    try:
        assert emu.mappo_gru_seq_forward(ctypes.byref(m), None) == 0
        assert emu.mappo_gru_seq_backward(ctypes.byref(m), None) == 0
    finally:
        emu.mappo_mlp_set_grid_cap(0)

    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    th0 = torch.tensor(h0, dtype=torch.float64, requires_grad=True)
    y_ref, h_ref = reference(tp, tx, th0, torch.tensor(masks, dtype=torch.float64), L, mb)
    np.testing.assert_allclose(y, y_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(h_last, h_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    out_ref = y_ref
    if head_out:        # the output Linear evaluated inside the launches
        thw = torch.tensor(hw, dtype=torch.float64, requires_grad=True)
        thb = torch.tensor(hb, dtype=torch.float64, requires_grad=True)
        out_ref = y_ref @ thw.t() + thb
        np.testing.assert_allclose(logits, out_ref.detach().numpy(), rtol=2e-4, atol=5e-5)
    ((out_ref * torch.tensor(dy, dtype=torch.float64)).sum() + (h_ref * torch.tensor(dhl, dtype=torch.float64)).sum()).backward()

    def close(got, ref, name):
        ref = ref.numpy() if torch.is_tensor(ref) else ref
        np.testing.assert_allclose(got, ref, rtol=3e-4, atol=3e-5 * max(1e-12, np.abs(ref).max()), err_msg=name)

    if 0 < head_out <= 6:       # the head's own gradients from the sums the backward launch leaves
        gh = ln_grads[384:384 + 64 * head_out].reshape(head_out, 64).astype(np.float64)
        dbh = ln_grads[768:768 + head_out].astype(np.float64)
        close(gh * P["ln_g"].astype(np.float64) + dbh[:, None] * P["ln_b"].astype(np.float64), thw.grad, "head weight")
        close(dbh, thb.grad, "head bias")
    close(dx, tx.grad, "dx")
    close(dh0, th0.grad, "dh0")
    close(ln_grads[:64], tp["ln_g"].grad, "ln weight")
    close(ln_grads[64:128], tp["ln_b"].grad, "ln bias")
    # the bias gradients come out of the launch (column sums of the gate gradients, folded over the rows in registers)
    close(ln_grads[128:320], tp["b_ih"].grad, "b_ih (in-kernel)")
    close(np.concatenate([ln_grads[128:256], ln_grads[320:384]]), tp["b_hh"].grad, "b_hh (in-kernel)")
    # what the caller forms from the gate gradients (onpolicy/algorithms/utils/rnn.py: _GRUChunkFn.backward)
    d64 = lambda a: a.astype(np.float64)
    close(d64(dgi).T @ d64(x), tp["w_ih"].grad, "w_ih")
    dgh = np.concatenate([d64(dgi)[:, :128], d64(dq)], 1)
    close(dgh.T @ d64(hm), tp["w_hh"].grad, "w_hh")
    close(d64(dgi).sum(0), tp["b_ih"].grad, "b_ih")
    close(dgh.sum(0), tp["b_hh"].grad, "b_hh")


@pytest.mark.parametrize("L,mb", [(1, 40), (3, 32), (5, 70), (10, 33)])
def test_chunk_kernels_vs_float64_reference(emu, L, mb):
    run(emu, L, mb, seed=L * 100 + mb)


@pytest.mark.parametrize("head_out,L,mb", [(1, 3, 40), (5, 4, 33), (6, 2, 32), (18, 3, 45), (2, 1, 70)])
def test_output_linear_inside_the_launches(emu, head_out, L, mb):
    """head_out > 0: logits = y head_w^T + head_b from the forward launch, dy = dlogits head_w formed by the backward launch
    on the MFMA (1, 3 or 9 k steps of two outputs; padded outputs meet zero weights)."""
    run(emu, L, mb, seed=head_out * 1000 + mb, head_out=head_out)


def test_waves_looping_over_several_tiles(emu):
    """Grid capped at one workgroup: every wave walks more than one 32-chunk tile (state, carry and the prefetched
    input are re-initialised per tile)."""
    run(emu, 4, 32 * 9 + 5, seed=7, grid_cap=1)


def weight_grads_case(lib, rows, seed, scale=None, device=None):
    """(dw from mappo_gru_weight_grads, float64 reference, sum of |terms|) for random gate gradients / inputs of `rows` rows; on
    the emulator (host arrays) or, with ``device``, through the product library on the GPU."""
    g = torch.Generator().manual_seed(seed)
    dgi = torch.randn(rows, 192, generator=g)
    dq = torch.randn(rows, 64, generator=g)
    x = torch.randn(rows, 64, generator=g)
    hm = torch.randn(rows, 64, generator=g) * (torch.rand(rows, 1, generator=g) > 0.1).float()
    if scale is not None:       # every row at its own magnitude
        s = 10.0 ** (scale * (torch.rand(rows, 1, generator=g) - 0.5))
        dgi, dq = dgi * s, dq * s
        x, hm = x / s.sqrt(), hm * s.sqrt()
    hid = torch.cat([dgi[:, :128], dq], 1).double()
    ref = torch.stack([dgi.double().t() @ x.double(), hid.t() @ hm.double()])
    mag = torch.stack([dgi.double().abs().t() @ x.double().abs(), hid.abs().t() @ hm.double().abs()])
    if device is None:
        a = [np.ascontiguousarray(t.numpy()) for t in (dgi, dq, x, hm)]
        dw = np.full((2, 192, 64), np.nan, np.float32)
        ws = np.full(lib.mappo_gru_weight_grads_workspace_floats(), np.nan, np.float32)
        rc = lib.mappo_gru_weight_grads(*[t.ctypes.data for t in a], rows, dw.ctypes.data, ws.ctypes.data, None)
        assert rc == 0
        return torch.from_numpy(dw).double(), ref, mag
    from onpolicy import _native
    a = [t.to(device).contiguous() for t in (dgi, dq, x, hm)]
    dw = torch.full((2, 192, 64), float("nan"), device=device)
    ws = torch.full((lib.mappo_gru_weight_grads_workspace_floats(),), float("nan"), device=device)
    _native.check(lib.mappo_gru_weight_grads(*[t.data_ptr() for t in a], rows, dw.data_ptr(), ws.data_ptr(),
                                             _native.stream_of(device)), "mappo_gru_weight_grads")
    return dw.cpu().double(), ref, mag


@pytest.mark.parametrize("rows,cap", [(1, 0), (16, 0), (16 * 7 + 5, 0), (16 * 40 + 3, 2), (16 * 9, 1), (16 * 64 + 15, 3)])
def test_weight_gradient_kernel_vs_float64(emu_lib, rows, cap):
    """mappo_gru_weight_grads: dW_ih = dgi^T x and dW_hh = [dgi_r | dgi_z | dq]^T hm in one launch, six-term arithmetic -- every
    output tile's owner wave, the slot ring over several tiles per workgroup (1, 2, 3 workgroups), ragged last tiles, the
    workgroups' partial sums; judged with the six-term bound against float64: |error| <= (16 + K / 6) 2^-24 sum |a||b|, K = rows."""
    emu_lib.mappo_mlp_set_grid_cap(cap)
    try:
        for scale in (None, 12.0):
            dw, ref, mag = weight_grads_case(emu_lib, rows, rows + (0 if scale is None else 1), scale)
            bound = (16 + rows / 6) * 2.0 ** -24 * mag + 1e-30
            worst = float(((dw - ref).abs() / bound).max())
            assert torch.isfinite(dw).all() and worst <= 1.0, (rows, scale, worst)
    finally:
        emu_lib.mappo_mlp_set_grid_cap(0)
