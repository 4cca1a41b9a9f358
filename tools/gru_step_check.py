"""Fused GRU step (MFMA) vs the cell kernels + library GEMM, and timing of both, on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "on-policy_amd"))
import torch
from onpolicy.algorithms.utils import rnn as R
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def run(L, B, fused):
    R._FUSED_STEP = bool(fused)
    H = 64
    g = torch.Generator(device="cpu").manual_seed(1)
    gi = torch.randn(L, B, 3 * H, generator=g).to(dev).requires_grad_(True)
    h0 = torch.randn(B, H, generator=g).to(dev).requires_grad_(True)
    masks = (torch.rand(L, B, 1, generator=g) > 0.1).float().to(dev)
    w = (torch.randn(3 * H, H, generator=g) * 0.2).to(dev).requires_grad_(True)
    bi = (torch.randn(3 * H, generator=g) * 0.3).to(dev).requires_grad_(True)
    bh = (torch.randn(3 * H, generator=g) * 0.3).to(dev).requires_grad_(True)
    dout = torch.randn(L, B, H, generator=g).to(dev)
    out = R._GRUSequenceFn.apply(gi, h0, masks, w, bi, bh)
    out.backward(dout)
    return [t.detach().clone() for t in (out, gi.grad, h0.grad, w.grad, bi.grad, bh.grad)]
for L, B in ((3, 100), (10, 70001)):
    a = run(L, B, False); b = run(L, B, True)
    for name, x, y in zip(("out", "dgi", "dh0", "dW", "db_ih", "db_hh"), a, b):
        err = float((x - y).abs().max()); scale = float(x.abs().max())
        print("L=%d B=%d %-6s max|diff| %.3e (scale %.3e)" % (L, B, name, err, scale))
        assert err <= 2e-5 * max(scale, 1.0) + 1e-6, name
# timing at the ns_rnn span size
L, B, H = 10, 262144, 64
for fused in (False, True):
    R._FUSED_STEP = bool(fused)
    gi = torch.randn(L, B, 3 * H, device=dev, requires_grad=True); h0 = torch.randn(B, H, device=dev)
    masks = torch.ones(L, B, 1, device=dev); w = torch.randn(3 * H, H, device=dev, requires_grad=True) * 0.1
    w = w.detach().requires_grad_(True)
    bi = torch.zeros(3 * H, device=dev, requires_grad=True); bh = torch.zeros(3 * H, device=dev, requires_grad=True)
    dout = torch.randn(L, B, H, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = R._GRUSequenceFn.apply(gi, h0, masks, w, bi, bh)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.backward(dout)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("fused_step=%s  forward %.2f ms  backward %.2f ms (L=%d, B=%d)" % (fused, (t1 - t0) * 1e3, (t2 - t1) * 1e3, L, B))
