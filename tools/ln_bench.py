"""Times the K6 act+LayerNorm kernels at the north-star span size (M = 2.62 M rows, D = 64)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "on-policy_amd"))
import torch
from onpolicy.algorithms.utils.fused_norm import _LayerNormFn, ACT_TANH
dev = torch.device("cuda", 0)
M, D = 2621440, 64
xs = [torch.randn(M, D, device=dev, requires_grad=True) for _ in range(3)]
w = torch.ones(D, device=dev, requires_grad=True); b = torch.zeros(D, device=dev, requires_grad=True)
pre = torch.zeros(D, device=dev, requires_grad=True)
dy = [torch.randn(M, D, device=dev) for _ in range(3)]
def run(n):
    tf = tb = 0.0
    for i in range(n):
        x = xs[i % 3]
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); y = _LayerNormFn.apply(x, w, b, 1e-5, ACT_TANH, pre); e1.record()
        y.backward(dy[i % 3]); e2.record(); torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    return tf / n * 1e3, tb / n * 1e3
run(3)
f, bwd = run(12)
print("act+LN fwd %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s)" % (f, 2 * M * D * 4 / f / 1e6, bwd, 3 * M * D * 4 / bwd / 1e6))
