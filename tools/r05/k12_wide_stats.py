"""Statistics behind tests/test_gpu_six_term_adversarial.py::test_gru_chunk_kernels_on_wide_magnitudes: the max error / largest
entry of both arithmetic forms against float64 over many instances (seeds, weight spreads, CPU thread counts -- the instance of
the test changed with torch.set_num_threads, which other tests call: call 15)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from onpolicy.algorithms.utils.rnn import RNNLayer          # noqa: E402
from test_gru_kernels_emulated import reference             # noqa: E402


def instance(seed, decades, threads):
    torch.set_num_threads(threads)
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    layer = RNNLayer(64, 64, 1, True)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for name in ("weight_ih_l0", "weight_hh_l0"):
            w = getattr(layer.rnn, name)
            w.mul_(10.0 ** (decades * torch.rand(w.shape, generator=g) - decades / 2))
        for p in (layer.rnn.bias_ih_l0, layer.rnn.bias_hh_l0, layer.norm.weight, layer.norm.bias):
            p.add_(0.1 * torch.randn(p.shape, generator=g))
    layer = layer.to(dev)
    L, B = 10, 32 * 40 + 7
    x = torch.randn(L * B, 64, generator=g) * 10.0 ** (8 * torch.rand(L * B, 1, generator=g) - 6)
    h0 = torch.randn(B, 1, 64, generator=g)
    masks = (torch.rand(L * B, 1, generator=g) > 0.1).float()
    dy = torch.randn(L * B, 64, generator=g)
    P = {"w_ih": layer.rnn.weight_ih_l0, "w_hh": layer.rnn.weight_hh_l0, "b_ih": layer.rnn.bias_ih_l0,
         "b_hh": layer.rnn.bias_hh_l0, "ln_g": layer.norm.weight, "ln_b": layer.norm.bias}
    tp = {k: v.detach().cpu().double().requires_grad_() for k, v in P.items()}
    tx, th = x.double().requires_grad_(), h0[:, 0].double().requires_grad_()
    y_ref, h_ref = reference(tp, tx, th, masks[:, 0].double(), L, B)
    (y_ref * dy.double()).sum().backward()
    ref = {"y": y_ref.detach(), "h_last": h_ref.detach(), "dx": tx.grad, "dh0": th.grad}
    ref.update({k: tp[k].grad for k in P})
    err = {}
    for arith in ("six_term", "f32_mfma"):
        os.environ["MAPPO_MATRIX_ARITHMETIC"] = arith
        for p in layer.parameters():
            p.grad = None
        xd, hd = x.to(dev).requires_grad_(), h0.to(dev).requires_grad_()
        y, h_last = layer(xd, hd, masks.to(dev))
        (y * dy.to(dev)).sum().backward()
        got = {"y": y.detach(), "h_last": h_last[:, 0].detach(), "dx": xd.grad, "dh0": hd.grad[:, 0]}
        got.update({k: P[k].grad for k in P})
        err[arith] = {k: float((got[k].cpu().double() - ref[k]).abs().max() / ref[k].abs().max()) for k in ref}
    return err


out = {}
for decades in (4, 2, 1):
    rows = []
    for seed in range(16):
        for threads in (1, 8):
            e = instance(seed, decades, threads)
            worst = max(e["six_term"][k] / (e["f32_mfma"][k] + 1e-30) for k in e["six_term"] if e["f32_mfma"][k] > 1e-6)
            rows.append({"seed": seed, "threads": threads, "max_six": max(e["six_term"].values()),
                         "max_f32": max(e["f32_mfma"].values()), "worst_ratio": worst})
    out[decades] = rows
    print("decades", decades, "max six %.3g  max f32 %.3g  worst ratio %.2f  median ratio %.2f" % (
        max(r["max_six"] for r in rows), max(r["max_f32"] for r in rows), max(r["worst_ratio"] for r in rows),
        sorted(r["worst_ratio"] for r in rows)[len(rows) // 2]))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r05"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05", "k12_wide_stats.json"), "w"), indent=1)
