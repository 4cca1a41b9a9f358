#!/usr/bin/env python
"""GPU-box probe: what the library GEMM makes of the first-layer weight gradient dW1 = dZ1^T X ([64, rows] x [rows, din]) when the
row table is the identity (one minibatch of all rows) -- the shape mlp_dw1_direct_kernel serves at 0.65-0.69 of the f32 matrix
pipe.  Prints ms and TFLOP/s per form.

    python tools/probe_dw1_gemm.py [--rows 13107200] [--din 384]
"""
import argparse
import json

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=13107200)
    ap.add_argument("--din", type=int, default=384)
    ap.add_argument("--reps", type=int, default=5)
    opt = ap.parse_args()
    dev = torch.device("cuda", 0)
    x = torch.randn(opt.rows, opt.din, device=dev)
    dz = torch.randn(opt.rows, 64, device=dev)
    out = {"rows": opt.rows, "din": opt.din, "flop": 2 * opt.rows * opt.din * 64, "forms": {}}
    forms = {"dz.T @ x": lambda: torch.mm(dz.t(), x), "(x.T @ dz)": lambda: torch.mm(x.t(), dz)}
    for name, fn in forms.items():
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(opt.reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / opt.reps
        out["forms"][name] = {"ms": round(ms, 3), "tflops": round(out["flop"] / ms / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
