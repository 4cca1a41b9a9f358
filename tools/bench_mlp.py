#!/usr/bin/env python
"""GPU-box microbenchmark of the fused hidden-64 trunk kernels (K9) at the north-star shapes: forward and backward
launch times with HIP events on the launch stream, TFLOP/s against the dense f32 MFMA peak (157.3 TFLOP/s,
MI355X_MICROARCH.md) and the HBM bytes each launch has to move.

    python tools/bench_mlp.py [--rows 2621440] [--din 384 48] [--reps 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

PEAK_TF = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2621440)
    ap.add_argument("--din", type=int, nargs="+", default=[384, 48])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", type=int, default=None)
    ap.add_argument("--grid-cap", type=int, default=0, help="upper bound on the persistent kernels' workgroups (tuning hook)")
    ap.add_argument("--sequential", action="store_true", help="rows in memory order instead of a random permutation")
    ap.add_argument("--stamps", action="store_true", help="print shader-clock stamps of workgroup 0 (forward kernel)")
    opt = ap.parse_args()
    from onpolicy.algorithms.utils import fused_mlp
    from onpolicy.algorithms.utils.mlp import MLPBase
    from helpers import make_args
    dev = torch.device("cuda", 0)
    if opt.grid_cap:
        from onpolicy import _native
        _native.lib().mappo_mlp_set_grid_cap(opt.grid_cap)
    res = []
    for din in opt.din:
        out = opt.out if opt.out is not None else (1 if din > 100 else 5)
        args = make_args(hidden_size=64, layer_N=1, use_ReLU=False)
        torch.manual_seed(0)
        base = MLPBase(args, (din,)).to(dev)
        head = torch.nn.Linear(64, out).to(dev)
        src_rows = opt.rows + 4096
        src = torch.randn(src_rows, din, device=dev)
        xhat = fused_mlp.standardize_rows(src)
        idx = torch.randperm(src_rows, device=dev)[:opt.rows]
        if opt.sequential:
            idx = torch.arange(opt.rows, device=dev)
        rs = fused_mlp.RowSource(xhat, idx, standardized=True)
        dy = torch.randn(opt.rows, out, device=dev)

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(opt.reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return sorted(ts)[len(ts) // 2]

        holder = {}

        def fwd():
            holder["y"] = fused_mlp.trunk_forward(base, rs, head)

        def bwd():
            for p in list(base.parameters()) + list(head.parameters()):
                p.grad = None
            holder["y"].backward(dy, retain_graph=True)

        if opt.stamps:
            from onpolicy import _native
            dbg = torch.zeros(2048, dtype=torch.int64, device=dev)
            with torch.no_grad():
                fused_mlp.trunk_forward(base, rs, head)
                _native.lib().mappo_mlp_set_debug(dbg.data_ptr())
                fused_mlp.trunk_forward(base, rs, head)
                torch.cuda.synchronize()
                _native.lib().mappo_mlp_set_debug(None)
            d = dbg.cpu().numpy()
            flags = int(os.environ.get("MAPPO_MLP_FLAGS", "0"))
            if not (flags & 4) and din % 4 == 0 and din <= 448:     # version 3 ran
                print("forward v3, waves 0 and 4 of workgroup 0 (they share SIMD 0): tile | chunk loop | tail | start -> "
                      "next start")
                for w in range(2):
                    t = d[64 * w:64 * w + 60].reshape(15, 4)
                    for m in range(1, 12):
                        print("wave", 4 * w, "tile", m, t[m][1] - t[m][0], t[m][2] - t[m][1], t[m + 1][0] - t[m][0],
                              "| start", t[m][0] - d[4])
                d = np.zeros_like(d)
            comp = d[:240].reshape(60, 4)
            load = d[256:256 + 240 + 16]
            print("compute: j, wait_at_barrier, mfma, tail(to next iteration start)")
            for j in range(24):
                nxt = comp[j + 1][0] if j + 1 < 60 else 0
                print(j, comp[j][1] - comp[j][0], comp[j][2] - comp[j][1], nxt - comp[j][2])
            print("loader (every 3 iterations): after barrier -> store done, -> issue done, -> next barrier passed")
            for j in range(0, 24, 3):
                b = load[4 * j:4 * j + 12]
                print(j, "data wait", b[3] - b[0], "lds", b[1] - b[3], "issue", b[2] - b[1], "barrier", b[4] - b[2], "|",
                      b[6] - b[4], b[8] - b[6], "|", b[10] - b[8])
            if True:
                fwd()
                dbg.zero_()
                _native.lib().mappo_mlp_set_debug(dbg.data_ptr())
                bwd()
                torch.cuda.synchronize()
                _native.lib().mappo_mlp_set_debug(None)
                dall = dbg.cpu().numpy()
                d = dall[512:512 + 160].reshape(40, 4)
                if din % 4 == 0 and din >= 192:
                    print("dw1 (wave 0 of workgroup 0): tile, wait for loads, barrier, issue, mfma steps")
                    for m in range(4, 12):
                        print(m, d[m][1] - d[m][0], d[m][2] - d[m][1], d[m][3] - d[m][2], d[m + 1][0] - d[m][3])
                b = dall[1024:1024 + 192].reshape(12, 16)
                if not (int(os.environ.get("MAPPO_MLP_FLAGS", "0")) & 4):
                    print("chain v2 (wave 0): tile | wait prefetch + issue loads | head | ln bwd | put dz + dX mfma | A regs |"
                          " put nhat' | G mfma | copies | ln bwd 0 | store | total")
                    ks = [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 15]
                    for t in range(1, 10):
                        r = b[t]
                        print(t, " ".join("%6d" % (r[ks[i + 1]] - r[ks[i]]) for i in range(len(ks) - 1)), "|", r[15] - r[0],
                              "| gap to next", b[t + 1][0] - r[15])
                    continue
                print("chain kernel (thread 0): tile | inputs+fetch | head | L1: ln+dz, sums, transposes+bias, dW+dX | "
                      "L0: ln+dz, sums | rest | total")
                for t in range(2, 10):
                    r = b[t]
                    print(t, r[1] - r[0], r[2] - r[1], "|", r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], "|",
                          r[7] - r[6], r[8] - r[7], "|", r[15] - r[8], "|", r[15] - r[0])
        t_stats = timed(lambda: fused_mlp.standardize_rows(src))
        t_f = timed(fwd)
        t_b = timed(bwd)
        R = opt.rows
        f_fwd = 2.0 * R * (din * 64 + 64 * 64 + 64 * out)
        f_bwd = 2.0 * R * (din * 64 + 2 * 64 * 64 + 2 * 64 * out)
        b_fwd = R * (4 * din + 8 + 8 + 2 * 256 + 4 * out)
        b_bwd = R * (4 * din + 8 + 8 + 2 * 256 + 4 * out + 2 * 256 * (1 + max(1, -(-din // 384))))
        rec = {"din": din, "out": out, "rows": R, "grid_cap": opt.grid_cap, "sequential": opt.sequential, "standardize_ms": round(t_stats, 3),
               "fwd_ms": round(t_f, 3), "fwd_tflops": round(f_fwd / t_f / 1e9, 1),
               "fwd_frac_mfma": round(f_fwd / t_f / 1e9 / PEAK_TF, 3), "fwd_gbs": round(b_fwd / t_f / 1e6, 1),
               "bwd_ms": round(t_b, 3), "bwd_tflops": round(f_bwd / t_b / 1e9, 1),
               "bwd_frac_mfma": round(f_bwd / t_b / 1e9 / PEAK_TF, 3), "bwd_gbs": round(b_bwd / t_b / 1e6, 1)}
        print(json.dumps(rec), flush=True)
        res.append(rec)
    return res


if __name__ == "__main__":
    main()
