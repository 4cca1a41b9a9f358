#!/usr/bin/env python
"""GPU box: K15 (mappo_linear512_*) alone -- forward, input gradient and weight gradient at one Hanabi row span (683 k rows,
K = 512 / 1288 / 1285) against the library's float32 GEMMs, event-timed.  Also the command tools/pmc_lin512.sh profiles.

    python tools/bench_lin512.py [--rows 682667] [--reps 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "on-policy_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=682667)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-library", action="store_true")
    opt = ap.parse_args()
    from onpolicy import _native
    lib, p = _native.lib(), _native.ptr
    dev = torch.device("cuda", 0)
    stream = _native.stream_of(dev)
    rows = opt.rows
    out = []
    for K, ldx in ((512, 512), (1288, 1288), (1285, 1285)):
        x = torch.randn(rows, ldx, device=dev)
        w = torch.randn(512, K, device=dev) * 0.05
        dy = torch.randn(rows, 512, device=dev)
        planes = torch.empty(lib.mappo_linear512_planes_floats(K), device=dev)
        y = torch.empty(rows, 512, device=dev)
        dw = torch.empty(512, K, device=dev)
        ws = torch.empty(lib.mappo_linear512_wgrad_workspace_floats(K), device=dev)
        _native.check(lib.mappo_linear512_prepare(p(w), K, K, 0, p(planes), stream), "prepare")
        fwd = lambda: _native.check(lib.mappo_linear512_forward(p(x), rows, K, ldx, p(planes), None, p(y), stream), "fwd")
        wg = lambda: _native.check(lib.mappo_linear512_wgrad(p(dy), p(x), rows, K, ldx, p(dw), p(ws), stream), "wgrad")
        flop = 2.0 * rows * K * 512
        rec = {"rows": rows, "K": K, "ldx": ldx}
        for name, fn in (("k15_forward", fwd), ("k15_wgrad", wg)):
            ms = timed(fn, opt.reps)
            rec[name] = {"ms": round(ms, 4), "tflops_f32_equivalent": round(flop / ms / 1e9, 1)}
        if not opt.no_library:
            xk = x[:, :K] if ldx != K else x
            for name, fn in (("library_forward", lambda: torch.nn.functional.linear(xk, w)),
                             ("library_wgrad", lambda: dy.t() @ xk)):
                ms = timed(fn, opt.reps)
                rec[name] = {"ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)}
        out.append(rec)
        del x, w, dy, y
    print(json.dumps({"what": "K15 against the library's float32 GEMMs at one Hanabi row span (tools/bench_lin512.py)", "runs": out}))


if __name__ == "__main__":
    main()
