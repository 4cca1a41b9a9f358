#!/usr/bin/env python
"""GPU-box probe: what the first GAE launch of a step pays beyond a back-to-back launch (56-62 us against 51 us at the north
star).  Every variant first evicts caches and TLBs with a 16 GB copy (what the update phase does to them), then times ONE
fused GAE launch with events:

  spin        behind a 0.1 ms single-thread spin (bench.py's in-situ measurement: the device sits idle before the launch)
  busy        directly behind a 200 MB device copy that is still running when the launch is queued (no idle gap)
  tlb         spin, but one 4-byte read per 64 KB of the six arrays first (translations warm, data cold)
  mall        spin, but the four input arrays read once first (translations and Infinity Cache warm)
  b2b         the same launch again right behind the previous one
  clean       (r6) spin, but a READ-ONLY sweep of an unrelated 1 GB first: the Infinity Cache holds clean foreign lines, so the
              launch evicts nothing dirty (the eviction copy leaves 256 MiB of dirty lines that the other cold modes write
              back during the launch: MI355X_MICROARCH.md "boundary": + B / 6 TB/s for B dirty bytes of the predecessor)
  k2          (r6) clean, then the step's real predecessor: one fused slab copy of after_update's size (57 MB written)

    python tools/gae_in_situ_probe.py [--T 400 --N 4096 --A 8] [--reps 6]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))

import torch

from onpolicy import _native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--A", type=int, default=8)
    ap.add_argument("--reps", type=int, default=6)
    opt = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib, p = _native.lib(), _native.ptr
    T, C = opt.T, opt.N * opt.A
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    d = dict(r=torch.randn(T, C, device=dev, generator=g), v=torch.randn(T + 1, C, device=dev, generator=g),
             m=(torch.rand(T + 1, C, device=dev, generator=g) > 0.04).float(), am=torch.ones(T + 1, C, device=dev),
             ret=torch.zeros(T + 1, C, device=dev), adv=torch.zeros(T, C, device=dev),
             nv=torch.randn(C, device=dev, generator=g))
    den = torch.tensor([0.1, 0.0], device=dev)
    partials = torch.zeros(lib.mappo_gae_partial_rows(C), 3, dtype=torch.float64, device=dev)
    big_a = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device=dev)
    big_b = torch.empty_like(big_a)
    mid_a = torch.empty(50 * 1024 ** 2, dtype=torch.float32, device=dev)
    mid_b = torch.empty_like(mid_a)

    def gae():
        code = lib.mappo_gae_f32(p(d["r"]), p(d["v"]), p(d["nv"]), p(d["m"]), None, p(d["ret"]), p(den), p(d["adv"]),
                                 p(d["am"]), p(partials), T, C, 0.99, 0.95, 1 | 4, stream)
        assert code == 0, code

    def timed():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gae()
        b.record()
        return a, b

    stride = 64 * 1024 // 4
    out = {"T": T, "C": C, "bytes": 24 * T * C, "us": {}}
    slab_src = torch.randn(opt.N * opt.A * 432, device=dev, generator=g)       # after_update's obs + share_obs slabs (57 MB)
    slab_dst = torch.empty_like(slab_src)
    slab = (_native.Slab * 1)(_native.Slab(slab_src.data_ptr(), slab_dst.data_ptr(), slab_src.numel()))
    for mode in ("spin", "busy", "tlb", "mall", "b2b", "clean", "k2"):
        ts = []
        for _ in range(opt.reps):
            big_b.copy_(big_a)
            torch.cuda.synchronize()
            sink = None
            if mode == "busy":
                mid_b.copy_(mid_a)
            else:
                torch.cuda._sleep(250000)
            if mode == "tlb":
                sink = sum(d[k].reshape(-1)[::stride].sum() for k in ("r", "v", "m", "am", "ret", "adv"))
            if mode == "mall":
                sink = sum(d[k].sum() for k in ("r", "v", "m", "am"))
            if mode == "b2b":
                gae()
            if mode in ("clean", "k2"):
                sink = big_a[:256 * 1024 ** 2].sum()
                if mode == "k2":
                    assert lib.mappo_slab_copy(slab, 1, stream) == 0
            a, b = timed()
            torch.cuda.synchronize()
            ts.append(1e3 * a.elapsed_time(b))
            del sink
        ts.sort()
        out["us"][mode] = {"median": round(ts[len(ts) // 2], 2), "min": round(ts[0], 2), "all": [round(t, 1) for t in ts]}
        out["us"][mode]["frac_of_8TBs"] = round(24 * T * C / (ts[len(ts) // 2] * 1e-6) / 8e12, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
