#!/usr/bin/env python
"""Micro-benchmarks of the HIP kernels on cold data (buffers rotated so that every launch streams
from HBM, not from the 256 MiB Infinity Cache).  Prints one JSON object; used for tuning and for
the per-kernel numbers quoted in DESIGN.md.

    python tools/bench_kernels.py [--T 400 --N 4096 --A 8] [--variants 1,2,3] [--gather-N 1024]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
sys.path.insert(0, ROOT)

import torch

from onpolicy import _native


def time_launches(fn, n_sets, iters, warm=3):
    for i in range(warm):
        fn(i % n_sets)
    torch.cuda.synchronize()
    evs = []
    for i in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(i % n_sets)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    # back-to-back total (launch overhead pipelined away)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % n_sets)
    b.record()
    torch.cuda.synchronize()
    return {"median_us": 1e3 * ms[len(ms) // 2], "min_us": 1e3 * ms[0], "b2b_us": 1e3 * a.elapsed_time(b) / iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--A", type=int, default=8)
    ap.add_argument("--variants", default="0,2,20,33,42,3042,51,3051,54,3054,57,3057,99")
    ap.add_argument("--gather-variants", default="1,5,2,3,113,133,69,101")
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--gather-N", type=int, default=1024)
    ap.add_argument("--skip-gather", action="store_true")
    ap.add_argument("--cold", action="store_true", help="also time single launches after flushing caches/TLBs with an 8 GB copy")
    opt = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _native.lib()
    T, C = opt.T, opt.N * opt.A
    stream = torch.cuda.current_stream().cuda_stream
    out = {"T": T, "C": C, "device": torch.cuda.get_device_name(0)}

    # reference point: large device copy
    x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    r = time_launches(lambda i: y.copy_(x), 1, 10)
    out["copy_1GiB_GBps"] = 2 * x.numel() * 4 / (r["b2b_us"] * 1e-6) / 1e9
    del x, y
    # ... and a device copy that moves as many bytes as one fused GAE launch at THIS size (24 B / element): what a plain
    # streaming kernel reaches when the whole launch lasts a few microseconds
    n_small = 24 * T * C // 8
    xs = [torch.empty(n_small, dtype=torch.float32, device=dev) for _ in range(opt.sets)]
    ys = [torch.empty(n_small, dtype=torch.float32, device=dev) for _ in range(opt.sets)]
    r = time_launches(lambda i: ys[i].copy_(xs[i]), opt.sets, opt.iters)
    out["copy_same_bytes"] = {"bytes": 8 * n_small, "median_us": r["median_us"], "b2b_us": r["b2b_us"],
                              "GBps_median": 8 * n_small / (r["median_us"] * 1e-6) / 1e9,
                              "GBps_b2b": 8 * n_small / (r["b2b_us"] * 1e-6) / 1e9}
    del xs, ys

    sets = []
    for s in range(opt.sets):
        g = torch.Generator(device=dev)
        g.manual_seed(s)
        d = dict(r=torch.randn(T, C, device=dev, generator=g), v=torch.randn(T + 1, C, device=dev, generator=g),
                 m=(torch.rand(T + 1, C, device=dev, generator=g) > 0.04).float(),
                 am=torch.ones(T + 1, C, device=dev), bad=torch.ones(T + 1, C, device=dev),
                 ret=torch.zeros(T + 1, C, device=dev), adv=torch.zeros(T, C, device=dev),
                 nv=torch.randn(C, device=dev, generator=g))
        sets.append(d)
    # ceilings for the scan's traffic mix (3 reads + 1 write), flat and in the strip pattern
    import ctypes
    probe_path = os.path.join(ROOT, "tools", "libprobe.so")
    if os.path.exists(probe_path):
        P = ctypes.CDLL(probe_path)
        vp = ctypes.c_void_p
        P.probe_flat.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, ctypes.c_int, vp]
        P.probe_strip.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, vp]
        pr = {}
        n = T * C
        for blocks in (2048, 4096, 8192):
            def fn(i):
                d = sets[i]
                assert P.probe_flat(d["r"].data_ptr(), d["v"].data_ptr(), d["m"].data_ptr(), d["ret"].data_ptr(),
                                    n, blocks, stream) == 0
            t = time_launches(fn, opt.sets, opt.iters)
            t["GBps_b2b"] = 16 * n / (t["b2b_us"] * 1e-6) / 1e9
            pr["flat_%d" % blocks] = t
        for v in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            def fn(i):
                d = sets[i]
                assert P.probe_strip(d["r"].data_ptr(), d["v"].data_ptr(), d["m"].data_ptr(), d["ret"].data_ptr(),
                                     T, C, v, stream) == 0
            t = time_launches(fn, opt.sets, opt.iters)
            t["GBps_b2b"] = 16 * n / (t["b2b_us"] * 1e-6) / 1e9
            pr["strip_v%d" % v] = t
        out["probe_3r1w"] = pr
    den = torch.tensor([0.1, 0.0], device=dev)
    rows = lib.mappo_gae_partial_rows(C)
    partials = torch.zeros(rows, 3, dtype=torch.float64, device=dev)
    p = _native.ptr
    res = {}
    for variant in [int(v) for v in opt.variants.split(",")]:
        lib.mappo_gae_set_variant(variant)
        for mode, fused in (("plain16", False), ("fused24", True)):
            def fn(i):
                d = sets[i]
                code = lib.mappo_gae_f32(p(d["r"]), p(d["v"]), p(d["nv"]), p(d["m"]), None, p(d["ret"]), p(den),
                                         p(d["adv"]) if fused else None, p(d["am"]) if fused else None,
                                         p(partials) if fused else None, T, C, 0.99, 0.95, 1 | 4, stream)
                assert code == 0, code
            t = time_launches(fn, opt.sets, opt.iters)
            nbytes = (24 if fused else 16) * T * C
            t["GBps_median"] = nbytes / (t["median_us"] * 1e-6) / 1e9
            t["GBps_b2b"] = nbytes / (t["b2b_us"] * 1e-6) / 1e9
            res["v%d_%s" % (variant, mode)] = t
    if opt.cold:
        big_a = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device=dev)
        big_b = torch.empty_like(big_a)
        cold = {}
        for variant in [int(v) for v in opt.variants.split(",")]:
            lib.mappo_gae_set_variant(variant)
            ms = []
            for i in range(6):
                big_b.copy_(big_a)            # 16 GB of traffic: evicts L2 / MALL / TLBs
                d = sets[i % opt.sets]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                code = lib.mappo_gae_f32(p(d["r"]), p(d["v"]), p(d["nv"]), p(d["m"]), None, p(d["ret"]), p(den),
                                         p(d["adv"]), p(d["am"]), p(partials), T, C, 0.99, 0.95, 1 | 4, stream)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            cold["v%d_fused24_cold_us" % variant] = [round(1e3 * x, 1) for x in ms]
        out["gae_cold"] = cold
        del big_a, big_b
    lib.mappo_gae_set_variant(0)
    out["gae"] = res
    del sets

    if not opt.skip_gather:
        # feed-forward gather at the north-star row shape, full permutation of T*N*A rows
        N = opt.gather_N
        B = T * N * opt.A
        widths = dict(share_obs=384, obs=48, actions=1, value_preds=1, returns=1, masks=1, active_masks=1,
                      logp=1, adv=1, avail=5)
        src = {k: torch.randn(B, w, device=dev) for k, w in widths.items()}
        dst = {k: torch.empty(B, w, device=dev) for k, w in widths.items()}
        stats = torch.tensor([0.0, 1.0], device=dev)
        gres = {}
        perms = [torch.randperm(B, device=dev) for _ in range(2)]
        combos = [("all_fields", list(widths)), ("share_obs_only", ["share_obs"]),
                  ("scalars_only", ["actions", "value_preds", "returns", "masks", "active_masks", "logp", "adv"])]
        for gv in [int(v) for v in opt.gather_variants.split(",")]:
            lib.mappo_gather_set_variant(gv)
            for label, names in combos[:2]:
                fields = (_native.Field * len(names))(*[
                    _native.Field(src[k].data_ptr(), dst[k].data_ptr(), widths[k], 0, 1 if k == "adv" else 0, 0)
                    for k in names])

                def fn(i):
                    code = lib.mappo_gather_rows(fields, len(names), perms[i].data_ptr(), B, p(stats), stream)
                    assert code == 0, code
                t = time_launches(fn, 2, 4, warm=1)
                nbytes = sum(2 * 4 * widths[k] for k in names) * B + 8 * B
                gres["gv%d_%s" % (gv, label)] = {"ms": t["median_us"] / 1e3,
                                                 "GBps": nbytes / (t["median_us"] * 1e-6) / 1e9}
        lib.mappo_gather_set_variant(0)
        for label, names in combos:
            fields = (_native.Field * len(names))(*[
                _native.Field(src[k].data_ptr(), dst[k].data_ptr(), widths[k], 0, 1 if k == "adv" else 0, 0)
                for k in names])

            def fn(i):
                code = lib.mappo_gather_rows(fields, len(names), perms[i].data_ptr(), B, p(stats), stream)
                assert code == 0, code
            t = time_launches(fn, 2, 6, warm=1)
            nbytes = sum(2 * 4 * widths[k] for k in names) * B + 8 * B
            t["GBps_median"] = nbytes / (t["median_us"] * 1e-6) / 1e9
            t["bytes"] = nbytes
            gres[label] = t
        # what torch's own indexing does for the wide field
        idx = perms[0]
        r = time_launches(lambda i: torch.index_select(src["share_obs"], 0, idx), 1, 4, warm=1)
        r["GBps_median"] = (2 * 4 * 384 + 8) * B / (r["median_us"] * 1e-6) / 1e9
        gres["torch_index_select_share_obs"] = r
        out["gather"] = gres
        out["gather_rows"] = B
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
