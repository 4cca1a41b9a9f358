#!/usr/bin/env python
"""GPU box: how far are the K9 kernels from float64 -- under both arithmetic forms of the matrix products
(include/mappo_hip.h MAPPO_ARITH_SIX_TERM -- the default -- and MAPPO_ARITH_F32_MFMA)?  One million rows through MLPBase + head at the
north-star widths, outputs and every parameter gradient against the float64 modules on the device (the comparison of
tests/test_gpu_mlp.py::test_trunk_at_scale_vs_float64, with the errors printed instead of asserted).

    python tools/six_term_accuracy.py > gpurun_out/six_term_accuracy.json
"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "on-policy_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.nn as nn


def main():
    from onpolicy import _native
    from onpolicy.algorithms.utils import fused_mlp
    from onpolicy.algorithms.utils.mlp import MLPBase
    from helpers import make_args
    dev = torch.device("cuda", 0)
    out_rows = []
    for din, out in ((384, 1), (48, 5), (152, 1)):
        args = make_args(hidden_size=64, layer_N=1, use_ReLU=False)
        torch.manual_seed(5)
        base = MLPBase(args, (din,))
        head = nn.Linear(64, out)
        ref_base, ref_head = copy.deepcopy(base).double().to(dev), copy.deepcopy(head).double().to(dev)
        base, head = base.to(dev), head.to(dev)
        rows, src_rows = (1 << 20) + 77, (1 << 20) + 5000
        g = torch.Generator(device=dev).manual_seed(din)
        src = torch.randn(src_rows, din, device=dev, generator=g) * 1.5 + 0.7
        idx = torch.randperm(src_rows, device=dev, generator=g)[:rows]
        dy = torch.randn(rows, out, device=dev, generator=g) / rows ** 0.5
        y_ref = ref_head(ref_base(src.double()[idx]))
        y_ref.backward(dy.double())
        for label, arith in (("float32 MFMA (MAPPO_ARITH_F32_MFMA)", "f32_mfma"), ("six-term bf16 (MAPPO_ARITH_SIX_TERM, the default)", "six_term")):
            fused_mlp.set_matrix_arithmetic(base, arith)
            try:
                for p in list(base.parameters()) + list(head.parameters()):
                    p.grad = None
                rs = fused_mlp.RowSource(fused_mlp.standardize_rows(src), idx, standardized=True, width=din)
                y = fused_mlp.trunk_forward(base, rs, head)
                y.backward(dy)
                torch.cuda.synchronize()
            finally:
                fused_mlp.set_matrix_arithmetic(base, "six_term")
            errs = {"y": float((y.detach().double() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())}
            for (name, p), q in list(zip(base.named_parameters(), ref_base.parameters())) + \
                    list(zip(head.named_parameters(), ref_head.parameters())):
                key = ("head." if p.shape[0] == out and p.dim() <= 2 and name in ("weight", "bias") else "") + name
                errs["grad " + key] = float((p.grad.double() - q.grad).abs().max() / q.grad.abs().max())
            out_rows.append({"din": din, "out": out, "rows": rows, "kernels": label,
                             "max_error_over_largest_entry_vs_float64": errs,
                             "worst": max(errs.values())})
    print(json.dumps({"what": "K9 forward + backward at one million rows against the float64 modules on the device",
                      "rows": out_rows}, indent=1))


if __name__ == "__main__":
    main()
