#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/norm_cases.npz from the REFERENCE's value normalisers:
ValueNorm (onpolicy/utils/valuenorm.py) after a sequence of updates -- statistics, normalize / denormalize outputs,
per_element_update -- and PopArt (onpolicy/algorithms/utils/popart.py) forward / normalize / denormalize with
statistics set by hand (its update() cannot run on the CPU: it assigns plain tensors to Parameters, popart.py:64).

    python oracle/make_golden_norms.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

ref = mg.ref


def main():
    out = {}
    rng = np.random.default_rng(12)
    batches = [(rng.standard_normal((40, 1)) * s + m).astype(np.float32) for s, m in ((3.0, 1.5), (0.5, -2.0), (10.0, 4.0))]
    probe = rng.standard_normal((17, 1)).astype(np.float32) * 5
    for tag, kw in (("vn", {}), ("vn_pe", dict(per_element_update=True)), ("vn_beta", dict(beta=0.9))):
        vn = ref.ValueNorm(1, **kw)
        for i, b in enumerate(batches):
            vn.update(torch.from_numpy(b))
            out["%s_stats%d" % (tag, i)] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                     float(vn.debiasing_term)], dtype=np.float64)
        out[tag + "_normalize"] = vn.normalize(torch.from_numpy(probe)).numpy()
        out[tag + "_denormalize"] = np.asarray(vn.denormalize(torch.from_numpy(probe)))
    for i, b in enumerate(batches):
        out["batch%d" % i] = b
    out["probe"] = probe
    torch.manual_seed(3)
    pa = ref.PopArt(6, 1)
    with torch.no_grad():
        pa.mean.fill_(2.5e-5)
        pa.mean_sq.fill_(9.0e-5)
        pa.debiasing_term.fill_(1.0e-5)
    x = torch.from_numpy(rng.standard_normal((9, 6)).astype(np.float32))
    out["pa_weight"], out["pa_bias"] = pa.weight.detach().numpy().copy(), pa.bias.detach().numpy().copy()
    out["pa_x"] = x.numpy()
    out["pa_forward"] = pa(x).detach().numpy()
    out["pa_normalize"] = pa.normalize(torch.from_numpy(probe)).numpy()
    out["pa_denormalize"] = np.asarray(pa.denormalize(torch.from_numpy(probe)))
    np.savez_compressed(os.path.join(mg.GOLD, "norm_cases.npz"), **out)
    print("norm_cases.npz: %d arrays" % len(out))


if __name__ == "__main__":
    main()
