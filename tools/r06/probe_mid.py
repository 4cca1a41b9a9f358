"""Diagnosis: mid_ns on the device route, deviations from the reference fixture under the two standardisation kernels."""
import os, sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, "on-policy_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import cfg_shapes as C
import parity
from conftest import GOLD

class G(object):
    def npz(self, name): return np.load(os.path.join(GOLD, name + ".npz"))
    def meta(self, name): return json.load(open(os.path.join(GOLD, name + ".json")))

from test_gpu_cfg_shapes import _device_buffer
from onpolicy.utils import shared_buffer as sb
from onpolicy.algorithms.utils import fused_mlp
orig = sb.SharedReplayBuffer._standardize_field
for mode in ("k2", "mlp", "k2"):
    if mode == "mlp":
        sb.SharedReplayBuffer._standardize_field = lambda self, rows, out=None, eps=1e-5: fused_mlp.standardize_rows(rows, out=out)
    else:
        sb.SharedReplayBuffer._standardize_field = orig
    for rng_mode in ("device", "host"):
        dev = torch.device("cuda", 0)
        z, key, meta, spec, args, spaces, policy, trainer = C.build(G(), "mid_ns", device=dev, fixture=C.MID_FIXTURE, sampler_rng=rng_mode)
        C.start_from_reference_weights(policy, z, key)
        arrays, nv = C.inputs(spec, z, key)
        buf = _device_buffer(args, spec, spaces, arrays, dev)
        buf.compute_returns(nv, trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(21)
        info = trainer.train(buf)
        loose = dict(info_rel=1, info_abs=1, weight_abs=1, weight_bulk=1, weight_rtol=1, grad_rel=1, norm_rtol=1)
        worst = parity.compare_update(z, key, meta, policy, trainer, info, tol=loose)
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
        print(mode, rng_mode, [(k[-40:], float("%.3g" % v)) for k, v in top])
