"""ValueNorm / PopArt against the reference's classes (fixtures: oracle/make_golden_norms.py)."""
import numpy as np
import pytest
import torch

from onpolicy.algorithms.utils.popart import PopArt
from onpolicy.utils.valuenorm import ValueNorm


@pytest.mark.parametrize("tag,kw", [("vn", {}), ("vn_pe", dict(per_element_update=True)), ("vn_beta", dict(beta=0.9))])
def test_valuenorm_matches_reference(gold, tag, kw):
    z = gold.npz("norm_cases")
    vn = ValueNorm(1, **kw)
    for i in range(3):
        vn.update(torch.from_numpy(z["batch%d" % i]))
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        np.testing.assert_allclose(got, z["%s_stats%d" % (tag, i)], rtol=1e-6, atol=1e-12)
    probe = torch.from_numpy(z["probe"])
    np.testing.assert_allclose(vn.normalize(probe).numpy(), z[tag + "_normalize"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(vn.denormalize(probe).numpy(), z[tag + "_denormalize"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(vn.denormalize(z["probe"]), z[tag + "_denormalize"], rtol=1e-6, atol=1e-6)   # ndarray in/out
    sigma, mu = [float(v) for v in vn.denorm_scalars()]
    np.testing.assert_allclose(z["probe"] * np.float32(sigma) + np.float32(mu), z[tag + "_denormalize"], rtol=1e-6, atol=1e-6)


def test_popart_layer_matches_reference(gold):
    z = gold.npz("norm_cases")
    pa = PopArt(6, 1)
    with torch.no_grad():
        pa.weight.copy_(torch.from_numpy(z["pa_weight"]))
        pa.bias.copy_(torch.from_numpy(z["pa_bias"]))
        pa.mean.fill_(2.5e-5)
        pa.mean_sq.fill_(9.0e-5)
        pa.debiasing_term.fill_(1.0e-5)
    np.testing.assert_allclose(pa(torch.from_numpy(z["pa_x"])).detach().numpy(), z["pa_forward"], rtol=1e-6, atol=1e-6)
    probe = torch.from_numpy(z["probe"])
    np.testing.assert_allclose(pa.normalize(probe).numpy(), z["pa_normalize"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pa.denormalize(probe).numpy(), z["pa_denormalize"], rtol=1e-6, atol=1e-6)
