"""One comparison of a finished ``R_MAPPO.train`` on the device with what the REFERENCE left behind on the same inputs
(tests/golden/trainer_*_cases.npz), shared by the -m gpu trainer tests, with ONE table of tolerances.

The tables sit at about three times the worst deviation measured over the whole device suite on the MI355X
(profiles/r06_parity_margins.json, written by the ``margins`` fixture of conftest.py; VERDICT r5 "next" #9): a kernel change that
costs accuracy shows up as a failing test instead of disappearing inside a 30x margin.  Units:
  info_rel    the six train_info scalars, relative (floor 1e-5 absolute);
  weight_abs  parameters after train(), absolute, largest entry (an Adam step moves a weight by <= lr = 5e-4 ... 1e-3);
  weight_bulk the same, 99.5th percentile per tensor (Adam divides a gradient by its own magnitude: a handful of entries whose
              gradient is noise move by noise / |noise| x lr -- the maximum is an ill-conditioned statistic, the bulk is not);
  grad_rel    what the last ppo_update left in .grad (after clipping), relative to the tensor's largest reference entry;
  norm_rtol   ValueNorm's running statistics, relative to the largest of the three.
"""
import numpy as np
import pytest

import cfg_shapes as C

# Measured worst cases over the device suite (round 6, profiles/r06_parity_margins.json, 26 comparisons):
#   hidden 64 (K9 / K12 routes: trainer_h64 x {graph, eager}, device_route, cfg4_shape, mid_size at 10^5 rows):
#       train_info 3.9e-7 relative, weights 7.4e-7 absolute, last gradients 4.2e-5 of the tensor's largest entry, ValueNorm 1.2e-7
#   hidden 512 (cfg5_shape through the library GEMMs and through K15): 4.6e-6, 1.6e-6, 2.1e-4, 1.1e-7
# Until round 6 every trainer test asserted 1e-3 / 5e-5 / 1e-3 (30-2500 x the measured values).
# (Second measurement, same round: the standardised observation copies moved to another kernel with the same accuracy against
# float64 -- inputs that differ in the last bit of 4 % of their elements.  train_info's worst case went from 3.9e-7 to 2.2e-6
# (critic_grad_norm of mid_ns after two epochs), and twelve of the 384 entries of the critic's feature_norm.bias of mid_ns moved
# by up to 1.0e-5: their gradients are at noise level, and Adam divides a gradient by its own running magnitude -- a parameter
# whose gradient is noise takes a step of noise / |noise| x lr.  That sensitivity, not the first run's luck, is what the margins
# have to cover: weights 3e-5 absolute (4 % of an Adam step; the 5e-5 of rounds 1-5 was not as loose as it looked).)
# (Third measurement -- the same perturbation, one kernel rebuild later: three of the 24 576 entries of mid_ns's first critic
# layer off by 4.7e-5.  The MAXIMUM over the weights is the statistic Adam makes ill-conditioned, so it only guards against gross
# errors (a wrong sign is 1.4e-3): 1.5e-4.  The bulk is the tight check: the 99.5th percentile of |difference| per tensor.)
TOL = {"info_rel": 7e-6, "info_abs": 1e-7, "weight_abs": 1.5e-4, "weight_bulk": 3e-6, "weight_rtol": 1e-5, "grad_rel": 1.5e-4,
       "norm_rtol": 5e-7}
# (hidden 512: the library route's GEMM kernels are picked per box by TunableOp, so the margin is 5 x, not 3 x)
TOL_H512 = {"info_rel": 2.5e-5, "info_abs": 1e-7, "weight_abs": 1.5e-4, "weight_bulk": 3e-6, "weight_rtol": 1e-5, "grad_rel": 1e-3,
            "norm_rtol": 5e-7}

def compare_update(z, key, meta, policy, trainer, info, tol=None):
    """Asserts every quantity against the fixture -> {quantity: deviation} (for the ``margins`` record)."""
    tol = dict(TOL, **(tol or {}))
    worst = {}
    for k, ref in meta["train_info"].items():
        worst["info." + k] = abs(info[k] - ref) / max(abs(ref), 1e-5)
        assert info[k] == pytest.approx(ref, rel=tol["info_rel"], abs=tol["info_abs"]), (k, info[k], ref)
    for net, pre in ((policy.actor, "final_actor."), (policy.critic, "final_critic.")):
        for k, v in net.state_dict().items():
            got = v.detach().cpu().numpy()
            sub, ref, mom = C.stored(z, key + pre + k, got)
            diff = np.abs(sub - ref).ravel()
            worst["w." + pre + k] = float(diff.max()) if diff.size else 0.0
            bulk = float(np.quantile(diff, 0.995)) if diff.size else 0.0
            worst["wq." + pre + k] = bulk
            np.testing.assert_allclose(sub, ref, rtol=tol["weight_rtol"], atol=tol["weight_abs"], err_msg=pre + k)
            assert bulk <= tol["weight_bulk"], (pre + k, "99.5th percentile of |difference|", bulk)
            if mom is not None:     # the elements in between, in aggregate
                g64 = got.astype(np.float64)
                assert abs(g64.sum() - mom[0]) <= tol["weight_abs"] * g64.size, (pre + k, g64.sum(), mom[0])
                np.testing.assert_allclose((g64 * g64).sum(), mom[1], rtol=1e-4, err_msg=pre + k + " (sum of squares)")
    for net, pre in ((policy.actor, "last_grad_actor."), (policy.critic, "last_grad_critic.")):
        for k, p in net.named_parameters():
            got = p.grad.detach().cpu().numpy()
            sub, ref, mom = C.stored(z, key + pre + k, got)
            scale = max(1e-12, float(np.abs(ref).max()))
            err = float(np.abs(sub - ref).max()) / scale
            worst["g." + pre + k] = err
            assert err < tol["grad_rel"], (pre + k, err)
            if mom is not None:
                g64 = got.astype(np.float64)
                np.testing.assert_allclose((g64 * g64).sum(), mom[1], rtol=2e-3, err_msg=pre + k + " (sum of squares)")
    vn = trainer.value_normalizer
    if vn is not None and (key + "final_norm") in z.files:
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        ref = z[key + "final_norm"]
        worst["valuenorm"] = float(np.abs(got - ref).max() / max(1e-12, np.abs(ref).max()))
        assert worst["valuenorm"] <= tol["norm_rtol"], (got, ref)        # (of the largest statistic: the three differ by 10^5)
    return worst


def top3(worst):
    def cls(k):
        return k.split(".", 1)[0]
    best = {}
    for k, v in worst.items():
        if v > best.get(cls(k), ("", -1.0))[1]:
            best[cls(k)] = (k, v)
    return {c: kv for c, kv in sorted(best.items())}
