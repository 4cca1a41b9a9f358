"""The PyTorch part of the path (networks + trainer on the host oracle buffer) against the reference-generated MID-SIZE
fixtures: >= 10^5 rows at the north-star flags (tests/golden/trainer_mid_cases.npz, oracle/make_golden_trainer.py:
CASES_MID; VERDICT r5 "weak" #1).  The device twin -- where the size matters: several tiles per wave, grid caps, split
reductions -- is tests/test_gpu_mid_size.py."""
import numpy as np
import pytest
import torch

import cfg_shapes as C
from oracle import oracle


@pytest.mark.parametrize("cname", C.MID_CASES)
def test_train_matches_reference_at_mid_size(gold, cname):
    z, key, meta, spec, args, spaces, policy, trainer = C.build(gold, cname, fixture=C.MID_FIXTURE)
    C.start_from_reference_weights(policy, z, key, rtol=1e-5, atol=1e-6)
    arrays, nv = C.inputs(spec, z, key)
    np.testing.assert_array_equal(nv, z[key + "next_value"])
    buf = oracle.OracleBuffer(args, spec["A"], *spaces)
    for name, arr in arrays.items():
        getattr(buf, name)[...] = arr
    buf.compute_returns(nv, trainer.value_normalizer)
    sub, ref, mom = C.stored(z, key + "returns", buf.returns)
    np.testing.assert_array_equal(sub, ref)
    r64 = buf.returns.astype(np.float64)
    np.testing.assert_allclose([r64.sum(), (r64 * r64).sum()], mom, rtol=1e-12)
    trainer.prep_training()
    torch.manual_seed(21)
    first, inner = {}, trainer._run_update

    def recording_update(sample, update_actor):
        out = inner(sample, update_actor)
        if not first:
            for net, pre in ((policy.actor, "first_grad_actor."), (policy.critic, "first_grad_critic.")):
                for k, p in net.named_parameters():
                    first[pre + k] = p.grad.detach().clone()
        return out
    trainer._run_update = recording_update
    info = trainer.train(buf)
    for name, got in first.items():             # the first update's gradients: the well-conditioned quantity at this size
        sub, ref, mom = C.stored(z, key + name, got.numpy())
        # (the reference itself moves by 3e-5 of such a tensor between one and four CPU threads: sums that nearly cancel)
        assert float(np.abs(sub - ref).max()) <= 1e-4 * max(1e-12, float(np.abs(ref).max())), name
    for k, ref in meta["train_info"].items():
        assert info[k] == pytest.approx(ref, rel=3e-4, abs=2e-6), (k, info[k], ref)
    C.check_weights(z, key + "final_actor.", policy.actor, rtol=1e-4, atol=2e-5)
    C.check_weights(z, key + "final_critic.", policy.critic, rtol=1e-4, atol=2e-5)
