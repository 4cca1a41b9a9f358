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
    info = trainer.train(buf)
    for k, ref in meta["train_info"].items():
        assert info[k] == pytest.approx(ref, rel=3e-4, abs=2e-6), (k, info[k], ref)
    C.check_weights(z, key + "final_actor.", policy.actor, rtol=1e-4, atol=2e-5)
    C.check_weights(z, key + "final_critic.", policy.critic, rtol=1e-4, atol=2e-5)
