"""The oracle buffers' storage methods (insert / chooseinsert / after_update / chooseafter_update) against
contents produced by the reference's own buffers on the same seeded stream (oracle/make_golden_storage.py)."""
import numpy as np
import pytest

from oracle import oracle
from storage_replay import FIELDS, replay
from test_oracle_separated import _Args, Box, Discrete


@pytest.mark.parametrize("mode", ["insert", "chooseinsert"])
def test_shared_storage_matches_reference(gold, mode):
    z = gold.npz("storage_cases")
    T, N, A, Do, Ds, na, H = [int(x) for x in z["dims"]]
    args = _Args(episode_length=T, n_rollout_threads=N, hidden_size=H)
    buf = replay(oracle.OracleBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(na)), mode, (N, A), z["dims"])
    assert buf.step == int(z["shared_%s_step" % mode])
    for name in FIELDS:
        np.testing.assert_array_equal(getattr(buf, name), z["shared_%s_%s" % (mode, name)], err_msg=name)


@pytest.mark.parametrize("mode", ["insert", "chooseinsert"])
def test_separated_storage_matches_reference(gold, mode):
    z = gold.npz("storage_cases")
    T, N, A, Do, Ds, na, H = [int(x) for x in z["dims"]]
    args = _Args(episode_length=T, n_rollout_threads=N, hidden_size=H)
    buf = replay(oracle.OracleSeparatedBuffer(args, Box((Do,)), Box((Ds,)), Discrete(na)), mode, (N,), z["dims"])
    for name in FIELDS:
        np.testing.assert_array_equal(getattr(buf, name), z["separated_%s_%s" % (mode, name)], err_msg=name)
