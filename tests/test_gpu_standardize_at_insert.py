"""-m gpu: the row-standardised observation copies the fused trunk kernels (K9) read stay resident after a train() and are
kept current slab by slab by insert / chooseinsert / after_update (utils/shared_buffer.py: _obs_slab_written; VERDICT r5
"next" #4a), inside K2's own launch (mappo_slab_copy_std) -- bit-identical to standardising the whole field again (the full
pass runs the same device code), and within float32 rounding of the float64 definition; never stale:

* slab writes of this class's kernels update exactly the slab they wrote (and nothing for row T, which is not part of the copy);
* in-place torch writes to the field (runners do ``buffer.obs[0] = ...``) are seen through the version counter -> full pass;
* a whole rollout + train() loop ends with the weights of the same loop under MAPPO_STANDARDIZE_AT_INSERT=0, bit for bit.
"""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args

pytestmark = pytest.mark.gpu

T, N, A, Do, Ds, NA = 6, 5, 3, 18, 54, 5


def _buffer(**kw):
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=64, **kw)
    return args, SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(NA), device=torch.device("cuda", 0))


def _step_values(rng):
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    # positional order of insert / chooseinsert (reference shared_buffer.py:90, :125)
    return (f(N, A, Ds), f(N, A, Do), f(N, A, 1, 64), f(N, A, 1, 64), rng.integers(0, NA, (N, A, 1)).astype(np.float32),
            f(N, A, 1), f(N, A, 1), f(N, A, 1), np.ones((N, A, 1), np.float32))


def _fresh(buf, name):
    """The full pass over the field as it is now (into fresh storage)."""
    field = getattr(buf, name)
    return buf._standardize_field(field[:T].reshape(T * N * A, -1).clone())


@pytest.mark.parametrize("mode", ["insert", "chooseinsert"])
def test_slab_writes_keep_the_resident_copy_current(mode):
    args, buf = _buffer()
    rng = np.random.default_rng(3)
    buf.share_obs.normal_()
    buf.obs.normal_()
    for name in ("share_obs", "obs"):
        first = buf._obs_rows(name, True)
        torch.testing.assert_close(first, _fresh(buf, name), rtol=0, atol=0)
    assert buf.std_full_passes == 2 and buf.std_slab_launches == 0
    buf.after_update()          # slab 0 <- slab T
    for t in range(T):
        getattr(buf, mode)(*_step_values(rng))
    # insert wrote slabs 1 .. T (T is not part of the copy), chooseinsert 0 .. T - 1; after_update slab 0
    expect = 2 * (1 + (T - 1 if mode == "insert" else T))
    assert buf.std_slab_launches == expect, (buf.std_slab_launches, expect)
    for name in ("share_obs", "obs"):
        ptr = buf._std_rows[name][1].data_ptr()
        got = buf._obs_rows(name, True)
        assert got.data_ptr() == ptr                       # same storage: an update graph's addresses survive
        torch.testing.assert_close(got, _fresh(buf, name), rtol=0, atol=0)
    assert buf.std_full_passes == 2                         # no full pass since the first one


def test_an_in_place_torch_write_forces_the_full_pass():
    args, buf = _buffer()
    buf.share_obs.normal_()
    buf.obs.normal_()
    buf._obs_rows("obs", True)
    buf._obs_rows("share_obs", True)
    buf.obs[2, 1] = 7.0                                   # the reference runners' idiom (buffer.obs[0] = obs.copy())
    rng = np.random.default_rng(5)
    buf.insert(*_step_values(rng))                       # a slab write on a stale copy must not mark it current
    assert buf.std_slab_launches == 1                     # (share_obs only)
    got = buf._obs_rows("obs", True)
    assert buf.std_full_passes == 3
    torch.testing.assert_close(got, _fresh(buf, "obs"), rtol=0, atol=0)
    torch.testing.assert_close(buf._obs_rows("share_obs", True), _fresh(buf, "share_obs"), rtol=0, atol=0)
    assert buf.std_full_passes == 3


def _loop(monkeypatch, at_insert):
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    monkeypatch.setenv("MAPPO_STANDARDIZE_AT_INSERT", "1" if at_insert else "0")
    dev = torch.device("cuda", 0)
    args, buf = _buffer(ppo_epoch=2, num_mini_batch=1)
    torch.manual_seed(1)
    policy = R_MAPPOPolicy(args, Box((Do,)), Box((Ds,)), Discrete(NA), device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    rng = np.random.default_rng(11)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    buf.share_obs.normal_(generator=g)
    buf.obs.normal_(generator=g)
    for it in range(3):
        for t in range(T):
            buf.insert(*_step_values(rng))
        buf.compute_returns(rng.standard_normal((N, A, 1)).astype(np.float32), trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(21 + it)
        trainer.train(buf)
        buf.after_update()
    w = torch.cat([p.detach().reshape(-1) for net in (policy.actor, policy.critic) for p in net.parameters()])
    return w.cpu().numpy(), buf.std_full_passes, buf.std_slab_launches


def test_training_loop_is_bit_identical_to_the_full_pass_per_train(monkeypatch):
    w1, full1, slabs1 = _loop(monkeypatch, True)
    w0, full0, slabs0 = _loop(monkeypatch, False)
    np.testing.assert_array_equal(w1, w0)
    assert slabs0 == 0 and full0 == 6                      # one full pass per field and train()
    assert full1 == 2 and slabs1 > 0, (full1, slabs1)      # only the first train() standardised whole fields


def test_standardised_rows_against_float64():
    """The device rows against (x - mean) / sqrt(var + 1e-5) in float64, for the widths of the BASELINE configs (incl. the odd
    ones, padded with zero columns)."""
    _, buf = _buffer()
    for D in (18, 30, 48, 54, 150, 370, 384, 435, 1285):
        x = torch.randn(333, D, device=buf.device) * 3 + 0.5
        got = buf._standardize_field(x)
        assert got.shape == (333, (D + 3) // 4 * 4) and torch.all(got[:, D:] == 0)
        x64 = x.double()
        ref = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5)
        assert (got[:, :D].double() - ref).abs().max().item() < 2e-6
