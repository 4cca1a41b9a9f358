"""-m gpu: the route bench.py TIMES, pinned to the reference -- ``sampler_rng="device"`` (K10 partition / identity index list
+ whole-batch cache) and the default GAE mode (time-parallel scan on narrow buffers), end to end through compute_returns +
R_MAPPO.train, against numbers the REFERENCE produced (reference onpolicy/utils/shared_buffer.py:358-361, :509-512 draw the
permutation; r_mappo.py:171-224 consumes it).

* one minibatch per epoch (``h64_ns``, ``h64_gru_straddle`` of trainer_h64_cases.npz): the device sampler's single slice is
  the whole batch, the same SET the reference's randperm covers, so the reference's numbers must hold at the tolerances of
  the host-permutation test; the cached tuple must have been handed out ppo_epoch - 1 times.
* several minibatches (``dev_*`` of trainer_dev_cases.npz): the reference ran with its torch.randperm replaced by the numpy
  restatement of K10 (oracle/k10_partition.py, keys from the same CPU generator under the same seed), so its minibatches
  are the sets the device sampler makes -- feed-forward, with a dropped tail, and chunked (recurrent).
* ``dev_scan_cfg2``: config-2 shapes at 2560 columns x 64 steps, where compute_returns takes the time-parallel scan
  (tolerance mode: returns to rtol 1e-5) and everything downstream runs on its output.
"""
import numpy as np
import pytest

import parity
import torch

from helpers import Box, Discrete, assert_k9_carried_the_updates, make_args

pytestmark = pytest.mark.gpu

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
          "active_masks", "action_log_probs", "available_actions", "rewards")


def _setup(gold, fixture, cname, dev, **extra):
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    z = gold.npz(fixture)
    meta = gold.meta(fixture)[cname]
    spec = meta["spec"]
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], sampler_rng="device", **dict(spec["args"], **extra))
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    buf = SharedReplayBuffer(args, spec["A"], *spaces, device=dev)
    key = "trn_%s_" % cname
    if spec.get("regen"):
        from oracle import synth
        arrays = synth.rollout(spec["T"], spec["N"], spec["A"], spec["Do"], spec["Ds"], spec["na"], seed=4242)
        nv = arrays.pop("next_value")
        np.testing.assert_allclose(synth.digest(arrays, nv), z[key + "input_digest"], rtol=1e-12,
                                   err_msg="the seeded inputs differ from the ones the reference was run on")
        for name, arr in arrays.items():
            getattr(buf, name).copy_(torch.from_numpy(arr))
    else:
        for name in FIELDS:
            dst = getattr(buf, name)
            if dst.stride()[0] != 0:
                dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    for net, pre in ((policy.actor, "init_actor."), (policy.critic, "init_critic.")):
        for k, v in net.state_dict().items():       # start from the reference's exact weights (host QR: last-bit noise)
            np.testing.assert_allclose(v.cpu().numpy(), z[key + pre + k], rtol=1e-4, atol=5e-6)
            v.copy_(torch.from_numpy(z[key + pre + k]))
    return z, key, meta, spec, policy, trainer, buf


def _launches():
    from onpolicy.algorithms.utils import fused_mlp
    t = fused_mlp.profile_times()
    return t.get("mappo_mlp_forward", (0,))[0], t.get("mappo_mlp_backward", (0,))[0]


def _train_and_compare(z, key, meta, spec, policy, trainer, buf, returns_exact, margins=None):
    from onpolicy.algorithms.utils import fused_mlp
    assert buf._sampler_rng == "device" and buf._gae_exact != (not returns_exact)     # the bench's sampler; the scan only on request
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    got = buf.returns.cpu().numpy()
    if returns_exact:
        np.testing.assert_array_equal(got, z[key + "returns"])
    else:
        np.testing.assert_allclose(got, z[key + "returns"], rtol=1e-5, atol=1e-5)    # (returns are O(1): 1e-5 of their scale)
    trainer.prep_training()
    torch.manual_seed(21)
    fused_mlp.profile(True)
    try:
        info = trainer.train(buf)
        torch.cuda.synchronize()
        n_fwd, n_bwd = _launches()
    finally:
        fused_mlp.profile(False)
    updates = spec["args"]["ppo_epoch"] * spec["args"]["num_mini_batch"]
    assert_k9_carried_the_updates(trainer, n_fwd, n_bwd, updates)          # K9 ran, forward and backward (eagerly or replayed)
    reuses = buf.whole_batch_reuses
    buf.after_update()

    worst = parity.compare_update(z, key, meta, policy, trainer, info)
    if margins is not None:
        margins("device_route/" + key[4:-1], worst)
    top = parity.top3(worst)
    print("\n[%s] device sampler: K9 launches fwd %d bwd %d, whole-batch reuses %d; largest relative errors: %s"
          % (key, n_fwd, n_bwd, reuses, top))
    return reuses


@pytest.mark.parametrize("cname", ["h64_ns", "h64_gru_straddle"])
def test_single_minibatch_device_route_vs_reference(gold, cname, margins):
    """num_mini_batch = 1: identity index list + whole-batch cache against the reference's randperm run."""
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_h64_cases", cname, dev)
    reuses = _train_and_compare(z, key, meta, spec, policy, trainer, buf, returns_exact=True, margins=margins)
    assert reuses == spec["args"]["ppo_epoch"] - 1


@pytest.mark.parametrize("cname", ["dev_relu2", "dev_tail", "dev_gru"])
def test_k10_minibatches_vs_reference_on_the_same_partition(gold, cname, margins):
    """Several minibatches per epoch: K10's slices on the device against the reference fed the same slices."""
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_dev_cases", cname, dev)
    reuses = _train_and_compare(z, key, meta, spec, policy, trainer, buf, returns_exact=True, margins=margins)
    assert reuses == 0          # every minibatch is a different set: nothing may be cached


def test_k10_slices_are_the_partition_the_reference_was_fed(gold):
    """The premise of the test above, checked directly: under the fixture's seed the device sampler emits the index lists
    recorded from the reference run (RandpermAsK10.calls)."""
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_dev_cases", "dev_tail", dev)
    n = spec["T"] * spec["N"] * spec["A"]
    n_mb = spec["args"]["num_mini_batch"]
    mb = n // n_mb
    torch.manual_seed(21)
    for epoch in range(meta["n_perms"]):
        idx = buf._sampler_indices(n, mb, n_mb).cpu().numpy()
        np.testing.assert_array_equal(idx, z[key + "perm%d" % epoch][:n_mb * mb])


def test_scan_gae_then_update_vs_reference(gold, margins):
    """Config-2 shapes, 2560 columns x 64 steps, with --gae_scan: compute_returns takes the time-parallel scan (asserted through
    mappo_gae_last_variant; without the flag the bit-exact kernels run, every other test of this file), train() runs on its output through the identity list + whole-batch cache."""
    from onpolicy import _native
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_dev_cases", "dev_scan_cfg2", dev, gae_scan=True)
    reuses = _train_and_compare(z, key, meta, spec, policy, trainer, buf, returns_exact=False, margins=margins)
    assert _native.lib().mappo_gae_last_variant() in (70, 71, 72, 73, 74, 75)
    assert reuses == spec["args"]["ppo_epoch"] - 1


def test_an_edited_whole_batch_tensor_is_gathered_again():
    """The cached one-minibatch tuple is shared between epochs (read-only contract); an in-place edit of a yielded tensor
    must not leak into the next epoch."""
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    dev = torch.device("cuda", 0)
    T, N, A = 6, 4, 3
    args = make_args(episode_length=T, n_rollout_threads=N, sampler_rng="device")
    buf = SharedReplayBuffer(args, A, Box((6,)), Box((18,)), Discrete(5), device=dev)
    buf.rewards.normal_()
    buf.value_preds.normal_()
    buf.compute_returns(torch.zeros(N, A, 1), None if not args.use_valuenorm else _vn(dev))
    adv = buf.normalized_advantages(_vn(dev))
    first = next(iter(buf.feed_forward_generator(adv, 1)))
    keep = first[6].clone()
    again = next(iter(buf.feed_forward_generator(adv, 1)))
    assert again[6] is first[6] and buf.whole_batch_reuses == 1
    first[6].add_(1.0)                                   # a trainer subclass scribbles on the returns it was handed
    third = next(iter(buf.feed_forward_generator(adv, 1)))
    assert third[6] is not first[6]
    torch.testing.assert_close(third[6], keep, rtol=0, atol=0)


def _vn(dev):
    from onpolicy.utils.valuenorm import ValueNorm
    return ValueNorm(1, device=dev)


@pytest.mark.parametrize("cname", ["h64_ns", "h64_gru"])
def test_rollout_forward_on_the_device_route_vs_reference(gold, cname):
    """``get_actions`` the way the rollout calls it in device mode (plain [N * A, dim] tensors, no autograd): trunk + head
    through K9 (``fused_mlp.rollout_rows``), sampling + log-prob through K14.  The values do not depend on the draw and
    must be the reference's (``act_values`` of the fixture); the log-probabilities must be those the reference-pinned
    ``evaluate_actions`` assigns to the sampled actions; sampled actions must be available ones."""
    from onpolicy.algorithms.utils import distributions, fused_mlp
    dev = torch.device("cuda", 0)
    z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_h64_cases", cname, dev)
    B = spec["N"] * spec["A"]
    trainer.prep_rollout()
    flat = lambda name: getattr(buf, name)[0].reshape(B, *getattr(buf, name).shape[3:]).contiguous()
    mode = distributions.SAMPLING_RNG
    distributions.set_sampling_rng("device")        # (a runner sets this from --sampler_rng; no runner here)
    fused_mlp.profile(True)
    try:
        with torch.no_grad():
            torch.manual_seed(7)
            values, actions, logp, h_a, h_c = policy.get_actions(
                flat("share_obs"), flat("obs"), flat("rnn_states"), flat("rnn_states_critic"), flat("masks"),
                flat("available_actions"))
        torch.cuda.synchronize()
        n_fwd, _ = _launches()
    finally:
        fused_mlp.profile(False)
        distributions.set_sampling_rng(mode)
    assert n_fwd == 2, n_fwd                                    # actor and critic trunks both went through K9
    np.testing.assert_allclose(values.cpu().numpy(), z[key + "act_values"], rtol=1e-4, atol=2e-5)
    if cname == "h64_gru":
        np.testing.assert_allclose(h_c.cpu().numpy(), z[key + "act_h_critic"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(h_a.cpu().numpy(), z[key + "act_h_actor"], rtol=1e-4, atol=2e-5)
    assert actions.dtype == torch.int64 and tuple(actions.shape) == (B, 1)
    assert bool((flat("available_actions").gather(-1, actions) == 1).all())
    with torch.no_grad():
        _, ev_logp, _ = policy.evaluate_actions(flat("share_obs"), flat("obs"), flat("rnn_states"),
                                                flat("rnn_states_critic"), actions.float(), flat("masks"),
                                                flat("available_actions"), flat("active_masks"))
    torch.testing.assert_close(logp, ev_logp, rtol=1e-4, atol=2e-5)


def test_whole_batch_views_equal_the_gathered_tuple_and_train_identically(gold, monkeypatch):
    """Round 6: on the trainer's private route (lazy_obs, one minibatch, device sampler) the 12-tuple is VIEWS of the buffer
    fields + the normalised advantages -- no record pack, no gather.  Every element must equal what the gather produces
    (MAPPO_WHOLE_BATCH_VIEWS=0) bit for bit, and a whole train() must end with identical weights."""
    from onpolicy.algorithms.utils.fused_mlp import RowSource
    dev = torch.device("cuda", 0)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MAPPO_WHOLE_BATCH_VIEWS", mode)
        z, key, meta, spec, policy, trainer, buf = _setup(gold, "trainer_h64_cases", "h64_ns", dev)
        buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
        adv = buf.normalized_advantages(trainer.value_normalizer)
        tup = next(iter(buf.feed_forward_generator(adv, 1, standardize_obs=True, lazy_obs=True)))
        again = next(iter(buf.feed_forward_generator(adv, 1, standardize_obs=True, lazy_obs=True)))
        assert all(a is b for a, b in zip(tup, again)) and buf.whole_batch_reuses == 1
        if mode == "1":     # zero-copy: the returns of the tuple ARE the buffer's
            assert tup[6].data_ptr() == buf.returns.data_ptr() and tup[10].data_ptr() != buf.advantages.data_ptr()
        fields = [x.materialize().clone() if isinstance(x, RowSource) else (None if x is None else x.clone()) for x in tup]
        trainer.prep_training()
        torch.manual_seed(21)
        trainer.train(buf)
        w = torch.cat([p.detach().reshape(-1) for net in (policy.actor, policy.critic) for p in net.parameters()]).clone()
        out[mode] = (fields, w)
    for a, b in zip(out["1"][0], out["0"][0]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert torch.equal(out["1"][1], out["0"][1])
