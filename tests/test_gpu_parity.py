"""-m gpu: the HIP path (through the device buffer, i.e. through the C ABI) against
  * the golden fixtures produced by the reference itself, and
  * the plain-C oracle on seeded inputs, up to the north-star size.
Bar: bit-exact for returns / value_preds / advantages / every gathered field / integer indices;
float32 tolerance (stated per assert) only where the reference reduces in float32 pairwise sums
(the two advantage moments) or where the maths runs through rocBLAS (trainer)."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args, fill_buffer_arrays, buffer_shapes, load_into
from oracle import oracle

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


def _native_lib():
    from onpolicy import _native
    return _native.lib()


def _vn(norm_triplet=None):
    from onpolicy.utils.valuenorm import ValueNorm
    vn = ValueNorm(1, device=_dev())
    if norm_triplet is not None:
        vn.running_mean.fill_(float(norm_triplet[0]))
        vn.running_mean_sq.fill_(float(norm_triplet[1]))
        vn.debiasing_term.fill_(float(norm_triplet[2]))
    return vn


def _buffer(args, A, Do=3, Ds=4, na=5):
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    return SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(na), device=_dev())


def test_native_library_is_loaded():
    lib = _native_lib()
    assert b"gfx950" in lib.mappo_build_info()
    assert "libmappo_hip" in open("/proc/self/maps").read()


# ------------------------------------------------------------------ K1 vs reference fixtures
def test_gae_kats(gold):
    z = gold.npz("kat_returns")
    kw = {"A": {}, "B": {}, "C": dict(use_valuenorm=False),
          "D": dict(use_valuenorm=False, use_proper_time_limits=True),
          "E": dict(use_valuenorm=False, use_gae=False)}
    for name in "ABCDE":
        args = make_args(episode_length=4, n_rollout_threads=1, **kw[name])
        buf = _buffer(args, 1)
        buf.rewards[:, 0, 0, 0] = torch.tensor([1., 2., 3., -1.])
        buf.value_preds[:4, 0, 0, 0] = torch.tensor([0.5, 0.4, 0.3, 0.25])
        buf.masks[:, 0, 0, 0] = torch.tensor([1., 1., 0., 1., 1.])
        buf.bad_masks[:, 0, 0, 0] = torch.tensor([1., 1., 1., 1., 0.])
        vn = _vn(z["kat_%s_norm" % name]) if name in "AB" else None
        buf.compute_returns(np.array([[[0.2]]], dtype=np.float32), vn)
        np.testing.assert_array_equal(buf.returns[:, 0, 0, 0].cpu().numpy(), z["kat_%s_returns" % name])


@pytest.mark.parametrize("variant", [0, 99])
def test_gae_matrix_vs_reference(gold, variant):
    """All 7 reference branches x normaliser states x ragged shapes, fused advantages included."""
    z = gold.npz("returns_cases")
    lib = _native_lib()
    old = lib.mappo_gae_set_variant(variant)
    try:
        for m in gold.meta("returns_cases"):
            key = "ret%03d_" % m["id"]
            args = make_args(episode_length=m["T"], n_rollout_threads=m["N"], use_gae=m["use_gae"],
                             use_proper_time_limits=m["use_proper_time_limits"],
                             use_valuenorm=m["use_valuenorm"])
            buf = _buffer(args, m["A"])
            for name, fk in (("rewards", "rewards"), ("value_preds", "value_preds_in"), ("masks", "masks"),
                             ("bad_masks", "bad_masks"), ("active_masks", "active_masks")):
                getattr(buf, name).copy_(torch.from_numpy(z[key + fk]))
            vn = _vn(z[key + "norm"]) if m["use_valuenorm"] else None
            buf.compute_returns(z[key + "next_value"], vn)
            np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"], err_msg=str(m))
            np.testing.assert_array_equal(buf.value_preds.cpu().numpy(), z[key + "value_preds_out"], err_msg=str(m))
            np.testing.assert_array_equal(buf.advantages.cpu().numpy(), z[key + "advantages"], err_msg=str(m))
            handle = buf.normalized_advantages(vn)
            mean, std = handle.stats.cpu().numpy()
            gm, gs = z[key + "adv_mean_std"]
            # numpy reduces in float32 pairwise sums, the kernel in float64: a few float32 ulp
            assert abs(mean - gm) <= 2e-6 * max(1.0, abs(gm)) + 1e-7, m
            if np.isfinite(gs):
                assert abs(std - gs) <= 2e-6 * max(1.0, abs(gs)) + 1e-7, m
                np.testing.assert_allclose(handle.materialize().cpu().numpy(), z[key + "advantages_normed"],
                                           rtol=2e-5, atol=2e-6, err_msg=str(m))
    finally:
        lib.mappo_gae_set_variant(old)


# ------------------------------------------------------------------ K1 vs oracle, every variant
def _random_case(T, N, A, seed, device):
    rng = np.random.default_rng(seed)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, 3, 4, 5, 8), rng, na=5, p_mask=1.0 - 1.0 / 25)
    return arrays


@pytest.mark.parametrize("variant", [2, 6, 20, 33, 34, 36, 42, 51, 54, 56, 57, 3033, 3042, 3057, 2057, 1054, 99])
@pytest.mark.parametrize("ptl,norm", [(False, True), (True, True), (False, False), (True, False)])
def test_gae_variants_vs_oracle(variant, ptl, norm):
    """Every kernel variant is bit-identical to the oracle, including ragged strips (C % W != 0),
    T not a multiple of the tile length and T smaller than a tile."""
    lib = _native_lib()
    old = lib.mappo_gae_set_variant(variant)
    try:
        for (T, N, A) in [(50, 33, 4), (7, 5, 4), (129, 16, 12), (64, 300, 8)]:
            args = make_args(episode_length=T, n_rollout_threads=N, use_proper_time_limits=ptl,
                             use_valuenorm=norm)
            arrays = _random_case(T, N, A, 7 + T, _dev())
            buf = _buffer(args, A)
            load_into(buf, arrays)
            vn = None
            sigma, mu = 1.0, 0.0
            if norm:
                vn = _vn([0.7e-4, 3.1e-4, 2.5e-5])
                sigma, mu = oracle.normalizer_scalars(0.7e-4, 3.1e-4, 2.5e-5)
                got = vn.denorm_scalars().cpu().numpy()
                np.testing.assert_allclose(got, [sigma, mu], rtol=1e-6)
                sigma, mu = float(got[0]), float(got[1])  # the kernel's own scalars
            buf.compute_returns(arrays["next_value"], vn)
            ret, v = oracle.compute_returns(arrays["rewards"], arrays["value_preds"], arrays["next_value"],
                                            arrays["masks"], arrays["bad_masks"], sigma=sigma, mu=mu,
                                            use_proper_time_limits=ptl, denorm=norm)
            np.testing.assert_array_equal(buf.returns.cpu().numpy(), ret)
            np.testing.assert_array_equal(buf.value_preds.cpu().numpy(), v)
            adv = oracle.advantages(ret, v, sigma=sigma, mu=mu, denorm=norm)
            np.testing.assert_array_equal(buf.advantages.cpu().numpy(), adv)
            mean, std, cnt = oracle.adv_moments(adv, arrays["active_masks"][:-1])
            h = buf.normalized_advantages(vn)
            assert float(buf._adv_sums[2]) == cnt
            np.testing.assert_allclose(h.stats.cpu().numpy(), [mean, std], rtol=1e-6, atol=1e-7)
    finally:
        lib.mappo_gae_set_variant(old)


def test_gae_north_star_size_vs_oracle():
    """T=400, N=4096, A=8 (13.1 M elements): bit-exact against the C oracle, plus the linearity
    property returns(r1 + r2 with zero values) consistency via an independent recomputation."""
    T, N, A = 400, 4096, 8
    C = N * A
    rng = np.random.default_rng(0)
    f32 = np.float32
    arrays = dict(rewards=rng.standard_normal((T, N, A, 1), dtype=f32),
                  value_preds=np.concatenate([rng.standard_normal((T, N, A, 1), dtype=f32),
                                              np.zeros((1, N, A, 1), f32)]),
                  masks=(rng.random((T + 1, N, A, 1)) >= 1.0 / 25).astype(f32),
                  active_masks=np.ones((T + 1, N, A, 1), f32), bad_masks=np.ones((T + 1, N, A, 1), f32))
    nv = rng.standard_normal((N, A, 1), dtype=f32)
    dev = _dev()
    from onpolicy import _native
    lib = _native.lib()
    t = {k: torch.from_numpy(v).to(dev) for k, v in arrays.items()}
    ret = torch.zeros((T + 1, N, A, 1), device=dev)
    adv = torch.zeros((T, N, A, 1), device=dev)
    rows = lib.mappo_gae_partial_rows(C)
    partials = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
    vn = _vn([0.7e-4, 3.1e-4, 2.5e-5])
    den = vn.denorm_scalars().contiguous()
    sigma, mu = [float(x) for x in den.cpu()]
    nvd = torch.from_numpy(nv).to(dev)
    exp_ret, exp_v = oracle.compute_returns(arrays["rewards"], arrays["value_preds"], nv, arrays["masks"],
                                            sigma=sigma, mu=mu, denorm=True)
    for variant in (0, 2, 20, 33, 57, 99):
        lib.mappo_gae_set_variant(variant)
        ret.zero_()
        code = lib.mappo_gae_f32(t["rewards"].data_ptr(), t["value_preds"].data_ptr(), nvd.data_ptr(),
                                 t["masks"].data_ptr(), None, ret.data_ptr(), den.data_ptr(), adv.data_ptr(),
                                 t["active_masks"].data_ptr(), partials.data_ptr(), T, C, 0.99, 0.95,
                                 1 | 4, torch.cuda.current_stream().cuda_stream)
        assert code == 0
        torch.cuda.synchronize()
        assert np.array_equal(ret.cpu().numpy(), exp_ret), variant
        assert np.array_equal(t["value_preds"].cpu().numpy(), exp_v), variant
    lib.mappo_gae_set_variant(0)
    # moments: count is exact, sums agree with a float64 numpy reduction
    sums = torch.zeros(3, dtype=torch.float64, device=dev)
    assert lib.mappo_adv_reduce(partials.data_ptr(), rows, sums.data_ptr(), None) == 0
    torch.cuda.synchronize()
    a64 = adv.cpu().numpy().astype(np.float64)
    assert float(sums[2]) == a64.size
    np.testing.assert_allclose(float(sums[0]), a64.sum(), rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(float(sums[1]), (a64 ** 2).sum(), rtol=1e-9)


# ------------------------------------------------------------------ K3 / K4 vs reference fixtures
FIELDS = ["share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
          "masks", "active_masks", "old_action_log_probs", "adv_targ", "available_actions"]


def _gen_buffer(z, recurrent=True):
    sh = z["gen_buf_share_obs"].shape
    T, N, A = sh[0] - 1, sh[1], sh[2]
    H = z["gen_buf_rnn_states"].shape[-1]
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, use_recurrent_policy=recurrent,
                     sampler_rng="host")
    buf = _buffer(args, A, Do=z["gen_buf_obs"].shape[-1], Ds=sh[-1], na=z["gen_buf_available_actions"].shape[-1])
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
                 "masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:      # a feed-forward buffer keeps no RNN-state storage
            dst.copy_(torch.from_numpy(z["gen_buf_" + name]))
    return buf


@pytest.mark.parametrize("case,call", [
    ("ff2", lambda b, a: b.feed_forward_generator(a, 2)),
    ("ff7", lambda b, a: b.feed_forward_generator(a, 7)),
    ("rec_L5", lambda b, a: b.recurrent_generator(a, 2, 5)),
    ("rec_L4", lambda b, a: b.recurrent_generator(a, 3, 4)),
    ("naive3", lambda b, a: b.naive_recurrent_generator(a, 3)),
])
def test_generators_vs_reference(gold, case, call):
    """Same CPU seed => same permutation (integer parity) => bit-identical 12-tuples."""
    z = gold.npz("generator_cases")
    buf = _gen_buffer(z)
    torch.manual_seed(5)
    batches = list(call(buf, z["gen_buf_advantages"]))
    n = [m for m in gold.meta("generator_cases") if m.get("case") == case][0]["n_batches"]
    assert len(batches) == n
    for bi, sample in enumerate(batches):
        assert len(sample) == 12
        for fname, t in zip(FIELDS, sample):
            exp = z["gen_%s_b%d_%s" % (case, bi, fname)]
            assert t.device.type == "cuda" and tuple(t.shape) == exp.shape, (fname, t.shape, exp.shape)
            np.testing.assert_array_equal(t.cpu().numpy(), exp, err_msg="%s b%d %s" % (case, bi, fname))


def test_feed_forward_lazy_rnn_state(gold):
    """A feed-forward buffer stores no RNN state and yields zeros of the right shape."""
    z = gold.npz("generator_cases")
    buf = _gen_buffer(z, recurrent=False)
    assert buf.rnn_states.stride()[0] == 0
    torch.manual_seed(5)
    sample = next(iter(buf.feed_forward_generator(z["gen_buf_advantages"], 2)))
    assert tuple(sample[2].shape) == z["gen_ff2_b0_rnn_states"].shape
    assert float(sample[2].abs().sum()) == 0.0


def test_gather_fused_normalisation_and_device_rng():
    """AdvantageHandle path: the gathers apply (adv - mean) / (std + 1e-5) exactly like the
    standalone pass, and the device permutation covers every row exactly once."""
    T, N, A = 20, 16, 3
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8)
    buf = _buffer(args, A, Do=6, Ds=18, na=5)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, 6, 18, 5, 8), np.random.default_rng(3), na=5)
    load_into(buf, arrays)
    vn = _vn([0.7e-4, 3.1e-4, 2.5e-5])
    buf.compute_returns(arrays["next_value"], vn)
    handle = buf.normalized_advantages(vn)
    full = handle.materialize().reshape(-1, 1)
    mean, std = [np.float32(x) for x in handle.stats.cpu().numpy()]
    exp = oracle.adv_normalize(buf.advantages.cpu().numpy(), mean, std).reshape(-1, 1)
    np.testing.assert_array_equal(full.cpu().numpy(), exp)
    # tag every row with its flat index so that the permutation can be read back
    B = T * N * A
    buf.rewards.copy_(torch.arange(B, dtype=torch.float32, device=buf.device).reshape(T, N, A, 1))
    buf.action_log_probs.copy_(buf.rewards)
    seen = []
    for sample in buf.feed_forward_generator(handle, 4):
        idx = sample[9].reshape(-1).long()       # old_action_log_probs carries the row id
        seen.append(idx)
        np.testing.assert_array_equal(sample[10].cpu().numpy(), full[idx].cpu().numpy())
        np.testing.assert_array_equal(sample[0].cpu().numpy(), buf.share_obs[:-1].reshape(B, -1)[idx].cpu().numpy())
    allidx = torch.cat(seen).cpu().numpy()
    assert np.array_equal(np.sort(allidx), np.arange(B))
    # later epochs: each a fresh, complete partition (device sampler K10, tests/test_gpu_sampler_indices.py)
    epochs = [allidx]
    buf.plan_epochs(3)
    for _ in range(3):
        order = torch.cat([smp[9].reshape(-1).long() for smp in buf.feed_forward_generator(handle, 4)]).cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(B))
        assert all(not np.array_equal(order, prev) for prev in epochs)
        epochs.append(order)


def test_gather_wide_odd_rows_vs_oracle():
    """Row widths that force the 8-byte and 4-byte access paths (SMAC 370 / 435, Hanabi 1285),
    chunk gather straddling trajectories, against the oracle."""
    T, N, A, L = 12, 6, 5, 8   # T % L != 0
    rng = np.random.default_rng(11)
    for Do, Ds in [(370, 435), (1285, 1385), (18, 54)]:
        args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8, use_recurrent_policy=True,
                         sampler_rng="host")
        buf = _buffer(args, A, Do=Do, Ds=Ds, na=7)
        arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, 7, 8), rng, na=7)
        load_into(buf, arrays)
        adv = rng.standard_normal((T, N, A, 1)).astype(np.float32)
        ob = oracle.OracleBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(7))
        load_into(ob, arrays)
        for gen in ("ff", "rec"):
            torch.manual_seed(9)
            got = list(buf.feed_forward_generator(adv, 3) if gen == "ff" else buf.recurrent_generator(adv, 2, L))
            torch.manual_seed(9)
            exp = list(ob.feed_forward_generator(adv, 3) if gen == "ff" else ob.recurrent_generator(adv, 2, L))
            assert len(got) == len(exp)
            for g, e in zip(got, exp):
                for fname, a, b in zip(FIELDS, g, e):
                    np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg="%s %s Do=%d" % (gen, fname, Do))


def test_gather_large_permutation_property():
    """1.0 M rows x (384 + 48 + scalars): a gather by a permutation preserves the multiset of rows;
    checked through per-field checksums and the inverse permutation."""
    T, N, A = 128, 1024, 8
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8)
    buf = _buffer(args, A, Do=48, Ds=384, na=5)
    g = torch.Generator(device=buf.device)
    g.manual_seed(1)
    buf.share_obs.normal_(generator=g)
    buf.obs.normal_(generator=g)
    buf.returns.normal_(generator=g)
    B = T * N * A
    buf.action_log_probs.copy_(torch.arange(B, dtype=torch.float32, device=buf.device).reshape(T, N, A, 1))
    sample = next(iter(buf.feed_forward_generator(None, 1)))
    idx = sample[9].reshape(-1).long()
    assert torch.equal(torch.sort(idx).values, torch.arange(B, device=buf.device))
    assert sample[10] is None
    assert torch.equal(sample[0], buf.share_obs[:-1].reshape(B, -1)[idx])
    assert torch.equal(sample[1], buf.obs[:-1].reshape(B, -1)[idx])
    assert torch.equal(sample[6], buf.returns[:-1].reshape(B, -1)[idx])
    np.testing.assert_allclose(float(sample[0].double().sum()), float(buf.share_obs[:-1].double().sum()), rtol=1e-9)


# ------------------------------------------------------------------ K2
def test_insert_and_after_update_vs_oracle():
    T, N, A, H = 5, 4, 3, 8
    for recurrent in (True, False):
        args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, use_recurrent_policy=recurrent)
        buf = _buffer(args, A, Do=6, Ds=18, na=5)
        ob = oracle.OracleBuffer(args, A, Box((6,)), Box((18,)), Discrete(5))
        rng = np.random.default_rng(2)
        f = lambda *s: rng.standard_normal(s).astype(np.float32)
        for step in range(2 * T + 1):
            data = dict(share_obs=f(N, A, 18), obs=f(N, A, 6),
                        rnn_states_actor=f(N, A, 1, H) if recurrent else np.zeros((N, A, 1, H), np.float32),
                        rnn_states_critic=f(N, A, 1, H) if recurrent else np.zeros((N, A, 1, H), np.float32),
                        actions=f(N, A, 1), action_log_probs=f(N, A, 1), value_preds=f(N, A, 1),
                        rewards=f(N, A, 1), masks=f(N, A, 1))
            extra = dict(bad_masks=f(N, A, 1), active_masks=f(N, A, 1), available_actions=f(N, A, 5)) \
                if step % 2 else {}
            if step % 3 == 0:   # tensors already on the device are accepted as well
                dev_data = {k: torch.from_numpy(v).to(buf.device) for k, v in data.items()}
                buf.insert(**dev_data, **extra)
            else:
                buf.insert(**data, **extra)
            ob.insert(**data, **extra)
            if buf.step == 0:
                buf.after_update()
                ob.after_update()
            assert buf.step == ob.step
        for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs",
                     "value_preds", "rewards", "masks", "bad_masks", "active_masks", "available_actions"):
            np.testing.assert_array_equal(getattr(buf, name).cpu().numpy(), getattr(ob, name), err_msg=name)


# ------------------------------------------------------------------ trainer end to end on the GPU
@pytest.mark.parametrize("cname", ["mlp", "mlp_relu", "gru", "mlp_nonorm"])
def test_train_on_device_vs_reference(gold, cname):
    """compute_returns + R_MAPPO.train on the device buffer with the reference's permutations
    (sampler_rng=host, same CPU seed).  rocBLAS / device transcendental rounding differs from the
    CPU reference, so losses are compared to 1e-3 relative and weights to 5e-5 absolute (one Adam
    step moves a weight by ~lr = 5e-4..7e-4)."""
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    z = gold.npz("trainer_cases")
    key = "trn_%s_" % cname
    meta = gold.meta("trainer_cases")[cname]
    spec = meta["spec"]
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], sampler_rng="host", **spec["args"])
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    dev = _dev()
    torch.manual_seed(1)
    np.random.seed(1)
    policy = R_MAPPOPolicy(args, *spaces, device=dev)
    trainer = R_MAPPO(args, policy, device=dev)
    # init draws come from the CPU generator on every device; orthogonal_ runs a host LAPACK QR whose
    # rounding depends on the host CPU / BLAS build, hence 1e-6 instead of bit equality across boxes
    for k, v in policy.actor.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), z[key + "init_actor." + k], rtol=1e-5, atol=1e-6)
    buf = SharedReplayBuffer(args, spec["A"], *spaces, device=dev)
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks",
                 "bad_masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"])
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    buf.after_update()
    for k, ref in meta["train_info"].items():
        assert info[k] == pytest.approx(ref, rel=1e-3, abs=1e-5), (k, info[k], ref)
    for net, prefix in ((policy.actor, "final_actor."), (policy.critic, "final_critic.")):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), z[key + prefix + k], rtol=1e-3, atol=5e-5, err_msg=k)
    if trainer.value_normalizer is not None:
        vn = trainer.value_normalizer
        got = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        np.testing.assert_allclose(got, z[key + "final_norm"], rtol=1e-5, atol=1e-9)


def test_train_with_observations_wider_than_the_standardising_gather():
    """simple_spread with 19 agents under a centralised critic has share_obs 2166 wide: no standardising-gather
    kernel covers it (and FusedLayerNorm falls back to PyTorch there).  train() must then gather plain rows and
    let the policy apply its own feature_norm instead of dying in the first minibatch; the update must equal
    the same update on the CPU port of the path (oracle buffer + the same trainer)."""
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A, Do, Ds, na = 6, 3, 2, 114, 2166, 5
    assert not SharedReplayBuffer._standardize_width_ok(Ds) and SharedReplayBuffer._standardize_width_ok(Do)
    args = make_args(episode_length=T, n_rollout_threads=N, ppo_epoch=2, num_mini_batch=1, sampler_rng="host")
    spaces = Box((Do,)), Box((Ds,)), Discrete(na)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, na, 64), np.random.default_rng(5), na=na)
    results = []
    for dev in (_dev(), torch.device("cpu")):
        torch.manual_seed(1)
        policy = R_MAPPOPolicy(args, *spaces, device=dev)
        trainer = R_MAPPO(args, policy, device=dev)
        buf = SharedReplayBuffer(args, A, *spaces, device=dev) if dev.type == "cuda" \
            else oracle.OracleBuffer(args, A, *spaces)
        if dev.type == "cuda":
            assert not buf.can_standardize_obs() and policy.can_fold_input_norm()
        load_into(buf, arrays)
        buf.compute_returns(arrays["next_value"], trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(33)
        info = trainer.train(buf)
        results.append((info, {k: v.detach().cpu().numpy() for k, v in policy.critic.state_dict().items()}))
    (gi, gw), (ci, cw) = results
    for k in ci:
        assert gi[k] == pytest.approx(ci[k], rel=1e-3, abs=1e-5), k
    for k in cw:
        np.testing.assert_allclose(gw[k], cw[k], rtol=1e-3, atol=5e-5, err_msg=k)


# ------------------------------------------------------------------ K6: LayerNorm kernels
@pytest.mark.parametrize("D", [48, 64, 384, 18, 54, 370, 435, 512, 1285, 30, 150, 4, 3, 2048, 1536])
def test_fused_layernorm_vs_torch(D):
    """Forward and backward of the HIP LayerNorm against torch's float32 LayerNorm on the same
    device, both judged against a float64 evaluation: the HIP kernel's error must be within
    4x torch's own float32 error + 1e-5 relative to the tensor's scale (rows with a tiny variance
    amplify rounding in ANY float32 implementation, so a fixed tolerance between the two float32
    results would be meaningless for small D)."""
    from onpolicy.algorithms.utils.fused_norm import FusedLayerNorm
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(D)

    def close(mine, theirs, exact, what):
        scale = float(exact.abs().max()) + 1e-30
        e_mine = float((mine.double() - exact).abs().max()) / scale
        e_torch = float((theirs.double() - exact).abs().max()) / scale
        assert e_mine <= 4 * e_torch + 1e-5, (what, D, e_mine, e_torch)

    for M in (1, 7, 1000, 70001):
        ln = FusedLayerNorm(D).to(dev)
        ref = torch.nn.LayerNorm(D).to(dev)
        ref64 = torch.nn.LayerNorm(D).to(dev).double()
        with torch.no_grad():
            ln.weight.copy_(torch.randn(D, device=dev, generator=g))
            ln.bias.copy_(torch.randn(D, device=dev, generator=g))
            ref.weight.copy_(ln.weight)
            ref.bias.copy_(ln.bias)
            ref64.weight.copy_(ln.weight.double())
            ref64.bias.copy_(ln.bias.double())
        for need_dx in (True, False):
            x = (torch.randn(M, D, device=dev, generator=g) * 3 + 1).requires_grad_(need_dx)
            xr = x.detach().clone().requires_grad_(need_dx)
            x64 = x.detach().double().requires_grad_(need_dx)
            dy = torch.randn(M, D, device=dev, generator=g)
            y, yr, y64 = ln(x), ref(xr), ref64(x64)
            close(y, yr, y64.detach(), "y")
            y.backward(dy)
            yr.backward(dy)
            y64.backward(dy.double())
            if need_dx:
                close(x.grad, xr.grad, x64.grad, "dx")
            close(ln.weight.grad, ref.weight.grad, ref64.weight.grad, "dw")
            close(ln.bias.grad, ref.bias.grad, ref64.bias.grad, "db")
            for m in (ln, ref, ref64):
                m.zero_grad()
    # 3-d input (the GRU path normalises [L*B, H] but keep the general case right)
    x = torch.randn(5, 9, D, device=dev, generator=g)
    close(ln(x), ref(x), ref64(x.double()).detach(), "y3d")


def test_tall_linear_backward_vs_torch():
    """Split-K weight / bias gradients against torch's own Linear backward (float64 as judge)."""
    from onpolicy.algorithms.utils.tall_linear import TallLinear
    dev = _dev()
    torch.manual_seed(3)
    for M, K, N in [(70001, 48, 64), (262144, 64, 5), (100000, 384, 64), (65536, 64, 1)]:
        lin = TallLinear(K, N).to(dev)
        ref = torch.nn.Linear(K, N).to(dev)
        ref64 = torch.nn.Linear(K, N).to(dev).double()
        with torch.no_grad():
            ref.weight.copy_(lin.weight)
            ref.bias.copy_(lin.bias)
            ref64.weight.copy_(lin.weight.double())
            ref64.bias.copy_(lin.bias.double())
        x = torch.randn(M, K, device=dev, requires_grad=True)
        xr = x.detach().clone().requires_grad_(True)
        x64 = x.detach().double().requires_grad_(True)
        dy = torch.randn(M, N, device=dev)
        y, yr, y64 = lin(x), ref(xr), ref64(x64)
        assert torch.equal(y, yr)
        y.backward(dy)
        yr.backward(dy)
        y64.backward(dy.double())
        for mine, theirs, exact, what in ((x.grad, xr.grad, x64.grad, "dx"),
                                          (lin.weight.grad, ref.weight.grad, ref64.weight.grad, "dw"),
                                          (lin.bias.grad, ref.bias.grad, ref64.bias.grad, "db")):
            scale = float(exact.abs().max())
            e_mine = float((mine.double() - exact).abs().max()) / scale
            e_torch = float((theirs.double() - exact).abs().max()) / scale
            assert e_mine <= 4 * e_torch + 1e-5, (what, M, K, N, e_mine, e_torch)


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_fused_act_layernorm_block_vs_torch(act):
    """DenseBlock (Linear -> act -> LayerNorm with the activation fused into the LayerNorm kernels)
    against the plain torch modules, float64 as judge."""
    from onpolicy.algorithms.utils.fused_norm import FusedLayerNorm, DenseBlock
    dev = _dev()
    torch.manual_seed(5)
    A = torch.nn.Tanh if act == "tanh" else torch.nn.ReLU
    # (.., 30): scalar-unit kernels; (.., 1100): too wide for the fused Linear-bias gradient -> plain bias
    for M, K, N in [(1000, 48, 64), (70001, 64, 64), (4097, 30, 512), (3001, 20, 30), (515, 16, 1100)]:
        blk = DenseBlock(torch.nn.Linear(K, N), A(), FusedLayerNorm(N)).to(dev)
        ref = torch.nn.Sequential(torch.nn.Linear(K, N), A(), torch.nn.LayerNorm(N)).to(dev)
        ref.load_state_dict(blk.state_dict())
        ref64 = torch.nn.Sequential(torch.nn.Linear(K, N), A(), torch.nn.LayerNorm(N)).to(dev).double()
        ref64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
        x = torch.randn(M, K, device=dev, requires_grad=True)
        xr = x.detach().clone().requires_grad_(True)
        x64 = x.detach().double().requires_grad_(True)
        dy = torch.randn(M, N, device=dev)
        y, yr, y64 = blk(x), ref(xr), ref64(x64)
        y.backward(dy)
        yr.backward(dy)
        y64.backward(dy.double())
        pairs = [(y, yr, y64.detach(), "y"), (x.grad, xr.grad, x64.grad, "dx")]
        for (n1, p1), (_, p2), (_, p3) in zip(blk.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
            pairs.append((p1.grad, p2.grad, p3.grad, n1))
        for mine, theirs, exact, what in pairs:
            scale = float(exact.abs().max())
            e_mine = float((mine.detach().double() - exact).abs().max()) / scale
            e_torch = float((theirs.detach().double() - exact).abs().max()) / scale
            assert e_mine <= 4 * e_torch + 2e-5, (what, act, M, e_mine, e_torch)


def test_gather_standardize_vs_torch():
    """standardize_obs=True: rows of share_obs / obs come out as (x - mean) / sqrt(var + 1e-5),
    for the vectorised and the odd-width paths, feed-forward and chunked."""
    T, N, A, L = 12, 6, 5, 4
    rng = np.random.default_rng(3)
    for Do, Ds in [(48, 384), (18, 54), (370, 435)]:
        args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8, use_recurrent_policy=True,
                         sampler_rng="host")
        buf = _buffer(args, A, Do=Do, Ds=Ds, na=7)
        arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, 7, 8), rng, na=7)
        load_into(buf, arrays)
        adv = rng.standard_normal((T, N, A, 1)).astype(np.float32)
        for gen in ("ff", "rec"):
            torch.manual_seed(9)
            plain = list(buf.feed_forward_generator(adv, 3) if gen == "ff" else buf.recurrent_generator(adv, 2, L))
            torch.manual_seed(9)
            std = list(buf.feed_forward_generator(adv, 3, standardize_obs=True) if gen == "ff"
                       else buf.recurrent_generator(adv, 2, L, standardize_obs=True))
            for p, s in zip(plain, std):
                for f in (0, 1):
                    x = p[f].double()
                    exp = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
                    torch.testing.assert_close(s[f].double(), exp, rtol=1e-5, atol=2e-5)
                for f in range(2, 12):      # every other field is untouched
                    if p[f] is not None:
                        assert torch.equal(p[f], s[f])


def test_chooseinsert_and_chooseafter_update_vs_oracle():
    """Turn-based (Hanabi) storage semantics: observations / active masks / available actions at row
    ``step`` (reference shared_buffer.py:125-158), chooseafter_update copies only states and masks."""
    T, N, A, H = 5, 4, 3, 8
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, use_recurrent_policy=True)
    buf = _buffer(args, A, Do=6, Ds=18, na=5)
    ob = oracle.OracleBuffer(args, A, Box((6,)), Box((18,)), Discrete(5))
    rng = np.random.default_rng(8)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    for step in range(2 * T):
        data = dict(share_obs=f(N, A, 18), obs=f(N, A, 6), rnn_states=f(N, A, 1, H), rnn_states_critic=f(N, A, 1, H),
                    actions=f(N, A, 1), action_log_probs=f(N, A, 1), value_preds=f(N, A, 1), rewards=f(N, A, 1),
                    masks=f(N, A, 1), bad_masks=f(N, A, 1), active_masks=f(N, A, 1), available_actions=f(N, A, 5))
        buf.chooseinsert(**data)
        ob.chooseinsert(**data)
        if buf.step == 0:
            buf.chooseafter_update()
            ob.chooseafter_update()
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs",
                 "value_preds", "rewards", "masks", "bad_masks", "active_masks", "available_actions"):
        np.testing.assert_array_equal(getattr(buf, name).cpu().numpy(), getattr(ob, name), err_msg=name)


def test_host_parity_action_sampling():
    """Integer-sampling parity mode: with the noise drawn on the CPU generator, device sampling gives
    exactly the actions torch.multinomial gives on the CPU for the same probabilities and seed."""
    from onpolicy.algorithms.utils import distributions as D
    dev = _dev()
    torch.manual_seed(3)
    logits = torch.randn(4096, 18)
    avail = (torch.rand(4096, 18) < 0.6).float()
    avail[:, 0] = 1
    logits = torch.where(avail == 0, torch.full_like(logits, -1e10), logits)
    cpu = D.FixedCategorical(logits)
    gpu = D.FixedCategorical(logits.to(dev))
    gpu._probs = cpu.probs.to(dev)          # identical probabilities on both sides
    D.set_sampling_rng("host")
    try:
        torch.manual_seed(77)
        a_cpu = cpu.sample()
        torch.manual_seed(77)
        a_gpu = gpu.sample()
    finally:
        D.set_sampling_rng("device")
    assert a_gpu.is_cuda and torch.equal(a_cpu, a_gpu.cpu())
    assert bool((torch.gather(avail, 1, a_cpu) == 1).all())


def test_gae_dispatch_timing_hook():
    """mappo_gae_time_next_launch / mappo_gae_timed_launch_ms (the measurement hook bench.py's roofline_gae uses): the armed
    launch is timed by events attached to its dispatch, its results are those of an unarmed launch, a slot that never
    launched reports MAPPO_E_FLAGS, and the duration is no longer than what an event pair around the same launch sees."""
    import ctypes
    import torch
    from onpolicy import _native
    lib = _native.lib()
    dev = torch.device("cuda", 0)
    T, C = 128, 32768
    g = torch.Generator(device=dev).manual_seed(11)
    rewards = torch.randn(T, C, device=dev, generator=g)
    values = torch.randn(T + 1, C, device=dev, generator=g)
    masks = (torch.rand(T + 1, C, device=dev, generator=g) > 0.05).float()
    nv = values[-1].clone()
    outs = []
    ms = ctypes.c_float(-1.0)
    for armed in (False, True):
        returns, adv = torch.zeros(T + 1, C, device=dev), torch.zeros(T, C, device=dev)
        partials = torch.zeros(lib.mappo_gae_partial_rows(C) * 4 + 64, dtype=torch.float64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        slot = lib.mappo_gae_time_next_launch() if armed else -1
        assert (slot >= 0) == armed
        torch.cuda._sleep(200000)
        e0.record()
        _native.check(lib.mappo_gae_f32(rewards.data_ptr(), values.data_ptr(), nv.data_ptr(), masks.data_ptr(), None,
                                        returns.data_ptr(), None, adv.data_ptr(), None, partials.data_ptr(), T, C, 0.99, 0.95,
                                        _native.GAE_USE_GAE, _native.stream_of(dev)), "mappo_gae_f32")
        e1.record()
        torch.cuda.synchronize()
        outs.append((returns, adv))
        if armed:
            assert lib.mappo_gae_timed_launch_ms(slot, ctypes.byref(ms)) == 0
            pair = e0.elapsed_time(e1)
            print("\n[GAE timing hook] dispatch %.2f us, event pair around the launch %.2f us" % (1e3 * ms.value, 1e3 * pair))
            assert 0.0 < ms.value <= pair * 1.02
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    slot = lib.mappo_gae_time_next_launch()                    # armed, never launched
    assert lib.mappo_gae_timed_launch_ms(slot, ctypes.byref(ms)) == -3
    assert lib.mappo_gae_timed_launch_ms(64, ctypes.byref(ms)) == -2 and lib.mappo_gae_timed_launch_ms(slot, None) == -1
    # (the armed slot is disarmed by the query above: the next launch is an ordinary one)
    returns, adv = torch.zeros(T + 1, C, device=dev), torch.zeros(T, C, device=dev)
    partials = torch.zeros(lib.mappo_gae_partial_rows(C) * 4 + 64, dtype=torch.float64, device=dev)
    _native.check(lib.mappo_gae_f32(rewards.data_ptr(), values.data_ptr(), nv.data_ptr(), masks.data_ptr(), None,
                                    returns.data_ptr(), None, adv.data_ptr(), None, partials.data_ptr(), T, C, 0.99, 0.95,
                                    _native.GAE_USE_GAE, _native.stream_of(dev)), "mappo_gae_f32")
    torch.cuda.synchronize()
    assert lib.mappo_gae_timed_launch_ms(slot, ctypes.byref(ms)) == -3 and torch.equal(returns, outs[0][0])


def test_gae_properties_at_scale():
    """Size-independent properties of the scan on a large buffer:
    (1) with all masks 1 and lambda = 1, returns are the discounted reward-to-go plus the discounted
        bootstrap value, whatever the value predictions are (float64 closed form, 1e-4 relative);
    (2) permuting the rollout threads permutes the outputs bit for bit (columns are independent)."""
    T, N, A = 200, 1024, 8
    dev = _dev()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    args = make_args(episode_length=T, n_rollout_threads=N, gae_lambda=1.0, use_valuenorm=False)
    buf = _buffer(args, A)
    buf.rewards.normal_(generator=g)
    buf.value_preds.normal_(generator=g)
    nv = torch.randn(N, A, 1, device=dev, generator=g)
    buf.compute_returns(nv, None)
    gamma = args.gamma
    ret64 = torch.zeros(T + 1, N, A, 1, dtype=torch.float64, device=dev)
    ret64[T] = nv.double()
    for t in range(T - 1, -1, -1):
        ret64[t] = buf.rewards[t].double() + gamma * ret64[t + 1]
    torch.testing.assert_close(buf.returns[:T].double(), ret64[:T], rtol=1e-4, atol=1e-4)

    # (2) thread permutation equivariance, default flags (valuenorm on), random masks
    args = make_args(episode_length=T, n_rollout_threads=N)
    a, b = _buffer(args, A), _buffer(args, A)
    perm = torch.randperm(N, device=dev, generator=g)
    for name in ("rewards", "value_preds", "masks", "active_masks"):
        src = getattr(a, name)
        if name in ("masks", "active_masks"):
            src.copy_((torch.rand(src.shape, device=dev, generator=g) > 0.05).float())
        else:
            src.normal_(generator=g)
        getattr(b, name).copy_(src[:, perm])
    vn = _vn([0.7e-4, 3.1e-4, 2.5e-5])
    a.compute_returns(nv, vn)
    b.compute_returns(nv[perm], vn)
    assert torch.equal(a.returns[:, perm], b.returns)
    assert torch.equal(a.advantages[:, perm], b.advantages)
    sa = a.normalized_advantages(vn).stats.clone()
    sb = b.normalized_advantages(vn).stats.clone()
    torch.testing.assert_close(sa, sb, rtol=1e-6, atol=1e-7)


def test_packed_record_cache_invalidation():
    """The per-epoch record pack is cached; any write to a packed field -- through the buffer's
    kernels or through plain torch in-place ops -- must invalidate it."""
    T, N, A = 6, 4, 2
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=8, sampler_rng="host")
    buf = _buffer(args, A, Do=5, Ds=9, na=4)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, 5, 9, 4, 8), np.random.default_rng(1), na=4)
    load_into(buf, arrays)
    adv = torch.zeros(T, N, A, 1, device=buf.device)
    B = T * N * A

    def first_sample():
        torch.manual_seed(3)
        return next(iter(buf.feed_forward_generator(adv, 1)))
    s0 = first_sample()
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    torch.manual_seed(3)
    idx = torch.randperm(B).to(buf.device)
    assert torch.equal(s0[6], buf.returns[:-1].reshape(B, 1)[idx])
    buf.returns[2] = 7.0                                     # torch in-place write
    s1 = first_sample()
    assert torch.equal(s1[6], buf.returns[:-1].reshape(B, 1)[idx]) and float(s1[6].max()) == 7.0
    buf.compute_returns(arrays["next_value"], _vn())         # kernel write
    s2 = first_sample()
    assert torch.equal(s2[6], buf.returns[:-1].reshape(B, 1)[idx])
    buf.insert(**{k: np.zeros_like(v) for k, v in dict(
        share_obs=arrays["share_obs"][0], obs=arrays["obs"][0], rnn_states_actor=arrays["rnn_states"][0],
        rnn_states_critic=arrays["rnn_states_critic"][0], actions=arrays["actions"][0],
        action_log_probs=arrays["action_log_probs"][0], value_preds=arrays["value_preds"][0] + 5,
        rewards=arrays["rewards"][0], masks=arrays["masks"][0]).items()})
    s3 = first_sample()
    assert torch.equal(s3[5], buf.value_preds[:-1].reshape(B, 1)[idx])


def test_gru_layer_device_vs_cpu():
    """RNNLayer on the GPU (fused gate kernel + split-K weight gradients) against the explicit CPU
    cell with the same weights: outputs, final state and every gradient within float32 tolerance,
    for the rollout (one step) and the update (L-step chunks with mask resets) call shapes."""
    from onpolicy.algorithms.utils.rnn import RNNLayer
    torch.manual_seed(2)
    from onpolicy.algorithms.utils import rnn as rnn_mod
    for H, R, L, B in [(64, 1, 6, 140001), (30, 2, 4, 1000)]:      # (64 at > 131 k rows: one-kernel forward step; 30: 4-byte path, 2 layers)
        _gru_case(RNNLayer, H, R, L, B)
    # below the row threshold H = 64 runs cell kernel + library GEMM; also force the one-kernel step on few rows
    _gru_case(RNNLayer, 64, 1, 5, 33333)
    old, rnn_mod._FUSED_STEP_MIN_ROWS = rnn_mod._FUSED_STEP_MIN_ROWS, 0
    try:
        _gru_case(RNNLayer, 64, 1, 3, 1001)
    finally:
        rnn_mod._FUSED_STEP_MIN_ROWS = old


def _gru_case(RNNLayer, H, R, L, B):
    cpu = RNNLayer(H, H, R, True)
    with torch.no_grad():
        for name, prm in cpu.rnn.named_parameters():
            if "bias" in name:
                prm.normal_(0.0, 0.3)       # the reference initialises them to 0; exercise their placement
    gpu = RNNLayer(H, H, R, True).to(_dev())
    gpu.load_state_dict(cpu.state_dict())
    x = torch.randn(L * B, H)
    h0 = torch.randn(B, R, H)
    masks = (torch.rand(L * B, 1) > 0.1).float()
    outs = []
    for layer, dev in ((cpu, "cpu"), (gpu, _dev())):
        xi = x.detach().clone().to(dev).requires_grad_(True)
        hi = h0.detach().clone().to(dev).requires_grad_(True)
        y, hT = layer(xi, hi, masks.to(dev))
        (y.sum() + (hT ** 2).sum()).backward()
        outs.append((y.detach().cpu(), hT.detach().cpu(), xi.grad.cpu(), hi.grad.cpu(),
                     {k: p.grad.detach().cpu() for k, p in layer.named_parameters()}))
    (y0, h0_, gx0, gh0, gp0), (y1, h1_, gx1, gh1, gp1) = outs
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(h1_, h0_, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(gx1, gx0, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gh1, gh0, rtol=1e-3, atol=1e-4)
    for k in gp0:
        scale = float(gp0[k].abs().max())
        assert float((gp1[k] - gp0[k]).abs().max()) <= 2e-4 * scale + 1e-5, k
    # rollout shape: one step, x rows == state rows
    xr, hr, mr = torch.randn(4096, H), torch.randn(4096, R, H), (torch.rand(4096, 1) > 0.2).float()
    with torch.no_grad():
        a = cpu(xr, hr, mr)
        b = gpu(xr.to(_dev()), hr.to(_dev()), mr.to(_dev()))
    torch.testing.assert_close(b[0].cpu(), a[0], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(b[1].cpu(), a[1], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind,mode", [("shared", "insert"), ("shared", "chooseinsert"), ("separated", "insert"),
                                       ("separated", "chooseinsert")])
def test_storage_vs_reference_fixtures(gold, kind, mode):
    """insert / chooseinsert / after_update / chooseafter_update through the slab kernel (K2) against the contents
    the reference's own buffers hold after the same seeded stream (oracle/make_golden_storage.py)."""
    from storage_replay import FIELDS as STORED, replay
    z = gold.npz("storage_cases")
    T, N, A, Do, Ds, na, H = [int(x) for x in z["dims"]]
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, use_recurrent_policy=True)
    if kind == "shared":
        buf, lead = _buffer(args, A, Do=Do, Ds=Ds, na=na), (N, A)
    else:
        from onpolicy.utils.separated_buffer import SeparatedReplayBuffer
        buf, lead = SeparatedReplayBuffer(args, Box((Do,)), Box((Ds,)), Discrete(na), device=_dev()), (N,)
    replay(buf, mode, lead, z["dims"])
    assert buf.step == int(z["%s_%s_step" % (kind, mode)])
    for name in STORED:
        np.testing.assert_array_equal(getattr(buf, name).cpu().numpy(), z["%s_%s_%s" % (kind, mode, name)], err_msg=name)


# ------------------------------------------------------------------ K1: time-parallel scan for narrow buffers
@pytest.mark.parametrize("T,N,A,ptl,norm", [(200, 1024, 5, False, True), (400, 512, 10, True, True),
                                            (400, 512, 8, False, False), (100, 2048, 2, True, False),
                                            (64, 1024, 3, False, True), (397, 700, 4, False, True)])
def test_gae_time_parallel_scan_vs_oracle(T, N, A, ptl, norm):
    """Narrow buffers (2048 <= N * A < 16384 columns) take the time-parallel scan (csrc/mappo_gae.hip gae_scan_kernel):
    16 segments folded independently and stitched through their affine maps.  Tolerance mode -- rtol 1e-5 against the
    oracle (the north star's "fp32 tolerance for returns / advantages"; measured ~1e-6), moments 2e-6; ``gae_exact``
    switches back to the bit-identical kernels for the same shape."""
    C = N * A
    assert 2048 <= C < 16384
    rng = np.random.default_rng(T + C)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, 3, 4, 5, 8), rng, na=5)
    triplet = [0.7e-4, 3.1e-4, 2.5e-5] if norm else None
    kw = dict(episode_length=T, n_rollout_threads=N, use_proper_time_limits=ptl, use_valuenorm=norm, hidden_size=8)
    ob = oracle.OracleBuffer(make_args(**kw), A, Box((3,)), Box((4,)), Discrete(5))
    load_into(ob, arrays)
    ob.compute_returns(arrays["next_value"], _vn(triplet) if norm else None)
    ref = ob.returns
    stats_exact = None
    for exact in (True, False):
        buf = _buffer(make_args(gae_scan=not exact, **kw), A)
        load_into(buf, arrays)
        vn = _vn(triplet) if norm else None
        buf.compute_returns(arrays["next_value"], vn)
        got = buf.returns.cpu().numpy()
        np.testing.assert_array_equal(buf.value_preds[-1].cpu().numpy(), arrays["next_value"])
        stats = buf.normalized_advantages(vn).stats.cpu().numpy()
        if exact:
            np.testing.assert_array_equal(got, ref)
            stats_exact = stats
        else:
            scale = np.abs(ref).max()
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * scale)
            assert not np.array_equal(got, ref)                  # it really is the other kernel
            np.testing.assert_allclose(stats, stats_exact, rtol=5e-5)     # advantage mean / std from the fused moments
