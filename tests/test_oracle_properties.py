"""Size-independent properties of the path, checked on the CPU oracle with hypothesis: the
same properties are asserted of the HIP path at scale in tests/test_gpu_parity.py::test_gae_properties_at_scale."""
import numpy as np
from hypothesis import given, settings, strategies as st

from helpers import Box, Discrete, make_args
from oracle import oracle

f32 = np.float32
SETTINGS = dict(max_examples=25, deadline=None, derandomize=True)


def _rollout(rng, T, N, A, p_mask=0.85):
    r = rng.standard_normal((T, N, A, 1)).astype(f32)
    v = rng.standard_normal((T + 1, N, A, 1)).astype(f32)
    m = (rng.random((T + 1, N, A, 1)) < p_mask).astype(f32)
    bad = (rng.random((T + 1, N, A, 1)) < 0.9).astype(f32)
    return r, v, v[-1].copy(), m, bad


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), T=st.integers(1, 40), N=st.integers(1, 6), A=st.integers(1, 4),
       gamma=st.floats(0.5, 1.0))
def test_lambda_one_without_episode_ends_is_the_discounted_reward_to_go(seed, T, N, A, gamma):
    """With every mask 1 and lambda = 1 the GAE return telescopes: ret_t = sum_k gamma^k r_{t+k} + gamma^(T-t) v_T."""
    rng = np.random.default_rng(seed)
    r, v, nv, m, _ = _rollout(rng, T, N, A, p_mask=2.0)
    ret, _ = oracle.compute_returns(r, v, nv, m, gamma=gamma, gae_lambda=1.0)
    expect = np.zeros((T + 1, N, A, 1))
    expect[T] = nv
    for t in range(T - 1, -1, -1):
        expect[t] = r[t].astype(np.float64) + gamma * expect[t + 1]
    np.testing.assert_allclose(ret[:T], expect[:T], rtol=2e-4, atol=2e-4)
    assert not ret[T].any()          # GAE leaves the last row untouched (shared_buffer.py:236-240)


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), T=st.integers(1, 30), N=st.integers(2, 7), A=st.integers(1, 3),
       ptl=st.booleans(), use_gae=st.booleans(), denorm=st.booleans())
def test_rollout_threads_are_independent_columns(seed, T, N, A, ptl, use_gae, denorm):
    """Permuting the rollout threads permutes the returns bit for bit, and a shard of the threads gives the same
    numbers as the same threads inside the full batch -- what sharding over ranks relies on."""
    rng = np.random.default_rng(seed)
    r, v, nv, m, bad = _rollout(rng, T, N, A)
    kw = dict(sigma=1.7, mu=-0.3, use_gae=use_gae, use_proper_time_limits=ptl, denorm=denorm)
    ret, _ = oracle.compute_returns(r, v, nv, m, bad, **kw)
    perm = rng.permutation(N)
    ret_p, _ = oracle.compute_returns(r[:, perm], v[:, perm], nv[perm], m[:, perm], bad[:, perm], **kw)
    assert np.array_equal(ret_p, ret[:, perm])
    cut = int(rng.integers(1, N))
    lo, _ = oracle.compute_returns(r[:, :cut], v[:, :cut], nv[:cut], m[:, :cut], bad[:, :cut], **kw)
    hi, _ = oracle.compute_returns(r[:, cut:], v[:, cut:], nv[cut:], m[:, cut:], bad[:, cut:], **kw)
    assert np.array_equal(np.concatenate([lo, hi], 1), ret)
    # advantage moments of the shards combine to those of the whole (the three all-reduced sums)
    v_full = v.copy()
    v_full[-1] = nv
    adv = oracle.advantages(ret, v_full, sigma=1.7, mu=-0.3, denorm=denorm)
    am = (rng.random(adv.shape) < 0.8).astype(f32)
    am[0, 0] = 1.0
    mean, std, cnt = oracle.adv_moments(adv, am)
    parts = [oracle.adv_moments(adv[:, s], am[:, s]) for s in (slice(0, cut), slice(cut, N))]
    parts = [p for p in parts if p[2] > 0]
    cnt2 = sum(p[2] for p in parts)
    s1 = sum(p[0] * p[2] for p in parts)
    s2 = sum((p[1] ** 2 + p[0] ** 2) * p[2] for p in parts)
    assert cnt2 == cnt
    np.testing.assert_allclose(s1 / cnt2, mean, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.sqrt(max(s2 / cnt2 - (s1 / cnt2) ** 2, 0.0)), std, rtol=1e-5, atol=1e-6)


@settings(**SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), T=st.integers(2, 12), N=st.integers(1, 4), A=st.integers(1, 3),
       mini=st.integers(1, 4), recurrent=st.booleans())
def test_samplers_visit_every_row_exactly_once(seed, T, N, A, mini, recurrent):
    """One epoch of the feed-forward sampler is a permutation of the T*N*A rows when the batch divides evenly; the
    recurrent sampler is a permutation of the whole chunks (rows tagged through the obs field)."""
    import torch
    L = 2
    rows = T * N * A
    units = rows // L if recurrent else rows
    if units < mini:
        mini = 1
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=4, data_chunk_length=L,
                     use_recurrent_policy=recurrent)
    buf = oracle.OracleBuffer(args, A, Box((1,)), Box((2,)), Discrete(3))
    tags = np.arange((T + 1) * N * A, dtype=f32).reshape(T + 1, N, A, 1)
    buf.obs[:] = tags
    torch.manual_seed(seed)
    adv = np.zeros((T, N, A, 1), dtype=f32)
    gen = buf.recurrent_generator(adv, mini, L) if recurrent else buf.feed_forward_generator(adv, mini)
    seen = np.concatenate([np.asarray(sample[1]).reshape(-1) for sample in gen])
    per = units // mini
    assert len(seen) == per * mini * (L if recurrent else 1)
    assert len(np.unique(seen)) == len(seen) and seen.max() < rows          # no row twice, none from row T
    if units % mini == 0 and (not recurrent or rows % L == 0):
        assert len(seen) == rows
