"""-m gpu: K10, the sort-free device sampler (mappo_minibatch_indices): minibatch index lists from a keyed bijection of
[0, n) instead of torch.randperm + slicing (reference onpolicy/utils/shared_buffer.py:360-361, :511-512).  Integer work:
checked bit for bit against a numpy restatement of the same Feistel network, plus the properties a sampler needs
(disjoint slices of exactly mb samples in ascending order, the right samples dropped, assignment that looks uniform
and changes with the keys)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle.k10_partition import permute as _permute_ref      # the numpy restatement (test infrastructure)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
M32 = 0xFFFFFFFF


def _indices(n, mb, n_mb, keys):
    from onpolicy import _native
    lib = _native.lib()
    idx = torch.full((n_mb * mb,), -1, dtype=torch.int64, device=DEV)
    ws = torch.empty(lib.mappo_minibatch_workspace_ints(n, n_mb), dtype=torch.int32, device=DEV)
    _native.check(lib.mappo_minibatch_indices(n, mb, n_mb, (ctypes.c_uint32 * 6)(*keys), idx.data_ptr(), ws.data_ptr(),
                                              None), "mappo_minibatch_indices")
    torch.cuda.synchronize()
    return idx.cpu().numpy()


@pytest.mark.parametrize("n,mb,n_mb", [(1000, 1000, 1), (1000, 333, 3), (4097, 1024, 4), (75000, 7500, 10),
                                       (300001, 9375, 32), (2048 * 3 + 5, 6149, 1), (17, 4, 4), (5, 1, 5)])
def test_slices_equal_numpy_restatement(n, mb, n_mb):
    keys = [int(k) for k in np.random.default_rng(n).integers(0, 1 << 32, 6)]
    got = _indices(n, mb, n_mb, keys)
    perm = _permute_ref(n, keys)
    assert np.array_equal(np.sort(perm), np.arange(n))                      # the restatement is a bijection
    for m in range(n_mb):
        members = np.flatnonzero((perm >= m * mb) & (perm < (m + 1) * mb))    # ascending by construction
        np.testing.assert_array_equal(got[m * mb:(m + 1) * mb], members)


def test_large_batch_properties_and_key_dependence():
    n, n_mb = 13_107_200, 4            # the north-star batch in four minibatches
    mb = n // n_mb
    a = _indices(n, mb, n_mb, [1, 2, 3, 4, 5, 6])
    b = _indices(n, mb, n_mb, [7, 2, 3, 4, 5, 6])
    for idx in (a, b):
        assert np.array_equal(np.sort(idx), np.arange(n))                    # a partition of all samples
        for m in range(n_mb):
            s = idx[m * mb:(m + 1) * mb]
            assert np.all(np.diff(s) > 0)                                     # ascending memory order
    # assignment looks uniform: every minibatch takes about a quarter of any contiguous region of the buffer
    region = a[:mb] < n // 8
    assert abs(region.mean() - 1 / 8) < 2e-3
    # and it changes with the keys: two draws share about 1/4 of a slice, like independent random partitions
    overlap = np.intersect1d(a[:mb], b[:mb]).size / mb
    assert 0.24 < overlap < 0.26


def test_buffer_generators_use_the_device_sampler():
    """feed_forward_generator / recurrent_generator in device mode: every minibatch ascending, the epoch a partition, a
    different partition every epoch, reproducible under torch.manual_seed."""
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A = 12, 10, 3
    args = make_args(episode_length=T, n_rollout_threads=N, data_chunk_length=4)
    buf = SharedReplayBuffer(args, A, Box((6,)), Box((18,)), Discrete(5), device=DEV)
    buf.obs.copy_(torch.arange((T + 1) * N * A * 6, device=DEV).float().reshape(buf.obs.shape))
    adv = torch.zeros(T, N, A, 1, device=DEV)

    def epoch():
        rows = []
        for sample in buf.feed_forward_generator(adv, num_mini_batch=3):
            r = (sample[1][:, 0] / 6).long().cpu().numpy()      # obs row -> flat sample index
            assert np.all(np.diff(r) > 0)
            rows.append(r)
        return rows
    torch.manual_seed(3)
    e1 = epoch()
    e2 = epoch()
    assert np.array_equal(np.sort(np.concatenate(e1)), np.arange(T * N * A))
    assert not np.array_equal(e1[0], e2[0])
    torch.manual_seed(3)
    e3 = epoch()
    assert all(np.array_equal(x, y) for x, y in zip(e1, e3))
    chunks = [s[1].shape[0] for s in buf.recurrent_generator(adv, 2, 4)]
    assert chunks == [T * N * A // 4 // 2 * 4] * 2


def test_single_minibatch_shortcut_equals_the_kernel():
    """One minibatch that takes every sample: the buffer hands out a cached 0 .. n - 1 list instead of running K10 -- which
    is exactly what K10 emits for one slice, whatever the keys (and the CPU generator still advances by the same draw)."""
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    n = 6 * 50 * 3
    for keys in ([1, 2, 3, 4, 5, 6], [0xDEADBEEF, 7, 0, M32, 99, 12345]):
        np.testing.assert_array_equal(_indices(n, n, 1, keys), np.arange(n))
    buf = SharedReplayBuffer(make_args(episode_length=6, n_rollout_threads=50), 3, Box((4,)), Box((5,)), Discrete(3), device=DEV)
    torch.manual_seed(5)
    a = buf._sampler_indices(n, n, 1)
    after_shortcut = torch.randint(0, 1 << 30, (1,)).item()
    torch.manual_seed(5)
    buf._sampler_indices(n, n // 2, 2)           # the kernel path draws the same six keys
    assert torch.randint(0, 1 << 30, (1,)).item() == after_shortcut
    np.testing.assert_array_equal(a.cpu().numpy(), np.arange(n))


def test_whole_batch_minibatch_is_gathered_once_per_buffer_content():
    """num_mini_batch = 1 under the device sampler: the epochs of one train() get the SAME 12-tuple (one gather, one row
    table) -- until anything writes the buffer: a torch in-place edit of a field, an insert, compute_returns,
    after_update (which also drops the cached tuple)."""
    from helpers import Box, Discrete, make_args
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A = 6, 20, 3
    buf = SharedReplayBuffer(make_args(episode_length=T, n_rollout_threads=N), A, Box((4,)), Box((8,)), Discrete(3), device=DEV)
    g = torch.Generator().manual_seed(0)
    for name in ("obs", "share_obs", "rewards", "value_preds", "returns", "action_log_probs"):
        getattr(buf, name).copy_(torch.randn(getattr(buf, name).shape, generator=g).to(DEV))
    adv = torch.randn(T, N, A, 1, generator=g).to(DEV)
    first = next(iter(buf.feed_forward_generator(adv, num_mini_batch=1)))
    # an external advantages array is the caller's: never cached
    assert next(iter(buf.feed_forward_generator(adv, num_mini_batch=1)))[6] is not first[6]
    from onpolicy.utils.shared_buffer import AdvantageHandle
    handle = AdvantageHandle(adv, None, buf)
    one = next(iter(buf.feed_forward_generator(handle, num_mini_batch=1)))
    two = next(iter(buf.feed_forward_generator(handle, num_mini_batch=1)))
    assert all(a is b for a, b in zip(one, two))
    torch.testing.assert_close(one[6], buf.returns[:-1].reshape(-1, 1), rtol=0, atol=0)
    buf.returns[0].add_(1.0)                                      # in-place torch edit of a field
    three = next(iter(buf.feed_forward_generator(handle, num_mini_batch=1)))
    assert three[6] is not one[6]
    torch.testing.assert_close(three[6], buf.returns[:-1].reshape(-1, 1), rtol=0, atol=0)
    assert all(a is b for a, b in zip(three, next(iter(buf.feed_forward_generator(handle, num_mini_batch=1)))))
    # several minibatches: fresh tensors every time
    parts = list(buf.feed_forward_generator(handle, num_mini_batch=2))
    assert len(parts) == 2 and parts[0][6].shape[0] == T * N * A // 2
    buf.after_update()
    assert buf._whole_batch is None
