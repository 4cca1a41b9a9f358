"""TEST INFRASTRUCTURE.  numpy restatement of K10, the product's sort-free device sampler
(on-policy_amd/csrc/mappo_perm.hip: mappo_minibatch_indices).

The reference draws ``rand = torch.randperm(n)`` and cuts it into slices (onpolicy/utils/shared_buffer.py:360-361,
:511-512).  A minibatch is a *set* -- its loss is a mean over its rows -- so the device sampler assigns sample ``r`` to
slice ``perm(r) // mb`` (``perm`` = a keyed bijection of [0, n): 6-round balanced Feistel network on the next even
power-of-two domain with cycle walking) and emits every slice in ascending memory order.  The six 32-bit round keys come
from the CPU generator (``torch.randint(0, 1 << 32, (6,), dtype=torch.int64)``), so ``torch.manual_seed`` fixes them.

Used by (a) tests/test_gpu_sampler_indices.py to check the kernel bit for bit, (b) oracle/make_golden_trainer.py to make
the REFERENCE's samplers draw exactly these slices (``RandpermAsK10``), which pins the device-sampler route of
``R_MAPPO.train`` to reference-generated numbers for several minibatches per epoch.
"""
import numpy as np

M32 = 0xFFFFFFFF


def _mix(r, k):
    h = (r * 0x9E3779B1 + k) & M32
    h ^= h >> 15
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    return h


def permute(n, keys):
    """perm(r) for r in [0, n)."""
    bits = 2
    while (1 << bits) < n:
        bits += 2
    half = bits // 2
    mask = (1 << half) - 1
    x = np.arange(n, dtype=np.uint64)
    out = np.empty(n, dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        l = (x[todo] >> np.uint64(half)).astype(np.uint64)
        r = (x[todo] & np.uint64(mask)).astype(np.uint64)
        for k in keys:
            t = l ^ (_mix(r, k) & np.uint64(mask))
            l, r = r, t
        y = (l << np.uint64(half)) | r
        x[todo] = y
        done = y < n
        out[todo[done]] = y[done]
        todo = todo[~done]
    return out.astype(np.int64)


def slices(n, mb, n_mb, keys):
    """The ``n_mb`` index lists of ``mb`` samples each, ascending, as K10 writes them back to back."""
    perm = permute(n, keys)
    return [np.flatnonzero((perm >= m * mb) & (perm < (m + 1) * mb)) for m in range(n_mb)]


def as_permutation(n, mb, n_mb, keys):
    """A full permutation of [0, n) whose first ``n_mb`` slices of ``mb`` entries are K10's minibatches (what the
    reference's ``rand[i * mb:(i + 1) * mb]`` then picks up); the samples K10 drops follow in ascending order."""
    perm = permute(n, keys)
    parts = [np.flatnonzero((perm >= m * mb) & (perm < (m + 1) * mb)) for m in range(n_mb)]
    parts.append(np.flatnonzero(perm >= n_mb * mb))
    out = np.concatenate(parts).astype(np.int64)
    assert out.size == n
    return out


class RandpermAsK10(object):
    """Context manager: ``torch.randperm(n)`` -> the permutation above, the keys drawn from the CPU generator the way
    the product's ``SharedReplayBuffer._sampler_indices`` draws them (one ``torch.randint`` of six int64 per call), so
    a reference ``train()`` under the same ``torch.manual_seed`` sees the minibatches the device sampler makes.
    ``n_mb``: the generator's ``num_mini_batch`` (``mb = n // n_mb`` is what both reference samplers compute)."""

    def __init__(self, n_mb):
        import torch
        self.torch = torch
        self.n_mb = int(n_mb)
        self.calls = []

    def __enter__(self):
        torch = self.torch
        self.orig = torch.randperm

        def k10(n, *a, **k):
            keys = torch.randint(0, 1 << 32, (6,), dtype=torch.int64).tolist()
            p = as_permutation(int(n), int(n) // self.n_mb, self.n_mb, keys)
            self.calls.append(p.copy())
            return torch.from_numpy(p)
        torch.randperm = k10
        return self

    def __exit__(self, *exc):
        self.torch.randperm = self.orig
