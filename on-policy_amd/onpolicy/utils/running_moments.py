"""Debiased exponential moving moments -- the statistics behind both value normalisers of the reference
(``ValueNorm``, onpolicy/utils/valuenorm.py:32-79, and the ``PopArt`` head, onpolicy/algorithms/utils/popart.py:49-98
use the same arithmetic under different attribute names):

    m1 <- w m1 + (1 - w) E[x],   m2 <- w m2 + (1 - w) E[x^2],   d <- w d + (1 - w)
    mean = m1 / max(d, eps),     var = max(m2 / max(d, eps) - mean^2, 1e-2)

``DebiasedMoments`` is a mix-in for an ``nn.Module`` that owns three buffers (their names are class attributes, so the
state dicts keep the reference's keys); it provides the update, (de)normalisation with numpy-in / numpy-out and
tensor-in / tensor-out, and ``denorm_scalars()`` -- the [sigma, mu] device tensor the GAE kernel reads without a
host sync.  ``update`` accepts batch moments that were all-reduced across GPUs in place of a local batch.
"""
import numpy as np
import torch

_VAR_FLOOR = 1e-2


class DebiasedMoments(object):
    _first_moment = "running_mean"
    _second_moment = "running_mean_sq"
    _debias = "debiasing_term"

    # -- storage
    def _moments(self):
        return getattr(self, self._first_moment), getattr(self, self._second_moment), getattr(self, self._debias)

    def _stats_device(self):
        return getattr(self, self._first_moment).device

    def _as_tensor(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(dtype=torch.float32, device=self._stats_device())

    def zero_moments(self):
        for buf in self._moments():
            buf.zero_()

    # -- statistics
    def _mean_var(self):
        m1, m2, d = self._moments()
        d = d.clamp(min=self.epsilon)
        mean = m1 / d
        return mean, (m2 / d - mean ** 2).clamp(min=_VAR_FLOOR)

    def _batch_moments(self, x):
        axes = tuple(range(self.norm_axes))
        return x.mean(dim=axes), (x ** 2).mean(dim=axes)

    # -- the HIP route for scalar statistics (ValueNorm(1) on the device): update + [sigma, mu] as two launches
    def _fused_ok(self, input_vector, batch_moments):
        import os
        m1 = getattr(self, self._first_moment)
        if not (m1.is_cuda and m1.numel() == 1 and self.norm_axes == 1 and os.environ.get("MAPPO_FUSED_VALUENORM", "1") != "0"
                and type(self).__name__ == "ValueNorm"):
            return False
        if batch_moments is not None:
            return all(torch.is_tensor(t) and t.is_cuda and t.numel() == 1 for t in batch_moments)
        return torch.is_tensor(input_vector) and input_vector.is_cuda and input_vector.dtype == torch.float32 \
            and input_vector.dim() == 2 and input_vector.shape[1] == 1 and input_vector.is_contiguous()

    def _stats_versions(self):
        return tuple((b.data_ptr(), b._version) for b in self._moments())

    @torch.no_grad()
    def _fold_in_fused(self, input_vector, batch_moments, weight):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        m1, m2, d = self._moments()
        dev = m1.device
        if getattr(self, "_denorm_cache", None) is None or self._denorm_cache.device != dev:
            self._denorm_cache = torch.empty(2, dtype=torch.float32, device=dev)
            self._vn_ws = torch.empty(lib.mappo_valuenorm_workspace_doubles(), dtype=torch.float64, device=dev)
        bm = None
        if batch_moments is not None:
            a, b = batch_moments
            # (mean, mean_sq) that already sit side by side in float32 -- DataParallel.minibatch_scales -- are read in place
            bm = a if (a.dtype == torch.float32 and b.dtype == torch.float32 and a.data_ptr() + 4 == b.data_ptr()) \
                else torch.cat([t.reshape(1).float() for t in batch_moments])
        x = None if batch_moments is not None else input_vector
        _native.check(lib.mappo_valuenorm_update(p(x), 0 if x is None else x.numel(), p(bm), float(weight), float(self.epsilon),
                                                 p(m1), p(m2), p(d), p(self._denorm_cache), p(self._vn_ws),
                                                 _native.stream_of(dev)), "mappo_valuenorm_update")
        # (the kernel writes the statistics through raw pointers: the tensors' version counters stay put, so any later
        # in-place torch edit of them -- load_state_dict, zero_moments, a test poking values in -- invalidates the cache)
        self._denorm_key = self._stats_versions()

    @torch.no_grad()
    def _fold_in(self, input_vector, batch_moments, weight):
        if self._fused_ok(input_vector, batch_moments):
            return self._fold_in_fused(input_vector, batch_moments, weight)
        mean, mean_sq = batch_moments if batch_moments is not None \
            else self._batch_moments(self._as_tensor(input_vector))
        m1, m2, d = self._moments()
        m1.mul_(weight).add_(mean * (1.0 - weight))
        m2.mul_(weight).add_(mean_sq * (1.0 - weight))
        d.mul_(weight).add_(1.0 * (1.0 - weight))

    def denorm_scalars(self):
        """float32 device tensor [sigma, mu] (scalar statistics, input_shape == 1).

        ALIASING CONTRACT: after a fused update this is the persistent two-float buffer the update kernel writes
        (no copy, no launch) -- valid until the next ``update()``, which overwrites it in place on the same stream.
        Consume it (hand it to a kernel / an op) before updating again; ``.clone()`` it to keep a snapshot.  Every
        caller in this package does the former (compute_returns, the fused loss).  Writers of the statistics that go
        through raw pointers must reset ``_denorm_key`` (the update kernel does); in-place torch edits are caught by
        the tensors' version counters."""
        if getattr(self, "_denorm_key", None) is not None and self._denorm_key == self._stats_versions():
            return self._denorm_cache        # written by the update kernel, statistics untouched since
        mean, var = self._mean_var()
        return torch.stack([torch.sqrt(var).reshape(()), mean.reshape(())])

    # -- maps
    def _broadcast(self):
        mean, var = self._mean_var()
        lead = (None,) * self.norm_axes
        return mean[lead], torch.sqrt(var)[lead]

    def normalize(self, input_vector):
        mean, sigma = self._broadcast()
        return (self._as_tensor(input_vector) - mean) / sigma

    def denormalize(self, input_vector):
        """x * sigma + mean.  ndarray -> ndarray (what the reference returns); tensor -> tensor on the device."""
        mean, sigma = self._broadcast()
        out = self._as_tensor(input_vector) * sigma + mean
        return out.cpu().numpy() if isinstance(input_vector, np.ndarray) else out
