"""HBM-resident rollout buffer for shared-policy MAPPO.

Drop-in for the reference's ``SharedReplayBuffer`` (onpolicy/utils/shared_buffer.py:21-608): same
constructor, attributes, methods and 12-tuple minibatch protocol, but

  * every field is a float32 torch tensor in device memory, in the reference's time-major layout
    ``[episode_length(+1), n_rollout_threads, num_agents, dim]`` (element (t, n, a) is row
    ``(t*N + n)*A + a`` of a [rows, dim] matrix), so one rollout step is one contiguous slab per
    field and the GAE scan reads coalesced rows;
  * ``compute_returns`` is one launch of the ``mappo_gae_f32`` HIP kernel (GAE / discounted returns
    with ValueNorm / PopArt de-normalisation, all seven non-MAT reference branches, bit-identical
    float32), which also emits the un-normalised advantages and their masked moments so that
    ``R_MAPPO.train`` needs no extra passes over the buffer (reference r_mappo.py:179-187);
  * the three samplers are one fused multi-field gather launch per minibatch
    (``mappo_gather_rows`` / ``mappo_gather_chunks``) reading straight from the time-major
    buffer -- no transposed copies, no Python loop over chunks, no host<->device traffic;
  * ``insert`` / ``after_update`` are one fused slab-copy launch (``mappo_slab_copy``).

There is no CPU implementation here: the constructor raises unless it gets a HIP device and
``libmappo_hip.so`` (see onpolicy/_native.py).
"""
import ctypes
import os

import numpy as np
import torch

from onpolicy import _native
from onpolicy.utils.util import get_shape_from_obs_space, get_shape_from_act_space


class AdvantageHandle(object):
    """Normalised advantages without materialising them: the raw ``returns - D(value_preds)`` tensor
    plus the device scalars (mean, std).  The samplers apply ``(adv - mean) / (std + 1e-5)``
    (reference r_mappo.py:187) while gathering.  ``materialize()`` gives the [T, N, A, 1] tensor the
    reference's ``train`` would have built."""

    def __init__(self, raw, stats, buffer):
        self.raw = raw
        self.stats = stats
        self._buffer = buffer

    def materialize(self):
        out = torch.empty_like(self.raw)
        L = _native.lib()
        _native.check(L.mappo_adv_normalize(self.raw.data_ptr(), self.stats.data_ptr(), out.data_ptr(),
                                            self.raw.numel(), _native.stream_of(self.raw.device)),
                      "mappo_adv_normalize")
        return out


class SharedReplayBuffer(object):
    """
    :param args: (argparse.Namespace) the reference's flag namespace (onpolicy/config.py).
    :param num_agents: (int) agents per environment.
    :param obs_space / cent_obs_space / act_space: gym-style spaces, recognised by class name.
    :param device: optional torch device; default ``args.buffer_device`` or the current HIP device.
    """

    def __init__(self, args, num_agents, obs_space, cent_obs_space, act_space, device=None):
        self.episode_length = args.episode_length
        self.n_rollout_threads = args.n_rollout_threads
        self.hidden_size = args.hidden_size
        self.recurrent_N = args.recurrent_N
        self.gamma = args.gamma
        self.gae_lambda = args.gae_lambda
        self._use_gae = args.use_gae
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_proper_time_limits = args.use_proper_time_limits
        self.algo = args.algorithm_name
        self.num_agents = num_agents
        self._recurrent = bool(getattr(args, "use_recurrent_policy", False) or
                               getattr(args, "use_naive_recurrent_policy", False))
        self._sampler_rng = getattr(args, "sampler_rng", "device")
        # Returns are bit-identical to the reference's loop for every buffer shape unless the caller opts into the
        # time-parallel scan for narrow buffers (--gae_scan / MAPPO_GAE_SCAN=1: 2048 <= N * A < 16384 columns, ~1e-6 relative;
        # VERDICT r5 "next" #9: the launch is < 0.2 % of a step either way, so bit-exactness is the default).  --gae_exact /
        # MAPPO_GAE_EXACT=1 and --sampler_rng host (the integer-parity mode of the whole path) always mean the exact kernels.
        scan = bool(getattr(args, "gae_scan", False)) or os.environ.get("MAPPO_GAE_SCAN", "0") == "1"
        self._gae_exact = (not scan) or bool(getattr(args, "gae_exact", False)) \
            or os.environ.get("MAPPO_GAE_EXACT", "0") == "1" or self._sampler_rng == "host"

        self.device = dev = self._resolve_device(args, device)
        self._lib = _native.lib()  # raises if the HIP library is not built

        obs_shape = get_shape_from_obs_space(obs_space)
        share_obs_shape = get_shape_from_obs_space(cent_obs_space)
        if type(obs_shape[-1]) == list:
            obs_shape = obs_shape[:1]
        if type(share_obs_shape[-1]) == list:
            share_obs_shape = share_obs_shape[:1]

        T, N, A = self.episode_length, self.n_rollout_threads, num_agents
        f32 = dict(dtype=torch.float32, device=dev)
        self.share_obs = torch.zeros((T + 1, N, A, *share_obs_shape), **f32)
        self.obs = torch.zeros((T + 1, N, A, *obs_shape), **f32)

        rnn_shape = (T + 1, N, A, self.recurrent_N, self.hidden_size)
        if self._recurrent:
            self.rnn_states = torch.zeros(rnn_shape, **f32)
            self.rnn_states_critic = torch.zeros(rnn_shape, **f32)
        else:
            # Feed-forward policies never read or change the RNN state (it is all zeros for the
            # whole run), so it gets no storage: a stride-0 view of one zero serves every reader.
            # (The reference allocates 2 x [T+1, N, A, R, H] regardless: 16.8 GB at Hanabi scale.)
            zero = torch.zeros(1, **f32)
            self.rnn_states = zero.expand(rnn_shape)
            self.rnn_states_critic = zero.expand(rnn_shape)

        self.value_preds = torch.zeros((T + 1, N, A, 1), **f32)
        self.returns = torch.zeros_like(self.value_preds)
        self.advantages = torch.zeros((T, N, A, 1), **f32)

        if act_space.__class__.__name__ == 'Discrete':
            self.available_actions = torch.ones((T + 1, N, A, act_space.n), **f32)
        else:
            self.available_actions = None

        act_shape = get_shape_from_act_space(act_space)
        self.actions = torch.zeros((T, N, A, act_shape), **f32)
        self.action_log_probs = torch.zeros((T, N, A, act_shape), **f32)
        self.rewards = torch.zeros((T, N, A, 1), **f32)

        self.masks = torch.ones((T + 1, N, A, 1), **f32)
        self.bad_masks = torch.ones_like(self.masks)
        self.active_masks = torch.ones_like(self.masks)

        self.step = 0
        # optional extra per-sample fields ([T, N, A, k] tensors) that the samplers gather after the
        # 12 standard ones (the separated buffer's HAPPO ``factor`` is one)
        self.extra_fields = {}

        # device workspaces of the GAE epilogue (tiny)
        rows = int(self._lib.mappo_gae_partial_rows(N * A))
        self._partial_rows = rows
        self._adv_partials = torch.zeros((rows, 3), dtype=torch.float64, device=dev)
        self._adv_sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self._adv_stats = torch.zeros(2, **f32)
        self._content_version = 0  # bumped by every method that writes buffer fields
        self._whole_batch = self._whole_batch_key = None    # the one-minibatch tuple of feed_forward_generator, see there
        self._whole_batch_versions = ()
        self.whole_batch_reuses = 0       # epochs that were handed the cached tuple (tests assert the route)
        self._std_rows = {}        # field name -> (key, row-standardised copy) for the fused trunk kernels
        self._std_keep = {}        # field name -> storage of the last copy, kept across train() calls when small (below)
        self._obs_writes = {"share_obs": 0, "obs": 0}   # slab writes of this class's kernels into the observation fields
        # MAPPO_STANDARDIZE_AT_INSERT (default on): once a train() has asked for the row-standardised copy of an
        # observation field, the copy stays resident and insert / chooseinsert / after_update keep it current slab by slab
        # (_write_slabs), so a train() finds it ready instead of re-standardising the whole field (VERDICT r5 "next" #4a)
        self._std_at_insert = os.environ.get("MAPPO_STANDARDIZE_AT_INSERT", "1") != "0"
        self.std_slab_launches = self.std_full_passes = 0      # (tests / bench.py: which of the two ran)
        # MAPPO_PINNED_INSERT=1: host inputs of insert() go through one pinned staging buffer + one async H2D copy.
        # Off by default: measured at the north star (tools/pcie_insert_bench.py, 57 MB per step) the single-threaded
        # memcpy into the staging buffer makes it slower (1.88 ms per step, 30 GB/s) than one pageable .to(device) per
        # field (1.11 ms, 51 GB/s), whose staging the runtime pipelines internally
        self._pinned_insert = os.environ.get("MAPPO_PINNED_INSERT", "0") == "1"
        self._host_stage = None
        self._adv_fresh = False   # advantages/moments match the current returns & value_preds
        self._adv_denormalized = False   # ... and were formed as returns - D(value_preds)
        self._adv_is_gae = False         # ... or are the GAE accumulator of the MAT branches
        self._stats_fresh = False
        self._events = None       # kernel name -> [(start, end, algorithmic bytes)], see profile_kernels

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _resolve_device(args, device):
        cand = device if device is not None else getattr(args, "buffer_device", None)
        if cand is None:
            if not torch.cuda.is_available():
                raise RuntimeError(
                    "SharedReplayBuffer is HBM-resident and needs a HIP device (torch.cuda.is_available() "
                    "is False). There is no CPU implementation of this path.")
            cand = torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(cand)
        if dev.type != "cuda":
            raise RuntimeError("SharedReplayBuffer needs a HIP ('cuda') device, got %r: the rollout "
                               "buffer, GAE and samplers are HIP kernels with no CPU fallback." % (dev,))
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def _stream(self):
        return _native.stream_of(self.device)

    # -- optional per-launch timing of the HIP kernels with events on the launch stream
    def profile_kernels(self, enabled=True):
        self._events = {} if enabled else None

    def _timed(self, name, nbytes, settle=False):
        if self._events is None or torch.cuda.is_current_stream_capturing():
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), nbytes)
        self._events.setdefault(name, []).append(ev)
        if settle:
            # (the GAE launch, first kernel of a step:) keep the stream busy for ~0.1 ms first: on an idle stream the
            # start event would fire at once and the interval would include the host's launch latency (tens of
            # microseconds of Python / ctypes), not just the kernel.  Behind the spin, start event, kernel and end event
            # are all queued before the GPU reaches them, so the interval is the kernel's duration.  Once per step:
            # the sampler's gathers run from a filled queue and are timed without it (ten spins per step were 2.5 % of
            # an 8-GPU shard's step).
            torch.cuda._sleep(250000)
        ev[0].record(torch.cuda.current_stream(self.device))
        return ev

    def _timed_end(self, ev):
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))

    def kernel_times(self, reset=True):
        """{kernel: (launches, mean ms, mean algorithmic bytes)} since profiling was enabled."""
        torch.cuda.synchronize(self.device)
        out = {}
        for name, evs in (self._events or {}).items():
            if evs:
                ms = [a.elapsed_time(b) for a, b, _ in evs]
                out[name] = (len(evs), sum(ms) / len(ms), sum(n for _, _, n in evs) / len(evs))
        import ctypes
        ms, nb = [], []
        for slot, nbytes in getattr(self, "_gae_dispatch", []):
            v = ctypes.c_float(0.0)
            if self._lib.mappo_gae_timed_launch_ms(slot, ctypes.byref(v)) == 0:     # (a launch without the hook: skipped)
                ms.append(float(v.value))
                nb.append(nbytes)
        if ms:      # the GAE launches once more, by the kernel's own begin / end timestamps
            out["mappo_gae_f32/dispatch"] = (len(ms), sum(ms) / len(ms), sum(nb) / len(nb))
        if reset and self._events is not None:
            self._events = {}
            self._gae_dispatch = []
        return out

    def _dev(self, x):
        """Anything array-like -> contiguous float32 tensor on the buffer's device."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        elif not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.detach()
        if x.device != self.device or x.dtype != torch.float32:
            x = x.to(device=self.device, dtype=torch.float32, non_blocking=True)
        return x.contiguous()

    # ---- host -> HBM staging of a rollout step (the caller side of the boundary: envs that live on the host)
    def _stage_host_values(self, pairs):
        """Values that arrive as host arrays (numpy / CPU tensors / lists) are packed into ONE page-locked staging
        buffer and cross PCIe as ONE asynchronous copy on the buffer's stream, instead of one synchronous pageable
        ``.to(device)`` per field; the fused slab write (K2) that follows reads them from a device staging area.
        ``insert()`` then returns as soon as the host memcpy into the staging buffer is done: the DMA (57 MB per step at
        the north star) overlaps the next env step.  Two staging buffers alternate; an event per buffer keeps a step
        from overwriting data whose copy is still in flight.  -> pairs with host values replaced by device views."""
        # (host values only: a tensor that lives on another GPU takes the ordinary device-to-device path of _dev())
        host = [(i, v) for i, (_, v) in enumerate(pairs) if not (torch.is_tensor(v) and v.device.type != "cpu")]
        if not host or not self._pinned_insert:
            return pairs
        arrays = []
        for i, v in host:
            a = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
            arrays.append((i, a))
        total = sum(a.size for _, a in arrays)
        st = self._host_stage
        if st is None or st["pin"][0].numel() < total:
            st = {"pin": [torch.empty(total, dtype=torch.float32, pin_memory=True) for _ in range(2)],
                  "dev": [torch.empty(total, dtype=torch.float32, device=self.device) for _ in range(2)],
                  "done": [None, None], "turn": 0}
            self._host_stage = st
        k = st["turn"]
        st["turn"] = 1 - k
        if st["done"][k] is not None:
            st["done"][k].synchronize()                 # the copy issued two inserts ago (long finished in practice)
        pin_np = st["pin"][k].numpy()
        out, off = list(pairs), 0
        for i, a in arrays:
            np.copyto(pin_np[off:off + a.size].reshape(a.shape), a, casting="unsafe")      # cast to float32 on the way
            out[i] = (pairs[i][0], st["dev"][k][off:off + a.size].view(a.shape))
            off += a.size
        st["dev"][k][:off].copy_(st["pin"][k][:off], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        st["done"][k] = ev
        return out

    def _write_slabs(self, pairs, obs_slabs=()):
        """One mappo_slab_copy launch for [(dst_view, value), ...] (K2).  ``obs_slabs``: (field name, time index) of the
        pairs that write a slab of share_obs / obs -- the resident row-standardised copy of that field (if a train() made
        one) is brought up to date for just that slab."""
        if self.device.type == "cuda":
            pairs = self._stage_host_values(pairs)
        keep, slabs, srcs = [], [], {}
        for dst, value in pairs:
            src = self._dev(value)
            if src.numel() != dst.numel() and src.numel() * dst.shape[-1] == dst.numel():
                # numpy assignment semantics of the reference (shared_buffer.py:107-121): a [.., 1] value is
                # broadcast over the last axis, e.g. the summed log-prob of a continuous action over act_dim
                src = src.reshape(*dst.shape[:-1], 1).expand(dst.shape).contiguous()
            if src.numel() != dst.numel():
                raise ValueError("cannot write %d elements into a buffer slab of %d (shape %s)"
                                 % (src.numel(), dst.numel(), tuple(dst.shape)))
            assert dst.is_contiguous()
            keep.append(src)
            srcs[dst.data_ptr()] = src
            slabs.append((src.data_ptr(), dst.data_ptr(), dst.numel()))
        arr = (_native.Slab * len(slabs))(*[_native.Slab(s, d, n) for s, d, n in slabs])
        # the slabs of the resident standardised copies that follow (same launch: K2's extra workgroups, csrc/mappo_copy.hip)
        std = []
        for name, t in obs_slabs:
            hit = self._obs_slab_written(name, t)
            if hit is not None:
                na = self.n_rollout_threads * self.num_agents
                field = getattr(self, name)
                src = srcs[field[t].data_ptr()]             # the value being written, not the buffer slab
                D = int(field.shape[-1])
                out = hit[t * na:(t + 1) * na]
                std.append(_native.StdSlab(src.data_ptr(), out.data_ptr(), na, D, int(out.shape[1]), 1e-5))
        if std:
            sarr = (_native.StdSlab * len(std))(*std)
            _native.check(self._lib.mappo_slab_copy_std(arr, len(slabs), sarr, len(std), self._stream()),
                          "mappo_slab_copy_std")
            self.std_slab_launches += len(std)      # (slabs standardised; they ride in K2's launch)
        else:
            _native.check(self._lib.mappo_slab_copy(arr, len(slabs), self._stream()), "mappo_slab_copy")
        self._content_version += 1

    def _obs_key(self, name):
        """What a standardised copy of field ``name`` is valid for: this class's slab writes + in-place torch writes."""
        return (name, self._obs_writes[name], getattr(self, name)._version)

    def _obs_slab_written(self, name, t):
        """Slab ``t`` of observation field ``name`` is being rewritten by K2.  A resident standardised copy that was current
        until now stays current: -> the copy, whose slab t the same launch recomputes from the value being written (row T is
        not part of the copy: -> None, nothing to do).  A copy that was already stale stays stale (-> None) and is rebuilt by
        the next train()."""
        hit = self._std_rows.get(name)
        current = self._std_at_insert and hit is not None and hit[0] == self._obs_key(name)
        self._obs_writes[name] += 1
        if not current:
            return None
        self._std_rows[name] = (self._obs_key(name), hit[1])
        return hit[1] if t < self.episode_length else None

    # ------------------------------------------------------------------ storage
    def insert(self, share_obs, obs, rnn_states_actor, rnn_states_critic, actions, action_log_probs,
               value_preds, rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        """Write one rollout step (reference shared_buffer.py:90-123): observation-like fields land in
        row ``step + 1``, action-like fields in row ``step``.  Values may be numpy arrays or tensors
        on any device, shaped [N, A, ...]."""
        s = self.step
        pairs = [(self.share_obs[s + 1], share_obs), (self.obs[s + 1], obs),
                 (self.actions[s], actions), (self.action_log_probs[s], action_log_probs),
                 (self.value_preds[s], value_preds), (self.rewards[s], rewards),
                 (self.masks[s + 1], masks)]
        if self._recurrent:
            pairs += [(self.rnn_states[s + 1], rnn_states_actor),
                      (self.rnn_states_critic[s + 1], rnn_states_critic)]
        if bad_masks is not None:
            pairs.append((self.bad_masks[s + 1], bad_masks))
        if active_masks is not None:
            pairs.append((self.active_masks[s + 1], active_masks))
        if available_actions is not None:
            pairs.append((self.available_actions[s + 1], available_actions))
        self._write_slabs(pairs, obs_slabs=(("share_obs", s + 1), ("obs", s + 1)))
        self._adv_fresh = False
        self.step = (s + 1) % self.episode_length

    def chooseinsert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs,
                     value_preds, rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        """Turn-based (Hanabi) insert (reference shared_buffer.py:125-158): observations, active masks
        and available actions go to row ``step`` instead of ``step + 1``."""
        s = self.step
        pairs = [(self.share_obs[s], share_obs), (self.obs[s], obs),
                 (self.actions[s], actions), (self.action_log_probs[s], action_log_probs),
                 (self.value_preds[s], value_preds), (self.rewards[s], rewards),
                 (self.masks[s + 1], masks)]
        if self._recurrent:
            pairs += [(self.rnn_states[s + 1], rnn_states), (self.rnn_states_critic[s + 1], rnn_states_critic)]
        if bad_masks is not None:
            pairs.append((self.bad_masks[s + 1], bad_masks))
        if active_masks is not None:
            pairs.append((self.active_masks[s], active_masks))
        if available_actions is not None:
            pairs.append((self.available_actions[s], available_actions))
        self._write_slabs(pairs, obs_slabs=(("share_obs", s), ("obs", s)))
        self._adv_fresh = False
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        """Row T becomes row 0 of the next rollout (reference shared_buffer.py:160-170)."""
        fields = [self.share_obs, self.obs, self.masks, self.bad_masks, self.active_masks]
        if self._recurrent:
            fields += [self.rnn_states, self.rnn_states_critic]
        if self.available_actions is not None:
            fields.append(self.available_actions)
        self._write_slabs([(f[0], f[-1]) for f in fields], obs_slabs=(("share_obs", 0), ("obs", 0)))
        self._release_update_scratch()

    def chooseafter_update(self):
        """Hanabi variant (reference shared_buffer.py:172-177)."""
        fields = [self.masks, self.bad_masks]
        if self._recurrent:
            fields += [self.rnn_states, self.rnn_states_critic]
        self._write_slabs([(f[0], f[-1]) for f in fields])
        self._release_update_scratch()

    def _release_update_scratch(self):
        """What a train() leaves behind.  The row-standardised observation copies the fused trunk kernels read (as large
        as the observation fields themselves: 23.5 GB at the north star) STAY resident and current (MAPPO_STANDARDIZE_AT_INSERT,
        the default: insert / after_update re-standardise the one slab they write, so the next train() starts without the
        full pass -- 8 ms and 45 GB of traffic per north-star step -- and the matrices a captured update graph reads keep
        their addresses) as long as a field's copy is at most MAPPO_KEEP_STANDARDIZED_BYTES (default 48 GiB; larger copies,
        or all of them with MAPPO_STANDARDIZE_AT_INSERT=0, go back to the allocator for the rollout; with the switch off,
        copies of up to 8 GiB keep their storage, not their content, for the update graph's sake)."""
        limit = int(os.environ.get("MAPPO_KEEP_STANDARDIZED_BYTES", str((48 if self._std_at_insert else 8) << 30)))
        for name, (_, t) in list(self._std_rows.items()):
            small = t.numel() * t.element_size() <= limit
            if small and self._std_at_insert:
                continue                    # stays in _std_rows: resident, kept current by _obs_slab_written
            if small:
                self._std_keep[name] = t
            del self._std_rows[name]
        self._whole_batch = self._whole_batch_key = None      # (holds RowSources on the standardised copies)

    # ------------------------------------------------------------------ returns
    def _denorm_scalars(self, value_normalizer):
        """Device tensor [sigma, mu] of the value normaliser, or None (identity)."""
        if not (self._use_popart or self._use_valuenorm):
            return None
        if value_normalizer is None:
            raise ValueError("use_popart / use_valuenorm is set but compute_returns got no value_normalizer")
        if hasattr(value_normalizer, "denorm_scalars"):
            s = value_normalizer.denorm_scalars()
        else:  # a foreign normaliser with the reference's attribute names
            vn = value_normalizer
            if hasattr(vn, "running_mean"):
                m, sq = vn.running_mean, vn.running_mean_sq
            else:
                m, sq = vn.mean, vn.mean_sq
            debias = vn.debiasing_term.clamp(min=vn.epsilon)
            mean = m / debias
            var = (sq / debias - mean ** 2).clamp(min=1e-2)
            s = torch.stack([torch.sqrt(var).reshape(()), mean.reshape(())])
        return s.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _gae_flags(self, denorm):
        flags = 0
        if self._use_gae:
            flags |= _native.GAE_USE_GAE
        if self._use_proper_time_limits:
            flags |= _native.GAE_PROPER_TIME_LIMITS
        if denorm is not None:
            flags |= _native.GAE_DENORM
        if self._gae_exact:
            flags |= _native.GAE_EXACT
        return flags

    def compute_returns(self, next_value, value_normalizer=None, _scan_denorm=True):
        """GAE / discounted returns over the whole buffer in one kernel launch
        (reference shared_buffer.py:179-262).  ``next_value``: [N, A, 1] (or [N*A, 1]) array / tensor.
        ``_scan_denorm=False`` (used by SeparatedReplayBuffer for the one branch where the reference's
        separated buffer does not de-normalise): identity D() in the scan; the fused advantages are
        then marked stale so that they are recomputed with the normaliser."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        nv = self._dev(next_value).reshape(-1)
        if nv.numel() != N * A:
            raise ValueError("next_value has %d elements, expected %d" % (nv.numel(), N * A))
        denorm = self._denorm_scalars(value_normalizer) if _scan_denorm else None
        p = _native.ptr
        if self.algo in ("mat", "mat_dec") and self._use_gae and not self._use_proper_time_limits:
            # transformer branches (reference shared_buffer.py:222-232, :241-251): the advantages are the
            # GAE accumulator itself and, without a normaliser, the TD error uses the agent-mean value
            ev = self._timed("mappo_gae_mat_f32", 24 * T * N * A, settle=True)
            code = self._lib.mappo_gae_mat_f32(
                p(self.rewards), p(self.value_preds), p(nv), p(self.masks), p(self.returns), p(denorm),
                p(self.advantages), p(self.active_masks), p(self._adv_partials), T, N * A, A,
                float(self.gamma), float(self.gae_lambda), _native.GAE_DENORM if denorm is not None else 0,
                self._stream())
            self._timed_end(ev)
            _native.check(code, "mappo_gae_mat_f32")
            self._content_version += 1
            self._adv_fresh = True
            self._adv_denormalized = bool(self._use_popart or self._use_valuenorm)   # = what MATTrainer reads
            self._adv_is_gae = True
            self._stats_fresh = False
            self._adv_inputs = self._adv_input_versions()
            return
        self._adv_is_gae = False
        # algorithmic bytes: r, v, m reads + returns write (16 B) + advantages write + active read
        # (+8 B) [+ bad_masks read 4 B] per (t, n, a) element  
        # while profiling (bench.py), the launches are timed ALTERNATELY by an event pair recorded around the launch (what
        # every round quoted) and by the kernel's own begin / end timestamps (events attached to the dispatch:
        # mappo_gae_time_next_launch).  Not both on one launch: the dispatch-level events lengthen what the pair around them
        # sees (65 us against 58 us), and the pair itself also times two packets of the command processor -- 5-6 us on this
        # 50 us kernel against rocprofv3's trace of the same run.
        nbytes = (24 + (4 if self._use_proper_time_limits else 0)) * T * N * A
        ev = None
        if self._events is not None and not torch.cuda.is_current_stream_capturing():
            self._gae_profiled = getattr(self, "_gae_profiled", 0) + 1
            if self._gae_profiled % 2 == 0:
                torch.cuda._sleep(250000)           # (as in _timed: the kernel starts from a filled queue)
                slot = self._lib.mappo_gae_time_next_launch()
                if slot >= 0:
                    self._gae_dispatch = (getattr(self, "_gae_dispatch", []) + [(slot, nbytes)])[-64:]
            else:
                ev = self._timed("mappo_gae_f32", nbytes, settle=True)
        code = self._lib.mappo_gae_f32(
            p(self.rewards), p(self.value_preds), p(nv), p(self.masks),
            p(self.bad_masks) if self._use_proper_time_limits else None, p(self.returns), p(denorm),
            p(self.advantages), p(self.active_masks), p(self._adv_partials),
            T, N * A, float(self.gamma), float(self.gae_lambda), self._gae_flags(denorm), self._stream())
        self._timed_end(ev)
        _native.check(code, "mappo_gae_f32")
        self._content_version += 1
        self._adv_fresh = True
        self._adv_denormalized = denorm is not None      # what the fused advantages subtracted
        self._stats_fresh = False
        self._adv_inputs = self._adv_input_versions()

    def _adv_input_versions(self):
        """In-place torch writes to the fields the fused advantages / moments were formed from bump these
        counters (runners do edit them between compute_returns and train, e.g. ``active_masks[-1] = ...``);
        this class's own kernels write through raw pointers and leave them alone."""
        return (self.returns._version, self.value_preds._version, self.active_masks._version)

    def normalized_advantages(self, value_normalizer=None, all_reduce=None, denormalize=None):
        """What the prologue of the reference's ``R_MAPPO.train`` computes (r_mappo.py:179-187), as an
        ``AdvantageHandle``.  Uses the advantages / moments the GAE launch already produced; if the
        buffer changed since, recomputes them with one ``mappo_advantages_f32`` launch.
        ``all_reduce``: optional callable applied in place to the float64 [3] moment sums
        (sum, sum of squares, count) -- data-parallel training passes an RCCL all-reduce here so that
        mean / std are global-batch statistics on every rank.
        ``denormalize``: None follows the buffer's popart / valuenorm flags (MAPPO); False forces the raw
        ``returns - value_preds`` (HAPPO under ValueNorm, happo_trainer.py:180-183)."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        p = _native.ptr
        flagged = bool(self._use_popart or self._use_valuenorm)
        want = flagged if denormalize is None else bool(denormalize)
        if want and not flagged:
            raise ValueError("denormalize=True needs a buffer built with use_popart / use_valuenorm")
        if self._adv_fresh and getattr(self, "_adv_inputs", None) != self._adv_input_versions():
            self._adv_fresh = False        # returns / value_preds / active_masks were edited in place since
        if self._adv_fresh and self._adv_is_gae and denormalize is None:
            pass       # MAT: buffer.advantages is what the trainer normalises (mat_trainer.py:160-164)
        elif not self._adv_fresh or self._adv_denormalized != want:
            self._adv_is_gae = False
            denorm = self._denorm_scalars(value_normalizer) if want else None
            self._adv_denormalized = want
            code = self._lib.mappo_advantages_f32(p(self.returns), p(self.value_preds), p(denorm),
                                                  p(self.active_masks), p(self.advantages),
                                                  p(self._adv_partials), T, N * A, self._stream())
            _native.check(code, "mappo_advantages_f32")
            self._content_version += 1
            self._adv_fresh = True
            self._stats_fresh = False
            self._adv_inputs = self._adv_input_versions()
        if not self._stats_fresh or all_reduce is not None:
            _native.check(self._lib.mappo_adv_reduce(p(self._adv_partials), self._partial_rows,
                                                     p(self._adv_sums), self._stream()), "mappo_adv_reduce")
            if all_reduce is not None:
                all_reduce(self._adv_sums)
            _native.check(self._lib.mappo_adv_stats(p(self._adv_sums), p(self._adv_stats), self._stream()),
                          "mappo_adv_stats")
            self._stats_fresh = True
        return AdvantageHandle(self.advantages, self._adv_stats, self)

    # ------------------------------------------------------------------ samplers
    def _sampler_indices(self, n, mb, n_mb):
        """Index lists of ``n_mb`` minibatches of ``mb`` samples out of ``n``, back to back in one int64 device tensor
        (minibatch i = ``[i * mb, (i + 1) * mb)``): what ``rand = torch.randperm(n)`` and its slices are to the reference
        (shared_buffer.py:360-361, :511-512).

        ``--sampler_rng host`` (integer parity): exactly that -- the CPU generator's permutation, uploaded.
        ``device`` (default): K10, ``mappo_minibatch_indices`` -- a keyed bijection of [0, n) assigns every sample its
        slice and each slice comes out in ascending memory order: no sort (the radix sort behind ``torch.randperm`` was
        5 % of the north-star step), no random-order gathers.  The six round keys come from the CPU generator (no device
        sync), so ``torch.manual_seed`` fixes the minibatches."""
        if self._sampler_rng == "host" or n_mb > _native.MAX_MINIBATCHES or n_mb * mb > n or n_mb < 1 or mb < 1:
            return torch.randperm(n).to(self.device, non_blocking=True) if self._sampler_rng == "host" \
                else torch.randperm(n, device=self.device)
        keys = torch.randint(0, 1 << 32, (6,), dtype=torch.int64).tolist()
        if n_mb == 1 and mb == n:
            # one slice that takes every sample: K10 would emit 0 .. n - 1 (slices come out in ascending order) whatever
            # the keys -- three launches per epoch saved (the generator's draw above still happens: same random stream)
            if getattr(self, "_identity_idx", None) is None or self._identity_idx.numel() != n:
                self._identity_idx = torch.arange(n, dtype=torch.int64, device=self.device)
            return self._identity_idx
        idx = torch.empty(n_mb * mb, dtype=torch.int64, device=self.device)
        ws = torch.empty(self._lib.mappo_minibatch_workspace_ints(n, n_mb), dtype=torch.int32, device=self.device)
        _native.check(self._lib.mappo_minibatch_indices(n, mb, n_mb, (ctypes.c_uint32 * 6)(*keys), idx.data_ptr(),
                                                        ws.data_ptr(), self._stream()), "mappo_minibatch_indices")
        return idx

    def plan_epochs(self, n_epochs):
        """Kept for callers of the previous sampler (which drew the next permutation ahead on a side stream): the
        device sampler no longer sorts, there is nothing to overlap."""

    def _field_table(self, advantages):
        """(name, source tensor whose rows are gathered, trailing shape, first_only, adv mode)."""
        T = self.episode_length
        table = [
            ("share_obs", self.share_obs, False),
            ("obs", self.obs, False),
            ("rnn_states", self.rnn_states, True),
            ("rnn_states_critic", self.rnn_states_critic, True),
            ("actions", self.actions, False),
            ("value_preds", self.value_preds, False),
            ("returns", self.returns, False),
            ("masks", self.masks, False),
            ("active_masks", self.active_masks, False),
            ("action_log_probs", self.action_log_probs, False),
        ]
        stats = None
        self._adv_external = False
        if advantages is None:
            adv = None
        elif isinstance(advantages, AdvantageHandle):
            adv, stats = advantages.raw, advantages.stats
        else:
            # the reference's protocol: an array / tensor the caller owns.  It is gathered as a field of its
            # own and never enters the packed records: the upload below is a temporary whose address and
            # version counter can repeat across calls, so it cannot key the record cache
            self._adv_external = True
            adv = self._dev(advantages)
            if adv.numel() != T * self.n_rollout_threads * self.num_agents:
                raise ValueError("advantages has the wrong number of elements")
        table.append(("advantages", adv, False))
        table.append(("available_actions", self.available_actions, False))
        for name, tensor in self.extra_fields.items():
            table.append((name, tensor, False))
        return table, stats

    supports_standardized_obs = True   # generators take standardize_obs=True (see feed_forward_generator)

    @staticmethod
    def _standardize_width_ok(width):
        """Widths the standardising gather has kernels for (pick_shape, csrc/mappo_norm.hip): rows of whole
        float4s up to 2048 floats, any other width up to 1536."""
        return (width % 4 == 0 and width <= 2048) or width <= 1536

    def can_standardize_obs(self):
        """Whether ``standardize_obs=True`` is available for BOTH observation fields of this buffer (vector
        observations of a supported width).  The trainer folds the input LayerNorm into the sampler only then;
        otherwise (e.g. simple_spread with >= 19 agents: share_obs 2166 wide) it gathers plain rows and the policy
        applies its own feature_norm."""
        for field in (self.share_obs, self.obs):
            tail = tuple(field.shape[3:])
            if len(tail) != 1 or not self._standardize_width_ok(int(tail[0])):
                return False
        return True

    # fields at most this wide are packed into one record per sample before sampling
    _NARROW = 8
    _MAX_RECORD = 32

    def _pack_records(self, table):
        """Pack the narrow fields (per-sample scalars, available_actions, ...) into one
        [T*N*A, RW] record array (mappo_pack_records) -> (records, RW, {name: (offset, width)}).
        Done once per generator call, i.e. per epoch: ~1 GB of traffic against ~47 GB per gather."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        layout, fields, off, versions = {}, [], 0, []
        if os.environ.get("MAPPO_PACK_RECORDS", "1") == "0":
            return None, 0, {}
        for name, src, is_state in table:
            if src is None or is_state or name in ("share_obs", "obs"):
                continue   # observations / RNN states keep their own (wide or standardising) path
            if name == "advantages" and self._adv_external:
                continue
            tail = tuple(src.shape[3:])
            width = int(np.prod(tail)) if tail else 1
            if width > self._NARROW or off + width > self._MAX_RECORD:
                continue
            layout[name] = (off, width)
            fields.append(_native.RecordField(src.data_ptr(), None, width, off, 0, 0))
            versions.append(src._version)      # in-place torch writes to the field (setitem, copy_, ...)
            off += width
        if not fields:
            return None, 0, {}
        if self._sampler_rng == "device":
            # the device sampler walks the records in ascending memory order: dense records (the 12 payload floats of the
            # north star in 48 bytes instead of 64) are read line by line with nothing skipped (PMC traffic 1.15 x -> <= 1.0 x
            # of the algorithmic bytes)
            rw = (off + 3) // 4 * 4
        else:
            rw = 4
            while rw < off:      # random permutations: 16 / 32 / 64 / 128-byte records never straddle more sectors than needed
                rw *= 2
        rows = T * N * A
        # the packed fields only change when the buffer is written -- by this class's kernels
        # (_content_version) or by in-place torch ops on the field tensors (tensor._version): within
        # one train() every epoch reuses the records of the first
        key = (self._content_version, tuple(versions), tuple((f.src, f.width, f.offset) for f in fields))
        if getattr(self, "_records_key", None) == key:
            return self._records, rw, layout
        if getattr(self, "_records", None) is None or self._records.numel() < rows * rw:
            self._records = torch.empty(rows * rw, dtype=torch.float32, device=self.device)
        arr = (_native.RecordField * len(fields))(*fields)
        _native.check(self._lib.mappo_pack_records(arr, len(fields), self._records.data_ptr(), rw, rows,
                                                   self._stream()), "mappo_pack_records")
        self._records_key = key
        return self._records, rw, layout

    supports_lazy_obs = True           # generators take lazy_obs=True (see feed_forward_generator)

    def _obs_rows(self, name, standardized):
        """The [T*N*A, D] matrix the fused trunk kernels read the rows of field ``name`` from: the field itself (a view)
        or, for networks with an input LayerNorm, its row-standardised copy -- made once per buffer content (the
        observations do not change during the ppo epochs) by ``mappo_standardize_rows``."""
        field = getattr(self, name)
        T = self.episode_length
        rows = field[:T].reshape(T * self.n_rollout_threads * self.num_agents, -1)
        if not standardized:
            return rows
        key = self._obs_key(name)
        hit = self._std_rows.get(name)
        if hit is None or hit[0] != key:
            out = hit[1] if hit is not None else self._std_keep.pop(name, None)
            hit = (key, self._standardize_field(rows, out))
            self._std_rows[name] = hit
            self.std_full_passes += 1
        return hit[1]

    def _standardize_field(self, rows2d, out=None, eps=1e-5):
        """(x - mean) / sqrt(var + eps) of every row of ``rows2d`` into a [rows, D padded to 4] matrix: the full pass, through
        the SAME device code that keeps single slabs current inside K2's launch (mappo_slab_copy_std with no copy slabs), so
        that a slab and the full pass agree bit for bit."""
        n, D = rows2d.shape
        ld = (D + 3) // 4 * 4 if os.environ.get("MAPPO_PAD_STANDARDIZED", "1") != "0" else D
        if out is None or tuple(out.shape) != (n, ld) or out.dtype != rows2d.dtype or out.device != rows2d.device:
            out = torch.empty((n, ld), dtype=rows2d.dtype, device=rows2d.device)
        assert rows2d.is_contiguous()
        done, per = 0, -(-n // _native.MAX_STD_SLABS)
        std = []
        while done < n:         # (cut into up to four descriptors: more workgroups on a large field)
            m = min(per, n - done)
            std.append(_native.StdSlab(rows2d[done:done + m].data_ptr(), out[done:done + m].data_ptr(), m, D, ld, float(eps)))
            done += m
        sarr = (_native.StdSlab * len(std))(*std)
        _native.check(self._lib.mappo_slab_copy_std(None, 0, sarr, len(std), self._stream()), "mappo_slab_copy_std")
        return out

    def _gather(self, table, stats, idx, mb, chunk_len=None, standardize_obs=False, packed=None, lazy_obs=False):
        """One minibatch -> the 12-tuple of fresh device tensors: wide fields through the fused tile
        gather / standardising gather (K3 / K4), narrow fields through one record gather.  ``lazy_obs``: the two
        observation fields are not gathered; their places hold ``RowSource`` objects (source matrix + this minibatch's
        indices) for the fused trunk kernels, which read the rows straight from the buffer."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        rows_out = mb if chunk_len is None else mb * chunk_len
        records, rw, layout = packed if packed is not None else (None, 0, {})
        outs, fields, rec_fields = [], [], []
        for name, src, is_state in table:
            if src is None:
                outs.append(None)
                continue
            tail = tuple(src.shape[3:])
            width = int(np.prod(tail)) if tail else 1
            if lazy_obs and name in ("share_obs", "obs") and len(tail) == 1:
                from onpolicy.algorithms.utils.fused_mlp import RowSource
                chunk = None if chunk_len is None else (int(chunk_len), T, N, A)
                outs.append(RowSource(self._obs_rows(name, standardize_obs), idx, chunk, standardized=standardize_obs,
                                      width=width))
                continue
            if is_state and not self._recurrent:
                outs.append(src[0, 0, 0].expand((mb,) + tail))  # zeros, no traffic
                continue
            first_only = 1 if (is_state and chunk_len is not None) else 0
            n_rows = mb if (first_only or chunk_len is None) else rows_out
            dst = torch.empty((n_rows,) + tail, dtype=torch.float32, device=self.device)
            normalize = 1 if (name == "advantages" and stats is not None) else 0
            if name in layout:
                off, w = layout[name]
                rec_fields.append(_native.RecordField(None, dst.data_ptr(), w, off, normalize, 0))
            else:
                standardize = 1 if (standardize_obs and name in ("share_obs", "obs") and len(tail) == 1) else 0
                fields.append(_native.Field(src.data_ptr(), dst.data_ptr(), width, first_only, normalize,
                                            standardize))
            outs.append(dst)
        sp = None if stats is None else stats.data_ptr()
        # algorithmic bytes: every gathered row is read once and written once, + the int64 indices
        nbytes = sum(2 * 4 * f.width * (mb if (f.first_only or chunk_len is None) else rows_out)
                     for f in fields) + sum(2 * 4 * f.width * rows_out for f in rec_fields) + 8 * mb
        ev = self._timed("mappo_gather_rows" if chunk_len is None else "mappo_gather_chunks", nbytes)
        if fields:
            arr = (_native.Field * len(fields))(*fields)
            if chunk_len is None:
                code = self._lib.mappo_gather_rows(arr, len(fields), idx.data_ptr(), mb, sp, self._stream())
                _native.check(code, "mappo_gather_rows")
            else:
                code = self._lib.mappo_gather_chunks(arr, len(fields), idx.data_ptr(), mb, chunk_len, T, N, A,
                                                     sp, self._stream())
                _native.check(code, "mappo_gather_chunks")
        if rec_fields:
            arr = (_native.RecordField * len(rec_fields))(*rec_fields)
            code = self._lib.mappo_gather_records(records.data_ptr(), rw, arr, len(rec_fields), idx.data_ptr(), mb,
                                                  0 if chunk_len is None else chunk_len, T, N, A, sp,
                                                  self._stream())
            _native.check(code, "mappo_gather_records")
        self._timed_end(ev)
        return tuple(outs)

    # One minibatch that takes every sample (chunk) in memory order -- the device sampler's single slice: the 12-tuple is
    # the same in every epoch of a train() -- nothing writes the buffer in between -- so it is gathered once and handed out
    # again (nine of ten gathers and row tables per north-star step saved).  Keyed on the buffer's content (the record
    # cache's key + the fields' tensor versions), dropped with the other scratch of an update.
    # CONTRACT: the tensors of such a tuple are SHARED between the epochs of one train() -- read-only for the caller.  (The
    # reference's generators yield independent copies, shared_buffer.py:379-396; they still do here for the reference
    # protocol -- an external ``advantages`` array never takes this route -- and for several minibatches per epoch.)  An
    # in-place edit of a yielded tensor is detected through its version counter and the batch is gathered again, so a
    # trainer that does edit them gets the reference's semantics at the price of the gather.
    def _whole_batch_ok(self, rand, packed):
        return rand is getattr(self, "_identity_idx", None) and packed[0] is not None and not self._adv_external

    @staticmethod
    def _output_versions(outs):
        return tuple(t._version for t in outs if torch.is_tensor(t))

    def _whole_batch_tuple(self, table, stats, rand, mb, chunk_len, standardize_obs, lazy_obs, packed):
        key = (self._records_key, chunk_len, bool(standardize_obs), bool(lazy_obs), None if stats is None else stats.data_ptr(),
               tuple((src.data_ptr(), src._version) for _, src, _ in table if src is not None))
        if self._whole_batch_key == key and self._output_versions(self._whole_batch) == self._whole_batch_versions:
            self.whole_batch_reuses += 1
            return self._whole_batch
        self._whole_batch = self._gather(table, stats, rand, mb, chunk_len=chunk_len, standardize_obs=standardize_obs,
                                         packed=packed, lazy_obs=lazy_obs)
        self._whole_batch_key = key
        self._whole_batch_versions = self._output_versions(self._whole_batch)
        return self._whole_batch

    # The trainer's private route (lazy_obs), one minibatch per epoch, device sampler: the single slice is every sample in
    # MEMORY ORDER, i.e. row r of a "gathered" field is row r of the buffer field itself.  Nothing is packed or gathered: the
    # tuple holds VIEWS of the buffer fields ([:T] flattened to [rows, dim]) and one fresh tensor, the normalised
    # advantages (mappo_adv_normalize, 8 bytes per sample) -- the north-star step loses its record pack + record gather
    # (1.0 ms, 2.9 GB of traffic; round 6).  Only on this route: its consumer, R_MAPPO.ppo_update, never writes its inputs.
    # Callers of the public protocol (neither lazy_obs nor standardize_obs) keep the copies and the edit detection of
    # _whole_batch_tuple.  With standardize_obs alone (networks that take tensors: hidden 512) the observation elements of the
    # tuple are the resident standardised copies themselves -- rows padded to 16 bytes, which K15's aligned loads want anyway
    # (Hanabi's 1285 / 1385-wide rows: first-layer forward 4.51 -> 4.16 ms, no standardising gather).
    def _whole_batch_views_ok(self, rand, mb, batch_size):
        return rand is getattr(self, "_identity_idx", None) and mb == batch_size and not self._adv_external \
            and os.environ.get("MAPPO_WHOLE_BATCH_VIEWS", "1") != "0"

    def _whole_batch_views(self, table, stats, rand, mb, standardize_obs, lazy_obs=True):
        T = self.episode_length
        key = ("views", self._content_version, bool(standardize_obs), bool(lazy_obs), None if stats is None else stats.data_ptr(),
               tuple((src.data_ptr(), src._version) for _, src, _ in table if src is not None))
        if self._whole_batch_key == key:
            self.whole_batch_reuses += 1
            return self._whole_batch
        from onpolicy.algorithms.utils.fused_mlp import RowSource
        outs = []
        for name, src, is_state in table:
            if src is None:
                outs.append(None)
                continue
            tail = tuple(src.shape[3:])
            if name in ("share_obs", "obs") and len(tail) == 1 and lazy_obs:
                width = int(tail[0])
                outs.append(RowSource(self._obs_rows(name, standardize_obs), rand, None, standardized=standardize_obs,
                                      width=width))
            elif name in ("share_obs", "obs") and len(tail) == 1 and standardize_obs:
                # networks that take tensors (hidden sizes without a fused trunk: Hanabi's 512): the resident standardised copy
                # itself, in memory order -- [rows, D padded to a multiple of 4] with zero columns behind the data (K15 takes the
                # padded rows through its aligned loads; tall_linear cuts them off for the library GEMM).  Nothing is gathered.
                outs.append(self._obs_rows(name, True))
            elif is_state and not self._recurrent:
                outs.append(src[0, 0, 0].expand((mb,) + tail))           # zeros, no traffic
            elif name == "advantages" and stats is not None:
                dst = torch.empty((mb,) + tail, dtype=torch.float32, device=self.device)
                ev = self._timed("mappo_adv_normalize", 8 * mb)
                _native.check(self._lib.mappo_adv_normalize(src.data_ptr(), stats.data_ptr(), dst.data_ptr(), mb,
                                                            self._stream()), "mappo_adv_normalize")
                self._timed_end(ev)
                outs.append(dst)
            else:
                outs.append(src[:T].reshape((mb,) + tail))               # a view: [T, N, A, ...] is contiguous
        self._whole_batch = tuple(outs)
        self._whole_batch_key = key
        self._whole_batch_versions = ()
        return self._whole_batch

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None,
                               standardize_obs=False, lazy_obs=False):
        """Minibatches of independent (t, n, a) samples for MLP policies
        (reference shared_buffer.py:340-400).  Yields
        (share_obs, obs, rnn_states, rnn_states_critic, actions, value_preds, returns, masks,
        active_masks, old_action_log_probs, adv_targ, available_actions) as device tensors.

        ``standardize_obs=True`` (all three generators): share_obs / obs rows come out standardised,
        ``(x - mean(x)) / sqrt(var(x) + 1e-5)`` per row -- the parameter-free half of the networks'
        input LayerNorm, computed while the row is being copied.  The trainer asks for it when the
        policy can fold the LayerNorm's affine half into its first Linear (MLPBase).

        ``lazy_obs=True`` (all three generators; the trainer's private route): share_obs / obs come out as
        ``fused_mlp.RowSource`` objects instead of gathered tensors -- the fused trunk kernels (K9) read the rows from the
        buffer through the minibatch's indices, so the [mb, obs_dim] copies are never written."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        batch_size = N * T * A
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch, (
                "PPO requires the number of processes ({}) "
                "* number of steps ({}) * number of agents ({}) = {} "
                "to be greater than or equal to the number of PPO mini batches ({})."
                "".format(N, T, A, batch_size, num_mini_batch))
            mini_batch_size = batch_size // num_mini_batch
        rand = self._sampler_indices(batch_size, mini_batch_size, num_mini_batch)
        table, stats = self._field_table(advantages)
        if num_mini_batch == 1 and (lazy_obs or standardize_obs) and self._whole_batch_views_ok(rand, mini_batch_size, batch_size) \
                and (lazy_obs or self.can_standardize_obs()):
            # (standardize_obs / lazy_obs are this implementation's extensions of the protocol: only its own trainer passes them)
            yield self._whole_batch_views(table, stats, rand, mini_batch_size, standardize_obs, lazy_obs)
            return
        packed = self._pack_records(table)
        if num_mini_batch == 1 and self._whole_batch_ok(rand, packed):
            yield self._whole_batch_tuple(table, stats, rand, mini_batch_size, None, standardize_obs, lazy_obs, packed)
            return
        for i in range(num_mini_batch):
            idx = rand[i * mini_batch_size:(i + 1) * mini_batch_size]
            yield self._gather(table, stats, idx, mini_batch_size, standardize_obs=standardize_obs, packed=packed,
                               lazy_obs=lazy_obs)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, standardize_obs=False, lazy_obs=False):
        """Minibatches of length-L chunks for truncated BPTT (reference shared_buffer.py:499-608):
        sequence fields come out as [L*mb, dim] (row l*mb + j), RNN states as [mb, R, H] (chunk
        start only)."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        batch_size = N * T * A
        data_chunks = batch_size // data_chunk_length
        mini_batch_size = data_chunks // num_mini_batch
        rand = self._sampler_indices(data_chunks, mini_batch_size, num_mini_batch)
        table, stats = self._field_table(advantages)
        packed = self._pack_records(table)
        if num_mini_batch == 1 and mini_batch_size == data_chunks and self._whole_batch_ok(rand, packed):
            yield self._whole_batch_tuple(table, stats, rand, mini_batch_size, data_chunk_length, standardize_obs, lazy_obs,
                                          packed)
            return
        for i in range(num_mini_batch):
            idx = rand[i * mini_batch_size:(i + 1) * mini_batch_size]
            yield self._gather(table, stats, idx, mini_batch_size, chunk_len=data_chunk_length,
                               standardize_obs=standardize_obs, packed=packed, lazy_obs=lazy_obs)

    def naive_recurrent_generator(self, advantages, num_mini_batch, standardize_obs=False, lazy_obs=False):
        """Whole-trajectory minibatches (reference shared_buffer.py:402-497): a chunk gather with
        L = T over a permutation of the N*A trajectories."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        batch_size = N * A
        assert batch_size >= num_mini_batch, (
            "PPO requires the number of processes ({})* number of agents ({}) "
            "to be greater than or equal to the number of "
            "PPO mini batches ({}).".format(N, A, num_mini_batch))
        num_envs_per_batch = batch_size // num_mini_batch
        if batch_size % num_envs_per_batch == 0:
            perm = self._sampler_indices(batch_size, num_envs_per_batch, batch_size // num_envs_per_batch)
        else:       # the reference's last, shorter slice (shared_buffer.py:417): a full permutation
            perm = torch.randperm(batch_size).to(self.device) if self._sampler_rng == "host" \
                else torch.randperm(batch_size, device=self.device)
        table, stats = self._field_table(advantages)
        packed = self._pack_records(table)
        for start in range(0, batch_size, num_envs_per_batch):
            idx = perm[start:start + num_envs_per_batch]
            yield self._gather(table, stats, idx, idx.numel(), chunk_len=T, standardize_obs=standardize_obs,
                               packed=packed, lazy_obs=lazy_obs)

    def feed_forward_generator_transformer(self, advantages, num_mini_batch=None, mini_batch_size=None):
        """Minibatches of whole (t, n) agent groups for the transformer policies (reference
        shared_buffer.py:264-338): a permutation of the T*N env steps; every drawn step contributes its
        A agents as consecutive rows, so a minibatch is [mb*A, dim] with row i*A + a.  Same fused row
        gather as feed_forward_generator over source rows idx*A + a (computed on the device)."""
        T, N, A = self.episode_length, self.n_rollout_threads, self.num_agents
        batch_size = N * T
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch, (
                "PPO requires the number of processes ({}) "
                "* number of steps ({}) = {} "
                "to be greater than or equal to the number of PPO mini batches ({})."
                "".format(N, T, batch_size, num_mini_batch))
            mini_batch_size = batch_size // num_mini_batch
        rand = self._sampler_indices(batch_size, mini_batch_size, num_mini_batch)
        table, stats = self._field_table(advantages)
        packed = self._pack_records(table)
        agents = torch.arange(A, device=self.device)
        for i in range(num_mini_batch):
            idx = rand[i * mini_batch_size:(i + 1) * mini_batch_size]
            rows = (idx[:, None] * A + agents[None, :]).reshape(-1).contiguous()
            yield self._gather(table, stats, rows, mini_batch_size * A, packed=packed)
