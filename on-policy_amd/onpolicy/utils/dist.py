"""Data parallelism over rollout threads: one process per GPU, RCCL (torch.distributed backend
"nccl" on ROCm) over xGMI; gloo on CPU for tests.

The reference is single-process / single-device (no torch.distributed call sites anywhere), so this
layer is new.  The MAPPO update shards naturally over ``n_rollout_threads``: every rank owns
N / world rollout threads in its own HBM buffer, runs GAE and the samplers locally, and the only
exchange is, per ``ppo_update``:

  1. a few float64 scalars that must be global before the loss is formed (masked-loss
     denominators; ValueNorm batch moments are reduced by the trainer the same way), and
  2. ONE sum all-reduce of the actor+critic gradients.  The gradients of both networks are gathered
     into a single flat float32 bucket with one batched copy (and ``param.grad`` become views of the
     reduced bucket), so there is exactly one collective: 151 KB for the 8-agent MPE MLP,
     454 KB for the SMAC GRU, 9.8 MB for Hanabi.  xGMI is a full mesh of point-to-point links, so
     buckets this small are latency-bound; a single collective per step is what matters.

plus one all-reduce of three float64 advantage moments per ``train()``.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    """True when a process group is up and there is more than one rank -- or MAPPO_FORCE_DIST=1, which
    sends a single rank through the same bucket / collective code path (used to smoke-test RCCL on a
    one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MAPPO_FORCE_DIST", "0") == "1"


def init_from_env(device=None):
    """Initialise the default process group from torchrun-style environment variables
    (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = os.environ.get("MAPPO_FORCE_DIST", "0") == "1"
    if (world <= 1 and not forced) or (dist.is_available() and dist.is_initialized()):
        return world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (required by RCCL on these hosts)
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
    # MAPPO_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices); used
    # to exercise the multi-rank device path on a single-GPU box
    backend = os.environ.get("MAPPO_DIST_BACKEND", backend)
    kwargs = {}
    if backend == "nccl":
        kwargs["device_id"] = torch.device(device)
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world, **kwargs)
    return world


def shard_threads(n_rollout_threads, rank=None, world_size=None):
    """Rollout threads [lo, hi) owned by ``rank``: contiguous, sizes differ by at most one."""
    if world_size is None:
        world_size = dist.get_world_size() if is_distributed() else 1
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    base, extra = divmod(n_rollout_threads, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class DataParallel(object):
    """Gradient / statistics exchange for one (actor, critic) pair."""

    def __init__(self, actor, critic, device, group=None):
        self.group = group
        self.active = is_distributed()       # collectives are issued (world size may still be 1 when forced)
        self.world_size = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.device = device
        self._flat = None
        self._timing = None                  # list of (start, end) events around the gradient all-reduce
        self._mb_ws = None                   # float64 scratch of minibatch_scales (per device, made on first use)
        self._scales_pending = []            # prologues of coming minibatches in flight (begin_scales)
        self._scales_done = None             # prologue of the current minibatch, reusable while its tensors are unchanged
        self.scalar_collectives = 0          # scalar all-reduces issued by begin_scales (tests / bench count them)
        self.scales_reused = 0               # updates served from the cached prologue
        if self.active:
            params = [p for net in (actor, critic) for p in net.parameters() if p.requires_grad]
            total = sum(p.numel() for p in params)
            self._flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            self._zeros = torch.zeros(total, dtype=torch.float32, device=params[0].device)   # stand-in for absent gradients
            self._params = params
            self._check_replicas()

    def _check_replicas(self):
        """All ranks must start from identical parameters (same seed => same CPU init stream)."""
        with torch.no_grad():
            digest = torch.stack([p.detach().double().sum() for p in self._params]).sum().reshape(1)
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if (hi - lo).abs().item() > 1e-6 * max(1.0, hi.abs().item()):
                raise RuntimeError("data-parallel replicas were initialised with different parameters; "
                                   "seed every rank identically before building the policy")

    def all_reduce(self, tensor):
        """In-place sum over ranks (no-op for one process)."""
        if self.active:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def zero_grad(self, *optimizers):
        """Gradients start from None in both modes: autograd then ASSIGNS the first gradient of every parameter instead
        of launching one accumulate kernel per parameter into a zeroed bucket view (16 tiny launches per update -- 3 % of
        a step on an 8-GPU shard of the north star)."""
        for opt in optimizers:
            opt.zero_grad(set_to_none=True)
        if self.active:
            for p in self._params:
                p.grad = None

    def all_reduce_grads(self):
        """Gather the parameters' gradients into the flat bucket (one batched copy), sum it over the ranks with ONE
        collective, and make every ``param.grad`` a view of the reduced bucket again (clipping and the optimiser read
        those)."""
        if not self.active:
            return
        pieces = []
        off = 0
        for p in self._params:
            n = p.numel()
            pieces.append(self._zeros[off:off + n] if p.grad is None else p.grad.reshape(-1))
            off += n
        torch.cat(pieces, out=self._flat)
        timed = self._timing is not None and self._flat.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        if timed:
            ev[1].record()
            self._timing.append(ev)
        off = 0
        for p in self._params:
            n = p.numel()
            p.grad = self._flat[off:off + n].view_as(p)
            off += n

    def all_reduce_from(self, grads):
        """``all_reduce_grads`` for gradients a captured graph left in ``grads`` (one entry per parameter of ``_params``, None
        = no gradient): batched copy into the flat bucket + ONE collective.  ``param.grad`` is not touched -- the graph that
        consumes the reduced gradients reads the bucket's views directly (algorithms/r_mappo/update_graph.py)."""
        if not self.active:
            return
        pieces, off = [], 0
        for p, g in zip(self._params, grads):
            n = p.numel()
            pieces.append(self._zeros[off:off + n] if g is None else g.reshape(-1))
            off += n
        torch.cat(pieces, out=self._flat)
        timed = self._timing is not None and self._flat.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        if timed:
            ev[1].record()
            self._timing.append(ev)

    def time_collectives(self, on=True):
        """Start (or stop) recording an event pair around every gradient all-reduce (bench.py)."""
        self._timing = [] if on else None

    def collective_times(self):
        """-> (number of gradient all-reduces recorded, their total milliseconds on the device, bucket bytes).
        The caller must have synchronised the device."""
        ev = self._timing or []
        ms = sum(a.elapsed_time(b) for a, b in ev)
        return len(ev), ms, (0 if self._flat is None else self._flat.numel() * 4)

    def loss_weights(self, active_masks, policy_masked, value_masked):
        """(w_actor, w_critic): local / global denominators of the masked means
        (reference r_mappo.py:135-139, 84-87)."""
        if not self.active:
            return 1.0, 1.0
        local = torch.stack([active_masks.detach().sum().double(),
                             torch.full((), float(active_masks.shape[0]), dtype=torch.float64,
                                        device=active_masks.device)])
        total = local.clone()
        self.all_reduce(total)
        w = (local / total).float()
        return (w[0] if policy_masked else w[1]), (w[0] if value_masked else w[1])

    def minibatch_stats(self, active_masks, return_batch, policy_masked, value_masked):
        """ONE collective for everything a minibatch needs globally before its loss is formed:
        -> (w_actor, w_critic, (mean, mean_sq) of the returns over the global minibatch).
        The loss weights are local / global denominators of the masked means (reference
        r_mappo.py:135-139, 84-87); the moments feed the ValueNorm / PopArt update (r_mappo.py:65)."""
        am = active_masks.detach()
        ret = return_batch.detach()
        # (the row count goes up with torch.full: torch.tensor(x, device=...) is a blocking copy that drains the stream)
        local = torch.stack([am.sum().double(), torch.full((), float(am.shape[0]), dtype=torch.float64, device=am.device),
                             ret.sum().double(), (ret * ret).sum().double()])
        total = local.clone()
        self.all_reduce(total)
        w = (local[:2] / total[:2]).float()
        mean = (total[2] / total[1]).float().reshape(1)
        mean_sq = (total[3] / total[1]).float().reshape(1)
        return (w[0] if policy_masked else w[1]), (w[0] if value_masked else w[1]), (mean, mean_sq)

    # ---- the scalar exchange of an update, off the critical path -------------------------------------------------------
    # What a minibatch needs globally before its loss is formed -- sum of active masks, rows, sum / sum of squares of the
    # returns (r_mappo.py:135-139, :84-87, :65) -- depends on the minibatch's rows only, not on the parameters.  So:
    #   * ``begin_scales`` launches the local sums and issues the (asynchronous) all-reduce as soon as the minibatch
    #     exists; the trainer calls it for minibatch i + 1 BEFORE it runs update i, so the collective travels under
    #     update i's kernels (its 32 bytes are pure latency on xGMI) and update i + 1 finds the result waiting;
    #   * a minibatch that is handed out again unchanged (the whole-batch tuple of a one-minibatch epoch: the same tensor
    #     objects in all ppo_epoch epochs) reuses the finished scales -- ONE scalar collective per train() instead of one
    #     per update.  Identity of the tensor OBJECTS (held here, so their memory cannot be recycled) + their version
    #     counters key the cache; a fresh gather is a new object and never hits.
    def _scales_ok(self, am, ret):
        return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (am, ret)) \
            and am.numel() == ret.numel() and am.numel() > 0 and os.environ.get("MAPPO_FUSED_PROLOGUE", "1") != "0"

    @staticmethod
    def _same_batch(rec, am, ret, key):
        return rec is not None and rec["am"] is am and rec["ret"] is ret and rec["key"] == key and \
            rec["versions"] == (am._version, ret._version)

    def begin_scales(self, active_masks, return_batch, policy_masked, value_masked):
        """Start the prologue of a minibatch (local sums + all-reduce in flight) -> True if one is now pending or cached
        for exactly these tensors, False if they do not qualify for the kernels."""
        am, ret = active_masks.detach(), return_batch.detach()
        if not self._scales_ok(am, ret):
            return False
        key = (bool(policy_masked), bool(value_masked))
        if self._same_batch(self._scales_done, active_masks, return_batch, key) or \
                any(self._same_batch(r, active_masks, return_batch, key) for r in self._scales_pending):
            return True
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        dev = am.device
        if self._mb_ws is None or self._mb_ws.device != dev:
            self._mb_ws = torch.empty(lib.mappo_minibatch_sums_workspace_doubles(), dtype=torch.float64, device=dev)
        stream = _native.stream_of(dev)
        local = torch.empty(4, dtype=torch.float64, device=dev)
        _native.check(lib.mappo_minibatch_sums(p(am), p(ret), am.numel(), p(local), p(self._mb_ws), stream),
                      "mappo_minibatch_sums")
        total, work = local, None
        if self.active:
            total = local.clone()
            work = dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.scalar_collectives += 1
        self._scales_pending.append({"am": active_masks, "ret": return_batch, "key": key,
                                     "versions": (active_masks._version, return_batch._version),
                                     "local": local, "total": total, "work": work})
        return True

    def minibatch_scales(self, active_masks, return_batch, policy_masked, value_masked):
        """The loss denominators and the returns' batch moments for the fused loss on a HIP device
        (``mappo_minibatch_sums`` / ``mappo_minibatch_scales`` + one collective in a multi-rank job) -> float32 [8] device
        tensor ``[1 / global policy denominator, 1 / global value denominator, 1 / local policy denominator (twice),
        1 / local value denominator, 1 / local rows, mean, mean of squares of the returns over the global minibatch]``,
        or None when the columns do not qualify (then ``minibatch_stats`` / PyTorch do the work).  Picks up what
        ``begin_scales`` started for these tensors (or the cached result of the same, unchanged tensors)."""
        if not self.begin_scales(active_masks, return_batch, policy_masked, value_masked):
            return None
        key = (bool(policy_masked), bool(value_masked))
        if self._same_batch(self._scales_done, active_masks, return_batch, key):
            self.scales_reused += 1
            return self._scales_done["out"]
        rec = next(r for r in self._scales_pending if self._same_batch(r, active_masks, return_batch, key))
        self._scales_pending = [r for r in self._scales_pending if r is not rec]      # (identity: == would compare tensors)
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        if rec["work"] is not None:
            rec["work"].wait()      # orders the current stream behind the collective (RCCL: no host block)
        out = torch.empty(8, dtype=torch.float32, device=rec["local"].device)
        _native.check(lib.mappo_minibatch_scales(p(rec["local"]), p(rec["total"]), int(key[0]), int(key[1]), p(out),
                                                 _native.stream_of(out.device)), "mappo_minibatch_scales")
        rec["out"] = out
        rec["work"] = None
        self._scales_done = rec
        return out

    def drop_scales(self):
        """Forget cached / pending prologues (end of a train(): the tensors they hold belong to the sampler)."""
        for rec in self._scales_pending:
            if rec["work"] is not None:
                rec["work"].wait()
        self._scales_pending = []
        self._scales_done = None

    def average_info(self, totals):
        """Logged scalars: mean over ranks of the per-rank means."""
        if self.active:
            self.all_reduce(totals)
            totals = totals / self.world_size
        return totals
