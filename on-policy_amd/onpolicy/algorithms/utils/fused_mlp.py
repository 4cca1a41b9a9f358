"""Fused hidden-64 trunk (K9, ``mappo_mlp_forward`` / ``mappo_mlp_backward``): the whole MLPBase chain plus the output
Linear of an actor / critic as one forward kernel and two backward kernels on the matrix cores (float32 products in the
arithmetic ``--matrix_arithmetic`` names, include/mappo_hip.h MAPPO_ARITH_*), reading the
observation rows of a sampler minibatch straight from the rollout buffer through the sampler's index list.

Reference modules it evaluates (same parameters, same function): onpolicy/algorithms/utils/mlp.py:6-58 (MLPLayer /
MLPBase: [LayerNorm] -> (Linear -> Tanh | ReLU -> LayerNorm) x (1 + layer_N)), the Categorical head's Linear
(distributions.py:55-68) and the critic's v_out (r_actor_critic.py:147-175).

``RowSource`` is what the buffer's samplers hand out instead of a gathered ``[mb, obs_dim]`` tensor when asked for
``lazy_obs=True``: the source matrix (for networks with an input LayerNorm: a copy of the observation field standardised
once per train()) and the minibatch's indices.  Networks that cannot take it call ``materialize()`` and get the tensor the
eager (standardising) gather would have produced.

Autograd sees one node per network (``_FusedTrunkFn``); the input LayerNorm's affine half is folded into the first
Linear with ordinary torch ops in front of it (``LN(x) W^T + b = xhat (W * gamma)^T + (b + W beta)``), so gamma / beta
receive their gradients from autograd.  ``MAPPO_FUSED_MLP=0`` disables the path.
"""
import os

import torch
import torch.nn as nn

from onpolicy import _native

HIDDEN = 64
MAX_LAYERS = 3
MAX_OUT = 64


def enabled():
    return os.environ.get("MAPPO_FUSED_MLP", "1") != "0"


# ---- launch timing (bench.py): HIP events on the launch stream around every K9 call while profiling is on
_PROFILE = None


def profile(on=True):
    """Start (or stop) recording an event pair + the algorithmic FLOPs / bytes of every fused-trunk launch."""
    global _PROFILE
    _PROFILE = {} if on else None


def profile_times():
    """-> {name: (launches, mean milliseconds, total FLOPs, total algorithmic HBM bytes)}; the caller has synchronised."""
    out = {}
    for name, recs in (_PROFILE or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _, _ in recs)
        out[name] = (len(recs), ms / max(1, len(recs)), sum(f for _, _, f, _ in recs), sum(b for _, _, _, b in recs))
    return out


class _Timed(object):
    def __init__(self, name, flops, nbytes):
        self.rec = None
        # (launches recorded into a HIP graph -- the captured rollout step -- have no events of their own to time)
        if _PROFILE is not None and not torch.cuda.is_current_stream_capturing():
            self.rec = (name, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), flops, nbytes)

    def __enter__(self):
        if self.rec:
            self.rec[1].record()

    def __exit__(self, *exc):
        if self.rec:
            self.rec[2].record()
            _PROFILE.setdefault(self.rec[0], []).append(self.rec[1:])


def _work(rows, din, n_layers, out, backward, fused_dw1=False):
    """Algorithmic FLOPs and HBM bytes of one launch: forward = every Linear once; backward = first-layer weight
    gradient, weight + input gradient of the hidden layers and of the head.  Bytes: the observation row (once forward,
    once more for the weight gradient), the saved activations / statistics, the head's output / its gradient, the
    first-layer gradient round trip through HBM (not with ``fused_dw1``: six-term two-layer trunks with aligned inputs of at
    most 64 columns accumulate that weight gradient inside the chain's launch)."""
    lin = din * 64 + (n_layers - 1) * 64 * 64 + 64 * out
    if not backward:
        return 2.0 * rows * lin, rows * (4 * din + 4 + n_layers * (256 + 8) + 4 * max(out, 64 if out == 0 else out))
    flops = 2.0 * rows * (din * 64 + 2 * (n_layers - 1) * 64 * 64 + 2 * 64 * out)
    return flops, rows * (4 * din + 4 + n_layers * (256 + 8) + 4 * max(out, 64 if out == 0 else out) + (0 if fused_dw1 else 2 * 256))


class RowSource(object):
    """Rows of a 2-D source matrix selected by a sampler minibatch, not yet gathered.

    ``src``   [src_rows, din] float32 device matrix the fused kernels read: a time-major buffer field viewed as rows or,
              for networks with an input LayerNorm, its standardised copy (``standardized`` = True)
    ``idx``   int64 device indices; ``chunk`` = (L, T, N, A) for recurrent_generator's chunk rows, None for row indices
    """

    def __init__(self, src, idx, chunk=None, standardized=False, width=None):
        self.src, self.idx, self.chunk, self.standardized = src, idx, chunk, bool(standardized)
        # columns that carry data: a standardised copy may be padded with zero columns (``standardize_rows``)
        self.width = int(src.shape[1]) if width is None else int(width)
        self.mb = int(idx.shape[0]) if idx is not None else int(src.shape[0])      # idx None: every row of src, in order
        self.rows = self.mb * (chunk[0] if chunk else 1)
        self._tab = None

    @classmethod
    def all_rows(cls, src, standardized=False, width=None):
        """Every row of ``src`` in order (the rollout's [N * A, dim] network inputs): no index list, and the row table is
        a cached identity table -- no launch."""
        rs = cls(src, None, None, standardized, width)
        rs._tab = _identity_table(rs.rows, src.device)
        return rs

    def table(self):
        """int32 row table for the kernels (``mappo_mlp_row_table``): the source row of every launch row; built on first
        use and kept for the other launches on this minibatch (forward, backward)."""
        if self._tab is None:
            lib = _native.lib()
            tab = torch.empty(lib.mappo_mlp_row_table_ints(self.rows), dtype=torch.int32, device=self.src.device)
            L, T, N, A = self.chunk if self.chunk else (0, 0, 0, 0)
            _native.check(lib.mappo_mlp_row_table(self.idx.data_ptr(), self.rows, self.mb, L, T, N, A, tab.data_ptr(),
                                                  _native.stream_of(self.src.device)), "mappo_mlp_row_table")
            self._tab = tab
        return self._tab

    @property
    def shape(self):
        return (self.rows, self.width)

    @property
    def device(self):
        return self.src.device

    is_cuda = True

    def rows_slice(self, lo, hi):
        """Row span [lo, hi) of a rows-mode minibatch / chunk span [lo, hi) of a chunk-mode one (every span keeps all
        L steps of its chunks, row l * (hi - lo) + j)."""
        if self.idx is None:
            return RowSource.all_rows(self.src[lo:hi], self.standardized, self.width)
        return RowSource(self.src, self.idx[lo:hi], self.chunk, self.standardized, self.width)

    def __getitem__(self, key):
        if not (isinstance(key, slice) and key.step in (None, 1)) or self.chunk:
            raise TypeError("RowSource supports contiguous row slices of rows-mode minibatches only")
        lo, hi, _ = key.indices(self.rows)
        return self.rows_slice(lo, hi)

    def source_rows(self):
        """int64 [rows] source row of every minibatch row (shared_buffer.py:379-396 / :554-604)."""
        if self.idx is None:
            return torch.arange(self.rows, device=self.src.device)
        if not self.chunk:
            return self.idx
        L, T, N, A = self.chunk
        r = torch.arange(self.rows, device=self.idx.device)
        l, j = r // self.mb, r % self.mb
        f = self.idx[j] * L + l
        n, rem = f // (A * T), f % (A * T)
        a, t = rem // T, rem % T
        return (t * N + n) * A + a

    def materialize(self):
        """The [rows, din] tensor an eager sampler would have produced from ``src``."""
        return self.src[self.source_rows()][:, :self.width]


_IDENTITY_TABLES = {}


def _identity_table(rows, device):
    """int32 [rows padded to the 128-row tile]: 0 .. rows - 1, then copies of the last row (what ``mappo_mlp_row_table``
    writes for identity rows); cached per (rows, device) -- the rollout asks for the same one every step."""
    key = (int(rows), str(device))
    tab = _IDENTITY_TABLES.get(key)
    if tab is None:
        padded = int(_native.lib().mappo_mlp_row_table_ints(rows))
        tab = torch.arange(padded, dtype=torch.int32, device=device).clamp_(max=rows - 1)
        # never evicted: a captured rollout graph bakes a table's device pointer in, and a table is rows * 4 bytes
        _IDENTITY_TABLES[key] = tab
    return tab


def rollout_rows(x, base):
    """The rollout side of the fused trunk: a plain [rows, dim] float32 device tensor (what ``collect`` / ``compute`` hand
    the networks, reference runner/shared/base_runner.py:120-134, mpe_runner.py:96-109) becomes a ``RowSource`` over all
    its rows -- standardised first when the trunk has an input LayerNorm -- so that trunk (+ head) run as K9 launches
    instead of ~12 framework launches per network and step.  Only without autograd (rollout / evaluation), only in the
    device-sampling mode (the integer-parity mode keeps the PyTorch modules the reference fixtures were pinned with), only
    for trunks the kernels take; otherwise ``x`` comes back unchanged."""
    from . import distributions
    if torch.is_grad_enabled() or not torch.is_tensor(x) or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32 \
            or distributions.SAMPLING_RNG != "device" or os.environ.get("MAPPO_FUSED_ROLLOUT", "1") == "0":
        return x
    if not trunk_supported(base) or x.shape[1] < 4 or x.shape[0] < 1:
        return x
    x = x.contiguous()
    if base._use_feature_normalization:
        w = int(x.shape[1])
        if not ((w % 4 == 0 and w <= 2048) or w <= 1536):       # widths the standardising kernel has shapes for
            return x
        return RowSource.all_rows(standardize_rows(x), standardized=True, width=int(x.shape[1]))
    return RowSource.all_rows(x, standardized=False)


def standardize_rows(src2d, eps=1e-5, pad=True, out=None):
    """(x - mean) / sqrt(var + eps) of every row of a float32 device matrix (``mappo_standardize_rows_ld``): the
    parameter-free half of the networks' input LayerNorm, applied to a whole observation field once per train().
    ``pad``: the copy's rows are padded with zero columns to a multiple of 4 floats (16 bytes), so that the trunk
    kernels take their aligned paths (direct-to-LDS weight gradient, no tail shifting) for odd observation widths;
    ``RowSource(..., width=D)`` remembers the true width."""
    rows, D = src2d.shape
    ld = (D + 3) // 4 * 4 if pad and os.environ.get("MAPPO_PAD_STANDARDIZED", "1") != "0" else D
    if out is None or tuple(out.shape) != (rows, ld) or out.dtype != src2d.dtype or out.device != src2d.device:
        out = torch.empty((rows, ld), dtype=src2d.dtype, device=src2d.device)       # (``out``: storage to write into again)
    _native.check(_native.lib().mappo_standardize_rows_ld(src2d.data_ptr(), rows, D, float(eps), out.data_ptr(), ld,
                                                          _native.stream_of(src2d.device)), "mappo_standardize_rows_ld")
    return out


def _act_kind(module):
    return 1 if isinstance(module, nn.Tanh) else 2 if isinstance(module, nn.ReLU) else None


def trunk_supported(base):
    """Whether an ``MLPBase`` can run through the fused kernels: hidden 64, <= 3 Linear blocks, Tanh / ReLU, float32
    parameters on a HIP device, input LayerNorm (if any) with the sampler's eps."""
    if not enabled() or not hasattr(base, "mlp"):
        return False
    mlp = base.mlp
    blocks = [mlp.fc1] + list(mlp.fc2)
    if len(blocks) > MAX_LAYERS or base.hidden_size != HIDDEN:
        return False
    for blk in blocks:
        lin, act, norm = blk[0], blk[1], blk[2]
        if _act_kind(act) is None or lin.out_features != HIDDEN or not isinstance(norm, nn.LayerNorm):
            return False
        if not norm.elementwise_affine or norm.bias is None or lin.bias is None:
            return False
    eps = {float(blk[2].eps) for blk in blocks}
    if len(eps) != 1:
        return False
    if base._use_feature_normalization and abs(base.feature_norm.eps - 1e-5) > 1e-12:
        return False
    p = lin.weight
    return p.is_cuda and p.dtype == torch.float32


class _FusedTrunkFn(torch.autograd.Function):
    """y = head(trunk(rows)).  Tensor inputs: w1, b1 (input LayerNorm already folded), then per layer ln weight / bias,
    per hidden layer weight / bias, then head weight / bias (absent for out = 0)."""

    @staticmethod
    def forward(ctx, rs, act, eps, n_layers, out, arith, save, *params):
        """``arith``: _native.ARITH_* (the call's matrix arithmetic).  ``save``: the caller's ``torch.is_grad_enabled()`` --
        grad mode is always off in here and ``needs_input_grad`` reflects the parameters' ``requires_grad`` whatever the
        mode, so without it every rollout step (under no_grad, also inside the captured rollout graph) would allocate and
        write the backward's scratch (n_layers * 264 B per row)."""
        lib = _native.lib()
        dev = rs.src.device
        params = [p.detach().contiguous() for p in params]
        m = _native.MLP()
        tab = rs.table()
        _fill_rows(m, rs, tab)
        m.n_layers, m.act, m.out, m.ln_eps, m.arith = n_layers, act, out, float(eps), int(arith)
        _fill_params(m, params, n_layers, out)
        rows = rs.rows
        y = torch.empty((rows, out if out else HIDDEN), dtype=torch.float32, device=dev)
        m.y = y.data_ptr()
        need_grad = bool(save) and any(ctx.needs_input_grad[7:])
        zs = []
        if need_grad:       # what the backward needs of every layer: normalised activations + {mean, rstd} per row
            # (opaque scratch in the kernels' own order, padded to the 128-row tile: include/mappo_hip.h)
            padded = lib.mappo_mlp_row_table_ints(rows)
            zbuf = torch.empty((n_layers, padded, HIDDEN), dtype=torch.float32, device=dev)
            sbuf = torch.empty((n_layers, padded, 2), dtype=torch.float32, device=dev)
            for l in range(n_layers):
                m.z[l] = zbuf[l].data_ptr()
                m.ln_stats[l] = sbuf[l].data_ptr()
            zs = [zbuf, sbuf]
        with _Timed("mappo_mlp_forward", *_work(rows, int(rs.src.shape[1]), n_layers, out, False)):
            _native.check(lib.mappo_mlp_forward(m, _native.stream_of(dev)), "mappo_mlp_forward")
        if need_grad:
            ctx.rs, ctx.cfg = rs, (act, eps, n_layers, out, int(arith))
            ctx.save_for_backward(*(zs + [tab] + params))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _native.lib()
        rs = ctx.rs
        act, eps, n_layers, out, arith = ctx.cfg
        saved = ctx.saved_tensors
        zbuf, sbuf, tab, params = saved[0], saved[1], saved[2], list(saved[3:])
        dev = rs.src.device
        din = int(rs.src.shape[1])
        m = _native.MLP()
        _fill_rows(m, rs, tab)
        m.n_layers, m.act, m.out, m.ln_eps, m.arith = n_layers, act, out, float(eps), arith
        _fill_params(m, params, n_layers, out)
        for l in range(n_layers):
            m.z[l] = zbuf[l].data_ptr()
            m.ln_stats[l] = sbuf[l].data_ptr()
        dy = dy.contiguous()
        grads = torch.empty(lib.mappo_mlp_grad_floats(din, n_layers, out), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.mappo_mlp_workspace_floats(din, n_layers, out), dtype=torch.float32, device=dev)
        dz1 = torch.empty((lib.mappo_mlp_row_table_ints(rs.rows), HIDDEN), dtype=torch.float32, device=dev)   # padded to the tile
        m.dy, m.dz1, m.workspace, m.grads = dy.data_ptr(), dz1.data_ptr(), ws.data_ptr(), grads.data_ptr()
        fused_dw1 = arith == _native.ARITH_SIX_TERM and n_layers == 2 and din % 4 == 0 and din <= 64   # (mappo_mlp_impl.h: DW1)
        with _Timed("mappo_mlp_backward", *_work(rs.rows, din, n_layers, out, True, fused_dw1)):
            _native.check(lib.mappo_mlp_backward(m, _native.stream_of(dev)), "mappo_mlp_backward")
        return (None, None, None, None, None, None, None) + tuple(_split_grads(grads, din, n_layers, out))


def _fill_rows(m, rs, tab):
    m.src = rs.src.data_ptr()
    m.row_tab = tab.data_ptr()
    m.rows, m.din = rs.rows, int(rs.src.shape[1])


def _fill_params(m, params, n_layers, out):
    """params: [w1, b1, (ln_g, ln_b) x n_layers, (w, b) x (n_layers - 1), (wh, bh) if out]."""
    it = iter(params)
    m.w1 = next(it).data_ptr()
    m.bias[0] = next(it).data_ptr()
    for l in range(n_layers):
        m.ln_g[l] = next(it).data_ptr()
        m.ln_b[l] = next(it).data_ptr()
    for l in range(1, n_layers):
        m.w2[l - 1] = next(it).data_ptr()
        m.bias[l] = next(it).data_ptr()
    if out:
        m.wh = next(it).data_ptr()
        m.bh = next(it).data_ptr()


def _split_grads(g, din, n_layers, out):
    """Flat library order [w1 | per layer: bias, ln weight, ln bias | hidden weights | wh | bh] -> the order of
    ``_fill_params``."""
    vec = lambda l, q: g[64 * din + 192 * l + 64 * q: 64 * din + 192 * l + 64 * (q + 1)]
    res = [g[:64 * din].view(64, din), vec(0, 0)]
    for l in range(n_layers):
        res += [vec(l, 1), vec(l, 2)]
    base = 64 * din + 192 * n_layers
    for l in range(1, n_layers):
        res += [g[base + 4096 * (l - 1): base + 4096 * l].view(64, 64), vec(l, 0)]
    if out:
        h0 = base + 4096 * (n_layers - 1)
        res += [g[h0:h0 + 64 * out].view(out, 64), g[h0 + 64 * out:h0 + 65 * out]]
    return res


class _FoldInputNormFn(torch.autograd.Function):
    """(W [64, din], b [64], gamma [din], beta [din]) -> (W * gamma zero-padded to ld columns, b + W beta): the input
    LayerNorm's affine half folded into the first Linear (``mappo_fold_input_norm_forward`` / ``_backward``, one launch per
    direction; as tensor ops it was 3 launches forward and 7 backward per network and update)."""

    @staticmethod
    def forward(ctx, w, b, gamma, beta, ld):
        lib, p = _native.lib(), _native.ptr
        w, b, gamma, beta = (t.detach().contiguous() for t in (w, b, gamma, beta))
        out_f, din = w.shape
        wf = torch.empty((out_f, ld), dtype=torch.float32, device=w.device)
        bf = torch.empty(out_f, dtype=torch.float32, device=w.device)
        _native.check(lib.mappo_fold_input_norm_forward(p(w), p(b), p(gamma), p(beta), out_f, din, ld, p(wf), p(bf),
                                                        _native.stream_of(w.device)), "mappo_fold_input_norm_forward")
        ctx.save_for_backward(w, gamma, beta)
        ctx.ld = ld
        return wf, bf

    @staticmethod
    def backward(ctx, dwf, dbf):
        lib, p = _native.lib(), _native.ptr
        w, gamma, beta = ctx.saved_tensors
        out_f, din = w.shape
        dwf, dbf = dwf.contiguous(), dbf.contiguous()
        dw = torch.empty_like(w)
        dgb = torch.empty((2, din), dtype=torch.float32, device=w.device)
        _native.check(lib.mappo_fold_input_norm_backward(p(w), p(gamma), p(beta), p(dwf), p(dbf), out_f, din, ctx.ld, p(dw),
                                                         p(dgb[0]), p(dgb[1]), _native.stream_of(w.device)),
                      "mappo_fold_input_norm_backward")
        return dw, dbf, dgb[0], dgb[1], None


def _fold_kernel_ok(*tensors):
    return os.environ.get("MAPPO_FUSED_FOLD", "1") != "0" and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def trunk_forward(base, rs, head=None):
    """``head(base(rows))`` for an ``MLPBase`` (see ``trunk_supported``) on the rows of a ``RowSource``; ``head``: an
    ``nn.Linear``-like module with ``weight`` [out, 64] / ``bias`` [out] (out <= 64) or None for the trunk's features."""
    mlp = base.mlp
    blocks = [mlp.fc1] + list(mlp.fc2)
    lin0 = blocks[0][0]
    if base._use_feature_normalization:
        if not rs.standardized:
            raise ValueError("the trunk has an input LayerNorm: the RowSource must read standardised rows")
        fn = base.feature_norm
        if _fold_kernel_ok(lin0.weight, lin0.bias, fn.weight, fn.bias):
            w1, b1 = _FoldInputNormFn.apply(lin0.weight, lin0.bias, fn.weight, fn.bias, int(rs.src.shape[1]))
        else:
            w1 = lin0.weight * fn.weight
            b1 = lin0.bias + lin0.weight @ fn.bias
    else:
        if rs.standardized:
            raise ValueError("the trunk has no input LayerNorm: the RowSource must read the rows as they are")
        w1, b1 = lin0.weight, lin0.bias
    ld = int(rs.src.shape[1])
    if ld != w1.shape[1]:           # zero-padded standardised copy: zero columns in the first Linear, exact zeros in the sums
        w1 = torch.nn.functional.pad(w1, (0, ld - w1.shape[1]))
    params = [w1, b1]
    for blk in blocks:
        params += [blk[2].weight, blk[2].bias]
    for blk in blocks[1:]:
        params += [blk[0].weight, blk[0].bias]
    out = 0
    if head is not None:
        out = int(head.weight.shape[0])
        params += [head.weight, head.bias]
    return _FusedTrunkFn.apply(rs, _act_kind(blocks[0][1]), blocks[0][2].eps, len(blocks), out, matrix_arithmetic_of(base),
                               torch.is_grad_enabled(), *params)


def matrix_arithmetic_of(module):
    """The _native.ARITH_* code a network part (``MLPBase`` / ``RNNLayer``) passes with its K9 / K12 calls: its own
    ``matrix_arithmetic`` attribute (set from ``--matrix_arithmetic`` by the actor / critic that owns it, or later through
    ``set_matrix_arithmetic``), else the process default (MAPPO_MATRIX_ARITHMETIC, else the six-term form)."""
    a = getattr(module, "matrix_arithmetic", None)
    return _native.default_arith() if a is None else _native.arith_code(a)


def set_matrix_arithmetic(root, name):
    """Select the arithmetic of the K9 / K12 matrix products ("six_term" | "f32_mfma") for every ``MLPBase`` / ``RNNLayer``
    below ``root`` -- a per-network choice carried by every call (``arith`` of mappo_mlp_t / mappo_gru_seq_t), nothing
    process-wide."""
    code = _native.arith_code(name)
    for mod in root.modules():
        if hasattr(mod, "matrix_arithmetic"):
            mod.matrix_arithmetic = code
    return code


def head_supported(head):
    return head is not None and getattr(head, "bias", None) is not None and head.weight.dim() == 2 \
        and head.weight.shape[1] == HIDDEN and head.weight.shape[0] <= MAX_OUT
