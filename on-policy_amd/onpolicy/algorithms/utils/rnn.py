"""GRU layer with episode-boundary resets.

Parameters live in an ``nn.GRU`` called ``rnn`` plus a LayerNorm ``norm`` -- the names, shapes and
init order of the reference's onpolicy/algorithms/utils/rnn.py (RNNLayer :7) -- but the recurrence
is evaluated here step by step from those weights:

  * the reference's training branch (rnn.py:30-77) copies the mask to the host to find the time
    steps where some trajectory restarts (rnn.py:43-47) and runs one cuDNN/MIOpen call per
    segment.  Multiplying the state by ``masks[t]`` at EVERY step is the same function (inside a
    segment all masks are 1) and needs no device->host sync, so the update stays asynchronous;
  * the input projection ``x @ W_ih^T + b_ih`` does not depend on the state and is done for all
    L steps in one GEMM; only the [mb, H] x [H, 3H] hidden GEMM and the gates are sequential.

GRU cell (PyTorch convention, gates stacked r|z|n):
    r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
    h' = (1 - z) * n + z * h.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_mlp
from .fused_norm import FusedLayerNorm
from .tall_linear import tall_linear, splitk_weight_grad, column_sums


class _GRUSequenceFn(torch.autograd.Function):
    """One GRU layer over a whole chunk on HIP tensors: gi_all [L, B, 3H] (input projection of all steps, no
    bias), h0 [B, H], masks [L, B, 1] -> out [L, B, H].  Per step: one GEMM for the hidden projection and
    one ``mappo_gru_cell_fwd`` launch (K8) that writes h_t into ``out[t]`` and the masked state of the next
    step in place; the backward mirrors it with ``mappo_gru_cell_bwd`` + one GEMM per step, and forms the
    recurrent weight gradient and both bias gradients ONCE over all L * B rows (split-K) instead of per
    step.  Nothing is stacked, unbound or accumulated by autograd."""

    @staticmethod
    def forward(ctx, gi_all, h0, masks, w_hh, b_ih, b_hh):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        L, B, G = gi_all.shape
        H = G // 3
        gi_all, masks = gi_all.contiguous(), masks.contiguous()
        b_ih, b_hh, w_hh = b_ih.contiguous(), b_hh.contiguous(), w_hh.contiguous()
        dev = gi_all.device
        stream = _native.stream_of(dev)
        hm_all = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        out = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        need_ws = any(ctx.needs_input_grad)
        ws_all = torch.empty(L, B, 4 * H, dtype=torch.float32, device=dev) if need_ws else None
        torch.mul(h0, masks[0], out=hm_all[0])
        w_t = w_hh.t()
        # hidden projection on the MFMA inside the step kernel: measured ahead of (tuned) library GEMM + cell kernel
        # from ~130 k rows per step upwards (1397 vs 1441 ms on ns_rnn at 262 k rows; 132 vs 128 ms on smac at 102 k)
        whole_step = _FUSED_STEP and H == 64 and B >= _FUSED_STEP_MIN_ROWS
        for t in range(L):
            last = t + 1 == L
            nxt = (None, None) if last else (p(masks[t + 1]), p(hm_all[t + 1]))
            wst = p(ws_all[t]) if need_ws else None
            if whole_step:
                _native.check(lib.mappo_gru_step_fwd(p(gi_all[t]), p(hm_all[t]), p(w_hh), p(b_ih), p(b_hh), nxt[0],
                                                     p(out[t]), nxt[1], wst, B, H, stream), "mappo_gru_step_fwd")
            else:
                gh = torch.mm(hm_all[t], w_t)
                _native.check(lib.mappo_gru_cell_fwd(p(gi_all[t]), p(gh), p(hm_all[t]), p(b_ih), p(b_hh), nxt[0],
                                                     p(out[t]), nxt[1], wst, B, H, stream), "mappo_gru_cell_fwd")
        if need_ws:
            ctx.save_for_backward(hm_all, masks, w_hh, ws_all)
        return out

    @staticmethod
    def backward(ctx, dout):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        hm_all, masks, w_hh, ws_all = ctx.saved_tensors
        L, B, H = hm_all.shape
        dev = hm_all.device
        stream = _native.stream_of(dev)
        dout = dout.contiguous()
        dgi_all = torch.empty(L, B, 3 * H, dtype=torch.float32, device=dev)
        dgh_all = torch.empty(L, B, 3 * H, dtype=torch.float32, device=dev)
        dhx = torch.empty(B, H, dtype=torch.float32, device=dev)
        carry = None
        for t in range(L - 1, -1, -1):
            mk = None if t + 1 == L else p(masks[t + 1])
            _native.check(lib.mappo_gru_cell_bwd(p(dout[t]), p(carry), mk, p(ws_all[t]), p(hm_all[t]),
                                                 p(dgi_all[t]), p(dgh_all[t]), p(dhx), B, H, stream),
                          "mappo_gru_cell_bwd")
            carry = torch.addmm(dhx, dgh_all[t], w_hh)            # d loss / d hm[t]
        dh0 = carry * masks[0] if ctx.needs_input_grad[1] else None
        flat_gh = dgh_all.view(L * B, 3 * H)
        dw = splitk_weight_grad(flat_gh, hm_all.view(L * B, H)) if ctx.needs_input_grad[3] else None
        # the r and z thirds of dgh equal those of dgi, so their bias gradients are shared; only the n third
        # (dn * r instead of dn) needs its own column sums
        db_ih = db_hh = None
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            db_ih = column_sums(dgi_all.view(L * B, 3 * H))
            db_hh = torch.cat([db_ih[:2 * H], column_sums(flat_gh[:, 2 * H:])])
        return dgi_all, dh0, None, dw, db_ih, db_hh


class _GRUChunkFn(torch.autograd.Function):
    """RNNLayer (one GRU layer of width 64 + its output LayerNorm) over a whole chunk as ONE launch per direction
    (K12, ``mappo_gru_seq_forward`` / ``_backward``): x [L * B, 64], h0 [B, 64], masks [L * B] -> (y [L * B, 64] =
    LayerNorm(h_l), h_last [B, 64]).  The backward kernel walks the chunk in reverse (truncated BPTT inside the launch)
    and hands back d x, d h0, the LayerNorm gradients, the bias gradients (column sums of the gate gradients, folded
    over the rows inside the launch) and the gate gradients; the two weight gradients are split-K GEMMs over all L * B
    rows here."""

    @staticmethod
    def forward(ctx, x, h0, masks, w_ih, w_hh, b_ih, b_hh, ln_g, ln_b, eps, L, save, arith, head_w=None, head_b=None):
        """``head_w`` [out, 64] / ``head_b`` [out] (out <= 18): an output Linear on y evaluated inside the launches -- the
        first result is then ``y head_w^T + head_b`` instead of y (y itself is kept for the head's weight gradient).
        ``save``: the caller's ``torch.is_grad_enabled()`` -- grad mode is always off in here and ``needs_input_grad``
        reflects the parameters' ``requires_grad`` whatever the mode, so without it every rollout step (L = 1, under
        no_grad) would allocate and write the backward's gates / states / statistics (1280 B per row)."""
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        B = h0.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        x, h0, masks = x.contiguous(), h0.contiguous(), masks.reshape(-1).contiguous()
        params = [t.detach().contiguous() for t in (w_ih, w_hh, b_ih, b_hh, ln_g, ln_b)]
        head = None if head_w is None else (head_w.detach().contiguous(), head_b.detach().contiguous())
        out = 0 if head is None else int(head[0].shape[0])
        need = bool(save) and any(ctx.needs_input_grad)
        # (a narrow head's own gradients come out of the backward launch as sums: the features are never written then)
        y = torch.empty(L * B, 64, **f32) if (head is None or (need and out > CHUNK_HEAD_SUMS)) else None
        logits = torch.empty(L * B, out, **f32) if head is not None else None
        h_last = torch.empty(B, 64, **f32)
        gates = torch.empty(lib.mappo_gru_seq_gates_floats(L, B), **f32) if need else None
        stats = torch.empty(lib.mappo_gru_seq_stats_floats(L, B), **f32) if need else None
        hm = torch.empty(L * B, 64, **f32) if need else None
        m = _native.GRUSeq(x=p(x), h0=p(h0), masks=p(masks), w_ih=p(params[0]), w_hh=p(params[1]), b_ih=p(params[2]),
                           b_hh=p(params[3]), ln_g=p(params[4]), ln_b=p(params[5]), ln_eps=float(eps), H=64, L=L, mb=B,
                           arith=int(arith), y=p(y), h_last=p(h_last), gates=p(gates), hm=p(hm), stats=p(stats),
                           head_w=p(head[0]) if head else None, head_b=p(head[1]) if head else None, head_out=out,
                           logits=p(logits))
        _native.check(lib.mappo_gru_seq_forward(m, _native.stream_of(dev)), "mappo_gru_seq_forward")
        if need:
            ctx.save_for_backward(x, h0, masks, gates, stats, hm, *params, *(head + ((y,) if y is not None else ()) if head else ()))
            ctx.cfg = (float(eps), L, out, int(arith))
        return (logits if head is not None else y), h_last

    @staticmethod
    def backward(ctx, dy, dh_last):
        from onpolicy import _native
        lib, p = _native.lib(), _native.ptr
        x, h0, masks, gates, stats, hm, w_ih, w_hh, b_ih, b_hh, ln_g, ln_b = ctx.saved_tensors[:12]
        eps, L, out, arith = ctx.cfg
        head_w, head_b = ctx.saved_tensors[12:14] if out else (None, None)
        y = ctx.saved_tensors[14] if out > CHUNK_HEAD_SUMS else None
        B = h0.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dy = torch.zeros(L * B, out or 64, **f32) if dy is None else dy.contiguous()
        dh_last = None if dh_last is None else dh_last.contiguous()
        dx = torch.empty(L * B, 64, **f32)
        dgi = torch.empty(L * B, 192, **f32)
        dq = torch.empty(L * B, 64, **f32)
        dh0 = torch.empty(B, 64, **f32) if ctx.needs_input_grad[1] else None
        # LayerNorm weight | bias gradients, column sums of dgi [192] and dq [64], a narrow head's sums [6, 64] + [6]
        ln_grads = torch.empty(800, **f32)
        ws = torch.empty(lib.mappo_gru_seq_workspace_floats(), **f32)
        m = _native.GRUSeq(x=p(x), h0=p(h0), masks=p(masks), w_ih=p(w_ih), w_hh=p(w_hh), b_ih=p(b_ih), b_hh=p(b_hh),
                           ln_g=p(ln_g), ln_b=p(ln_b), ln_eps=eps, H=64, L=L, mb=B, arith=arith, gates=p(gates), hm=p(hm),
                           stats=p(stats), dy=None if out else p(dy), dx=p(dx), dgi=p(dgi), dq=p(dq), dh0=p(dh0),
                           dh_last=p(dh_last), ln_grads=p(ln_grads), workspace=p(ws),
                           head_w=p(head_w), head_b=p(head_b), head_out=out, dlogits=p(dy) if out else None,
                           head_sums=int(0 < out <= CHUNK_HEAD_SUMS))
        _native.check(lib.mappo_gru_seq_backward(m, _native.stream_of(dev)), "mappo_gru_seq_backward")
        # dW_ih = dgi^T x; the hidden side's gate gradient is [dgi_r | dgi_z | dq]
        if arith == _native.ARITH_SIX_TERM and _WEIGHT_GRAD_KERNEL:
            # one launch that reads the four matrices once, in the arithmetic of the rest of the call (mappo_gru_impl.h)
            dw = torch.empty(2, 192, 64, **f32)
            ws2 = torch.empty(lib.mappo_gru_weight_grads_workspace_floats(), **f32)
            _native.check(lib.mappo_gru_weight_grads(p(dgi), p(dq), p(x), p(hm), L * B, p(dw), p(ws2), _native.stream_of(dev)),
                          "mappo_gru_weight_grads")
            dw_ih, dw_hh = dw[0], dw[1]
        else:       # float32 route: three split-K library GEMMs
            dw_ih = splitk_weight_grad(dgi, x)
            dw_hh = torch.cat([splitk_weight_grad(dgi[:, :128], hm), splitk_weight_grad(dq, hm)], 0)
        db_ih = ln_grads[128:320]
        db_hh = torch.cat([ln_grads[128:256], ln_grads[320:384]])
        if not out:
            d_head = (None, None)
        elif out <= CHUNK_HEAD_SUMS:        # dW_h = gamma (.) sum dlogits n^ + beta (x) sum dlogits (y = n^ gamma + beta)
            dbh = ln_grads[768:768 + out]
            d_head = (ln_grads[384:384 + 64 * out].view(out, 64) * ln_g + dbh[:, None] * ln_b, dbh)
        else:
            d_head = (splitk_weight_grad(dy, y), column_sums(dy))
        return (dx, dh0, None, dw_ih, dw_hh, db_ih, db_hh, ln_grads[:64], ln_grads[64:128], None, None, None, None) + d_head


# the widest output Linear the chunk kernels evaluate themselves (k steps of 2 on the MFMA in the backward)
CHUNK_HEAD_MAX = 18
# ... and the widest whose weight / bias gradients it also sums itself (one column-sum butterfly per output and step)
CHUNK_HEAD_SUMS = 6 if __import__("os").environ.get("MAPPO_GRU_HEAD_SUMS", "1") != "0" else 0

# MAPPO_GRU_CHUNK=0 keeps the step-by-step kernels below for the update (one launch per step and direction)
_CHUNK_KERNEL = __import__("os").environ.get("MAPPO_GRU_CHUNK", "1") != "0"
# the chunk kernels' weight gradients under the six-term arithmetic: mappo_gru_weight_grads (0: the library GEMMs of the float32 route)
_WEIGHT_GRAD_KERNEL = __import__("os").environ.get("MAPPO_GRU_WEIGHT_GRAD_KERNEL", "1") != "0"
# MAPPO_GRU_HEAD=0 keeps the output Linear behind the GRU (action head / v_out) a separate GEMM
_CHUNK_HEAD = __import__("os").environ.get("MAPPO_GRU_HEAD", "1") != "0"
# MAPPO_GRU_SEQUENCE=0 falls back to aten::_thnn_fused_gru_cell driven step by step through autograd
_SEQUENCE_KERNELS = __import__("os").environ.get("MAPPO_GRU_SEQUENCE", "1") != "0"
# MAPPO_GRU_FUSED_STEP=0 keeps the forward hidden projection a library GEMM next to the K8 cell kernel
# (default for H = 64: one kernel per step with the projection on the f32 MFMA)
_FUSED_STEP = __import__("os").environ.get("MAPPO_GRU_FUSED_STEP", "1") != "0"
_FUSED_STEP_MIN_ROWS = 1 << 17


class RNNLayer(nn.Module):
    # arithmetic of K12's matrix products (_native.ARITH_* or None = the process default; fused_mlp.set_matrix_arithmetic)
    matrix_arithmetic = None

    def __init__(self, inputs_dim, outputs_dim, recurrent_N, use_orthogonal):
        super(RNNLayer, self).__init__()
        self._recurrent_N = recurrent_N
        self._use_orthogonal = use_orthogonal
        self.rnn = nn.GRU(inputs_dim, outputs_dim, num_layers=self._recurrent_N)
        for name, param in self.rnn.named_parameters():
            if 'bias' in name:
                nn.init.constant_(param, 0)
            elif 'weight' in name:
                if self._use_orthogonal:
                    nn.init.orthogonal_(param)
                else:
                    nn.init.xavier_uniform_(param)
        self.norm = FusedLayerNorm(outputs_dim)

    def _layer_weights(self, k):
        g = self.rnn
        return (getattr(g, 'weight_ih_l%d' % k), getattr(g, 'weight_hh_l%d' % k),
                getattr(g, 'bias_ih_l%d' % k), getattr(g, 'bias_hh_l%d' % k))

    @staticmethod
    def _cell(gi, h, w_hh, b_hh):
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, -1)
        h_r, h_z, h_n = gh.chunk(3, -1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        return (1.0 - z) * n + z * h

    def _run(self, x, h0, masks):
        """x [L, B, in], h0 [B, R, H], masks [L, B, 1] -> (y [L, B, H], h_last [B, R, H]).

        On HIP tensors a layer runs through ``_GRUSequenceFn`` (K8: in-place cell kernels, one hidden GEMM per
        step or, from 131 k rows per step, the projection inside the step kernel; weight / bias gradients once
        over all L * B rows).  ``MAPPO_GRU_SEQUENCE=0`` selects the older device path instead -- aten's fused GRU
        cell driven step by step through autograd, time steps taken apart with ``unbind`` / ``stack``.  CPU
        tensors use the explicit cell above (same formulas)."""
        L, B = x.size(0), x.size(1)
        fused = x.is_cuda
        mask_steps = masks.unbind(0)
        layer_in = x
        finals = []
        for k in range(self._recurrent_N):
            w_ih, w_hh, b_ih, b_hh = self._layer_weights(k)
            if fused and _SEQUENCE_KERNELS:     # K8: whole chunk through the in-place cell kernels
                gi_all = tall_linear(layer_in.reshape(L * B, -1), w_ih, None).view(L, B, -1)
                layer_in = _GRUSequenceFn.apply(gi_all, h0[:, k], masks, w_hh, b_ih, b_hh)
                finals.append(layer_in[-1])
                continue
            if fused:       # biases are added inside the fused cell
                gi_all = tall_linear(layer_in.reshape(L * B, -1), w_ih, None).view(L, B, -1)
            else:
                gi_all = F.linear(layer_in, w_ih, b_ih)          # one GEMM for all L steps
            gi_steps = gi_all.unbind(0)
            h = h0[:, k]
            outs = []
            for t in range(L):
                hm = h * mask_steps[t]
                if fused:
                    gh = tall_linear(hm, w_hh, None)
                    h = torch.ops.aten._thnn_fused_gru_cell(gi_steps[t], gh, hm, b_ih, b_hh)[0]
                else:
                    h = self._cell(gi_steps[t], hm, w_hh, b_hh)
                outs.append(h)
            layer_in = torch.stack(outs, 0)
            finals.append(h)
        return layer_in, torch.stack(finals, 1)

    def _chunk_kernel_ok(self, x):
        """One GRU layer of width 64 on float32 HIP tensors, LayerNorm with affine parameters: K12 takes the whole layer."""
        return _CHUNK_KERNEL and x.is_cuda and x.dtype == torch.float32 and self._recurrent_N == 1 and x.size(-1) == 64 \
            and self.rnn.hidden_size == 64 and self.norm.elementwise_affine and self.norm.bias is not None

    def head_ok(self, x, head):
        """Whether ``forward(..., head=head)`` evaluates the output Linear inside the chunk kernels: K12 takes the layer,
        ``head`` is Linear-like (weight [out <= 18, 64], bias [out]) and the ``MAPPO_GRU_HEAD`` switch is on."""
        w, b = getattr(head, "weight", None), getattr(head, "bias", None)
        return _CHUNK_HEAD and self._chunk_kernel_ok(x) and torch.is_tensor(w) and torch.is_tensor(b) and w.dim() == 2 \
            and w.shape[1] == 64 and 0 < w.shape[0] <= CHUNK_HEAD_MAX and w.is_cuda and w.dtype == torch.float32

    def forward(self, x, hxs, masks, head=None):
        """-> (LayerNorm(h_l) rows, final states); with ``head`` (see ``head_ok``) the first result is ``head(...)`` of
        those rows, evaluated inside the K12 launches."""
        if self._chunk_kernel_ok(x):
            B = hxs.size(0)
            L = x.size(0) // B
            w_ih, w_hh, b_ih, b_hh = self._layer_weights(0)
            extra = ()
            if head is not None:
                if not self.head_ok(x, head):
                    raise ValueError("this head cannot be evaluated inside the GRU chunk kernels (see head_ok)")
                extra = (head.weight, head.bias)
            y, h_last = _GRUChunkFn.apply(x, hxs[:, 0], masks, w_ih, w_hh, b_ih, b_hh, self.norm.weight, self.norm.bias,
                                          self.norm.eps, L, torch.is_grad_enabled(),
                                          fused_mlp.matrix_arithmetic_of(self), *extra)
            return y, h_last.unsqueeze(1)
        if head is not None:
            raise ValueError("this head cannot be evaluated inside the GRU chunk kernels (see head_ok)")
        if x.size(0) == hxs.size(0):
            # rollout: one step for every (env, agent) row
            y, hxs = self._run(x.unsqueeze(0), hxs, masks.reshape(1, -1, 1))
            x = y.squeeze(0)
        else:
            # update: x is [L*B, .] (time-major chunks from the recurrent sampler), hxs [B, R, H]
            B = hxs.size(0)
            L = x.size(0) // B
            y, hxs = self._run(x.view(L, B, x.size(1)), hxs, masks.view(L, B, 1))
            x = y.reshape(L * B, -1)
        return self.norm(x), hxs
