"""TEST INFRASTRUCTURE shared by tests/test_six_term_adversarial_emulated.py (host SIMT emulator) and
tests/test_gpu_six_term_adversarial.py (the MI355X): worst-case inputs for the matrix products of K9 under BOTH arithmetic
forms (include/mappo_hip.h MAPPO_ARITH_SIX_TERM / MAPPO_ARITH_F32_MFMA), judged against float64 with bounds stated relative to
sum |x| |w| -- the quantity a float32 dot product's error is proportional to.

Products are isolated through the C ABI's own outputs:
* first layer:   z1 = W1 x + b1 is rebuilt from what mappo_mlp_forward saves for the backward with an identity activation
                 (z[0] = (z1 - mean) * rstd in fragment order, ln_stats[0] = {mean, rstd}): z1 = z[0] / rstd + mean;
* hidden layer:  z2 = (W2 diag(gamma1)) n1 + (b2 + W2 beta1) likewise from z[1] / ln_stats[1], with n1 = the kernel's own saved z[0]
                 as the float64 reference's input (so only the hidden product is judged);
* first-layer weight gradient: dW1 = dz1^T x with dz1 read back from the backward's scratch argument.

``Backend``: host arrays + the emulator library, or device tensors + libmappo_hip.so.
"""
import ctypes

import numpy as np

import mlp_reference as R

U = 2.0 ** -24          # float32 unit round-off


class HostBackend(object):
    def __init__(self, lib):
        self.lib = lib

    def put(self, a):
        return np.ascontiguousarray(a)

    def ptr(self, h):
        return None if h is None else h.ctypes.data

    def get(self, h):
        return h

    def sync(self):
        pass


class DeviceBackend(object):
    def __init__(self, lib, device):
        import torch
        self.lib, self.dev, self.torch = lib, device, torch

    def put(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def ptr(self, h):
        return None if h is None else h.data_ptr()

    def get(self, h):
        self.torch.cuda.synchronize(self.dev)
        return h.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize(self.dev)


class Net(object):
    """A two-layer hidden-64 trunk with identity activation and a 1-wide head on ``rows`` rows of ``x`` (through an explicit
    row table, so that rows may repeat)."""

    def __init__(self, be, x, srows, params, arith, act=0):
        lib = be.lib
        self.be, self.rows, self.din = be, len(srows), x.shape[1]
        rows, din = self.rows, self.din
        padded = int(lib.mappo_mlp_row_table_ints(rows))
        tab = np.full(padded, srows[-1], np.int32)
        tab[:rows] = srows
        f32 = np.float32
        self.h = dict(x=be.put(x.astype(f32)), tab=be.put(tab), y=be.put(np.full((rows, 1), np.nan, f32)))
        for k, v in params.items():
            self.h[k] = be.put(v.astype(f32))
        for l in range(2):
            self.h["z%d" % l] = be.put(np.full((padded, 64), np.nan, f32))
            self.h["st%d" % l] = be.put(np.full((padded, 2), np.nan, f32))
        p = lambda k: be.ptr(self.h[k])
        m = R.MLP(src=p("x"), row_tab=p("tab"), rows=rows, din=din, n_layers=2, act=act, out=1, ln_eps=1e-5, arith=arith,
                  w1=p("w1"), wh=p("wh"), bh=p("bh"), y=p("y"))
        for l in range(2):
            m.bias[l], m.ln_g[l], m.ln_b[l] = p("bias%d" % l), p("ln_g%d" % l), p("ln_b%d" % l)
            m.z[l], m.ln_stats[l] = p("z%d" % l), p("st%d" % l)
        m.w2[0] = p("w2_0")
        self.m = m

    def forward(self):
        assert self.be.lib.mappo_mlp_forward(ctypes.byref(self.m), None) == 0
        self.be.sync()

    def saved(self, l):
        """-> (n^ [rows, 64], mean [rows], rstd [rows]) as the forward saved them for layer l."""
        z = self.be.get(self.h["z%d" % l])
        st = self.be.get(self.h["st%d" % l])
        return R.rows_of_fragments(z, self.rows), st[:self.rows, 0], st[:self.rows, 1]

    def pre_activation(self, l):
        """z_l of every row rebuilt in float64 from the saved statistics (identity activation)."""
        n, mean, rstd = self.saved(l)
        return n.astype(np.float64) / rstd.astype(np.float64)[:, None] + mean.astype(np.float64)[:, None]

    def y(self):
        return self.be.get(self.h["y"])

    def backward(self, dy):
        lib, be = self.be.lib, self.be
        f32 = np.float32
        din = self.din
        padded = int(lib.mappo_mlp_row_table_ints(self.rows))
        self.h["dy"] = be.put(dy.astype(f32))
        self.h["grads"] = be.put(np.full(int(lib.mappo_mlp_grad_floats(din, 2, 1)), np.nan, f32))
        self.h["ws"] = be.put(np.full(int(lib.mappo_mlp_workspace_floats(din, 2, 1)), np.nan, f32))
        self.h["dz1"] = be.put(np.full((padded, 64), np.nan, f32))
        m = self.m
        m.dy, m.dz1, m.workspace, m.grads = (be.ptr(self.h[k]) for k in ("dy", "dz1", "ws", "grads"))
        assert lib.mappo_mlp_backward(ctypes.byref(m), None) == 0
        be.sync()
        return be.get(self.h["grads"]), be.get(self.h["dz1"])[:self.rows]


def params(rng, din, w1=None, scale=0.3):
    p = R.random_net(rng, din, 2, 1, scale)
    if w1 is not None:
        p["w1"] = w1.astype(np.float32)
    return p


# ------------------------------------------------------------------------------------------------ input generators
def wide_rows(rng, rows, din):
    """every row at its own magnitude, 1e-20 .. 1e16 (beyond that the sum of 64 squares in the LayerNorm behind the product
    overflows float32 -- under either arithmetic form)"""
    x = rng.standard_normal((rows, din)) * 10.0 ** rng.uniform(-20, 16, (rows, 1))
    return x.astype(np.float32), (0.3 * rng.standard_normal((64, din))).astype(np.float32)


def wide_elements(rng, rows, din):
    """every element of x and of W1 at its own magnitude: products spread over 24 decades inside one dot product"""
    x = rng.standard_normal((rows, din)) * 10.0 ** rng.uniform(-6, 6, (rows, din))
    w = rng.standard_normal((64, din)) * 10.0 ** rng.uniform(-6, 6, (64, din))
    return x.astype(np.float32), w.astype(np.float32)


def cancelling(rng, rows, din, delta=1e-6):
    """columns in pairs with x[2j + 1] = x[2j] and w[2j + 1] = -w[2j] (1 + d), |d| <= delta: every dot product cancels to
    <= delta of sum |x| |w|"""
    assert din % 2 == 0
    half = (rng.standard_normal((rows, din // 2)) * 10.0 ** rng.uniform(-3, 3, (rows, 1))).astype(np.float32)
    x = np.repeat(half, 2, axis=1)
    wh = (0.5 * rng.standard_normal((64, din // 2))).astype(np.float32)
    w = np.empty((64, din), np.float32)
    w[:, 0::2] = wh
    w[:, 1::2] = (-wh.astype(np.float64) * (1.0 + delta * rng.uniform(-1, 1, wh.shape))).astype(np.float32)
    return x, w


def subnormal(rng, rows, din):
    """float32 subnormal inputs (|x| < 1.18e-38): their bf16 split keeps 7 bits per part, so the products carry an absolute
    error of up to 2^-133 |w| each (and a matrix pipe may flush them): judged with that allowance"""
    x = (rng.standard_normal((rows, din)) * 3e-39).astype(np.float32)
    assert (np.abs(x[x != 0]) < 1.1755e-38).mean() > 0.95
    return x, (rng.standard_normal((64, din))).astype(np.float32)


GENERATORS = {"wide_rows": wide_rows, "wide_elements": wide_elements, "cancelling": cancelling, "subnormal": subnormal}


def product_bound(S, K, absolute=0.0):
    """What a float32 dot product of K terms may miss: each product rounded once (u = 2^-24), a float32 accumulation chain of K
    (six-term: 6 K) partial sums in an order the hardware chooses -- K u sum|terms| is the classical worst case, ~sqrt(K) u in
    practice -- plus the six-term form's dropped terms (< 2 u per product).  The tests hold BOTH forms to
        |error| <= (16 + K / 6) u sum |x| |w|        (K = 384: 80 u = 4.8e-6 of sum |x||w|; measured: six-term <= 44 u, f32 <= 28 u)
    plus 8 u max_f sum|x||w_f| of the row for the float32 LayerNorm statistics the value is rebuilt from, plus ``absolute``."""
    return (16.0 + K / 6.0) * U * S + 8.0 * U * S.max(axis=1, keepdims=True) + absolute


def first_layer_errors(be, arith, gen, rows, din, seed):
    """-> (max error / bound, max error / (u sum|x||w|)) of z1 = W1 x + b1 over all rows and features."""
    rng = np.random.default_rng(seed)
    x, w1 = GENERATORS[gen](rng, rows, din)
    p = params(rng, din, w1)
    if gen == "subnormal":
        p["bias0"] = np.zeros(64, np.float32)       # so that the subnormal products are the result
    net = Net(be, x, np.arange(rows), p, arith)
    net.forward()
    got = net.pre_activation(0)
    x64, w64, b64 = x.astype(np.float64), p["w1"].astype(np.float64), p["bias0"].astype(np.float64)
    ref = x64 @ w64.T + b64
    S = np.abs(x64) @ np.abs(w64).T + np.abs(b64)
    # subnormal operands: 2^-133 |w| per product from the inexact split, or the whole product where a pipe flushes them
    absolute = (np.abs(x64) @ np.abs(w64).T) * 1.0 + 1e-37 if gen == "subnormal" else 0.0
    assert np.isfinite(got).all(), "non-finite pre-activations from finite inputs"
    err = np.abs(got - ref)
    return float((err / product_bound(S, din, absolute)).max()), float((err / (U * S + 1e-300)).max())


def hidden_layer_errors(be, arith, gen, rows, din, seed):
    """The hidden 64 x 64 product on the kernel's own normalised first-layer activations: gen = "wide_gamma" (the LayerNorm
    weight folded into W2 spans 12 decades) or "cancelling" (first-layer features in identical pairs, hidden weights in
    opposite pairs)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, din)) * 1.5 + 0.7).astype(np.float32)
    p = params(rng, din)
    if gen == "wide_gamma":
        p["ln_g0"] = (rng.choice([-1.0, 1.0], 64) * 10.0 ** rng.uniform(-6, 6, 64)).astype(np.float32)
    else:
        # (duplicated features halve the row's variance structure but keep n^ well defined)
        p["w1"][1::2] = p["w1"][0::2]
        p["bias0"][1::2] = p["bias0"][0::2]
        p["ln_g0"] = np.ones(64, np.float32)
        p["ln_b0"] = np.zeros(64, np.float32)
        wh = p["w2_0"][:, 0::2].copy()
        p["w2_0"][:, 1::2] = (-wh.astype(np.float64) * (1.0 + 1e-6 * rng.uniform(-1, 1, wh.shape))).astype(np.float32)
    net = Net(be, x, np.arange(rows), p, arith)
    net.forward()
    n1 = net.saved(0)[0].astype(np.float64)
    if gen == "cancelling":
        np.testing.assert_array_equal(n1[:, 1::2], n1[:, 0::2])
    w = p["w2_0"].astype(np.float64) * p["ln_g0"].astype(np.float64)[None, :]
    b = p["bias1"].astype(np.float64) + p["w2_0"].astype(np.float64) @ p["ln_b0"].astype(np.float64)
    ref = n1 @ w.T + b
    # the fold W2 * gamma and b + W2 beta are float32 operations of the kernel: one more rounding per term
    S = np.abs(n1) @ np.abs(w).T + np.abs(p["bias1"].astype(np.float64)) + np.abs(p["w2_0"].astype(np.float64)) @ np.abs(p["ln_b0"].astype(np.float64))
    got = net.pre_activation(1)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    return float((err / product_bound(S, 64 + 64)).max()), float((err / (U * S + 1e-300)).max())


def weight_gradient_errors(be, arith, gen, rows, din, seed):
    """dW1 = dz1^T x over ``rows`` rows (the contraction index), dz1 read back from the call: gen = "wide_rows" (dy and x rows at
    their own magnitudes) or "cancelling" (every row twice through the row table, the second copy with dy' = -dy (1 + d))."""
    rng = np.random.default_rng(seed)
    if gen == "wide_rows":
        x = (rng.standard_normal((rows, din)) * 10.0 ** rng.uniform(-8, 8, (rows, 1))).astype(np.float32)
        srows = np.arange(rows)
        dy = rng.standard_normal((rows, 1)) * 10.0 ** rng.uniform(-8, 8, (rows, 1))
    else:
        assert rows % 2 == 0
        x = (rng.standard_normal((rows // 2, din)) * 1.5 + 0.7).astype(np.float32)
        srows = np.concatenate([np.arange(rows // 2), np.arange(rows // 2)])
        half = rng.standard_normal((rows // 2, 1))
        dy = np.concatenate([half, -half * (1.0 + 1e-6 * rng.uniform(-1, 1, half.shape))])
    p = params(rng, din)
    net = Net(be, x, srows, p, arith, act=1)
    net.forward()
    grads, dz1 = net.backward(dy.astype(np.float32))
    got = grads[:64 * din].reshape(64, din).astype(np.float64)
    d64, x64 = dz1.astype(np.float64), x.astype(np.float64)[srows]
    assert np.isfinite(d64).all() and np.isfinite(got).all()
    ref = d64.T @ x64
    S = np.abs(d64).T @ np.abs(x64)
    err = np.abs(got - ref)
    # K = rows terms per sum, accumulated in float32 per wave, then over waves / workgroups in a fixed order
    bound = (16.0 + rows / 6.0) * U * S + 1e-300
    return float((err / bound).max()), float((err / (U * S + 1e-300)).max())


# ------------------------------------------------------------------------------------------------ non-finite operands
BF16_MAX = float(np.float32(3.3895313892515355e38))         # 0x7F7F0000: the largest value whose bf16 rounding is finite


def non_finite_rows(be, arith, din, rows, seed, act=1):
    """Rows 3, 40, 77, 100, 130 of a plain (not standardised) input carry +inf / -inf / NaN / 3.4e38 / the largest bf16 value in
    one column.  Returns (y, y of the clean run, the poisoned row numbers in that order)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, din)) * 1.5).astype(np.float32)
    p = params(rng, din)
    clean = Net(be, x, np.arange(rows), p, arith, act=act)
    clean.forward()
    y0 = clean.y().copy()
    bad = [3, 40, 77, 100, 130]
    vals = [np.inf, -np.inf, np.nan, np.float32(3.4e38), np.float32(BF16_MAX)]
    xb = x.copy()
    for r, v in zip(bad, vals):
        xb[r, (7 * r) % din] = v
    net = Net(be, xb, np.arange(rows), p, arith, act=act)
    net.forward()
    return net, net.y().copy(), y0, bad
