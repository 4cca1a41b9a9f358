"""-m gpu: K13 (mappo_clip_adam): gradient clipping + Adam of one network as two launches, against
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (the calls of the reference's ppo_update, r_mappo.py:146-167;
optimiser of rMAPPOPolicy.py:31-37) on the same tensors over several steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(seed, shapes, dev):
    g = torch.Generator().manual_seed(seed)
    make = lambda: [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    g.manual_seed(seed)
    a = make()
    g.manual_seed(seed)
    b = make()
    return a, b


@pytest.mark.parametrize("max_norm,wd", [(10.0, 0.0), (0.05, 0.0), (None, 0.0), (0.5, 0.01)])
def test_clip_adam_matches_torch(max_norm, wd):
    from onpolicy.algorithms.utils import fused_optim
    dev = torch.device("cuda", 0)
    shapes = [(64, 48), (64,), (64,), (64,), (64, 64), (64,), (5, 64), (5,), (192, 64), (3, 1000, 7)]
    pa, pb = _nets(3, shapes, dev)
    kw = dict(lr=7e-4, eps=1e-5, weight_decay=wd)
    oa = torch.optim.Adam(pa, fused=True, **kw)
    ob = torch.optim.Adam(pb, fused=True, **kw)
    g = torch.Generator().manual_seed(11)
    for step in range(5):
        grads = [torch.randn(s, generator=g).to(dev) * (0.1 if step % 2 else 3.0) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if step == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-4          # lr_decay between updates
        assert fused_optim.supported(oa, pa)
        na = fused_optim.clip_and_step(oa, pa, max_norm)
        if max_norm:
            nb = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        else:
            nb = torch.sqrt(sum(q.grad.norm() ** 2 for q in pb))
        ob.step()
        torch.testing.assert_close(na, nb.reshape(()), rtol=2e-6, atol=0)
        for i, (p, q) in enumerate(zip(pa, pb)):
            torch.testing.assert_close(p.grad, q.grad, rtol=2e-6, atol=1e-9, msg="grad %d step %d" % (i, step))
            torch.testing.assert_close(p.data, q.data, rtol=3e-7, atol=1e-8, msg="param %d step %d" % (i, step))      # ~2 ulp
            for k in ("exp_avg", "exp_avg_sq", "step"):     # (sums that can cancel: absolute floor of a few ulp of the terms)
                torch.testing.assert_close(oa.state[p][k], ob.state[q][k], rtol=2e-6, atol=3e-7, msg=k)
    # the state is torch.optim.Adam's own: a plain step() continues from it
    for p, q in zip(pa, pb):
        p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
    oa.step()
    ob.step()
    for p, q in zip(pa, pb):
        torch.testing.assert_close(p.data, q.data, rtol=5e-7, atol=1e-8)


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_gradient_norm_behaves_like_clip_grad_norm(bad):
    """torch.nn.utils.clip_grad_norm_ (error_if_nonfinite=False): a NaN norm makes the clip coefficient NaN and poisons
    EVERY gradient; an infinite norm gives coefficient 0 (finite entries -> 0, the infinite one -> NaN).  The kernel must
    not quietly map a NaN coefficient to 1 and step on unclipped gradients."""
    from onpolicy.algorithms.utils import fused_optim
    dev = torch.device("cuda", 0)
    shapes = [(64, 48), (64,), (5, 64)]
    pa, pb = _nets(5, shapes, dev)
    oa = torch.optim.Adam(pa, fused=True, lr=7e-4, eps=1e-5)
    g = torch.Generator().manual_seed(12)
    grads = [torch.randn(s, generator=g).to(dev) for s in shapes]
    grads[1][7] = bad
    for p, q, gr in zip(pa, pb, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    na = fused_optim.clip_and_step(oa, pa, 10.0)
    nb = torch.nn.utils.clip_grad_norm_(pb, 10.0)
    torch.testing.assert_close(na, nb.reshape(()), rtol=0, atol=0, equal_nan=True)
    for i, (p, q) in enumerate(zip(pa, pb)):
        torch.testing.assert_close(p.grad, q.grad, rtol=0, atol=0, equal_nan=True, msg="grad %d" % i)
    if bad != bad:
        assert all(torch.isnan(p.grad).all() for p in pa)


def test_unsupported_optimisers_fall_back():
    from onpolicy.algorithms.utils import fused_optim
    dev = torch.device("cuda", 0)
    p = [torch.nn.Parameter(torch.randn(4, 4, device=dev))]
    p[0].grad = torch.randn(4, 4, device=dev)
    assert not fused_optim.supported(torch.optim.Adam(p, amsgrad=True), p)
    assert not fused_optim.supported(torch.optim.SGD(p, lr=0.1), p)
    q = [torch.nn.Parameter(torch.randn(4, 4, device=dev))]
    assert not fused_optim.supported(torch.optim.Adam(q), q)        # no gradient


def test_fused_valuenorm_update_matches_the_tensor_ops():
    """mappo_valuenorm_update (ValueNorm.update + running_mean_var, valuenorm.py:32-55) against the same module on the
    CPU: statistics and the [sigma, mu] pair after several batches, local batches and given (all-reduced) moments; the
    cached pair is dropped as soon as somebody edits the statistics."""
    from onpolicy.utils.valuenorm import ValueNorm
    dev = torch.device("cuda", 0)
    a, b = ValueNorm(1, device=dev), ValueNorm(1)
    g = torch.Generator().manual_seed(2)
    for i in range(6):
        x = torch.randn(100003 if i % 2 else 257, 1, generator=g) * (3.0 + i) + 1.5
        if i == 4:
            mom = (x.mean(0), (x ** 2).mean(0))
            a.update(None, batch_moments=tuple(t.to(dev) for t in mom))
            b.update(None, batch_moments=mom)
        else:
            a.update(x.to(dev))
            b.update(x)
        assert a._denorm_key is not None
        for name in ("running_mean", "running_mean_sq", "debiasing_term"):
            torch.testing.assert_close(getattr(a, name).cpu(), getattr(b, name), rtol=2e-5, atol=1e-10, msg=name)
        torch.testing.assert_close(a.denorm_scalars().cpu(), b.denorm_scalars(), rtol=2e-5, atol=1e-8)
        torch.testing.assert_close(a.normalize(x.to(dev)).cpu(), b.normalize(x), rtol=1e-4, atol=1e-5)
    a.running_mean.fill_(7.0)       # an in-place edit: the cached pair must not be served any more
    b.running_mean.fill_(7.0)
    torch.testing.assert_close(a.denorm_scalars().cpu(), b.denorm_scalars(), rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("policy_masked,value_masked", [(True, True), (True, False), (False, False)])
def test_minibatch_scales_match_the_tensor_ops(policy_masked, value_masked):
    """mappo_minibatch_sums / mappo_minibatch_scales (DataParallel.minibatch_scales: loss denominators of r_mappo.py:135-139,
    :84-87 and the returns' batch moments of :65) against float64 tensor arithmetic, one rank."""
    from onpolicy.utils.dist import DataParallel
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for n in (1, 257, 1_000_003):
        active = (torch.rand(n, 1, generator=g) > 0.3).float()
        active[0] = 1.0
        ret = torch.randn(n, 1, generator=g) * 4.0 + 2.0
        dp = DataParallel(torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), dev)
        out = dp.minibatch_scales(active.to(dev), ret.to(dev), policy_masked, value_masked)
        assert out is not None and out.shape == (8,)
        a, r = active.double(), ret.double()
        den_p = a.sum() if policy_masked else float(n)
        den_v = a.sum() if value_masked else float(n)
        want = torch.tensor([1 / den_p, 1 / den_v, 1 / den_p, 1 / den_p, 1 / den_v, 1 / n, r.mean(), (r * r).mean()])
        torch.testing.assert_close(out.cpu().double(), want, rtol=3e-7, atol=0)
    # columns that do not qualify fall back to the tensor ops
    assert dp.minibatch_scales(active.to(dev).double(), ret.to(dev), True, True) is None


@pytest.mark.parametrize("din,ld", [(48, 48), (30, 32), (435, 436), (7, 8)])
def test_fold_input_norm_kernels_match_autograd(din, ld):
    """mappo_fold_input_norm_forward / _backward (the input LayerNorm's affine half folded into the first Linear,
    mlp.py:47-48 + :20) against the tensor expression they replace, values and all four gradients."""
    from onpolicy.algorithms.utils.fused_mlp import _FoldInputNormFn
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(din)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    w, b, gamma, beta = mk(64, din), mk(64), mk(din), mk(din)
    wf, bf = _FoldInputNormFn.apply(w, b, gamma, beta, ld)
    ref = [t.detach().double().requires_grad_(True) for t in (w, b, gamma, beta)]
    wr = torch.nn.functional.pad(ref[0] * ref[2], (0, ld - din))
    br = ref[1] + ref[0] @ ref[3]
    torch.testing.assert_close(wf.double(), wr.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(bf.double(), br.detach(), rtol=1e-5, atol=1e-5)
    dwf, dbf = torch.randn(64, ld, generator=g).to(dev), torch.randn(64, generator=g).to(dev)
    torch.autograd.backward([wf, bf], [dwf, dbf])
    torch.autograd.backward([wr, br], [dwf.double(), dbf.double()])
    for got, want, name in zip((w, b, gamma, beta), ref, ("w", "b", "gamma", "beta")):
        torch.testing.assert_close(got.grad.double(), want.grad, rtol=2e-5, atol=2e-5, msg=name)
