"""-m gpu: the fused PPO loss kernel (K7, mappo_ppo_loss_f32) against the same loss written with torch ops
and differentiated by autograd (a float32 torch reference of the op: r_mappo.py:52-89, :119-153), for every
flag combination, and the trainer with / without the fused path."""
import itertools

import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args, fill_buffer_arrays, buffer_shapes, load_into

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _torch_loss(logits, avail, actions, old_logp, adv, active, factor, values, value_preds, returns, norm, *,
                clip, huber_delta, entropy_coef, value_loss_coef, use_huber, use_clipped, p_active, v_active):
    """The reference's formulas (FixedCategorical on masked logits; clipped surrogate; clipped huber / mse value
    loss) with torch ops."""
    x = logits if avail is None else torch.where(avail == 0, torch.full_like(logits, -1e10), logits)
    dist = torch.distributions.Categorical(logits=x)
    logp = dist.log_prob(actions.squeeze(-1).long()).unsqueeze(-1)
    ent = dist.entropy()
    ratio = torch.exp(logp - old_logp)
    surr = torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv)
    if factor is not None:
        surr = factor * surr
    per = -surr.sum(-1, keepdim=True)
    if p_active:
        policy_loss = (per * active).sum() / active.sum()
        entropy = (ent * active.squeeze(-1)).sum() / active.sum()
    else:
        policy_loss, entropy = per.mean(), ent.mean()
    target = returns if norm is None else (returns - norm[1]) / norm[0]
    vpc = value_preds + (values - value_preds).clamp(-clip, clip)
    e_c, e_o = target - vpc, target - values

    def loss(e):
        if not use_huber:
            return e ** 2 / 2
        a = (e.abs() <= huber_delta).float()
        return a * e ** 2 / 2 + (1 - a) * huber_delta * (e.abs() - huber_delta / 2)
    vl = torch.max(loss(e_o), loss(e_c)) if use_clipped else loss(e_o)
    value_loss = (vl * active).sum() / active.sum() if v_active else vl.mean()
    return policy_loss, entropy, value_loss, ratio


@pytest.mark.parametrize("na,with_avail,with_factor,with_norm", [(5, False, False, True), (5, True, True, False),
                                                                   (48, True, False, True), (64, False, False, True), (200, True, False, True),
                                                                  (19, True, False, True), (1, False, False, False),
                                                                  (38, False, True, True), (40, True, False, True)])
# (40 / 48 actions + mask: 84 / 100 KB of LDS, granted above 64 KB -- round 4; 200 + mask: wider than the staged variant takes)
def test_fused_loss_matches_autograd(na, with_avail, with_factor, with_norm):
    from onpolicy.algorithms.utils import fused_loss
    R = 10007
    g = torch.Generator(device="cpu").manual_seed(na * 7 + 1)
    rnd = lambda *s: torch.randn(*s, generator=g)
    logits0 = (rnd(R, na) * 2).to(DEV)
    avail = None
    if with_avail:
        avail = (torch.rand(R, na, generator=g) < 0.6).float()
        avail[:, 0] = 1.0
        avail = avail.to(DEV)
    probs = torch.softmax(logits0 if avail is None else torch.where(avail == 0, torch.full_like(logits0, -1e10), logits0), -1)
    actions = torch.multinomial(probs.cpu(), 1, generator=g).float().to(DEV)
    old_logp = (torch.log(probs.gather(1, actions.long())) + rnd(R, 1).to(DEV) * 0.2)
    adv = rnd(R, 1).to(DEV)
    active = (torch.rand(R, 1, generator=g) < 0.8).float().to(DEV)
    factor = (torch.rand(R, 1, generator=g) + 0.5).to(DEV) if with_factor else None
    values0 = rnd(R, 1).to(DEV)
    value_preds = (values0 + rnd(R, 1).to(DEV) * 0.3)
    returns = (rnd(R, 1) * 3 + 1).to(DEV)
    norm = torch.tensor([2.5, 0.7], device=DEV) if with_norm else None
    hp = dict(clip=0.2, huber_delta=0.8, entropy_coef=0.01, value_loss_coef=1.3)
    for use_huber, use_clipped, p_active, v_active in itertools.product([True, False], repeat=4):
        logits = logits0.clone().requires_grad_(True)
        values = values0.clone().requires_grad_(True)
        pl, ent, vl, ratio = _torch_loss(logits, avail, actions, old_logp, adv, active, factor, values, value_preds,
                                         returns, norm, use_huber=use_huber, use_clipped=use_clipped,
                                         p_active=p_active, v_active=v_active, **hp)
        (pl - ent * hp["entropy_coef"]).backward()
        (vl * hp["value_loss_coef"]).backward()
        n = torch.tensor(float(R), device=DEV)
        inv = (1.0 / torch.stack([active.sum() if p_active else n, active.sum() if v_active else n])).contiguous()
        sums = torch.zeros(4, dtype=torch.float64, device=DEV)
        dlogits, dvalues = fused_loss.ppo_loss(
            logits.detach(), avail, actions, old_logp, adv, active, factor, values.detach(), value_preds, returns, norm,
            inv, sums, use_huber=use_huber, use_clipped_value_loss=use_clipped, policy_active_masks=p_active,
            value_active_masks=v_active, **hp)
        tag = (use_huber, use_clipped, p_active, v_active)
        s = sums.cpu().numpy()
        assert s[0] * float(inv[0]) == pytest.approx(float(pl.detach()), rel=2e-5, abs=1e-6), tag
        assert s[1] * float(inv[0]) == pytest.approx(float(ent.detach()), rel=2e-5, abs=1e-6), tag
        assert s[2] * float(inv[1]) == pytest.approx(float(vl.detach()), rel=2e-5, abs=1e-6), tag
        assert s[3] / R == pytest.approx(float(ratio.detach().mean()), rel=2e-5), tag
        scale = float(logits.grad.abs().max())
        np.testing.assert_allclose(dlogits.cpu().numpy(), logits.grad.cpu().numpy(), rtol=2e-4, atol=2e-6 * max(scale, 1e-3) + 1e-10,
                                   err_msg=str(tag))
        np.testing.assert_allclose(dvalues.cpu().numpy(), values.grad.cpu().numpy(), rtol=2e-5, atol=1e-10, err_msg=str(tag))
        if avail is not None:
            assert float(dlogits[avail == 0].abs().sum()) == 0.0


@pytest.mark.parametrize("recurrent", [False, True])
def test_trainer_fused_equals_unfused(monkeypatch, recurrent):
    """One train() with the fused loss and one without, from identical seeds: same logged scalars and
    parameters within float32 tolerance."""
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    T, N, A, Do, Ds, na, H = 20, 16, 3, 6, 18, 5, 16
    results = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MAPPO_FUSED_LOSS", flag)
        args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, ppo_epoch=3, num_mini_batch=2,
                         use_recurrent_policy=recurrent, data_chunk_length=5, sampler_rng="host",
                         algorithm_name="rmappo" if recurrent else "mappo")
        spaces = Box((Do,)), Box((Ds,)), Discrete(na)
        torch.manual_seed(5)
        policy = R_MAPPOPolicy(args, *spaces, device=DEV)
        trainer = R_MAPPO(args, policy, device=DEV)
        assert trainer._fused_loss == (flag == "1")
        buf = SharedReplayBuffer(args, A, *spaces, device=DEV)
        arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, na, H), np.random.default_rng(3), na=na)
        load_into(buf, arrays)
        buf.compute_returns(arrays["next_value"], trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(9)
        info = trainer.train(buf)
        results.append((info, [p.detach().clone() for p in policy.actor.parameters()],
                        [p.detach().clone() for p in policy.critic.parameters()],
                        trainer.value_normalizer.running_mean.clone()))
    (i1, a1, c1, n1), (i0, a0, c0, n0) = results
    for k in i0:
        assert i1[k] == pytest.approx(i0[k], rel=2e-4, abs=2e-6), (k, i1[k], i0[k])
    for p, q in zip(a1 + c1, a0 + c0):
        np.testing.assert_allclose(p.cpu().numpy(), q.cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert torch.allclose(n1, n0, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("na,with_avail", [(5, False), (5, True), (18, True), (48, True), (1, False)])
def test_categorical_sample_kernel_vs_the_framework_rule(na, with_avail):
    """K14 (``mappo_categorical_sample``): masking + one draw per row + its log-probability in one launch, against the
    framework formulas on the SAME Exponential(1) noise (torch.multinomial's rule for one sample: argmax p / q; reference
    distributions.py:14-28, :55-68).  Actions identical, log-probs to float32 rounding; the empirical action frequencies of
    many draws match the probabilities."""
    from onpolicy.algorithms.utils import distributions, fused_loss
    dev = torch.device("cuda", 0)
    distributions.set_sampling_rng("device")
    g = torch.Generator(device=dev).manual_seed(na)
    rows = 4099
    logits = torch.randn(rows, na, device=dev, generator=g) * 2.0
    avail = None
    if with_avail:
        avail = (torch.rand(rows, na, device=dev, generator=g) < 0.7).float()
        avail[:, 0] = 1.0
    with torch.no_grad():
        assert fused_loss.sample_supported(logits)
        torch.manual_seed(123)
        actions, logp = fused_loss.sample_categorical(logits, avail)
        torch.manual_seed(123)
        noise = torch.empty_like(logits).exponential_(1.0)
    x = logits if avail is None else torch.where(avail == 0, torch.full_like(logits, -1e10), logits)
    ref_l = x - x.logsumexp(-1, keepdim=True)
    ref_a = (ref_l.exp() / noise).argmax(-1, keepdim=True)
    assert actions.shape == (rows, 1) and actions.dtype == torch.int64 and logp.shape == (rows, 1)
    same = (actions == ref_a)
    # (a near-tie of p / q can flip under expf's last bit: allow a handful, and require equal probabilities there)
    assert same.float().mean() > 0.999
    torch.testing.assert_close(logp[same.squeeze(-1)], ref_l.gather(-1, ref_a)[same.squeeze(-1)], rtol=1e-5, atol=1e-6)
    if avail is not None:
        assert bool((avail.gather(-1, actions) == 1).all())             # never an unavailable action
    # frequencies: one row's distribution sampled 20 000 times
    if na > 1:
        row = logits[:1].expand(20000, na).contiguous()
        with torch.no_grad():
            a, _ = fused_loss.sample_categorical(row, None)
        freq = torch.bincount(a.reshape(-1), minlength=na).float() / 20000
        torch.testing.assert_close(freq, torch.softmax(logits[0], -1), rtol=0, atol=0.015)
