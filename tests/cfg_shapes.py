"""Shared by tests/test_cfg_shapes_cpu.py and tests/test_gpu_cfg_shapes.py: the reference-generated end-to-end fixtures at the
layer shapes of BASELINE.json configs[3] (SMAC MMM2: obs 370 / share_obs 435 / 18 actions / 10 agents, rmappo, chunk 10, two
minibatches, gain 1 -- reference scripts/train_smac_scripts/train_smac_MMM2.sh:12-14) and configs[4] (Hanabi-Full, 5 players:
obs 1285 / share_obs 1385 / 48 actions, hidden 512, layer_N 2, critic_lr 1e-3 -- scripts/train_hanabi_forward.sh:15-17),
``tests/golden/trainer_cfg_cases.npz`` written by ``oracle/make_golden_trainer.py: CASES_CFG`` from the reference's
R_MAPPOPolicy / R_MAPPO / SharedReplayBuffer.

The inputs are rebuilt from the seed (oracle/synth.py) and checked against the digest stored with the fixture; tensors of
more than 65 536 elements are stored as every 8th element + their float64 sum and sum of squares.
"""
import numpy as np
import torch

from helpers import Box, Discrete, make_args

FIXTURE = "trainer_cfg_cases"
CASES = ["cfg4_shape", "cfg4_shape_dev", "cfg5_shape"]


MID_FIXTURE = "trainer_mid_cases"          # oracle/make_golden_trainer.py: CASES_MID (>= 10^5 rows, north-star flags)
MID_CASES = ["mid_ns", "mid_ns_rnn"]


def build(gold, cname, device=None, fixture=None, **extra):
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    fixture = fixture or FIXTURE
    z = gold.npz(fixture)
    meta = gold.meta(fixture)[cname]
    spec = meta["spec"]
    kw = dict(spec["args"])
    kw.update(extra)
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], **kw)
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    dev = {} if device is None else dict(device=device)
    policy = R_MAPPOPolicy(args, *spaces, **dev)
    trainer = R_MAPPO(args, policy, **dev)
    return z, "trn_%s_" % cname, meta, spec, args, spaces, policy, trainer


def inputs(spec, z, key):
    """The seeded rollout the reference was run on (digest-checked) -> (arrays, next_value)."""
    from oracle import synth
    rnn_hidden = spec["args"]["hidden_size"] if spec["args"].get("use_recurrent_policy") else 0
    arrays = synth.rollout(spec["T"], spec["N"], spec["A"], spec["Do"], spec["Ds"], spec["na"], seed=4242,
                           rnn_hidden=rnn_hidden)
    nv = arrays.pop("next_value")
    np.testing.assert_allclose(synth.digest(arrays, nv), z[key + "input_digest"], rtol=1e-12,
                               err_msg="the seeded inputs differ from the ones the reference was run on")
    return arrays, nv


def start_from_reference_weights(policy, z, key, rtol=1e-4, atol=5e-6):
    """Same seed => the reference's initial weights up to the host LAPACK's last bits (orthogonal_'s QR); the update then
    starts from the fixture's exact values."""
    for net, pre in ((policy.actor, "init_actor."), (policy.critic, "init_critic.")):
        sd = net.state_dict()
        keys = [k[len(key + pre):] for k in z.files if k.startswith(key + pre)]
        assert sorted(keys) == sorted(sd.keys())
        for k, v in sd.items():
            ref = z[key + pre + k]
            np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=rtol, atol=atol, err_msg=pre + k)
            v.copy_(torch.from_numpy(ref))


def stored(z, name, got):
    """(got restricted to what the fixture stores, the stored reference, [sum, sum of squares] or None)."""
    ref = z[name]
    got = np.asarray(got)
    if ref.shape == got.shape:
        return got, ref, None
    stride = 8
    sub = got.ravel()[::stride]
    assert sub.shape == ref.shape, (name, got.shape, ref.shape)
    return sub, ref, z[name + "#moments"]


def check_weights(z, prefix, module, rtol, atol):
    for k, v in module.state_dict().items():
        got = v.detach().cpu().numpy()
        sub, ref, mom = stored(z, prefix + k, got)
        np.testing.assert_allclose(sub, ref, rtol=rtol, atol=atol, err_msg=prefix + k)
        if mom is not None:     # the elements in between, in aggregate: an Adam step moves a weight by <= ~lr
            g64 = got.astype(np.float64)
            n = g64.size
            assert abs(g64.sum() - mom[0]) <= atol * n, (prefix + k, g64.sum(), mom[0])
            np.testing.assert_allclose((g64 * g64).sum(), mom[1], rtol=1e-4, err_msg=prefix + k + " (sum of squares)")


def check_grads(z, prefix, module, rel, worst=None):
    """.grad of every parameter against what the reference's last ppo_update left (after clipping), relative to each tensor's
    largest stored entry."""
    for k, p in module.named_parameters():
        got = p.grad.detach().cpu().numpy()
        sub, ref, mom = stored(z, prefix + k, got)
        scale = max(1e-12, float(np.abs(ref).max()))
        err = float(np.abs(sub - ref).max()) / scale
        if worst is not None:
            worst[prefix + k] = err
        assert err < rel, (prefix + k, err)
        if mom is not None:
            g64 = got.astype(np.float64)
            np.testing.assert_allclose((g64 * g64).sum(), mom[1], rtol=2e-3, err_msg=prefix + k + " (sum of squares)")
