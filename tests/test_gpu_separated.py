"""-m gpu: the HBM SeparatedReplayBuffer and the separated runner against
fixtures produced by the reference's own SeparatedReplayBuffer (oracle/make_golden_separated.py).
Bit-exact: returns in every flag combination, and the 12- / 13-tuples of the three samplers."""
import os

import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args
from fake_envs import FakeMPEVecEnv
from test_oracle_separated import FIELDS, BUF_FIELDS, CASES, separated_returns_cases

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _vn(n):
    from onpolicy.utils.valuenorm import ValueNorm
    vn = ValueNorm(1, device=DEV)
    vn.running_mean.fill_(float(n[0]))
    vn.running_mean_sq.fill_(float(n[1]))
    vn.debiasing_term.fill_(float(n[2]))
    return vn


def _buffer(args, Do=3, Ds=4, na=5):
    from onpolicy.utils.separated_buffer import SeparatedReplayBuffer
    return SeparatedReplayBuffer(args, Box((Do,)), Box((Ds,)), Discrete(na), device=DEV)


def test_separated_compute_returns_vs_reference(gold):
    for z, m, key in separated_returns_cases(gold):
        args = make_args(episode_length=m["T"], n_rollout_threads=m["N"], use_gae=m["use_gae"],
                         use_popart=m["use_popart"], use_valuenorm=m["use_valuenorm"],
                         use_proper_time_limits=m["use_proper_time_limits"])
        buf = _buffer(args)
        assert tuple(buf.rewards.shape) == (m["T"], m["N"], 1)
        for name in ("rewards", "masks", "bad_masks", "active_masks"):
            getattr(buf, name).copy_(torch.from_numpy(z[key + name]))
        buf.value_preds.copy_(torch.from_numpy(z[key + "value_preds_in"]))
        vn = _vn(z[key + "norm"]) if (key + "norm") in z else None
        buf.compute_returns(z[key + "next_value"], vn)
        np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"], err_msg=str(m))
        # advantages of the trainer prologue (r_mappo.py:179-182) from whatever the scan left behind
        handle = buf.normalized_advantages(vn)
        ret, vp = z[key + "returns"][:-1], buf.value_preds.cpu().numpy()[:-1]
        if vn is not None:
            sigma, mu = [np.float32(x) for x in vn.denorm_scalars().cpu().numpy()]
            vp = vp * sigma + mu
        np.testing.assert_array_equal(handle.raw.cpu().numpy().reshape(ret.shape), ret - vp, err_msg=str(m))


def _gen_buffer(z, recurrent=True):
    sh = z["sgen_buf_share_obs"].shape
    args = make_args(episode_length=sh[0] - 1, n_rollout_threads=sh[1], hidden_size=z["sgen_buf_rnn_states"].shape[-1],
                     use_recurrent_policy=recurrent, sampler_rng="host")
    buf = _buffer(args, Do=z["sgen_buf_obs"].shape[-1], Ds=sh[-1], na=z["sgen_buf_available_actions"].shape[-1])
    for name in BUF_FIELDS:
        getattr(buf, name).copy_(torch.from_numpy(z["sgen_buf_" + name]))
    return buf


@pytest.mark.parametrize("with_factor", [False, True])
@pytest.mark.parametrize("case,call", CASES)
def test_separated_generators_vs_reference(gold, case, call, with_factor):
    z = gold.npz("separated_cases")
    buf = _gen_buffer(z)
    if with_factor:
        buf.update_factor(z["sgen_buf_factor"])
        case += "_factor"
    torch.manual_seed(9)
    batches = list(call(buf, z["sgen_buf_advantages"]))
    n = [m for m in gold.meta("separated_cases")["generators"] if m.get("case") == case][0]["n_batches"]
    assert len(batches) == n
    for bi, sample in enumerate(batches):
        assert len(sample) == (13 if with_factor else 12)
        for fname, t in zip(FIELDS, sample):
            exp = z["sgen_%s_b%d_%s" % (case, bi, fname)]
            assert t.is_cuda and tuple(t.shape) == exp.shape, (fname, tuple(t.shape), exp.shape)
            np.testing.assert_array_equal(t.cpu().numpy(), exp, err_msg="%s b%d %s" % (case, bi, fname))


def test_separated_storage_matches_oracle():
    """insert / after_update through the slab kernel: same contents as the host restatement."""
    from oracle import oracle
    T, N, Do, Ds, na, H = 6, 5, 4, 9, 3, 8
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, use_recurrent_policy=True)
    dev, ref = _buffer(args, Do, Ds, na), oracle.OracleSeparatedBuffer(args, Box((Do,)), Box((Ds,)), Discrete(na))
    rng = np.random.default_rng(5)
    r = lambda *s: rng.standard_normal(s).astype(np.float32)
    for step in range(T + 2):
        if step == T:
            dev.after_update()
            ref.after_update()
        row = (r(N, Ds), r(N, Do), r(N, 1, H), r(N, 1, H), r(N, 1), r(N, 1), r(N, 1), r(N, 1),
               (rng.random((N, 1)) < 0.8).astype(np.float32), (rng.random((N, 1)) < 0.8).astype(np.float32),
               (rng.random((N, 1)) < 0.8).astype(np.float32), (rng.random((N, na)) < 0.7).astype(np.float32))
        dev.insert(*row)
        ref.insert(*row)
    assert dev.step == 2
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
                 "rewards", "masks", "bad_masks", "active_masks", "available_actions"):
        np.testing.assert_array_equal(getattr(dev, name).cpu().numpy(), getattr(ref, name), err_msg=name)


@pytest.mark.parametrize("recurrent", [False, True])
def test_separated_mpe_runner(tmp_path, recurrent):
    from onpolicy.runner.separated.mpe_runner import MPERunner
    T, N, A, Do, na = 8, 4, 3, 6, 5
    args = make_args(env_name="MPE", episode_length=T, n_rollout_threads=N, num_env_steps=2 * T * N,
                     hidden_size=16, ppo_epoch=2, num_mini_batch=2, use_recurrent_policy=recurrent,
                     algorithm_name="rmappo" if recurrent else "mappo", data_chunk_length=4, log_interval=1,
                     use_wandb=False, share_policy=False, n_eval_rollout_threads=2)
    args.scenario_name = "fake_spread"
    envs = FakeMPEVecEnv(N, A, Do, na)
    torch.manual_seed(1)
    runner = MPERunner({"all_args": args, "envs": envs, "eval_envs": FakeMPEVecEnv(2, A, Do, na), "num_agents": A,
                        "device": DEV, "run_dir": tmp_path})
    assert len(runner.policy) == len(runner.trainer) == len(runner.buffer) == A
    runner.warmup()
    for a, b in enumerate(runner.buffer):
        np.testing.assert_array_equal(b.obs[0].cpu().numpy(), envs.log[0]["obs"][:, a])
        np.testing.assert_array_equal(b.share_obs[0].cpu().numpy(), envs.log[0]["obs"].reshape(N, -1))
    for step in range(T):
        out = runner.collect(step)
        obs, rewards, dones, infos = envs.step(out[5])
        runner.insert((obs, rewards, dones, infos) + tuple(out[:5]))
        rec = envs.log[-1]
        for a, b in enumerate(runner.buffer):
            np.testing.assert_array_equal(b.obs[step + 1].cpu().numpy(), rec["obs"][:, a])
            np.testing.assert_array_equal(b.rewards[step].cpu().numpy(), rec["rewards"][:, a])
            np.testing.assert_array_equal(b.actions[step, :, 0].cpu().numpy(), rec["actions"][:, a])
            np.testing.assert_array_equal(b.masks[step + 1, :, 0].cpu().numpy(), 1.0 - rec["dones"][:, a])
            np.testing.assert_array_equal(b.value_preds[step].cpu().numpy(), out[0][a].cpu().numpy())
            if recurrent:
                assert float(b.rnn_states[step + 1][torch.as_tensor(rec["dones"][:, a])].abs().sum()) == 0.0
    runner.compute()
    before = [p.detach().clone() for p in runner.policy[0].actor.parameters()]
    infos = runner.train()
    assert len(infos) == A and all(np.isfinite(v) for info in infos for v in info.values())
    assert any(not torch.equal(p0, p1) for p0, p1 in zip(before, runner.policy[0].actor.parameters()))
    for a, b in enumerate(runner.buffer):
        np.testing.assert_array_equal(b.obs[0].cpu().numpy(), envs.log[-1]["obs"][:, a])      # after_update
        # the factor each agent trained with: product of the ratios of the agents updated before it
        assert tuple(b.factor.shape) == (T, N, 1) and bool(torch.isfinite(b.factor).all())
    assert sum(bool((b.factor == 1).all()) for b in runner.buffer) >= 1     # the first agent in the order
    runner.save()
    for a in range(A):
        for stem in ("actor_agent", "critic_agent", "vnrom_agent"):
            assert os.path.exists(os.path.join(runner.save_dir, "%s%d.pt" % (stem, a)))
    # reload into a fresh runner: same parameters and normaliser statistics
    args2 = make_args(**{k: getattr(args, k) for k in ("env_name", "episode_length", "n_rollout_threads",
                                                       "num_env_steps", "hidden_size", "ppo_epoch", "num_mini_batch",
                                                       "use_recurrent_policy", "algorithm_name", "data_chunk_length",
                                                       "log_interval", "use_wandb", "share_policy")},
                      model_dir=runner.save_dir)
    args2.scenario_name = "fake_spread"
    again = MPERunner({"all_args": args2, "envs": envs, "eval_envs": None, "num_agents": A, "device": DEV,
                       "run_dir": tmp_path / "again"})
    for po, po2, tr, tr2 in zip(runner.policy, again.policy, runner.trainer, again.trainer):
        for p, q in zip(po.actor.parameters(), po2.actor.parameters()):
            assert torch.equal(p, q)
        assert torch.equal(tr.value_normalizer.running_mean, tr2.value_normalizer.running_mean)
    runner.run()
    runner.eval(0)
    lines = open(os.path.join(runner.log_dir, "scalars.jsonl")).read()
    assert "agent0/value_loss" in lines and "agent2/eval_average_episode_rewards" in lines


@pytest.mark.parametrize("cname", ["mlp", "mlp_popart", "mlp_nonorm", "gru"])
def test_happo_on_device_buffer_vs_reference(gold, cname):
    """The reference's HAPPO.train on its SeparatedReplayBuffer (factor set) vs ours through the HBM
    buffer: same CPU seed => same permutations; train_info / parameters within float32 tolerance
    (the network maths runs through rocBLAS)."""
    from test_happo_cpu import build_happo, check_happo_result, BUF
    from onpolicy.utils.separated_buffer import SeparatedReplayBuffer
    z = gold.npz("happo_cases")
    key = "hap_%s_" % cname
    meta, spec, args, spaces, policy, trainer = build_happo(gold, cname, device=DEV)
    args.sampler_rng = "host"
    buf = SeparatedReplayBuffer(args, *spaces, device=DEV)
    for name in BUF:
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"])
    buf.update_factor(z[key + "factor"])
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    check_happo_result(z, key, meta, info, policy, trainer, rel=1e-3, atol=1e-4)


def test_separated_runner_happo(tmp_path):
    from onpolicy.runner.separated.mpe_runner import MPERunner
    from onpolicy.algorithms.happo.happo_trainer import HAPPO
    T, N, A, Do, na = 8, 4, 3, 6, 5
    args = make_args(env_name="MPE", episode_length=T, n_rollout_threads=N, num_env_steps=2 * T * N, hidden_size=16,
                     ppo_epoch=2, num_mini_batch=2, algorithm_name="happo", log_interval=1, use_wandb=False,
                     share_policy=False)
    args.scenario_name = "fake_spread"
    torch.manual_seed(2)
    runner = MPERunner({"all_args": args, "envs": FakeMPEVecEnv(N, A, Do, na), "eval_envs": None, "num_agents": A,
                        "device": DEV, "run_dir": tmp_path})
    assert all(isinstance(tr, HAPPO) for tr in runner.trainer)
    runner.run()
    factors = [b.factor for b in runner.buffer]
    assert sum(bool((f == 1).all()) for f in factors) == 1          # only the first agent of the random order
    assert all(bool(torch.isfinite(f).all()) for f in factors)


@pytest.mark.parametrize("algo", ["happo", "rmappo"])
def test_separated_smac_runner(tmp_path, algo):
    """Per-agent buffers get the team-level mask logic of the SMAC runner (episode end, dead agents,
    time-limit truncations) and the availability masks; HAPPO / recurrent MAPPO train on them."""
    from onpolicy.runner.separated.smac_runner import SMACRunner
    from fake_envs import FakeSMACVecEnv
    T, N, A, Do, Ds, na = 8, 3, 4, 7, 9, 6
    args = make_args(env_name="StarCraft2", episode_length=T, n_rollout_threads=N, num_env_steps=2 * T * N,
                     hidden_size=16, ppo_epoch=1, num_mini_batch=1, use_recurrent_policy=(algo == "rmappo"),
                     algorithm_name=algo, data_chunk_length=4, log_interval=1, use_wandb=False,
                     use_proper_time_limits=True, share_policy=False, use_eval=True, n_eval_rollout_threads=2,
                     eval_episodes=2, use_linear_lr_decay=True)
    args.map_name = "fake"
    envs = FakeSMACVecEnv(N, A, Do, Ds, na)
    torch.manual_seed(1)
    runner = SMACRunner({"all_args": args, "envs": envs, "eval_envs": FakeSMACVecEnv(2, A, Do, Ds, na, seed=3),
                         "num_agents": A, "device": DEV, "run_dir": tmp_path})
    runner.warmup()
    for a, b in enumerate(runner.buffer):
        np.testing.assert_array_equal(b.available_actions[0].cpu().numpy(), envs.log[0]["available_actions"][:, a])
    for step in range(T):
        out = runner.collect(step)
        actions_env = np.stack([x.cpu().numpy() for x in out[1]], axis=1)
        obs, share_obs, rewards, dones, infos, avail = envs.step(actions_env)
        runner.insert((obs, share_obs, rewards, dones, infos, avail) + tuple(out))
        rec = envs.log[-1]
        dones_env = rec["dones"].all(1)
        exp_masks = np.ones((N, A)); exp_masks[dones_env] = 0
        exp_active = np.ones((N, A)); exp_active[rec["dones"]] = 0; exp_active[dones_env] = 1
        exp_bad = np.array([[0.0 if i[a]["bad_transition"] else 1.0 for a in range(A)] for i in rec["infos"]])
        for a, b in enumerate(runner.buffer):
            np.testing.assert_array_equal(b.masks[step + 1, :, 0].cpu().numpy(), exp_masks[:, a])
            np.testing.assert_array_equal(b.active_masks[step + 1, :, 0].cpu().numpy(), exp_active[:, a])
            np.testing.assert_array_equal(b.bad_masks[step + 1, :, 0].cpu().numpy(), exp_bad[:, a])
            np.testing.assert_array_equal(b.share_obs[step + 1].cpu().numpy(), rec["share_obs"][:, a])
            np.testing.assert_array_equal(b.available_actions[step + 1].cpu().numpy(), rec["available_actions"][:, a])
            np.testing.assert_array_equal(b.actions[step, :, 0].cpu().numpy(), rec["actions"][:, a])
    runner.compute()
    infos = runner.train()
    assert len(infos) == A and all(np.isfinite(v) for info in infos for v in info.values())
    runner.run()
    lines = open(os.path.join(runner.log_dir, "scalars.jsonl")).read()
    assert "agent3/dead_ratio" in lines and "eval_win_rate" in lines and "incre_win_rate" in lines
