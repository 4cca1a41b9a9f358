"""TEST INFRASTRUCTURE shared by tests/test_runners_cpu.py (host stand-in buffer, CPU) and tests/test_gpu_runners.py
(the HBM buffer, policy on the GPU): drive this package's shared runners exactly like oracle/make_golden_runners.py
drove the REFERENCE's runners on the same deterministic fake envs and seeds, and compare buffers, parameters, training
info and eval logs with what the reference produced (tests/golden/runner_cases.npz)."""
import json
import os

import numpy as np
import pytest
import torch

import fake_envs
from helpers import make_args

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
          "rewards", "masks", "bad_masks", "active_masks", "returns", "available_actions")
TOL = dict(rtol=2e-4, atol=2e-5)


def to_np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def config(args, envs, A, tmp_path, eval_envs=None, device=torch.device("cpu")):
    return {"all_args": args, "envs": envs, "eval_envs": eval_envs, "num_agents": A, "device": device,
            "run_dir": tmp_path}


def check_eval(runner, step, expected):
    """runner.eval(step) must log what the reference runner logged (tag, value) for the same policy and eval envs."""
    path = os.path.join(runner.log_dir, "scalars.jsonl")
    before = len(open(path).read().splitlines()) if os.path.exists(path) else 0
    runner.eval(step)
    got = [json.loads(l) for l in open(path).read().splitlines()[before:]]
    assert [g["tag"] for g in got] == [e[0] for e in expected]
    for g, (tag, vals, at) in zip(got, expected):
        assert g["step"] == at
        for k, v in vals.items():
            assert g[k] == pytest.approx(v, rel=2e-4, abs=2e-6), (tag, g[k], v)


def check_params(z, prefix, policy, exact=False, rtol=1e-4, atol=3e-5):
    nets = (("transformer.", policy.transformer),) if hasattr(policy, "transformer") else \
        (("actor.", policy.actor), ("critic.", policy.critic))
    for net, mod in nets:
        for k, v in mod.state_dict().items():
            ref = z[prefix + net + k]
            if exact:
                np.testing.assert_array_equal(to_np(v), ref, err_msg=prefix + net + k)
            else:
                np.testing.assert_allclose(to_np(v), ref, rtol=rtol, atol=atol, err_msg=prefix + net + k)


def check_buffer(z, prefix, buf):
    for name in FIELDS:
        got = to_np(getattr(buf, name))
        ref = z[prefix + name]
        if name == "actions":
            np.testing.assert_array_equal(got, ref, err_msg=name)          # integer sampling parity
        else:
            np.testing.assert_allclose(got, ref, err_msg=name, **TOL)


def replay_shared_case(gold, tmp_path, cname, device=torch.device("cpu"), init_exact=True, **over):
    """MPE / SMAC shared runner: warmup, T x (collect, env step, insert), compute, train, eval."""
    from onpolicy.runner.shared.mpe_runner import MPERunner
    from onpolicy.runner.shared.smac_runner import SMACRunner
    z, meta = gold.npz("runner_cases"), gold.meta("runner_cases")[cname]
    sp = meta["spec"]
    T, N, A = sp["T"], sp["N"], sp["A"]
    smac = sp["env"] == "StarCraft2"
    kw = dict(sp["args"])
    kw.update(over)
    args = make_args(env_name=sp["env"], episode_length=T, n_rollout_threads=N, num_env_steps=T * N, use_wandb=False,
                     use_eval=True, n_eval_rollout_threads=2, eval_episodes=4, **kw)
    args.scenario_name = args.map_name = "fake"
    envs = fake_envs.FakeSMACVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"]) if smac \
        else fake_envs.FakeMPEVecEnv(N, A, sp["Do"], sp["na"])
    eval_envs = fake_envs.FakeSMACVecEnv(2, A, sp["Do"], sp["Ds"], sp["na"], seed=3) if smac \
        else fake_envs.FakeMPEVecEnv(2, A, sp["Do"], sp["na"], seed=3)
    torch.manual_seed(1)
    np.random.seed(1)
    runner = (SMACRunner if smac else MPERunner)(config(args, envs, A, tmp_path, eval_envs, device))
    key = "run_%s_" % cname
    # initial weights: drawn on the CPU generator on every device; bit-identical on the host that made the fixtures,
    # 1e-6 across hosts (orthogonal_'s LAPACK QR)
    wide = kw.get("hidden_size", 0) >= 64 and hasattr(runner.policy, "actor")     # (64-wide QR factors differ a little more between hosts)
    check_params(z, key + "init_", runner.policy, exact=init_exact and not wide, rtol=1e-4 if wide else 1e-5,
                 atol=5e-6 if wide else 1e-6)
    if wide:        # ... so a wide case starts from the reference's exact weights: its sampled actions must not depend on them
        with torch.no_grad():
            for net, mod in (("actor.", runner.policy.actor), ("critic.", runner.policy.critic)):
                for k, v in mod.state_dict().items():
                    v.copy_(torch.from_numpy(z[key + "init_" + net + k]))
    torch.manual_seed(5)
    runner.warmup()
    for step in range(T):
        res = runner.collect(step)
        if smac:
            values, actions, action_log_probs, rnn_states, rnn_states_critic = res
            obs, share_obs, rewards, dones, infos, avail = envs.step(to_np(actions))
            runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, action_log_probs, rnn_states,
                           rnn_states_critic))
        else:
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = res
            obs, rewards, dones, infos = envs.step(actions_env)
            runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))
    runner.compute()
    check_buffer(z, key + "rollout_", runner.buffer)
    torch.manual_seed(9)
    info = runner.train()
    for k, v in meta["train_info"].items():
        assert info[k] == pytest.approx(v, rel=5e-4, abs=5e-6), (k, info[k], v)
    check_buffer(z, key + "after_", runner.buffer)
    check_params(z, key + "final_", runner.policy)
    check_eval(runner, 777, meta["eval_logged"])
    return runner


def replay_hanabi_case(gold, tmp_path, device=torch.device("cpu"), init_exact=True, **over):
    """The whole turn-based Hanabi loop (4 episodes) on the fake choose-env."""
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner
    z, meta = gold.npz("runner_cases"), gold.meta("runner_cases")["hanabi"]
    sp = meta["spec"]
    T, N, A = sp["T"], sp["N"], sp["A"]
    args = make_args(env_name="Hanabi", episode_length=T, n_rollout_threads=N, num_env_steps=4 * T * N, hidden_size=16,
                     ppo_epoch=2, num_mini_batch=1, algorithm_name="mappo", use_linear_lr_decay=True, log_interval=1000,
                     save_interval=1000, use_wandb=False, **over)
    args.hanabi_name = "fake"
    args.n_eval_rollout_threads = 3
    envs = fake_envs.FakeChooseVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    runner = HanabiRunner(config(args, envs, A, tmp_path,
                                 fake_envs.FakeChooseVecEnv(3, A, sp["Do"], sp["Ds"], sp["na"], seed=4), device))
    check_params(z, "run_hanabi_init_", runner.policy, exact=init_exact, rtol=1e-5, atol=1e-6)
    torch.manual_seed(5)
    runner.run()
    assert runner.true_total_num_steps == meta["true_total_num_steps"]
    assert envs.steps == meta["env_steps"] and envs.games == meta["games"]
    check_buffer(z, "run_hanabi_after_", runner.buffer)
    check_params(z, "run_hanabi_final_", runner.policy)
    check_eval(runner, 888, meta["eval_logged"])
    return runner
