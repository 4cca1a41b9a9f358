#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/runner_cases.npz by driving the REFERENCE's runners
(onpolicy/runner/shared/{mpe,smac,hanabi_runner_forward}.py with the reference buffer, policy and trainer, on the
CPU) against the deterministic fake envs of tests/fake_envs.py: buffer contents after a rollout + compute, the
train_info of one update, the parameters afterwards; for Hanabi the state after a few episodes of ``run()``.
wandb / tensorboardX / imageio are replaced by empty stand-ins for the import only.

    python oracle/make_golden_runners.py
"""
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402  (loads the reference)

for name in ("wandb", "imageio", "tensorboardX"):
    sys.modules.setdefault(name, types.ModuleType(name))


LOGGED = []      # (main_tag, {key: value}, step) of every add_scalars call of the reference runners


class _Writer(object):
    def __init__(self, *a, **k):
        pass

    def add_scalars(self, main_tag, tag_scalar_dict, global_step=None):
        LOGGED.append((main_tag, {k: float(v) for k, v in tag_scalar_dict.items()}, global_step))


sys.modules["tensorboardX"].SummaryWriter = _Writer

import fake_envs  # noqa: E402  (tests/fake_envs.py)

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
          "rewards", "masks", "bad_masks", "active_masks", "returns")


def dump(out, key, runner, with_avail):
    for name in FIELDS + (("available_actions",) if with_avail else ()):
        out[key + name] = np.array(getattr(runner.buffer, name), dtype=np.float32)


def params(out, key, policy):
    if hasattr(policy, "transformer"):          # MAT: one network
        for k, v in policy.transformer.state_dict().items():
            out[key + "transformer." + k] = v.detach().numpy().copy()
        return
    for k, v in policy.actor.state_dict().items():
        out[key + "actor." + k] = v.detach().numpy().copy()
    for k, v in policy.critic.state_dict().items():
        out[key + "critic." + k] = v.detach().numpy().copy()


def config(args, envs, A, run_dir, eval_envs=None):
    return {"all_args": args, "envs": envs, "eval_envs": eval_envs, "num_agents": A, "device": torch.device("cpu"),
            "run_dir": Path(run_dir)}


def logged_since(mark):
    return [[tag, vals, step] for tag, vals, step in LOGGED[mark:]]


def main():
    out, meta = {}, {}
    tmp = tempfile.mkdtemp()
    from onpolicy.runner.shared.mpe_runner import MPERunner
    from onpolicy.runner.shared.smac_runner import SMACRunner
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner

    # ---- MPE (feed-forward and recurrent) and SMAC: one rollout, compute, one update
    specs = {
        "mpe_mlp": dict(env="MPE", runner=MPERunner, T=8, N=4, A=3, Do=6, na=5,
                        args=dict(algorithm_name="mappo", hidden_size=16, ppo_epoch=2, num_mini_batch=2)),
        "mpe_rnn": dict(env="MPE", runner=MPERunner, T=8, N=4, A=3, Do=6, na=5,
                        args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=16, ppo_epoch=2,
                                  num_mini_batch=2, data_chunk_length=4)),
        # (no football case: the reference's FootballRunner.insert passes rnn_states= to a buffer method whose
        #  parameter is rnn_states_actor, football_runner.py:132-142 vs shared_buffer.py:90 -- it raises TypeError)
        "smac_rnn": dict(env="StarCraft2", runner=SMACRunner, T=8, N=3, A=4, Do=7, Ds=9, na=6,
                         args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=16, ppo_epoch=1,
                                   num_mini_batch=1, data_chunk_length=4, use_proper_time_limits=True)),
        # Multi-Agent Transformer through the same SMAC runner (base_runner.py:66-93, :124-128; smac_runner.py:176-182)
        "smac_mat": dict(env="StarCraft2", runner=SMACRunner, T=8, N=3, A=4, Do=7, Ds=9, na=6,
                         args=dict(algorithm_name="mat", n_embd=16, n_head=2, n_block=1, ppo_epoch=2, num_mini_batch=2)),
        "smac_mat_dec": dict(env="StarCraft2", runner=SMACRunner, T=8, N=3, A=4, Do=7, Ds=9, na=6,
                             args=dict(algorithm_name="mat_dec", dec_actor=True, share_actor=True, n_embd=16, n_head=1,
                                       n_block=1, ppo_epoch=1, num_mini_batch=1)),
    }
    # hidden 64 (the width of every shipped MPE / SMAC configuration: on the device these go through the fused trunk, K9,
    # and -- recurrent -- the GRU chunk kernels, K12); generated after everything else so that the older cases keep their
    # place in the run
    h64_specs = {
        "mpe_mlp_h64": dict(env="MPE", runner=MPERunner, T=12, N=6, A=3, Do=18, na=5,
                            args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2,
                                      num_mini_batch=1)),
        "smac_rnn_h64": dict(env="StarCraft2", runner=SMACRunner, T=10, N=5, A=4, Do=22, Ds=30, na=6,
                             args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=64, layer_N=1,
                                       ppo_epoch=1, num_mini_batch=1, data_chunk_length=5, gain=1.0)),
    }

    def shared_case(cname, sp):
        T, N, A = sp["T"], sp["N"], sp["A"]
        args = mg.make_args(env_name=sp["env"], episode_length=T, n_rollout_threads=N, num_env_steps=T * N,
                            use_wandb=False, use_eval=True, n_eval_rollout_threads=2, eval_episodes=4, **sp["args"])
        args.scenario_name = args.map_name = "fake"
        smac = sp["env"] == "StarCraft2"
        envs = fake_envs.FakeSMACVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"]) if smac \
            else fake_envs.FakeMPEVecEnv(N, A, sp["Do"], sp["na"])
        eval_envs = fake_envs.FakeSMACVecEnv(2, A, sp["Do"], sp["Ds"], sp["na"], seed=3) if smac \
            else fake_envs.FakeMPEVecEnv(2, A, sp["Do"], sp["na"], seed=3)
        torch.manual_seed(1)
        np.random.seed(1)
        runner = sp["runner"](config(args, envs, A, os.path.join(tmp, cname), eval_envs))
        key = "run_%s_" % cname
        params(out, key + "init_", runner.policy)
        torch.manual_seed(5)
        runner.warmup()
        for step in range(T):
            res = runner.collect(step)
            if smac:
                values, actions, action_log_probs, rnn_states, rnn_states_critic = res
                obs, share_obs, rewards, dones, infos, avail = envs.step(actions)
                runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, action_log_probs,
                               rnn_states, rnn_states_critic))
            else:
                values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = res
                obs, rewards, dones, infos = envs.step(actions_env)
                runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states,
                               rnn_states_critic))
        runner.compute()
        dump(out, key + "rollout_", runner, True)
        torch.manual_seed(9)
        info = runner.train()
        dump(out, key + "after_", runner, True)
        params(out, key + "final_", runner.policy)
        mark = len(LOGGED)
        runner.eval(777)                      # deterministic policy on the eval envs; what it logs is the result
        meta[cname] = dict(spec={k: v for k, v in sp.items() if k != "runner"},
                           train_info={k: float(v) for k, v in info.items()}, eval_logged=logged_since(mark))

    for cname, sp in specs.items():
        shared_case(cname, sp)

    # ---- separated policies (one policy / trainer / buffer per agent, HAPPO factor bookkeeping in train())
    from onpolicy.runner.separated.mpe_runner import MPERunner as SepMPERunner
    from onpolicy.runner.separated.smac_runner import SMACRunner as SepSMACRunner
    sep_specs = {
        "sep_mpe_mlp": dict(env="MPE", runner=SepMPERunner, T=8, N=4, A=3, Do=6, na=5,
                            args=dict(algorithm_name="mappo", hidden_size=16, ppo_epoch=2, num_mini_batch=2,
                                      share_policy=False)),
        "sep_smac_happo": dict(env="StarCraft2", runner=SepSMACRunner, T=8, N=3, A=4, Do=7, Ds=9, na=6,
                               args=dict(algorithm_name="happo", hidden_size=16, ppo_epoch=1, num_mini_batch=1,
                                         use_proper_time_limits=True, share_policy=False)),
        # trust-region updates; one minibatch per agent, so that the result does not depend on the permutation drawn
        # after the reference's throw-away actor has advanced the random stream (hatrpo_trainer.py:222-225)
        "sep_smac_hatrpo": dict(env="StarCraft2", runner=SepSMACRunner, T=8, N=3, A=4, Do=7, Ds=9, na=6,
                                args=dict(algorithm_name="hatrpo", hidden_size=16, num_mini_batch=1, share_policy=False)),
    }
    SEP_FIELDS = tuple(f for f in FIELDS)
    for cname, sp in sep_specs.items():
        T, N, A = sp["T"], sp["N"], sp["A"]
        args = mg.make_args(env_name=sp["env"], episode_length=T, n_rollout_threads=N, num_env_steps=T * N,
                            use_wandb=False, use_eval=True, n_eval_rollout_threads=2, eval_episodes=4, **sp["args"])
        args.scenario_name = args.map_name = "fake"
        smac = sp["env"] == "StarCraft2"
        envs = fake_envs.FakeSMACVecEnv(N, A, sp["Do"], sp["Ds"], sp["na"]) if smac \
            else fake_envs.FakeMPEVecEnv(N, A, sp["Do"], sp["na"])
        eval_envs = fake_envs.FakeSMACVecEnv(2, A, sp["Do"], sp["Ds"], sp["na"], seed=3) if smac \
            else fake_envs.FakeMPEVecEnv(2, A, sp["Do"], sp["na"], seed=3)
        torch.manual_seed(1)
        np.random.seed(1)
        runner = sp["runner"](config(args, envs, A, os.path.join(tmp, cname), eval_envs))
        key = "run_%s_" % cname
        for a in range(A):
            params(out, key + "init%d_" % a, runner.policy[a])
        torch.manual_seed(5)
        runner.warmup()
        for step in range(T):
            res = runner.collect(step)
            if smac:
                values, actions, action_log_probs, rnn_states, rnn_states_critic = res
                obs, share_obs, rewards, dones, infos, avail = envs.step(actions)
                runner.insert((obs, share_obs, rewards, dones, infos, avail, values, actions, action_log_probs,
                               rnn_states, rnn_states_critic))
            else:
                values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = res
                obs, rewards, dones, infos = envs.step(actions_env)
                runner.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states,
                               rnn_states_critic))
        runner.compute()
        for a in range(A):
            for name in SEP_FIELDS + ("available_actions",):
                out[key + "rollout%d_" % a + name] = np.array(getattr(runner.buffer[a], name), dtype=np.float32)
        torch.manual_seed(9)
        infos_train = runner.train()
        for a in range(A):
            out[key + "factor%d" % a] = np.array(runner.buffer[a].factor, dtype=np.float32)
            params(out, key + "final%d_" % a, runner.policy[a])
        mark = len(LOGGED)
        try:
            runner.eval(555)
            logged = logged_since(mark)
        except ValueError:
            # the reference's separated SMAC eval concatenates per-thread lists and fails when a thread has not
            # finished an episode by the time eval_episodes is reached (smac_runner.py:220); no fixture then
            logged = None
        meta[cname] = dict(spec={k: v for k, v in sp.items() if k != "runner"},
                           train_info=[{k: float(v) for k, v in info.items()} for info in infos_train],
                           eval_logged=logged)

    # ---- Hanabi: the whole turn-based loop for a few episodes
    T, N, A, Do, Ds, na = 6, 5, 3, 9, 12, 7
    args = mg.make_args(env_name="Hanabi", episode_length=T, n_rollout_threads=N, num_env_steps=4 * T * N,
                        hidden_size=16, ppo_epoch=2, num_mini_batch=1, algorithm_name="mappo", use_linear_lr_decay=True, log_interval=1000,
                        save_interval=1000, use_wandb=False)
    args.hanabi_name = "fake"
    envs = fake_envs.FakeChooseVecEnv(N, A, Do, Ds, na)
    torch.manual_seed(1)
    np.random.seed(1)
    args.n_eval_rollout_threads = 3
    runner = HanabiRunner(config(args, envs, A, os.path.join(tmp, "hanabi"),
                                 fake_envs.FakeChooseVecEnv(3, A, Do, Ds, na, seed=4)))
    params(out, "run_hanabi_init_", runner.policy)
    torch.manual_seed(5)
    runner.run()
    dump(out, "run_hanabi_after_", runner, True)
    params(out, "run_hanabi_final_", runner.policy)
    mark = len(LOGGED)
    runner.eval(888)
    meta["hanabi"] = dict(spec=dict(T=T, N=N, A=A, Do=Do, Ds=Ds, na=na), true_total_num_steps=int(runner.true_total_num_steps),
                          env_steps=int(envs.steps), games=int(envs.games), eval_logged=logged_since(mark))
    for cname, sp in h64_specs.items():
        shared_case(cname, sp)
    np.savez_compressed(os.path.join(mg.GOLD, "runner_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "runner_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("runner_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "runner_cases.npz"))))
    print(json.dumps(meta, indent=1)[:1500])


if __name__ == "__main__":
    main()
