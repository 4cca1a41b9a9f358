#!/usr/bin/env python
"""MPE training entry point with the command line of the reference's
onpolicy/scripts/train/train_mpe.py (main :64, parse_args :52-61, env factories :16-49): same flags,
same algorithm-name -> recurrent-flag rewrite, same config dict handed to ``MPERunner``.

Differences: simple_spread comes from the in-tree vectorised implementation (no gym / subprocess workers;
``--use_device_env`` keeps the training worlds on the GPU next to policy and buffer);
other scenarios are built per worker from an external env tree (MAPPO_ENVS_PATH); wandb / setproctitle are optional.
Example (BASELINE.json configs[0]):
    python -m onpolicy.scripts.train.train_mpe --env_name MPE --scenario_name simple_spread \\
        --num_agents 3 --num_landmarks 3 --n_rollout_threads 8 --episode_length 25 \\
        --num_env_steps 20000 --ppo_epoch 10 --use_ReLU --use_wandb
"""
import sys

from onpolicy.config import get_config
from onpolicy.scripts.train import _launch


def make_train_env(all_args, n_threads=None, seed_offset=0, device=None):
    n = all_args.n_rollout_threads if n_threads is None else n_threads
    seed_offset += 1000 * getattr(all_args, "rollout_thread_offset", 0)      # data-parallel rank: its own worlds
    if all_args.scenario_name == "simple_spread" and device is not None:    # worlds held on the device (row f1)
        from onpolicy.envs.mpe.simple_spread import TorchSimpleSpread
        return TorchSimpleSpread(n, all_args.num_agents, all_args.num_landmarks, all_args.episode_length,
                                 seed=all_args.seed + seed_offset, device=device)
    if all_args.scenario_name == "simple_spread":       # built in, all worlds advanced by one numpy pass
        from onpolicy.envs.mpe.simple_spread import VecSimpleSpread
        return VecSimpleSpread(n, all_args.num_agents, all_args.num_landmarks, all_args.episode_length,
                               seed=all_args.seed + seed_offset)
    # any other scenario: one MPEEnv per worker like the reference (train_mpe.py:16-32); the scenario modules
    # come from an external env tree (MAPPO_ENVS_PATH)
    from onpolicy.envs.env_wrappers import DummyVecEnv, SubprocVecEnv
    from onpolicy.envs.mpe.MPE_env import MPEEnv

    def get_env_fn(rank):
        def init_env():
            env = MPEEnv(all_args)
            env.seed(all_args.seed + seed_offset + rank * 1000)
            return env
        return init_env
    return DummyVecEnv([get_env_fn(0)]) if n == 1 else SubprocVecEnv([get_env_fn(i) for i in range(n)])


def parse_args(args, parser):
    parser.add_argument('--scenario_name', type=str, default='simple_spread', help="Which scenario to run on")
    parser.add_argument("--num_landmarks", type=int, default=3)
    parser.add_argument('--num_agents', type=int, default=2, help="number of players")
    parser.add_argument('--use_device_env', action='store_true', default=False,
                        help="simple_spread with the training worlds held as tensors on the policy's device "
                             "(shared-policy runner): no per-step host round trip")
    return parser.parse_known_args(args)[0]


def main(args):
    all_args = parse_args(args, get_config())
    _launch.apply_algorithm_flags(all_args, ("rmappo", "mappo", "ippo", "happo", "hatrpo"))   # happo / hatrpo: separated runner
    assert (all_args.share_policy is True and all_args.scenario_name == 'simple_speaker_listener') is False, (
        "The simple_speaker_listener scenario can not use shared policy. Please check the config.py.")
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.scenario_name)
    _launch.seed_everything(all_args)
    shared = all_args.share_policy and all_args.algorithm_name not in ("happo", "hatrpo")
    if all_args.use_device_env and not (shared and all_args.scenario_name == "simple_spread"):
        raise NotImplementedError("--use_device_env: simple_spread with the shared-policy runner only")
    envs = make_train_env(all_args, device=device if all_args.use_device_env else None)
    eval_envs = make_train_env(all_args, all_args.n_eval_rollout_threads, 50000) if all_args.use_eval else None
    if shared:
        from onpolicy.runner.shared.mpe_runner import MPERunner as Runner
    else:
        from onpolicy.runner.separated.mpe_runner import MPERunner as Runner
    return _launch.run(Runner, all_args, envs, eval_envs, all_args.num_agents, device, run_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
