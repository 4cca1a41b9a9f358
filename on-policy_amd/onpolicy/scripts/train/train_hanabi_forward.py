#!/usr/bin/env python
"""Train script for Hanabi (turn-based runner) -- flags and flow of the reference's
onpolicy/scripts/train/train_hanabi_forward.py (parse_args :61-69, env factories :16-58, runner :158-162).
The env is the in-tree batched stepper (``onpolicy.envs.hanabi``): all rollout threads advance in one native
call (``HanabiBatchVecEnv``).  ``--use_subproc_envs`` builds the reference's layout instead -- one ``HanabiEnv``
per thread behind ``ChooseSubprocVecEnv`` / ``ChooseDummyVecEnv``; both give identical games for the same seeds.

    python -m onpolicy.scripts.train.train_hanabi_forward --env_name Hanabi --hanabi_name Hanabi-Full --num_agents 2 ...
"""
import sys

from onpolicy.config import get_config
from onpolicy.envs.env_wrappers import ChooseDummyVecEnv, ChooseSubprocVecEnv
from onpolicy.scripts.train import _launch


def make_env(all_args, n_threads, seed_of_rank):
    if all_args.env_name != "Hanabi":
        raise NotImplementedError("Can not support the " + all_args.env_name + " environment.")
    assert 1 < all_args.num_agents < 6, "num_agents can be only between 2-5."
    if not all_args.use_subproc_envs:
        from onpolicy.envs.hanabi.batch import HanabiBatchVecEnv
        # the runner copies every row it keeps, so the stepper's own arrays are handed out as they are
        return HanabiBatchVecEnv(all_args, [seed_of_rank(rank) for rank in range(n_threads)], copy=False)

    def get_env_fn(rank):
        def init_env():
            if all_args.env_name != "Hanabi":
                raise NotImplementedError("Can not support the " + all_args.env_name + " environment.")
            assert 1 < all_args.num_agents < 6, "num_agents can be only between 2-5."
            from onpolicy.envs.hanabi.Hanabi_Env import HanabiEnv
            env = HanabiEnv(all_args, seed_of_rank(rank))
            env.seed(seed_of_rank(rank))
            return env
        return init_env
    if n_threads == 1:
        return ChooseDummyVecEnv([get_env_fn(0)])
    return ChooseSubprocVecEnv([get_env_fn(i) for i in range(n_threads)])


def parse_args(args, parser):
    parser.add_argument('--hanabi_name', type=str, default='Hanabi-Very-Small', help="Which env to run on")
    parser.add_argument('--num_agents', type=int, default=2, help="number of players")
    parser.add_argument('--use_subproc_envs', action='store_true', default=False,
                        help="one HanabiEnv per rollout thread behind the Choose* VecEnv wrappers (reference layout) "
                             "instead of the batched stepper")
    return parser.parse_known_args(args)[0]


def main(args):
    all_args = _launch.apply_algorithm_flags(parse_args(args, get_config()), ("rmappo", "mappo", "ippo"))
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.hanabi_name)
    _launch.seed_everything(all_args)
    envs = make_env(all_args, all_args.n_rollout_threads, lambda rank: all_args.seed + (getattr(all_args, "rollout_thread_offset", 0) + rank) * 1000)
    eval_envs = make_env(all_args, all_args.n_eval_rollout_threads,
                         lambda rank: all_args.seed * 50000 + rank * 10000) if all_args.use_eval else None
    if not all_args.share_policy:
        raise NotImplementedError("the separated Hanabi runner is outside this implementation")
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner as Runner
    return _launch.run(Runner, all_args, envs, eval_envs, all_args.num_agents, device, run_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
