#!/usr/bin/env python
"""Train script for Google Research Football -- flags and flow of the reference's
onpolicy/scripts/train/train_football.py (parse_args :58-94, env factories :20-55, runner :186-189).
``onpolicy.envs.football`` must come from an external env tree (``MAPPO_ENVS_PATH``).

    python -m onpolicy.scripts.train.train_football --env_name Football --scenario_name academy_3_vs_1_with_keeper ...
"""
import sys

from onpolicy.config import get_config
from onpolicy.envs.env_wrappers import DummyVecEnv, SubprocVecEnv
from onpolicy.scripts.train import _launch


def make_env(all_args, n_threads, seed_of_rank):
    def get_env_fn(rank):
        def init_env():
            if all_args.env_name != "Football":
                raise NotImplementedError("Can not support the " + all_args.env_name + " environment.")
            from onpolicy.envs.football.Football_Env import FootballEnv
            env = FootballEnv(all_args)
            env.seed(seed_of_rank(rank))
            return env
        return init_env
    if n_threads == 1:
        return DummyVecEnv([get_env_fn(0)])
    return SubprocVecEnv([get_env_fn(i) for i in range(n_threads)])


def parse_args(args, parser):
    parser.add_argument("--scenario_name", type=str, default="academy_3_vs_1_with_keeper",
                        help="which scenario to run on.")
    parser.add_argument("--num_agents", type=int, default=3, help="number of controlled players.")
    parser.add_argument("--representation", type=str, default="simple115v2",
                        choices=["simple115v2", "extracted", "pixels_gray", "pixels"],
                        help="representation used to build the observation.")
    parser.add_argument("--rewards", type=str, default="scoring", help="comma separated list of rewards to be added.")
    parser.add_argument("--smm_width", type=int, default=96, help="width of super minimap.")
    parser.add_argument("--smm_height", type=int, default=72, help="height of super minimap.")
    parser.add_argument("--remove_redundancy", action="store_true", default=False,
                        help="by default False. If True, remove redundancy features")
    parser.add_argument("--zero_feature", action="store_true", default=False,
                        help="by default False. If True, replace -1 by 0")
    parser.add_argument("--eval_deterministic", action="store_false", default=True,
                        help="by default True. If False, sample action according to probability")
    parser.add_argument("--share_reward", action='store_false', default=True,
                        help="by default true. If false, use different reward for each agent.")
    parser.add_argument("--save_videos", action="store_true", default=False,
                        help="by default, do not save render video. If set, save video.")
    parser.add_argument("--video_dir", type=str, default="", help="directory to save videos.")
    return parser.parse_known_args(args)[0]


def main(args):
    all_args = _launch.apply_algorithm_flags(parse_args(args, get_config()), ("rmappo", "mappo", "ippo"))
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.scenario_name)
    _launch.seed_everything(all_args)
    envs = make_env(all_args, all_args.n_rollout_threads, lambda rank: all_args.seed + (getattr(all_args, "rollout_thread_offset", 0) + rank) * 1000)
    eval_envs = make_env(all_args, all_args.n_eval_rollout_threads,
                         lambda rank: all_args.seed * 50000 + rank * 10000) if all_args.use_eval else None
    if not all_args.share_policy:
        raise NotImplementedError("the separated football runner is outside this implementation")
    from onpolicy.runner.shared.football_runner import FootballRunner as Runner
    return _launch.run(Runner, all_args, envs, eval_envs, all_args.num_agents, device, run_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
