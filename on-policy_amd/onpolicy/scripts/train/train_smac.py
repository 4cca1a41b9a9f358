#!/usr/bin/env python
"""Train script for StarCraft II micromanagement -- flags and flow of the reference's
onpolicy/scripts/train/train_smac.py (parse_args :106-122, env factories :50-103, runner choice :232-240).
The SMAC environments themselves are not part of this package: ``onpolicy.envs.starcraft2`` must come from an
external env tree (``MAPPO_ENVS_PATH``, see onpolicy/envs/__init__.py).

    python -m onpolicy.scripts.train.train_smac --env_name StarCraft2 --map_name MMM2 --algorithm_name rmappo ...
"""
import sys

from onpolicy.config import get_config
from onpolicy.envs.env_wrappers import ShareDummyVecEnv, ShareSubprocVecEnv
from onpolicy.scripts.train import _launch


def parse_smacv2_distribution(args):
    """Unit-type / start-position distributions of SMACv2 by race (train_smac.py:15-48)."""
    n_units, n_enemies = (int(x) for x in args.units.split('v'))
    teams = {"protoss": (["stalker", "zealot", "colossus"], [0.45, 0.45, 0.1]),
             "zerg": (["zergling", "baneling", "hydralisk"], [0.45, 0.1, 0.45]),
             "terran": (["marine", "marauder", "medivac"], [0.45, 0.45, 0.1])}
    config = {"n_units": n_units, "n_enemies": n_enemies,
              "start_positions": {"dist_type": "surrounded_and_reflect", "p": 0.5, "map_x": 32, "map_y": 32}}
    for race, (unit_types, weights) in teams.items():
        if race in args.map_name:
            config["team_gen"] = {"dist_type": "weighted_teams", "unit_types": unit_types, "weights": weights,
                                  "observe": True}
            break
    return config


def _make_env(all_args):
    name = all_args.env_name
    if name == "StarCraft2":
        from onpolicy.envs.starcraft2.StarCraft2_Env import StarCraft2Env
        return StarCraft2Env(all_args)
    if name == "StarCraft2v2":
        from onpolicy.envs.starcraft2.SMACv2_modified import SMACv2
        return SMACv2(capability_config=parse_smacv2_distribution(all_args), map_name=all_args.map_name)
    if name == "SMAC":
        from onpolicy.envs.starcraft2.SMAC import SMAC
        return SMAC(map_name=all_args.map_name)
    if name == "SMACv2":
        from onpolicy.envs.starcraft2.SMACv2 import SMACv2
        return SMACv2(capability_config=parse_smacv2_distribution(all_args), map_name=all_args.map_name)
    raise NotImplementedError("Can not support the " + name + " environment.")


def make_env(all_args, n_threads, seed_of_rank):
    def get_env_fn(rank):
        def init_env():
            env = _make_env(all_args)
            env.seed(seed_of_rank(rank))
            return env
        return init_env
    if n_threads == 1:
        return ShareDummyVecEnv([get_env_fn(0)])
    return ShareSubprocVecEnv([get_env_fn(i) for i in range(n_threads)])


def parse_args(args, parser):
    parser.add_argument('--map_name', type=str, default='3m', help="Which smac map to run on")
    parser.add_argument('--units', type=str, default='10v10')       # for smac v2
    for flag in ("add_move_state", "add_local_obs", "add_distance_state", "add_enemy_action_state", "add_agent_id",
                 "add_visible_state", "add_xy_state"):
        parser.add_argument("--" + flag, action='store_true', default=False)
    for flag in ("use_state_agent", "use_mustalive", "add_center_xy"):
        parser.add_argument("--" + flag, action='store_false', default=True)
    return parser.parse_known_args(args)[0]


def num_agents_of(all_args):
    if all_args.env_name in ("SMACv2", "StarCraft2v2"):
        return parse_smacv2_distribution(all_args)['n_units']
    from onpolicy.envs.starcraft2.smac_maps import get_map_params
    return get_map_params(all_args.map_name)["n_agents"]


def main(args):
    all_args = _launch.apply_algorithm_flags(parse_args(args, get_config()), ("rmappo", "mappo", "ippo", "happo", "hatrpo", "mat", "mat_dec"))
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.map_name)
    _launch.seed_everything(all_args)
    envs = make_env(all_args, all_args.n_rollout_threads, lambda rank: all_args.seed + (getattr(all_args, "rollout_thread_offset", 0) + rank) * 1000)
    eval_envs = make_env(all_args, all_args.n_eval_rollout_threads,
                         lambda rank: all_args.seed * 50000 + rank * 10000) if all_args.use_eval else None
    if all_args.share_policy and all_args.algorithm_name not in ("happo", "hatrpo"):
        from onpolicy.runner.shared.smac_runner import SMACRunner as Runner
    else:
        from onpolicy.runner.separated.smac_runner import SMACRunner as Runner
    return _launch.run(Runner, all_args, envs, eval_envs, num_agents_of(all_args), device, run_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
