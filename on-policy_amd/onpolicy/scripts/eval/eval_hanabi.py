#!/usr/bin/env python
"""Score a saved Hanabi policy over many deterministic games -- flags and flow of the reference's
onpolicy/scripts/eval/eval_hanabi.py (``--use_eval`` and ``--model_dir`` are required, :93-94; the runner's
``eval_100k`` plays the games, :166).  ``--eval_games`` (default 100000, the reference's fixed count) is the one
extra flag.

    python -m onpolicy.scripts.eval.eval_hanabi --env_name Hanabi --hanabi_name Hanabi-Full --num_agents 2 \
        --use_eval --n_eval_rollout_threads 1000 --model_dir results/Hanabi/.../models
"""
import sys

from onpolicy.config import get_config
from onpolicy.scripts.train import _launch
from onpolicy.scripts.train.train_hanabi_forward import make_env
from onpolicy.scripts.train.train_hanabi_forward import parse_args as _train_flags


def parse_args(args, parser):
    parser.add_argument('--eval_games', type=int, default=100000, help="number of games to average the score over")
    return _train_flags(args, parser)


def main(args):
    all_args = _launch.apply_algorithm_flags(parse_args(args, get_config()), ("rmappo", "mappo", "ippo"))
    assert all_args.use_eval, ("u need to set use_eval be True")
    assert not (all_args.model_dir is None or all_args.model_dir == ""), ("set model_dir first")
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.hanabi_name)
    _launch.seed_everything(all_args)
    envs = make_env(all_args, all_args.n_rollout_threads, lambda rank: all_args.seed + rank * 1000)
    eval_envs = make_env(all_args, all_args.n_eval_rollout_threads, lambda rank: all_args.seed * 50000 + rank * 10000)
    if not all_args.share_policy:
        raise NotImplementedError("the separated Hanabi runner is outside this implementation")
    from onpolicy.runner.shared.hanabi_runner_forward import HanabiRunner as Runner
    runner = Runner({"all_args": all_args, "envs": envs, "eval_envs": eval_envs, "num_agents": all_args.num_agents,
                     "device": device, "run_dir": run_dir})
    score = runner.eval_100k(all_args.eval_games)
    envs.close()
    eval_envs.close()
    return score


if __name__ == "__main__":
    main(sys.argv[1:])
