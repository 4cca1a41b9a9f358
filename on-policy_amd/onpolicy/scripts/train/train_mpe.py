#!/usr/bin/env python
"""MPE training entry point with the command line of the reference's
onpolicy/scripts/train/train_mpe.py (main :64, parse_args :52-61, env factories :16-49): same flags,
same algorithm-name -> recurrent-flag rewrite, same config dict handed to ``MPERunner``.

Differences: the envs come from the in-tree vectorised simple_spread (no gym / subprocess workers;
other scenarios need the reference's env package on the path); wandb / setproctitle are optional.
Example (BASELINE.json configs[0]):
    python -m onpolicy.scripts.train.train_mpe --env_name MPE --scenario_name simple_spread \\
        --num_agents 3 --num_landmarks 3 --n_rollout_threads 8 --episode_length 25 \\
        --num_env_steps 20000 --ppo_epoch 10 --use_ReLU --use_wandb
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

from onpolicy.config import get_config


def make_train_env(all_args, n_threads=None, seed_offset=0):
    if all_args.scenario_name != "simple_spread":
        raise NotImplementedError("only simple_spread is built in; scenario %r needs the reference's "
                                  "onpolicy.envs.mpe package" % all_args.scenario_name)
    from onpolicy.envs.mpe.simple_spread import VecSimpleSpread
    n = all_args.n_rollout_threads if n_threads is None else n_threads
    return VecSimpleSpread(n, all_args.num_agents, all_args.num_landmarks, all_args.episode_length,
                           seed=all_args.seed + seed_offset)


def parse_args(args, parser):
    parser.add_argument('--scenario_name', type=str, default='simple_spread', help="Which scenario to run on")
    parser.add_argument("--num_landmarks", type=int, default=3)
    parser.add_argument('--num_agents', type=int, default=2, help="number of players")
    return parser.parse_known_args(args)[0]


def main(args):
    parser = get_config()
    all_args = parse_args(args, parser)

    if all_args.algorithm_name == "rmappo":
        all_args.use_recurrent_policy = True
        all_args.use_naive_recurrent_policy = False
    elif all_args.algorithm_name in ("mappo", "happo"):     # happo: separated runner only (--share_policy false)
        all_args.use_recurrent_policy = False
        all_args.use_naive_recurrent_policy = False
    elif all_args.algorithm_name == "ippo":
        all_args.use_centralized_V = False
    else:
        raise NotImplementedError("algorithm %s is outside this implementation" % all_args.algorithm_name)
    assert (all_args.share_policy is True and all_args.scenario_name == 'simple_speaker_listener') is False, (
        "The simple_speaker_listener scenario can not use shared policy. Please check the config.py.")

    if all_args.cuda and torch.cuda.is_available():
        print("choose to use gpu...")
        device = torch.device("cuda:0")
        torch.set_num_threads(all_args.n_training_threads)
    else:
        raise RuntimeError("the rollout buffer of this implementation lives in HBM: a HIP device is required")

    run_dir = Path(os.environ.get("MAPPO_RESULTS_DIR", os.path.join(os.getcwd(), "results"))) / all_args.env_name \
        / all_args.scenario_name / all_args.algorithm_name / all_args.experiment_name
    run_dir.mkdir(parents=True, exist_ok=True)
    existing = [int(p.name[3:]) for p in run_dir.iterdir() if p.name.startswith("run") and p.name[3:].isdigit()]
    run_dir = run_dir / ("run%d" % (max(existing) + 1 if existing else 1))
    run_dir.mkdir(parents=True)
    try:
        import setproctitle
        setproctitle.setproctitle("-".join([all_args.algorithm_name, all_args.env_name, all_args.experiment_name]))
    except Exception:
        pass

    from onpolicy.utils import gemm_tuning
    gemm_tuning.enable()          # best GEMM kernel per shape (PyTorch TunableOp), winners cached per user

    torch.manual_seed(all_args.seed)
    torch.cuda.manual_seed_all(all_args.seed)
    np.random.seed(all_args.seed)

    envs = make_train_env(all_args)
    eval_envs = make_train_env(all_args, all_args.n_eval_rollout_threads, 50000) if all_args.use_eval else None
    config = {"all_args": all_args, "envs": envs, "eval_envs": eval_envs, "num_agents": all_args.num_agents,
              "device": device, "run_dir": run_dir}

    if all_args.share_policy:
        from onpolicy.runner.shared.mpe_runner import MPERunner as Runner
    else:
        from onpolicy.runner.separated.mpe_runner import MPERunner as Runner
    runner = Runner(config)
    runner.run()

    envs.close()
    if eval_envs is not None:
        eval_envs.close()
    if not runner.use_wandb:
        runner.writter.export_scalars_to_json(str(runner.log_dir + '/summary.json'))
        runner.writter.close()
    return runner


if __name__ == "__main__":
    main(sys.argv[1:])
