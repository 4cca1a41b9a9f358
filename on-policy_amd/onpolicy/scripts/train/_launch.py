"""What the train scripts share: algorithm-name flag rewrites, device, run directory, seeding, GEMM tuning, the
runner's life cycle.  The reference repeats these ~90 lines in every script (e.g. scripts/train/train_smac.py:
131-252); the per-env scripts here keep only their flags, env factories and runner choice.
"""
import os
from pathlib import Path

import numpy as np
import torch


def apply_algorithm_flags(all_args, allowed=("rmappo", "mappo", "ippo")):
    """scripts/train/train_*.py: the algorithm name decides the recurrent / centralised-V flags."""
    name = all_args.algorithm_name
    if name not in allowed:
        raise NotImplementedError("algorithm %s is outside this implementation (available here: %s)"
                                  % (name, ", ".join(allowed)))
    if name == "rmappo":
        all_args.use_recurrent_policy = True
        all_args.use_naive_recurrent_policy = False
    elif name in ("mappo", "happo", "hatrpo", "mat", "mat_dec"):
        all_args.use_recurrent_policy = False
        all_args.use_naive_recurrent_policy = False
    elif name == "ippo":
        all_args.use_centralized_V = False
    if name == "mat_dec":          # decentralised actor variant of MAT (train_smac.py:151-153)
        all_args.dec_actor = True
        all_args.share_actor = True
    return all_args


def device_of(all_args):
    if all_args.cuda and torch.cuda.is_available():
        print("choose to use gpu...")
        torch.set_num_threads(all_args.n_training_threads)
        return torch.device("cuda:0")
    raise RuntimeError("the rollout buffer of this implementation lives in HBM: a HIP device is required")


def new_run_dir(all_args, *levels):
    """<results>/<env_name>/<levels...>/<algorithm>/<experiment>/run<k+1> (results root: $MAPPO_RESULTS_DIR or ./results)."""
    root = Path(os.environ.get("MAPPO_RESULTS_DIR", os.path.join(os.getcwd(), "results")))
    run_dir = root.joinpath(all_args.env_name, *[str(x) for x in levels]) / all_args.algorithm_name / all_args.experiment_name
    run_dir.mkdir(parents=True, exist_ok=True)
    existing = [int(p.name[3:]) for p in run_dir.iterdir() if p.name.startswith("run") and p.name[3:].isdigit()]
    run_dir = run_dir / ("run%d" % (max(existing) + 1 if existing else 1))
    run_dir.mkdir(parents=True)
    return run_dir


def seed_everything(all_args):
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


def run(runner_cls, all_args, envs, eval_envs, num_agents, device, run_dir):
    config = {"all_args": all_args, "envs": envs, "eval_envs": eval_envs, "num_agents": num_agents,
              "device": device, "run_dir": run_dir}
    runner = runner_cls(config)
    runner.run()
    envs.close()
    if all_args.use_eval and eval_envs is not None and eval_envs is not envs:
        eval_envs.close()
    if not runner.use_wandb:
        runner.writter.export_scalars_to_json(str(runner.log_dir + '/summary.json'))
        runner.writter.close()
    return runner
