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
    """The HIP device of this process.  Single process: cuda:0 like the reference (train_mpe.py:86-96).  Under a
    one-process-per-GPU launcher (``torchrun --nproc-per-node W train_*.py ...``: WORLD_SIZE / RANK / LOCAL_RANK in
    the environment) the job is data parallel over rollout threads: this rank takes cuda:<LOCAL_RANK>, joins the
    RCCL process group and keeps its contiguous share of ``--n_rollout_threads`` -- ``all_args.n_rollout_threads``
    becomes the LOCAL count, ``all_args.rollout_thread_offset`` the index of its first thread (env seeds are
    offset by it so that the union of the ranks' envs is the single-process set) -- and R_MAPPO all-reduces the
    gradients (onpolicy/utils/dist.py).  ``--num_env_steps`` remains the budget of the WHOLE job: the runners derive
    episode counts, the learning-rate schedule and the logged step counters from ``all_args.global_n_rollout_threads``,
    so every rank makes the same number of train() calls; thread counts that do not divide evenly are rejected."""
    if not (all_args.cuda and torch.cuda.is_available()):
        raise RuntimeError("the rollout buffer of this implementation lives in HBM: a HIP device is required")
    print("choose to use gpu...")
    torch.set_num_threads(all_args.n_training_threads)
    all_args.rollout_thread_offset = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return torch.device("cuda:0")
    from onpolicy.utils import dist as mdist
    single = os.environ.get("MAPPO_SINGLE_DEVICE", "0") == "1"     # test mode: every rank on GPU 0, gloo collectives
    device = torch.device("cuda", 0 if single else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    mdist.init_from_env(device)
    if all_args.n_rollout_threads < world or all_args.n_rollout_threads % world != 0:
        # equal shards only: every rank then runs the same number of episodes and train() calls (each one a collective),
        # and --num_env_steps stays a budget of the whole job (the runners count the job-wide threads)
        raise ValueError("--n_rollout_threads %d must be a positive multiple of the number of ranks %d"
                         % (all_args.n_rollout_threads, world))
    lo, hi = mdist.shard_threads(all_args.n_rollout_threads)
    all_args.global_n_rollout_threads = all_args.n_rollout_threads
    all_args.n_rollout_threads, all_args.rollout_thread_offset = hi - lo, lo
    return device


def new_run_dir(all_args, *levels):
    """<results>/<env_name>/<levels...>/<algorithm>/<experiment>/run<k+1> (results root: $MAPPO_RESULTS_DIR or ./results)."""
    root = Path(os.environ.get("MAPPO_RESULTS_DIR", os.path.join(os.getcwd(), "results")))
    run_dir = root.joinpath(all_args.env_name, *[str(x) for x in levels]) / all_args.algorithm_name / all_args.experiment_name
    run_dir.mkdir(parents=True, exist_ok=True)
    existing = [int(p.name[3:]) for p in run_dir.iterdir() if p.name.startswith("run") and p.name[3:].isdigit()]
    run_dir = run_dir / ("run%d" % (max(existing) + 1 if existing else 1))
    rank = int(os.environ.get("RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and rank > 0:    # ranks of one job never share a directory
        run_dir = run_dir.parent / ("%s_rank%d_%d" % (run_dir.name, rank, os.getpid()))
    run_dir.mkdir(parents=True)
    return run_dir


def seed_everything(all_args):
    try:
        import setproctitle
        setproctitle.setproctitle("-".join([all_args.algorithm_name, all_args.env_name, all_args.experiment_name]))
    except Exception:
        pass
    from onpolicy.utils import gemm_tuning
    # best GEMM kernel per shape (PyTorch TunableOp), winners cached per user; new shapes are only benchmarked inside
    # R_MAPPO.train() (fixed shapes), never during rollouts (row counts vary from step to step)
    gemm_tuning.enable(tune_new=False)
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
