"""Worker for the world_size-2 gloo tests: one rank of a data-parallel MAPPO update on a shard of the
rollout threads (host OracleBuffer as the minibatch source, product trainer + DataParallel)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "on-policy_amd"), ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def build_case(n_threads, args_over):
    from helpers import Box, Discrete, make_args, fill_buffer_arrays, buffer_shapes
    T, A, Do, Ds, na, H = 6, 3, 5, 9, 4, 16
    args = make_args(episode_length=T, n_rollout_threads=n_threads, hidden_size=H, ppo_epoch=2,
                     num_mini_batch=1, **args_over)
    spaces = Box((Do,)), Box((Ds,)), Discrete(na)
    return args, spaces, (T, A, Do, Ds, na, H)


def full_arrays(N, dims):
    from helpers import fill_buffer_arrays, buffer_shapes
    T, A, Do, Ds, na, H = dims
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, na, H), np.random.default_rng(5), na=na)
    av = arrays["available_actions"][:-1]
    pick = np.random.default_rng(6).random(av.shape) * av
    arrays["actions"] = pick.argmax(-1)[..., None].astype(np.float32)
    return arrays


def run_update_mat(N_global, lo, hi, args_over):
    """The same for the Multi-Agent Transformer: host OracleBuffer (mat branches of compute_returns, transformer
    sampler) + TransformerPolicy / MATTrainer on rollout threads [lo, hi)."""
    from helpers import load_into
    from oracle import oracle
    from onpolicy.algorithms.mat.algorithm.transformer_policy import TransformerPolicy
    from onpolicy.algorithms.mat.mat_trainer import MATTrainer
    args, spaces, dims = build_case(hi - lo, dict(args_over, algorithm_name="mat", n_embd=16, n_head=2))
    arrays = full_arrays(N_global, dims)
    shard = {k: (v[:, lo:hi] if k != "next_value" else v[lo:hi]) for k, v in arrays.items()}
    torch.manual_seed(1)
    np.random.seed(1)
    policy = TransformerPolicy(args, *spaces, dims[1])
    trainer = MATTrainer(args, policy, dims[1])
    buf = oracle.OracleBuffer(args, dims[1], *spaces)
    load_into(buf, shard)
    buf.compute_returns(shard["next_value"], trainer.value_normalizer)
    trainer.prep_training()
    torch.manual_seed(100)
    info = trainer.train(buf)
    sd = {"transformer." + k: v.detach().cpu().clone() for k, v in policy.transformer.state_dict().items()}
    if trainer.value_normalizer is not None:
        sd["vn.mean"] = trainer.value_normalizer.running_mean.cpu().clone()
        sd["vn.sq"] = trainer.value_normalizer.running_mean_sq.cpu().clone()
    return info, sd, trainer.dp.world_size


def run_update(N_global, lo, hi, args_over, device=None):
    """compute_returns + train on rollout threads [lo, hi) of the global case.  ``device`` None: host
    OracleBuffer + CPU trainer; a HIP device: the HBM buffer and the trainer (fused loss) on it."""
    from helpers import load_into
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    args, spaces, dims = build_case(hi - lo, args_over)
    arrays = full_arrays(N_global, dims)
    shard = {k: (v[:, lo:hi] if k != "next_value" else v[lo:hi]) for k, v in arrays.items()}
    torch.manual_seed(1)
    np.random.seed(1)
    if device is None:
        from oracle import oracle
        policy = R_MAPPOPolicy(args, *spaces)
        trainer = R_MAPPO(args, policy)
        buf = oracle.OracleBuffer(args, dims[1], *spaces)
    else:
        from onpolicy.utils.shared_buffer import SharedReplayBuffer
        args.sampler_rng = "host"
        policy = R_MAPPOPolicy(args, *spaces, device=device)
        trainer = R_MAPPO(args, policy, device=device)
        buf = SharedReplayBuffer(args, dims[1], *spaces, device=device)
    load_into(buf, shard)
    buf.compute_returns(shard["next_value"], trainer.value_normalizer)
    trainer.prep_training()
    torch.manual_seed(100)
    info = trainer.train(buf)
    sd = {"actor." + k: v.detach().cpu().clone() for k, v in policy.actor.state_dict().items()}
    sd.update({"critic." + k: v.detach().cpu().clone() for k, v in policy.critic.state_dict().items()})
    if trainer.value_normalizer is not None and hasattr(trainer.value_normalizer, "running_mean"):
        sd["vn.mean"] = trainer.value_normalizer.running_mean.cpu().clone()
        sd["vn.sq"] = trainer.value_normalizer.running_mean_sq.cpu().clone()
    return info, sd, trainer.dp.world_size


def run_fixture(cname, lo, hi, device=None, sampler_rng="host", fname="trainer_h64_cases"):
    """compute_returns + train on rollout threads [lo, hi) of a REFERENCE-generated hidden-64 case
    (tests/golden/trainer_h64_cases.npz, oracle/make_golden_trainer.py), starting from the reference's initial weights.
    With one minibatch per epoch the union of the ranks' minibatches is the reference's batch, so the data-parallel result
    is comparable with what the reference's single process produced."""
    import json
    from helpers import Box, Discrete, make_args
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    z = np.load(os.path.join(HERE, "golden", fname + ".npz"))
    spec = json.load(open(os.path.join(HERE, "golden", fname + ".json")))[cname]["spec"]
    key = "trn_%s_" % cname
    extra = {} if device is None else {"sampler_rng": sampler_rng}
    args = make_args(episode_length=spec["T"], n_rollout_threads=hi - lo, **dict(spec["args"], **extra))
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["na"])
    torch.manual_seed(1)
    np.random.seed(1)
    if device is None:
        from oracle import oracle
        policy = R_MAPPOPolicy(args, *spaces)
        trainer = R_MAPPO(args, policy)
        buf = oracle.OracleBuffer(args, spec["A"], *spaces)
    else:
        from onpolicy.utils.shared_buffer import SharedReplayBuffer
        policy = R_MAPPOPolicy(args, *spaces, device=device)
        trainer = R_MAPPO(args, policy, device=device)
        buf = SharedReplayBuffer(args, spec["A"], *spaces, device=device)
    with torch.no_grad():
        for net, pre in ((policy.actor, "init_actor."), (policy.critic, "init_critic.")):
            for k, v in net.state_dict().items():
                v.copy_(torch.from_numpy(z[key + pre + k]))
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
                 "active_masks", "action_log_probs", "available_actions", "rewards"):
        dst, src = getattr(buf, name), z[key + "buf_" + name][:, lo:hi]
        if device is None:
            dst[...] = src
        elif dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(np.ascontiguousarray(src)))
    buf.compute_returns(z[key + "next_value"][lo:hi], trainer.value_normalizer)
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    sd = {"final_actor." + k: v.detach().cpu().clone() for k, v in policy.actor.state_dict().items()}
    sd.update({"final_critic." + k: v.detach().cpu().clone() for k, v in policy.critic.state_dict().items()})
    vn = trainer.value_normalizer
    sd["final_norm"] = torch.tensor([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)],
                                    dtype=torch.float64)
    # how the scalar prologue of the updates travelled (DataParallel.begin_scales / minibatch_scales)
    info = dict(info, _scalar_collectives=trainer.dp.scalar_collectives, _scales_reused=trainer.dp.scales_reused,
                _whole_batch_reuses=getattr(buf, "whole_batch_reuses", 0),
                _capture_failures=getattr(getattr(trainer, "_update_graph", None), "capture_failures", 0))
    return info, sd, trainer.dp.world_size


def worker(rank, world, port, N_global, args_over, out_dir, device=None, mat=False, fixture=None, fixture_opts=None):
    """``device``: None = CPU ranks; "cuda:0" = every rank on GPU 0 with the gloo backend (RCCL refuses
    duplicate devices), which exercises the device buffer + fused loss under data parallelism."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from onpolicy.utils import dist as mdist
    if device is not None:
        os.environ["MAPPO_DIST_BACKEND"] = "gloo"
        device = torch.device(device)
        torch.cuda.set_device(device)
    mdist.init_from_env(torch.device("cpu") if device is None else device)
    assert dist.get_backend() == "gloo"
    lo, hi = mdist.shard_threads(N_global, rank, world)
    if fixture is not None:
        info, sd, ws = run_fixture(fixture, lo, hi, device, **(fixture_opts or {}))
    else:
        info, sd, ws = run_update_mat(N_global, lo, hi, args_over) if mat else run_update(N_global, lo, hi, args_over, device)
    assert ws == world
    torch.save({"info": info, "sd": sd, "span": (lo, hi)}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()
