"""-m gpu, PARKED (opt-in with MAPPO_PENDING_GPU_TESTS=1): device runs of the Multi-Agent Transformer and HATRPO trainers,
which are outside SURVEY.md section 8's hot-path rows (only the MAT *buffer hooks* are in scope and those are covered by
tests/test_gpu_mat.py).  Both trainers are pinned to the reference on the CPU (tests/test_mat_trainer_cpu.py,
tests/test_hatrpo_cpu.py); no GPU time is spent on them."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import Box, Discrete, load_into, make_args

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("MAPPO_PENDING_GPU_TESTS") != "1",
                                 reason="not yet verified on the GPU box (set MAPPO_PENDING_GPU_TESTS=1)")]
DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("idx", range(6))
def test_mat_trainer_on_device_vs_reference(gold, idx):
    """tests/test_mat_trainer_cpu.py::test_policy_and_trainer_match_reference with the policy on the GPU and the HBM
    buffer (mat GAE kernel, moments from its epilogue, transformer sampler) in place of the oracle buffer."""
    from test_mat_trainer_cpu import INFO_KEYS, T, N, Do, Ds, _BoundedBox, _cases, _parse
    from onpolicy.algorithms.mat.algorithm.transformer_policy import TransformerPolicy
    from onpolicy.algorithms.mat.mat_trainer import MATTrainer
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    z, cases = _cases(gold)
    name, algo, extra = cases[idx]
    key = "mt_%s_" % name
    A, k, is_box = (int(v) for v in z[key + "spec"])
    args = make_args(episode_length=T, n_rollout_threads=N, algorithm_name=algo, ppo_epoch=2, num_mini_batch=2,
                     sampler_rng="host", **_parse(extra))
    act_space = _BoundedBox((k,)) if is_box else Discrete(k)
    torch.manual_seed(11)
    np.random.seed(11)
    policy = TransformerPolicy(args, Box((Do,)), Box((Ds,)), act_space, A, device=DEV)
    for pname, p in policy.transformer.state_dict().items():
        np.testing.assert_allclose(p.cpu().numpy(), z[key + "init_" + pname], rtol=0, atol=2e-6, err_msg=pname)
    buf = SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), act_space, device=DEV)
    load_into(buf, {f[len(key) + 4:]: z[f] for f in z.files if f.startswith(key + "buf_")})
    rows = lambda a: a.reshape(-1, *a.shape[2:])          # noqa: E731
    avail = rows(buf.available_actions[0]) if buf.available_actions is not None else None
    policy.eval()
    torch.manual_seed(23)
    with torch.no_grad():
        values, actions, logp, _, _ = policy.get_actions(rows(buf.share_obs[0]), rows(buf.obs[0]), rows(buf.rnn_states[0]),
                                                         rows(buf.rnn_states_critic[0]), rows(buf.masks[0]), avail, True)
    if not is_box:
        np.testing.assert_array_equal(actions.cpu().numpy(), z[key + "det_actions"])
    np.testing.assert_allclose(values.cpu().numpy(), z[key + "det_values"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), z[key + "det_logp"], rtol=1e-3, atol=1e-4)
    trainer = MATTrainer(args, policy, A, device=DEV)
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"])
    np.testing.assert_array_equal(buf.advantages.cpu().numpy(), z[key + "advantages"])
    torch.manual_seed(31)
    info = trainer.train(buf)
    np.testing.assert_allclose([info[k] for k in INFO_KEYS], z[key + "info"], rtol=5e-3, atol=5e-4)
    for pname, p in policy.transformer.state_dict().items():
        np.testing.assert_allclose(p.cpu().numpy(), z[key + "final_" + pname], rtol=5e-3, atol=5e-4, err_msg=pname)


@pytest.mark.parametrize("algo", ["mat", "mat_dec"])
def test_smac_runner_with_the_transformer_on_device(tmp_path, algo):
    import fake_envs
    from onpolicy.runner.shared.smac_runner import SMACRunner
    from onpolicy.scripts.train import _launch
    T, N, A, Do, Ds, na = 6, 3, 4, 7, 9, 6
    args = make_args(env_name="StarCraft2", algorithm_name=algo, episode_length=T, n_rollout_threads=N,
                     num_env_steps=3 * T * N, n_embd=16, n_head=2, ppo_epoch=2, num_mini_batch=1, use_wandb=False,
                     use_eval=True, n_eval_rollout_threads=2, eval_episodes=2, eval_interval=1, log_interval=1)
    _launch.apply_algorithm_flags(args, ("mat", "mat_dec"))
    args.map_name = "fake"
    runner = SMACRunner({"all_args": args, "envs": fake_envs.FakeSMACVecEnv(N, A, Do, Ds, na),
                         "eval_envs": fake_envs.FakeSMACVecEnv(2, A, Do, Ds, na, seed=3), "num_agents": A,
                         "device": DEV, "run_dir": tmp_path})
    runner.run()
    tags = {json.loads(l)["tag"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))}
    assert {"value_loss", "policy_loss", "ratio", "eval_win_rate"} <= tags
    assert os.path.exists(os.path.join(runner.save_dir, "transformer_0.pt"))


@pytest.mark.parametrize("cname", ["mlp", "mlp_popart", "mlp_nonorm", "gru", "rejected"])
def test_hatrpo_on_device_buffer_vs_reference(gold, cname):
    """tests/test_hatrpo_cpu.py on the HBM separated buffer with the networks on the GPU."""
    from test_hatrpo_cpu import BUF
    from onpolicy.algorithms.hatrpo.hatrpo_trainer import HATRPO
    from onpolicy.algorithms.hatrpo.policy import HATRPO_Policy
    from onpolicy.utils.separated_buffer import SeparatedReplayBuffer
    z, meta = gold.npz("hatrpo_cases"), gold.meta("hatrpo_cases")[cname]
    spec, key = meta["spec"], "hat_%s_" % cname
    args = make_args(episode_length=spec["T"], n_rollout_threads=spec["N"], sampler_rng="host", **spec["args"])
    spaces = Box((spec["Do"],)), Box((spec["Ds"],)), Discrete(spec["act"][1])
    torch.manual_seed(1)
    np.random.seed(1)
    policy = HATRPO_Policy(args, *spaces, device=DEV)
    trainer = HATRPO(args, policy, device=DEV)
    buf = SeparatedReplayBuffer(args, *spaces, device=DEV)
    for name in BUF:
        dst = getattr(buf, name)
        if dst.stride()[0] != 0:
            dst.copy_(torch.from_numpy(z[key + "buf_" + name]))
    buf.compute_returns(z[key + "next_value"], trainer.value_normalizer)
    np.testing.assert_array_equal(buf.returns.cpu().numpy(), z[key + "returns"])
    buf.update_factor(z[key + "factor"])
    trainer.prep_training()
    torch.manual_seed(21)
    info = trainer.train(buf)
    for k, v in meta["train_info"].items():
        assert info[k] == pytest.approx(v, rel=1e-2, abs=1e-4), (k, info[k], v)
    for prefix, module in ((key + "final_actor.", policy.actor), (key + "final_critic.", policy.critic)):
        for k, v in module.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), z[prefix + k], rtol=1e-2, atol=5e-4, err_msg=prefix + k)
