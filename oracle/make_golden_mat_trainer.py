#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/mat_trainer_cases.npz from the REFERENCE's Multi-Agent Transformer:
TransformerPolicy (onpolicy/algorithms/mat/algorithm/transformer_policy.py) over MultiAgentTransformer
(ma_transformer.py) and MATTrainer.train (mat_trainer.py) on the reference's SharedReplayBuffer built with
algorithm_name mat / mat_dec.  Per case: initial parameters under a seed, sampled and deterministic actions / log-probs /
values, evaluate_actions outputs, and the parameters + logged scalars after one train() call.

    python oracle/make_golden_mat_trainer.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference)

from onpolicy.algorithms.mat.algorithm.transformer_policy import TransformerPolicy  # noqa: E402
from onpolicy.algorithms.mat.mat_trainer import MATTrainer  # noqa: E402

ref = mg.ref


class BoxWithBounds(mg.Box):
    def __init__(self, shape):
        mg.Box.__init__(self, shape)
        self.high = np.ones(shape, dtype=np.float32)


BoxWithBounds.__name__ = "Box"

CASES = [
    # name, algo, act space, A, extra args
    ("discrete", "mat", ("Discrete", 5), 3, dict(n_block=1, n_embd=16, n_head=1)),
    ("discrete_deep", "mat", ("Discrete", 6), 4, dict(n_block=2, n_embd=16, n_head=2, use_valuenorm=False,
                                                      use_huber_loss=False, use_policy_active_masks=False)),
    ("encode_state", "mat", ("Discrete", 4), 2, dict(n_block=1, n_embd=8, n_head=1, encode_state=True,
                                                     use_clipped_value_loss=False, use_value_active_masks=False)),
    ("mat_dec", "mat_dec", ("Discrete", 5), 3, dict(n_block=1, n_embd=16, n_head=1, dec_actor=True, share_actor=True)),
    ("dec_actor_own", "mat", ("Discrete", 5), 3, dict(n_block=1, n_embd=16, n_head=1, dec_actor=True,
                                                      use_max_grad_norm=False)),
    ("continuous", "mat", ("Box", 2), 3, dict(n_block=2, n_embd=16, n_head=2)),
]


def main():
    out, names = {}, []
    T, N, Do, Ds = 5, 4, 7, 9
    for name, algo, (kind, k), A, extra in CASES:
        args = mg.make_args(episode_length=T, n_rollout_threads=N, algorithm_name=algo, ppo_epoch=2, num_mini_batch=2,
                            **extra)
        act_space = mg.Discrete(k) if kind == "Discrete" else BoxWithBounds((k,))
        torch.manual_seed(11)
        np.random.seed(11)
        policy = TransformerPolicy(args, mg.Box((Do,)), mg.Box((Ds,)), act_space, A)
        trainer = MATTrainer(args, policy, A)
        key = "mt_%s_" % name
        for pname, p in policy.transformer.state_dict().items():
            out[key + "init_" + pname] = p.detach().numpy().copy()
        rng = np.random.default_rng(500 + len(names))
        buf = ref.SharedReplayBuffer(args, A, mg.Box((Do,)), mg.Box((Ds,)), act_space)
        nv = mg.fill_buffer(buf, rng)
        if kind == "Box":
            buf.actions[:] = rng.standard_normal(buf.actions.shape).astype(np.float32)
            buf.action_log_probs[:] = rng.standard_normal(buf.action_log_probs.shape).astype(np.float32) * 0.1 - 1.0
        # rollout-side calls on the rows of step 0
        rows = lambda a: np.concatenate(a)          # noqa: E731  ([N, A, .] -> [N*A, .], what the runners pass)
        avail = rows(buf.available_actions[0]) if buf.available_actions is not None else None
        policy.eval()
        for mode, det in (("sample", False), ("det", True)):
            torch.manual_seed(23)
            with torch.no_grad():
                values, actions, logp, rs, rc = policy.get_actions(rows(buf.share_obs[0]), rows(buf.obs[0]),
                                                                   rows(buf.rnn_states[0]), rows(buf.rnn_states_critic[0]),
                                                                   rows(buf.masks[0]), avail, det)
            out[key + mode + "_values"], out[key + mode + "_actions"] = values.numpy(), actions.numpy()
            out[key + mode + "_logp"] = logp.numpy()
        with torch.no_grad():
            out[key + "get_values"] = policy.get_values(rows(buf.share_obs[0]), rows(buf.obs[0]),
                                                        rows(buf.rnn_states_critic[0]), rows(buf.masks[0])).numpy()
            ev = policy.evaluate_actions(rows(buf.share_obs[1]), rows(buf.obs[1]), None, None, rows(buf.actions[1]),
                                         None, rows(buf.available_actions[1]) if avail is not None else None,
                                         torch.from_numpy(rows(buf.active_masks[1])))
        out[key + "eval_values"], out[key + "eval_logp"], out[key + "eval_entropy"] = (x.numpy() for x in ev)
        # one update phase
        for fname in ("share_obs", "obs", "rewards", "value_preds", "masks", "bad_masks", "active_masks", "actions",
                      "action_log_probs", "available_actions"):
            arr = getattr(buf, fname)
            if arr is not None:
                out[key + "buf_" + fname] = arr.copy()
        out[key + "next_value"] = nv.copy()
        buf.compute_returns(nv, trainer.value_normalizer)
        out[key + "returns"], out[key + "advantages"] = buf.returns.copy(), buf.advantages.copy()
        torch.manual_seed(31)
        with mg.PermRecorder() as rec:
            info = trainer.train(buf)
        out[key + "perms"] = np.stack([c.astype(np.int64) for c in rec.calls])
        out[key + "info"] = np.array([float(info[k]) for k in ("value_loss", "policy_loss", "dist_entropy",
                                                               "actor_grad_norm", "critic_grad_norm", "ratio")])
        for pname, p in policy.transformer.state_dict().items():
            out[key + "final_" + pname] = p.detach().numpy().copy()
        if trainer.value_normalizer is not None:
            vn = trainer.value_normalizer
            out[key + "norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq), float(vn.debiasing_term)])
        out[key + "spec"] = np.array([A, k, int(kind == "Box")])
        names.append("%s|%s|%s" % (name, algo, ",".join("%s=%s" % kv for kv in sorted(extra.items()))))
        print(name, {k: round(float(v), 5) for k, v in info.items()})
    out["cases"] = np.array(names)
    path = os.path.join(mg.GOLD, "mat_trainer_cases.npz")
    np.savez_compressed(path, **out)
    print("mat_trainer_cases.npz: %d arrays, %d KiB" % (len(out), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
