#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/space_cases.{npz,json}: the REFERENCE's R_MAPPOPolicy / R_MAPPO on
non-Discrete action spaces (continuous Box, MultiDiscrete) -- seeded initial parameters, one
evaluate_actions call, compute_returns + train on a seeded buffer (actions drawn by the reference policy itself),
final parameters and train_info.  Pins the DiagGaussian / multi-head Categorical / Bernoulli heads
(onpolicy/algorithms/utils/act.py, distributions.py) and the stored action widths (utils/util.py:40-52).

    python oracle/make_golden_spaces.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference)
from make_golden_trainer import PermRecorder, _sd  # noqa: E402

ref = mg.ref


class MultiDiscrete(object):           # duck-typed like the reference's utils/multi_discrete.py (gym is absent)
    def __init__(self, pairs):
        self.low = np.array([p[0] for p in pairs])
        self.high = np.array([p[1] for p in pairs])
        self.shape = len(pairs)


class MultiBinary(object):
    def __init__(self, n):
        self.shape = (n,)


# (MultiBinary is not generated: the reference's ACTLayer passes available_actions to Bernoulli.forward, act.py:87,
# which takes no such argument -- its rollout path raises TypeError)
SPACES = {"box": lambda: mg.Box((3,)), "multidiscrete": lambda: MultiDiscrete([[0, 2], [0, 3]]),
          # image observations: CNNBase trunk (algorithms/utils/cnn.py) for actor and critic, Discrete head
          "cnn": lambda: mg.Discrete(4),
          # whole-trajectory sampler (--use_naive_recurrent_policy) and a 2-layer GRU with xavier init and no
          # input LayerNorm: flag combinations the main trainer fixtures do not reach
          "naive_gru": lambda: mg.Discrete(5), "gru2_xavier": lambda: mg.Discrete(5)}
EXTRA_ARGS = {"naive_gru": dict(use_naive_recurrent_policy=True, algorithm_name="rmappo"),
              "gru2_xavier": dict(use_recurrent_policy=True, recurrent_N=2, use_orthogonal=False,
                                  use_feature_normalization=False, data_chunk_length=4, algorithm_name="rmappo")}
BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "rewards")


def main():
    out, meta = {}, {}
    T, N, A, Do, Ds = 8, 4, 2, 6, 10
    for cname, make_space in SPACES.items():
        base_args = dict(hidden_size=16, layer_N=1, ppo_epoch=2, num_mini_batch=2, algorithm_name="mappo")
        base_args.update(EXTRA_ARGS.get(cname, {}))
        args = mg.make_args(episode_length=T, n_rollout_threads=N, **base_args)
        act_space = make_space()
        spaces = (mg.Box((Do,)), mg.Box((Ds,)), act_space) if cname != "cnn" else \
            (mg.Box((3, 9, 9)), mg.Box((3, 9, 9)), act_space)
        torch.manual_seed(1)
        np.random.seed(1)
        policy = ref.R_MAPPOPolicy(args, *spaces)
        trainer = ref.R_MAPPO(args, policy)
        key = "spc_%s_" % cname
        _sd(key + "init_actor.", policy.actor, out)
        _sd(key + "init_critic.", policy.critic, out)
        rng = np.random.default_rng(99)
        buf = ref.SharedReplayBuffer(args, A, *spaces)
        assert (buf.available_actions is None) == (cname in ("box", "multidiscrete"))
        next_value = mg.fill_buffer(buf, rng)
        # actions and their log-probs from the reference policy itself, stored the way the runners store them
        B = N * A
        flat = lambda x: x.reshape(B, *x.shape[2:])
        trainer.prep_rollout()
        torch.manual_seed(7)
        with torch.no_grad():
            for t in range(T):
                avail = None if buf.available_actions is None else flat(buf.available_actions[t])
                _, a, lp, _, _ = policy.get_actions(flat(buf.share_obs[t]), flat(buf.obs[t]), flat(buf.rnn_states[t]),
                                                    flat(buf.rnn_states_critic[t]), flat(buf.masks[t]), avail)
                buf.actions[t] = a.numpy().reshape(N, A, -1)
                buf.action_log_probs[t] = lp.numpy().reshape(N, A, -1)       # [., 1] broadcasts for Box
            ev = policy.evaluate_actions(flat(buf.share_obs[0]), flat(buf.obs[0]), flat(buf.rnn_states[0]),
                                         flat(buf.rnn_states_critic[0]), flat(buf.actions[0]), flat(buf.masks[0]),
                                         None if buf.available_actions is None else flat(buf.available_actions[0]),
                                         flat(buf.active_masks[0]))
        out[key + "eval_values"] = ev[0].numpy().copy()
        out[key + "eval_logp"] = ev[1].numpy().copy()
        out[key + "eval_entropy"] = np.array(float(ev[2]), dtype=np.float32)
        for name in BUF + (("available_actions",) if buf.available_actions is not None else ()):
            out[key + "buf_" + name] = getattr(buf, name).copy()
        out[key + "next_value"] = next_value
        buf.compute_returns(next_value, trainer.value_normalizer)
        trainer.prep_training()
        torch.manual_seed(21)
        with PermRecorder() as rec:
            info = trainer.train(buf)
        info = {k: float(v) for k, v in info.items()}
        _sd(key + "final_actor.", policy.actor, out)
        _sd(key + "final_critic.", policy.critic, out)
        meta[cname] = dict(T=T, N=N, A=A, Do=Do, Ds=Ds, act_width=int(buf.actions.shape[-1]), train_info=info,
                           args=base_args)
    np.savez_compressed(os.path.join(mg.GOLD, "space_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "space_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("space_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "space_cases.npz"))))
    print(json.dumps({k: v["train_info"] for k, v in meta.items()}, indent=1))


if __name__ == "__main__":
    main()
