#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/happo_cases.{npz,json} from the REFERENCE's HAPPO trainer
(onpolicy/algorithms/happo/happo_trainer.py), HAPPO_Policy and SeparatedReplayBuffer with a factor set:
seeded initial parameters, the permutations drawn during train(), the six train_info scalars, the final
parameters and normaliser statistics.

    python oracle/make_golden_happo.py
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

from onpolicy.utils.separated_buffer import SeparatedReplayBuffer as RefSeparated  # noqa: E402  (reference)
from onpolicy.algorithms.happo.happo_trainer import HAPPO as RefHAPPO  # noqa: E402
from onpolicy.algorithms.happo.policy import HAPPO_Policy as RefPolicy  # noqa: E402

CASES = {
    # defaults: ValueNorm that the trainer never feeds
    "mlp": dict(args=dict(algorithm_name="happo", hidden_size=16, layer_N=1, ppo_epoch=2, num_mini_batch=2),
                T=10, N=6, Do=7, Ds=11, na=5),
    # the self-updating stand-alone PopArt
    "mlp_popart": dict(args=dict(algorithm_name="happo", hidden_size=16, layer_N=1, ppo_epoch=1, num_mini_batch=2,
                                 use_popart=True, use_valuenorm=False, use_policy_active_masks=False),
                       T=8, N=4, Do=5, Ds=9, na=4),
    "mlp_nonorm": dict(args=dict(algorithm_name="happo", hidden_size=16, ppo_epoch=1, num_mini_batch=3,
                                 use_valuenorm=False, use_huber_loss=False),
                       T=6, N=5, Do=6, Ds=6, na=3),
    # recurrent policy on the separated buffer's chunk-major minibatches
    "gru": dict(args=dict(algorithm_name="happo", use_recurrent_policy=True, hidden_size=16, layer_N=1, ppo_epoch=2,
                          num_mini_batch=2, data_chunk_length=5),
                T=10, N=6, Do=7, Ds=11, na=6),
}
BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "available_actions", "rewards")


def main():
    out, meta = {}, {}
    for cname, spec in CASES.items():
        T, N, Do, Ds, na = (spec[k] for k in ("T", "N", "Do", "Ds", "na"))
        args = mg.make_args(episode_length=T, n_rollout_threads=N, **spec["args"])
        spaces = mg.Box((Do,)), mg.Box((Ds,)), mg.Discrete(na)
        torch.manual_seed(1)
        np.random.seed(1)
        policy = RefPolicy(args, *spaces)
        trainer = RefHAPPO(args, policy)
        key = "hap_%s_" % cname
        _sd(key + "init_actor.", policy.actor, out)
        _sd(key + "init_critic.", policy.critic, out)
        rng = np.random.default_rng(777)
        buf = RefSeparated(args, spaces[0], spaces[1], spaces[2])
        next_value = mg.fill_buffer(buf, rng)
        av = buf.available_actions[:-1]
        buf.actions[:] = (rng.random(av.shape) * av).argmax(-1)[..., None].astype(np.float32)
        factor = (rng.random((T, N, 1)) + 0.5).astype(np.float32)
        for name in BUF:
            out[key + "buf_" + name] = getattr(buf, name).copy()
        out[key + "next_value"] = next_value
        out[key + "factor"] = factor
        buf.compute_returns(next_value, trainer.value_normalizer)
        out[key + "returns"] = buf.returns.copy()
        buf.update_factor(factor)
        trainer.prep_training()
        torch.manual_seed(21)
        with PermRecorder() as rec:
            info = trainer.train(buf)
        info = {k: float(v) for k, v in info.items()}
        _sd(key + "final_actor.", policy.actor, out)
        _sd(key + "final_critic.", policy.critic, out)
        if trainer.value_normalizer is not None:
            vn = trainer.value_normalizer
            out[key + "final_norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                float(vn.debiasing_term)], dtype=np.float64)
        meta[cname] = dict(spec=spec, train_info=info, n_perms=len(rec.calls))
    np.savez_compressed(os.path.join(mg.GOLD, "happo_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "happo_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("happo_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "happo_cases.npz"))))
    print(json.dumps({k: v["train_info"] for k, v in meta.items()}, indent=1))


if __name__ == "__main__":
    main()
