#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/hatrpo_cases.{npz,json} from the REFERENCE's HATRPO trainer
(onpolicy/algorithms/hatrpo/hatrpo_trainer.py), HATRPO_Policy and SeparatedReplayBuffer with a factor set: seeded
initial parameters, the seven train_info scalars, the final parameters and normaliser statistics of one train() call;
plus the six outputs of HATRPO_Policy.evaluate_actions on one minibatch.

    python oracle/make_golden_hatrpo.py
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
from onpolicy.algorithms.hatrpo.hatrpo_trainer import HATRPO as RefHATRPO  # noqa: E402
from onpolicy.algorithms.hatrpo.policy import HATRPO_Policy as RefPolicy  # noqa: E402

CASES = {
    "mlp": dict(args=dict(algorithm_name="hatrpo", hidden_size=16, layer_N=1, num_mini_batch=2),
                T=10, N=6, Do=7, Ds=11, act=("Discrete", 5)),
    "mlp_popart": dict(args=dict(algorithm_name="hatrpo", hidden_size=16, layer_N=1, num_mini_batch=1, use_popart=True,
                                 use_valuenorm=False, use_policy_active_masks=False, kl_threshold=0.005),
                       T=8, N=4, Do=5, Ds=9, act=("Discrete", 4)),
    "mlp_nonorm": dict(args=dict(algorithm_name="hatrpo", hidden_size=16, num_mini_batch=3, use_valuenorm=False,
                                 use_huber_loss=False, use_max_grad_norm=False, accept_ratio=0.1, ls_step=4),
                       T=6, N=5, Do=6, Ds=6, act=("Discrete", 3)),
    "gru": dict(args=dict(algorithm_name="hatrpo", use_recurrent_policy=True, hidden_size=16, layer_N=1,
                          num_mini_batch=2, data_chunk_length=5),
                T=10, N=6, Do=7, Ds=11, act=("Discrete", 6)),
    # (no Box case: the reference's evaluate_actions_trpo passes available_actions to the Gaussian head, whose forward
    #  takes features only -- act.py:219 vs distributions.py:84 -- and raises TypeError)
    # a KL ball so small that no backtracking step is accepted: the actor must come back unchanged
    "rejected": dict(args=dict(algorithm_name="hatrpo", hidden_size=16, layer_N=1, num_mini_batch=1, accept_ratio=50.0,
                               ls_step=3),
                     T=6, N=4, Do=5, Ds=7, act=("Discrete", 4)),
}
BUF = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "masks", "bad_masks",
       "active_masks", "action_log_probs", "available_actions", "rewards")
KEYS = ('value_loss', 'kl', 'dist_entropy', 'loss_improve', 'expected_improve', 'critic_grad_norm', 'ratio')


def main():
    out, meta = {}, {}
    for cname, spec in CASES.items():
        T, N, Do, Ds = (spec[k] for k in ("T", "N", "Do", "Ds"))
        kind, k = spec["act"]
        args = mg.make_args(episode_length=T, n_rollout_threads=N, **spec["args"])
        act_space = mg.Discrete(k) if kind == "Discrete" else mg.Box((k,))
        spaces = mg.Box((Do,)), mg.Box((Ds,)), act_space
        torch.manual_seed(1)
        np.random.seed(1)
        policy = RefPolicy(args, *spaces)
        trainer = RefHATRPO(args, policy)
        key = "hat_%s_" % cname
        _sd(key + "init_actor.", policy.actor, out)
        _sd(key + "init_critic.", policy.critic, out)
        rng = np.random.default_rng(778)
        buf = RefSeparated(args, spaces[0], spaces[1], spaces[2])
        next_value = mg.fill_buffer(buf, rng)
        if kind == "Discrete":
            av = buf.available_actions[:-1]
            buf.actions[:] = (rng.random(av.shape) * av).argmax(-1)[..., None].astype(np.float32)
        else:
            buf.actions[:] = rng.standard_normal(buf.actions.shape).astype(np.float32) * 0.5
            buf.action_log_probs[:] = (rng.standard_normal(buf.action_log_probs.shape) * 0.1 - 1.0).astype(np.float32)
        factor = (rng.random((T, N, 1)) + 0.5).astype(np.float32)
        for name in BUF:
            arr = getattr(buf, name)
            if arr is not None:
                out[key + "buf_" + name] = arr.copy()
        out[key + "next_value"], out[key + "factor"] = next_value, factor
        # evaluate_actions: the 6-tuple on the rows of the first two steps
        flat = lambda a: a.reshape(-1, *a.shape[2:])      # noqa: E731
        with torch.no_grad():
            ev = policy.evaluate_actions(flat(buf.share_obs[:2]), flat(buf.obs[:2]), flat(buf.rnn_states[0:1]),
                                         flat(buf.rnn_states_critic[0:1]), flat(buf.actions[:2]), flat(buf.masks[:2]),
                                         None if buf.available_actions is None else flat(buf.available_actions[:2]),
                                         torch.from_numpy(flat(buf.active_masks[:2])))
        for name, t in zip(("values", "logp", "entropy", "mean", "std", "logits"), ev):
            if t is not None:
                out[key + "eval_" + name] = t.numpy()
        buf.compute_returns(next_value, trainer.value_normalizer)
        out[key + "returns"] = buf.returns.copy()
        buf.update_factor(factor)
        trainer.prep_training()
        torch.manual_seed(21)
        with PermRecorder() as rec:
            info = trainer.train(buf)
        info = {k: float(np.asarray(info[k].detach() if torch.is_tensor(info[k]) else info[k]).reshape(-1)[0]) for k in KEYS}
        _sd(key + "final_actor.", policy.actor, out)
        _sd(key + "final_critic.", policy.critic, out)
        if trainer.value_normalizer is not None:
            vn = trainer.value_normalizer
            out[key + "final_norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                float(vn.debiasing_term)], dtype=np.float64)
        moved = max(float(np.abs(out[key + "final_actor." + n] - out[key + "init_actor." + n]).max())
                    for n in policy.actor.state_dict())
        meta[cname] = dict(spec=spec, train_info=info, n_perms=len(rec.calls), actor_moved=moved)
        print(cname, "actor moved %.3g" % moved, {k: round(v, 5) for k, v in info.items()})
    np.savez_compressed(os.path.join(mg.GOLD, "hatrpo_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "hatrpo_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("hatrpo_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "hatrpo_cases.npz"))))


if __name__ == "__main__":
    main()
