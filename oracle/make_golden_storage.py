#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/storage_cases.npz from the REFERENCE's SharedReplayBuffer and
SeparatedReplayBuffer: seeded per-step data pushed through insert / chooseinsert for more than one rollout with
after_update / chooseafter_update at the wrap -- the final contents of every field pin the row conventions
(observation-like fields at step + 1 vs step, what the wrap copies) of shared_buffer.py:90-177 and
separated_buffer.py:65-120.

    python oracle/make_golden_storage.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference)
from onpolicy.utils.separated_buffer import SeparatedReplayBuffer as RefSeparated  # noqa: E402  (reference)

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
          "rewards", "masks", "bad_masks", "active_masks", "available_actions")
T, N, A, Do, Ds, na, H = 5, 4, 3, 6, 18, 5, 8


def step_data(rng, lead):
    f = lambda *s: rng.standard_normal(lead + s).astype(np.float32)
    return dict(share_obs=f(Ds), obs=f(Do), rnn_states=f(1, H), rnn_states_critic=f(1, H), actions=f(1),
                action_log_probs=f(1), value_preds=f(1), rewards=f(1), masks=f(1), bad_masks=f(1),
                active_masks=f(1), available_actions=f(na))


def main():
    out = {}
    args = mg.make_args(episode_length=T, n_rollout_threads=N, hidden_size=H)
    for kind in ("shared", "separated"):
        for mode in ("insert", "chooseinsert"):
            if kind == "shared":
                buf = mg.ref.SharedReplayBuffer(args, A, mg.Box((Do,)), mg.Box((Ds,)), mg.Discrete(na))
                lead = (N, A)
            else:
                buf = RefSeparated(args, mg.Box((Do,)), mg.Box((Ds,)), mg.Discrete(na))
                lead = (N,)
            rng = np.random.default_rng(2024)         # the tests regenerate the same stream
            for step in range(2 * T + 2):
                d = step_data(rng, lead)
                order = (d["share_obs"], d["obs"], d["rnn_states"], d["rnn_states_critic"], d["actions"],
                         d["action_log_probs"], d["value_preds"], d["rewards"], d["masks"], d["bad_masks"],
                         d["active_masks"], d["available_actions"])
                getattr(buf, mode)(*order)
                if buf.step == 0:
                    (buf.after_update if mode == "insert" else buf.chooseafter_update)()
            for name in FIELDS:
                out["%s_%s_%s" % (kind, mode, name)] = getattr(buf, name).copy()
            out["%s_%s_step" % (kind, mode)] = np.array(buf.step)
    out["dims"] = np.array([T, N, A, Do, Ds, na, H])
    np.savez_compressed(os.path.join(mg.GOLD, "storage_cases.npz"), **out)
    print("storage_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "storage_cases.npz"))))


if __name__ == "__main__":
    main()
