"""Replays the seeded insert / chooseinsert stream of oracle/make_golden_storage.py into a buffer object."""
import numpy as np

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "action_log_probs", "value_preds",
          "rewards", "masks", "bad_masks", "active_masks", "available_actions")


def replay(buf, mode, lead, dims):
    T, N, A, Do, Ds, na, H = [int(x) for x in dims]
    rng = np.random.default_rng(2024)
    f = lambda *s: rng.standard_normal(lead + s).astype(np.float32)
    for step in range(2 * T + 2):
        d = dict(share_obs=f(Ds), obs=f(Do), rnn_states=f(1, H), rnn_states_critic=f(1, H), actions=f(1),
                 action_log_probs=f(1), value_preds=f(1), rewards=f(1), masks=f(1), bad_masks=f(1),
                 active_masks=f(1), available_actions=f(na))
        getattr(buf, mode)(d["share_obs"], d["obs"], d["rnn_states"], d["rnn_states_critic"], d["actions"],
                           d["action_log_probs"], d["value_preds"], d["rewards"], d["masks"], d["bad_masks"],
                           d["active_masks"], d["available_actions"])
        if buf.step == 0:
            (buf.after_update if mode == "insert" else buf.chooseafter_update)()
    return buf
