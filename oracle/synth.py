"""TEST INFRASTRUCTURE.  Seeded synthetic rollouts that are too large to commit as fixtures: the generator script
(oracle/make_golden_trainer.py, which feeds them to the REFERENCE) and the device test rebuild the same arrays from the
seed, and the fixture carries a digest of them so that a drifting random stream fails loudly instead of silently
comparing different inputs.  Distributions: SURVEY.md section 8d (feed-forward policies: the RNN states stay zero).
"""
import numpy as np


def rollout(T, N, A, Do, Ds, na, seed, p_mask=0.96, p_active=0.9, p_avail=0.7, rnn_hidden=0, recurrent_N=1):
    """dict of float32 arrays in the reference buffer's shapes (+ "next_value" [N, A, 1]).  ``rnn_hidden`` > 0 (recurrent
    policies) adds N(0, 1) ``rnn_states`` / ``rnn_states_critic`` [T + 1, N, A, recurrent_N, rnn_hidden], drawn AFTER everything
    else so that the arrays of the feed-forward cases do not depend on it."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    out = {}
    out["share_obs"] = rng.standard_normal((T + 1, N, A, Ds), dtype=f32)
    out["obs"] = rng.standard_normal((T + 1, N, A, Do), dtype=f32)
    out["rewards"] = rng.standard_normal((T, N, A, 1), dtype=f32)
    vp = np.zeros((T + 1, N, A, 1), dtype=f32)
    vp[:-1] = rng.standard_normal((T, N, A, 1), dtype=f32)
    out["value_preds"] = vp
    out["masks"] = (rng.random((T + 1, N, A, 1)) < p_mask).astype(f32)
    out["bad_masks"] = np.ones((T + 1, N, A, 1), dtype=f32)
    out["active_masks"] = (rng.random((T + 1, N, A, 1)) < p_active).astype(f32)
    av = (rng.random((T + 1, N, A, na)) < p_avail).astype(f32)
    av[..., 0] = 1.0
    out["available_actions"] = av
    # a valid action under the availability mask
    pick = rng.random((T, N, A, na)) * av[:-1]
    out["actions"] = pick.argmax(-1)[..., None].astype(f32)
    out["action_log_probs"] = np.full((T, N, A, 1), -np.log(na), dtype=f32)
    out["next_value"] = rng.standard_normal((N, A, 1), dtype=f32)
    if rnn_hidden > 0:
        out["rnn_states"] = rng.standard_normal((T + 1, N, A, recurrent_N, rnn_hidden), dtype=f32)
        out["rnn_states_critic"] = rng.standard_normal((T + 1, N, A, recurrent_N, rnn_hidden), dtype=f32)
    return out


def digest(arrays, next_value):
    """float64 [2 * fields + 2]: per field (sorted by name) its sum and its sum of squares, then next_value's."""
    vals = []
    for name in sorted(arrays):
        a = np.asarray(arrays[name], dtype=np.float64)
        vals += [a.sum(), (a * a).sum()]
    nv = np.asarray(next_value, dtype=np.float64)
    vals += [nv.sum(), (nv * nv).sum()]
    return np.array(vals, dtype=np.float64)
