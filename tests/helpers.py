"""Shared test helpers: duck-typed spaces, reference-default args, buffer fillers."""

import numpy as np
import torch

from onpolicy.config import get_config


class Box(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class Discrete(object):
    def __init__(self, n):
        self.n = n


def make_args(**kw):
    """Defaults of onpolicy/config.py with the algorithm-name rewrite of the train scripts
    (reference scripts/train/train_mpe.py:68-80) applied for mappo."""
    args = get_config().parse_known_args([])[0]
    args.use_recurrent_policy = False
    args.use_naive_recurrent_policy = False
    for k, v in kw.items():
        assert hasattr(args, k), k
        setattr(args, k, v)
    return args


def fill_buffer_arrays(shapes, rng, na=None, p_mask=0.9, p_bad=0.9, p_active=0.8, p_avail=0.7):
    """Seeded synthetic trajectory with the distributions of SURVEY.md section 8d.
    shapes: dict name -> shape.  Returns dict of float32 arrays (+ next_value)."""
    f32 = np.float32
    out = {}
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "rewards"):
        out[name] = rng.standard_normal(shapes[name]).astype(f32)
    vp = np.zeros(shapes["value_preds"], dtype=f32)
    vp[:-1] = rng.standard_normal(vp[:-1].shape).astype(f32)
    out["value_preds"] = vp
    out["masks"] = (rng.random(shapes["masks"]) < p_mask).astype(f32)
    out["bad_masks"] = (rng.random(shapes["masks"]) < p_bad).astype(f32)
    out["active_masks"] = (rng.random(shapes["masks"]) < p_active).astype(f32)
    if na is not None:
        av = (rng.random(shapes["available_actions"]) < p_avail).astype(f32)
        av[..., 0] = 1.0
        out["available_actions"] = av
        out["actions"] = rng.integers(0, na, size=shapes["actions"]).astype(f32)
        out["action_log_probs"] = np.full(shapes["actions"], -np.log(na), dtype=f32)
    out["next_value"] = rng.standard_normal(shapes["value_preds"][1:]).astype(f32)
    return out


def buffer_shapes(T, N, A, Do, Ds, na, H, R=1):
    return dict(share_obs=(T + 1, N, A, Ds), obs=(T + 1, N, A, Do), rnn_states=(T + 1, N, A, R, H),
                rnn_states_critic=(T + 1, N, A, R, H), rewards=(T, N, A, 1), value_preds=(T + 1, N, A, 1),
                masks=(T + 1, N, A, 1), available_actions=(T + 1, N, A, na), actions=(T, N, A, 1))


def load_into(buf, arrays):
    """Copy a dict of arrays into a buffer object (numpy OracleBuffer or device SharedReplayBuffer)."""
    for name, arr in arrays.items():
        if name == "next_value" or not hasattr(buf, name):
            continue
        dst = getattr(buf, name)
        if dst is None:
            continue
        if torch.is_tensor(dst):
            if dst.stride()[0] == 0:      # lazily-allocated RNN state of a feed-forward buffer
                continue
            dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)))
        else:
            dst[...] = arr


def graph_replays(trainer):
    """Updates of ``trainer`` so far that were replays of a captured HIP graph (algorithms/r_mappo/update_graph.py)."""
    ug = getattr(trainer, "_update_graph", None)
    return 0 if ug is None else ug.replays


def assert_k9_carried_the_updates(trainer, n_fwd, n_bwd, updates, signatures=1):
    """Every update ran the fused trunk kernels, forward and backward, for both networks: eagerly (event pairs recorded:
    ``n_fwd`` / ``n_bwd`` launches) or as a replay of a graph captured from exactly such an update.  With the update graph on,
    the first update of a minibatch shape is the eager warm-up, all later ones replay."""
    import os
    replays = graph_replays(trainer)
    assert n_fwd == 2 * (updates - replays) and n_bwd == 2 * (updates - replays), (n_fwd, n_bwd, updates, replays)
    if os.environ.get("MAPPO_UPDATE_GRAPH", "1") != "0":
        assert replays == updates - signatures, (replays, updates)
    else:
        assert replays == 0
