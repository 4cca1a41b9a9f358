#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/separated_cases.{npz,json} by running the REFERENCE's
SeparatedReplayBuffer (onpolicy/utils/separated_buffer.py, imported from /root/reference through
oracle/ref_import.py) on seeded inputs: compute_returns in every flag combination and the three
samplers with and without the HAPPO factor.

    python oracle/make_golden_separated.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference)

from onpolicy.utils.separated_buffer import SeparatedReplayBuffer as RefSeparated  # noqa: E402  (reference)

FIELD_NAMES = mg.FIELD_NAMES + ["factor"]


def gen_returns(out, meta):
    cid = 0
    for (T, N) in [(7, 5), (25, 12)]:
        for flags in mg.FLAGSETS:
            for popart in ([False, True] if flags["use_valuenorm"] else [False]):
                rng = np.random.default_rng(4000 + cid)
                f = dict(flags)
                if popart:
                    f["use_valuenorm"], f["use_popart"] = False, True
                args = mg.make_args(episode_length=T, n_rollout_threads=N, **f)
                buf = RefSeparated(args, mg.Box((3,)), mg.Box((4,)), mg.Discrete(5))
                nv = mg.fill_buffer(buf, rng)
                vn = mg.updated_valuenorm(rng) if (f["use_valuenorm"] or f.get("use_popart")) else None
                key = "sret%03d_" % cid
                for name in ("rewards", "masks", "bad_masks", "active_masks"):
                    out[key + name] = getattr(buf, name).copy()
                out[key + "value_preds_in"] = buf.value_preds.copy()
                out[key + "next_value"] = nv.copy()
                buf.compute_returns(nv, vn)
                out[key + "returns"] = buf.returns.copy()
                if vn is not None:
                    out[key + "norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                  float(vn.debiasing_term)], dtype=np.float32)
                meta.append(dict(id=cid, T=T, N=N, use_popart=bool(f.get("use_popart", False)),
                                 **{k: v for k, v in f.items() if k != "use_popart"}))
                cid += 1


def gen_generators(out, meta):
    T, N, Do, Ds, na, H = 10, 6, 7, 11, 5, 8
    rng = np.random.default_rng(91)
    args = mg.make_args(episode_length=T, n_rollout_threads=N, hidden_size=H)
    buf = RefSeparated(args, mg.Box((Do,)), mg.Box((Ds,)), mg.Discrete(na))
    nv = mg.fill_buffer(buf, rng)
    buf.compute_returns(nv, mg.ref.ValueNorm(1))
    adv = rng.standard_normal(buf.rewards.shape).astype(np.float32)
    factor = rng.random((T, N, 1)).astype(np.float32) + 0.5
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
                 "masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        out["sgen_buf_" + name] = getattr(buf, name).copy()
    out["sgen_buf_advantages"] = adv
    out["sgen_buf_factor"] = factor
    cases = [("ff3", lambda: buf.feed_forward_generator(adv, 3)),
             ("ff7", lambda: buf.feed_forward_generator(adv, 7)),
             ("rec_L5", lambda: buf.recurrent_generator(adv, 2, 5)),
             ("rec_L4", lambda: buf.recurrent_generator(adv, 3, 4)),       # chunks straddle trajectories
             ("naive3", lambda: buf.naive_recurrent_generator(adv, 3))]
    for with_factor in (False, True):
        if with_factor:
            buf.update_factor(factor)
        for cname, fn in cases:
            cname = cname + ("_factor" if with_factor else "")
            torch.manual_seed(9)
            with mg.PermRecorder() as rec:
                batches = list(fn())
            assert len(rec.calls) == 1
            out["sgen_%s_perm" % cname] = rec.calls[0].astype(np.int64)
            for bi, sample in enumerate(batches):
                assert len(sample) == (13 if with_factor else 12)
                for fname, arr in zip(FIELD_NAMES, sample):
                    out["sgen_%s_b%d_%s" % (cname, bi, fname)] = np.asarray(arr, dtype=np.float32)
            meta.append(dict(case=cname, n_batches=len(batches), factor=with_factor))
    meta.append(dict(shape=dict(T=T, N=N, Do=Do, Ds=Ds, na=na, H=H)))


def main():
    out, rmeta, gmeta = {}, [], []
    gen_returns(out, rmeta)
    gen_generators(out, gmeta)
    np.savez_compressed(os.path.join(mg.GOLD, "separated_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "separated_cases.json"), "w") as f:
        json.dump(dict(returns=rmeta, generators=gmeta), f, indent=0)
    print("separated_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(mg.GOLD, "separated_cases.npz"))))


if __name__ == "__main__":
    main()
