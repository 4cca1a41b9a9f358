#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/mat_cases.{npz,json} from the REFERENCE's SharedReplayBuffer
built with algorithm_name "mat" / "mat_dec": the transformer branches of compute_returns
(onpolicy/utils/shared_buffer.py:222-232, :241-251; advantages stored by the scan) and
feed_forward_generator_transformer (:264-338).

    python oracle/make_golden_mat.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference)

ref = mg.ref


def gen_returns(out, meta):
    cid = 0
    for (T, N, A) in [(7, 5, 2), (12, 3, 3), (9, 4, 8), (6, 3, 9), (5, 2, 17)]:
        for flags in (dict(use_valuenorm=True), dict(use_valuenorm=False),
                      dict(use_valuenorm=True, use_proper_time_limits=True),    # falls through to the MAPPO branch
                      dict(use_valuenorm=False, use_gae=False)):
            if (flags.get("use_proper_time_limits") or flags.get("use_gae") is False) and A != 3:
                continue
            for norm_state in (["fresh", "updated"] if flags["use_valuenorm"] else ["none"]):
                rng = np.random.default_rng(7000 + cid)
                algo = "mat" if cid % 2 == 0 else "mat_dec"
                args = mg.make_args(episode_length=T, n_rollout_threads=N, algorithm_name=algo, **flags)
                buf = ref.SharedReplayBuffer(args, A, mg.Box((3,)), mg.Box((4,)), mg.Discrete(5))
                nv = mg.fill_buffer(buf, rng)
                vn = {"fresh": lambda: ref.ValueNorm(1), "updated": lambda: mg.updated_valuenorm(rng),
                      "none": lambda: None}[norm_state]()
                key = "mat%03d_" % cid
                for name in ("rewards", "masks", "bad_masks", "active_masks"):
                    out[key + name] = getattr(buf, name).copy()
                out[key + "value_preds_in"] = buf.value_preds.copy()
                out[key + "next_value"] = nv.copy()
                buf.compute_returns(nv, vn)
                out[key + "returns"] = buf.returns.copy()
                out[key + "advantages"] = buf.advantages.copy()
                if vn is not None:
                    out[key + "norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                  float(vn.debiasing_term)], dtype=np.float32)
                full = dict(use_gae=True, use_proper_time_limits=False)
                full.update(flags)
                meta.append(dict(id=cid, T=T, N=N, A=A, algo=algo, norm=norm_state, **full))
                cid += 1


def gen_generators(out, meta):
    T, N, A, Do, Ds, na, H = 6, 5, 3, 7, 11, 5, 8
    rng = np.random.default_rng(313)
    args = mg.make_args(episode_length=T, n_rollout_threads=N, hidden_size=H, algorithm_name="mat")
    buf = ref.SharedReplayBuffer(args, A, mg.Box((Do,)), mg.Box((Ds,)), mg.Discrete(na))
    nv = mg.fill_buffer(buf, rng)
    buf.compute_returns(nv, ref.ValueNorm(1))
    adv = rng.standard_normal(buf.advantages.shape).astype(np.float32)
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns",
                 "masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        out["mgen_buf_" + name] = getattr(buf, name).copy()
    out["mgen_buf_advantages"] = adv
    for cname, nmb in (("tf1", 1), ("tf2", 2), ("tf7", 7)):
        torch.manual_seed(4)
        with mg.PermRecorder() as rec:
            batches = list(buf.feed_forward_generator_transformer(adv, nmb))
        assert len(rec.calls) == 1
        out["mgen_%s_perm" % cname] = rec.calls[0].astype(np.int64)
        for bi, sample in enumerate(batches):
            for fname, arr in zip(mg.FIELD_NAMES, sample):
                out["mgen_%s_b%d_%s" % (cname, bi, fname)] = np.asarray(arr, dtype=np.float32)
        meta.append(dict(case=cname, n_batches=len(batches), num_mini_batch=nmb))
    meta.append(dict(shape=dict(T=T, N=N, A=A, Do=Do, Ds=Ds, na=na, H=H)))


def main():
    out, rmeta, gmeta = {}, [], []
    gen_returns(out, rmeta)
    gen_generators(out, gmeta)
    np.savez_compressed(os.path.join(mg.GOLD, "mat_cases.npz"), **out)
    with open(os.path.join(mg.GOLD, "mat_cases.json"), "w") as f:
        json.dump(dict(returns=rmeta, generators=gmeta), f, indent=0)
    print("mat_cases.npz: %d arrays, %d return cases, %d B" % (len(out), len(rmeta),
                                                                os.path.getsize(os.path.join(mg.GOLD, "mat_cases.npz"))))


if __name__ == "__main__":
    main()
