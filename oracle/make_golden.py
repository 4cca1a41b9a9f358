#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by running the REFERENCE itself
(marlbenchmark/on-policy imported from /root/reference, see oracle/ref_import.py) on seeded inputs.

    python oracle/make_golden.py            # rewrites every fixture

The fixtures pin (a) the plain-C / numpy oracle and (b) the HIP path; they are committed because
/root/reference does not exist on the GPU box.  Everything here calls reference code only -- the
few lines that are not callable as a function (the R_MAPPO.train prologue,
onpolicy/algorithms/r_mappo/r_mappo.py:179-187) are executed verbatim on reference objects.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

ref = ref_import.load_reference()
Box, Discrete = ref.Box, ref.Discrete


def make_args(**kw):
    argv = []
    parser = ref.get_config()
    args = parser.parse_known_args(argv)[0]
    # the train scripts rewrite these from algorithm_name (scripts/train/train_mpe.py:68-80)
    args.use_recurrent_policy = False
    args.use_naive_recurrent_policy = False
    for k, v in kw.items():
        assert hasattr(args, k), k
        setattr(args, k, v)
    return args


def fill_buffer(buf, rng, p_mask=0.9, p_bad=0.9, p_active=0.8, p_avail=0.7):
    """Seeded synthetic trajectory (SURVEY.md section 8d distributions)."""
    f32 = np.float32
    buf.share_obs[:] = rng.standard_normal(buf.share_obs.shape).astype(f32)
    buf.obs[:] = rng.standard_normal(buf.obs.shape).astype(f32)
    buf.rnn_states[:] = rng.standard_normal(buf.rnn_states.shape).astype(f32)
    buf.rnn_states_critic[:] = rng.standard_normal(buf.rnn_states_critic.shape).astype(f32)
    buf.rewards[:] = rng.standard_normal(buf.rewards.shape).astype(f32)
    buf.value_preds[:-1] = rng.standard_normal(buf.value_preds[:-1].shape).astype(f32)
    buf.masks[:] = (rng.random(buf.masks.shape) < p_mask).astype(f32)
    buf.bad_masks[:] = (rng.random(buf.bad_masks.shape) < p_bad).astype(f32)
    buf.active_masks[:] = (rng.random(buf.active_masks.shape) < p_active).astype(f32)
    if buf.available_actions is not None:
        av = (rng.random(buf.available_actions.shape) < p_avail).astype(f32)
        av[..., 0] = 1.0
        buf.available_actions[:] = av
        na = av.shape[-1]
        buf.actions[:] = rng.integers(0, na, size=buf.actions.shape).astype(f32)
        buf.action_log_probs[:] = f32(-np.log(na))
    next_value = rng.standard_normal(buf.value_preds[-1].shape).astype(f32)
    return next_value


def updated_valuenorm(rng):
    vn = ref.ValueNorm(1)
    for _ in range(3):
        vn.update(torch.from_numpy((rng.standard_normal((64, 1)) * 3.0 + 1.5).astype(np.float32)))
    return vn


# ----------------------------------------------------------------------------- KATs
def gen_kats(out):
    """The five known-answer vectors of SURVEY.md section 8c."""
    cases = {"A": dict(norm="fresh"), "B": dict(norm="updated"),
             "C": dict(use_valuenorm=False), "D": dict(use_valuenorm=False, use_proper_time_limits=True),
             "E": dict(use_valuenorm=False, use_gae=False)}
    for name, kw in cases.items():
        norm = kw.pop("norm", None)
        args = make_args(episode_length=4, n_rollout_threads=1, **kw)
        buf = ref.SharedReplayBuffer(args, 1, Box((3,)), Box((3,)), Discrete(5))
        buf.rewards[:, 0, 0, 0] = [1, 2, 3, -1]
        buf.value_preds[:4, 0, 0, 0] = [0.5, 0.4, 0.3, 0.25]
        buf.masks[:, 0, 0, 0] = [1, 1, 0, 1, 1]
        buf.bad_masks[:, 0, 0, 0] = [1, 1, 1, 1, 0]
        vn = None
        if norm is not None:
            vn = ref.ValueNorm(1)
            if norm == "updated":
                vn.update(np.array([[1], [2], [3], [6]], dtype=np.float32))
        buf.compute_returns(np.array([[[0.2]]], dtype=np.float32), vn)
        out["kat_%s_returns" % name] = buf.returns[:, 0, 0, 0].copy()
        if vn is not None:
            out["kat_%s_norm" % name] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                  float(vn.debiasing_term)], dtype=np.float32)


# -------------------------------------------------------- compute_returns flag matrix
SHAPES = [(25, 8, 3), (7, 3, 5), (33, 4, 4), (1, 1, 1), (70, 20, 4)]
FLAGSETS = [
    dict(use_gae=True, use_proper_time_limits=False, use_valuenorm=True),
    dict(use_gae=True, use_proper_time_limits=False, use_valuenorm=False),
    dict(use_gae=True, use_proper_time_limits=True, use_valuenorm=True),
    dict(use_gae=True, use_proper_time_limits=True, use_valuenorm=False),
    dict(use_gae=False, use_proper_time_limits=True, use_valuenorm=True),
    dict(use_gae=False, use_proper_time_limits=True, use_valuenorm=False),
    dict(use_gae=False, use_proper_time_limits=False, use_valuenorm=True),
]


def gen_returns(out, meta):
    cid = 0
    for (T, N, A) in SHAPES:
        for fi, flags in enumerate(FLAGSETS):
            for norm_state in (["fresh", "updated"] if flags["use_valuenorm"] else ["none"]):
                rng = np.random.default_rng(1000 + cid)
                args = make_args(episode_length=T, n_rollout_threads=N, **flags)
                buf = ref.SharedReplayBuffer(args, A, Box((3,)), Box((4,)), Discrete(5))
                nv = fill_buffer(buf, rng)
                vn = None
                if norm_state == "fresh":
                    vn = ref.ValueNorm(1)
                elif norm_state == "updated":
                    vn = updated_valuenorm(rng)
                key = "ret%03d_" % cid
                out[key + "rewards"] = buf.rewards.copy()
                out[key + "value_preds_in"] = buf.value_preds.copy()
                out[key + "masks"] = buf.masks.copy()
                out[key + "bad_masks"] = buf.bad_masks.copy()
                out[key + "active_masks"] = buf.active_masks.copy()
                out[key + "next_value"] = nv.copy()
                buf.compute_returns(nv, vn)
                out[key + "returns"] = buf.returns.copy()
                out[key + "value_preds_out"] = buf.value_preds.copy()
                # R_MAPPO.train prologue, r_mappo.py:179-187, verbatim on reference objects
                if vn is not None:
                    advantages = buf.returns[:-1] - vn.denormalize(buf.value_preds[:-1])
                else:
                    advantages = buf.returns[:-1] - buf.value_preds[:-1]
                advantages_copy = advantages.copy()
                advantages_copy[buf.active_masks[:-1] == 0.0] = np.nan
                mean_advantages = np.nanmean(advantages_copy)
                std_advantages = np.nanstd(advantages_copy)
                normed = (advantages - mean_advantages) / (std_advantages + 1e-5)
                out[key + "advantages"] = advantages.astype(np.float32)
                out[key + "adv_mean_std"] = np.array([mean_advantages, std_advantages], dtype=np.float32)
                out[key + "advantages_normed"] = normed.astype(np.float32)
                if vn is not None:
                    out[key + "norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                  float(vn.debiasing_term)], dtype=np.float32)
                meta.append(dict(id=cid, T=T, N=N, A=A, norm=norm_state, **flags))
                cid += 1


# --------------------------------------------------------------------------- generators
class PermRecorder(object):
    def __init__(self):
        self.orig = torch.randperm
        self.calls = []

    def __enter__(self):
        def rec(*a, **k):
            p = self.orig(*a, **k)
            self.calls.append(p.numpy().copy())
            return p
        torch.randperm = rec
        return self

    def __exit__(self, *exc):
        torch.randperm = self.orig


FIELD_NAMES = ["share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds",
               "returns", "masks", "active_masks", "old_action_log_probs", "adv_targ",
               "available_actions"]


def gen_generators(out, meta):
    T, N, A, Do, Ds, na, H = 10, 4, 3, 7, 11, 5, 8
    rng = np.random.default_rng(77)
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=H)
    buf = ref.SharedReplayBuffer(args, A, Box((Do,)), Box((Ds,)), Discrete(na))
    nv = fill_buffer(buf, rng)
    buf.compute_returns(nv, ref.ValueNorm(1))
    adv = rng.standard_normal(buf.advantages.shape).astype(np.float32)
    for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds",
                 "returns", "masks", "active_masks", "action_log_probs", "available_actions", "rewards"):
        out["gen_buf_" + name] = getattr(buf, name).copy()
    out["gen_buf_advantages"] = adv
    cases = [("ff2", lambda: buf.feed_forward_generator(adv, 2)),
             ("ff7", lambda: buf.feed_forward_generator(adv, 7)),           # 120 // 7: tail dropped
             ("rec_L5", lambda: buf.recurrent_generator(adv, 2, 5)),        # T % L == 0
             ("rec_L4", lambda: buf.recurrent_generator(adv, 3, 4)),        # chunks straddle (n, a)
             ("naive3", lambda: buf.naive_recurrent_generator(adv, 3))]
    for cname, fn in cases:
        torch.manual_seed(5)
        with PermRecorder() as rec:
            batches = list(fn())
        assert len(rec.calls) == 1
        out["gen_%s_perm" % cname] = rec.calls[0].astype(np.int64)
        for bi, sample in enumerate(batches):
            for fname, arr in zip(FIELD_NAMES, sample):
                out["gen_%s_b%d_%s" % (cname, bi, fname)] = np.asarray(arr, dtype=np.float32)
        meta.append(dict(case=cname, n_batches=len(batches)))
    meta.append(dict(shape=dict(T=T, N=N, A=A, Do=Do, Ds=Ds, na=na, H=H)))


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--only-mid" in sys.argv:        # just trainer_mid_cases (>= 10^5 rows at the north-star flags)
        import make_golden_trainer as mgt
        mgt.generate(ref, make_args, fill_buffer, GOLD, mgt.CASES_MID, "trainer_mid_cases", with_grads=True)
        return
    if "--only-cfg" in sys.argv:        # just trainer_cfg_cases (BASELINE configs[3] / [4] layer shapes)
        import make_golden_trainer as mgt
        mgt.generate(ref, make_args, fill_buffer, GOLD, mgt.CASES_CFG, "trainer_cfg_cases", with_grads=True)
        return
    kats = {}
    gen_kats(kats)
    np.savez_compressed(os.path.join(GOLD, "kat_returns.npz"), **kats)

    rets, meta = {}, []
    gen_returns(rets, meta)
    np.savez_compressed(os.path.join(GOLD, "returns_cases.npz"), **rets)
    with open(os.path.join(GOLD, "returns_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)

    gens, gmeta = {}, []
    gen_generators(gens, gmeta)
    np.savez_compressed(os.path.join(GOLD, "generator_cases.npz"), **gens)
    with open(os.path.join(GOLD, "generator_cases.json"), "w") as f:
        json.dump(gmeta, f, indent=0)

    if "--trainer" in sys.argv or True:
        import make_golden_trainer
        make_golden_trainer.main(ref, make_args, fill_buffer, GOLD)
    print("fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(GOLD, fn))))


if __name__ == "__main__":
    main()
