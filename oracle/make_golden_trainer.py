"""TEST INFRASTRUCTURE.  Trainer / network fixtures generated from the REFERENCE's R_MAPPOPolicy and
R_MAPPO (called by oracle/make_golden.py).  For each case the fixture stores: the seeded initial
parameters, one evaluate_actions / get_actions call, the permutations the reference drew during
train(), the six train_info scalars, the parameters and ValueNorm statistics after train().
"""
import json
import os

import numpy as np
import torch

CASES = {
    # feed-forward MAPPO, flags of train_mpe_spread.sh (--use_ReLU passed => Tanh)
    "mlp": dict(args=dict(algorithm_name="mappo", hidden_size=16, layer_N=1, use_ReLU=False, ppo_epoch=2,
                          num_mini_batch=2, lr=7e-4, critic_lr=7e-4),
                T=10, N=4, A=3, Do=7, Ds=11, na=5),
    # default ReLU, one minibatch, active-mask / huber variations off
    "mlp_relu": dict(args=dict(algorithm_name="mappo", hidden_size=16, layer_N=2, ppo_epoch=1,
                               num_mini_batch=1, use_huber_loss=False, use_clipped_value_loss=False,
                               use_value_active_masks=False, use_policy_active_masks=False,
                               use_max_grad_norm=False),
                     T=6, N=3, A=2, Do=5, Ds=9, na=4),
    # recurrent MAPPO (GRU), chunked sampler
    "gru": dict(args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=16, layer_N=1,
                          ppo_epoch=2, num_mini_batch=2, data_chunk_length=5, gain=1.0),
                T=10, N=4, A=3, Do=7, Ds=11, na=6),
    # no value normaliser, proper time limits
    "mlp_nonorm": dict(args=dict(algorithm_name="mappo", hidden_size=16, ppo_epoch=1, num_mini_batch=2,
                                 use_valuenorm=False, use_proper_time_limits=True),
                       T=8, N=2, A=3, Do=6, Ds=6, na=3),
}


# Hidden 64: the width the fused trunk kernels (K9, on-policy_amd/csrc/mappo_mlp_impl.h) take -- every shipped MPE /
# SMAC configuration.  Reference modules on this route: algorithms/utils/mlp.py:6-58, algorithms/utils/act.py:44-60,
# algorithms/r_mappo/algorithm/r_actor_critic.py:147-175, driven by algorithms/r_mappo/r_mappo.py:91-169.
# Written to trainer_h64_cases.npz (>= 192 rows per case: more than one 128-row kernel tile).
CASES_H64 = {
    # the north-star flags (train_mpe_spread.sh: tanh, layer_N 1, one minibatch) at Do 48 / Ds 384, 8 agents
    "h64_ns": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2,
                             num_mini_batch=1, lr=7e-4, critic_lr=7e-4),
                   T=6, N=4, A=8, Do=48, Ds=384, na=5),
    # ReLU, three Linear blocks, two minibatches, simple_spread's own shapes (config 1 / 3)
    "h64_relu2": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=2, use_ReLU=True, ppo_epoch=2,
                                num_mini_batch=2),
                      T=10, N=8, A=3, Do=18, Ds=54, na=5),
    # no input LayerNorm: the kernels read the buffer rows as they are
    "h64_nofeat": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2,
                                 num_mini_batch=2, use_feature_normalization=False),
                       T=10, N=8, A=3, Do=18, Ds=54, na=5),
    # observation widths that are not a multiple of 4 floats (config 2: 30 / 150), mse loss without clipping
    "h64_odd": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2,
                              num_mini_batch=2, use_huber_loss=False, use_clipped_value_loss=False),
                    T=8, N=6, A=5, Do=30, Ds=150, na=5),
    # recurrent MAPPO: fused trunk in front of the GRU, chunked sampler, gain 1 (train_smac_MMM2.sh)
    "h64_gru": dict(args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=64, layer_N=1,
                              use_ReLU=False, ppo_epoch=2, num_mini_batch=2, data_chunk_length=5, gain=1.0),
                    T=10, N=8, A=3, Do=22, Ds=37, na=6),
    # a chunk length that does not divide T (chunks straddle trajectories, shared_buffer.py:554-566), ReLU
    "h64_gru_straddle": dict(args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=64, layer_N=1,
                                       use_ReLU=True, ppo_epoch=1, num_mini_batch=1, data_chunk_length=4),
                             T=10, N=6, A=4, Do=16, Ds=40, na=7),
}


# The device-sampler route (what bench.py times): K10's partition instead of torch.randperm, the whole-batch cache, the
# default (time-parallel) GAE mode on narrow buffers.  The reference runs with its torch.randperm replaced by the numpy
# restatement of K10 (oracle/k10_partition.py: same keys from the same CPU generator), so its minibatches are the SETS the
# device sampler makes; for one minibatch per epoch the set is the whole batch whatever the order, so no patch is needed.
# Written to trainer_dev_cases.npz.
CASES_DEV = {
    # two minibatches per epoch, feed-forward (h64_relu2's spec)
    "dev_relu2": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=2, use_ReLU=True, ppo_epoch=2,
                                num_mini_batch=2),
                      T=10, N=8, A=3, Do=18, Ds=54, na=5, k10=True),
    # four minibatches with a dropped tail (210 samples -> 4 x 52, two samples dropped), tanh
    "dev_tail": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2,
                               num_mini_batch=4),
                     T=10, N=7, A=3, Do=18, Ds=54, na=5, k10=True),
    # recurrent, two minibatches of chunks (h64_gru's spec: the SMAC route)
    "dev_gru": dict(args=dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=64, layer_N=1,
                              use_ReLU=False, ppo_epoch=2, num_mini_batch=2, data_chunk_length=5, gain=1.0),
                    T=10, N=8, A=3, Do=22, Ds=37, na=6, k10=True),
    # config-2 shapes wide enough for the time-parallel GAE scan (C = N * A = 2560 columns, T = 64), one minibatch.
    # The inputs are regenerated from the seed by the test (oracle/synth.py), only outputs are stored.
    "dev_scan_cfg2": dict(args=dict(algorithm_name="mappo", hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=3,
                                    num_mini_batch=1, lr=7e-4, critic_lr=7e-4),
                          T=64, N=512, A=5, Do=30, Ds=150, na=5, k10=False, regen=True),
}


# BASELINE.json configs[3] / configs[4] at their OWN layer shapes (VERDICT r4 "missing" #2): the observation / action widths of
# SMAC MMM2 (reference envs/starcraft2/StarCraft2_Env.py:1625-1677: obs 370, share_obs 435,
# 18 actions, 10 agents) with the flags of scripts/train_smac_scripts/train_smac_MMM2.sh:12-14 (rmappo, 2 minibatches,
# gain 1, --use_value_active_masks = store_false, chunk 10 and hidden 64 from config.py), and of Hanabi-Full with 5 players
# (obs 1285, share_obs 1385, 48 actions) with scripts/train_hanabi_forward.sh:15-17 (hidden 512, layer_N 2, ReLU,
# critic_lr 1e-3, entropy_coef 0.015, gain 0.01, one minibatch).  Few rows (the widths are the point), two epochs.
# Inputs are rebuilt from the seed by the tests (oracle/synth.py, digest in the fixture); tensors of more than 65 536
# elements are stored as every 8th element + their float64 sum / sum of squares (``subsample``), except the initial
# weights, which the tests start from and therefore need in full.  Written to trainer_cfg_cases.npz.
_CFG4 = dict(algorithm_name="rmappo", use_recurrent_policy=True, hidden_size=64, layer_N=1, ppo_epoch=2, num_mini_batch=2,
             data_chunk_length=10, gain=1.0, use_value_active_masks=False)
CASES_CFG = {
    "cfg4_shape": dict(args=_CFG4, T=20, N=6, A=10, Do=370, Ds=435, na=18, k10=False, regen=True, store_perms=True,
                       subsample=8),
    "cfg4_shape_dev": dict(args=_CFG4, T=20, N=6, A=10, Do=370, Ds=435, na=18, k10=True, regen=True, store_perms=True,
                           subsample=8),
    "cfg5_shape": dict(args=dict(algorithm_name="mappo", hidden_size=512, layer_N=2, use_ReLU=True, ppo_epoch=2,
                                 num_mini_batch=1, lr=7e-4, critic_lr=1e-3, entropy_coef=0.015, gain=0.01),
                       T=4, N=8, A=5, Do=1285, Ds=1385, na=48, k10=False, regen=True, store_perms=True, subsample=8),
}
# Mid-size cases (VERDICT r5 "weak" #1): >= 10^5 rows at the north-star flags, so that what only happens on large minibatches --
# persistent grids walking several tiles per wave, grid caps, multi-tile accumulation of the weight gradients, split
# reductions, the multi-workgroup GAE kernels -- is pinned by the REFERENCE and not only by float64 restatements.
#   mid_ns      train_mpe_spread.sh flags (mappo, tanh, hidden 64, layer_N 1, one minibatch, gain 0.01) at Do 48 / Ds 384,
#               8 agents, T = 100 x N = 256: 204 800 rows, 2 048 buffer columns;
#   mid_ns_rnn  the same shapes with the recurrent policy (chunk 10), T = 100 x N = 128: 102 400 rows = 10 240 chunks.
# One minibatch per epoch: the device sampler's single slice is the reference's batch as a set, no permutation patch needed.
# Inputs rebuilt from the seed (digest in the fixture); tensors above 65 536 elements stored subsampled (incl. the returns).
# ``first_grads``: also what the reference's FIRST ppo_update left in .grad.  That is the well-conditioned quantity at this size:
# under ValueNorm the first update's targets have the batch mean subtracted, so the first gradient of the value head's bias is a
# sum that cancels to rounding noise; Adam (eps 1e-5) turns that noise into a step whose size depends on its sign and magnitude,
# and everything after it -- the second update's gradients (0.2 % of the critic's), a handful of weights (5e-5) -- lands on
# one of two branches depending on the last bit of the inputs (measured on the device: tools/r06/probe_mid3.py).
# ``torch_threads`` = 1: at this size PyTorch's CPU reductions split over the threads, and the reference's own numbers move in
# their sixth digit with the thread count -- one thread makes the fixture regenerate bit for bit on any machine.
_MID = dict(hidden_size=64, layer_N=1, use_ReLU=False, ppo_epoch=2, num_mini_batch=1, lr=7e-4, critic_lr=7e-4, gain=0.01)
CASES_MID = {
    "mid_ns": dict(args=dict(algorithm_name="mappo", **_MID), T=100, N=256, A=8, Do=48, Ds=384, na=5, k10=False,
                   regen=True, subsample=8, first_grads=True, torch_threads=1),
    "mid_ns_rnn": dict(args=dict(algorithm_name="rmappo", use_recurrent_policy=True, data_chunk_length=10, **_MID),
                       T=100, N=128, A=8, Do=48, Ds=384, na=5, k10=False, regen=True, subsample=8, first_grads=True,
                       torch_threads=1),
}
SUBSAMPLE_ABOVE = 65536


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


def _store(out, name, arr, stride):
    """arr itself, or (large tensors of the ``subsample`` cases) every ``stride``-th element + [sum, sum of squares]."""
    arr = np.asarray(arr)
    if stride and arr.size > SUBSAMPLE_ABOVE:
        out[name] = arr.ravel()[::stride].copy()
        a64 = arr.astype(np.float64)
        out[name + "#moments"] = np.array([a64.sum(), (a64 * a64).sum()], dtype=np.float64)
    else:
        out[name] = arr.copy()


def _sd(prefix, module, out, stride=0):
    for k, v in module.state_dict().items():
        _store(out, prefix + k, v.detach().cpu().numpy(), stride)


def main(ref, make_args, fill_buffer, gold_dir):
    generate(ref, make_args, fill_buffer, gold_dir, CASES, "trainer_cases")
    generate(ref, make_args, fill_buffer, gold_dir, CASES_H64, "trainer_h64_cases", with_grads=True)
    generate(ref, make_args, fill_buffer, gold_dir, CASES_DEV, "trainer_dev_cases", with_grads=True)
    generate(ref, make_args, fill_buffer, gold_dir, CASES_CFG, "trainer_cfg_cases", with_grads=True)
    generate(ref, make_args, fill_buffer, gold_dir, CASES_MID, "trainer_mid_cases", with_grads=True)


def generate(ref, make_args, fill_buffer, gold_dir, cases, fname, with_grads=False):
    out, meta = {}, {}
    threads_before = torch.get_num_threads()
    for cname, spec in cases.items():
        torch.set_num_threads(int(spec.get("torch_threads", threads_before)))
        T, N, A, Do, Ds, na = (spec[k] for k in ("T", "N", "A", "Do", "Ds", "na"))
        args = make_args(episode_length=T, n_rollout_threads=N, **spec["args"])
        obs_space, cent_space, act_space = ref.Box((Do,)), ref.Box((Ds,)), ref.Discrete(na)
        torch.manual_seed(1)
        np.random.seed(1)
        policy = ref.R_MAPPOPolicy(args, obs_space, cent_space, act_space)
        trainer = ref.R_MAPPO(args, policy)
        key = "trn_%s_" % cname
        _sd(key + "init_actor.", policy.actor, out)
        _sd(key + "init_critic.", policy.critic, out)

        rng = np.random.default_rng(4242)
        buf = ref.SharedReplayBuffer(args, A, obs_space, cent_space, act_space)
        if spec.get("regen"):
            # large case: seeded inputs that the test rebuilds itself (oracle/synth.py); the fixture keeps a digest of them
            import synth
            rnn_hidden = spec["args"]["hidden_size"] if spec["args"].get("use_recurrent_policy") else 0
            arrays = synth.rollout(T, N, A, Do, Ds, na, seed=4242, rnn_hidden=rnn_hidden)
            next_value = arrays.pop("next_value")
            for name, arr in arrays.items():
                getattr(buf, name)[...] = arr
            out[key + "input_digest"] = synth.digest(arrays, next_value)
        else:
            next_value = fill_buffer(buf, rng)
            # valid action ids under the availability mask: pick the first available action at random
            av = buf.available_actions[:-1]
            pick = rng.random(av.shape) * av
            buf.actions[:] = pick.argmax(-1)[..., None].astype(np.float32)
            for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds",
                         "masks", "bad_masks", "active_masks", "action_log_probs", "available_actions",
                         "rewards"):
                out[key + "buf_" + name] = getattr(buf, name).copy()
        out[key + "next_value"] = next_value

        # one rollout-side and one update-side forward on step 0 of the buffer
        B = N * A
        flat = lambda x: x.reshape(B, *x.shape[2:])
        trainer.prep_rollout()
        torch.manual_seed(11)
        with torch.no_grad():
            values, actions, logp, h_a, h_c = policy.get_actions(
                flat(buf.share_obs[0]), flat(buf.obs[0]), flat(buf.rnn_states[0]),
                flat(buf.rnn_states_critic[0]), flat(buf.masks[0]), flat(buf.available_actions[0]))
            out[key + "act_values"] = values.numpy().copy()
            out[key + "act_actions"] = actions.numpy().astype(np.int64)
            out[key + "act_logp"] = logp.numpy().copy()
            if not spec.get("regen"):
                out[key + "act_h_actor"] = h_a.numpy().copy()
                out[key + "act_h_critic"] = h_c.numpy().copy()
            ev_values, ev_logp, ev_ent = policy.evaluate_actions(
                flat(buf.share_obs[0]), flat(buf.obs[0]), flat(buf.rnn_states[0]),
                flat(buf.rnn_states_critic[0]), flat(buf.actions[0]), flat(buf.masks[0]),
                flat(buf.available_actions[0]), flat(buf.active_masks[0]))
            out[key + "eval_values"] = ev_values.numpy().copy()
            out[key + "eval_logp"] = ev_logp.numpy().copy()
            out[key + "eval_entropy"] = np.array(float(ev_ent), dtype=np.float32)

        buf.compute_returns(next_value, trainer.value_normalizer)
        _store(out, key + "returns", buf.returns, int(spec.get("subsample", 0)))
        trainer.prep_training()
        torch.manual_seed(21)
        if spec.get("k10"):
            from k10_partition import RandpermAsK10
            recorder = RandpermAsK10(spec["args"]["num_mini_batch"])
        else:
            recorder = PermRecorder()
        first = {}
        if spec.get("first_grads"):
            inner = trainer.ppo_update

            def recording_update(*a, **k):
                res = inner(*a, **k)
                if not first:
                    for net, pre in ((policy.actor, "first_grad_actor."), (policy.critic, "first_grad_critic.")):
                        for kk, p in net.named_parameters():
                            first[pre + kk] = p.grad.detach().cpu().numpy().copy()
                return res
            trainer.ppo_update = recording_update
        with recorder as rec:
            info = trainer.train(buf)
        for kk, g in first.items():
            _store(out, key + kk, g, int(spec.get("subsample", 0)))
        for i, p in enumerate(rec.calls if (spec.get("store_perms") or not spec.get("regen")) else []):
            out[key + "perm%d" % i] = p.astype(np.int64)
        info = {k: float(v) for k, v in info.items()}
        stride = int(spec.get("subsample", 0))
        _sd(key + "final_actor.", policy.actor, out, stride)
        _sd(key + "final_critic.", policy.critic, out, stride)
        if with_grads:      # what the reference's last ppo_update left in .grad (after clip_grad_norm_, r_mappo.py:149,163)
            for net, pre in ((policy.actor, "last_grad_actor."), (policy.critic, "last_grad_critic.")):
                for k, p in net.named_parameters():
                    _store(out, key + pre + k, p.grad.detach().cpu().numpy(), stride)
        if trainer.value_normalizer is not None:
            vn = trainer.value_normalizer
            out[key + "final_norm"] = np.array([float(vn.running_mean), float(vn.running_mean_sq),
                                                float(vn.debiasing_term)], dtype=np.float64)
        meta[cname] = dict(spec=spec, train_info=info, n_perms=len(rec.calls))
    torch.set_num_threads(threads_before)
    np.savez_compressed(os.path.join(gold_dir, fname + ".npz"), **out)
    with open(os.path.join(gold_dir, fname + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
