"""Command-line contract of the training scripts.

``get_config()`` returns an argparse parser with exactly the flag names, types, defaults and
store_true / store_false polarities of the reference's onpolicy/config.py (get_config :4, flags
:160-305) -- the shipped shell recipes (onpolicy/scripts/train_*_scripts/*.sh) must parse
unchanged.  Mind the inverted flags: passing ``--use_ReLU``, ``--use_valuenorm``,
``--use_value_active_masks`` ... turns them OFF (store_false).

The flag table below is data, not a transcription of argparse calls; flags added by this
implementation sit at the end and default to the reference's behaviour.
"""
import argparse

ON = "store_true"     # default False, passing the flag enables
OFF = "store_false"   # default True, passing the flag disables

# (name, kind, default[, choices]) ; kind is a type or ON / OFF
_FLAGS = [
    # prepare
    ("algorithm_name", str, "mappo", ["rmappo", "mappo", "happo", "hatrpo", "mat", "mat_dec"]),
    ("experiment_name", str, "check"),
    ("seed", int, 1),
    ("cuda", OFF, True),
    ("cuda_deterministic", OFF, True),
    ("n_training_threads", int, 1),
    ("n_rollout_threads", int, 32),
    ("n_eval_rollout_threads", int, 1),
    ("n_render_rollout_threads", int, 1),
    ("num_env_steps", int, 10e6),
    ("user_name", str, "marl"),
    ("use_wandb", OFF, True),
    # env
    ("env_name", str, "StarCraft2"),
    ("use_obs_instead_of_state", ON, False),
    # replay buffer
    ("episode_length", int, 200),
    # network
    ("share_policy", OFF, True),
    ("use_centralized_V", OFF, True),
    ("stacked_frames", int, 1),
    ("use_stacked_frames", ON, False),
    ("hidden_size", int, 64),
    ("layer_N", int, 1),
    ("use_ReLU", OFF, True),
    ("use_popart", ON, False),
    ("use_valuenorm", OFF, True),
    ("use_feature_normalization", OFF, True),
    ("use_orthogonal", OFF, True),
    ("gain", float, 0.01),
    # recurrent
    ("use_naive_recurrent_policy", ON, False),
    ("use_recurrent_policy", OFF, True),
    ("recurrent_N", int, 1),
    ("data_chunk_length", int, 10),
    # optimizer
    ("lr", float, 5e-4),
    ("critic_lr", float, 5e-4),
    ("opti_eps", float, 1e-5),
    ("weight_decay", float, 0),
    # trpo
    ("kl_threshold", float, 0.01),
    ("ls_step", int, 10),
    ("accept_ratio", float, 0.5),
    # ppo
    ("ppo_epoch", int, 15),
    ("use_clipped_value_loss", OFF, True),
    ("clip_param", float, 0.2),
    ("num_mini_batch", int, 1),
    ("entropy_coef", float, 0.01),
    ("value_loss_coef", float, 1),
    ("use_max_grad_norm", OFF, True),
    ("max_grad_norm", float, 10.0),
    ("use_gae", OFF, True),
    ("gamma", float, 0.99),
    ("gae_lambda", float, 0.95),
    ("use_proper_time_limits", ON, False),
    ("use_huber_loss", OFF, True),
    ("use_value_active_masks", OFF, True),
    ("use_policy_active_masks", OFF, True),
    ("huber_delta", float, 10.0),
    # run / save / log / eval / render
    ("use_linear_lr_decay", ON, False),
    ("save_interval", int, 1),
    ("log_interval", int, 5),
    ("use_eval", ON, False),
    ("eval_interval", int, 25),
    ("eval_episodes", int, 32),
    ("save_gifs", ON, False),
    ("use_render", ON, False),
    ("render_episodes", int, 5),
    ("ifi", float, 0.1),
    ("model_dir", str, None),
    # transformer (MAT) flags: parsed for script compatibility, unused by this path
    ("encode_state", ON, False),
    ("n_block", int, 1),
    ("n_embd", int, 64),
    ("n_head", int, 1),
    ("dec_actor", ON, False),
    ("share_actor", ON, False),
]

_LIST_FLAGS = [("train_maps", str), ("eval_maps", str)]

# Added by the MI355X implementation (all default to reference behaviour)
_NEW_FLAGS = [
    # where the rollout buffer lives; None = cuda:<LOCAL_RANK> (the buffer is HBM-resident)
    ("buffer_device", str, None),
    # 'device': minibatch permutations from the GPU generator (fast);
    # 'host': torch.randperm on the CPU generator, bit-identical index streams to the reference
    ("sampler_rng", str, "device", ["device", "host"]),
    # GAE returns are bit-identical to the reference's numpy loop for every buffer shape (the default since round 6).
    # --gae_scan (or MAPPO_GAE_SCAN=1) lets narrow buffers, 2048 <= N * A < 16384, take the time-parallel scan instead, which
    # agrees with the reference to ~1e-6 relative and is 2-3 x faster on a launch that is < 0.2 % of an update step
    # (include/mappo_hip.h K1).  --gae_exact is kept for command lines written against earlier rounds (it wins over --gae_scan).
    ("gae_exact", ON, False),
    ("gae_scan", ON, False),
    # how the hidden-64 kernels (K9 trunk, K12 GRU) form their float32 matrix products (include/mappo_hip.h MAPPO_ARITH_*):
    # 'six_term' = six bf16 x bf16 terms of the operands' exact three-way splits on the bf16 matrix cores, float32
    # accumulation (error of the float32 MFMA chain's order); 'f32_mfma' = the float32 matrix instruction.  None = the
    # process default (environment variable MAPPO_MATRIX_ARITHMETIC, else six_term).  A per-policy choice.
    ("matrix_arithmetic", str, None, ["six_term", "f32_mfma"]),
]


def get_config():
    parser = argparse.ArgumentParser(description="onpolicy",
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    for spec in _FLAGS + _NEW_FLAGS:
        name, kind, default = spec[0], spec[1], spec[2]
        if kind in (ON, OFF):
            parser.add_argument("--" + name, action=kind, default=default)
        else:
            extra = {"choices": spec[3]} if len(spec) > 3 else {}
            parser.add_argument("--" + name, type=kind, default=default, **extra)
    for name, kind in _LIST_FLAGS:
        parser.add_argument("--" + name, type=kind, nargs="+", default=None)
    return parser
