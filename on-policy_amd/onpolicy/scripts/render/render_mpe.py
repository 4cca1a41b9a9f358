#!/usr/bin/env python
"""Roll out a saved MPE policy and record the frames -- flags and checks of the reference's
onpolicy/scripts/render/render_mpe.py (``--use_render`` and ``--model_dir`` required, one rollout thread, :66-68; the
runner's ``render`` does the work, :124).  The built-in simple_spread draws its frames in numpy (``--save_gifs``);
other scenarios render through an external env tree (MAPPO_ENVS_PATH).

    python -m onpolicy.scripts.render.render_mpe --env_name MPE --scenario_name simple_spread --num_agents 3 \
        --num_landmarks 3 --n_rollout_threads 1 --use_render --save_gifs --render_episodes 5 --model_dir <run>/models
"""
import sys

from onpolicy.config import get_config
from onpolicy.envs.env_wrappers import DummyVecEnv
from onpolicy.scripts.train import _launch
from onpolicy.scripts.train.train_mpe import parse_args


def make_render_env(all_args):
    from onpolicy.envs.mpe.MPE_env import MPEEnv

    def init_env():
        env = MPEEnv(all_args)
        env.seed(all_args.seed)
        return env
    return DummyVecEnv([init_env])


def main(args):
    all_args = parse_args(args, get_config())
    _launch.apply_algorithm_flags(all_args, ("rmappo", "mappo", "ippo", "happo", "hatrpo"))
    assert (all_args.share_policy is True and all_args.scenario_name == 'simple_speaker_listener') is False, (
        "The simple_speaker_listener scenario can not use shared policy. Please check the config.py.")
    assert all_args.use_render, ("u need to set use_render be True")
    assert not (all_args.model_dir is None or all_args.model_dir == ""), ("set model_dir first")
    assert all_args.n_rollout_threads == 1, ("only support to use 1 env to render.")
    device = _launch.device_of(all_args)
    run_dir = _launch.new_run_dir(all_args, all_args.scenario_name)
    _launch.seed_everything(all_args)
    envs = make_render_env(all_args)
    if all_args.share_policy and all_args.algorithm_name not in ("happo", "hatrpo"):
        from onpolicy.runner.shared.mpe_runner import MPERunner as Runner
    else:
        from onpolicy.runner.separated.mpe_runner import MPERunner as Runner
    runner = Runner({"all_args": all_args, "envs": envs, "eval_envs": None, "num_agents": all_args.num_agents,
                     "device": device, "run_dir": run_dir})
    runner.render()
    envs.close()
    return runner


if __name__ == "__main__":
    main(sys.argv[1:])
