#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/mpe_spread_cases.npz by stepping the REFERENCE's multi-agent particle
environment (onpolicy/envs/mpe: core.py physics, environment.py step / spaces, scenarios/simple_spread.py reward and
observation) from seeded initial states with seeded actions.  gym and seaborn are not installed, so the two are
replaced by minimal stand-ins for the import only (spaces are containers; the colour palette is cosmetic).

    python oracle/make_golden_mpe.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _stub_modules():
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Space(object):
        pass
    gym.Env, gym.Space = Env, Space
    spaces = types.ModuleType("gym.spaces")

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete(Space):
        def __init__(self, n):
            self.n = n

    class Tuple(Space):
        def __init__(self, spaces):
            self.spaces = spaces
    spaces.Box, spaces.Discrete, spaces.Tuple = Box, Discrete, Tuple
    gym.spaces = spaces
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.EnvSpec = object
    envs.registration = reg
    gym.envs = envs
    sns = types.ModuleType("seaborn")
    sns.color_palette = lambda *a, **k: [(0.1, 0.1, 0.1)] * 64
    imp = types.ModuleType("imp")          # removed from Python 3.12; scenarios/__init__ only needs load_source
    import importlib.util

    def load_source(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    imp.load_source = load_source
    for name, mod in (("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs), ("gym.envs.registration", reg),
                      ("seaborn", sns)):
        sys.modules.setdefault(name, mod)
    if "imp" not in sys.modules:
        try:
            import imp as _real  # noqa: F401
        except Exception:
            sys.modules["imp"] = imp


def main():
    _stub_modules()
    ref_import.load_reference()
    # onpolicy/envs/__init__.py only initialises absl flags for SMAC (absl is not installed): register the
    # package without running it, like ref_import does for the top-level package
    envs = types.ModuleType("onpolicy.envs")
    envs.__path__ = [os.path.join(ref_import.REFERENCE_ROOT, "onpolicy", "envs")]
    sys.modules["onpolicy.envs"] = envs
    from onpolicy.envs.mpe.MPE_env import MPEEnv          # the reference's
    out = {}
    cases = [(3, 3, 25, 7), (8, 8, 12, 11), (2, 3, 6, 5)]      # (agents, landmarks, episode_length, seed)
    for ci, (A, L, T, seed) in enumerate(cases):
        args = types.SimpleNamespace(scenario_name="simple_spread", num_agents=A, num_landmarks=L, episode_length=T)
        env = MPEEnv(args)
        env.seed(seed)
        obs0 = np.array(env.reset(), dtype=np.float32)
        w = env.world
        key = "mpe%d_" % ci
        out[key + "dims"] = np.array([A, L, T])
        out[key + "pos0"] = np.array([a.state.p_pos for a in w.agents])
        out[key + "vel0"] = np.array([a.state.p_vel for a in w.agents])
        out[key + "landmarks"] = np.array([l.state.p_pos for l in w.landmarks])
        out[key + "obs0"] = obs0
        rng = np.random.default_rng(seed)
        acts, obs_all, rew_all, done_all, pos_all = [], [], [], [], []
        for t in range(T):
            a = np.eye(5)[rng.integers(0, 5, A)]
            obs, rew, done, info = env.step(list(a))
            acts.append(a)
            obs_all.append(np.array(obs, dtype=np.float32))
            rew_all.append(np.array(rew, dtype=np.float64))
            done_all.append(np.array(done))
            pos_all.append(np.array([ag.state.p_pos for ag in w.agents]))
        out[key + "actions"] = np.array(acts)
        out[key + "obs"] = np.array(obs_all)
        out[key + "rewards"] = np.array(rew_all)
        out[key + "dones"] = np.array(done_all)
        out[key + "pos"] = np.array(pos_all)
        out[key + "obs_dim"] = np.array(env.observation_space[0].shape)
        out[key + "share_obs_dim"] = np.array(env.share_observation_space[0].shape)
    np.savez_compressed(os.path.join(GOLD, "mpe_spread_cases.npz"), **out)
    print("mpe_spread_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(GOLD, "mpe_spread_cases.npz"))))
    print("case 0: obs", out["mpe0_obs"].shape, "rewards[0]", out["mpe0_rewards"][0].ravel(), "dones[-1]", out["mpe0_dones"][-1])


if __name__ == "__main__":
    main()
