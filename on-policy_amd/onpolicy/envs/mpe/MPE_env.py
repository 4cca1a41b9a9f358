"""``MPEEnv(args)``: one multi-agent particle world behind the single-env protocol that the VecEnv wrappers
drive (reference onpolicy/envs/mpe/MPE_env.py:4-33 builds a ``MultiAgentEnv`` from a scenario module;
environment.py:100-165 is its ``reset`` / ``step``).  Built in: ``simple_spread`` (one world of the vectorised
implementation in simple_spread.py, without its internal auto-reset -- the VecEnv worker resets a finished env).
Other scenarios come from an external env tree (``MAPPO_ENVS_PATH``, see onpolicy/envs/__init__.py), whose
``MPE_env`` module then shadows nothing: this one is found first, so it delegates by scenario name.
"""
import importlib

import numpy as np

from onpolicy.envs.mpe.simple_spread import VecSimpleSpread


class SimpleSpreadEnv(object):
    """reset() -> obs [A, Do];  step(actions [A, 5] one-hot) -> obs, rewards [A, 1], dones [A], infos [A dicts]."""

    def __init__(self, num_agents=3, num_landmarks=None, episode_length=25, seed=1):
        self._kw = dict(num_agents=num_agents, num_landmarks=num_landmarks, episode_length=episode_length)
        self._world = VecSimpleSpread(1, seed=seed, auto_reset=False, **self._kw)
        self.n = self.num_agents = int(num_agents)
        self.observation_space = self._world.observation_space
        self.share_observation_space = self._world.share_observation_space
        self.action_space = self._world.action_space

    def seed(self, seed=None):
        self._world.rng = np.random.default_rng(1 if seed is None else seed)

    def reset(self):
        return self._world.reset()[0]

    def step(self, action_n):
        obs, rewards, dones, infos = self._world.step(np.asarray(action_n, dtype=np.float64)[None])
        return obs[0], rewards[0], dones[0], infos[0]

    def render(self, mode="human"):
        raise NotImplementedError("rendering needs the reference's pyglet-based environment (MAPPO_ENVS_PATH)")

    def close(self):
        pass


def MPEEnv(args):
    """Factory with the reference's signature: ``args.scenario_name``, ``num_agents``, ``num_landmarks``,
    ``episode_length``."""
    if args.scenario_name == "simple_spread":
        return SimpleSpreadEnv(args.num_agents, getattr(args, "num_landmarks", None), args.episode_length)
    try:    # the reference's layout: scenarios/<name>.py with a Scenario class + environment.MultiAgentEnv
        scenarios = importlib.import_module("onpolicy.envs.mpe.scenarios")
        environment = importlib.import_module("onpolicy.envs.mpe.environment")
    except ImportError as e:
        raise NotImplementedError("scenario %r is not built in; point MAPPO_ENVS_PATH at an env tree that "
                                  "provides onpolicy.envs.mpe.scenarios (%s)" % (args.scenario_name, e))
    scenario = scenarios.load(args.scenario_name + ".py").Scenario()
    world = scenario.make_world(args)
    return environment.MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation, scenario.info)
