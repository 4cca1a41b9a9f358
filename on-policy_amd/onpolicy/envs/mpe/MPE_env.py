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

    def render(self, mode="human", size=350):
        """``rgb_array``: [frame], one uint8 [size, size, 3] image of the world rasterised in numpy -- agents blue,
        landmarks dark grey, view [-1.5, 1.5]^2 centred on the origin (the reference draws the same entities
        through pyglet, environment.py:219-301, one viewer per env when ``shared_viewer``).  ``human`` needs a
        display and the reference's renderer (MAPPO_ENVS_PATH)."""
        if mode != "rgb_array":
            raise NotImplementedError("on-screen rendering needs the reference's pyglet-based environment "
                                      "(MAPPO_ENVS_PATH); use --save_gifs / mode='rgb_array'")
        w = self._world
        frame = np.full((size, size, 3), 255, dtype=np.uint8)
        span = 1.5
        ys, xs = np.mgrid[0:size, 0:size]
        wx = (xs + 0.5) / size * 2 * span - span
        wy = span - (ys + 0.5) / size * 2 * span              # image rows grow downwards
        for centres, radius, colour in ((w.landmarks[0], 0.05, (64, 64, 64)), (w.pos[0], 0.15, (89, 89, 217))):
            for cx, cy in centres:
                frame[(wx - cx) ** 2 + (wy - cy) ** 2 <= radius ** 2] = colour
        return [frame]

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
