"""``HanabiEnv(args, seed)``: the per-env class of the reference (onpolicy/envs/hanabi/Hanabi_Env.py:82-500) as a
one-table view of the batched stepper.  Same constructor arguments (``args.hanabi_name``, ``args.num_agents``,
``args.use_obs_instead_of_state``), spaces, ``reset(choose)`` / ``step(action)`` return values and score-differential
reward; observations come out as float32 vectors instead of Python lists of ints."""
import numpy as np

from . import batch as _batch
from onpolicy.envs.spaces import Discrete


class HanabiEnv(object):
    def __init__(self, args, seed):
        self._seed = seed
        rules = _batch.rules_for(args.hanabi_name, args.num_agents)
        self.obs_instead_of_state = args.use_obs_instead_of_state
        share = _batch.SHARE_ALL_PLAYERS if self.obs_instead_of_state else _batch.SHARE_OWN_HAND
        self._batch = _batch.HanabiBatch(rules, [seed], share)
        b = self._batch
        self.players = b.players
        self.action_space = [Discrete(b.num_moves) for _ in range(self.players)]
        self.observation_space = [[b.obs_len + self.players] for _ in range(self.players)]
        self.share_observation_space = [[b.share_len + self.players] for _ in range(self.players)]

    def seed(self, seed=None):
        np.random.seed(1 if seed is None else seed)

    def vectorized_observation_shape(self):
        return [self._batch.obs_len]

    def vectorized_share_observation_shape(self):
        return [self._batch.share_len]

    def num_moves(self):
        return self._batch.num_moves

    def _rows(self):
        b = self._batch
        return b.obs[0].copy(), b.share_obs[0].copy(), b.available_actions[0].copy()

    def reset(self, choose=True):
        b = self._batch
        if choose:
            b.reset()
            b.encode()
        else:
            b.encode([False])
        return self._rows()

    def step(self, action):
        b = self._batch
        action = int(action[0])
        b.step([action])
        if action == -1:      # not this env's turn to be stepped (Hanabi_Env.py:460-468)
            b.encode([False])
            obs, share_obs, available_actions = self._rows()
            return obs, share_obs, np.zeros((self.players, 1)), None, {"score": int(b.scores[0])}, available_actions
        b.encode()
        obs, share_obs, available_actions = self._rows()
        rewards = [[float(b.rewards[0])]] * self.players
        return obs, share_obs, rewards, bool(b.status[0]), {"score": int(b.scores[0])}, available_actions

    def state(self):
        """Tokens, deck size, score, fireworks, ... of the table (for logging)."""
        return self._batch.table_state(0)

    def close(self):
        self._batch.close()
