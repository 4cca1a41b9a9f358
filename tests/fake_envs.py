"""Deterministic stand-ins for the vectorised envs the runners talk to (same reset / step
protocol and space lists as onpolicy/envs/env_wrappers.py of the reference, class-name-typed
spaces).  They record what they emitted so that tests can compare it with the buffer."""
import numpy as np

from helpers import Box, Discrete


class FakeMPEVecEnv(object):
    """SubprocVecEnv protocol of the MPE wrappers: reset() -> obs [N, A, Do];
    step(one_hot_actions [N, A, na]) -> obs, rewards [N, A, 1], dones [N, A], infos."""

    def __init__(self, n_threads, n_agents, obs_dim, n_actions, done_every=5, seed=0):
        self.n, self.a, self.do, self.na = n_threads, n_agents, obs_dim, n_actions
        self.observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.share_observation_space = [Box((obs_dim * n_agents,)) for _ in range(n_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(n_agents)]
        self.done_every = done_every
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.log = []

    def _obs(self):
        return self.rng.standard_normal((self.n, self.a, self.do)).astype(np.float32)

    def reset(self):
        self.t = 0
        obs = self._obs()
        self.log.append(dict(kind="reset", obs=obs))
        return obs

    def step(self, actions):
        actions = np.asarray(actions)        # the separated runner passes [envs][agents] lists
        assert actions.shape == (self.n, self.a, self.na) and np.all(actions.sum(-1) == 1)
        self.t += 1
        obs = self._obs()
        act_id = actions.argmax(-1)
        rewards = (act_id[..., None] * 0.1 + obs[..., :1]).astype(np.float32)
        dones = np.zeros((self.n, self.a), dtype=bool)
        if self.t % self.done_every == 0:
            dones[self.t % self.n] = True
        infos = [[{"individual_reward": float(rewards[i, j, 0])} for j in range(self.a)] for i in range(self.n)]
        self.log.append(dict(kind="step", obs=obs, rewards=rewards, dones=dones, actions=act_id))
        return obs, rewards, dones, infos

    def close(self):
        pass


class FakeSMACVecEnv(object):
    """ShareSubprocVecEnv protocol of the SMAC wrappers: reset() -> obs, share_obs, available_actions;
    step(actions [N, A, 1]) -> obs, share_obs, rewards, dones, infos, available_actions."""

    def __init__(self, n_threads, n_agents, obs_dim, state_dim, n_actions, seed=0):
        self.n, self.a, self.do, self.ds, self.na = n_threads, n_agents, obs_dim, state_dim, n_actions
        self.observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.share_observation_space = [Box((state_dim,)) for _ in range(n_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(n_agents)]
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.avail = None
        self.log = []

    def _emit(self):
        obs = self.rng.standard_normal((self.n, self.a, self.do)).astype(np.float32)
        share = self.rng.standard_normal((self.n, self.a, self.ds)).astype(np.float32)
        avail = (self.rng.random((self.n, self.a, self.na)) < 0.6).astype(np.float32)
        avail[..., 0] = 1.0
        self.avail = avail
        return obs, share, avail

    def reset(self):
        self.t = 0
        obs, share, avail = self._emit()
        self.log.append(dict(kind="reset", obs=obs, share_obs=share, available_actions=avail))
        return obs, share, avail

    def step(self, actions):
        actions = np.asarray(actions)
        assert actions.shape == (self.n, self.a, 1)
        act = actions[..., 0].astype(np.int64)
        # the policy must respect the availability mask it was given
        assert np.all(np.take_along_axis(self.avail, act[..., None], -1) == 1.0)
        self.t += 1
        obs, share, avail = self._emit()
        rewards = self.rng.standard_normal((self.n, self.a, 1)).astype(np.float32)
        dones = self.rng.random((self.n, self.a)) < 0.15            # individual deaths
        if self.t % 4 == 0:
            dones[self.t % self.n] = True                            # a whole team finishes
        infos = [[{"bad_transition": bool((self.t + i + j) % 7 == 0), "battles_won": self.t // 4,
                   "battles_game": self.t // 2, "won": True} for j in range(self.a)] for i in range(self.n)]
        self.log.append(dict(kind="step", obs=obs, share_obs=share, rewards=rewards, dones=dones,
                             available_actions=avail, actions=act, infos=infos))
        return obs, share, rewards, dones, infos, avail

    def close(self):
        pass


class FakeChooseVecEnv(object):
    """ChooseSubprocVecEnv protocol of the Hanabi wrappers: one acting player per env and step.
    reset(choose [N] bool) -> obs [N, Do], share_obs [N, Ds], available_actions [N, na] (zeros for
    envs that are not reset); step(actions [N, 1], -1 = idle) -> obs, share_obs, rewards [N, A, 1]...
    here rewards are [N, 1] broadcast to all players like the Hanabi env, dones (True / False / None
    for idle envs), infos, available_actions."""

    def __init__(self, n_threads, n_agents, obs_dim, state_dim, n_actions, seed=0):
        self.n, self.a, self.do, self.ds, self.na = n_threads, n_agents, obs_dim, state_dim, n_actions
        self.observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.share_observation_space = [Box((state_dim,)) for _ in range(n_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(n_agents)]
        self.rng = np.random.default_rng(seed)
        self.left = np.zeros(n_threads, dtype=np.int64)       # moves left in the current game
        self.steps = 0
        self.games = 0

    def _emit(self, which):
        obs = np.zeros((self.n, self.do), np.float32)
        share = np.zeros((self.n, self.ds), np.float32)
        avail = np.zeros((self.n, self.na), np.float32)
        k = int(which.sum())
        if k:
            obs[which] = self.rng.standard_normal((k, self.do))
            share[which] = self.rng.standard_normal((k, self.ds))
            av = (self.rng.random((k, self.na)) < 0.5).astype(np.float32)
            av[:, 0] = 1.0
            avail[which] = av
        return obs, share, avail

    def reset(self, choose):
        choose = np.asarray(choose, dtype=bool)
        self.left[choose] = self.rng.integers(5, 14, int(choose.sum()))
        return self._emit(choose)

    def step(self, actions):
        actions = np.asarray(actions)
        assert actions.shape == (self.n, 1)
        acting = actions[:, 0] >= 0
        assert np.all(self.left[acting] > 0), "an env without a running game was asked to act"
        self.steps += int(acting.sum())
        self.left[acting] -= 1
        finished = acting & (self.left == 0)
        running = acting & ~finished
        obs, share, avail = self._emit(running)
        rewards = np.zeros((self.n, self.a, 1), np.float32)
        rewards[acting] = self.rng.random((int(acting.sum()), 1, 1)).astype(np.float32)
        dones = np.array([True if f else (False if r else None) for f, r in zip(finished, running)], dtype=object)
        infos = [{"score": float(self.rng.integers(0, 25))} if f else {} for f in finished]
        self.games += int(finished.sum())
        return obs, share, rewards, dones, infos, avail

    def close(self):
        pass


class TinyEnv(object):
    """One multi-agent env (not vectorised) for the VecEnv wrapper tests.  ``share=True`` speaks the SMAC
    protocol (obs, share_obs, rewards, dones, infos, available_actions), otherwise the MPE one.
    ``choose=True``: ``reset(choose)`` returns zeros when the flag is False (Hanabi protocol)."""

    def __init__(self, seed, n_agents=2, obs_dim=3, horizon=3, share=False, choose=False):
        self.a, self.do, self.h, self.share, self.choose = n_agents, obs_dim, horizon, share, choose
        self.observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.share_observation_space = [Box((obs_dim * n_agents,)) for _ in range(n_agents)]
        self.action_space = [Discrete(4) for _ in range(n_agents)]
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.resets = 0

    def _obs(self):
        obs = self.rng.standard_normal((self.a, self.do)).astype(np.float32)
        if not self.share:
            return obs
        return obs, np.tile(obs.reshape(1, -1), (self.a, 1)), np.ones((self.a, 4), np.float32)

    def reset(self, choose=True):
        if self.choose and not choose:
            z = np.zeros((self.a, self.do), np.float32)
            return (z, np.zeros((self.a, self.do * self.a), np.float32), np.zeros((self.a, 4), np.float32)) \
                if self.share else z
        self.t = 0
        self.resets += 1
        out = self._obs()
        if self.share:
            out[0][0, 0] = 1000.0 + self.resets      # marks observations that come from a reset
        else:
            out[0, 0] = 1000.0 + self.resets
        return out

    def step(self, action):
        self.t += 1
        rewards = np.full((self.a, 1), float(np.sum(action)) + self.t, dtype=np.float32)
        dones = np.full(self.a, self.t >= self.h)
        infos = [{"t": self.t} for _ in range(self.a)]
        if self.share:
            obs, share_obs, avail = self._obs()
            return obs, share_obs, rewards, dones, infos, avail
        return self._obs(), rewards, dones, infos

    def close(self):
        pass


class FakeFootballVecEnv(object):
    """DummyVecEnv protocol of the football wrapper: reset() -> obs [N, A, Do]; step(list of [A] int arrays) ->
    obs, rewards [N, A, 1], dones [N, A], infos (one dict per env)."""

    def __init__(self, n_threads, n_agents, obs_dim, n_actions, horizon=5, seed=0):
        self.n, self.a, self.do, self.na, self.h = n_threads, n_agents, obs_dim, n_actions, horizon
        self.observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.share_observation_space = [Box((obs_dim,)) for _ in range(n_agents)]
        self.action_space = [Discrete(n_actions) for _ in range(n_agents)]
        self.rng = np.random.default_rng(seed)
        self.left = np.full(n_threads, horizon)
        self.log = []

    def reset(self):
        self.left[:] = self.h
        obs = self.rng.standard_normal((self.n, self.a, self.do)).astype(np.float32)
        self.log.append(dict(kind="reset", obs=obs))
        return obs

    def step(self, actions):
        assert len(actions) == self.n and all(np.asarray(a).shape == (self.a,) for a in actions)
        act = np.stack([np.asarray(a) for a in actions]).astype(np.int64)
        assert act.min() >= 0 and act.max() < self.na
        self.left -= 1
        self.left[1] -= 1 if self.left[1] > 0 else 0          # env 1 finishes twice as fast
        done_env = self.left <= 0
        obs = self.rng.standard_normal((self.n, self.a, self.do)).astype(np.float32)
        rewards = self.rng.standard_normal((self.n, self.a, 1)).astype(np.float32)
        dones = np.repeat(done_env[:, None], self.a, axis=1)
        infos = [{"score_reward": int(i % 2), "max_steps": self.h, "steps_left": int(max(self.left[i], 0))}
                 for i in range(self.n)]
        self.left[done_env] = self.h
        self.log.append(dict(kind="step", obs=obs, rewards=rewards, dones=dones, actions=act))
        return obs, rewards, dones, infos

    def close(self):
        pass
