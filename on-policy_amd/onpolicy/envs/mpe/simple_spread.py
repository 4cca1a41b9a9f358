"""Cooperative navigation ("simple_spread") of the multi-agent particle environment, vectorised over
rollout threads in numpy.

Scope note: environments are outside the hot path of this repository ("next"
row f2).  This module exists so that BASELINE.json's configs[0] (MPE simple_spread, 3 agents,
8 rollout threads, episode length 25) runs end to end without gym / seaborn / subprocess workers.
It follows the published MPE dynamics (Lowe et al. 2017; the reference vendors them under
onpolicy/envs/mpe/{core,environment}.py and scenarios/simple_spread.py): point-mass agents with
damping 0.25, dt 0.1, action sensitivity 5, soft contact forces, shared reward
-sum_landmarks min_agents dist - collisions; trajectories are pinned to the reference's own environment by
tests/test_mpe_env_cpu.py (fixtures: oracle/make_golden_mpe.py).  All ``n_threads`` worlds advance in one set of array
operations, so a rollout step costs one numpy pass instead of ``n_threads`` pipe round trips.

VecEnv protocol (reference onpolicy/envs/env_wrappers.py:235-298): ``reset() -> obs [N, A, Do]``,
``step(one_hot_actions [N, A, 5]) -> obs, rewards [N, A, 1], dones [N, A], infos``.
"""
import numpy as np

from onpolicy.envs.spaces import Box, Discrete

_DT = 0.1
_DAMPING = 0.25
_SENSITIVITY = 5.0
_CONTACT_FORCE = 1e2
_CONTACT_MARGIN = 1e-3
_AGENT_SIZE = 0.15


class VecSimpleSpread(object):
    def __init__(self, n_threads, num_agents=3, num_landmarks=None, episode_length=25, seed=1, auto_reset=True):
        self.auto_reset = auto_reset      # finished worlds restart inside step(), as the vec-env workers do
        self.n, self.a = int(n_threads), int(num_agents)
        self.l = int(num_landmarks) if num_landmarks is not None else self.a
        self.world_length = int(episode_length)
        self.rng = np.random.default_rng(seed)
        obs_dim = 4 + 2 * self.l + 4 * (self.a - 1)
        self.observation_space = [Box(shape=(obs_dim,)) for _ in range(self.a)]
        self.share_observation_space = [Box(shape=(obs_dim * self.a,)) for _ in range(self.a)]
        self.action_space = [Discrete(5) for _ in range(self.a)]
        self.pos = np.zeros((self.n, self.a, 2))
        self.vel = np.zeros((self.n, self.a, 2))
        self.landmarks = np.zeros((self.n, self.l, 2))
        self.t = np.zeros(self.n, dtype=np.int64)

    # -- helpers
    def _reset_worlds(self, which):
        k = int(which.sum())
        if k:
            self.pos[which] = self.rng.uniform(-1, 1, (k, self.a, 2))
            self.vel[which] = 0.0
            self.landmarks[which] = self.rng.uniform(-1, 1, (k, self.l, 2))
            self.t[which] = 0

    def _obs(self):
        n, a = self.n, self.a
        rel_land = (self.landmarks[:, None, :, :] - self.pos[:, :, None, :]).reshape(n, a, -1)
        rel_other = self.pos[:, None, :, :] - self.pos[:, :, None, :]          # [n, i, j, 2] = pos_j - pos_i
        keep = ~np.eye(a, dtype=bool)
        rel_other = rel_other[:, keep].reshape(n, a, (a - 1) * 2)
        comm = np.zeros((n, a, (a - 1) * 2))                                   # agents are silent
        return np.concatenate([self.vel, self.pos, rel_land, rel_other, comm], -1).astype(np.float32)

    def _collision_forces(self):
        delta = self.pos[:, :, None, :] - self.pos[:, None, :, :]              # [n, i, j, 2]
        dist = np.sqrt((delta ** 2).sum(-1))
        dist_min = 2 * _AGENT_SIZE
        k = _CONTACT_MARGIN
        pen = np.logaddexp(0.0, -(dist - dist_min) / k) * k
        with np.errstate(divide="ignore", invalid="ignore"):
            f = _CONTACT_FORCE * delta / dist[..., None] * pen[..., None]
        f[:, np.arange(self.a), np.arange(self.a)] = 0.0                        # no self force
        return np.nan_to_num(f).sum(2)                                          # force on i

    def _reward(self):
        d = np.sqrt(((self.pos[:, :, None, :] - self.landmarks[:, None, :, :]) ** 2).sum(-1))  # [n, a, l]
        cover = -d.min(1).sum(-1)                                                               # [n]
        dd = np.sqrt(((self.pos[:, :, None, :] - self.pos[:, None, :, :]) ** 2).sum(-1))
        # the reference's loop runs over ALL agents, the agent itself included (scenarios/simple_spread.py:78-81:
        # is_collision(agent, agent) is true), so every agent carries a constant -1; kept for identical rewards
        hits = dd < 2 * _AGENT_SIZE
        per_agent = cover[:, None] - hits.sum(-1)                                               # [n, a]
        return per_agent

    # -- VecEnv protocol
    def reset(self):
        self._reset_worlds(np.ones(self.n, dtype=bool))
        return self._obs()

    def step(self, actions):
        actions = np.asarray(actions, dtype=np.float64)
        assert actions.shape == (self.n, self.a, 5), actions.shape
        u = np.stack([actions[..., 1] - actions[..., 2], actions[..., 3] - actions[..., 4]], -1) * _SENSITIVITY
        force = u + self._collision_forces()
        self.vel = self.vel * (1 - _DAMPING) + force * _DT
        self.pos = self.pos + self.vel * _DT
        self.t += 1
        per_agent = self._reward()
        shared = per_agent.sum(-1, keepdims=True)                              # shared reward: sum over agents
        rewards = np.repeat(shared, self.a, 1)[..., None].astype(np.float32)
        done_env = self.t >= self.world_length
        dones = np.repeat(done_env[:, None], self.a, 1)
        infos = [[{"individual_reward": float(per_agent[i, j])} for j in range(self.a)] for i in range(self.n)]
        if self.auto_reset:
            self._reset_worlds(done_env)
        return self._obs(), rewards, dones, infos

    def close(self):
        pass


class _LazyInfos(object):
    """``infos[i][j]['individual_reward']`` of the reference protocol, materialised from the device only if somebody
    looks (the runner reads the last step's infos once per log interval)."""

    def __init__(self, per_agent):
        self._per_agent, self._rows = per_agent, None

    def _materialise(self):
        if self._rows is None:
            self._rows = [[{"individual_reward": float(v)} for v in row] for row in self._per_agent.cpu().tolist()]
        return self._rows

    def __len__(self):
        return self._per_agent.shape[0]

    def __iter__(self):
        return iter(self._materialise())

    def __getitem__(self, i):
        return self._materialise()[i]


class TorchSimpleSpread(object):
    """The same worlds as ``VecSimpleSpread`` held as tensors on ``device``: with the policy, the rollout buffer and
    the env on the GPU, a rollout step moves nothing over PCIe.  State is float64 like
    the reference's numpy physics; observations and rewards leave as float32.

    ``device_resident = True`` tells the runner to hand over the integer action tensor [N, A, 1] as it comes out of
    the policy (one-hot [N, A, 5] arrays are accepted as well) and to expect tensors back: obs [N, A, Do] float32,
    rewards [N, A, 1] float32, dones [N, A] bool, infos (lazy)."""
    device_resident = True

    def __init__(self, n_threads, num_agents=3, num_landmarks=None, episode_length=25, seed=1, auto_reset=True,
                 device="cpu"):
        import torch
        self._torch = torch
        self.auto_reset = auto_reset
        self.device = torch.device(device)
        self.n, self.a = int(n_threads), int(num_agents)
        self.l = int(num_landmarks) if num_landmarks is not None else self.a
        self.world_length = int(episode_length)
        self.rng = torch.Generator(device=self.device)
        self.rng.manual_seed(int(seed))
        obs_dim = 4 + 2 * self.l + 4 * (self.a - 1)
        self.observation_space = [Box(shape=(obs_dim,)) for _ in range(self.a)]
        self.share_observation_space = [Box(shape=(obs_dim * self.a,)) for _ in range(self.a)]
        self.action_space = [Discrete(5) for _ in range(self.a)]
        f64 = dict(dtype=torch.float64, device=self.device)
        self.pos = torch.zeros(self.n, self.a, 2, **f64)
        self.vel = torch.zeros(self.n, self.a, 2, **f64)
        self.landmarks = torch.zeros(self.n, self.l, 2, **f64)
        self.t = torch.zeros(self.n, dtype=torch.int64, device=self.device)
        self._others = ~torch.eye(self.a, dtype=torch.bool, device=self.device)
        # action index -> force direction (environment.py: u[0] += a[1] - a[2], u[1] += a[3] - a[4])
        self._directions = torch.tensor([[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1]], **f64) * _SENSITIVITY

    def _uniform(self, *shape):
        torch = self._torch
        # U(-1, 1) in one launch: uniform_ evaluates rand * (to - from) + from on the same draws as torch.rand
        return torch.empty(*shape, dtype=torch.float64, device=self.device).uniform_(-1.0, 1.0, generator=self.rng)

    def _reset_worlds(self, which):
        """Branch-free (no host sync): fresh positions are drawn for every world and kept where ``which`` is set."""
        torch = self._torch
        w = which.view(-1, 1, 1)
        self.pos = torch.where(w, self._uniform(self.n, self.a, 2), self.pos)
        self.vel = torch.where(w, torch.zeros_like(self.vel), self.vel)
        self.landmarks = torch.where(w, self._uniform(self.n, self.l, 2), self.landmarks)
        self.t = torch.where(which, torch.zeros_like(self.t), self.t)

    def _obs(self):
        torch = self._torch
        n, a = self.n, self.a
        rel_land = (self.landmarks[:, None, :, :] - self.pos[:, :, None, :]).reshape(n, a, -1)
        rel_other = self.pos[:, None, :, :] - self.pos[:, :, None, :]
        rel_other = rel_other[:, self._others].reshape(n, a, (a - 1) * 2)
        comm = torch.zeros(n, a, (a - 1) * 2, dtype=torch.float64, device=self.device)
        return torch.cat([self.vel, self.pos, rel_land, rel_other, comm], -1).to(torch.float32)

    def _collision_forces(self):
        torch = self._torch
        delta = self.pos[:, :, None, :] - self.pos[:, None, :, :]
        dist = torch.sqrt((delta ** 2).sum(-1))
        pen = torch.logaddexp(torch.zeros_like(dist), -(dist - 2 * _AGENT_SIZE) / _CONTACT_MARGIN) * _CONTACT_MARGIN
        f = _CONTACT_FORCE * delta / dist[..., None] * pen[..., None]
        f = torch.where(self._others[None, :, :, None], f, torch.zeros_like(f))       # no self force (0 / 0 there)
        return torch.nan_to_num(f, nan=0.0, posinf=0.0, neginf=0.0).sum(2)

    def _reward(self):
        torch = self._torch
        d = torch.sqrt(((self.pos[:, :, None, :] - self.landmarks[:, None, :, :]) ** 2).sum(-1))
        cover = -d.min(1).values.sum(-1)
        dd = torch.sqrt(((self.pos[:, :, None, :] - self.pos[:, None, :, :]) ** 2).sum(-1))
        hits = dd < 2 * _AGENT_SIZE            # the agent itself included, as in the reference (see VecSimpleSpread)
        return cover[:, None] - hits.sum(-1)

    def reset(self):
        torch = self._torch
        self._reset_worlds(torch.ones(self.n, dtype=torch.bool, device=self.device))
        return self._obs()

    @property
    def graph_safe(self):
        """True when ``step`` advances pos / vel / landmarks / t IN PLACE (the K11 kernel path): a captured rollout graph
        (runner/shared/rollout_graph.py) may then replay it.  The tensor-op path rebinds them."""
        return self.device.type == "cuda" and self.a <= 16 and self.l <= 16

    def step(self, actions):
        torch = self._torch
        actions = torch.as_tensor(actions, device=self.device)
        if self.device.type == "cuda" and self.a <= 16 and self.l <= 16:
            return self._step_kernel(actions)
        return self._step_ops(actions)

    def _step_kernel(self, actions):
        """The whole step as one launch (K11, ``mappo_simple_spread_step``): same arithmetic and the same generator
        draws as ``_step_ops`` (~60 small launches; 140 ms per step at 4096 worlds, against 30 us)."""
        torch = self._torch
        from onpolicy import _native
        if actions.shape == (self.n, self.a, 5):                         # one-hot (the host protocol)
            idx = actions.argmax(-1)
        else:
            idx = actions.reshape(self.n, self.a)
        idx = idx.to(torch.int64).contiguous()
        fresh_pos = self._uniform(self.n, self.a, 2) if self.auto_reset else None
        fresh_land = self._uniform(self.n, self.l, 2) if self.auto_reset else None
        obs_dim = 4 + 2 * self.l + 4 * (self.a - 1)
        obs = torch.empty(self.n, self.a, obs_dim, dtype=torch.float32, device=self.device)
        rewards = torch.empty(self.n, self.a, 1, dtype=torch.float32, device=self.device)
        dones = torch.empty(self.n, self.a, dtype=torch.bool, device=self.device)
        per_agent = torch.empty(self.n, self.a, dtype=torch.float64, device=self.device)
        for name in ("pos", "vel", "landmarks", "t"):
            setattr(self, name, getattr(self, name).contiguous())
        p = _native.ptr
        _native.check(_native.lib().mappo_simple_spread_step(
            p(self.pos), p(self.vel), p(self.landmarks), p(self.t), p(idx), p(fresh_pos), p(fresh_land), p(obs),
            p(rewards), p(dones), p(per_agent), self.n, self.a, self.l, self.world_length, int(self.auto_reset),
            _native.stream_of(self.device)), "mappo_simple_spread_step")
        return obs, rewards, dones, _LazyInfos(per_agent)

    def _step_ops(self, actions):
        torch = self._torch
        if actions.shape == (self.n, self.a, 5):                         # one-hot (the host protocol)
            a = actions.to(torch.float64)
            u = torch.stack([a[..., 1] - a[..., 2], a[..., 3] - a[..., 4]], -1) * _SENSITIVITY
        else:                                                            # action indices straight from the policy
            assert actions.shape in ((self.n, self.a, 1), (self.n, self.a)), tuple(actions.shape)
            u = self._directions[actions.reshape(self.n, self.a).long()]
        force = u + self._collision_forces()
        self.vel = self.vel * (1 - _DAMPING) + force * _DT
        self.pos = self.pos + self.vel * _DT
        self.t = self.t + 1
        per_agent = self._reward()
        rewards = per_agent.sum(-1, keepdim=True).expand(self.n, self.a).unsqueeze(-1).to(torch.float32)
        done_env = self.t >= self.world_length
        dones = done_env[:, None].expand(self.n, self.a)
        if self.auto_reset:
            self._reset_worlds(done_env)
        return self._obs(), rewards, dones, _LazyInfos(per_agent)

    def close(self):
        pass
