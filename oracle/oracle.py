"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front end of the plain-C oracle (oracle/mappo_oracle.c) plus ``OracleBuffer``, a
host-side restatement of the reference's ``SharedReplayBuffer``
(reference onpolicy/utils/shared_buffer.py:21-608).  Only tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg may import this module; nothing under on-policy_amd/ does.

Pinned against the reference by tests/test_oracle_golden.py (fixtures made by
oracle/make_golden.py from the reference itself).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

USE_GAE, PROPER_TIME_LIMITS, DENORM = 1, 2, 4

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build():
    """Compile oracle/_build/liboracle.so with gcc (idempotent)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_compute_returns.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_uint]
        L.orc_compute_returns.restype = None
        L.orc_compute_returns_mat.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.orc_compute_returns_mat.restype = None
        L.orc_advantages.argtypes = [_f32p, _f32p, _f32p, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_int, ctypes.c_int64]
        L.orc_advantages.restype = None
        L.orc_adv_moments.argtypes = [_f32p, _f32p, ctypes.c_int64, _f64p]
        L.orc_adv_moments.restype = None
        L.orc_adv_normalize.argtypes = [_f32p, ctypes.c_float, ctypes.c_float, _f32p, ctypes.c_int64]
        L.orc_adv_normalize.restype = None
        L.orc_gather_rows.argtypes = [_f32p, _i64p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.orc_gather_rows.restype = None
        L.orc_gather_chunks.argtypes = [_f32p, _i64p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.orc_gather_chunks.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(_f32p)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def normalizer_scalars(running_mean, running_mean_sq, debiasing_term, epsilon=1e-5):
    """(sigma, mu) of ValueNorm.running_mean_var / PopArt.debiased_mean_var
    (reference onpolicy/utils/valuenorm.py:32-37, onpolicy/algorithms/utils/popart.py:72-76),
    evaluated with float32 torch ops exactly like the reference does."""
    m = torch.as_tensor(running_mean, dtype=torch.float32).reshape(())
    sq = torch.as_tensor(running_mean_sq, dtype=torch.float32).reshape(())
    d = torch.as_tensor(debiasing_term, dtype=torch.float32).reshape(())
    mean = m / d.clamp(min=epsilon)
    mean_sq = sq / d.clamp(min=epsilon)
    var = (mean_sq - mean ** 2).clamp(min=1e-2)
    return float(torch.sqrt(var)), float(mean)


def compute_returns(rewards, value_preds, next_value, masks, bad_masks=None, *, sigma=1.0, mu=0.0,
                    gamma=0.99, gae_lambda=0.95, use_gae=True, use_proper_time_limits=False,
                    denorm=False):
    """SharedReplayBuffer.compute_returns (shared_buffer.py:179-262) on [T(+1), ...] float32 arrays.
    Returns (returns[T+1,...], value_preds[T+1,...]) as new arrays."""
    T = rewards.shape[0]
    shape1 = value_preds.shape
    C = int(np.prod(rewards.shape[1:]))
    r = _c(rewards).reshape(T, C)
    v = _c(value_preds).reshape(T + 1, C).copy()
    nv = _c(next_value).reshape(C)
    m = _c(masks).reshape(T + 1, C)
    b = None if bad_masks is None else _c(bad_masks).reshape(T + 1, C)
    ret = np.zeros((T + 1, C), dtype=np.float32)
    flags = (USE_GAE if use_gae else 0) | (PROPER_TIME_LIMITS if use_proper_time_limits else 0) | \
            (DENORM if denorm else 0)
    if use_proper_time_limits:
        assert b is not None
    lib().orc_compute_returns(_p(r), _p(v), _p(nv), _p(m), _p(b), _p(ret), float(sigma), float(mu),
                              T, C, float(gamma), float(gae_lambda), flags)
    return ret.reshape(shape1), v.reshape(shape1)


def compute_returns_mat(rewards, value_preds, next_value, masks, *, num_agents, sigma=1.0, mu=0.0, gamma=0.99,
                        gae_lambda=0.95, denorm=False):
    """The mat / mat_dec branches of compute_returns (shared_buffer.py:222-232, :241-251) on
    [T(+1), N, A, 1] arrays -> (returns [T+1, ...], value_preds [T+1, ...], advantages [T, ...])."""
    T = rewards.shape[0]
    C = int(np.prod(rewards.shape[1:]))
    v = _c(value_preds).reshape(T + 1, C).copy()
    ret = np.zeros((T + 1, C), dtype=np.float32)
    adv = np.zeros((T, C), dtype=np.float32)
    lib().orc_compute_returns_mat(_p(_c(rewards).reshape(T, C)), _p(v), _p(_c(next_value).reshape(C)),
                                  _p(_c(masks).reshape(T + 1, C)), _p(ret), _p(adv), float(sigma), float(mu), T, C,
                                  int(num_agents), float(gamma), float(gae_lambda), int(bool(denorm)))
    return ret.reshape(value_preds.shape), v.reshape(value_preds.shape), adv.reshape(rewards.shape)


def advantages(returns, value_preds, *, sigma=1.0, mu=0.0, denorm=False):
    """r_mappo.py:179-182 on the first T rows."""
    T = returns.shape[0] - 1
    ret = _c(returns[:T])
    v = _c(value_preds[:T])
    out = np.empty_like(ret)
    lib().orc_advantages(_p(ret), _p(v), _p(out), float(sigma), float(mu), int(denorm), ret.size)
    return out


def adv_moments(adv, active_masks=None):
    """r_mappo.py:183-186 -> (mean, std, count) as Python floats (float64 accumulation)."""
    a = _c(adv)
    am = None if active_masks is None else _c(active_masks)
    out = np.zeros(3, dtype=np.float64)
    lib().orc_adv_moments(_p(a), _p(am), a.size, out.ctypes.data_as(_f64p))
    return float(out[0]), float(out[1]), float(out[2])


def adv_normalize(adv, mean, std):
    a = _c(adv)
    out = np.empty_like(a)
    lib().orc_adv_normalize(_p(a), float(np.float32(mean)), float(np.float32(std)), _p(out), a.size)
    return out


def gather_rows(field, idx):
    """field: [rows, D] float32; idx int64 -> [mb, D] (shared_buffer.py:379-396)."""
    f = _c(field)
    D = f.shape[-1] if f.ndim > 1 else 1
    f2 = f.reshape(-1, D)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty((idx.size, D), dtype=np.float32)
    lib().orc_gather_rows(_p(f2), idx.ctypes.data_as(_i64p), idx.size, D, _p(out))
    return out


def gather_chunks(field, idx, L, first_only=False):
    """field: [T, N, A, D...] float32 (the [:-1] slice for T+1-row fields); idx = chunk ids.
    Returns [L*mb, D...] (or [mb, D...] for first_only) like recurrent_generator
    (shared_buffer.py:499-608)."""
    f = _c(field)
    T, N, A = f.shape[:3]
    tail = f.shape[3:]
    D = int(np.prod(tail)) if tail else 1
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    rows = idx.size if first_only else idx.size * L
    out = np.empty((rows, D), dtype=np.float32)
    lib().orc_gather_chunks(_p(f.reshape(-1, D)), idx.ctypes.data_as(_i64p), idx.size, L, T, N, A, D,
                            int(first_only), _p(out))
    return out.reshape((rows,) + tuple(tail))


class OracleBuffer(object):
    """Host restatement of the reference SharedReplayBuffer (shared_buffer.py:21-608): numpy
    storage with the reference's shapes, compute_returns through the C oracle, generators through
    the C gathers.  Space objects are recognised by class name like the reference does
    (onpolicy/utils/util.py:31-52)."""

    def __init__(self, args, num_agents, obs_space, cent_obs_space, act_space):
        self.episode_length = T = args.episode_length
        self.n_rollout_threads = N = args.n_rollout_threads
        self.hidden_size = args.hidden_size
        self.recurrent_N = args.recurrent_N
        self.gamma = args.gamma
        self.gae_lambda = args.gae_lambda
        self._use_gae = args.use_gae
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_proper_time_limits = args.use_proper_time_limits
        self.algo = getattr(args, "algorithm_name", "mappo")
        self.num_agents = A = num_agents
        obs_shape = tuple(_shape_of(obs_space))
        share_shape = tuple(_shape_of(cent_obs_space))
        z = lambda *s: np.zeros(s, dtype=np.float32)
        self.share_obs = z(T + 1, N, A, *share_shape)          # :54
        self.obs = z(T + 1, N, A, *obs_shape)                  # :56
        self.rnn_states = z(T + 1, N, A, self.recurrent_N, self.hidden_size)  # :58
        self.rnn_states_critic = np.zeros_like(self.rnn_states)
        self.value_preds = z(T + 1, N, A, 1)
        self.returns = z(T + 1, N, A, 1)
        self.advantages = z(T, N, A, 1)
        if act_space.__class__.__name__ == 'Discrete':          # :69-73
            self.available_actions = np.ones((T + 1, N, A, act_space.n), dtype=np.float32)
        else:
            self.available_actions = None
        act_dim = _act_dim(act_space)
        self.actions = z(T, N, A, act_dim)
        self.action_log_probs = z(T, N, A, act_dim)
        self.rewards = z(T, N, A, 1)
        self.masks = np.ones((T + 1, N, A, 1), dtype=np.float32)
        self.bad_masks = np.ones_like(self.masks)
        self.active_masks = np.ones_like(self.masks)
        self.step = 0

    # -- shared_buffer.py:90-123
    def insert(self, share_obs, obs, rnn_states_actor, rnn_states_critic, actions, action_log_probs,
               value_preds, rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        s = self.step
        self.share_obs[s + 1] = share_obs
        self.obs[s + 1] = obs
        self.rnn_states[s + 1] = rnn_states_actor
        self.rnn_states_critic[s + 1] = rnn_states_critic
        self.actions[s] = actions
        self.action_log_probs[s] = action_log_probs
        self.value_preds[s] = value_preds
        self.rewards[s] = rewards
        self.masks[s + 1] = masks
        if bad_masks is not None:
            self.bad_masks[s + 1] = bad_masks
        if active_masks is not None:
            self.active_masks[s + 1] = active_masks
        if available_actions is not None:
            self.available_actions[s + 1] = available_actions
        self.step = (s + 1) % self.episode_length

    # -- shared_buffer.py:125-158
    def chooseinsert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs,
                     value_preds, rewards, masks, bad_masks=None, active_masks=None,
                     available_actions=None):
        s = self.step
        self.share_obs[s] = share_obs
        self.obs[s] = obs
        self.rnn_states[s + 1] = rnn_states
        self.rnn_states_critic[s + 1] = rnn_states_critic
        self.actions[s] = actions
        self.action_log_probs[s] = action_log_probs
        self.value_preds[s] = value_preds
        self.rewards[s] = rewards
        self.masks[s + 1] = masks
        if bad_masks is not None:
            self.bad_masks[s + 1] = bad_masks
        if active_masks is not None:
            self.active_masks[s] = active_masks
        if available_actions is not None:
            self.available_actions[s] = available_actions
        self.step = (s + 1) % self.episode_length

    # -- shared_buffer.py:160-170
    def after_update(self):
        for name in ("share_obs", "obs", "rnn_states", "rnn_states_critic", "masks", "bad_masks",
                     "active_masks", "available_actions"):
            arr = getattr(self, name)
            if arr is not None:
                arr[0] = arr[-1]

    # -- shared_buffer.py:172-177
    def chooseafter_update(self):
        for name in ("rnn_states", "rnn_states_critic", "masks", "bad_masks"):
            arr = getattr(self, name)
            arr[0] = arr[-1]

    def _scalars(self, value_normalizer):
        if (self._use_popart or self._use_valuenorm) and value_normalizer is not None:
            if hasattr(value_normalizer, "running_mean"):
                return normalizer_scalars(value_normalizer.running_mean, value_normalizer.running_mean_sq,
                                          value_normalizer.debiasing_term) + (True,)
            return normalizer_scalars(value_normalizer.mean, value_normalizer.mean_sq,
                                      value_normalizer.debiasing_term) + (True,)
        return 1.0, 0.0, False

    # -- shared_buffer.py:179-262
    def compute_returns(self, next_value, value_normalizer=None):
        sigma, mu, dn = self._scalars(value_normalizer)
        if self.algo in ("mat", "mat_dec") and self._use_gae and not self._use_proper_time_limits:   # :222,:241
            ret, v, adv = compute_returns_mat(self.rewards, self.value_preds, np.asarray(next_value, dtype=np.float32),
                                              self.masks, num_agents=self.num_agents, sigma=sigma, mu=mu,
                                              gamma=self.gamma, gae_lambda=self.gae_lambda, denorm=dn)
            self.returns[...] = ret
            self.value_preds[...] = v
            self.advantages[...] = adv
            return
        ret, v = compute_returns(self.rewards, self.value_preds, np.asarray(next_value, dtype=np.float32),
                                 self.masks, self.bad_masks, sigma=sigma, mu=mu, gamma=self.gamma,
                                 gae_lambda=self.gae_lambda, use_gae=self._use_gae,
                                 use_proper_time_limits=self._use_proper_time_limits, denorm=dn)
        self.returns[...] = ret
        self.value_preds[...] = v

    def _fields(self, advantages):
        T = self.episode_length
        f = [("share_obs", self.share_obs[:T]), ("obs", self.obs[:T]),
             ("rnn_states", self.rnn_states[:T]), ("rnn_states_critic", self.rnn_states_critic[:T]),
             ("actions", self.actions), ("value_preds", self.value_preds[:T]),
             ("returns", self.returns[:T]), ("masks", self.masks[:T]),
             ("active_masks", self.active_masks[:T]), ("action_log_probs", self.action_log_probs),
             ("advantages", advantages),
             ("available_actions", None if self.available_actions is None else self.available_actions[:T])]
        return f

    # -- shared_buffer.py:264-338: minibatches of whole (t, n) agent groups, agents consecutive
    def feed_forward_generator_transformer(self, advantages, num_mini_batch=None, mini_batch_size=None):
        T, N, A = self.rewards.shape[0:3]
        batch_size = N * T
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch
            mini_batch_size = batch_size // num_mini_batch
        rand = torch.randperm(batch_size).numpy()                                  # :284
        sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(num_mini_batch)]
        fields = self._fields(advantages)
        for indices in sampler:
            out = []
            for name, arr in fields:
                if arr is None:
                    out.append(None)
                    continue
                tail = arr.shape[3:]
                g = gather_rows(arr.reshape(batch_size, -1), indices)              # [mb, A*D]
                out.append(g.reshape((len(indices) * A,) + tuple(tail)))           # :315-331
            yield tuple(out)

    # -- shared_buffer.py:340-400
    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        T, N, A = self.rewards.shape[0:3]
        batch_size = N * T * A
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch
            mini_batch_size = batch_size // num_mini_batch
        rand = torch.randperm(batch_size).numpy()                                  # :360
        sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(num_mini_batch)]
        fields = self._fields(advantages)
        for indices in sampler:
            out = []
            for name, arr in fields:
                if arr is None:
                    out.append(None)
                    continue
                tail = arr.shape[3:]
                g = gather_rows(arr.reshape(batch_size, -1), indices)
                out.append(g.reshape((len(indices),) + tuple(tail)))
            yield tuple(out)

    # -- shared_buffer.py:499-608
    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):
        T, N, A = self.rewards.shape[0:3]
        batch_size = N * T * A
        data_chunks = batch_size // data_chunk_length                              # :508
        mini_batch_size = data_chunks // num_mini_batch
        rand = torch.randperm(data_chunks).numpy()                                 # :511
        sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(num_mini_batch)]
        fields = self._fields(advantages)
        for indices in sampler:
            out = []
            for name, arr in fields:
                if arr is None:
                    out.append(None)
                    continue
                first = name in ("rnn_states", "rnn_states_critic")
                out.append(gather_chunks(arr, indices, data_chunk_length, first_only=first))
            yield tuple(out)

    # -- shared_buffer.py:402-497
    def naive_recurrent_generator(self, advantages, num_mini_batch):
        T, N, A = self.rewards.shape[0:3]
        batch_size = N * A
        assert batch_size >= num_mini_batch
        per = batch_size // num_mini_batch
        perm = torch.randperm(batch_size).numpy()                                  # :415
        fields = self._fields(advantages)
        for start in range(0, batch_size, per):
            indices = perm[start:start + per]
            out = []
            for name, arr in fields:
                if arr is None:
                    out.append(None)
                    continue
                first = name in ("rnn_states", "rnn_states_critic")
                out.append(gather_chunks(arr, indices, T, first_only=first))
            yield tuple(out)


class OracleSeparatedBuffer(object):
    """Host restatement of the reference SeparatedReplayBuffer (separated_buffer.py:12-424): the
    shared layout with one agent, fields exposed without the agent axis.  Differences restated:
    the non-GAE proper-time-limits branch de-normalises only under PopArt (:146-151), the recurrent
    sampler emits chunk-major rows (:372-404), ``factor`` rides along as a 13th element (:62,:225)."""

    _VIEWS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns",
              "available_actions", "actions", "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")

    def __init__(self, args, obs_space, share_obs_space, act_space):
        self._inner = OracleBuffer(args, 1, obs_space, share_obs_space, act_space)
        self.factor = None

    def __getattr__(self, name):
        if name in self._VIEWS:
            a = getattr(self._inner, name)
            return None if a is None else a[:, :, 0]
        raise AttributeError(name)

    def update_factor(self, factor):                                    # :62-63
        self.factor = np.array(factor, dtype=np.float32)

    @property
    def step(self):
        return self._inner.step

    @staticmethod
    def _with_agent_axis(a, k):
        lift = lambda x: None if x is None else np.asarray(x)[:, None]
        return [lift(x) for x in a], {n: lift(x) for n, x in k.items()}

    def insert(self, *a, **k):                                          # :65-83
        a, k = self._with_agent_axis(a, k)
        self._inner.insert(*a, **k)

    def chooseinsert(self, *a, **k):                                    # :85-103
        a, k = self._with_agent_axis(a, k)
        self._inner.chooseinsert(*a, **k)

    def after_update(self):                                             # :105-114
        self._inner.after_update()

    def chooseafter_update(self):                                       # :116-120
        self._inner.chooseafter_update()

    def compute_returns(self, next_value, value_normalizer=None):       # :122-167
        b = self._inner
        sigma, mu, dn = b._scalars(value_normalizer)
        if not b._use_gae and b._use_proper_time_limits and not b._use_popart:
            sigma, mu, dn = 1.0, 0.0, False                             # :149-151
        ret, v = compute_returns(b.rewards, b.value_preds, np.asarray(next_value, dtype=np.float32), b.masks,
                                 b.bad_masks, sigma=sigma, mu=mu, gamma=b.gamma, gae_lambda=b.gae_lambda,
                                 use_gae=b._use_gae, use_proper_time_limits=b._use_proper_time_limits, denorm=dn)
        b.returns[...] = ret
        b.value_preds[...] = v

    def _with_factor(self, gen, rows_of):
        for sample in gen:
            if self.factor is None:
                yield sample
            else:
                yield sample + (rows_of(sample),)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):   # :169-227
        b = self._inner
        T, N = b.rewards.shape[0:2]
        adv = None if advantages is None else np.asarray(advantages, dtype=np.float32).reshape(T, N, 1, 1)
        fields = b._fields(adv)
        if self.factor is not None:
            fields = fields + [("factor", self.factor.reshape(T, N, 1, -1))]
        batch_size = N * T
        if mini_batch_size is None:
            assert batch_size >= num_mini_batch
            mini_batch_size = batch_size // num_mini_batch
        rand = torch.randperm(batch_size).numpy()
        for i in range(num_mini_batch):
            indices = rand[i * mini_batch_size:(i + 1) * mini_batch_size]
            yield tuple(None if arr is None else
                        gather_rows(arr.reshape(batch_size, -1), indices).reshape((len(indices),) + tuple(arr.shape[3:]))
                        for _, arr in fields)

    def _chunk_sampler(self, advantages, batches, L, chunk_major):
        b = self._inner
        T, N = b.rewards.shape[0:2]
        adv = None if advantages is None else np.asarray(advantages, dtype=np.float32).reshape(T, N, 1, 1)
        fields = b._fields(adv)
        if self.factor is not None:
            fields = fields + [("factor", self.factor.reshape(T, N, 1, -1))]
        for indices in batches:
            out = []
            for name, arr in fields:
                if arr is None:
                    out.append(None)
                    continue
                first = name in ("rnn_states", "rnn_states_critic")
                g = gather_chunks(arr, indices, L, first_only=first)          # rows l*mb + j
                if chunk_major and not first:                                  # :372-404 -> rows j*L + l
                    mb = len(indices)
                    g = np.ascontiguousarray(g.reshape((L, mb) + g.shape[1:]).swapaxes(0, 1)).reshape(g.shape)
                out.append(g)
            yield tuple(out)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):              # :315-424
        T, N = self._inner.rewards.shape[0:2]
        data_chunks = N * T // data_chunk_length
        mb = data_chunks // num_mini_batch
        rand = torch.randperm(data_chunks).numpy()
        batches = [rand[i * mb:(i + 1) * mb] for i in range(num_mini_batch)]
        return self._chunk_sampler(advantages, batches, data_chunk_length, True)

    def naive_recurrent_generator(self, advantages, num_mini_batch):                           # :229-313
        T, N = self._inner.rewards.shape[0:2]
        assert N >= num_mini_batch
        per = N // num_mini_batch
        perm = torch.randperm(N).numpy()
        batches = [perm[s:s + per] for s in range(0, N, per)]
        return self._chunk_sampler(advantages, batches, T, False)


def _shape_of(space):
    """onpolicy/utils/util.py:31-38 + shared_buffer.py:48-52."""
    name = space.__class__.__name__
    if name == 'Box':
        shape = space.shape
    elif name == 'list':
        shape = space
    else:
        raise NotImplementedError
    if type(shape[-1]) == list:
        shape = shape[:1]
    return shape


def _act_dim(act_space):
    """onpolicy/utils/util.py:40-52."""
    name = act_space.__class__.__name__
    if name == 'Discrete':
        return 1
    if name == 'MultiDiscrete':
        return act_space.shape
    if name in ('Box', 'MultiBinary'):
        return act_space.shape[0]
    return act_space[0].shape[0] + 1
