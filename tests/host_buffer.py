"""TEST INFRASTRUCTURE: a host (CPU) stand-in for the HBM ``SharedReplayBuffer`` so that the runners' host logic
can be exercised -- and compared with the reference's runners -- without a GPU.  Storage, returns and samplers are
the oracle's (oracle/oracle.py: the plain-C / numpy restatement pinned to the reference); the fields are exposed as
torch CPU tensors that share memory with the oracle's numpy arrays, which is the surface the runners use.
Never imported by anything under on-policy_amd/."""
import numpy as np
import torch

from oracle import oracle

_FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "advantages",
           "available_actions", "actions", "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")


def _np(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


class HostSharedBuffer(object):
    def __init__(self, args, num_agents, obs_space, cent_obs_space, act_space, device=None):
        self._o = oracle.OracleBuffer(args, num_agents, obs_space, cent_obs_space, act_space)
        self.device = torch.device("cpu")
        self.episode_length, self.n_rollout_threads, self.num_agents = args.episode_length, args.n_rollout_threads, num_agents
        for name in _FIELDS:
            arr = getattr(self._o, name)
            setattr(self, name, None if arr is None else torch.from_numpy(arr))

    @property
    def step(self):
        return self._o.step

    def insert(self, *a, **k):
        self._o.insert(*[_np(x) for x in a], **{n: _np(x) for n, x in k.items()})

    def chooseinsert(self, *a, **k):
        self._o.chooseinsert(*[_np(x) for x in a], **{n: _np(x) for n, x in k.items()})

    def after_update(self):
        self._o.after_update()

    def chooseafter_update(self):
        self._o.chooseafter_update()

    def compute_returns(self, next_value, value_normalizer=None):
        nv = _np(next_value).reshape(self.n_rollout_threads, self.num_agents, 1)
        self._o.compute_returns(nv, value_normalizer)

    def _wrap(self, gen):
        for sample in gen:
            yield tuple(None if x is None else torch.from_numpy(np.ascontiguousarray(x)) for x in sample)

    def feed_forward_generator_transformer(self, advantages, num_mini_batch=None, mini_batch_size=None):
        return self._wrap(self._o.feed_forward_generator_transformer(_np(advantages), num_mini_batch, mini_batch_size))

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        return self._wrap(self._o.feed_forward_generator(_np(advantages), num_mini_batch, mini_batch_size))

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):
        return self._wrap(self._o.recurrent_generator(_np(advantages), num_mini_batch, data_chunk_length))

    def naive_recurrent_generator(self, advantages, num_mini_batch):
        return self._wrap(self._o.naive_recurrent_generator(_np(advantages), num_mini_batch))


class HostSeparatedBuffer(object):
    """The same stand-in for the per-agent ``SeparatedReplayBuffer`` (oracle.OracleSeparatedBuffer underneath)."""

    def __init__(self, args, obs_space, share_obs_space, act_space, device=None):
        self._o = oracle.OracleSeparatedBuffer(args, obs_space, share_obs_space, act_space)
        self.device = torch.device("cpu")
        self.episode_length, self.n_rollout_threads = args.episode_length, args.n_rollout_threads
        for name in oracle.OracleSeparatedBuffer._VIEWS:
            arr = getattr(self._o, name)
            setattr(self, name, None if arr is None else torch.from_numpy(arr))
        self.factor = None

    @property
    def step(self):
        return self._o.step

    def update_factor(self, factor):
        self._o.update_factor(_np(factor))
        self.factor = torch.from_numpy(self._o.factor)

    def insert(self, *a, **k):
        self._o.insert(*[_np(x) for x in a], **{n: _np(x) for n, x in k.items()})

    def after_update(self):
        self._o.after_update()

    def compute_returns(self, next_value, value_normalizer=None):
        self._o.compute_returns(_np(next_value).reshape(self.n_rollout_threads, 1), value_normalizer)

    _wrap = HostSharedBuffer._wrap

    def feed_forward_generator_transformer(self, advantages, num_mini_batch=None, mini_batch_size=None):
        return self._wrap(self._o.feed_forward_generator_transformer(_np(advantages), num_mini_batch, mini_batch_size))

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        return self._wrap(self._o.feed_forward_generator(_np(advantages), num_mini_batch, mini_batch_size))

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):
        return self._wrap(self._o.recurrent_generator(_np(advantages), num_mini_batch, data_chunk_length))

    def naive_recurrent_generator(self, advantages, num_mini_batch):
        return self._wrap(self._o.naive_recurrent_generator(_np(advantages), num_mini_batch))
