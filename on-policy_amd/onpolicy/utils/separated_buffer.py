"""HBM-resident per-agent rollout buffer (``share_policy=False`` runners).

Drop-in for the reference's ``SeparatedReplayBuffer`` (onpolicy/utils/separated_buffer.py:12-424):
the same storage without the agent axis -- fields are ``[episode_length(+1), n_rollout_threads, dim]``
-- plus the HAPPO ``factor`` that the samplers append as a 13th element when it is set
(separated_buffer.py:62-63,197-227).

It is the shared buffer with ``num_agents = 1``: every attribute below is a view of the inner
``SharedReplayBuffer``'s ``[T(+1), N, 1, dim]`` tensors with the agent axis squeezed, so all storage,
GAE and sampling go through the same HIP kernels (time-major rows are ``t*N + n``).  Differences from
the shared buffer that the reference has and this class reproduces:
  * the discounted-return branch with proper time limits de-normalises only under PopArt, not under
    ValueNorm (separated_buffer.py:146-151 vs shared_buffer.py:207-211);
  * the recurrent generator emits CHUNK-major rows: the reference stacks the chunks on axis 0 and then
    reshapes [mb, L, dim] to [L*mb, dim] (separated_buffer.py:372-404, ``np.stack(x)`` where the shared
    buffer has ``np.stack(x, axis=1)``), so output row ``j*L + l`` holds step ``l`` of chunk ``j``.
    Reproduced as is (the parity fixtures pin it); it is a row gather over indices computed on the
    device;
  * ``factor`` rides along in the minibatches as a 13th element.
"""
import numpy as np
import torch

from onpolicy.utils.shared_buffer import SharedReplayBuffer

_VIEWS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "advantages",
          "available_actions", "actions", "action_log_probs", "rewards", "masks", "bad_masks", "active_masks")


class SeparatedReplayBuffer(object):
    def __init__(self, args, obs_space, share_obs_space, act_space, device=None):
        self._inner = SharedReplayBuffer(args, 1, obs_space, share_obs_space, act_space, device=device)
        self.episode_length = args.episode_length
        self.n_rollout_threads = args.n_rollout_threads
        self.rnn_hidden_size = args.hidden_size
        self.recurrent_N = args.recurrent_N
        self.gamma = args.gamma
        self.gae_lambda = args.gae_lambda
        self._use_gae = args.use_gae
        self._use_popart = args.use_popart
        self._use_valuenorm = args.use_valuenorm
        self._use_proper_time_limits = args.use_proper_time_limits
        self.device = self._inner.device
        self.factor = None

    def __getattr__(self, name):
        # views of the inner [T(+1), N, 1, ...] tensors without the agent axis
        if name in _VIEWS:
            t = getattr(self._inner, name)
            return None if t is None else t[:, :, 0]
        raise AttributeError(name)

    @property
    def step(self):
        return self._inner.step

    supports_standardized_obs = True

    def can_standardize_obs(self):
        return self._inner.can_standardize_obs()

    def update_factor(self, factor):
        """HAPPO's running product of the other agents' probability ratios, [T, N, k]."""
        f = self._inner._dev(factor).reshape(self.episode_length, self.n_rollout_threads, 1, -1).clone()
        self.factor = f[:, :, 0]
        self._inner.extra_fields["factor"] = f
        self._inner._content_version += 1     # packed sampler records are stale

    # -- storage: the inner buffer only checks element counts, so [N, ...] slabs go straight through
    def insert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds,
               rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        self._inner.insert(share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds,
                           rewards, masks, bad_masks, active_masks, available_actions)

    def chooseinsert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds,
                     rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        self._inner.chooseinsert(share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs,
                                 value_preds, rewards, masks, bad_masks, active_masks, available_actions)

    def after_update(self):
        self._inner.after_update()

    def chooseafter_update(self):
        self._inner.chooseafter_update()

    def compute_returns(self, next_value, value_normalizer=None):
        """reference separated_buffer.py:122-167."""
        scan_denorm = True
        if not self._use_gae and self._use_proper_time_limits and not self._use_popart:
            scan_denorm = False                      # :146-151: only PopArt de-normalises in this branch
        self._inner.compute_returns(next_value, value_normalizer, _scan_denorm=scan_denorm)

    def normalized_advantages(self, value_normalizer=None, all_reduce=None, denormalize=None):
        return self._inner.normalized_advantages(value_normalizer, all_reduce, denormalize)

    def plan_epochs(self, n_epochs):
        self._inner.plan_epochs(n_epochs)

    # -- samplers: 12-tuples, or 13-tuples ending in factor once update_factor() has been called
    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, standardize_obs=False):
        return self._inner.feed_forward_generator(self._adv(advantages), num_mini_batch, mini_batch_size,
                                                  standardize_obs=standardize_obs)

    def naive_recurrent_generator(self, advantages, num_mini_batch, standardize_obs=False):
        return self._inner.naive_recurrent_generator(self._adv(advantages), num_mini_batch,
                                                     standardize_obs=standardize_obs)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, standardize_obs=False):
        """reference separated_buffer.py:315-424.  Sequence fields [mb*L, dim] with row j*L + l (see the
        module docstring), RNN states [mb, R, H] of each chunk's first step."""
        inner, T, N, L = self._inner, self.episode_length, self.n_rollout_threads, data_chunk_length
        batch_size = N * T
        data_chunks = batch_size // L
        mb = data_chunks // num_mini_batch
        assert batch_size >= L, (
            "PPO requires the number of processes ({}) * episode length ({}) "
            "to be greater than or equal to the number of "
            "data chunk length ({}).".format(N, T, L))
        assert data_chunks >= 2, ("need larger batch size")
        rand = inner._sampler_indices(data_chunks, mb, num_mini_batch)
        table, stats = inner._field_table(self._adv(advantages))
        packed = inner._pack_records(table)
        seq_table = [(name, None if is_state else src, is_state) for name, src, is_state in table]
        state_table = [(name, src if is_state else None, is_state) for name, src, is_state in table]
        steps = torch.arange(L, device=self.device)
        for i in range(num_mini_batch):
            idx = rand[i * mb:(i + 1) * mb]
            flat = (idx[:, None] * L + steps[None, :]).reshape(-1)        # position in the [N, T] order
            rows = ((flat % T) * N + flat // T).contiguous()              # row in the time-major buffer
            seq = inner._gather(seq_table, stats, rows, mb * L, standardize_obs=standardize_obs, packed=packed)
            first = rows[::L].contiguous()
            states = inner._gather(state_table, None, first, mb)
            yield tuple(st if is_state else sq for sq, st, (_, _, is_state) in zip(seq, states, table))

    def _adv(self, advantages):
        """[T, N, 1] arrays / tensors -> the inner buffer's [T, N, 1, 1]; handles pass through."""
        if advantages is None or not (isinstance(advantages, np.ndarray) or torch.is_tensor(advantages)):
            return advantages
        return self._inner._dev(advantages).reshape(self.episode_length, self.n_rollout_threads, 1, 1)
