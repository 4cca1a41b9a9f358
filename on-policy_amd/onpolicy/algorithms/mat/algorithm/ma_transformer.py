"""Multi-Agent Transformer (encoder over the agents' observations -> values; causal decoder over the agents'
actions -> logits / means).  Parameter names, shapes and construction order are those of the reference's
onpolicy/algorithms/mat/algorithm/ma_transformer.py (SelfAttention :20, EncodeBlock :64, DecodeBlock :86,
Encoder :110, Decoder :147, MultiAgentTransformer :222), so checkpoints are interchangeable and the same seed gives
the same initial weights.  What differs is how it is evaluated:

  * attention goes through ``scaled_dot_product_attention`` (one fused kernel per block on the GPU);
  * acting is incremental.  The reference re-runs the WHOLE decoder once per agent and keeps one row of the
    result (transformer_act.py:14-15: A decoder passes over A positions each).  The decoder is causal, so the
    row of agent i only depends on positions <= i: here every block keeps the keys / values of the positions
    already decoded and each agent costs one position (``Decoder.begin`` / ``_Incremental.logits``) -- the same
    numbers for 1/A of the work.

The reference feeds the critic's "state" branch zeros of width 37 (ma_transformer.py:236-237, :248-250); that
input carries no information, so ``encode_state`` runs the state encoder on a zero tensor exactly like it.
"""
import torch
import torch.nn as nn
from torch.nn import functional as F

from onpolicy.algorithms.utils.util import check, init
from onpolicy.algorithms.utils import transformer_act

_UNUSED_STATE_DIM = 37


def init_(m, gain=0.01, activate=False):
    if activate:
        gain = nn.init.calculate_gain('relu')
    return init(m, nn.init.orthogonal_, lambda x: nn.init.constant_(x, 0), gain=gain)


def _two_layer(n_embd):
    return nn.Sequential(init_(nn.Linear(n_embd, n_embd), activate=True), nn.GELU(), init_(nn.Linear(n_embd, n_embd)))


def _input_encoder(dim, n_embd):
    return nn.Sequential(nn.LayerNorm(dim), init_(nn.Linear(dim, n_embd), activate=True), nn.GELU())


def _head(n_embd, out_dim):
    return nn.Sequential(init_(nn.Linear(n_embd, n_embd), activate=True), nn.GELU(), nn.LayerNorm(n_embd),
                         init_(nn.Linear(n_embd, out_dim)))


def _actor_mlp(obs_dim, n_embd, action_dim):
    return nn.Sequential(nn.LayerNorm(obs_dim), init_(nn.Linear(obs_dim, n_embd), activate=True), nn.GELU(),
                         nn.LayerNorm(n_embd), init_(nn.Linear(n_embd, n_embd), activate=True), nn.GELU(),
                         nn.LayerNorm(n_embd), init_(nn.Linear(n_embd, action_dim)))


class SelfAttention(nn.Module):
    def __init__(self, n_embd, n_head, n_agent, masked=False):
        super(SelfAttention, self).__init__()
        assert n_embd % n_head == 0
        self.masked, self.n_head = masked, n_head
        for name in ("key", "query", "value", "proj"):          # construction order fixes the init random stream
            setattr(self, name, init_(nn.Linear(n_embd, n_embd)))
        # part of the reference's state dict; causality itself is handled by the attention kernel
        self.register_buffer("mask", torch.tril(torch.ones(n_agent + 1, n_agent + 1))
                             .view(1, 1, n_agent + 1, n_agent + 1))

    def _heads(self, x):
        B, L, D = x.shape
        return x.view(B, L, self.n_head, D // self.n_head).transpose(1, 2)          # [B, heads, L, D / heads]

    def keys_values(self, x):
        return self._heads(self.key(x)), self._heads(self.value(x))

    def attend(self, query, k, v, causal):
        """query [B, Lq, D] against projected keys / values [B, heads, Lk, .] -> [B, Lq, D]."""
        B, Lq, D = query.shape
        y = F.scaled_dot_product_attention(self._heads(self.query(query)), k, v, is_causal=causal)
        return self.proj(y.transpose(1, 2).reshape(B, Lq, D))

    def forward(self, key, value, query):
        k, v = self._heads(self.key(key)), self._heads(self.value(value))
        return self.attend(query, k, v, causal=self.masked and query.shape[1] > 1)


class EncodeBlock(nn.Module):
    def __init__(self, n_embd, n_head, n_agent):
        super(EncodeBlock, self).__init__()
        self.ln1, self.ln2 = nn.LayerNorm(n_embd), nn.LayerNorm(n_embd)
        self.attn = SelfAttention(n_embd, n_head, n_agent, masked=False)
        self.mlp = _two_layer(n_embd)

    def forward(self, x):
        """Post-norm residual block: agents attend to each other without a mask."""
        attended = self.ln1(x + self.attn(x, x, x))
        return self.ln2(attended + self.mlp(attended))


class DecodeBlock(nn.Module):
    def __init__(self, n_embd, n_head, n_agent):
        super(DecodeBlock, self).__init__()
        self.ln1, self.ln2, self.ln3 = (nn.LayerNorm(n_embd) for _ in range(3))
        self.attn1 = SelfAttention(n_embd, n_head, n_agent, masked=True)      # over the action tokens
        self.attn2 = SelfAttention(n_embd, n_head, n_agent, masked=True)      # encoder rows query the action stream
        self.mlp = _two_layer(n_embd)

    def forward(self, x, rep_enc):
        """All positions at once; both attentions are causal over the agent order."""
        tokens = self.ln1(x + self.attn1(x, x, x))
        mixed = self.ln2(rep_enc + self.attn2(key=tokens, value=tokens, query=rep_enc))
        return self.ln3(mixed + self.mlp(mixed))

    def step(self, x, rep_enc, cache):
        """One new position: x, rep_enc [B, 1, D]; ``cache`` holds this block's keys / values of the earlier ones."""
        def extend(name, attn, src):
            k, v = attn.keys_values(src)
            if name in cache:
                k, v = torch.cat([cache[name][0], k], 2), torch.cat([cache[name][1], v], 2)
            cache[name] = (k, v)
            return k, v
        tokens = self.ln1(x + self.attn1.attend(x, *extend("self", self.attn1, x), causal=False))
        mixed = self.ln2(rep_enc + self.attn2.attend(rep_enc, *extend("cross", self.attn2, tokens), causal=False))
        return self.ln3(mixed + self.mlp(mixed))


class Encoder(nn.Module):
    def __init__(self, state_dim, obs_dim, n_block, n_embd, n_head, n_agent, encode_state):
        super(Encoder, self).__init__()
        self.state_dim, self.obs_dim, self.n_embd, self.n_agent = state_dim, obs_dim, n_embd, n_agent
        self.encode_state = encode_state
        self.state_encoder = _input_encoder(state_dim, n_embd)
        self.obs_encoder = _input_encoder(obs_dim, n_embd)
        self.ln = nn.LayerNorm(n_embd)
        self.blocks = nn.Sequential(*[EncodeBlock(n_embd, n_head, n_agent) for _ in range(n_block)])
        self.head = _head(n_embd, 1)

    def forward(self, state, obs):
        x = self.state_encoder(state) if self.encode_state else self.obs_encoder(obs)
        rep = self.blocks(self.ln(x))
        return self.head(rep), rep


class _Incremental(object):
    """Decoder state while the agents act one after the other (see the module docstring)."""

    def __init__(self, decoder, obs_rep, obs):
        self.decoder, self.obs_rep = decoder, obs_rep
        self.caches = [{} for _ in decoder.blocks] if not decoder.dec_actor else None
        self.fixed = decoder(None, obs_rep, obs) if decoder.dec_actor else None     # logits do not depend on actions

    def logits(self, i, shifted_action_i):
        """Output row of agent i given its input token [B, width] (agents 0 .. i-1 must have been decoded)."""
        if self.fixed is not None:
            return self.fixed[:, i, :]
        d = self.decoder
        x = d.ln(d.action_encoder(shifted_action_i.unsqueeze(1)))
        rep_i = self.obs_rep[:, i:i + 1, :]
        for block, cache in zip(d.blocks, self.caches):
            x = block.step(x, rep_i, cache)
        return d.head(x)[:, 0, :]


class Decoder(nn.Module):
    def __init__(self, obs_dim, action_dim, n_block, n_embd, n_head, n_agent,
                 action_type='Discrete', dec_actor=False, share_actor=False):
        super(Decoder, self).__init__()
        self.action_dim, self.n_embd = action_dim, n_embd
        self.dec_actor, self.share_actor, self.action_type = dec_actor, share_actor, action_type
        if action_type != 'Discrete':
            self.log_std = torch.nn.Parameter(torch.ones(action_dim))
        if self.dec_actor:
            if self.share_actor:
                self.mlp = _actor_mlp(obs_dim, n_embd, action_dim)
            else:
                self.mlp = nn.ModuleList([_actor_mlp(obs_dim, n_embd, action_dim) for _ in range(n_agent)])
        else:
            if action_type == 'Discrete':      # token = [start flag | one-hot of the previous agent's action]
                self.action_encoder = nn.Sequential(
                    init_(nn.Linear(action_dim + 1, n_embd, bias=False), activate=True), nn.GELU())
            else:
                self.action_encoder = nn.Sequential(init_(nn.Linear(action_dim, n_embd), activate=True), nn.GELU())
            self.obs_encoder = _input_encoder(obs_dim, n_embd)       # unused by forward, as in the reference
            self.ln = nn.LayerNorm(n_embd)
            self.blocks = nn.Sequential(*[DecodeBlock(n_embd, n_head, n_agent) for _ in range(n_block)])
            self.head = _head(n_embd, action_dim)

    def zero_std(self, device):
        if self.action_type != 'Discrete':
            self.log_std.data = torch.zeros(self.action_dim).to(device)

    def forward(self, action, obs_rep, obs):
        """All positions at once (training): shifted actions [B, A, width] -> logits / means [B, A, action_dim]."""
        if self.dec_actor:
            if self.share_actor:
                return self.mlp(obs)
            return torch.stack([actor(obs[:, n, :]) for n, actor in enumerate(self.mlp)], dim=1)
        x = self.ln(self.action_encoder(action))
        for block in self.blocks:
            x = block(x, obs_rep)
        return self.head(x)

    def begin(self, obs_rep, obs):
        return _Incremental(self, obs_rep, obs)


class MultiAgentTransformer(nn.Module):
    def __init__(self, state_dim, obs_dim, action_dim, n_agent, n_block, n_embd, n_head, encode_state=False,
                 device=torch.device("cpu"), action_type='Discrete', dec_actor=False, share_actor=False):
        super(MultiAgentTransformer, self).__init__()
        self.n_agent, self.action_dim, self.action_type = n_agent, action_dim, action_type
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.device = device
        self.encoder = Encoder(_UNUSED_STATE_DIM, obs_dim, n_block, n_embd, n_head, n_agent, encode_state)
        self.decoder = Decoder(obs_dim, action_dim, n_block, n_embd, n_head, n_agent, self.action_type,
                               dec_actor=dec_actor, share_actor=share_actor)
        self.to(device)

    def zero_std(self):
        if self.action_type != 'Discrete':
            self.decoder.zero_std(self.device)

    def _encode(self, obs):
        obs = check(obs).to(**self.tpdv)
        state = obs.new_zeros(*obs.shape[:-1], _UNUSED_STATE_DIM) if self.encoder.encode_state else None
        values, rep = self.encoder(state, obs)
        return obs, values, rep

    def forward(self, state, obs, action, available_actions=None):
        """(log-probs, values, entropies) of the given joint actions [B, A, act] (teacher forcing)."""
        obs, v_loc, obs_rep = self._encode(obs)
        action = check(action).to(**self.tpdv)
        if available_actions is not None:
            available_actions = check(available_actions).to(**self.tpdv)
        B = obs.shape[0]
        if self.action_type == 'Discrete':
            action_log, entropy = transformer_act.discrete_parallel_act(
                self.decoder, obs_rep, obs, action.long(), B, self.n_agent, self.action_dim, self.tpdv, available_actions)
        else:
            action_log, entropy = transformer_act.continuous_parallel_act(
                self.decoder, obs_rep, obs, action, B, self.n_agent, self.action_dim, self.tpdv)
        return action_log, v_loc, entropy

    def get_actions(self, state, obs, available_actions=None, deterministic=False):
        obs, v_loc, obs_rep = self._encode(obs)
        if available_actions is not None:
            available_actions = check(available_actions).to(**self.tpdv)
        B = obs.shape[0]
        if self.action_type == "Discrete":
            output_action, output_action_log = transformer_act.discrete_autoregreesive_act(
                self.decoder, obs_rep, obs, B, self.n_agent, self.action_dim, self.tpdv, available_actions, deterministic)
        else:
            output_action, output_action_log = transformer_act.continuous_autoregreesive_act(
                self.decoder, obs_rep, obs, B, self.n_agent, self.action_dim, self.tpdv, deterministic)
        return output_action, output_action_log, v_loc

    def get_values(self, state, obs):
        return self._encode(obs)[1]
