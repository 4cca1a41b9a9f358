"""Action heads of the Multi-Agent Transformer: the four entry points of the reference's
onpolicy/algorithms/utils/transformer_act.py (discrete / continuous x autoregressive acting / parallel
evaluation; the reference's spelling of the function names is kept, they are the module's interface).

The decoder's input token for agent i is the previous agent's action (one-hot behind a leading start flag for
Discrete heads, the raw action vector for Box heads; agent 0 sees the start token / zeros).  Both kinds of head run
through the same two routines below, which differ only in their ``_Head``: how a token is built from an action and
which distribution the decoder's output row parameterises.  Acting asks the decoder for one agent's row at a time:
decoders that offer ``begin(obs_rep, obs)`` (ours) are stepped incrementally with cached keys / values, any other
callable ``decoder(tokens, obs_rep, obs)`` is re-run on the whole sequence per agent like the reference does.  The
random draws -- one sample of batch size B per agent, in agent order -- are the reference's, so a seed gives the
same actions.
"""
import torch
from torch.distributions import Categorical, Normal
from torch.nn import functional as F

_UNAVAILABLE = -1e10


class _DiscreteHead(object):
    def __init__(self, action_dim, available_actions):
        self.action_dim, self.available = action_dim, available_actions
        self.token_width, self.action_width, self.action_dtype = action_dim + 1, 1, torch.long

    def start(self, tokens):
        tokens[:, 0, 0] = 1                                         # start flag of the first agent

    def token_of(self, action):                                     # action: [..] int64 -> [.., action_dim]
        return F.one_hot(action, num_classes=self.action_dim).to(torch.float32)

    def write_token(self, tokens, i, action):
        tokens[:, i, 1:] = self.token_of(action)

    def dist(self, decoder, rows, agent=None):
        if self.available is not None:
            mask = self.available if agent is None else self.available[:, agent, :]
            rows = rows.masked_fill(mask == 0, _UNAVAILABLE)
        return Categorical(logits=rows)

    def mode(self, dist):
        return dist.probs.argmax(dim=-1)

    def store(self, dst, i, value):
        dst[:, i, 0] = value


class _GaussianHead(object):
    def __init__(self, action_dim):
        self.token_width = self.action_width = action_dim
        self.action_dtype = torch.float32

    def start(self, tokens):
        pass                                                         # agent 0 sees zeros

    def write_token(self, tokens, i, action):
        tokens[:, i, :] = action

    def dist(self, decoder, rows, agent=None):
        return Normal(rows, torch.sigmoid(decoder.log_std) * 0.5)   # std in (0, 0.5), shared by all agents

    def mode(self, dist):
        return dist.mean

    def store(self, dst, i, value):
        dst[:, i, :] = value


def _act(head, decoder, obs_rep, obs, batch_size, n_agent, tpdv, deterministic):
    tokens = torch.zeros((batch_size, n_agent, head.token_width)).to(**tpdv)
    head.start(tokens)
    dev = tokens.device
    actions = torch.zeros((batch_size, n_agent, head.action_width), dtype=head.action_dtype, device=dev)
    log_probs = torch.zeros((batch_size, n_agent, head.action_width), dtype=torch.float32, device=dev)
    if hasattr(decoder, "begin"):
        state = decoder.begin(obs_rep, obs)
        row_of = lambda i: state.logits(i, tokens[:, i, :])                    # noqa: E731
    else:
        row_of = lambda i: decoder(tokens, obs_rep, obs)[:, i, :]              # noqa: E731
    for i in range(n_agent):
        dist = head.dist(decoder, row_of(i), agent=i)
        action = head.mode(dist) if deterministic else dist.sample()
        head.store(actions, i, action)
        head.store(log_probs, i, dist.log_prob(action))
        if i + 1 < n_agent:
            head.write_token(tokens, i + 1, action)
    return actions, log_probs


def _evaluate(head, decoder, obs_rep, obs, action, batch_size, n_agent, tpdv):
    """Teacher forcing: every agent's token is the GIVEN action of the agent before it -> one decoder pass."""
    tokens = torch.zeros((batch_size, n_agent, head.token_width)).to(**tpdv)
    head.start(tokens)
    taken = action.squeeze(-1) if head.action_dtype == torch.long else action
    if n_agent > 1:
        shifted = taken[:, :-1]
        if head.action_dtype == torch.long:
            tokens[:, 1:, 1:] = head.token_of(shifted)
        else:
            tokens[:, 1:, :] = shifted
    dist = head.dist(decoder, decoder(tokens, obs_rep, obs))
    log_prob, entropy = dist.log_prob(taken), dist.entropy()
    if head.action_dtype == torch.long:
        return log_prob.unsqueeze(-1), entropy.unsqueeze(-1)
    return log_prob, entropy


def discrete_autoregreesive_act(decoder, obs_rep, obs, batch_size, n_agent, action_dim, tpdv,
                                available_actions=None, deterministic=False):
    return _act(_DiscreteHead(action_dim, available_actions), decoder, obs_rep, obs, batch_size, n_agent, tpdv,
                deterministic)


def discrete_parallel_act(decoder, obs_rep, obs, action, batch_size, n_agent, action_dim, tpdv,
                          available_actions=None):
    return _evaluate(_DiscreteHead(action_dim, available_actions), decoder, obs_rep, obs, action, batch_size, n_agent,
                     tpdv)


def continuous_autoregreesive_act(decoder, obs_rep, obs, batch_size, n_agent, action_dim, tpdv,
                                  deterministic=False):
    return _act(_GaussianHead(action_dim), decoder, obs_rep, obs, batch_size, n_agent, tpdv, deterministic)


def continuous_parallel_act(decoder, obs_rep, obs, action, batch_size, n_agent, action_dim, tpdv):
    return _evaluate(_GaussianHead(action_dim), decoder, obs_rep, obs, action, batch_size, n_agent, tpdv)
