"""Action heads of the Multi-Agent Transformer: the four entry points of the reference's
onpolicy/algorithms/utils/transformer_act.py (discrete / continuous x autoregressive acting / parallel
evaluation; the reference's spelling of the function names is kept, they are the module's interface).

The decoder's input token for agent i is the previous agent's action (one-hot with a leading start flag for
Discrete heads, the raw action vector for Box heads; agent 0 sees the start token / zeros).  Acting asks the
decoder for one agent's output at a time: decoders that offer ``begin(obs_rep, obs)`` (ours) are stepped
incrementally, any other callable ``decoder(shifted_action, obs_rep, obs)`` is re-run on the whole sequence per
agent like the reference does.  The random draws (one ``Categorical`` / ``Normal`` sample of batch size B per agent,
in agent order) are the reference's, so a seed gives the same actions.
"""
import torch
from torch.distributions import Categorical, Normal
from torch.nn import functional as F


def _row_source(decoder, obs_rep, obs, shifted):
    """-> f(i) giving the decoder's output row of agent i for the tokens written into ``shifted`` so far."""
    if hasattr(decoder, "begin"):
        state = decoder.begin(obs_rep, obs)
        return lambda i: state.logits(i, shifted[:, i, :])
    return lambda i: decoder(shifted, obs_rep, obs)[:, i, :]


def _action_std(decoder):
    return torch.sigmoid(decoder.log_std) * 0.5


def discrete_autoregreesive_act(decoder, obs_rep, obs, batch_size, n_agent, action_dim, tpdv,
                                available_actions=None, deterministic=False):
    shifted = torch.zeros((batch_size, n_agent, action_dim + 1)).to(**tpdv)
    shifted[:, 0, 0] = 1
    actions = torch.zeros((batch_size, n_agent, 1), dtype=torch.long, device=shifted.device)
    log_probs = torch.zeros((batch_size, n_agent, 1), dtype=torch.float32, device=shifted.device)
    row = _row_source(decoder, obs_rep, obs, shifted)
    for i in range(n_agent):
        logit = row(i)
        if available_actions is not None:
            logit = logit.masked_fill(available_actions[:, i, :] == 0, -1e10)
        dist = Categorical(logits=logit)
        action = dist.probs.argmax(dim=-1) if deterministic else dist.sample()
        actions[:, i, 0] = action
        log_probs[:, i, 0] = dist.log_prob(action)
        if i + 1 < n_agent:
            shifted[:, i + 1, 1:] = F.one_hot(action, num_classes=action_dim)
    return actions, log_probs


def discrete_parallel_act(decoder, obs_rep, obs, action, batch_size, n_agent, action_dim, tpdv,
                          available_actions=None):
    one_hot = F.one_hot(action.squeeze(-1), num_classes=action_dim)
    shifted = torch.zeros((batch_size, n_agent, action_dim + 1)).to(**tpdv)
    shifted[:, 0, 0] = 1
    shifted[:, 1:, 1:] = one_hot[:, :-1, :]
    logit = decoder(shifted, obs_rep, obs)
    if available_actions is not None:
        logit = logit.masked_fill(available_actions == 0, -1e10)
    dist = Categorical(logits=logit)
    return dist.log_prob(action.squeeze(-1)).unsqueeze(-1), dist.entropy().unsqueeze(-1)


def continuous_autoregreesive_act(decoder, obs_rep, obs, batch_size, n_agent, action_dim, tpdv,
                                  deterministic=False):
    shifted = torch.zeros((batch_size, n_agent, action_dim)).to(**tpdv)
    actions = torch.zeros((batch_size, n_agent, action_dim), dtype=torch.float32, device=shifted.device)
    log_probs = torch.zeros_like(actions)
    row = _row_source(decoder, obs_rep, obs, shifted)
    for i in range(n_agent):
        mean = row(i)
        dist = Normal(mean, _action_std(decoder))
        action = mean if deterministic else dist.sample()
        actions[:, i, :] = action
        log_probs[:, i, :] = dist.log_prob(action)
        if i + 1 < n_agent:
            shifted[:, i + 1, :] = action
    return actions, log_probs


def continuous_parallel_act(decoder, obs_rep, obs, action, batch_size, n_agent, action_dim, tpdv):
    shifted = torch.zeros((batch_size, n_agent, action_dim)).to(**tpdv)
    shifted[:, 1:, :] = action[:, :-1, :]
    dist = Normal(decoder(shifted, obs_rep, obs), _action_std(decoder))
    return dist.log_prob(action), dist.entropy()
