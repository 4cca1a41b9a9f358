"""``TransformerPolicy``: the policy object the runners drive when ``--algorithm_name mat / mat_dec`` -- interface
of the reference's onpolicy/algorithms/mat/algorithm/transformer_policy.py (:9-214): the per-agent rows
[batch * A, dim] of the runner become [batch, A, dim] sequences for the transformer and go back to rows.
RNN states are accepted and returned untouched (MAT has none)."""
import numpy as np
import torch

from onpolicy.algorithms.mat.algorithm.ma_transformer import MultiAgentTransformer
from onpolicy.algorithms.utils.util import check
from onpolicy.utils.util import get_shape_from_obs_space, update_linear_schedule


class TransformerPolicy(object):
    def __init__(self, args, obs_space, cent_obs_space, act_space, num_agents, device=torch.device("cpu")):
        self.device = device
        self.lr, self.opti_eps, self.weight_decay = args.lr, args.opti_eps, args.weight_decay
        self._use_policy_active_masks = args.use_policy_active_masks
        self.action_type = 'Continuous' if act_space.__class__.__name__ == 'Box' else 'Discrete'
        self.obs_dim = get_shape_from_obs_space(obs_space)[0]
        self.share_obs_dim = get_shape_from_obs_space(cent_obs_space)[0]
        if self.action_type == 'Discrete':
            self.act_dim, self.act_num = act_space.n, 1
        else:
            self.act_dim = self.act_num = act_space.shape[0]
        self.num_agents = num_agents
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.transformer = MultiAgentTransformer(
            self.share_obs_dim, self.obs_dim, self.act_dim, num_agents, n_block=args.n_block, n_embd=args.n_embd,
            n_head=args.n_head, encode_state=args.encode_state, device=device, action_type=self.action_type,
            dec_actor=args.dec_actor, share_actor=args.share_actor)
        if args.env_name == "hands":
            self.transformer.zero_std()
        fused = {"fused": True} if torch.device(device).type == "cuda" else {}
        self.optimizer = torch.optim.Adam(self.transformer.parameters(), lr=self.lr, eps=self.opti_eps,
                                          weight_decay=self.weight_decay, **fused)

    def lr_decay(self, episode, episodes):
        update_linear_schedule(self.optimizer, episode, episodes, self.lr)

    def _sequences(self, x, width):
        return None if x is None else check(x).reshape(-1, self.num_agents, width)

    def get_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, masks, available_actions=None,
                    deterministic=False):
        actions, action_log_probs, values = self.transformer.get_actions(
            self._sequences(cent_obs, self.share_obs_dim), self._sequences(obs, self.obs_dim),
            self._sequences(available_actions, self.act_dim), deterministic)
        return (values.view(-1, 1), actions.view(-1, self.act_num), action_log_probs.view(-1, self.act_num),
                check(rnn_states_actor).to(**self.tpdv), check(rnn_states_critic).to(**self.tpdv))

    def get_values(self, cent_obs, obs, rnn_states_critic, masks):
        return self.transformer.get_values(self._sequences(cent_obs, self.share_obs_dim),
                                           self._sequences(obs, self.obs_dim)).view(-1, 1)

    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, actions, masks,
                         available_actions=None, active_masks=None):
        action_log_probs, values, entropy = self.transformer(
            self._sequences(cent_obs, self.share_obs_dim), self._sequences(obs, self.obs_dim),
            self._sequences(actions, self.act_num), self._sequences(available_actions, self.act_dim))
        entropy = entropy.view(-1, self.act_num)
        if self._use_policy_active_masks and active_masks is not None:
            entropy = (entropy * active_masks).sum() / active_masks.sum()
        else:
            entropy = entropy.mean()
        return values.view(-1, 1), action_log_probs.view(-1, self.act_num), entropy

    def act(self, cent_obs, obs, rnn_states_actor, masks, available_actions=None, deterministic=True):
        rnn_states_critic = np.zeros_like(rnn_states_actor) if isinstance(rnn_states_actor, np.ndarray) \
            else torch.zeros_like(rnn_states_actor)
        _, actions, _, rnn_states_actor, _ = self.get_actions(cent_obs, obs, rnn_states_actor, rnn_states_critic,
                                                              masks, available_actions, deterministic)
        return actions, rnn_states_actor

    def save(self, save_dir, episode):
        torch.save(self.transformer.state_dict(), str(save_dir) + "/transformer_" + str(episode) + ".pt")

    def restore(self, model_dir):
        """``model_dir`` is the checkpoint FILE (transformer_<episode>.pt), as in the reference (:204-206)."""
        self.transformer.load_state_dict(torch.load(model_dir, map_location=self.device))

    def train(self):
        self.transformer.train()

    def eval(self):
        self.transformer.eval()
