"""Policy wrapper: one shared actor, one (centralised) critic, one Adam optimiser each.

Surface of the reference's onpolicy/algorithms/r_mappo/algorithm/rMAPPOPolicy.py (R_MAPPOPolicy :6,
lr_decay :39, get_actions :48, get_values :76, evaluate_actions :88, act :116).
"""
import torch

from onpolicy.algorithms.r_mappo.algorithm.r_actor_critic import R_Actor, R_Critic
from onpolicy.utils.util import update_linear_schedule


class R_MAPPOPolicy:
    def __init__(self, args, obs_space, cent_obs_space, act_space, device=torch.device("cpu")):
        self.device = device
        self.lr = args.lr
        self.critic_lr = args.critic_lr
        self.opti_eps = args.opti_eps
        self.weight_decay = args.weight_decay
        self.obs_space = obs_space
        self.share_obs_space = cent_obs_space
        self.act_space = act_space

        self.actor = R_Actor(args, self.obs_space, self.act_space, self.device)
        self.critic = R_Critic(args, self.share_obs_space, self.device)
        self.actor_optimizer = self._adam(self.actor, self.lr)
        self.critic_optimizer = self._adam(self.critic, self.critic_lr)

    def _adam(self, net, lr):
        # reference rMAPPOPolicy.py:31-37: torch.optim.Adam(lr, eps=opti_eps, weight_decay).  On the GPU the
        # single-kernel ("fused") implementation of the same update replaces ~14 launches per step
        fused = {"fused": True} if torch.device(self.device).type == "cuda" else {}
        return torch.optim.Adam(net.parameters(), lr=lr, eps=self.opti_eps, weight_decay=self.weight_decay, **fused)

    def set_matrix_arithmetic(self, name):
        """"six_term" | "f32_mfma": the arithmetic of the K9 / K12 matrix products of THIS policy's networks from the next
        call on (``--matrix_arithmetic``; carried per call in the ``arith`` field of the C ABI structs, so several policies
        in one process may differ)."""
        from onpolicy.algorithms.utils import fused_mlp
        for net in (self.actor, self.critic):
            fused_mlp.set_matrix_arithmetic(net, name)

    def lr_decay(self, episode, episodes):
        update_linear_schedule(self.actor_optimizer, episode, episodes, self.lr)
        update_linear_schedule(self.critic_optimizer, episode, episodes, self.critic_lr)

    def get_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, masks, available_actions=None,
                    deterministic=False):
        """-> (values, actions, action_log_probs, rnn_states_actor, rnn_states_critic)."""
        actions, action_log_probs, rnn_states_actor = self.actor(obs, rnn_states_actor, masks,
                                                                 available_actions, deterministic)
        values, rnn_states_critic = self.critic(cent_obs, rnn_states_critic, masks)
        return values, actions, action_log_probs, rnn_states_actor, rnn_states_critic

    def get_values(self, cent_obs, rnn_states_critic, masks):
        return self.critic(cent_obs, rnn_states_critic, masks)[0]

    def can_fold_input_norm(self):
        """Both trunks can take row-standardised observations (see MLPBase.forward)."""
        bases = (self.actor.base, self.critic.base)
        return all(hasattr(b, "can_fold_input_norm") and b.can_fold_input_norm() for b in bases)

    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks,
                         available_actions=None, active_masks=None, obs_standardized=False):
        """-> (values, action_log_probs, dist_entropy).  ``obs_standardized``: cent_obs / obs rows were
        standardised by the sampler (SharedReplayBuffer generators, standardize_obs=True)."""
        action_log_probs, dist_entropy = self.actor.evaluate_actions(obs, rnn_states_actor, action, masks,
                                                                     available_actions, active_masks,
                                                                     obs_standardized=obs_standardized)
        values = self.critic(cent_obs, rnn_states_critic, masks, obs_standardized=obs_standardized)[0]
        return values, action_log_probs, dist_entropy

    def evaluate_logits(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, masks, obs_standardized=False,
                        need_actor=True):
        """-> (values, logits of the Discrete action head or None): the inputs of the fused PPO loss.

        Evaluations of at most MAPPO_TWO_STREAM_MAX_ROWS rows (default 2^20: the minibatches whose ``ppo_update`` is replayed
        from a HIP graph, update_graph.py, and the row spans a larger hidden-512 minibatch is cut into) run the critic on a
        SIDE STREAM next to the actor (MAPPO_TWO_STREAM_UPDATE=0: one stream).  The two networks share nothing until the loss
        kernel.  On such sizes one network's launches occupy a fraction of the chip (the 64-threads-per-GPU shard of BASELINE
        configs[3]: K12 is one 32-chunk tile per wave on 400 of 1 024 SIMDs) or leave tails the other network's fill.
        Autograd runs every backward node on its forward's stream, so the backward passes overlap the same way; captured into
        the update graph the fork / join become two branches of the graph.  Same kernels on the same data: bit-identical to
        the one-stream order (tests/test_gpu_update_graph.py).  Measured, alternating on one box
        (profiles/r06_ab_two_streams.json): SMAC shard 16.3 -> 13.8 ms per step, 128-thread recurrent north-star shard
        24.7 -> 21.7, SMAC shapes at 512 threads 64.7 -> 61.4, Hanabi shapes (six spans of 683 k rows) 4.70 -> 4.57-4.64 s,
        configs[1] unchanged.  One evaluation of more rows (north star, config 3: 13.1 M / 4.9 M rows) stays on one stream:
        -0.7 % at best, and launches that share the chip cannot be timed against a roofline -- bench.py times the K9 / K15
        launches of workloads that do use the side stream in one extra one-stream step."""
        side = self._critic_stream(masks) if need_actor else None
        if side is None:
            logits = self.actor.evaluate_logits(obs, rnn_states_actor, masks, obs_standardized=obs_standardized) \
                if need_actor else None
            values = self.critic(cent_obs, rnn_states_critic, masks, obs_standardized=obs_standardized)[0]
            return values, logits
        main = torch.cuda.current_stream(masks.device)
        side.wait_stream(main)                  # the minibatch (index lists, masks, RNN states) is ready
        with torch.cuda.stream(side):
            values = self.critic(cent_obs, rnn_states_critic, masks, obs_standardized=obs_standardized)[0]
        logits = self.actor.evaluate_logits(obs, rnn_states_actor, masks, obs_standardized=obs_standardized)
        main.wait_stream(side)                  # the loss kernel reads both
        return values, logits

    def _critic_stream(self, masks):
        """The side stream of ``evaluate_logits`` for this minibatch, or None (CPU tensors, no autograd,
        MAPPO_TWO_STREAM_UPDATE=0, more rows than MAPPO_TWO_STREAM_MAX_ROWS)."""
        import os
        if not (torch.is_tensor(masks) and masks.is_cuda and torch.is_grad_enabled()) \
                or os.environ.get("MAPPO_TWO_STREAM_UPDATE", "1") == "0" \
                or masks.shape[0] > int(os.environ.get("MAPPO_TWO_STREAM_MAX_ROWS", str(1 << 20))):
            return None
        # (only under the six-term arithmetic, whose matrix products are in-tree kernels made of independent workgroups: the
        # float32 route goes through library GEMMs, and nothing says those tolerate a second GEMM taking CUs away under them)
        from onpolicy import _native
        from onpolicy.algorithms.utils.fused_mlp import matrix_arithmetic_of
        if any(matrix_arithmetic_of(net.base) != _native.ARITH_SIX_TERM for net in (self.actor, self.critic)
               if hasattr(net, "base")):
            return None
        streams = self.__dict__.setdefault("_side_streams", {})
        key = masks.device.index
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=masks.device)
        return streams[key]

    def act(self, obs, rnn_states_actor, masks, available_actions=None, deterministic=False):
        actions, _, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        return actions, rnn_states_actor
