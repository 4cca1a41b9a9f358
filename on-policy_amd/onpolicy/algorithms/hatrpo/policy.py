"""HATRPO policy: the MAPPO actor / critic pair whose ``evaluate_actions`` also reports the action distribution's
parameters (reference onpolicy/algorithms/hatrpo/policy.py:5-135; the actor switches on ``args.algorithm_name ==
"hatrpo"``).  One such policy exists per agent."""
from onpolicy.algorithms.happo.policy import HAPPO_Policy


class HATRPO_Policy(HAPPO_Policy):
    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks,
                         available_actions=None, active_masks=None):
        """-> (values, action_log_probs, dist_entropy, action_mean, action_std, normalised logits or None)."""
        actor_out = self.actor.evaluate_actions(obs, rnn_states_actor, action, masks, available_actions, active_masks)
        assert len(actor_out) == 5, "build the policy with args.algorithm_name == 'hatrpo'"
        values = self.critic(cent_obs, rnn_states_critic, masks)[0]
        return (values,) + tuple(actor_out)
