"""Rollout / update loop of the ``share_policy=False`` runners: one policy, trainer and HBM buffer per
agent.  Same class, config contract and method names as the reference's
onpolicy/runner/separated/base_runner.py (Runner :15, compute :125, train :135, save :185, restore :196,
log_train :205, log_env :214).

What changes against the reference:
  * every agent's ``SeparatedReplayBuffer`` lives in HBM; ``compute`` and the two whole-buffer
    ``evaluate_actions`` passes of ``train`` read device views of it (the reference reshapes numpy
    copies and moves them to the device each time, base_runner.py:146-175);
  * the HAPPO ``factor`` (running product over the agents already updated of
    ``prod_k exp(new_logp_k - old_logp_k)``, base_runner.py:177) stays a device tensor;
  * the two whole-buffer passes run without autograd and in thread spans, so their activations are
    bounded however long the rollout is.
"""
import os

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner as _SharedRunner, _t2n, wandb  # noqa: F401
from onpolicy.utils.separated_buffer import SeparatedReplayBuffer

# rows x widest activation kept below this many elements per evaluate_actions span
_SPAN_ELEMENTS = 1 << 28


class Runner(_SharedRunner):
    def _init_learner(self):
        a = self.all_args
        if self.algorithm_name in ("mat", "mat_dec"):
            raise NotImplementedError("algorithm %r needs the shared-policy runner (one transformer for the team)"
                                      % self.algorithm_name)
        if self.algorithm_name == "hatrpo":
            from onpolicy.algorithms.hatrpo.hatrpo_trainer import HATRPO as TrainAlgo
            from onpolicy.algorithms.hatrpo.policy import HATRPO_Policy as Policy
        elif self.algorithm_name == "happo":
            from onpolicy.algorithms.happo.happo_trainer import HAPPO as TrainAlgo
            from onpolicy.algorithms.happo.policy import HAPPO_Policy as Policy
        else:
            from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO as TrainAlgo
            from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy as Policy
        print("share_observation_space: ", self.envs.share_observation_space)
        print("observation_space: ", self.envs.observation_space)
        print("action_space: ", self.envs.action_space)

        spaces = []
        for agent_id in range(self.num_agents):
            share = self.envs.share_observation_space[agent_id] if self.use_centralized_V \
                else self.envs.observation_space[agent_id]
            spaces.append((self.envs.observation_space[agent_id], share, self.envs.action_space[agent_id]))
        self.policy = [Policy(a, obs, share, act, device=self.device) for obs, share, act in spaces]
        if self.model_dir is not None:
            self.restore()
        buffer_device = self.device if torch.device(self.device).type == "cuda" else None
        self.trainer = [TrainAlgo(a, po, device=self.device) for po in self.policy]
        if self.model_dir is not None:
            self._restore_normalizers()
        self.buffer = [SeparatedReplayBuffer(a, obs, share, act, device=buffer_device) for obs, share, act in spaces]

    @torch.no_grad()
    def compute(self):
        for tr, b in zip(self.trainer, self.buffer):
            tr.prep_rollout()
            next_value = tr.policy.get_values(b.share_obs[-1], b.rnn_states_critic[-1], b.masks[-1])
            b.compute_returns(next_value, tr.value_normalizer)

    @torch.no_grad()
    def _buffer_log_probs(self, agent_id):
        """log pi(a_t | o_t) of every stored action under the agent's current actor, [T, N, act_dim]."""
        b, actor = self.buffer[agent_id], self.trainer[agent_id].policy.actor
        T, N = self.episode_length, self.n_rollout_threads
        widest = max(self.hidden_size, b.obs.shape[-1] if b.obs.dim() == 3 else int(np.prod(b.obs.shape[2:])))
        span = max(1, min(N, _SPAN_ELEMENTS // max(1, T * widest)))
        flat = lambda x: x.reshape(-1, *x.shape[2:])
        out = []
        for lo in range(0, N, span):
            hi = min(N, lo + span)
            avail = None if b.available_actions is None else flat(b.available_actions[:-1, lo:hi])
            logp = actor.evaluate_actions(flat(b.obs[:-1, lo:hi]), flat(b.rnn_states[0:1, lo:hi]),
                                          flat(b.actions[:, lo:hi]), flat(b.masks[:-1, lo:hi]), avail,
                                          flat(b.active_masks[:-1, lo:hi]))[0]
            out.append(logp.reshape(T, hi - lo, -1))
        return out[0] if len(out) == 1 else torch.cat(out, dim=1)

    def train(self):
        """Sequential update in random agent order with the HAPPO factor bookkeeping
        (reference base_runner.py:135-183)."""
        # like the reference, the infos are returned in UPDATE order (base_runner.py:178 appends), so
        # log_train's "agent%i/" prefix counts positions in this round's random order, not agent ids
        train_infos = []
        factor = torch.ones(self.episode_length, self.n_rollout_threads, 1, dtype=torch.float32,
                            device=self.buffer[0].device)
        for agent_id in torch.randperm(self.num_agents).tolist():
            tr, b = self.trainer[agent_id], self.buffer[agent_id]
            tr.prep_training()
            b.update_factor(factor)
            old_logp = self._buffer_log_probs(agent_id)
            train_infos.append(tr.train(b))
            new_logp = self._buffer_log_probs(agent_id)
            factor = factor * torch.prod(torch.exp(new_logp - old_logp), dim=-1, keepdim=True)
            b.after_update()
        return train_infos

    def save(self):
        for agent_id, tr in enumerate(self.trainer):
            torch.save(tr.policy.actor.state_dict(), str(self.save_dir) + "/actor_agent" + str(agent_id) + ".pt")
            torch.save(tr.policy.critic.state_dict(), str(self.save_dir) + "/critic_agent" + str(agent_id) + ".pt")
            if tr._use_valuenorm:
                # sic: the reference's file name (base_runner.py:193)
                torch.save(tr.value_normalizer.state_dict(),
                           str(self.save_dir) + "/vnrom_agent" + str(agent_id) + ".pt")

    def restore(self):
        for agent_id, po in enumerate(self.policy):
            load = lambda name: torch.load(str(self.model_dir) + '/' + name + str(agent_id) + '.pt',
                                           map_location=self.device)
            po.actor.load_state_dict(load('actor_agent'))
            po.critic.load_state_dict(load('critic_agent'))
        # the reference reloads the value normalisers in the same loop, but its restore() runs before
        # the trainers exist (base_runner.py:96-97 vs :99-111) and fails there; here the normalisers
        # are reloaded once the trainers have been built (_restore_normalizers)

    def _restore_normalizers(self):
        for agent_id, tr in enumerate(self.trainer):
            path = str(self.model_dir) + '/vnrom_agent' + str(agent_id) + '.pt'
            if tr._use_valuenorm and os.path.exists(path):
                tr.value_normalizer.load_state_dict(torch.load(path, map_location=self.device))

    def log_train(self, train_infos, total_num_steps):
        for agent_id in range(self.num_agents):
            for k, v in train_infos[agent_id].items():
                self._log_scalar("agent%i/" % agent_id + k, float(v), total_num_steps)
