"""Separated-policy runner for StarCraft II micromanagement (``train_smac.py`` with ``--share_policy false``
or ``--algorithm_name happo``).  Interface of the reference's onpolicy/runner/separated/smac_runner.py
(SMACRunner: run :16, warmup :104, collect :116, insert :148, eval :178).

Per step each agent's policy reads device views of its own HBM buffer and its outputs go straight back into
it; only the integer actions leave the GPU.  The team-level mask logic (episode end, dead agents, time-limit
truncations) is the shared runner's.
"""
import time

import numpy as np
import torch

from onpolicy.runner.separated.base_runner import Runner, _t2n
from onpolicy.runner.shared import _smac_common as common


class SMACRunner(Runner):
    def __init__(self, config):
        super(SMACRunner, self).__init__(config)

    # ------------------------------------------------------------------ training loop
    def run(self):
        self.warmup()
        started = time.time()
        T, N = self.episode_length, self.n_rollout_threads
        NJ = self.n_rollout_threads_job     # job-wide thread count (== N in a single-process run)
        episodes = int(self.num_env_steps) // T // NJ
        battles = common.BattleLog(N)
        infos = []
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                for tr in self.trainer:     # the reference calls .policy on the list here (smac_runner.py:27)
                    tr.policy.lr_decay(episode, episodes)
            for step in range(T):
                outputs = self.collect(step)
                team_actions = np.stack([_t2n(a) for a in outputs[1]], axis=1)            # [N, A, act_dim]
                env_out = self.envs.step(team_actions)
                infos = env_out[4]
                self.insert(tuple(env_out) + tuple(outputs))
            self.compute()
            train_infos = self.train()

            steps_done = (episode + 1) * T * NJ
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0:
                print(common.progress_line(self.all_args, self.algorithm_name, self.experiment_name, episode, episodes,
                                           steps_done, self.num_env_steps, time.time() - started))
                if self.env_name in common.SMAC_ENV_NAMES:
                    incre_win_rate = battles.incremental_win_rate(infos)
                    print("incre win rate is {}.".format(incre_win_rate))
                    self._log_scalar("incre_win_rate", incre_win_rate, steps_done)
                for agent_id, b in enumerate(self.buffer):
                    # sic: the reference divides by num_agents times the entries of ONE agent's masks
                    # (smac_runner.py:96), so the logged value is not a ratio of this agent's steps
                    entries = self.num_agents * common.mask_entries(b.active_masks)
                    train_infos[agent_id]['dead_ratio'] = 1 - float(b.active_masks.sum()) / entries
                self.log_train(train_infos, steps_done)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(steps_done)

    def warmup(self):
        obs, share_obs, available_actions = self.envs.reset()
        first = dict(obs=obs, share_obs=share_obs if self.use_centralized_V else obs,
                     available_actions=available_actions)
        for agent_id, b in enumerate(self.buffer):
            for name, value in first.items():
                getattr(b, name)[0] = torch.as_tensor(np.ascontiguousarray(value[:, agent_id]), dtype=torch.float32)

    @torch.no_grad()
    def collect(self, step):
        """-> (values, actions, action_log_probs, rnn_states, rnn_states_critic), each a per-agent list of device
        tensors [N, ...]."""
        per_output = ([], [], [], [], [])
        for tr, b in zip(self.trainer, self.buffer):
            tr.prep_rollout()
            outputs = tr.policy.get_actions(b.share_obs[step], b.obs[step], b.rnn_states[step],
                                            b.rnn_states_critic[step], b.masks[step], b.available_actions[step])
            for bucket, x in zip(per_output, outputs):
                bucket.append(x)
        return per_output

    def insert(self, data):
        obs, share_obs, rewards, dones, infos, available_actions, \
            values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        team_done, masks, active_masks, bad_masks = common.team_masks(dones, infos, self.num_agents)
        keep = torch.as_tensor(~team_done, dtype=torch.float32, device=self.buffer[0].device).view(-1, 1, 1)
        critic_obs = share_obs if self.use_centralized_V else obs
        rewards = np.asarray(rewards, dtype=np.float32)
        for a, b in enumerate(self.buffer):
            b.insert(critic_obs[:, a], obs[:, a], rnn_states[a] * keep, rnn_states_critic[a] * keep, actions[a],
                     action_log_probs[a], values[a], rewards[:, a], masks[:, a], bad_masks[:, a], active_masks[:, a],
                     available_actions[:, a])

    @torch.no_grad()
    def eval(self, total_num_steps):
        """Deterministic policies on the eval envs until ``eval_episodes`` episodes have finished (reference
        smac_runner.py:178-253: returns are collected per thread and concatenated)."""
        n, A = self.n_eval_rollout_threads, self.num_agents
        finished, won = 0, 0
        episode_returns = [[] for _ in range(n)]
        running = [[] for _ in range(n)]
        obs, _, available_actions = self.eval_envs.reset()
        rnn_states = np.zeros((n, A, self.recurrent_N, self.hidden_size), dtype=np.float32)
        masks = np.ones((n, A, 1), dtype=np.float32)
        while finished < self.all_args.eval_episodes:
            team_actions = []
            for a, tr in enumerate(self.trainer):
                tr.prep_rollout()
                action, state = tr.policy.act(obs[:, a], rnn_states[:, a], masks[:, a], available_actions[:, a],
                                              deterministic=True)
                rnn_states[:, a] = _t2n(state)
                team_actions.append(_t2n(action))
            obs, _, rewards, dones, infos, available_actions = self.eval_envs.step(
                np.array(team_actions).transpose(1, 0, 2))
            for i in range(n):
                running[i].append(rewards[i])
            team_done = np.all(dones, axis=1)
            rnn_states[team_done] = 0.0
            masks = np.ones((n, A, 1), dtype=np.float32)
            masks[team_done] = 0.0
            for i in np.flatnonzero(team_done):
                finished += 1
                episode_returns[i].append(np.sum(running[i], axis=0))
                running[i] = []
                won += bool(infos[i][0]['won'])
        returns = np.concatenate([r for r in episode_returns if len(r) > 0])
        self.log_env({'eval_average_episode_rewards': returns}, total_num_steps)
        eval_win_rate = won / finished
        print("eval win rate is {}.".format(eval_win_rate))
        self._log_scalar("eval_win_rate", eval_win_rate, total_num_steps)
