"""Separated-policy runner for StarCraft II micromanagement (``train_smac.py`` with ``--share_policy false``
or ``--algorithm_name happo``).  Interface of the reference's onpolicy/runner/separated/smac_runner.py
(SMACRunner: run :16, warmup :104, collect :116, insert :148, eval :178).

Per step each agent's policy reads device views of its own HBM buffer and its outputs go straight back into
it; only the integer actions leave the GPU.  The team-level mask logic (episode end, dead agents, time-limit
truncations) is the shared runner's.
"""
import time
from functools import reduce

import numpy as np
import torch

from onpolicy.runner.separated.base_runner import Runner, _t2n
from onpolicy.runner.shared.smac_runner import _SMAC_NAMES


class SMACRunner(Runner):
    def __init__(self, config):
        super(SMACRunner, self).__init__(config)

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads
        last_battles_game = np.zeros(self.n_rollout_threads, dtype=np.float32)
        last_battles_won = np.zeros(self.n_rollout_threads, dtype=np.float32)
        infos = []
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                for tr in self.trainer:     # the reference calls .policy on the list here (smac_runner.py:27)
                    tr.policy.lr_decay(episode, episodes)
            for step in range(self.episode_length):
                values, actions, action_log_probs, rnn_states, rnn_states_critic = self.collect(step)
                actions_env = np.stack([_t2n(a) for a in actions], axis=1)              # [N, A, act_dim]
                obs, share_obs, rewards, dones, infos, available_actions = self.envs.step(actions_env)
                self.insert((obs, share_obs, rewards, dones, infos, available_actions, values, actions,
                             action_log_probs, rnn_states, rnn_states_critic))
            self.compute()
            train_infos = self.train()

            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0:
                end = time.time()
                print("\n Map {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n"
                      .format(getattr(self.all_args, "map_name", "?"), self.algorithm_name, self.experiment_name,
                              episode, episodes, total_num_steps, self.num_env_steps,
                              int(total_num_steps / (end - start))))
                if self.env_name in _SMAC_NAMES:
                    won, game, d_won, d_game = [], [], [], []
                    for i, info in enumerate(infos):
                        if 'battles_won' in info[0].keys():
                            won.append(info[0]['battles_won'])
                            d_won.append(info[0]['battles_won'] - last_battles_won[i])
                        if 'battles_game' in info[0].keys():
                            game.append(info[0]['battles_game'])
                            d_game.append(info[0]['battles_game'] - last_battles_game[i])
                    incre_win_rate = np.sum(d_won) / np.sum(d_game) if np.sum(d_game) > 0 else 0.0
                    print("incre win rate is {}.".format(incre_win_rate))
                    self._log_scalar("incre_win_rate", incre_win_rate, total_num_steps)
                    last_battles_game, last_battles_won = game, won
                for agent_id, b in enumerate(self.buffer):
                    # sic: the reference divides by num_agents times the entries of ONE agent's masks
                    # (smac_runner.py:96), so the logged value is not a ratio of this agent's steps
                    n_entries = self.num_agents * reduce(lambda x, y: x * y, list(b.active_masks.shape))
                    train_infos[agent_id]['dead_ratio'] = 1 - float(b.active_masks.sum()) / n_entries
                self.log_train(train_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def warmup(self):
        obs, share_obs, available_actions = self.envs.reset()
        if not self.use_centralized_V:
            share_obs = obs
        f32 = torch.float32
        for agent_id, b in enumerate(self.buffer):
            b.share_obs[0] = torch.as_tensor(np.ascontiguousarray(share_obs[:, agent_id]), dtype=f32)
            b.obs[0] = torch.as_tensor(np.ascontiguousarray(obs[:, agent_id]), dtype=f32)
            b.available_actions[0] = torch.as_tensor(np.ascontiguousarray(available_actions[:, agent_id]), dtype=f32)

    @torch.no_grad()
    def collect(self, step):
        """-> per-agent lists of device tensors."""
        out = ([], [], [], [], [])
        for tr, b in zip(self.trainer, self.buffer):
            tr.prep_rollout()
            result = tr.policy.get_actions(b.share_obs[step], b.obs[step], b.rnn_states[step],
                                           b.rnn_states_critic[step], b.masks[step], b.available_actions[step])
            for lst, x in zip(out, result):
                lst.append(x)
        return out

    def insert(self, data):
        obs, share_obs, rewards, dones, infos, available_actions, \
            values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        dev = self.buffer[0].device
        dones = np.asarray(dones, dtype=bool)
        dones_env = np.all(dones, axis=1)                                   # the whole team is done
        env_alive = torch.as_tensor(~dones_env, dtype=torch.float32, device=dev)

        masks = np.ones((self.n_rollout_threads, self.num_agents, 1), dtype=np.float32)
        masks[dones_env] = 0.0
        active_masks = np.ones((self.n_rollout_threads, self.num_agents, 1), dtype=np.float32)
        active_masks[dones] = 0.0                                            # dead agents ...
        active_masks[dones_env] = 1.0                                        # ... revive with the reset
        bad_masks = np.array([[[0.0] if info[agent_id]['bad_transition'] else [1.0]
                               for agent_id in range(self.num_agents)] for info in infos], dtype=np.float32)
        if not self.use_centralized_V:
            share_obs = obs
        rewards = np.asarray(rewards, dtype=np.float32)
        for a, b in enumerate(self.buffer):
            b.insert(share_obs[:, a], obs[:, a], rnn_states[a] * env_alive.view(-1, 1, 1),
                     rnn_states_critic[a] * env_alive.view(-1, 1, 1), actions[a], action_log_probs[a], values[a],
                     rewards[:, a], masks[:, a], bad_masks[:, a], active_masks[:, a], available_actions[:, a])

    @torch.no_grad()
    def eval(self, total_num_steps):
        n = self.n_eval_rollout_threads
        eval_battles_won, eval_episode = 0, 0
        eval_episode_rewards = [[] for _ in range(n)]
        one_episode_rewards = [[] for _ in range(n)]
        eval_obs, eval_share_obs, eval_available_actions = self.eval_envs.reset()
        eval_rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
        eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        while True:
            collected = []
            for a, tr in enumerate(self.trainer):
                tr.prep_rollout()
                act, state = tr.policy.act(eval_obs[:, a], eval_rnn_states[:, a], eval_masks[:, a],
                                           eval_available_actions[:, a], deterministic=True)
                eval_rnn_states[:, a] = _t2n(state)
                collected.append(_t2n(act))
            eval_actions = np.array(collected).transpose(1, 0, 2)
            eval_obs, eval_share_obs, eval_rewards, eval_dones, eval_infos, eval_available_actions = \
                self.eval_envs.step(eval_actions)
            for i in range(n):
                one_episode_rewards[i].append(eval_rewards[i])
            eval_dones_env = np.all(eval_dones, axis=1)
            eval_rnn_states[eval_dones_env] = 0.0
            eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            eval_masks[eval_dones_env] = 0.0
            for i in range(n):
                if eval_dones_env[i]:
                    eval_episode += 1
                    eval_episode_rewards[i].append(np.sum(one_episode_rewards[i], axis=0))
                    one_episode_rewards[i] = []
                    if eval_infos[i][0]['won']:
                        eval_battles_won += 1
            if eval_episode >= self.all_args.eval_episodes:
                rewards = np.concatenate([r for r in eval_episode_rewards if len(r) > 0])
                self.log_env({'eval_average_episode_rewards': rewards}, total_num_steps)
                eval_win_rate = eval_battles_won / eval_episode
                print("eval win rate is {}.".format(eval_win_rate))
                self._log_scalar("eval_win_rate", eval_win_rate, total_num_steps)
                break
