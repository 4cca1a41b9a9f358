"""Runner for StarCraft II micromanagement (``train_smac.py`` / ``train_smacv2.py``).  Interface of the
reference's onpolicy/runner/shared/smac_runner.py (SMACRunner: run :16, warmup :97, collect :111,
insert :129, log_train :153, eval :162).  SMAC adds three per-step fields to the buffer --
``available_actions``, ``active_masks`` (dead agents) and ``bad_masks`` (episode-limit truncations) --
which is exactly the mask logic the GAE kernel's proper-time-limits branch and the masked losses
consume.  Network outputs stay on the device between ``collect`` and ``insert``.
"""
import time
from functools import reduce

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner, _t2n

_SMAC_NAMES = ("StarCraft2", "SMACv2", "SMAC", "StarCraft2v2")


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
                self.trainer.policy.lr_decay(episode, episodes)
            for step in range(self.episode_length):
                values, actions, action_log_probs, rnn_states, rnn_states_critic = self.collect(step)
                obs, share_obs, rewards, dones, infos, available_actions = self.envs.step(_t2n(actions))
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
                n_entries = reduce(lambda x, y: x * y, list(self.buffer.active_masks.shape))
                train_infos['dead_ratio'] = 1 - float(self.buffer.active_masks.sum()) / n_entries
                self.log_train(train_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def warmup(self):
        obs, share_obs, available_actions = self.envs.reset()
        if not self.use_centralized_V:
            share_obs = obs
        f32 = torch.float32
        self.buffer.share_obs[0] = torch.as_tensor(share_obs, dtype=f32)
        self.buffer.obs[0] = torch.as_tensor(obs, dtype=f32)
        self.buffer.available_actions[0] = torch.as_tensor(available_actions, dtype=f32)

    @torch.no_grad()
    def collect(self, step):
        self.trainer.prep_rollout()
        b = self.buffer
        value, action, action_log_prob, rnn_state, rnn_state_critic = self.trainer.policy.get_actions(
            self._rows(b.share_obs[step]), self._rows(b.obs[step]), self._rows(b.rnn_states[step]),
            self._rows(b.rnn_states_critic[step]), self._rows(b.masks[step]),
            self._rows(b.available_actions[step]))
        return (self._per_env(value), self._per_env(action), self._per_env(action_log_prob),
                self._per_env(rnn_state), self._per_env(rnn_state_critic))

    def insert(self, data):
        obs, share_obs, rewards, dones, infos, available_actions, \
            values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        dev = self.buffer.device
        dones = np.asarray(dones, dtype=bool)
        dones_env = np.all(dones, axis=1)                                   # the whole team is done
        env_alive = torch.as_tensor(~dones_env, dtype=torch.float32, device=dev).view(-1, 1, 1, 1)
        rnn_states = rnn_states * env_alive
        rnn_states_critic = rnn_states_critic * env_alive

        masks = np.ones((self.n_rollout_threads, self.num_agents, 1), dtype=np.float32)
        masks[dones_env] = 0.0
        active_masks = np.ones((self.n_rollout_threads, self.num_agents, 1), dtype=np.float32)
        active_masks[dones] = 0.0                                            # dead agents ...
        active_masks[dones_env] = 1.0                                        # ... revive with the reset
        bad_masks = np.array([[[0.0] if info[agent_id]['bad_transition'] else [1.0]
                               for agent_id in range(self.num_agents)] for info in infos], dtype=np.float32)
        if not self.use_centralized_V:
            share_obs = obs
        self.buffer.insert(share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, values,
                           rewards, masks, bad_masks, active_masks, available_actions)

    def log_train(self, train_infos, total_num_steps):
        train_infos["average_step_rewards"] = float(self.buffer.rewards.mean())
        for k, v in train_infos.items():
            self._log_scalar(k, float(v), total_num_steps)

    @torch.no_grad()
    def eval(self, total_num_steps):
        n = self.n_eval_rollout_threads
        eval_battles_won, eval_episode = 0, 0
        eval_episode_rewards, one_episode_rewards = [], []
        eval_obs, eval_share_obs, eval_available_actions = self.eval_envs.reset()
        eval_rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
        eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        while True:
            self.trainer.prep_rollout()
            critic_input = [np.concatenate(eval_share_obs)] if self._mat else []     # smac_runner.py:176-182
            eval_actions, eval_rnn_states = self.trainer.policy.act(
                *critic_input, np.concatenate(eval_obs), np.concatenate(eval_rnn_states), np.concatenate(eval_masks),
                np.concatenate(eval_available_actions), deterministic=True)
            eval_actions = np.array(np.split(_t2n(eval_actions), n))
            eval_rnn_states = np.array(np.split(_t2n(eval_rnn_states), n))
            eval_obs, eval_share_obs, eval_rewards, eval_dones, eval_infos, eval_available_actions = \
                self.eval_envs.step(eval_actions)
            one_episode_rewards.append(eval_rewards)
            eval_dones_env = np.all(eval_dones, axis=1)
            eval_rnn_states[eval_dones_env] = 0.0
            eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            eval_masks[eval_dones_env] = 0.0
            for i in range(n):
                if eval_dones_env[i]:
                    eval_episode += 1
                    eval_episode_rewards.append(np.sum(one_episode_rewards, axis=0))
                    one_episode_rewards = []
                    if eval_infos[i][0]['won']:
                        eval_battles_won += 1
            if eval_episode >= self.all_args.eval_episodes:
                self.log_env({'eval_average_episode_rewards': np.array(eval_episode_rewards)}, total_num_steps)
                eval_win_rate = eval_battles_won / eval_episode
                print("eval win rate is {}.".format(eval_win_rate))
                self._log_scalar("eval_win_rate", eval_win_rate, total_num_steps)
                break
