"""Runner for StarCraft II micromanagement (``train_smac.py`` / ``train_smacv2.py``).  Interface of the
reference's onpolicy/runner/shared/smac_runner.py (SMACRunner: run :16, warmup :97, collect :111,
insert :129, log_train :153, eval :162).  SMAC adds three per-step fields to the buffer --
``available_actions``, ``active_masks`` (dead agents) and ``bad_masks`` (episode-limit truncations) --
which is exactly the mask logic the GAE kernel's proper-time-limits branch and the masked losses
consume.  Network outputs stay on the device between ``collect`` and ``insert``.
"""
import time

import numpy as np
import torch

from onpolicy.runner.shared import _smac_common as common
from onpolicy.runner.shared.base_runner import Runner, _t2n

_SMAC_NAMES = common.SMAC_ENV_NAMES


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
                self.trainer.policy.lr_decay(episode, episodes)
            for step in range(T):
                outputs = self.collect(step)
                env_out = self.envs.step(_t2n(outputs[1]))           # actions: the one D2H copy of the step
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
                if self.env_name in _SMAC_NAMES:
                    incre_win_rate = battles.incremental_win_rate(infos)
                    print("incre win rate is {}.".format(incre_win_rate))
                    self._log_scalar("incre_win_rate", incre_win_rate, steps_done)
                alive = float(self.buffer.active_masks.sum()) / common.mask_entries(self.buffer.active_masks)
                train_infos['dead_ratio'] = 1 - alive
                self.log_train(train_infos, steps_done)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(steps_done)

    def warmup(self):
        obs, share_obs, available_actions = self.envs.reset()
        first = dict(obs=obs, share_obs=share_obs if self.use_centralized_V else obs,
                     available_actions=available_actions)
        for name, value in first.items():
            getattr(self.buffer, name)[0] = torch.as_tensor(value, dtype=torch.float32)

    @torch.no_grad()
    def collect(self, step):
        """-> (values, actions, action_log_probs, rnn_states, rnn_states_critic), each [N, A, ...] on the device."""
        self.trainer.prep_rollout()
        b = self.buffer
        fields = (b.share_obs, b.obs, b.rnn_states, b.rnn_states_critic, b.masks, b.available_actions)
        outputs = self.trainer.policy.get_actions(*[self._rows(f[step]) for f in fields])
        return tuple(self._per_env(x) for x in outputs)

    def insert(self, data):
        obs, share_obs, rewards, dones, infos, available_actions, \
            values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        team_done, masks, active_masks, bad_masks = common.team_masks(dones, infos, self.num_agents)
        # a finished team restarts from zero recurrent state
        keep = torch.as_tensor(~team_done, dtype=torch.float32, device=self.buffer.device).view(-1, 1, 1, 1)
        self.buffer.insert(share_obs if self.use_centralized_V else obs, obs, rnn_states * keep,
                           rnn_states_critic * keep, actions, action_log_probs, values, rewards, masks, bad_masks,
                           active_masks, available_actions)

    def log_train(self, train_infos, total_num_steps):
        train_infos["average_step_rewards"] = float(self.buffer.rewards.mean())
        for k, v in train_infos.items():
            self._log_scalar(k, float(v), total_num_steps)

    @torch.no_grad()
    def eval(self, total_num_steps):
        """Deterministic policy on the eval envs until ``eval_episodes`` episodes have finished; logs the episode
        returns and the win rate (reference smac_runner.py:162-221)."""
        n, A = self.n_eval_rollout_threads, self.num_agents
        finished, won = 0, 0
        episode_returns, running = [], []                 # ``running`` is shared by all threads, as in the reference
        obs, share_obs, available_actions = self.eval_envs.reset()
        rnn_states = np.zeros((n, A, self.recurrent_N, self.hidden_size), dtype=np.float32)
        masks = np.ones((n, A, 1), dtype=np.float32)
        while finished < self.all_args.eval_episodes:
            self.trainer.prep_rollout()
            inputs = [np.concatenate(x) for x in (obs, rnn_states, masks, available_actions)]
            if self._mat:                                  # the transformer also takes the centralised observation
                inputs.insert(0, np.concatenate(share_obs))
            actions, states = self.trainer.policy.act(*inputs, deterministic=True)
            actions = np.array(np.split(_t2n(actions), n))
            rnn_states = np.array(np.split(_t2n(states), n))
            obs, share_obs, rewards, dones, infos, available_actions = self.eval_envs.step(actions)
            running.append(rewards)
            team_done = np.all(dones, axis=1)
            rnn_states[team_done] = 0.0
            masks = np.ones((n, A, 1), dtype=np.float32)
            masks[team_done] = 0.0
            for i in np.flatnonzero(team_done):
                finished += 1
                episode_returns.append(np.sum(running, axis=0))
                running = []
                won += bool(infos[i][0]['won'])
        self.log_env({'eval_average_episode_rewards': np.array(episode_returns)}, total_num_steps)
        eval_win_rate = won / finished
        print("eval win rate is {}.".format(eval_win_rate))
        self._log_scalar("eval_win_rate", eval_win_rate, total_num_steps)
