"""Runner for Google Research Football (``train_football.py``).  Interface of the reference's
onpolicy/runner/shared/football_runner.py (FootballRunner: run :20, warmup :76, collect :84, insert :108,
log_env :142, eval :151, render :229).

Football has no separate state: the critic sees the agents' observations (``share_obs = obs``), actions go to
the envs as one integer per agent, and ``infos`` is one dict per env (score, steps left).  As in the other
shared runners the policy reads device views of the HBM buffer and its outputs go straight back into it;
only the integer actions and the new observations / rewards / dones cross the PCIe bus.  Save / log / eval
intervals count env steps here, not episodes (reference :48-69).
"""
import time
from collections import defaultdict

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner, _t2n


class FootballRunner(Runner):
    def __init__(self, config):
        super(FootballRunner, self).__init__(config)
        self.env_infos = defaultdict(list)

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads_job
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                self.trainer.policy.lr_decay(episode, episodes)
            for step in range(self.episode_length):
                values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = self.collect(step)
                obs, rewards, dones, infos = self.envs.step(actions_env)
                self.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states,
                             rnn_states_critic))
            self.compute()
            train_infos = self.train()

            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads_job
            if total_num_steps % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if total_num_steps % self.log_interval == 0:
                end = time.time()
                print("\n Env {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n"
                      .format(self.env_name, self.algorithm_name, self.experiment_name, episode, episodes,
                              total_num_steps, self.num_env_steps, int(total_num_steps / (end - start))))
                train_infos["average_episode_rewards"] = float(self.buffer.rewards.mean()) * self.episode_length
                print("average episode rewards is {}".format(train_infos["average_episode_rewards"]))
                self.log_train(train_infos, total_num_steps)
                self.log_env(self.env_infos, total_num_steps)
                self.env_infos = defaultdict(list)
            if total_num_steps % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def warmup(self):
        obs = torch.as_tensor(np.asarray(self.envs.reset()), dtype=torch.float32)
        self.buffer.share_obs[0] = obs
        self.buffer.obs[0] = obs

    @torch.no_grad()
    def collect(self, step):
        self.trainer.prep_rollout()
        b = self.buffer
        value, action, action_log_prob, rnn_states, rnn_states_critic = self.trainer.policy.get_actions(
            self._rows(b.share_obs[step]), self._rows(b.obs[step]), self._rows(b.rnn_states[step]),
            self._rows(b.rnn_states_critic[step]), self._rows(b.masks[step]))
        actions = self._per_env(action)
        host = _t2n(actions)                                              # the one D2H copy
        actions_env = [host[idx, :, 0] for idx in range(self.n_rollout_threads)]
        return (self._per_env(value), actions, self._per_env(action_log_prob), self._per_env(rnn_states),
                self._per_env(rnn_states_critic), actions_env)

    def _record_finished(self, dones_env, infos, sink):
        for done, info in zip(dones_env, infos):
            if done:
                sink["goal"].append(info["score_reward"])
                sink["win_rate"].append(1 if info["score_reward"] > 0 else 0)
                sink["steps"].append(info["max_steps"] - info["steps_left"])

    def insert(self, data):
        obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        dones_env = np.all(np.asarray(dones, dtype=bool), axis=-1)
        if np.any(dones_env):
            self._record_finished(dones_env, infos, self.env_infos)
        alive = torch.as_tensor(~dones_env, dtype=torch.float32, device=self.buffer.device)      # [N]
        keep = alive.view(-1, 1, 1, 1)
        masks = alive.view(-1, 1, 1).expand(-1, self.num_agents, 1)      # whole team restarts together (:125-129)
        self.buffer.insert(share_obs=obs, obs=obs, rnn_states_actor=rnn_states * keep,
                           rnn_states_critic=rnn_states_critic * keep, actions=actions,
                           action_log_probs=action_log_probs, value_preds=values, rewards=rewards, masks=masks)

    @torch.no_grad()
    def _act(self, obs, rnn_states, masks, n, deterministic):
        action, rnn_states = self.trainer.policy.act(np.concatenate(obs), np.concatenate(rnn_states),
                                                     np.concatenate(masks), deterministic=deterministic)
        actions = np.array(np.split(_t2n(action), n))
        return [actions[idx, :, 0] for idx in range(n)], np.array(np.split(_t2n(rnn_states), n))

    @torch.no_grad()
    def eval(self, total_num_steps):
        n, want = self.n_eval_rollout_threads, self.all_args.eval_episodes
        eval_obs = self.eval_envs.reset()
        eval_rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
        eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        results = defaultdict(list)
        # every thread plays its share of the episodes, the first `want % n` threads one more
        quota = np.full(n, want // n, dtype=int)
        quota[:want % n] += 1
        played = np.zeros(n, dtype=int)
        step = 0
        while len(results["goal"]) < want and step < self.episode_length:
            self.trainer.prep_rollout()
            eval_actions_env, eval_rnn_states = self._act(eval_obs, eval_rnn_states, eval_masks, n,
                                                          getattr(self.all_args, "eval_deterministic", True))   # a train_football.py flag
            eval_obs, _, eval_dones, eval_infos = self.eval_envs.step(eval_actions_env)
            eval_dones_env = np.all(np.asarray(eval_dones, dtype=bool), axis=-1)
            counted = eval_dones_env & (played < quota)
            self._record_finished(counted, eval_infos, results)
            played += counted
            eval_rnn_states[eval_dones_env] = 0.0
            eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            eval_masks[eval_dones_env] = 0.0
            step += 1
        # the reference averages over eval_episodes slots, unfinished ones counting as 0 (:216-218)
        mean = lambda xs: float(np.sum(xs)) / want
        eval_goal, eval_win_rate, eval_step = mean(results["goal"]), mean(results["win_rate"]), mean(results["steps"])
        print("eval expected goal is {}.".format(eval_goal))
        if self.use_wandb:
            self._log_scalar("eval_goal", eval_goal, total_num_steps)
            self._log_scalar("eval_win_rate", eval_win_rate, total_num_steps)
            self._log_scalar("eval_step", eval_step, total_num_steps)
        else:
            self.writter.add_scalars("eval_goal", {"expected_goal": eval_goal}, total_num_steps)
            self.writter.add_scalars("eval_win_rate", {"eval_win_rate": eval_win_rate}, total_num_steps)
            self.writter.add_scalars("eval_step", {"expected_step": eval_step}, total_num_steps)

    @torch.no_grad()
    def render(self):
        envs, n = self.envs, self.n_rollout_threads
        render_goals = np.zeros(self.all_args.render_episodes)
        for i_episode in range(self.all_args.render_episodes):
            obs = envs.reset()
            rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
            masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            frames = []
            if self.all_args.save_gifs:
                frames.append(envs.envs[0].env.unwrapped.observation()[0]["frame"])
            dones, rewards = False, None
            while not np.any(dones):
                self.trainer.prep_rollout()
                actions_env, rnn_states = self._act(obs, rnn_states, masks, n, True)
                obs, rewards, dones, infos = envs.step(actions_env)
                if self.all_args.save_gifs:
                    frames.append(infos[0]["frame"])
            render_goals[i_episode] = rewards[0, 0]
            print("goal in episode {}: {}".format(i_episode, rewards[0, 0]))
            if self.all_args.save_gifs:
                import imageio
                imageio.mimsave(uri="{}/episode{}.gif".format(str(self.gif_dir), i_episode), ims=frames,
                                format="GIF", duration=self.all_args.ifi)
        print("expected goal: {}".format(np.mean(render_goals)))
