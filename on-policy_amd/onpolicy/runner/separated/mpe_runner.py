"""Separated-policy runner for the multi-agent particle envs (``train_mpe.py --share_policy false``).
Interface of the reference's onpolicy/runner/separated/mpe_runner.py (MPERunner: run :18, warmup :79,
collect :94, insert :148, eval :179, render :237).

As in the shared runner only the integer actions (device -> host) and the new observations / rewards /
dones (host -> device) cross the PCIe bus per step; each agent's values, log-probs and RNN states go
from its policy straight into its HBM buffer.
"""
import time

import numpy as np
import torch

from onpolicy.runner.separated.base_runner import Runner, _t2n


def _one_hot(space, action):
    """Integer actions [N, k] of one agent -> the one-hot layout the MPE envs expect."""
    kind = space.__class__.__name__
    if kind == 'MultiDiscrete':
        return np.concatenate([np.eye(space.high[i] + 1)[action[:, i]] for i in range(space.shape)], axis=1)
    if kind == 'Discrete':
        return np.squeeze(np.eye(space.n)[action], 1)
    raise NotImplementedError(kind)


def _agent_obs(obs, agent_id):
    """Observation of one agent over all envs.  ``obs`` is [N, A, D] or an object array of per-agent
    vectors of different lengths (heterogeneous scenarios)."""
    return np.array(list(obs[:, agent_id]), dtype=np.float32)


def _joint_obs(obs):
    """Centralised critic input: all agents' observations of an env concatenated, [N, sum D]."""
    if isinstance(obs, np.ndarray) and obs.dtype != object:
        return np.asarray(obs, dtype=np.float32).reshape(obs.shape[0], -1)
    return np.array([np.concatenate([np.asarray(x, dtype=np.float32).ravel() for x in o]) for o in obs])


class MPERunner(Runner):
    def __init__(self, config):
        super(MPERunner, self).__init__(config)

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads_job
        infos = []
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                for tr in self.trainer:
                    tr.policy.lr_decay(episode, episodes)

            for step in range(self.episode_length):
                values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = self.collect(step)
                obs, rewards, dones, infos = self.envs.step(actions_env)
                self.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states,
                             rnn_states_critic))

            self.compute()
            train_infos = self.train()

            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads_job
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0:
                end = time.time()
                print("\n Scenario {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n"
                      .format(getattr(self.all_args, "scenario_name", "?"), self.algorithm_name,
                              self.experiment_name, episode, episodes, total_num_steps, self.num_env_steps,
                              int(total_num_steps / (end - start))))
                if self.env_name == "MPE":
                    for agent_id in range(self.num_agents):
                        idv_rews = [info[agent_id]['individual_reward'] for info in infos
                                    if 'individual_reward' in info[agent_id].keys()]
                        if idv_rews:
                            train_infos[agent_id]['individual_rewards'] = float(np.mean(idv_rews))
                        train_infos[agent_id]["average_episode_rewards"] = \
                            float(self.buffer[agent_id].rewards.mean()) * self.episode_length
                self.log_train(train_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def _share_obs(self, obs, agent_id, joint):
        return joint if self.use_centralized_V else _agent_obs(obs, agent_id)

    def warmup(self):
        obs = self.envs.reset()
        joint = _joint_obs(obs) if self.use_centralized_V else None
        for agent_id, b in enumerate(self.buffer):
            b.share_obs[0] = torch.as_tensor(self._share_obs(obs, agent_id, joint), dtype=torch.float32)
            b.obs[0] = torch.as_tensor(_agent_obs(obs, agent_id), dtype=torch.float32)

    @torch.no_grad()
    def collect(self, step):
        """-> per-agent lists of device tensors (values, actions, log-probs, RNN states) and the
        [envs][agents] one-hot actions for the envs."""
        values, actions, action_log_probs, rnn_states, rnn_states_critic, one_hot = [], [], [], [], [], []
        for agent_id, (tr, b) in enumerate(zip(self.trainer, self.buffer)):
            tr.prep_rollout()
            value, action, action_log_prob, rnn_state, rnn_state_critic = tr.policy.get_actions(
                b.share_obs[step], b.obs[step], b.rnn_states[step], b.rnn_states_critic[step], b.masks[step])
            values.append(value)
            actions.append(action)
            action_log_probs.append(action_log_prob)
            rnn_states.append(rnn_state)
            rnn_states_critic.append(rnn_state_critic)
            one_hot.append(_one_hot(self.envs.action_space[agent_id], _t2n(action).astype(np.int64)))
        actions_env = [[one_hot[a][i] for a in range(self.num_agents)] for i in range(self.n_rollout_threads)]
        return values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env

    def insert(self, data):
        obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        dev = self.buffer[0].device
        alive = torch.as_tensor(~np.asarray(dones, dtype=bool), dtype=torch.float32, device=dev)   # [N, A]
        rewards = np.asarray(rewards, dtype=np.float32)
        joint = _joint_obs(obs) if self.use_centralized_V else None
        for agent_id, b in enumerate(self.buffer):
            keep = alive[:, agent_id]
            # finished agents restart from a zero RNN state and get mask 0 (reference mpe_runner.py:151-154)
            b.insert(self._share_obs(obs, agent_id, joint), _agent_obs(obs, agent_id),
                     rnn_states[agent_id] * keep.view(-1, 1, 1), rnn_states_critic[agent_id] * keep.view(-1, 1, 1),
                     actions[agent_id], action_log_probs[agent_id], values[agent_id], rewards[:, agent_id],
                     keep.view(-1, 1))

    @torch.no_grad()
    def _act(self, envs, obs, rnn_states, masks):
        """Deterministic actions of all agents -> [envs][agents] one-hot actions; rnn_states updated in place."""
        one_hot = []
        for agent_id, tr in enumerate(self.trainer):
            tr.prep_rollout()
            action, rnn_state = tr.policy.act(_agent_obs(obs, agent_id), rnn_states[:, agent_id], masks[:, agent_id],
                                              deterministic=True)
            one_hot.append(_one_hot(envs.action_space[agent_id], _t2n(action).astype(np.int64)))
            rnn_states[:, agent_id] = _t2n(rnn_state)
        return [[one_hot[a][i] for a in range(self.num_agents)] for i in range(rnn_states.shape[0])]

    @torch.no_grad()
    def eval(self, total_num_steps):
        n = self.n_eval_rollout_threads
        eval_episode_rewards = []
        eval_obs = self.eval_envs.reset()
        eval_rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
        eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        for _ in range(self.episode_length):
            eval_actions_env = self._act(self.eval_envs, eval_obs, eval_rnn_states, eval_masks)
            eval_obs, eval_rewards, eval_dones, _ = self.eval_envs.step(eval_actions_env)
            eval_episode_rewards.append(eval_rewards)
            done = np.asarray(eval_dones, dtype=bool)
            eval_rnn_states[done] = 0.0
            eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            eval_masks[done] = 0.0
        eval_episode_rewards = np.array(eval_episode_rewards)
        eval_train_infos = []
        for agent_id in range(self.num_agents):
            mean = float(np.mean(np.sum(eval_episode_rewards[:, :, agent_id], axis=0)))
            eval_train_infos.append({'eval_average_episode_rewards': mean})
            print("eval average episode rewards of agent%i: " % agent_id + str(mean))
        self.log_train(eval_train_infos, total_num_steps)

    @torch.no_grad()
    def render(self):
        envs, n = self.envs, self.n_rollout_threads
        all_frames = []
        for _ in range(self.all_args.render_episodes):
            episode_rewards = []
            obs = envs.reset()
            if self.all_args.save_gifs:
                all_frames.append(envs.render('rgb_array')[0][0])
            rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
            masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            for _ in range(self.episode_length):
                t0 = time.time()
                obs, rewards, dones, _ = envs.step(self._act(envs, obs, rnn_states, masks))
                episode_rewards.append(rewards)
                done = np.asarray(dones, dtype=bool)
                rnn_states[done] = 0.0
                masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
                masks[done] = 0.0
                if self.all_args.save_gifs:
                    all_frames.append(envs.render('rgb_array')[0][0])
                    spare = self.all_args.ifi - (time.time() - t0)
                    if spare > 0:
                        time.sleep(spare)
            episode_rewards = np.array(episode_rewards)
            for agent_id in range(self.num_agents):
                print("eval average episode rewards of agent%i: " % agent_id
                      + str(np.mean(np.sum(episode_rewards[:, :, agent_id], axis=0))))
        if self.all_args.save_gifs:
            self._save_frames(all_frames)
