"""Runner for the multi-agent particle envs (``train_mpe.py``).  Interface of the reference's
onpolicy/runner/shared/mpe_runner.py (MPERunner: run :16, warmup :81, collect :96, insert :125,
eval :142, render :192).

Per rollout step only two things cross the PCIe bus: the integer actions go to the (CPU) envs and
the new observations / rewards / dones come back.  Values, log-probs and RNN states never leave
the GPU -- ``collect`` reads device views of the buffer and ``insert`` hands device tensors to the
fused slab-write kernel (the reference moves every field to numpy and back each step,
mpe_runner.py:105-109).
"""
import time

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner, _t2n


def _one_hot_actions(space, actions):
    """Integer actions [N, A, k] -> the one-hot layout the MPE envs expect."""
    kind = space.__class__.__name__
    if kind == 'MultiDiscrete':
        parts = [np.eye(space.high[i] + 1)[actions[:, :, i]] for i in range(space.shape)]
        return np.concatenate(parts, axis=2)
    if kind == 'Discrete':
        return np.squeeze(np.eye(space.n)[actions], 2)
    raise NotImplementedError(kind)


class MPERunner(Runner):
    def __init__(self, config):
        super(MPERunner, self).__init__(config)

    def _share(self, obs, n_threads):
        """Centralised critic input: every agent sees the concatenation of all observations."""
        if self.use_centralized_V:
            share = obs.reshape(n_threads, -1)
            if torch.is_tensor(share):       # device-resident env: a broadcast view, materialised by the K2 slab write
                return share.unsqueeze(1).expand(-1, self.num_agents, -1)
            return np.expand_dims(share, 1).repeat(self.num_agents, axis=1)
        return obs

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads_job
        infos = []
        # worlds on the device: the whole step (policy forward, sampling, env step, bookkeeping) as one captured graph
        # launch + the slab write (runner/shared/rollout_graph.py); None = the eager loop below
        from onpolicy.runner.shared import rollout_graph
        self.rollout_graph = rollout_graph.build(self) if episodes > 0 else None
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                self.trainer.policy.lr_decay(episode, episodes)

            if self.rollout_graph is not None:
                self.trainer.prep_rollout()
                self.rollout_graph.begin_episode()
                for step in range(self.episode_length):
                    infos = self.rollout_graph.step()
            for step in range(self.episode_length if self.rollout_graph is None else 0):
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
                env_infos = {}
                if self.env_name == "MPE":
                    for agent_id in range(self.num_agents):
                        rews = [info[agent_id]['individual_reward'] for info in infos
                                if 'individual_reward' in info[agent_id].keys()]
                        env_infos['agent%i/individual_rewards' % agent_id] = rews
                train_infos["average_episode_rewards"] = float(self.buffer.rewards.mean()) * self.episode_length
                print("average episode rewards is {}".format(train_infos["average_episode_rewards"]))
                self.log_train(train_infos, total_num_steps)
                self.log_env(env_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def warmup(self):
        obs = self.envs.reset()
        self.buffer.share_obs[0] = torch.as_tensor(self._share(obs, self.n_rollout_threads), dtype=torch.float32)
        self.buffer.obs[0] = torch.as_tensor(obs, dtype=torch.float32)

    @torch.no_grad()
    def collect(self, step):
        self.trainer.prep_rollout()
        b = self.buffer
        value, action, action_log_prob, rnn_states, rnn_states_critic = self.trainer.policy.get_actions(
            self._rows(b.share_obs[step]), self._rows(b.obs[step]), self._rows(b.rnn_states[step]),
            self._rows(b.rnn_states_critic[step]), self._rows(b.masks[step]))
        values = self._per_env(value)
        actions = self._per_env(action)
        action_log_probs = self._per_env(action_log_prob)
        rnn_states = self._per_env(rnn_states)
        rnn_states_critic = self._per_env(rnn_states_critic)
        if getattr(self.envs, "device_resident", False):
            actions_env = actions            # env state lives on the device: the action indices never leave it
        else:
            actions_env = _one_hot_actions(self.envs.action_space[0], _t2n(actions))   # the one D2H copy
        return values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env

    def insert(self, data):
        obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        if torch.is_tensor(dones):
            alive = (~dones).to(dtype=torch.float32, device=self.buffer.device)
        else:
            alive = torch.as_tensor(~np.asarray(dones, dtype=bool), dtype=torch.float32, device=self.buffer.device)
        masks = alive.unsqueeze(-1)                                         # 0 where the episode ended
        # finished agents restart from a zero RNN state (reference mpe_runner.py:128-129)
        rnn_states = rnn_states * alive.view(*alive.shape, 1, 1)
        rnn_states_critic = rnn_states_critic * alive.view(*alive.shape, 1, 1)
        share_obs = self._share(obs, self.n_rollout_threads)
        self.buffer.insert(share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, values,
                           rewards, masks)

    @torch.no_grad()
    def _act(self, obs, rnn_states, masks, n_threads):
        action, rnn_states = self.trainer.policy.act(np.concatenate(obs), np.concatenate(rnn_states),
                                                     np.concatenate(masks), deterministic=True)
        actions = np.array(np.split(_t2n(action), n_threads))
        rnn_states = np.array(np.split(_t2n(rnn_states), n_threads))
        return actions, rnn_states

    @torch.no_grad()
    def eval(self, total_num_steps):
        n = self.n_eval_rollout_threads
        eval_episode_rewards = []
        eval_obs = self.eval_envs.reset()
        eval_rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
        eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        for _ in range(self.episode_length):
            self.trainer.prep_rollout()
            eval_actions, eval_rnn_states = self._act(eval_obs, eval_rnn_states, eval_masks, n)
            eval_obs, eval_rewards, eval_dones, _ = self.eval_envs.step(
                _one_hot_actions(self.eval_envs.action_space[0], eval_actions))
            eval_episode_rewards.append(eval_rewards)
            done = np.asarray(eval_dones, dtype=bool)
            eval_rnn_states[done] = 0.0
            eval_masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            eval_masks[done] = 0.0
        totals = np.sum(np.array(eval_episode_rewards), axis=0)
        print("eval average episode rewards of agent: " + str(np.mean(totals)))
        self.log_env({'eval_average_episode_rewards': totals}, total_num_steps)

    @torch.no_grad()
    def render(self):
        """Roll out the deterministic policy and show / record the frames."""
        envs = self.envs
        n = self.n_rollout_threads
        all_frames = []
        for _ in range(self.all_args.render_episodes):
            obs = envs.reset()
            if self.all_args.save_gifs:
                all_frames.append(envs.render('rgb_array')[0][0])
            else:
                envs.render('human')
            rnn_states = np.zeros((n, self.num_agents, self.recurrent_N, self.hidden_size), dtype=np.float32)
            masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
            episode_rewards = []
            for _ in range(self.episode_length):
                t0 = time.time()
                self.trainer.prep_rollout()
                actions, rnn_states = self._act(obs, rnn_states, masks, n)
                obs, rewards, dones, _ = envs.step(_one_hot_actions(envs.action_space[0], actions))
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
                else:
                    envs.render('human')
            print("average episode rewards is: " + str(np.mean(np.sum(np.array(episode_rewards), axis=0))))
        if self.all_args.save_gifs:
            self._save_frames(all_frames)
