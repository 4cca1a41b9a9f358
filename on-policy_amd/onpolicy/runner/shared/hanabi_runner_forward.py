"""Runner for turn-based Hanabi (``train_hanabi_forward.py``).  Interface of the reference's
onpolicy/runner/shared/hanabi_runner_forward.py (HanabiRunner: run :20, warmup :125, collect :138,
train :222, eval :229).

Hanabi is turn based: within one buffer "step" the agents act one after the other, only the envs
whose current player has a legal move take part (``choose``), and a reward arrives only after the
*other* players have moved.  That bookkeeping is per-environment host logic and stays in numpy (the
``turn_*`` arrays, one row per rollout thread); what goes through the HBM path is

  * ``buffer.chooseinsert`` / ``chooseafter_update`` (fused slab writes),
  * the in-buffer reward shift before each update (reference :59-63) as device copies,
  * ``compute`` (GAE kernel) and ``train`` (fused samplers + PPO update).

The turn state is kept in one small container instead of twelve attributes; the control flow and
the order of the env calls are the reference's.
"""
import time

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner, _t2n


class _TurnState(object):
    """Per-thread data of the turn in progress, shaped like one buffer row [N, A, ...]."""

    def __init__(self, n, buffer):
        z = lambda t: np.zeros((n,) + tuple(t.shape[2:]), dtype=np.float32)
        o = lambda t: np.ones((n,) + tuple(t.shape[2:]), dtype=np.float32)
        self.obs, self.share_obs = z(buffer.obs), z(buffer.share_obs)
        self.available_actions = z(buffer.available_actions)
        self.values, self.actions = z(buffer.value_preds), z(buffer.actions)
        self.action_log_probs = z(buffer.action_log_probs)
        self.rnn_states, self.rnn_states_critic = z(buffer.rnn_states), z(buffer.rnn_states_critic)
        self.masks, self.active_masks, self.bad_masks = o(buffer.masks), o(buffer.masks), o(buffer.masks)
        self.rewards = z(buffer.rewards)
        self.rewards_since_last_action = z(buffer.rewards)


class HanabiRunner(Runner):
    def __init__(self, config):
        super(HanabiRunner, self).__init__(config)
        self.true_total_num_steps = 0

    def run(self):
        self.turn = _TurnState(self.n_rollout_threads, self.buffer)
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads
        T = self.episode_length
        b, turn = self.buffer, self.turn
        train_infos = {}
        for episode in range(episodes):
            if self.use_linear_lr_decay:
                self.trainer.policy.lr_decay(episode, episodes)
            self.scores = []
            for step in range(T):
                self.reset_choose = np.zeros(self.n_rollout_threads, dtype=bool)
                self.collect(step)

                if step == 0 and episode > 0:
                    # the last buffer row gets the turn that has just been played ...
                    f32 = torch.float32
                    b.share_obs[-1] = torch.as_tensor(turn.share_obs, dtype=f32)
                    b.obs[-1] = torch.as_tensor(turn.obs, dtype=f32)
                    b.available_actions[-1] = torch.as_tensor(turn.available_actions, dtype=f32)
                    b.active_masks[-1] = torch.as_tensor(turn.active_masks, dtype=f32)
                    # ... and every reward moves one step earlier (it is only known a turn later)
                    b.rewards[0:T - 1] = b.rewards[1:].clone()
                    b.rewards[-1] = torch.as_tensor(turn.rewards, dtype=f32)
                    self.compute()
                    train_infos = self.train()

                b.chooseinsert(turn.share_obs, turn.obs, turn.rnn_states, turn.rnn_states_critic, turn.actions,
                               turn.action_log_probs, turn.values, turn.rewards, turn.masks, turn.bad_masks,
                               turn.active_masks, turn.available_actions)
                obs, share_obs, available_actions = self.envs.reset(self.reset_choose)
                share_obs = share_obs if self.use_centralized_V else obs
                rc = self.reset_choose
                self.use_obs[rc] = obs[rc]
                self.use_share_obs[rc] = share_obs[rc]
                self.use_available_actions[rc] = available_actions[rc]

            total_num_steps = (episode + 1) * T * self.n_rollout_threads
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0 and episode > 0:
                end = time.time()
                print("\n Env {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n"
                      .format(getattr(self.all_args, "hanabi_name", "?"), self.algorithm_name, self.experiment_name,
                              episode, episodes, total_num_steps, self.num_env_steps,
                              int(total_num_steps / (end - start))))
                if self.env_name == "Hanabi":
                    average_score = np.mean(self.scores) if len(self.scores) > 0 else 0.0
                    print("average score is {}.".format(average_score))
                    self._log_scalar('average_score', average_score, self.true_total_num_steps)
                train_infos["average_step_rewards"] = float(b.rewards.mean())
                self.log_train(train_infos, self.true_total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(self.true_total_num_steps)

    def warmup(self):
        self.reset_choose = np.ones(self.n_rollout_threads, dtype=bool)
        obs, share_obs, available_actions = self.envs.reset(self.reset_choose)
        share_obs = share_obs if self.use_centralized_V else obs
        self.use_obs = obs.copy()
        self.use_share_obs = share_obs.copy()
        self.use_available_actions = available_actions.copy()

    @torch.no_grad()
    def collect(self, step):
        turn, n, A = self.turn, self.n_rollout_threads, self.num_agents
        for agent_id in range(A):
            env_actions = -np.ones((n,) + tuple(self.buffer.actions.shape[3:]), dtype=np.float32)
            choose = np.any(self.use_available_actions == 1, axis=1)    # envs whose player can move
            if not np.any(choose):
                self.reset_choose = np.ones(n, dtype=bool)
                break

            # rows of the envs that move: a plain slice (views, no gather) when every env does, else their indices
            sel = slice(None) if choose.all() else np.flatnonzero(choose)
            obs_c, share_c, avail_c = self.use_obs[sel], self.use_share_obs[sel], self.use_available_actions[sel]
            self.trainer.prep_rollout()
            value, action, action_log_prob, rnn_state, rnn_state_critic = self.trainer.policy.get_actions(
                share_c, obs_c, turn.rnn_states[sel, agent_id], turn.rnn_states_critic[sel, agent_id],
                turn.masks[sel, agent_id], avail_c)
            action_np = _t2n(action)
            turn.obs[sel, agent_id] = obs_c
            turn.share_obs[sel, agent_id] = share_c
            turn.available_actions[sel, agent_id] = avail_c
            turn.values[sel, agent_id] = _t2n(value)
            turn.actions[sel, agent_id] = action_np
            env_actions[sel] = action_np
            turn.action_log_probs[sel, agent_id] = _t2n(action_log_prob)
            turn.rnn_states[sel, agent_id] = _t2n(rnn_state)
            turn.rnn_states_critic[sel, agent_id] = _t2n(rnn_state_critic)

            obs, share_obs, rewards, dones, infos, available_actions = self.envs.step(env_actions)
            self.true_total_num_steps += int(choose.sum())
            share_obs = share_obs if self.use_centralized_V else obs
            self.use_obs = obs.copy()
            self.use_share_obs = share_obs.copy()
            self.use_available_actions = available_actions.copy()

            # the acting player collects what accumulated since its previous move; everybody accrues
            # the new reward (the reward of buffer step 0 is discarded by the shift in run())
            turn.rewards[sel, agent_id] = turn.rewards_since_last_action[sel, agent_id]
            turn.rewards_since_last_action[sel, agent_id] = 0.0
            turn.rewards_since_last_action[sel] += rewards[sel]

            done = np.asarray(dones) == True   # noqa: E712  (dones may hold None for idle envs)
            alive = np.asarray(dones) == False  # noqa: E712
            self.reset_choose[done] = True
            turn.masks[alive, agent_id] = 1.0                 # running games: the current player stays live
            turn.active_masks[alive, agent_id] = 1.0
            if not done.any():
                continue
            # finished games: nobody may act, states restart, players after the current one are inactive
            self.use_available_actions[done] = 0.0
            turn.masks[done] = 0.0
            turn.rnn_states[done] = 0.0
            turn.rnn_states_critic[done] = 0.0
            turn.active_masks[done, agent_id] = 1.0
            rest = slice(agent_id + 1, A)
            turn.active_masks[done, rest] = 0.0
            turn.rewards[done, rest] = turn.rewards_since_last_action[done, rest]
            turn.rewards_since_last_action[done, rest] = 0.0
            turn.values[done, rest] = 0.0
            turn.obs[done, rest] = 0.0
            turn.share_obs[done, rest] = 0.0
            for i in np.flatnonzero(done):
                if 'score' in infos[i].keys():
                    self.scores.append(infos[i]['score'])

    def train(self):
        self.trainer.prep_training()
        train_infos = self.trainer.train(self.buffer)
        self.buffer.chooseafter_update()
        return train_infos

    @torch.no_grad()
    def _eval_games(self):
        """Play the eval envs to the end with the deterministic policy -> list of final scores."""
        n, envs = self.n_eval_rollout_threads, self.eval_envs
        scores = []
        obs, share_obs, available_actions = envs.reset(np.ones(n, dtype=bool))
        rnn_states = np.zeros((n,) + tuple(self.buffer.rnn_states.shape[2:]), dtype=np.float32)
        masks = np.ones((n, self.num_agents, 1), dtype=np.float32)
        while True:
            for agent_id in range(self.num_agents):
                actions = -np.ones((n, 1), dtype=np.float32)
                choose = np.any(available_actions == 1, axis=1)
                if not np.any(choose):
                    return scores
                self.trainer.prep_rollout()
                action, rnn_state = self.trainer.policy.act(obs[choose], rnn_states[choose, agent_id],
                                                            masks[choose, agent_id], available_actions[choose],
                                                            deterministic=True)
                actions[choose] = _t2n(action)
                rnn_states[choose, agent_id] = _t2n(rnn_state)
                obs, share_obs, rewards, dones, infos, available_actions = envs.step(actions)
                available_actions[np.asarray(dones) == True] = 0.0   # noqa: E712
                for d, info in zip(dones, infos):
                    if d and 'score' in info.keys():
                        scores.append(info['score'])

    @torch.no_grad()
    def eval(self, total_num_steps):
        eval_average_score = np.mean(self._eval_games())
        print("eval average score is {}.".format(eval_average_score))
        self._log_scalar('eval_average_score', eval_average_score, total_num_steps)

    @torch.no_grad()
    def eval_100k(self, eval_games=100000):
        scores = []
        for trial in range(int(eval_games / self.n_eval_rollout_threads)):
            print("trail is {}".format(trial))
            scores.extend(self._eval_games())
        eval_average_score = np.mean(scores)
        print("eval average score is {}.".format(eval_average_score))
        return eval_average_score
