"""Runner for turn-based Hanabi (``train_hanabi_forward.py``).  Interface of the reference's
onpolicy/runner/shared/hanabi_runner_forward.py (HanabiRunner: run :20, warmup :125, collect :138,
train :222, eval :229).

Hanabi is turn based: within one buffer "step" the agents act one after the other, only the envs
whose current player has a legal move take part (``choose``), and a reward arrives only after the
*other* players have moved.  The reference keeps that bookkeeping in a dozen numpy arrays shaped like
a buffer row and re-uploads all of them with every ``chooseinsert``.  Here the turn state lives on the
policy's device (``_TurnState``: one tensor per buffer field, [N, A, ...]):

  * per player move the only host -> device traffic is what the env just produced -- the movers'
    observation / centralised observation / legal-move rows, the rewards and the done flags -- packed
    into one page-locked staging buffer and sent as ONE asynchronous copy (``_HostStage``);
  * the policy's outputs never leave the device except the action indices the env needs;
  * the reference's masked numpy assignments become ``index_put`` / ``where`` updates of the turn
    tensors (no boolean-mask indexing on the device: that would synchronise);
  * ``chooseinsert`` then is a device-to-device slab write (K2), the reward shift before each update
    (reference :59-63) device copies, ``compute`` / ``train`` the GAE kernel and the fused update.

The host keeps what decides control flow: the legal-move table (who moves), the done flags (which
games reset) and the scores.  Control flow and the order of the env calls are the reference's.
"""
import time

import numpy as np
import torch

from onpolicy.runner.shared.base_runner import Runner, _t2n


class _TurnState(object):
    """Per-thread data of the turn in progress, shaped like one buffer row [N, A, ...], on ``device``."""

    def __init__(self, n, buffer, device):
        z = lambda t: torch.zeros((n,) + tuple(t.shape[2:]), dtype=torch.float32, device=device)
        o = lambda t: torch.ones((n,) + tuple(t.shape[2:]), dtype=torch.float32, device=device)
        self.obs, self.share_obs = z(buffer.obs), z(buffer.share_obs)
        self.available_actions = z(buffer.available_actions)
        self.values, self.actions = z(buffer.value_preds), z(buffer.actions)
        self.action_log_probs = z(buffer.action_log_probs)
        self.rnn_states, self.rnn_states_critic = z(buffer.rnn_states), z(buffer.rnn_states_critic)
        self.masks, self.active_masks, self.bad_masks = o(buffer.masks), o(buffer.masks), o(buffer.masks)
        self.rewards = z(buffer.rewards)
        self.rewards_since_last_action = z(buffer.rewards)


class _HostStage(object):
    """Host arrays of one env call -> device tensors through one pinned staging buffer and one async copy (two
    buffers alternate; an event per buffer guards reuse).  On a CPU device: plain tensor views of copies."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs, self.turn = None, 0

    def upload(self, arrays):
        """arrays: list of numpy arrays -> list of float32 device tensors of the same shapes."""
        arrays = [np.asarray(a) for a in arrays]
        if self.device.type != "cuda":
            return [torch.from_numpy(np.array(a, dtype=np.float32)) for a in arrays]
        total = sum(a.size for a in arrays)
        if self.bufs is None or self.bufs[0][0].numel() < total:
            self.bufs = [(torch.empty(total, dtype=torch.float32, pin_memory=True),
                          torch.empty(total, dtype=torch.float32, device=self.device), [None]) for _ in range(2)]
        pin, dev, done = self.bufs[self.turn]
        self.turn = 1 - self.turn
        if done[0] is not None:
            done[0].synchronize()
        pin_np, out, off = pin.numpy(), [], 0
        for a in arrays:
            np.copyto(pin_np[off:off + a.size].reshape(a.shape), a, casting="unsafe")
            out.append(dev[off:off + a.size].view(a.shape))
            off += a.size
        dev[:off].copy_(pin[:off], non_blocking=True)
        done[0] = torch.cuda.Event()
        done[0].record(torch.cuda.current_stream(self.device))
        return out


class HanabiRunner(Runner):
    def __init__(self, config):
        super(HanabiRunner, self).__init__(config)
        self.true_total_num_steps = 0

    def run(self):
        self.turn = _TurnState(self.n_rollout_threads, self.buffer, self.buffer.device)
        self._stage = _HostStage(self.buffer.device)
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads_job
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
                    b.share_obs[-1] = turn.share_obs
                    b.obs[-1] = turn.obs
                    b.available_actions[-1] = turn.available_actions
                    b.active_masks[-1] = turn.active_masks
                    # ... and every reward moves one step earlier (it is only known a turn later)
                    b.rewards[0:T - 1] = b.rewards[1:].clone()
                    b.rewards[-1] = turn.rewards
                    self.compute()
                    train_infos = self.train()

                b.chooseinsert(turn.share_obs, turn.obs, turn.rnn_states, turn.rnn_states_critic, turn.actions,
                               turn.action_log_probs, turn.values, turn.rewards, turn.masks, turn.bad_masks,
                               turn.active_masks, turn.available_actions)
                obs, share_obs, available_actions = self.envs.reset(self.reset_choose)
                share_obs = share_obs if self.use_centralized_V else obs
                rc = self.reset_choose
                if rc.any():        # only the restarted games' rows cross to the device
                    rows = torch.as_tensor(np.flatnonzero(rc), device=self.buffer.device)
                    o_d, s_d, a_d = self._stage.upload([obs[rc], share_obs[rc], available_actions[rc]])
                    self.use_obs[rows] = o_d
                    self.use_share_obs[rows] = s_d
                    self.use_available_actions[rows] = a_d
                    self.avail_host[rc] = available_actions[rc]

            total_num_steps = (episode + 1) * T * self.n_rollout_threads_job
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
        self._take_env_rows(obs, share_obs, available_actions)

    def _take_env_rows(self, obs, share_obs, available_actions, extra=()):
        """The env's current rows -> device (``use_*``); the legal-move table also stays on the host, where it
        decides who moves.  -> the device tensors of ``extra``."""
        up = self._stage.upload([obs, share_obs, available_actions] + list(extra))
        # clones: the staging area is reused two uploads later, these rows live until the next move
        self.use_obs, self.use_share_obs, self.use_available_actions = (t.clone() for t in up[:3])
        self.avail_host = np.array(available_actions, dtype=np.float32)
        return up[3:]

    @torch.no_grad()
    def collect(self, step):
        turn, n, A = self.turn, self.n_rollout_threads, self.num_agents
        dev = self.buffer.device
        for agent_id in range(A):
            env_actions = -np.ones((n,) + tuple(self.buffer.actions.shape[3:]), dtype=np.float32)
            choose = np.any(self.avail_host == 1, axis=1)    # envs whose player can move
            if not np.any(choose):
                self.reset_choose = np.ones(n, dtype=bool)
                break

            # rows of the envs that move: a plain slice (views, no gather) when every env does, else their indices
            every = bool(choose.all())
            sel_host = slice(None) if every else np.flatnonzero(choose)
            sel = slice(None) if every else torch.as_tensor(sel_host, device=dev)
            obs_c, share_c, avail_c = self.use_obs[sel], self.use_share_obs[sel], self.use_available_actions[sel]
            self.trainer.prep_rollout()
            value, action, action_log_prob, rnn_state, rnn_state_critic = self.trainer.policy.get_actions(
                share_c, obs_c, turn.rnn_states[sel, agent_id], turn.rnn_states_critic[sel, agent_id],
                turn.masks[sel, agent_id], avail_c)
            action_np = _t2n(action)             # the one device -> host transfer of a move: the env needs the actions
            to = dict(device=dev, dtype=torch.float32)
            turn.obs[sel, agent_id] = obs_c
            turn.share_obs[sel, agent_id] = share_c
            turn.available_actions[sel, agent_id] = avail_c
            turn.values[sel, agent_id] = value.to(**to)
            turn.actions[sel, agent_id] = action.to(**to)
            env_actions[sel_host] = action_np
            turn.action_log_probs[sel, agent_id] = action_log_prob.to(**to)
            turn.rnn_states[sel, agent_id] = rnn_state.to(**to)
            turn.rnn_states_critic[sel, agent_id] = rnn_state_critic.to(**to)

            obs, share_obs, rewards, dones, infos, available_actions = self.envs.step(env_actions)
            self.true_total_num_steps += int(choose.sum())
            share_obs = share_obs if self.use_centralized_V else obs
            done = np.asarray(dones) == True   # noqa: E712  (dones may hold None for idle envs)
            alive = np.asarray(dones) == False  # noqa: E712
            rewards_d, done_d, alive_d = self._take_env_rows(
                obs, share_obs, available_actions,
                extra=[np.asarray(rewards, dtype=np.float32).reshape(turn.rewards.shape), done, alive])
            done_d, alive_d = done_d > 0, alive_d > 0

            # the acting player collects what accumulated since its previous move; everybody accrues
            # the new reward (the reward of buffer step 0 is discarded by the shift in run())
            turn.rewards[sel, agent_id] = turn.rewards_since_last_action[sel, agent_id]
            turn.rewards_since_last_action[sel, agent_id] = 0.0
            if every:
                turn.rewards_since_last_action += rewards_d
            else:
                turn.rewards_since_last_action[sel] += rewards_d[sel]

            self.reset_choose[done] = True
            # running games: the current player stays live
            live_col = alive_d.view(n, 1)
            turn.masks[:, agent_id] = torch.where(live_col, torch.ones_like(turn.masks[:, agent_id]), turn.masks[:, agent_id])
            turn.active_masks[:, agent_id] = torch.where(live_col, torch.ones_like(turn.active_masks[:, agent_id]),
                                                         turn.active_masks[:, agent_id])
            if not done.any():
                continue
            # finished games: nobody may act, states restart, players after the current one are inactive
            self.avail_host[done] = 0.0
            row = lambda t: done_d.view((n,) + (1,) * (t.dim() - 1))
            self.use_available_actions.masked_fill_(row(self.use_available_actions), 0.0)
            turn.masks.masked_fill_(row(turn.masks), 0.0)
            turn.rnn_states.masked_fill_(row(turn.rnn_states), 0.0)
            turn.rnn_states_critic.masked_fill_(row(turn.rnn_states_critic), 0.0)
            turn.active_masks[:, agent_id] = torch.where(done_d.view(n, 1), torch.ones_like(turn.active_masks[:, agent_id]),
                                                         turn.active_masks[:, agent_id])
            rest = slice(agent_id + 1, A)
            if agent_id + 1 < A:
                d3 = done_d.view(n, 1, 1)
                turn.active_masks[:, rest] = torch.where(d3, torch.zeros_like(turn.active_masks[:, rest]),
                                                         turn.active_masks[:, rest])
                turn.rewards[:, rest] = torch.where(d3, turn.rewards_since_last_action[:, rest], turn.rewards[:, rest])
                turn.rewards_since_last_action[:, rest] = torch.where(
                    d3, torch.zeros_like(turn.rewards_since_last_action[:, rest]), turn.rewards_since_last_action[:, rest])
                turn.values[:, rest] = torch.where(d3, torch.zeros_like(turn.values[:, rest]), turn.values[:, rest])
                turn.obs[:, rest] = torch.where(d3, torch.zeros_like(turn.obs[:, rest]), turn.obs[:, rest])
                turn.share_obs[:, rest] = torch.where(d3, torch.zeros_like(turn.share_obs[:, rest]), turn.share_obs[:, rest])
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
