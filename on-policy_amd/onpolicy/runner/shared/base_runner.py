"""Rollout / update loop shared by the shared-policy runners.

Same class, config contract (``all_args, envs, eval_envs, num_agents, device, run_dir[, render_envs]``)
and method names as the reference's onpolicy/runner/shared/base_runner.py (Runner :12, compute :120,
train :136, save :143, restore :153, log_train :164, log_env :176), so the reference's train scripts
construct and drive it unchanged.  Differences:

  * the buffer lives in HBM: ``compute`` feeds the critic with device views of the last buffer row
    ([N, A, D] -> [N*A, D] is a free reshape; the reference concatenates / splits numpy copies and
    round-trips through the host, base_runner.py:124-134) and hands the bootstrap values to the GAE
    kernel as a device tensor;
  * wandb / tensorboardX are optional: without them scalars go to ``<run_dir>/logs/scalars.jsonl``
    through a writer object with the tensorboardX method names the scripts call
    (``add_scalars``, ``export_scalars_to_json``, ``close``).
"""
import json
import os

import numpy as np
import torch

from onpolicy.utils.shared_buffer import SharedReplayBuffer

try:  # optional logging back ends
    import wandb  # noqa: F401
except Exception:  # pragma: no cover - not installed in the build image
    wandb = None
try:
    from tensorboardX import SummaryWriter as _SummaryWriter
except Exception:  # pragma: no cover
    _SummaryWriter = None


def _t2n(x):
    """Tensor -> numpy (device -> host)."""
    return x.detach().cpu().numpy()


class JsonlWriter(object):
    """Minimal stand-in for tensorboardX.SummaryWriter."""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        self._path = os.path.join(log_dir, "scalars.jsonl")
        self._scalars = {}

    def add_scalars(self, main_tag, tag_scalar_dict, global_step=None):
        rec = {"tag": main_tag, "step": global_step}
        for k, v in tag_scalar_dict.items():
            val = float(v)
            rec[k] = val
            self._scalars.setdefault(main_tag + "/" + k, []).append([global_step, val])
        with open(self._path, "a") as f:
            f.write(json.dumps(rec) + "\n")

    def export_scalars_to_json(self, path):
        with open(path, "w") as f:
            json.dump(self._scalars, f)

    def close(self):
        pass


class Runner(object):
    """Base class for training recurrent policies.
    :param config: (dict) Config dictionary containing parameters for training."""

    def __init__(self, config):
        self._init_common(config)
        self._init_learner()

    def _init_common(self, config):
        """Config unpacking, run / log / model directories and the scalar writer (reference
        base_runner.py:17-66; the separated runner's are the same, runner/separated/base_runner.py:17-66)."""
        self.all_args = config['all_args']
        self.envs = config['envs']
        self.eval_envs = config['eval_envs']
        self.device = config['device']
        self.num_agents = config['num_agents']
        if 'render_envs' in config:
            self.render_envs = config['render_envs']

        a = self.all_args
        # --sampler_rng host = integer-parity mode: minibatch permutations AND the action
        # noise are drawn on the CPU generator like the reference does, so a device run reproduces its action stream
        from onpolicy.algorithms.utils import distributions
        distributions.set_sampling_rng(getattr(a, "sampler_rng", "device"))
        self.env_name = a.env_name
        self.algorithm_name = a.algorithm_name
        self.experiment_name = a.experiment_name
        self.use_centralized_V = a.use_centralized_V
        self.use_obs_instead_of_state = a.use_obs_instead_of_state
        self.num_env_steps = a.num_env_steps
        self.episode_length = a.episode_length
        self.n_rollout_threads = a.n_rollout_threads
        # a data-parallel job (scripts/train/_launch.py: device_of) keeps this rank's share in n_rollout_threads;
        # --num_env_steps is a budget of the WHOLE job, so episode counts, the learning-rate schedule and the step
        # counters of the logs use the job-wide count -- identical on every rank (they must agree on the number of
        # train() calls: every one of them is a collective)
        self.n_rollout_threads_job = getattr(a, "global_n_rollout_threads", a.n_rollout_threads)
        self.n_eval_rollout_threads = a.n_eval_rollout_threads
        self.n_render_rollout_threads = a.n_render_rollout_threads
        self.use_linear_lr_decay = a.use_linear_lr_decay
        self.hidden_size = a.hidden_size
        self.use_wandb = a.use_wandb and wandb is not None
        self.use_render = a.use_render
        self.recurrent_N = a.recurrent_N
        self.save_interval = a.save_interval
        self.use_eval = a.use_eval
        self.eval_interval = a.eval_interval
        self.log_interval = a.log_interval
        self.model_dir = a.model_dir

        if self.use_render:      # reference base_runner.py:48-52
            self.gif_dir = str(os.path.join(str(config["run_dir"]), 'gifs'))
            os.makedirs(self.gif_dir, exist_ok=True)
        if self.use_wandb and getattr(wandb, "run", None) is not None:
            self.save_dir = str(wandb.run.dir)
            self.run_dir = str(wandb.run.dir)
        else:
            self.use_wandb = False
            self.run_dir = config["run_dir"]
            self.log_dir = str(os.path.join(str(self.run_dir), 'logs'))
            os.makedirs(self.log_dir, exist_ok=True)
            self.writter = _SummaryWriter(self.log_dir) if _SummaryWriter is not None else JsonlWriter(self.log_dir)
            self.save_dir = str(os.path.join(str(self.run_dir), 'models'))
            os.makedirs(self.save_dir, exist_ok=True)

    _mat = False          # True when the learner is the Multi-Agent Transformer (set by _init_learner)

    def _init_learner(self):
        """One policy / trainer / HBM buffer shared by all agents (reference base_runner.py:68-108)."""
        a = self.all_args
        self._mat = self.algorithm_name in ("mat", "mat_dec")
        if self._mat:
            from onpolicy.algorithms.mat.mat_trainer import MATTrainer as TrainAlgo
            from onpolicy.algorithms.mat.algorithm.transformer_policy import TransformerPolicy as Policy
        else:
            from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO as TrainAlgo
            from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy as Policy

        share_observation_space = self.envs.share_observation_space[0] if self.use_centralized_V \
            else self.envs.observation_space[0]
        print("obs_space: ", self.envs.observation_space)
        print("share_obs_space: ", self.envs.share_observation_space)
        print("act_space: ", self.envs.action_space)

        extra = (self.num_agents,) if self._mat else ()       # the transformer is built for a fixed team size
        self.policy = Policy(a, self.envs.observation_space[0], share_observation_space,
                             self.envs.action_space[0], *extra, device=self.device)
        if self.model_dir is not None:
            self.restore(self.model_dir)
        self.trainer = TrainAlgo(a, self.policy, *extra, device=self.device)
        self.buffer = SharedReplayBuffer(a, self.num_agents, self.envs.observation_space[0],
                                         share_observation_space, self.envs.action_space[0],
                                         device=self.device if torch.device(self.device).type == "cuda" else None)

    # -- hooks of the concrete runners
    def run(self):
        raise NotImplementedError

    def warmup(self):
        raise NotImplementedError

    def collect(self, step):
        raise NotImplementedError

    def insert(self, data):
        raise NotImplementedError

    # -- helpers shared by the concrete runners
    def _rows(self, x):
        """[N, A, ...] buffer slab -> [N*A, ...] (a view: what np.concatenate did on the host)."""
        return x.reshape(x.shape[0] * x.shape[1], *x.shape[2:])

    def _per_env(self, x, n_threads=None):
        """[N*A, ...] network output -> [N, A, ...] (what np.split + np.array did on the host)."""
        n = self.n_rollout_threads if n_threads is None else n_threads
        return x.reshape(n, -1, *x.shape[1:])

    @torch.no_grad()
    def compute(self):
        """Bootstrap value of the last state, then returns / advantages for the whole rollout."""
        self.trainer.prep_rollout()
        b = self.buffer
        inputs = [self._rows(b.share_obs[-1]), self._rows(b.rnn_states_critic[-1]), self._rows(b.masks[-1])]
        if self._mat:                     # the transformer's critic reads the observations (base_runner.py:124-128)
            inputs.insert(1, self._rows(b.obs[-1]))
        next_values = self.trainer.policy.get_values(*inputs)
        b.compute_returns(self._per_env(next_values), self.trainer.value_normalizer)

    def train(self):
        """One PPO update phase on the collected rollout."""
        self.trainer.prep_training()
        train_infos = self.trainer.train(self.buffer)
        self.buffer.after_update()
        return train_infos

    def save(self, episode=0):
        """actor.pt / critic.pt state dicts (transformer_<episode>.pt for MAT), the reference's checkpoint format."""
        if self._mat:
            return self.policy.save(self.save_dir, episode)
        torch.save(self.trainer.policy.actor.state_dict(), str(self.save_dir) + "/actor.pt")
        torch.save(self.trainer.policy.critic.state_dict(), str(self.save_dir) + "/critic.pt")

    def restore(self, model_dir):
        if self._mat:
            return self.policy.restore(model_dir)
        self.policy.actor.load_state_dict(torch.load(str(model_dir) + '/actor.pt', map_location=self.device))
        if not self.all_args.use_render:
            self.policy.critic.load_state_dict(torch.load(str(model_dir) + '/critic.pt', map_location=self.device))

    def _save_frames(self, frames):
        """render.gif through imageio when it is installed, else the raw frames as render.npz."""
        try:
            import imageio
        except ImportError:
            path = str(self.gif_dir) + '/render.npz'
            np.savez_compressed(path, frames=np.asarray(frames), ifi=self.all_args.ifi)
            print("imageio is not installed: frames written to " + path)
            return path
        path = str(self.gif_dir) + '/render.gif'
        imageio.mimsave(path, frames, duration=self.all_args.ifi)
        return path

    def _log_scalar(self, key, value, step):
        if self.use_wandb:
            wandb.log({key: value}, step=step)
        else:
            self.writter.add_scalars(key, {key: value}, step)

    def log_train(self, train_infos, total_num_steps):
        for k, v in train_infos.items():
            self._log_scalar(k, float(v), total_num_steps)

    def log_env(self, env_infos, total_num_steps):
        for k, v in env_infos.items():
            if len(v) > 0:
                self._log_scalar(k, float(np.mean(v)), total_num_steps)
