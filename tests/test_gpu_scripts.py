"""-m gpu: the SMAC / football train scripts end to end against tiny fake single-env classes placed in an external env
tree (MAPPO_ENVS_PATH), and the Hanabi train script on the in-tree engine (batched stepper and one env per thread)
-- flags, VecEnv wrappers, runner choice, logging and checkpoints."""
import json
import os
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

SMAC_ENV = '''
import numpy as np
from onpolicy.envs.spaces import Box, Discrete
class StarCraft2Env(object):
    def __init__(self, args):
        self.a, self.do, self.ds, self.na, self.t = 3, 7, 9, 6, 0
        self.observation_space = [Box(shape=(self.do,)) for _ in range(self.a)]
        self.share_observation_space = [Box(shape=(self.ds,)) for _ in range(self.a)]
        self.action_space = [Discrete(self.na) for _ in range(self.a)]
        self.rng = np.random.default_rng(0)
    def seed(self, s): self.rng = np.random.default_rng(s)
    def _emit(self):
        av = (self.rng.random((self.a, self.na)) < 0.6).astype(np.float32); av[:, 0] = 1
        self.av = av
        return (self.rng.standard_normal((self.a, self.do)).astype(np.float32),
                self.rng.standard_normal((self.a, self.ds)).astype(np.float32), av)
    def reset(self):
        self.t = 0
        return self._emit()
    def step(self, actions):
        act = np.asarray(actions).reshape(self.a).astype(int)
        assert all(self.av[i, act[i]] == 1 for i in range(self.a))
        self.t += 1
        obs, share, av = self._emit()
        done = self.t >= 5
        infos = [{"bad_transition": False, "battles_won": self.t // 5, "battles_game": 1 + self.t // 5, "won": True}
                 for _ in range(self.a)]
        return obs, share, np.ones((self.a, 1), np.float32), np.full(self.a, done), infos, av
    def close(self): pass
'''
SMAC_MAPS = "def get_map_params(name):\n    return {'n_agents': 3}\n"
FOOTBALL_ENV = '''
import numpy as np
from onpolicy.envs.spaces import Box, Discrete
class FootballEnv(object):
    def __init__(self, args):
        self.a, self.do, self.t = args.num_agents, 11, 0
        self.observation_space = [Box(shape=(self.do,)) for _ in range(self.a)]
        self.share_observation_space = [Box(shape=(self.do,)) for _ in range(self.a)]
        self.action_space = [Discrete(5) for _ in range(self.a)]
        self.rng = np.random.default_rng(0)
    def seed(self, s): self.rng = np.random.default_rng(s)
    def reset(self):
        self.t = 0
        return self.rng.standard_normal((self.a, self.do)).astype(np.float32)
    def step(self, actions):
        assert np.asarray(actions).shape == (self.a,)
        self.t += 1
        done = self.t >= 4
        info = {"score_reward": 1, "max_steps": 4, "steps_left": 4 - self.t}
        return (self.rng.standard_normal((self.a, self.do)).astype(np.float32), np.ones((self.a, 1), np.float32),
                np.full(self.a, done), info)
    def close(self): pass
'''
@pytest.fixture
def env_tree(tmp_path, monkeypatch):
    ext = tmp_path / "ext_envs"
    for pkg, files in (("starcraft2", {"StarCraft2_Env.py": SMAC_ENV, "smac_maps.py": SMAC_MAPS}),
                       ("football", {"Football_Env.py": FOOTBALL_ENV})):
        (ext / pkg).mkdir(parents=True)
        (ext / pkg / "__init__.py").write_text("")
        for name, body in files.items():
            (ext / pkg / name).write_text(textwrap.dedent(body))
    monkeypatch.setenv("MAPPO_ENVS_PATH", str(ext))
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    import onpolicy.envs as envs
    before = list(envs.__path__)
    envs._extend(envs.__path__)
    yield ext
    envs.__path__[:] = before
    for mod in [m for m in sys.modules if m.startswith(("onpolicy.envs.starcraft2", "onpolicy.envs.football"))]:
        del sys.modules[mod]


def _tags(runner):
    return {json.loads(l)["tag"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))}


COMMON = ["--n_rollout_threads", "1", "--episode_length", "10", "--num_env_steps", "30", "--ppo_epoch", "2",
          "--hidden_size", "16", "--use_wandb", "--log_interval", "1", "--n_training_threads", "1"]


@pytest.mark.parametrize("algo,share", [("rmappo", True), ("happo", False)])
def test_train_smac_script(env_tree, algo, share):
    from onpolicy.scripts.train import train_smac
    argv = ["--env_name", "StarCraft2", "--map_name", "fake3", "--algorithm_name", algo, "--num_mini_batch", "1",
            "--data_chunk_length", "5"] + COMMON + ([] if share else ["--share_policy"])
    runner = train_smac.main(argv)
    assert runner.num_agents == 3
    assert type(runner).__module__.endswith(("shared.smac_runner" if share else "separated.smac_runner"))
    tags = _tags(runner)
    assert ("value_loss" in tags) if share else ("agent0/value_loss" in tags)
    assert "incre_win_rate" in tags
    assert os.path.exists(os.path.join(runner.log_dir, "summary.json"))


def test_train_football_script(env_tree):
    from onpolicy.scripts.train import train_football
    runner = train_football.main(["--env_name", "Football", "--algorithm_name", "mappo", "--num_agents", "2",
                                  "--save_interval", "10", "--log_interval", "10"] + COMMON[:-4] +
                                 ["--use_wandb", "--n_training_threads", "1"])
    assert {"value_loss", "goal", "win_rate"} <= _tags(runner)
    assert os.path.exists(os.path.join(runner.save_dir, "actor.pt"))


@pytest.mark.parametrize("layout", [[], ["--use_subproc_envs"]])
def test_train_hanabi_script(tmp_path, monkeypatch, layout):
    """Real games (in-tree engine): HanabiBatchVecEnv by default, ChooseSubprocVecEnv over HanabiEnv on request."""
    from onpolicy.scripts.train import train_hanabi_forward
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    runner = train_hanabi_forward.main(["--env_name", "Hanabi", "--hanabi_name", "Hanabi-Very-Small", "--num_agents", "2",
                                        "--algorithm_name", "mappo", "--n_rollout_threads", "2", "--episode_length", "6",
                                        "--num_env_steps", "48", "--ppo_epoch", "2", "--hidden_size", "16",
                                        "--use_wandb", "--log_interval", "1", "--n_training_threads", "1"] + layout)
    assert type(runner.envs).__name__ == ("ChooseSubprocVecEnv" if layout else "HanabiBatchVecEnv")
    assert runner.true_total_num_steps > 0
    assert {"value_loss", "average_score"} <= _tags(runner)
    runner.envs.close()
