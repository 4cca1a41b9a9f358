"""The VecEnv wrappers (onpolicy/envs/env_wrappers.py): every reference class name exists, subprocess and
in-process variants return identical batches, envs are reset inside ``step`` when all agents are done
(non-choose protocols) and only on request with per-env flags (choose protocols)."""
import functools

import numpy as np
import pytest

from fake_envs import TinyEnv
from onpolicy.envs import env_wrappers as W

NAMES = ["CloudpickleWrapper", "ShareVecEnv", "SubprocVecEnv", "GuardSubprocVecEnv", "ShareSubprocVecEnv",
         "ChooseSimpleSubprocVecEnv", "ChooseSubprocVecEnv", "ChooseGuardSubprocVecEnv", "DummyVecEnv",
         "ShareDummyVecEnv", "ChooseDummyVecEnv", "ChooseSimpleDummyVecEnv"]


def test_reference_class_names_exist():
    for name in NAMES:
        assert hasattr(W, name), name


def _fns(n, **kw):
    return [functools.partial(TinyEnv, seed=10 + i, **kw) for i in range(n)]


def _same(a, b):
    if isinstance(a, (tuple, list)) and not isinstance(a, np.ndarray):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif isinstance(a, dict):
        assert a == b
    else:
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("share,sub,dummy", [(False, "SubprocVecEnv", "DummyVecEnv"),
                                             (True, "ShareSubprocVecEnv", "ShareDummyVecEnv"),
                                             (False, "GuardSubprocVecEnv", "DummyVecEnv")])
def test_auto_reset_protocols(share, sub, dummy):
    n = 3
    venv_s, venv_d = getattr(W, sub)(_fns(n, share=share)), getattr(W, dummy)(_fns(n, share=share))
    try:
        assert venv_s.num_envs == n and len(venv_s.action_space) == 2
        _same(venv_s.reset(), venv_d.reset())
        rng = np.random.default_rng(0)
        for t in range(1, 8):
            actions = rng.integers(0, 4, size=(n, 2, 1))
            out_s, out_d = venv_s.step(actions), venv_d.step(actions)
            assert len(out_s) == (6 if share else 4)
            _same(out_s, out_d)
            obs, dones = out_d[0], out_d[3 if share else 2]
            assert obs.shape == (n, 2, 3) and dones.shape == (n, 2)
            # horizon 3: every third step all agents are done and the returned obs is the reset's
            assert bool(dones.all()) == (t % 3 == 0)
            assert bool((obs[:, 0, 0] > 999).all()) == (t % 3 == 0)
    finally:
        venv_s.close()
        venv_d.close()


@pytest.mark.parametrize("share,sub,dummy", [(True, "ChooseSubprocVecEnv", "ChooseDummyVecEnv"),
                                             (False, "ChooseSimpleSubprocVecEnv", "ChooseSimpleDummyVecEnv"),
                                             (False, "ChooseGuardSubprocVecEnv", "ChooseSimpleDummyVecEnv")])
def test_choose_protocols(share, sub, dummy):
    n = 3
    venv_s = getattr(W, sub)(_fns(n, share=share, choose=True))
    venv_d = getattr(W, dummy)(_fns(n, share=share, choose=True))
    try:
        choose = np.array([True, False, True])
        out_s, out_d = venv_s.reset(choose), venv_d.reset(choose)
        _same(out_s, out_d)
        obs = out_d[0] if share else out_d
        assert float(np.abs(obs[1]).sum()) == 0.0 and obs[0, 0, 0] > 999
        for t in range(1, 5):
            actions = np.zeros((n, 2, 1))
            out_s, out_d = venv_s.step(actions), venv_d.step(actions)
            _same(out_s, out_d)
            obs = out_d[0]
            assert not bool((obs[:, 0, 0] > 999).any())       # no automatic reset, even past the horizon
    finally:
        venv_s.close()
        venv_d.close()


def test_external_env_tree_is_a_fallback(tmp_path, monkeypatch):
    """MAPPO_ENVS_PATH: env packages missing here resolve from an external tree; ours win when both exist."""
    import importlib
    import subprocess
    import sys
    ext = tmp_path / "envs"
    (ext / "starcraft2").mkdir(parents=True)
    (ext / "starcraft2" / "__init__.py").write_text("NAME = 'external smac'\n")
    # an external MPE tree in the reference's layout: scenarios.load(name).Scenario + environment.MultiAgentEnv
    (ext / "mpe" / "scenarios").mkdir(parents=True)
    (ext / "mpe" / "scenarios" / "__init__.py").write_text(
        "import types\n"
        "def load(name):\n"
        "    m = types.SimpleNamespace()\n"
        "    class Scenario:\n"
        "        def make_world(self, args): return 'world of ' + name\n"
        "        reset_world = reward = observation = info = None\n"
        "    m.Scenario = Scenario\n"
        "    return m\n")
    (ext / "mpe" / "environment.py").write_text(
        "def MultiAgentEnv(world, *callbacks):\n    return 'external env: ' + world\n")
    (ext / "env_wrappers.py").write_text("raise RuntimeError('must not shadow the package module')\n")
    code = ("import types, onpolicy.envs.starcraft2 as s, onpolicy.envs.env_wrappers as w, onpolicy.envs.mpe.MPE_env as m, "
            "onpolicy.envs.mpe.simple_spread as ss; a = types.SimpleNamespace(scenario_name='simple_reference'); "
            "print(s.NAME, '|', m.MPEEnv(a), '|', hasattr(w, 'DummyVecEnv'), hasattr(ss, 'VecSimpleSpread'))")
    import os
    from conftest import ROOT
    env = dict(os.environ, MAPPO_ENVS_PATH=str(ext), PYTHONPATH=os.path.join(str(ROOT), "on-policy_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout.strip() == "external smac | external env: world of simple_reference.py | True True"


def test_mpe_env_factory_with_vec_wrappers():
    """MPEEnv(args) single worlds behind DummyVecEnv / SubprocVecEnv give the [N, A, .] batches the MPE runner
    expects and restart after episode_length steps; same seeds => same trajectories in both wrappers."""
    from types import SimpleNamespace
    from onpolicy.envs.mpe.MPE_env import MPEEnv
    args = SimpleNamespace(scenario_name="simple_spread", num_agents=3, num_landmarks=3, episode_length=4)

    def make(rank):
        def init():
            env = MPEEnv(args)
            env.seed(100 + rank)
            return env
        return init
    n = 3
    dummy, sub = W.DummyVecEnv([make(i) for i in range(n)]), W.SubprocVecEnv([make(i) for i in range(n)])
    try:
        o1, o2 = dummy.reset(), sub.reset()
        assert o1.shape == (n, 3, 18) and o1.dtype == np.float32
        np.testing.assert_array_equal(o1, o2)
        rng = np.random.default_rng(0)
        for t in range(1, 7):
            act = np.eye(5)[rng.integers(0, 5, (n, 3))]
            r1, r2 = dummy.step(act), sub.step(act)
            obs, rew, done, info = r1
            assert obs.shape == (n, 3, 18) and rew.shape == (n, 3, 1) and done.shape == (n, 3)
            assert bool(done.all()) == (t == 4)                 # episode_length 4, then a fresh episode
            for a, b in zip(r1[:3], r2[:3]):
                np.testing.assert_array_equal(a, b)
            assert "individual_reward" in info[0][0]
    finally:
        dummy.close()
        sub.close()
    with pytest.raises(NotImplementedError):
        MPEEnv(SimpleNamespace(scenario_name="simple_reference", num_agents=2, episode_length=4))


@pytest.mark.parametrize("name,share,choose", [
    ("DummyVecEnv", False, False), ("SubprocVecEnv", False, False), ("ShareDummyVecEnv", True, False),
    ("ShareSubprocVecEnv", True, False), ("ChooseDummyVecEnv", True, True), ("ChooseSubprocVecEnv", True, True),
    ("ChooseSimpleDummyVecEnv", False, True), ("ChooseSimpleSubprocVecEnv", False, True)])
def test_wrappers_match_reference_outputs(gold, name, share, choose):
    """Every array a reset / step of the reference's wrapper class returns on the tiny env with seeded actions
    (oracle/make_golden_wrappers.py) -- auto-reset timing, choose-resets, stacking -- reproduced bit for bit."""
    z = gold.npz("wrapper_cases")

    def flatten(x, prefix, out):
        if isinstance(x, (tuple, list)) and not (len(x) and isinstance(x[0], dict)) and not isinstance(x, np.ndarray):
            for i, y in enumerate(x):
                flatten(y, prefix + "_%d" % i, out)
        else:
            arr = np.asarray(x)
            if arr.dtype != object:
                out[prefix] = arr
    got = {}
    n = 3
    venv = getattr(W, name)([functools.partial(TinyEnv, seed=10 + i, share=share, choose=choose) for i in range(n)])
    try:
        rng = np.random.default_rng(0)
        flatten(venv.reset(np.array([True, False, True])) if choose else venv.reset(), name + "_reset", got)
        for t in range(7):
            actions = rng.integers(0, 4, size=(n, 2, 1))
            flatten(venv.step(actions), name + "_step%d" % t, got)
            if choose and t == 3:
                flatten(venv.reset(np.array([False, True, True])), name + "_reset_mid", got)
    finally:
        venv.close()
    expected = sorted(k for k in z.files if k.startswith(name + "_"))
    assert sorted(got) == expected
    for k in expected:
        np.testing.assert_array_equal(got[k], z[k], err_msg=k)
