"""Vectorised-environment wrappers: N copies of a multi-agent env behind one batched ``reset`` / ``step``.

Class names, constructor arguments and call protocol of the reference's onpolicy/envs/env_wrappers.py
(ShareVecEnv :26, SubprocVecEnv :236, ShareSubprocVecEnv :338, ChooseSimpleSubprocVecEnv :430,
ChooseSubprocVecEnv :524, GuardSubprocVecEnv :176, ChooseGuardSubprocVecEnv :615, DummyVecEnv :683 and the
Share / Choose / ChooseSimple dummies :727-822), so the reference's ``make_train_env`` factories work against
this package.  The reference spells each of the twelve classes out; here they are one subprocess class and
one in-process class parameterised by the three things that actually differ:

  share   -- the env also returns a centralised observation and the available actions
             (step -> obs, share_obs, rewards, dones, infos, available_actions; reset -> obs, share_obs,
             available_actions), as SMAC / Hanabi / football do; otherwise step -> obs, rewards, dones, infos;
  choose  -- turn-based envs: ``reset(reset_choose)`` passes one flag per env and there is no automatic
             reset on ``done``; otherwise an env is reset inside ``step`` as soon as all its agents are done;
  daemon  -- whether the worker processes are daemonic (the "Guard" variants are not).

Envs run on the host; what they return is plain numpy that the runners copy into the HBM buffer.
"""
from abc import ABC, abstractmethod
from multiprocessing import Pipe, Process

import numpy as np

from onpolicy.utils.util import tile_images


class CloudpickleWrapper(object):
    """Carries an env factory across the process boundary (closures do not survive plain pickle)."""

    def __init__(self, x):
        self.x = x

    def __getstate__(self):
        import cloudpickle
        return cloudpickle.dumps(self.x)

    def __setstate__(self, ob):
        import pickle
        self.x = pickle.loads(ob)


def _all_done(done):
    return bool(done) if 'bool' in done.__class__.__name__ else bool(np.all(done))


class ShareVecEnv(ABC):
    """Batched env interface: ``reset()``, ``step(actions)`` (= ``step_async`` + ``step_wait``), ``close()``."""
    closed = False
    viewer = None
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, num_envs, observation_space, share_observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.share_observation_space = share_observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        if self.viewer is not None:
            self.viewer.close()
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode='human'):
        bigimg = tile_images(self.get_images())
        if mode == 'human':
            self.get_viewer().imshow(bigimg)
            return self.get_viewer().isopen
        if mode == 'rgb_array':
            return bigimg
        raise NotImplementedError

    def get_images(self):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self.venv.unwrapped if hasattr(self, "venv") else self

    def get_viewer(self):
        if self.viewer is None:
            from gym.envs.classic_control import rendering
            self.viewer = rendering.SimpleImageViewer()
        return self.viewer


# ------------------------------------------------------------------ one env, either side of a pipe
def _env_step(env, action, share, choose):
    out = env.step(action)
    if choose:
        return out
    if share:
        ob, s_ob, reward, done, info, avail = out
        if _all_done(done):
            ob, s_ob, avail = env.reset()
        return ob, s_ob, reward, done, info, avail
    ob, reward, done, info = out
    if _all_done(done):
        ob = env.reset()
    return ob, reward, done, info


def _serve(remote, parent_remote, env_fn_wrapper, share, choose):
    """Worker loop of the subprocess wrappers (reference worker :140, shareworker :296,
    choosesimpleworker :399, chooseworker :495, chooseguardworker :590)."""
    parent_remote.close()
    env = env_fn_wrapper.x()
    while True:
        cmd, data = remote.recv()
        if cmd == 'step':
            remote.send(_env_step(env, data, share, choose))
        elif cmd == 'reset':
            remote.send(env.reset(data) if choose else env.reset())
        elif cmd == 'render':
            if data == "rgb_array":
                remote.send(env.render(mode=data))
            elif data == "human":
                env.render(mode=data)
        elif cmd == 'reset_task':
            remote.send(env.reset_task())
        elif cmd == 'render_vulnerability':
            remote.send(env.render_vulnerability(data))
        elif cmd == 'get_spaces':
            remote.send((env.observation_space, env.share_observation_space, env.action_space))
        elif cmd == 'close':
            env.close()
            remote.close()
            break
        else:
            raise NotImplementedError(cmd)


def _stack_step(results, share):
    if share:
        obs, share_obs, rews, dones, infos, avail = zip(*results)
        return np.stack(obs), np.stack(share_obs), np.stack(rews), np.stack(dones), infos, np.stack(avail)
    obs, rews, dones, infos = zip(*results)
    return np.stack(obs), np.stack(rews), np.stack(dones), infos


def _stack_reset(results, share):
    if share:
        obs, share_obs, avail = zip(*results)
        return np.stack(obs), np.stack(share_obs), np.stack(avail)
    return np.stack(results)


class _SubprocVecEnv(ShareVecEnv):
    """One process per env, commands over pipes."""
    _share = False
    _choose = False
    _daemon = True      # if the main process dies the workers must not keep it hanging

    def __init__(self, env_fns, spaces=None):
        self.waiting = False
        self.closed = False
        self.remotes, self.work_remotes = zip(*[Pipe() for _ in range(len(env_fns))])
        self.ps = [Process(target=_serve, args=(work_remote, remote, CloudpickleWrapper(env_fn), self._share,
                                                self._choose))
                   for work_remote, remote, env_fn in zip(self.work_remotes, self.remotes, env_fns)]
        for p in self.ps:
            p.daemon = self._daemon
            p.start()
        for remote in self.work_remotes:
            remote.close()
        self.remotes[0].send(('get_spaces', None))
        observation_space, share_observation_space, action_space = self.remotes[0].recv()
        ShareVecEnv.__init__(self, len(env_fns), observation_space, share_observation_space, action_space)

    def step_async(self, actions):
        for remote, action in zip(self.remotes, actions):
            remote.send(('step', action))
        self.waiting = True

    def step_wait(self):
        results = [remote.recv() for remote in self.remotes]
        self.waiting = False
        return _stack_step(results, self._share)

    def reset(self, reset_choose=None):
        if self._choose:
            for remote, choose in zip(self.remotes, reset_choose):
                remote.send(('reset', choose))
        else:
            for remote in self.remotes:
                remote.send(('reset', None))
        return _stack_reset([remote.recv() for remote in self.remotes], self._share)

    def reset_task(self):
        for remote in self.remotes:
            remote.send(('reset_task', None))
        return np.stack([remote.recv() for remote in self.remotes])

    def render(self, mode="rgb_array"):
        for remote in self.remotes:
            remote.send(('render', mode))
        if mode == "rgb_array":
            return np.stack([remote.recv() for remote in self.remotes])

    def close(self):
        if self.closed:
            return
        if self.waiting:
            for remote in self.remotes:
                remote.recv()
        for remote in self.remotes:
            remote.send(('close', None))
        for p in self.ps:
            p.join()
        self.closed = True


class _DummyVecEnv(ShareVecEnv):
    """All envs in this process, stepped one after the other."""
    _share = False
    _choose = False

    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        env = self.envs[0]
        ShareVecEnv.__init__(self, len(env_fns), env.observation_space, env.share_observation_space,
                             env.action_space)
        self.actions = None

    def step_async(self, actions):
        self.actions = actions

    def step_wait(self):
        results = [_env_step(env, a, self._share, self._choose) for a, env in zip(self.actions, self.envs)]
        self.actions = None
        return tuple(map(np.array, zip(*results)))

    def reset(self, reset_choose=None):
        if self._choose:
            results = [env.reset(choose) for env, choose in zip(self.envs, reset_choose)]
        else:
            results = [env.reset() for env in self.envs]
        if self._share:
            return tuple(map(np.array, zip(*results)))
        return np.array(results)

    def close(self):
        for env in self.envs:
            env.close()

    def render(self, mode="human"):
        if mode == "rgb_array":
            return np.array([env.render(mode=mode) for env in self.envs])
        if mode == "human":
            for env in self.envs:
                env.render(mode=mode)
            return None
        raise NotImplementedError


# ------------------------------------------------------------------ the reference's class names
class SubprocVecEnv(_SubprocVecEnv):
    """MPE-style envs (obs, rewards, dones, infos), auto-reset."""


class GuardSubprocVecEnv(_SubprocVecEnv):
    _daemon = False


class ShareSubprocVecEnv(_SubprocVecEnv):
    """SMAC / football-style envs (+ share_obs, available_actions), auto-reset."""
    _share = True


class ChooseSimpleSubprocVecEnv(_SubprocVecEnv):
    _choose = True


class ChooseSubprocVecEnv(_SubprocVecEnv):
    """Hanabi-style turn-based envs: share protocol, ``reset(reset_choose)``."""
    _share = True
    _choose = True


class ChooseGuardSubprocVecEnv(_SubprocVecEnv):
    _choose = True
    _daemon = False


class DummyVecEnv(_DummyVecEnv):
    pass


class ShareDummyVecEnv(_DummyVecEnv):
    _share = True


class ChooseDummyVecEnv(_DummyVecEnv):
    _share = True
    _choose = True


class ChooseSimpleDummyVecEnv(_DummyVecEnv):
    _choose = True
