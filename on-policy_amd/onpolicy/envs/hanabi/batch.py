"""ctypes binding of libhanabi_batch.so (include/hanabi_batch.h) and the vectorised env built on it.

``HanabiBatch`` owns N tables and numpy arrays the native calls fill in place.  ``HanabiBatchVecEnv`` exposes them
through the interface the turn-based runner uses on the reference's ``ChooseSubprocVecEnv`` (reference
onpolicy/envs/env_wrappers.py:661-704: ``reset(reset_choose)`` -> (obs, share_obs, available_actions),
``step(actions)`` -> (obs, share_obs, rewards, dones, infos, available_actions)) with the per-env semantics of the
reference's ``HanabiEnv`` (onpolicy/envs/hanabi/Hanabi_Env.py:278-312, :451-500): action -1 leaves a table alone and
yields zero rows, ``done = None`` and the current score; ``reset`` with ``choose`` False yields zero rows.
There are no worker processes and no pickled Python lists: one call advances every table.
"""
import ctypes
import os

import numpy as np

from onpolicy.envs.spaces import Discrete

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAPPO_HANABI_LIB", os.path.normpath(
    os.path.join(_HERE, "..", "..", "..", "lib", "libhanabi_batch.so")))

OBSERVATION_MINIMAL, OBSERVATION_CARD_KNOWLEDGE, OBSERVATION_SEER = 0, 1, 2
SHARE_OWN_HAND, SHARE_ALL_PLAYERS = 0, 1
_ERRORS = {-1: "bad argument", -2: "illegal move", -3: "table was never reset"}


class Rules(ctypes.Structure):
    """struct hanabi_rules (include/hanabi_batch.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("colors", "ranks", "players", "hand_size", "max_information_tokens",
                                              "max_life_tokens", "observation_type", "random_start_player")]


_vp, _i32 = ctypes.c_void_p, ctypes.c_int32
# symbol -> (restype, argtypes); must list every function include/hanabi_batch.h declares
SIGNATURES = {
    "hanabi_batch_create": (_vp, [ctypes.POINTER(Rules), _i32, _vp]),
    "hanabi_batch_destroy": (None, [_vp]),
    "hanabi_batch_tables": (_i32, [_vp]),
    "hanabi_batch_players": (_i32, [_vp]),
    "hanabi_batch_num_moves": (_i32, [_vp]),
    "hanabi_batch_obs_len": (_i32, [_vp]),
    "hanabi_batch_own_hand_len": (_i32, [_vp]),
    "hanabi_batch_failed_table": (_i32, [_vp]),
    "hanabi_batch_reset": (ctypes.c_int, [_vp, _vp]),
    "hanabi_batch_step": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "hanabi_batch_encode": (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "hanabi_batch_player_view": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "hanabi_batch_table_state": (ctypes.c_int, [_vp, _i32, _vp]),
}
_lib = None


def lib():
    """Load (once) and return the bound library; raises if it is not built -- there is no Python stand-in."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libhanabi_batch.so not found at %s -- build it with `make -C on-policy_amd/csrc` "
                               "(or __graft_entry__.build())" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


# the named games of the reference env (Hanabi_Env.py:117-160); hand_size 0 = by the rules (5 below four players)
GAMES = {
    "Hanabi-Full": dict(colors=5, ranks=5, hand_size=0, max_information_tokens=8, max_life_tokens=3,
                        observation_type=OBSERVATION_CARD_KNOWLEDGE),
    "Hanabi-Full-Minimal": dict(colors=5, ranks=5, hand_size=0, max_information_tokens=8, max_life_tokens=3,
                                observation_type=OBSERVATION_MINIMAL),
    "Hanabi-Small": dict(colors=2, ranks=5, hand_size=2, max_information_tokens=3, max_life_tokens=1,
                         observation_type=OBSERVATION_CARD_KNOWLEDGE),
    "Hanabi-Very-Small": dict(colors=1, ranks=5, hand_size=2, max_information_tokens=3, max_life_tokens=1,
                              observation_type=OBSERVATION_CARD_KNOWLEDGE),
}
GAMES["Hanabi-Full-CardKnowledge"] = GAMES["Hanabi-Full"]


def rules_for(hanabi_name, num_agents):
    if hanabi_name not in GAMES:
        raise ValueError("Unknown environment {}".format(hanabi_name))
    return dict(GAMES[hanabi_name], players=int(num_agents), random_start_player=0)


def _ptr(a):
    return None if a is None else a.ctypes.data


class HanabiBatch(object):
    """N tables under one set of rules; every array below is written in place by the native calls."""

    def __init__(self, rules, seeds, share_mode=SHARE_OWN_HAND):
        self._lib = lib()
        # the engine seeds std::mt19937, which takes the seed modulo 2^32: wider Python ints wrap the same way
        seeds = (np.asarray(seeds, dtype=np.int64).reshape(-1) % (1 << 32)).astype(np.uint32).view(np.int32)
        seeds = np.ascontiguousarray(seeds)
        self._rules = Rules(**rules)
        self._handle = self._lib.hanabi_batch_create(ctypes.byref(self._rules), len(seeds), _ptr(seeds))
        if not self._handle:
            raise ValueError("invalid Hanabi rules %r" % (rules,))
        L = self._lib
        self.n = len(seeds)
        self.players = L.hanabi_batch_players(self._handle)
        self.num_moves = L.hanabi_batch_num_moves(self._handle)
        self.obs_len = L.hanabi_batch_obs_len(self._handle)
        self.own_hand_len = L.hanabi_batch_own_hand_len(self._handle)
        self.share_mode = share_mode
        self.share_len = self.own_hand_len + self.obs_len if share_mode == SHARE_OWN_HAND else self.players * self.obs_len
        self.obs = np.zeros((self.n, self.obs_len + self.players), dtype=np.float32)
        self.share_obs = np.zeros((self.n, self.share_len + self.players), dtype=np.float32)
        self.available_actions = np.zeros((self.n, self.num_moves), dtype=np.float32)
        self.to_move = np.full(self.n, -1, dtype=np.int32)
        self.rewards = np.zeros(self.n, dtype=np.float32)
        self.status = np.zeros(self.n, dtype=np.uint8)
        self.scores = np.zeros(self.n, dtype=np.int32)

    def _check(self, code, what):
        if code != 0:
            detail = _ERRORS.get(code, "error %d" % code)
            if code == -2:
                detail += " on table %d" % self._lib.hanabi_batch_failed_table(self._handle)
            raise ValueError("%s: %s" % (what, detail))

    def reset(self, choose=None):
        mask = None if choose is None else np.ascontiguousarray(choose, dtype=np.uint8)
        self._check(self._lib.hanabi_batch_reset(self._handle, _ptr(mask)), "hanabi_batch_reset")

    def step(self, actions):
        """actions [n] int (move uid, -1 = leave alone) -> fills rewards / status / scores."""
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
        assert a.shape[0] == self.n
        self._check(self._lib.hanabi_batch_step(self._handle, _ptr(a), _ptr(self.rewards), _ptr(self.status),
                                                _ptr(self.scores)), "hanabi_batch_step")

    def encode(self, active=None):
        """Fills obs / share_obs / available_actions / to_move (zero rows where ``active`` is False)."""
        mask = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        self._check(self._lib.hanabi_batch_encode(self._handle, self.share_mode, _ptr(mask), _ptr(self.obs),
                                                  _ptr(self.share_obs), _ptr(self.available_actions),
                                                  _ptr(self.to_move)), "hanabi_batch_encode")

    def player_view(self, table, player):
        """(observation, own hand) of one player as the reference encoder returns them (int vectors)."""
        obs = np.zeros(self.obs_len, dtype=np.int32)
        own = np.zeros(self.own_hand_len, dtype=np.int32)
        self._check(self._lib.hanabi_batch_player_view(self._handle, table, player, _ptr(obs), _ptr(own)),
                    "hanabi_batch_player_view")
        return obs, own

    def table_state(self, table):
        out = np.zeros(8 + 5, dtype=np.int32)
        self._check(self._lib.hanabi_batch_table_state(self._handle, table, _ptr(out)), "hanabi_batch_table_state")
        keys = ("life_tokens", "information_tokens", "deck_size", "score", "current_player", "end_of_game",
                "turns_left", "discards")
        state = dict(zip(keys, (int(v) for v in out[:8])))
        state["fireworks"] = [int(v) for v in out[8:8 + self._rules.colors]]
        return state

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.hanabi_batch_destroy(self._handle)
            self._handle = None

    __del__ = close


class _ScoreInfos(object):
    """``infos[i] == {"score": s_i}`` without building N dicts per step (the runner reads the finished tables only)."""

    def __init__(self, scores):
        self._scores = scores.copy()

    def __len__(self):
        return len(self._scores)

    def __getitem__(self, i):
        return {"score": int(self._scores[i])}

    def __iter__(self):
        return ({"score": int(s)} for s in self._scores)


_DONE_OF_STATUS = np.array([False, True, None], dtype=object)      # status 2 = idle: the reference env's done = None


class HanabiBatchVecEnv(object):
    """All rollout threads of the turn-based Hanabi runner in one native batch (see the module docstring).
    ``seeds[i]`` is what the reference's train script passes to env i (train_hanabi_forward.py:24-26)."""

    def __init__(self, all_args, seeds, copy=True):
        """``copy=False`` returns the batch's own arrays (overwritten by the next reset / step) instead of fresh
        ones -- for callers that, like the turn-based runner, copy what they keep anyway."""
        self._out = (lambda a: a.copy()) if copy else (lambda a: a)
        rules = rules_for(all_args.hanabi_name, all_args.num_agents)
        share = SHARE_ALL_PLAYERS if all_args.use_obs_instead_of_state else SHARE_OWN_HAND
        self.batch = HanabiBatch(rules, seeds, share)
        b = self.batch
        self.num_envs = b.n
        self.players = b.players
        self.action_space = [Discrete(b.num_moves) for _ in range(b.players)]
        self.observation_space = [[b.obs_len + b.players] for _ in range(b.players)]
        self.share_observation_space = [[b.share_len + b.players] for _ in range(b.players)]

    def reset(self, reset_choose=None):
        b = self.batch
        choose = np.ones(b.n, dtype=bool) if reset_choose is None else np.asarray(reset_choose, dtype=bool)
        b.reset(choose)
        b.encode(choose)
        return self._out(b.obs), self._out(b.share_obs), self._out(b.available_actions)

    def step(self, actions):
        b = self.batch
        a = np.asarray(actions).reshape(b.n, -1)[:, 0].astype(np.int32)
        b.step(a)
        b.encode(a != -1)
        rewards = np.repeat(b.rewards[:, None, None], b.players, axis=1)
        dones = _DONE_OF_STATUS[b.status]
        infos = _ScoreInfos(b.scores)
        return self._out(b.obs), self._out(b.share_obs), rewards, dones, infos, self._out(b.available_actions)

    def close(self):
        self.batch.close()
