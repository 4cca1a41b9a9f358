#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/hanabi_cases.npz by playing the REFERENCE's Hanabi env --
onpolicy/envs/hanabi/Hanabi_Env.py over pyhanabi.py over the reference's C++ engine, compiled from the sources where
they lie into oracle/_ref/libpyhanabi.so (`make -C oracle ref`) -- and recording every returned observation,
centralised observation, legal-action mask, reward, done flag and score together with the actions taken.

Three things the container lacks are stood in for, none of them arithmetic: ``cffi`` (oracle/cffi_ctypes.py, a
ctypes shim for the four FFI calls pyhanabi.py makes), ``gym.spaces.Discrete`` and the ``np.int`` alias that
numpy 2 removed (Hanabi_Env.py:293).

    make -C oracle ref && python oracle/make_golden_hanabi.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE_ROOT = os.environ.get("MAPPO_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, HERE)


def load_reference_env():
    import cffi_ctypes
    sys.modules["cffi"] = cffi_ctypes
    gym, spaces = types.ModuleType("gym"), types.ModuleType("gym.spaces")

    class Discrete(object):
        def __init__(self, n):
            self.n = n
    spaces.Discrete = Discrete
    gym.spaces = spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    if not hasattr(np, "int"):
        np.int = int
    for name, sub in (("onpolicy", "onpolicy"), ("onpolicy.envs", "onpolicy/envs"),
                      ("onpolicy.envs.hanabi", "onpolicy/envs/hanabi")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REFERENCE_ROOT, sub)]
        sys.modules[name] = mod
    from onpolicy.envs.hanabi import pyhanabi
    assert pyhanabi.try_cdef(prefixes=[os.path.join(REFERENCE_ROOT, "onpolicy/envs/hanabi")])
    assert pyhanabi.try_load(prefixes=[os.path.join(HERE, "_ref")]), "run `make -C oracle ref` first"
    from onpolicy.envs.hanabi.Hanabi_Env import HanabiEnv
    return HanabiEnv


def pick_action(env, avail, rng, skill):
    """uid of a legal move.  ``skill`` is the probability of peeking at the true hands and playing a card that fits
    the fireworks (long games: empty deck, finished stacks, regained tokens); otherwise uniform over the legal moves
    (short games: lost lives)."""
    legal = np.nonzero(avail)[0]
    if rng.random() < skill:
        state = env.state
        me = state.cur_player()
        fireworks = state.fireworks()
        hand = state.player_hands()[me]
        h = env.game.hand_size()
        for i, card in enumerate(hand):
            if card.rank() == fireworks[card.color()] and avail[h + i]:
                return h + i
        safe = [u for u in legal if not (h <= u < 2 * h)]        # never gamble a life
        if safe:
            return int(rng.choice(safe))
    return int(rng.choice(legal))


CASES = [
    # name, hanabi_name, players, use_obs_instead_of_state, seed, steps, skill
    ("full2", "Hanabi-Full", 2, False, 1, 260, 0.9),
    ("full2_random", "Hanabi-Full", 2, False, 7, 120, 0.0),
    ("full3_allobs", "Hanabi-Full", 3, True, 1001, 200, 0.95),
    ("full5", "Hanabi-Full", 5, False, 2001, 220, 0.97),
    ("full5_allobs", "Hanabi-Full", 5, True, 3, 120, 0.5),
    ("minimal4", "Hanabi-Full-Minimal", 4, False, 11, 200, 0.9),
    ("minimal2_allobs", "Hanabi-Full-Minimal", 2, True, 12, 150, 0.6),
    ("small2", "Hanabi-Small", 2, False, 5, 150, 0.95),
    ("small3_allobs", "Hanabi-Small", 3, True, 6, 150, 0.9),
    ("verysmall2", "Hanabi-Very-Small", 2, False, 1, 120, 0.95),
    ("verysmall4", "Hanabi-Very-Small", 4, True, 4, 100, 0.8),
    ("cardknowledge2", "Hanabi-Full-CardKnowledge", 2, False, 50000, 80, 0.7),
]


# rule sets the named games never use, driven through the reference's pyhanabi classes directly
RULE_CASES = [
    # name, config (HanabiGame parameters, hanabi_game.cc:31-49), steps, skill
    ("seer3", dict(colors=3, ranks=4, players=3, hand_size=3, max_information_tokens=5, max_life_tokens=2,
                   observation_type=2, seed=21), 160, 0.9),
    ("random_start4", dict(colors=4, ranks=5, players=4, max_information_tokens=6, max_life_tokens=3,
                           observation_type=1, random_start_player=1, seed=22), 200, 0.9),
    ("tall_hands2", dict(colors=5, ranks=5, players=2, hand_size=4, max_information_tokens=2, max_life_tokens=5,
                         observation_type=1, seed=23), 200, 0.8),
    ("two_ranks5", dict(colors=5, ranks=2, players=5, hand_size=2, max_information_tokens=4, max_life_tokens=1,
                        observation_type=0, seed=24), 150, 0.9),
    ("no_tokens3", dict(colors=2, ranks=3, players=3, hand_size=2, max_information_tokens=0, max_life_tokens=2,
                        observation_type=1, random_start_player=1, seed=25), 120, 0.7),
]


def play_rule_case(pyhanabi, config, steps, skill, rng):
    game = pyhanabi.HanabiGame(config)
    encoder = pyhanabi.ObservationEncoder(game, pyhanabi.ObservationEncoderType.CANONICAL)
    P, h = game.num_players(), game.hand_size()
    rec = dict(views=[], own=[], legal=[], to_move=[], actions=[], scores=[], status=[], resets=[])

    def new_state():
        state = game.new_initial_state()
        while state.cur_player() == pyhanabi.CHANCE_PLAYER_ID:
            state.deal_random_card()
        return state

    def snapshot(state):
        obs = [state.observation(p) for p in range(P)]
        rec["views"].append([encoder.encode(o) for o in obs])
        rec["own"].append([encoder.encodeownhand(o) for o in obs])
        legal = np.zeros(game.max_moves(), dtype=np.uint8)
        legal[[game.get_move_uid(m) for m in obs[state.cur_player()].legal_moves()]] = 1
        rec["legal"].append(legal)
        rec["to_move"].append(state.cur_player())
        return legal

    state = new_state()
    legal = snapshot(state)
    ends = set()
    for t in range(steps):
        moves = np.nonzero(legal)[0]
        a = None
        if rng.random() < skill:
            me, fireworks = state.cur_player(), state.fireworks()
            for i, card in enumerate(state.player_hands()[me]):
                if card.rank() == fireworks[card.color()]:
                    a = h + i
                    break
            if a is None:
                safe = [u for u in moves if not (h <= u < 2 * h)]
                a = int(rng.choice(safe)) if safe else None
        if a is None:
            a = int(rng.choice(moves))
        state.apply_move(game.get_move(a))
        while state.cur_player() == pyhanabi.CHANCE_PLAYER_ID:
            state.deal_random_card()
        rec["actions"].append(a)
        rec["scores"].append(state.score())
        rec["status"].append(state.end_of_game_status().value)
        legal = snapshot(state)
        if state.is_terminal():
            ends.add(state.end_of_game_status().name)
            rec["resets"].append(t)
            state = new_state()
            legal = snapshot(state)
    return rec, game, encoder, ends


def main():
    HanabiEnv = load_reference_env()
    out, names = {}, []
    from onpolicy.envs.hanabi import pyhanabi
    rule_names = []
    for name, config, steps, skill in RULE_CASES:
        rec, game, encoder, ends = play_rule_case(pyhanabi, config, steps, skill, np.random.default_rng(config["seed"]))
        key = "rules_" + name
        out[key + "_config"] = np.array([config.get(k, d) for k, d in (
            ("colors", 5), ("ranks", 5), ("players", 2), ("hand_size", 0), ("max_information_tokens", 8),
            ("max_life_tokens", 3), ("observation_type", 1), ("random_start_player", 0), ("seed", 0))])
        out[key + "_dims"] = np.array([game.max_moves(), encoder.shape()[0], encoder.ownhandshape()[0], game.hand_size()])
        for k in ("views", "own", "legal"):
            arr = np.asarray(rec[k])
            assert np.array_equal(arr, arr.astype(np.uint8))
            out["%s_%s" % (key, k)] = arr.astype(np.uint8)
        for k in ("to_move", "actions", "scores", "status", "resets"):
            out["%s_%s" % (key, k)] = np.asarray(rec[k], dtype=np.int32)
        rule_names.append(name)
        print("%-16s episodes %2d  best score %2d  first movers %s  endings %s" % (
            name, len(rec["resets"]), max(rec["scores"]), sorted(set(rec["to_move"][i + 1] for i in rec["resets"])),
            sorted(ends)))
    out["rule_cases"] = np.array(rule_names)
    for name, game, players, all_obs, seed, steps, skill in CASES:
        args = types.SimpleNamespace(hanabi_name=game, num_agents=players, use_obs_instead_of_state=all_obs)
        env = HanabiEnv(args, seed)
        rng = np.random.default_rng(seed + 99)
        obs, share, avail = env.reset()
        rec = dict(obs=[obs], share=[share], avail=[avail], actions=[], rewards=[], dones=[], scores=[], resets=[])
        episodes, best, ends = 0, 0, set()
        for t in range(steps):
            a = pick_action(env, avail, rng, skill)
            obs, share, rewards, done, info, avail = env.step([a])
            assert all(r == rewards[0] for r in rewards) and len(rewards) == players
            rec["actions"].append(a)
            rec["rewards"].append(rewards[0][0])
            rec["dones"].append(bool(done))
            rec["scores"].append(info["score"])
            rec["obs"].append(obs)
            rec["share"].append(share)
            rec["avail"].append(avail)
            best = max(best, info["score"])
            if done:                      # the runner resets a finished env before its next step
                ends.add(env.state.end_of_game_status().name)
                episodes += 1
                obs, share, avail = env.reset()
                rec["resets"].append(t)
                rec["obs"].append(obs)
                rec["share"].append(share)
                rec["avail"].append(avail)
        # the idle protocol: action -1 and reset(choose=False)
        idle = env.step([-1])
        assert idle[3] is None and not np.any(idle[0]) and not np.any(idle[1]) and not np.any(idle[5])
        out[name + "_idle_score"] = np.array(idle[4]["score"])
        skipped = env.reset(False)
        assert not np.any(skipped[0]) and not np.any(skipped[1]) and not np.any(skipped[2])
        out[name + "_meta"] = np.array([players, int(all_obs), seed, env.num_moves(),
                                        env.vectorized_observation_shape()[0],
                                        env.vectorized_share_observation_shape()[0]])
        for k in ("obs", "share", "avail"):
            arr = np.asarray(rec[k])
            assert np.array_equal(arr, arr.astype(np.uint8)), k
            out["%s_%s" % (name, k)] = arr.astype(np.uint8)
        out[name + "_actions"] = np.asarray(rec["actions"], dtype=np.int16)
        out[name + "_rewards"] = np.asarray(rec["rewards"], dtype=np.int16)
        out[name + "_dones"] = np.asarray(rec["dones"], dtype=np.uint8)
        out[name + "_scores"] = np.asarray(rec["scores"], dtype=np.int16)
        out[name + "_resets"] = np.asarray(rec["resets"], dtype=np.int32)
        names.append("%s|%s" % (name, game))
        print("%-16s %-26s episodes %2d  best score %2d  endings %s" % (name, game, episodes, best, sorted(ends)))
    out["cases"] = np.array(names)
    path = os.path.join(GOLD, "hanabi_cases.npz")
    np.savez_compressed(path, **out)
    print("hanabi_cases.npz: %d arrays, %d KiB" % (len(out), os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
