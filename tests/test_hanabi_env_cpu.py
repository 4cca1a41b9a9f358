"""The batched Hanabi stepper (csrc/hanabi_batch.cc, include/hanabi_batch.h) against the reference's env.

tests/golden/hanabi_cases.npz holds games played on the reference's HanabiEnv (its Python glue over its C++ engine,
oracle/make_golden_hanabi.py): replaying the recorded actions here must give identical observations, centralised
observations, legal-action masks, rewards, done flags and scores -- bit for bit, over several episodes per case, since
every deal comes from the same per-env generator that keeps running across episodes."""
import os
import re
import subprocess
import types

import numpy as np
import pytest

from conftest import ROOT
from onpolicy.envs.hanabi import batch as hb
from onpolicy.envs.hanabi.Hanabi_Env import HanabiEnv

HEADER = os.path.join(ROOT, "include", "hanabi_batch.h")


def _cases(gold):
    z = gold.npz("hanabi_cases")
    return z, [c.split("|") for c in z["cases"]]


@pytest.fixture
def cpu_launch(monkeypatch, tmp_path):
    """Lets the train / eval scripts run here: host buffer stand-in, CPU device (threads set as the real device_of
    does), results under tmp_path."""
    import torch
    import onpolicy.runner.shared.base_runner as base
    from host_buffer import HostSharedBuffer
    from onpolicy.scripts.train import _launch

    def device_of(all_args):
        torch.set_num_threads(all_args.n_training_threads)
        return torch.device("cpu")
    threads = torch.get_num_threads()
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    monkeypatch.setattr(_launch, "device_of", device_of)
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    yield
    torch.set_num_threads(threads)


def _args(game, players, all_obs):
    return types.SimpleNamespace(hanabi_name=game, num_agents=int(players), use_obs_instead_of_state=bool(all_obs))


def test_env_replays_reference_games_exactly(gold):
    z, cases = _cases(gold)
    assert len(cases) >= 12
    endings = set()
    for name, game in cases:
        players, all_obs, seed, n_moves, obs_len, share_len = (int(v) for v in z[name + "_meta"])
        env = HanabiEnv(_args(game, players, all_obs), seed)
        assert env.num_moves() == n_moves == env.action_space[0].n
        assert env.vectorized_observation_shape() == [obs_len]
        assert env.vectorized_share_observation_shape() == [share_len]
        assert env.observation_space == [[obs_len + players]] * players
        assert env.share_observation_space == [[share_len + players]] * players
        exp_obs, exp_share, exp_avail = z[name + "_obs"], z[name + "_share"], z[name + "_avail"]
        resets = set(int(t) for t in z[name + "_resets"])
        row = 0

        def check(got, what):
            for g, e, k in zip(got, (exp_obs[row], exp_share[row], exp_avail[row]), ("obs", "share_obs", "available")):
                assert g.dtype == np.float32 and np.array_equal(g, e), "%s: %s differs at %s" % (name, k, what)

        check(env.reset(), "the first reset")
        for t, a in enumerate(z[name + "_actions"]):
            obs, share, rewards, done, info, avail = env.step([int(a)])
            row += 1
            check((obs, share, avail), "step %d" % t)
            assert rewards == [[float(z[name + "_rewards"][t])]] * players
            assert done is bool(z[name + "_dones"][t]) and info == {"score": int(z[name + "_scores"][t])}
            if done:
                endings.add(env.state()["end_of_game"])
                assert t in resets
                row += 1
                check(env.reset(), "the reset after step %d" % t)
        assert row + 1 == len(exp_obs)
        # idle protocol (Hanabi_Env.py:460-468, :307-311)
        obs, share, rewards, done, info, avail = env.step([-1])
        assert done is None and not obs.any() and not share.any() and not avail.any()
        assert np.array_equal(rewards, np.zeros((players, 1))) and info == {"score": int(z[name + "_idle_score"])}
        assert not any(x.any() for x in env.reset(False))
        env.close()
    assert endings == {1, 2, 3}          # out of life tokens, out of cards, completed fireworks


def test_rule_sets_outside_the_named_games_match_the_reference_engine(gold):
    """Seer observations, random start player, non-default hand sizes / token counts / rank counts: every player's
    encoded view, own-hand vector, the legal moves, the player to move, score and end-of-game status after each move,
    against the reference's HanabiGame / HanabiState / ObservationEncoder played with the same seed and actions."""
    z = gold.npz("hanabi_cases")
    keys = ("colors", "ranks", "players", "hand_size", "max_information_tokens", "max_life_tokens",
            "observation_type", "random_start_player")
    first_movers = set()
    for name in z["rule_cases"]:
        k = "rules_%s_" % name
        config = [int(v) for v in z[k + "config"]]
        rules, seed = dict(zip(keys, config[:8])), config[8]
        b = hb.HanabiBatch(rules, [seed])
        n_moves, obs_len, own_len, hand = (int(v) for v in z[k + "dims"])
        assert (b.num_moves, b.obs_len, b.own_hand_len) == (n_moves, obs_len, own_len)
        views, own, legal, to_move = z[k + "views"], z[k + "own"], z[k + "legal"], z[k + "to_move"]
        resets = set(int(t) for t in z[k + "resets"])
        row = 0

        def check(what):
            b.encode()
            assert int(b.to_move[0]) == int(to_move[row]), "%s: player to move at %s" % (name, what)
            assert np.array_equal(b.available_actions[0], legal[row]), "%s: legal moves at %s" % (name, what)
            for p in range(b.players):
                got_view, got_own = b.player_view(0, p)
                assert np.array_equal(got_view, views[row, p]), "%s: view of player %d at %s" % (name, p, what)
                assert np.array_equal(got_own, own[row, p]), "%s: own hand of player %d at %s" % (name, p, what)
            assert np.array_equal(b.obs[0, :obs_len], views[row, to_move[row]])

        b.reset()
        check("the first deal")
        first_movers.add((str(name), int(b.to_move[0])))
        for t, a in enumerate(z[k + "actions"]):
            b.step([int(a)])
            row += 1
            check("move %d" % t)
            state = b.table_state(0)
            assert state["score"] == int(z[k + "scores"][t]) == int(b.scores[0])
            assert state["end_of_game"] == int(z[k + "status"][t]) and bool(b.status[0]) == (state["end_of_game"] != 0)
            if t in resets:
                assert b.status[0] == 1
                b.reset()
                row += 1
                check("the deal after move %d" % t)
                first_movers.add((str(name), int(b.to_move[0])))
        assert row + 1 == len(views)
    assert len({m for n, m in first_movers if n == "random_start4"}) >= 3      # the start player really is drawn


def test_batched_vec_env_equals_one_env_per_thread(gold):
    """HanabiBatchVecEnv (one native call for all threads) == the per-env protocol of ChooseDummyVecEnv over
    HanabiEnv objects, including threads that sit a step out (action -1) and selective resets."""
    from onpolicy.envs.env_wrappers import ChooseDummyVecEnv
    n, players = 7, 3
    args = _args("Hanabi-Small", players, False)
    seeds = [5 + 1000 * i for i in range(n)]
    vec = hb.HanabiBatchVecEnv(args, seeds)
    ref = ChooseDummyVecEnv([(lambda s=s: HanabiEnv(args, s)) for s in seeds])
    assert vec.num_envs == n and vec.action_space[0].n == ref.action_space[0].n
    assert vec.observation_space == ref.observation_space
    assert vec.share_observation_space == ref.share_observation_space
    rng = np.random.default_rng(0)
    choose = np.ones(n, dtype=bool)
    got, exp = vec.reset(choose), ref.reset(choose)
    finished = 0
    for t in range(120):
        for g, e in zip(got[:2] + got[-1:], exp[:2] + exp[-1:]):
            assert np.array_equal(g, np.asarray(e, dtype=np.float32))
        avail = got[-1]
        actions = np.full((n, 1), -1, dtype=np.int64)
        for i in range(n):
            if avail[i].any() and rng.random() < 0.8:
                actions[i, 0] = rng.choice(np.nonzero(avail[i])[0])
        got, exp = vec.step(actions), ref.step(actions)
        assert np.array_equal(got[2], np.asarray(exp[2], dtype=np.float32)) and got[2].shape == (n, players, 1)
        assert list(got[3]) == list(exp[3]) and list(got[4]) == list(exp[4])
        choose = np.array([d is True or not a.any() for d, a in zip(got[3], got[5])])
        finished += sum(d is True for d in got[3])
        if choose.any():         # finished tables and those that sat out with nothing to do start a new game
            new_got, new_exp = vec.reset(choose), ref.reset(choose)
            got = tuple(np.where(choose[:, None], ng, g) for ng, g in zip(new_got, (got[0], got[1], got[5])))
            exp = tuple(np.where(choose[:, None], np.asarray(ne), np.asarray(e))
                        for ne, e in zip(new_exp, (exp[0], exp[1], exp[5])))
        else:
            got, exp = (got[0], got[1], got[5]), (exp[0], exp[1], exp[5])
    assert finished >= 5
    vec.close()
    ref.close()


def test_illegal_moves_and_bad_rules_are_rejected():
    args = _args("Hanabi-Very-Small", 2, False)
    env = HanabiEnv(args, 3)
    with pytest.raises(ValueError, match="never reset"):
        env.step([0])
    _, _, avail = env.reset()
    before = env.state()
    illegal = int(np.nonzero(avail == 0)[0][0])          # all tokens in hand: discarding is illegal
    with pytest.raises(ValueError, match="illegal move on table 0"):
        env.step([illegal])
    with pytest.raises(ValueError, match="illegal move"):
        env.step([env.num_moves()])
    assert env.state() == before                           # nothing was applied
    with pytest.raises(ValueError, match="Unknown environment"):
        HanabiEnv(_args("Hanabi-Huge", 2, False), 1)
    for bad in (dict(players=6), dict(players=1), dict(colors=6), dict(hand_size=6), dict(max_life_tokens=0)):
        with pytest.raises(ValueError, match="invalid Hanabi rules"):
            hb.HanabiBatch(dict(hb.rules_for("Hanabi-Full", 2), **bad), [1])


def test_seer_and_random_start_rules():
    """The two rule switches the named games never set: seer observations start with every card hinted, and a
    random start player is drawn from the table's generator before the first deal (hanabi_game.cc:150-157)."""
    rules = dict(hb.rules_for("Hanabi-Small", 3), observation_type=hb.OBSERVATION_SEER, random_start_player=1)
    b = hb.HanabiBatch(rules, np.arange(40))
    b.reset()
    b.encode()
    assert set(b.to_move.tolist()) == {0, 1, 2}
    K, C, R, H, P = 10, 2, 5, 2, 3
    belief = b.player_view(0, 0)[0][-(P * H * (K + C + R)):].reshape(P, H, K + C + R)
    assert (belief[..., :K].sum(-1) == 1).all() and (belief[..., K:K + C].sum(-1) == 1).all()
    assert (belief[..., K + C:].sum(-1) == 1).all()
    own = b.player_view(0, 0)[1].reshape(H, K)
    assert np.array_equal(own, belief[0, :, :K])          # the observer's own slots name its true cards


def test_header_symbols_exported_bound_and_plain_c():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(hanabi_batch_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 13 and sorted(hb.SIGNATURES) == names
    lib = hb.lib()
    for n in names:
        assert hasattr(lib, n), "libhanabi_batch.so does not export %s" % n
    import ctypes
    assert ctypes.sizeof(hb.Rules) == 32
    for lang, std in (("c", "c99"), ("c++", "c++17")):
        out = subprocess.run(["gcc", "-fsyntax-only", "-x", lang, "-std=" + std, "-Wall", "-Werror", HEADER],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(hb, "_lib", None)
    monkeypatch.setattr(hb, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="libhanabi_batch.so not found"):
        hb.lib()


@pytest.mark.parametrize("game,players", [("Hanabi-Very-Small", 2), ("Hanabi-Small", 3)])
def test_train_script_plays_real_games(cpu_launch, game, players):
    """train_hanabi_forward end to end on the real engine (host buffer stand-in, CPU): the batched stepper and the
    reference layout (one HanabiEnv per thread behind the Choose* wrappers) play the same games, so the two runs log
    the same scores and end with the same parameters."""
    import json
    import torch
    from onpolicy.scripts.train import train_hanabi_forward
    argv = ["--env_name", "Hanabi", "--hanabi_name", game, "--num_agents", str(players), "--algorithm_name", "mappo",
            "--n_rollout_threads", "3", "--episode_length", "8", "--num_env_steps", "96", "--ppo_epoch", "2",
            "--hidden_size", "16", "--use_wandb", "--log_interval", "1", "--n_training_threads", "1", "--use_eval",
            "--n_eval_rollout_threads", "2", "--eval_interval", "2", "--seed", "3"]
    runs = []
    for extra in ([], ["--use_subproc_envs"]):
        runner = train_hanabi_forward.main(argv + extra)
        assert type(runner.envs).__name__ == ("ChooseSubprocVecEnv" if extra else "HanabiBatchVecEnv")
        assert runner.true_total_num_steps > 0
        logged = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
        tags = {r["tag"] for r in logged}
        assert {"value_loss", "average_score", "eval_average_score"} <= tags
        params = torch.cat([p.detach().reshape(-1) for p in runner.policy.actor.parameters()])
        runs.append(([sorted(r.items()) for r in logged if "score" in r["tag"]], params,
                     runner.true_total_num_steps))
        runner.envs.close()
        runner.eval_envs.close()
    assert len(runs[0][0]) >= 4 and runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2]
    assert torch.equal(runs[0][1], runs[1][1])


def test_eval_script_scores_a_saved_policy(cpu_launch):
    """scripts/eval/eval_hanabi.py: restore the checkpoint a training run saved and average the score of
    deterministic games (reference scripts/eval/eval_hanabi.py + HanabiRunner.eval_100k)."""
    from onpolicy.scripts.eval import eval_hanabi
    from onpolicy.scripts.train import train_hanabi_forward
    common = ["--env_name", "Hanabi", "--hanabi_name", "Hanabi-Very-Small", "--num_agents", "2", "--algorithm_name",
              "mappo", "--n_rollout_threads", "2", "--hidden_size", "16", "--use_wandb", "--n_training_threads", "1"]
    trained = train_hanabi_forward.main(common + ["--episode_length", "8", "--num_env_steps", "32", "--ppo_epoch", "1"])
    model_dir = str(trained.save_dir)
    assert os.path.exists(os.path.join(model_dir, "actor.pt"))
    with pytest.raises(AssertionError, match="use_eval"):
        eval_hanabi.main(common + ["--model_dir", model_dir])
    with pytest.raises(AssertionError, match="model_dir"):
        eval_hanabi.main(common + ["--use_eval"])
    argv = common + ["--use_eval", "--model_dir", model_dir, "--n_eval_rollout_threads", "4", "--eval_games", "12"]
    score = eval_hanabi.main(argv)
    assert 0.0 <= score <= 5.0
    assert eval_hanabi.main(argv) == score              # deterministic policy, seeded tables


def test_policy_learns_to_score_on_the_real_engine(cpu_launch):
    """Learning signal through the whole turn-based path (batched engine -> runner bookkeeping -> GAE -> PPO): on
    Hanabi-Very-Small the average score of finished games climbs from ~0 (random play loses the single life) to ~1
    within a few dozen updates."""
    import json
    from onpolicy.scripts.train import train_hanabi_forward
    runner = train_hanabi_forward.main(
        ["--env_name", "Hanabi", "--hanabi_name", "Hanabi-Very-Small", "--num_agents", "2", "--algorithm_name", "mappo",
         "--n_rollout_threads", "64", "--episode_length", "40", "--num_env_steps", str(64 * 40 * 36), "--ppo_epoch", "5",
         "--num_mini_batch", "1", "--hidden_size", "64", "--layer_N", "1", "--lr", "1e-3", "--critic_lr", "1e-3",
         "--entropy_coef", "0.015", "--use_wandb", "--log_interval", "1", "--n_training_threads", "1"])
    scores = [json.loads(l)["average_score"] for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))
              if json.loads(l)["tag"] == "average_score"]
    assert len(scores) >= 30
    assert np.mean(scores[:3]) < 0.3 and np.mean(scores[-5:]) > 0.8, (scores[:3], scores[-5:])


def test_engine_invariants_under_random_rules_and_play():
    """Property test over random rule sets and uniformly random legal play: cards are conserved (deck + hands +
    discards + fireworks = the full deck), scores and tokens stay in range, a running game always offers a legal
    move, finished games report a consistent ending, and identical seeds replay identically."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=30, deadline=None, derandomize=True)
    @given(seed=st.integers(0, 2 ** 31 - 1), colors=st.integers(1, 5), ranks=st.integers(2, 5),
           players=st.integers(2, 5), hand=st.integers(1, 5), info=st.integers(0, 8), lives=st.integers(1, 4),
           obs_type=st.integers(0, 2))
    def run(seed, colors, ranks, players, hand, info, lives, obs_type):
        copies = sum(3 if r == 0 else (1 if r == ranks - 1 else 2) for r in range(ranks))
        deck = copies * colors
        rules = dict(colors=colors, ranks=ranks, players=players, hand_size=hand, max_information_tokens=info,
                     max_life_tokens=lives, observation_type=obs_type, random_start_player=seed % 2)
        if hand * players > deck:
            with pytest.raises(ValueError, match="invalid Hanabi rules"):
                hb.HanabiBatch(rules, [seed])
            return
        n = 3
        a, b = hb.HanabiBatch(rules, [seed, seed + 1, seed + 2]), hb.HanabiBatch(rules, [seed, seed + 1, seed + 2])
        rng = np.random.default_rng(seed)
        for batch in (a, b):
            batch.reset()
            batch.encode()
        for _ in range(60):
            assert np.array_equal(a.obs, b.obs) and np.array_equal(a.available_actions, b.available_actions)
            assert a.available_actions.any(axis=1).all()             # somebody can always move in a running game
            moves = np.array([rng.choice(np.flatnonzero(row)) for row in a.available_actions], dtype=np.int32)
            for batch in (a, b):
                batch.step(moves)
            assert np.array_equal(a.rewards, b.rewards) and np.array_equal(a.status, b.status)
            for i in range(n):
                s = a.table_state(i)
                in_hands = sum(int(a.player_view(i, p)[1].sum()) for p in range(players))
                assert s["deck_size"] + in_hands + s["discards"] + sum(s["fireworks"]) == deck
                assert 0 <= s["information_tokens"] <= info and 0 <= s["life_tokens"] <= lives
                assert 0 <= s["score"] <= colors * ranks and all(0 <= f <= ranks for f in s["fireworks"])
                assert (s["end_of_game"] != 0) == bool(a.status[i])
                if s["end_of_game"] == 1:
                    assert s["life_tokens"] == 0 and s["score"] == 0
                if s["end_of_game"] == 3:
                    assert s["score"] == colors * ranks
            done = a.status == 1
            for batch in (a, b):
                if done.any():
                    batch.reset(done)
                batch.encode()
    run()
    # seeds wider than 32 bits wrap like the generator's own seeding does
    wide, wrapped = hb.HanabiBatch(hb.rules_for("Hanabi-Small", 2), [(1 << 32) + 7]), \
        hb.HanabiBatch(hb.rules_for("Hanabi-Small", 2), [7])
    for batch in (wide, wrapped):
        batch.reset()
        batch.encode()
    assert np.array_equal(wide.share_obs, wrapped.share_obs)
