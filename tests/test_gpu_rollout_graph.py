"""-m gpu: the device-resident rollout step as one captured HIP graph launch (onpolicy/runner/shared/rollout_graph.py)
against the eager loop it replaces (reference onpolicy/runner/shared/mpe_runner.py:96-139: collect -> envs.step -> insert):
from the same worlds, generator states and policy a whole rollout must fill the buffer with the same trajectory."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ("share_obs", "obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "rnn_states",
          "rnn_states_critic")


def _runner(tmp_path, monkeypatch, algo, N=64, T=12, episodes=2, graph="1", extra=()):
    from onpolicy.scripts.train import train_mpe
    monkeypatch.setenv("MAPPO_RESULTS_DIR", str(tmp_path / "results"))
    monkeypatch.setenv("MAPPO_ROLLOUT_GRAPH", graph)
    return train_mpe.main(["--env_name", "MPE", "--scenario_name", "simple_spread", "--num_agents", "3",
                           "--num_landmarks", "3", "--algorithm_name", algo, "--n_rollout_threads", str(N),
                           "--episode_length", str(T), "--num_env_steps", str(episodes * N * T), "--ppo_epoch", "2",
                           "--num_mini_batch", "1", "--data_chunk_length", "4", "--hidden_size", "64", "--use_ReLU",
                           "--use_wandb", "--log_interval", "1", "--n_training_threads", "1", "--use_device_env"]
                          + list(extra))


def _snapshot(runner):
    e = runner.envs
    dev = runner.buffer.device
    return ({k: getattr(e, k).clone() for k in ("pos", "vel", "landmarks", "t")}, e.rng.get_state(),
            torch.cuda.get_rng_state(dev), runner.buffer.step)


def _restore(runner, snap):
    e = runner.envs
    state, env_rng, dev_rng, step = snap
    for k, v in state.items():
        getattr(e, k).copy_(v)
    e.rng.set_state(env_rng)
    torch.cuda.set_rng_state(dev_rng, runner.buffer.device)
    runner.buffer.step = step


def _fields(runner):
    out = {}
    for name in FIELDS:
        t = getattr(runner.buffer, name)
        if t.stride()[0] != 0:
            out[name] = t.clone()
    return out


# --use_popart (with --use_valuenorm = store_false: the trainer asserts they are exclusive): the PopArt value head REBINDS its
# parameters' storage on every update (algorithms/utils/popart.py), which a captured graph must not be blind to
POPART = ("--use_popart", "--use_valuenorm")


@pytest.mark.parametrize("algo,extra", [("mappo", ()), ("rmappo", ()), ("mappo", POPART), ("rmappo", POPART)],
                         ids=["mappo", "rmappo", "mappo_popart", "rmappo_popart"])
def test_graphed_rollout_equals_eager_rollout(tmp_path, monkeypatch, algo, extra):
    runner = _runner(tmp_path, monkeypatch, algo, extra=extra)
    if extra:       # two train() calls have run: the head the graph was captured with is long gone
        from onpolicy.algorithms.utils.popart import PopArt
        assert isinstance(runner.trainer.policy.critic.v_out, PopArt)
    rg = runner.rollout_graph
    assert rg is not None and rg.graph is not None, "the rollout step was not captured"
    assert rg.replays == 2 * runner.episode_length          # both training episodes ran through the graph
    T = runner.episode_length
    snap = _snapshot(runner)
    row0 = {k: getattr(runner.buffer, k)[0].clone() for k in ("obs", "share_obs", "masks")}

    # eager loop
    for step in range(T):
        values, actions, logp, rnn_a, rnn_c, actions_env = runner.collect(step)
        obs, rewards, dones, infos = runner.envs.step(actions_env)
        runner.insert((obs, rewards, dones, infos, values, actions, logp, rnn_a, rnn_c))
    torch.cuda.synchronize()
    eager = _fields(runner)
    eager_infos = [[d["individual_reward"] for d in row] for row in infos]

    # the same rollout through the graph
    _restore(runner, snap)
    for k, v in row0.items():
        getattr(runner.buffer, k)[0].copy_(v)
    runner.trainer.prep_rollout()
    rg.begin_episode()
    for step in range(T):
        g_infos = rg.step()
    torch.cuda.synchronize()
    graphed = _fields(runner)
    assert float(graphed["masks"].min()) == 0.0 or T < 25        # (episodes of 25 steps: ends only show in long rollouts)
    np.testing.assert_array_equal(graphed["actions"].cpu().numpy(), eager["actions"].cpu().numpy())
    for name in eager:
        torch.testing.assert_close(graphed[name], eager[name], rtol=1e-5, atol=1e-6, msg=name)
    np.testing.assert_allclose([[d["individual_reward"] for d in row] for row in g_infos], eager_infos, rtol=1e-6)


@pytest.mark.parametrize("extra", [(), POPART], ids=["valuenorm", "popart"])
def test_training_through_the_graph_logs_like_the_eager_run(tmp_path, monkeypatch, extra):
    """Whole runs (rollouts + updates) with and without the graph from the same seed: same logged rewards / losses."""
    logs = {}
    for mode in ("1", "0"):
        runner = _runner(tmp_path / mode, monkeypatch, "mappo", N=32, T=10, episodes=3, graph=mode, extra=extra)
        assert (runner.rollout_graph is not None) == (mode == "1")
        recs = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
        logs[mode] = {(r["tag"], i): r[r["tag"]] for i, r in enumerate(recs)}
    assert logs["1"].keys() == logs["0"].keys() and len(logs["1"]) > 10
    for k in logs["1"]:
        assert logs["1"][k] == pytest.approx(logs["0"][k], rel=2e-3, abs=1e-5), k
