"""The built-in simple_spread (vectorised numpy, onpolicy/envs/mpe/simple_spread.py) against trajectories of the
reference's own particle environment (core.py physics + environment.py + scenarios/simple_spread.py, fixtures from
oracle/make_golden_mpe.py): same initial state + same actions => same observations, rewards, dones and positions."""
import numpy as np
import pytest

from onpolicy.envs.mpe.simple_spread import VecSimpleSpread


@pytest.mark.parametrize("case", [0, 1, 2])
def test_simple_spread_matches_reference_trajectories(gold, case):
    z = gold.npz("mpe_spread_cases")
    key = "mpe%d_" % case
    A, L, T = [int(x) for x in z[key + "dims"]]
    env = VecSimpleSpread(2, num_agents=A, num_landmarks=L, episode_length=T, seed=0, auto_reset=False)
    assert env.observation_space[0].shape == tuple(z[key + "obs_dim"])
    assert env.share_observation_space[0].shape == tuple(z[key + "share_obs_dim"])
    env.reset()
    # world 0 replays the reference trajectory, world 1 stays random (worlds must not interact)
    env.pos[0], env.vel[0], env.landmarks[0] = z[key + "pos0"], z[key + "vel0"], z[key + "landmarks"]
    np.testing.assert_allclose(env._obs()[0], z[key + "obs0"], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(99)
    for t in range(T):
        act = np.stack([z[key + "actions"][t], np.eye(5)[rng.integers(0, 5, A)]])
        obs, rew, done, info = env.step(act)
        np.testing.assert_allclose(env.pos[0], z[key + "pos"][t], rtol=1e-9, atol=1e-9, err_msg="t=%d" % t)
        np.testing.assert_allclose(obs[0], z[key + "obs"][t], rtol=2e-6, atol=2e-6, err_msg="t=%d" % t)
        np.testing.assert_allclose(rew[0], z[key + "rewards"][t].reshape(A, 1), rtol=1e-6, atol=1e-6, err_msg="t=%d" % t)
        np.testing.assert_array_equal(done[0], z[key + "dones"][t])
