"""CPU checks of the pieces that are not on the benchmarked configuration but belong to the same
API surface: PopArt, the non-Discrete action heads, ValueNorm numpy/tensor round trips."""
import numpy as np
import pytest
import torch

from helpers import Box, Discrete, make_args

from onpolicy.algorithms.utils.popart import PopArt
from onpolicy.algorithms.utils.act import ACTLayer
from onpolicy.utils.valuenorm import ValueNorm


def test_popart_update_preserves_unnormalised_output():
    """PopArt's defining property: after a statistics update the de-normalised output of the layer is
    unchanged (the reference's update raises TypeError on CPU, popart.py:64,69-70; this one works)."""
    torch.manual_seed(0)
    layer = PopArt(16, 1)
    x = torch.randn(32, 16)
    layer.update(torch.randn(64, 1) * 3 + 2)          # warm statistics
    before = layer.denormalize(layer(x)).detach().clone()
    layer.update(torch.randn(64, 1) * 5 - 1)
    after = layer.denormalize(layer(x)).detach()
    torch.testing.assert_close(after, before, rtol=1e-4, atol=1e-4)
    s = layer.denorm_scalars()
    mean, var = layer.debiased_mean_var()
    assert float(s[0]) == pytest.approx(float(var.sqrt())) and float(s[1]) == pytest.approx(float(mean))
    # numpy in -> numpy out, tensor in -> tensor out
    assert isinstance(layer.denormalize(np.zeros((3, 1), np.float32)), np.ndarray)
    assert torch.is_tensor(layer.denormalize(torch.zeros(3, 1)))


def test_popart_trainer_runs():
    from oracle import oracle
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from helpers import fill_buffer_arrays, buffer_shapes, load_into
    T, N, A, Do, Ds, na = 6, 3, 2, 5, 9, 4
    args = make_args(episode_length=T, n_rollout_threads=N, hidden_size=16, ppo_epoch=2, num_mini_batch=2,
                     use_popart=True, use_valuenorm=False)
    spaces = Box((Do,)), Box((Ds,)), Discrete(na)
    torch.manual_seed(1)
    policy = R_MAPPOPolicy(args, *spaces)
    trainer = R_MAPPO(args, policy)
    assert trainer.value_normalizer is policy.critic.v_out
    buf = oracle.OracleBuffer(args, A, *spaces)
    arrays = fill_buffer_arrays(buffer_shapes(T, N, A, Do, Ds, na, 16), np.random.default_rng(0), na=na)
    load_into(buf, arrays)
    buf.compute_returns(arrays["next_value"], trainer.value_normalizer)
    info = trainer.train(buf)
    assert all(np.isfinite(v) for v in info.values())
    assert float(policy.critic.v_out.debiasing_term) > 0


def test_valuenorm_roundtrip_and_types():
    vn = ValueNorm(1)
    x = torch.randn(100, 1) * 4 + 3
    vn.update(x)
    torch.testing.assert_close(vn.denormalize(vn.normalize(x)), x, rtol=1e-5, atol=1e-5)
    out = vn.denormalize(x.numpy())
    assert isinstance(out, np.ndarray)
    np.testing.assert_allclose(out, vn.denormalize(x).numpy(), rtol=1e-6)
    assert set(vn.state_dict()) == {"running_mean", "running_mean_sq", "debiasing_term"}


class _MultiDiscrete(object):
    def __init__(self, nvec):
        self.high = np.array(nvec) - 1
        self.low = np.zeros(len(nvec), dtype=np.int64)
        self.shape = len(nvec)


_MultiDiscrete.__name__ = "MultiDiscrete"


class _MultiBinary(object):
    def __init__(self, n):
        self.shape = (n,)


_MultiBinary.__name__ = "MultiBinary"


@pytest.mark.parametrize("space,act_dim", [(Discrete(5), 1), (Box((3,)), 3), (_MultiDiscrete([3, 4]), 2),
                                           (_MultiBinary(4), 4)])
def test_action_heads_shapes_and_consistency(space, act_dim):
    torch.manual_seed(0)
    layer = ACTLayer(space, 16, True, 0.01)
    x = torch.randn(10, 16)
    actions, logp = layer(x)
    assert actions.shape == (10, act_dim)
    det, _ = layer(x, deterministic=True)
    assert det.shape == actions.shape
    ev_logp, ent = layer.evaluate_actions(x, actions.float(), active_masks=torch.ones(10, 1))
    assert torch.isfinite(ent)
    if space.__class__.__name__ in ("Discrete", "Box", "MultiBinary"):
        torch.testing.assert_close(ev_logp, logp, rtol=1e-5, atol=1e-6)      # log-prob of the drawn action
    probs = layer.get_probs(x) if space.__class__.__name__ != "Box" else None
    if space.__class__.__name__ == "Discrete":
        torch.testing.assert_close(probs.sum(-1), torch.ones(10), rtol=1e-5, atol=1e-6)


def test_gemm_tuning_is_a_no_op_without_a_gpu(monkeypatch):
    """onpolicy.utils.gemm_tuning.enable() must not touch anything on a CPU-only host, honours its off switch,
    and the shipped winners file is a TunableOp CSV for gfx950."""
    import torch
    from onpolicy.utils import gemm_tuning
    monkeypatch.setenv("MAPPO_GEMM_TUNING", "1")
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() is False
        assert tuple(gemm_tuning.results()) == ()
    monkeypatch.setenv("MAPPO_GEMM_TUNING", "0")
    assert gemm_tuning.enable() is False
    lines = open(gemm_tuning.SHIPPED).read().splitlines()
    assert lines[0].startswith("Validator,PT_VERSION") and any("gfx950" in l for l in lines[:6])
    assert sum(l.startswith("Gemm") for l in lines) > 50


def test_multi_discrete_space_drives_the_action_head():
    """onpolicy.utils.multi_discrete.MultiDiscrete (reference utils/multi_discrete.py) is recognised by name by the
    action head and the shape helpers."""
    from onpolicy.utils.multi_discrete import MultiDiscrete
    from onpolicy.utils.util import get_shape_from_act_space
    space = MultiDiscrete([[0, 4], [0, 1], [0, 2]])
    assert space.shape == 3 and space.n == 9 and space.contains([4, 1, 2]) and not space.contains([5, 0, 0])
    np.random.seed(0)
    assert all(space.contains(space.sample()) for _ in range(50))
    assert get_shape_from_act_space(space) == 3
    layer = ACTLayer(space, 8, True, 0.01)
    actions, logp = layer(torch.randn(6, 8))
    assert actions.shape == (6, 3) and logp.shape == (6, 3)
    assert bool((actions[:, 0] <= 4).all()) and bool((actions[:, 1] <= 1).all()) and bool((actions[:, 2] <= 2).all())


def test_stream_capture_keeps_the_cyclic_collector_out(monkeypatch):
    """onpolicy/utils/graph_capture.py: dead cycles (which may hold CUDAGraph objects of an earlier trainer / runner) are collected
    BEFORE a capture begins and the collector stays off until it has ended -- a graph destroyed in the middle of another capture
    aborts the process (round 5, device suite).  CPU check with a stand-in for torch.cuda.graph."""
    import contextlib
    import gc
    import torch
    from onpolicy.utils import graph_capture

    class Node(object):
        died = []

        def __init__(self):
            self.me = self          # a reference cycle: only the collector frees it

        def __del__(self):
            Node.died.append(inside[0])

    inside = [False]

    @contextlib.contextmanager
    def fake_graph(graph, **kw):
        inside[0] = True
        try:
            yield
        finally:
            inside[0] = False

    monkeypatch.setattr(torch.cuda, "graph", fake_graph)
    gc.collect()
    Node()                      # dead before the capture begins
    assert gc.isenabled()
    with graph_capture.capturing(object(), pool=None):
        assert not gc.isenabled()
        for _ in range(3):
            Node()              # dies during the capture: must wait
        junk = [[i] for i in range(20000)]     # enough allocations to trigger an automatic collection if it were enabled
        del junk
    assert gc.isenabled()
    gc.collect()
    assert len(Node.died) == 4 and not any(Node.died), Node.died
    gc.disable()
    try:                        # a caller that runs with the collector off gets it back off
        with graph_capture.capturing(object()):
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
