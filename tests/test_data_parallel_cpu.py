"""The N > 1 path on CPU: world_size-2 gloo processes, rollout threads sharded over ranks, one flat
gradient all-reduce + global statistics per update.  A data-parallel update over two shards must
equal the single-process update over the whole buffer (num_mini_batch=1 => the union of the ranks'
minibatches is the full batch) up to float32 summation order, and the replicas must stay identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import dp_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_threads_partition():
    from onpolicy.utils.dist import shard_threads
    for n in (1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard_threads(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("args_over", [
    dict(),                                                      # masked means, ValueNorm, huber
    dict(use_policy_active_masks=False, use_value_active_masks=False, use_valuenorm=False),
    dict(use_recurrent_policy=True, data_chunk_length=3, algorithm_name="rmappo"),
], ids=["default", "unmasked_nonorm", "recurrent"])
def test_two_rank_update_equals_single_process(tmp_path, args_over):
    N = 6
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dp_worker.worker, args=(r, world, port, N, args_over, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert [r["span"] for r in ranks] == [(0, 3), (3, 6)]
    # replicas stay bit-identical: same all-reduced gradients, same Adam step
    for k in ranks[0]["sd"]:
        assert torch.equal(ranks[0]["sd"][k], ranks[1]["sd"][k]), k
    # and match the single-process update on the whole buffer
    info, sd, ws = dp_worker.run_update(N, 0, N, args_over)
    assert ws == 1
    recurrent = args_over.get("use_recurrent_policy", False)
    for k in sd:
        # the recurrent sampler drops a different tail of chunks per shard (6*3*6/3 chunks split
        # unevenly), so only the feed-forward cases are exactly the same minibatch
        if not recurrent:
            np.testing.assert_allclose(ranks[0]["sd"][k].numpy(), sd[k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    for k in ("actor_grad_norm", "critic_grad_norm"):
        assert ranks[0]["info"][k] == pytest.approx(ranks[1]["info"][k], rel=1e-6)
        if not recurrent:
            assert ranks[0]["info"][k] == pytest.approx(info[k], rel=1e-3, abs=1e-6)


@pytest.mark.parametrize("args_over", [
    dict(),
    dict(use_policy_active_masks=False, use_value_active_masks=False, use_valuenorm=False),
], ids=["default", "unmasked_nonorm"])
def test_two_rank_transformer_update_equals_single_process(tmp_path, args_over):
    """MATTrainer under data parallelism: globally normalised advantages (three all-reduced sums), loss terms weighted
    to global denominators, ValueNorm fed the global moments, one flat-bucket gradient all-reduce per minibatch."""
    N, world = 6, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dp_worker.worker, args=(r, world, port, N, args_over, str(tmp_path), None, True))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    for k in ranks[0]["sd"]:
        assert torch.equal(ranks[0]["sd"][k], ranks[1]["sd"][k]), k              # replicas stay identical
    info, sd, ws = dp_worker.run_update_mat(N, 0, N, args_over)
    assert ws == 1
    for k in sd:
        np.testing.assert_allclose(ranks[0]["sd"][k].numpy(), sd[k].numpy(), rtol=5e-4, atol=5e-6, err_msg=k)
    # logged losses are global-batch values too: rank means weighted by local / global denominators, summed
    for k, rel in (("actor_grad_norm", 2e-3), ("value_loss", 2e-3), ("policy_loss", 2e-3), ("dist_entropy", 2e-3)):
        assert ranks[0]["info"][k] == pytest.approx(ranks[1]["info"][k], rel=1e-6)
        assert ranks[0]["info"][k] == pytest.approx(info[k], rel=rel, abs=2e-3), k


def test_job_wide_episode_budget_and_even_shards(monkeypatch):
    """Under a one-process-per-GPU launcher ``--num_env_steps`` is the budget of the whole job: every rank derives the
    same episode count from the job-wide thread count (unequal counts would leave a rank waiting in a collective), and
    thread counts that do not divide over the ranks are refused."""
    import types
    import torch
    from onpolicy.scripts.train import _launch
    from onpolicy.utils import dist as mdist
    from onpolicy.runner.shared import base_runner

    with monkeypatch.context() as m:     # (undone before the session fixtures look at torch.cuda again)
        m.setenv("WORLD_SIZE", "4")
        m.setenv("RANK", "1")
        m.setenv("LOCAL_RANK", "0")
        m.setenv("MAPPO_SINGLE_DEVICE", "1")
        m.setattr(torch.cuda, "is_available", lambda: True)
        m.setattr(torch.cuda, "set_device", lambda d: None)
        m.setattr(mdist, "init_from_env", lambda device: None)
        m.setattr(mdist, "shard_threads", lambda n, rank=1, world=4: (n // world * rank, n // world * (rank + 1)))
        args = types.SimpleNamespace(cuda=True, n_training_threads=1, n_rollout_threads=10)
        with pytest.raises(ValueError, match="multiple of the number of ranks"):
            _launch.device_of(args)
        args.n_rollout_threads = 12
        _launch.device_of(args)
    assert (args.n_rollout_threads, args.rollout_thread_offset, args.global_n_rollout_threads) == (3, 3, 12)
    # the runners' episode count: job-wide threads (12), not this rank's 3
    src = open(base_runner.__file__).read()
    assert 'getattr(a, "global_n_rollout_threads", a.n_rollout_threads)' in src
    from onpolicy.runner.shared import mpe_runner
    assert "// self.n_rollout_threads_job" in open(mpe_runner.__file__).read()


def _run_two_ranks(tmp_path, cname, device, fixture_opts=None, fname="trainer_h64_cases"):
    import json
    spec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fname + ".json")))[cname]
    N, world, port = spec["spec"]["N"], 2, _free_port()
    ctx = mp.get_context("spawn")
    opts = dict(fixture_opts or {}, fname=fname)
    procs = [ctx.Process(target=dp_worker.worker, args=(r, world, port, N, {}, str(tmp_path), device, False, cname, opts))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert [r["span"] for r in ranks] == [(0, N // 2), (N // 2, N)]
    return spec, ranks


def _two_ranks_on_reference_fixture(tmp_path, cname, device, fixture_opts=None):
    """Two data-parallel ranks (gloo), each with half of the rollout threads of a reference-generated hidden-64 trainer
    case, against what the REFERENCE's single process produced (tests/golden/trainer_h64_cases.npz).  -> the ranks'
    records (info incl. the DataParallel counters, final state)."""
    import json
    spec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_h64_cases.json")))[cname]
    N, world, port = spec["spec"]["N"], 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dp_worker.worker, args=(r, world, port, N, {}, str(tmp_path), device, False, cname,
                                                        fixture_opts))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ranks = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert [r["span"] for r in ranks] == [(0, N // 2), (N // 2, N)]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trainer_h64_cases.npz"))
    for k in ranks[0]["sd"]:
        assert torch.equal(ranks[0]["sd"][k], ranks[1]["sd"][k]), k                  # replicas stay identical
        ref = z["trn_%s_%s" % (cname, k)]
        if k == "final_norm":
            np.testing.assert_allclose(ranks[0]["sd"][k].numpy(), ref, rtol=1e-5, atol=1e-9)
        else:
            np.testing.assert_allclose(ranks[0]["sd"][k].numpy(), ref, rtol=1e-3, atol=5e-5, err_msg=k)
    # global-batch quantities of the log: the gradient norms (the losses are rank means of per-rank means)
    for k in ("actor_grad_norm", "critic_grad_norm"):
        assert ranks[0]["info"][k] == pytest.approx(spec["train_info"][k], rel=2e-3), k
    return ranks


def test_two_rank_update_on_hidden64_reference_fixture(tmp_path):
    """(CPU ranks: the PyTorch modules at width 64; the device variant with the fused trunk is
    tests/test_gpu_bench.py::test_two_rank_device_update_on_hidden64_reference_fixture.)"""
    _two_ranks_on_reference_fixture(tmp_path, "h64_ns", None)
