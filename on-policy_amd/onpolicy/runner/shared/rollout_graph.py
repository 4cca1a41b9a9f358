"""One rollout step of the device-resident loop as ONE graph launch.

With the worlds on the GPU (``--use_device_env``: K11) a rollout step of the shared MPE runner -- policy forward of
actor and critic, action sampling, env step, mask / RNN-state bookkeeping (reference
onpolicy/runner/shared/mpe_runner.py:96-139: collect -> envs.step -> insert) -- is ~30 small kernels on a few thousand
rows: pure launch latency (0.75 ms per step at 4096 worlds x 3 agents, five times the update's share of a config-3
iteration).  Nothing in it depends on the host, and every step runs the same kernels on the same buffers except for the
slab of the rollout buffer it writes.  So the step is captured once into a HIP graph that reads and advances a small
carried state (current observation, masks, RNN states) held in static tensors and leaves its results in static output
tensors; a rollout step is then one graph launch + the one fused slab write (K2) into row ``step`` of the buffer, whose
destination pointers are the only thing that changes from step to step.

Random numbers: the action sampler draws from torch's default device generator and the worlds from their own
``torch.Generator``; both are registered with the graph (philox offsets advance with every replay), so a graphed
rollout consumes the same random stream as the eager loop (tests/test_gpu_rollout_graph.py compares whole rollouts).

Falls back to the eager loop (returns None from ``build``) when the step cannot be captured: host envs, worlds whose step
rebinds its state tensors (more than 16 agents / landmarks: the tensor-op path), the integer-parity sampler
(``--sampler_rng host`` draws on the CPU generator), no HIP device.  A PopArt value head is read through static copies that
every ``begin_episode`` refreshes (its ``update`` rebinds the parameters' storage).
"""
import os

import torch

from onpolicy.utils.graph_capture import capturing


class RolloutGraph(object):
    WARMUP = 3

    def __init__(self, runner):
        self.r = runner
        b = runner.buffer
        self.N, self.A = runner.n_rollout_threads, runner.num_agents
        dev = b.device
        f32 = dict(dtype=torch.float32, device=dev)
        # carried state of the loop
        self.cur_obs = torch.empty(b.obs.shape[1:], **f32)
        self.cur_masks = torch.empty(b.masks.shape[1:], **f32)
        self.recurrent = b.rnn_states.stride()[0] != 0
        if self.recurrent:
            self.cur_rnn_a = torch.empty(b.rnn_states.shape[1:], **f32)
            self.cur_rnn_c = torch.empty(b.rnn_states_critic.shape[1:], **f32)
        else:       # feed-forward policies: the buffer's zero view serves every step
            self.cur_rnn_a, self.cur_rnn_c = b.rnn_states[0], b.rnn_states_critic[0]
        self.side = torch.cuda.Stream(device=dev)      # the critic's branch of the captured step
        # A PopArt value head REBINDS its weight / bias storage on every update (algorithms/utils/popart.py: the running
        # minibatch's backward still needs the old tensors), so the addresses a capture bakes in would go stale with the first
        # train().  The graph therefore reads static copies of the head, refreshed at the start of every episode.
        head = getattr(runner.trainer.policy.critic, "v_out", None)
        self.popart = head if type(head).__name__ == "PopArt" else None
        if self.popart is not None:
            self.v_w = torch.empty_like(self.popart.weight.data)
            self.v_b = torch.empty_like(self.popart.bias.data)
        self.graph = None
        self.out = None
        self.infos = None
        self.replays = 0

    # -- the step itself: eager code, captured once
    @torch.no_grad()
    def _body(self):
        r, N, A = self.r, self.N, self.A
        policy = r.trainer.policy
        dev = r.buffer.device
        share = r._share(self.cur_obs, N)
        # The critic's branch (centralised observation -> value) depends on nothing the actor's branch produces, and a
        # forward launch on N * A rows fills a third of the CUs: it is recorded on a side stream, i.e. as a parallel
        # branch of the graph, and runs under the actor forward / sampling / env step of the main branch.
        main = torch.cuda.current_stream(dev)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            values, rnn_c = policy.critic(r._rows(share), r._rows(self.cur_rnn_c), r._rows(self.cur_masks))
            values, rnn_c = r._per_env(values), r._per_env(rnn_c)
        actions, logp, rnn_a = policy.actor(r._rows(self.cur_obs), r._rows(self.cur_rnn_a), r._rows(self.cur_masks))
        actions, logp, rnn_a = r._per_env(actions), r._per_env(logp), r._per_env(rnn_a)
        obs, rewards, dones, infos = r.envs.step(actions)
        masks = torch.where(dones, 0.0, 1.0).unsqueeze(-1)          # 0 where the episode ended (one launch)
        main.wait_stream(self.side)
        if self.recurrent:
            # finished agents restart from a zero RNN state (reference mpe_runner.py:128-129)
            rnn_a = rnn_a * masks.view(N, A, 1, 1)
            rnn_c = rnn_c * masks.view(N, A, 1, 1)
        # the carried state of the next step
        self.cur_obs.copy_(obs)
        self.cur_masks.copy_(masks)
        if self.recurrent:
            self.cur_rnn_a.copy_(rnn_a)
            self.cur_rnn_c.copy_(rnn_c)
        # (the centralised observation is materialised here, inside the graph, so that the slab write takes it as it is)
        share_next = r._share(obs, N).contiguous() if r.use_centralized_V else obs
        return (share_next, obs, rnn_a, rnn_c, actions, logp, values, rewards, masks), infos

    def _env_state(self):
        e = self.r.envs
        return {k: getattr(e, k).clone() for k in ("pos", "vel", "landmarks", "t")}, e.rng.get_state(), \
            torch.cuda.get_rng_state(self.r.buffer.device)

    class _StaticHead(object):
        """While active, the PopArt head's parameters live in the graph's static tensors (same values)."""

        def __init__(self, g):
            self.g = g

        def __enter__(self):
            g = self.g
            if g.popart is not None:
                self.saved = (g.popart.weight.data, g.popart.bias.data)
                g.v_w.copy_(self.saved[0])
                g.v_b.copy_(self.saved[1])
                g.popart.weight.data, g.popart.bias.data = g.v_w, g.v_b

        def __exit__(self, *exc):
            g = self.g
            if g.popart is not None:
                g.popart.weight.data, g.popart.bias.data = self.saved

    def _restore(self, snap):
        e = self.r.envs
        state, env_rng, dev_rng = snap
        for k, v in state.items():
            getattr(e, k).copy_(v)
        e.rng.set_state(env_rng)
        torch.cuda.set_rng_state(dev_rng, self.r.buffer.device)

    def capture(self):
        """Warm the step up on a side stream (library workspaces, lazy module state), capture it, and put worlds and
        generators back where they were: building the graph must not show in the trajectories."""
        r = self.r
        dev = r.buffer.device
        r.trainer.prep_rollout()
        self.begin_episode()
        keep = (self.cur_obs.clone(), self.cur_masks.clone(),
                self.cur_rnn_a.clone() if self.recurrent else None, self.cur_rnn_c.clone() if self.recurrent else None)
        snap = self._env_state()
        state_ptrs = {k: getattr(r.envs, k).data_ptr() for k in ("pos", "vel", "landmarks", "t")}
        torch.cuda.synchronize(dev)
        try:        # (warm-up included: whatever fails in here, worlds / generators / carried state go back where they were)
            with self._StaticHead(self):
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(self.WARMUP):
                        self._body()
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                graph.register_generator_state(r.envs.rng)
                with capturing(graph):
                    self.out, self.infos = self._body()
                torch.cuda.synchronize(dev)
            # the graph reads and advances the worlds IN PLACE: a step path that rebinds its state tensors cannot be replayed
            moved = [k for k, p in state_ptrs.items() if getattr(r.envs, k).data_ptr() != p]
            if moved:
                raise RuntimeError("env.step rebinds its state tensors (%s): not capturable" % ", ".join(moved))
            self.graph = graph
        finally:
            self._restore(snap)
            self.cur_obs.copy_(keep[0])
            self.cur_masks.copy_(keep[1])
            if self.recurrent:
                self.cur_rnn_a.copy_(keep[2])
                self.cur_rnn_c.copy_(keep[3])
        return self

    def begin_episode(self):
        """Row 0 of the buffer is the state the rollout starts from (warmup() / after_update() put it there)."""
        b = self.r.buffer
        self.cur_obs.copy_(b.obs[0])
        self.cur_masks.copy_(b.masks[0])
        if self.recurrent:
            self.cur_rnn_a.copy_(b.rnn_states[0])
            self.cur_rnn_c.copy_(b.rnn_states_critic[0])
        if self.popart is not None:         # the head as the last train() left it
            self.v_w.copy_(self.popart.weight.data)
            self.v_b.copy_(self.popart.bias.data)

    def step(self):
        """One rollout step: graph launch + the fused slab write into the buffer's current row."""
        self.graph.replay()
        self.replays += 1
        self.r.buffer.insert(*self.out)
        # (a fresh lazy view of the static per-agent rewards: the previous one may have cached an older step)
        per_agent = getattr(self.infos, "_per_agent", None)
        return type(self.infos)(per_agent) if per_agent is not None else self.infos


def build(runner):
    """-> a captured RolloutGraph for ``runner``, or None when the step has to stay eager."""
    from onpolicy.algorithms.utils import distributions
    if os.environ.get("MAPPO_ROLLOUT_GRAPH", "1") == "0":
        return None
    envs = runner.envs
    if not getattr(envs, "device_resident", False) or not hasattr(envs, "rng") or not hasattr(envs, "pos"):
        return None
    if not getattr(envs, "graph_safe", False):      # the step must advance its state tensors in place (K11 does)
        return None
    if torch.device(runner.buffer.device).type != "cuda" or distributions.SAMPLING_RNG != "device":
        return None
    if getattr(runner, "_mat", False):
        return None
    try:
        return RolloutGraph(runner).capture()
    except Exception as exc:        # capture is an optimisation: anything it cannot take stays on the eager loop
        print("rollout graph: capture failed (%s: %s); the rollout stays eager" % (type(exc).__name__, exc))
        try:
            torch.cuda.synchronize(runner.buffer.device)
        except Exception:
            pass
        return None
