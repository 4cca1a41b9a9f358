"""Stream capture with Python's cyclic garbage collector out of the way.

``torch.cuda.graph`` no longer collects garbage when a capture begins (PyTorch 2.10: only with
``torch.compiler.config.force_cudagraph_gc``).  A dead reference cycle that holds a ``torch.cuda.CUDAGraph`` -- a trainer and its
``UpdateGraph``, a runner and its ``RolloutGraph`` of an earlier run in the same process -- is then destroyed whenever the
collector happens to run, and if that is in the middle of ANOTHER capture on the same thread the graph's destruction is an
illegal call during capture: the process aborts (seen once in the device suite, round 5: ``Fatal Python error: Aborted`` with
``Garbage-collecting`` under ``UpdateGraph._capture``).  So: collect before the capture begins, keep the collector off until it
has ended."""
import contextlib
import gc

import torch


@contextlib.contextmanager
def capturing(graph, **kw):
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()
