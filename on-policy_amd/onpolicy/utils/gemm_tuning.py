"""GEMM kernel selection for the tall, skinny matrix products of the MAPPO update.

The Linear layers of the actor / critic see [10^6..10^7, 48..384] x [.., 64] products.  PyTorch-ROCm's default
pick (hipBLASLt's heuristic) is 15-35 % slower on these shapes than the best kernel the two BLAS libraries
offer (e.g. 1151 -> 925 us for the 384 -> 64 critic layer at 2.6 M rows, 358 -> 243 us for the 64 -> 64 layers).
PyTorch's own TunableOp benchmarks the candidates the first time a shape is seen and remembers the winner;
this module switches it on, pre-loads the winners for the shapes of the shipped workloads
(``tuned_gemms_gfx950.csv``: recurrent north star incl. its 8-GPU shard, SMAC, cfg2, Hanabi shapes -- the hidden-64 MLP
trunks no longer reach the BLAS libraries; refresh with tools/tune_gemms.sh + tools/merge_tuned_gemms.py) so that no tuning is needed
for them, and keeps whatever gets tuned later in a per-user cache file (one per device ordinal).

The maths is unchanged (float32 GEMMs; only the tile configuration / summation order differs).
``MAPPO_GEMM_TUNING=0`` disables it, ``=tune`` also benchmarks shapes that have no stored winner yet (inside the update
phase only); ``MAPPO_GEMM_TUNING_CACHE`` moves the cache directory.
"""
import os
import tempfile

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuned_gemms_gfx950.csv")


# whether shapes without a stored winner are benchmarked inside R_MAPPO.train() (``with tuning():``)
_TUNE_NEW_SHAPES = False


def enable(tune_new=True):
    """Turn TunableOp on for this process.  -> True if it is active.

    ``tune_new=False`` (the train scripts): stored winners only -- the shipped table and the per-user cache -- and
    nothing is ever benchmarked on line: tuning the GEMM shapes of an unseen configuration costs tens of seconds per
    shape, which made the FIRST update of e.g. ``train_hanabi_forward.py`` at 1024 tables take four minutes (round 2).
    ``MAPPO_GEMM_TUNING=tune`` (or ``tune_new=True``: bench.py, tools/tune_gemms.sh) benchmarks new shapes inside the
    update phase and stores the winners in the cache for later runs."""
    global _TUNE_NEW_SHAPES
    mode = os.environ.get("MAPPO_GEMM_TUNING", "1")
    if mode == "0" or not torch.cuda.is_available():
        return False
    if mode == "tune":
        tune_new = True
    _TUNE_NEW_SHAPES = bool(tune_new)
    t = torch.cuda.tunable
    try:
        t.enable(True)
        t.tuning_enable(bool(tune_new))
        t.set_max_tuning_duration(30)        # ms per candidate
        t.set_max_tuning_iterations(5)
        cache = os.environ.get("MAPPO_GEMM_TUNING_CACHE") or \
            os.path.join(tempfile.gettempdir(), "mappo_amd_gemm_tuning")
        os.makedirs(cache, exist_ok=True)
        t.set_filename(os.path.join(cache, "tunableop_results.csv"), insert_device_ordinal=True)
        for path in (SHIPPED, t.get_filename()):
            if os.path.exists(path):
                t.read_file(path)            # ignored (with a warning) if it was made by other library versions
    except Exception as e:                   # kernel selection is an optimisation: never fail a run over it
        print("gemm_tuning: not enabled (%s: %s)" % (type(e).__name__, e))
        try:
            t.enable(False)
        except Exception:
            pass
        return False
    return True


class tuning(object):
    """``with gemm_tuning.tuning():`` -- shapes first seen inside the block are benchmarked when the process asked for
    on-line tuning (``enable(tune_new=True)`` / ``MAPPO_GEMM_TUNING=tune``); outside it only stored winners are used.  The update phase runs the same few shapes every iteration and is worth
    tuning; a rollout is not: e.g. the turn-based Hanabi loop evaluates the policy on a different number of rows at
    almost every move, and benchmarking each of them (seconds per shape) stalls it for minutes."""

    def __enter__(self):
        self.was = None
        if _TUNE_NEW_SHAPES and torch.cuda.is_available() and torch.cuda.tunable.is_enabled():
            self.was = torch.cuda.tunable.tuning_is_enabled()
            torch.cuda.tunable.tuning_enable(True)
        return self

    def __exit__(self, *exc):
        if self.was is not None:
            torch.cuda.tunable.tuning_enable(self.was)


def results():
    """The (op, shape, solution, time) tuples TunableOp currently holds."""
    return torch.cuda.tunable.get_results() if torch.cuda.is_available() else ()
