"""TEST INFRASTRUCTURE.  Imports the *reference* (marlbenchmark/on-policy, mounted read-only at
/root/reference in the build container) without running onpolicy/__init__.py, whose env imports
need packages that are not installed (absl, gym, ...).  Used only by oracle/make_golden.py and by
tests that are skipped when /root/reference is absent (it never exists on the GPU box).

The reference package is registered under the module name ``onpolicy`` -- so this must run in a
process that has NOT imported the product's own ``onpolicy`` package (make_golden.py is run as a
script; tests call it through a subprocess).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MAPPO_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "onpolicy"))


class Box(object):
    """Duck-typed gym.spaces.Box: the reference dispatches on the class NAME
    (onpolicy/utils/util.py:31-52)."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class Discrete(object):
    def __init__(self, n):
        self.n = n


def load_reference():
    if "onpolicy" in sys.modules and not getattr(sys.modules["onpolicy"], "_is_reference", False):
        raise RuntimeError("a different 'onpolicy' package is already imported in this process")
    pkg = types.ModuleType("onpolicy")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "onpolicy")]
    pkg._is_reference = True
    sys.modules["onpolicy"] = pkg
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.utils.valuenorm import ValueNorm
    from onpolicy.algorithms.utils.popart import PopArt
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.config import get_config
    return types.SimpleNamespace(SharedReplayBuffer=SharedReplayBuffer, ValueNorm=ValueNorm,
                                 PopArt=PopArt, R_MAPPO=R_MAPPO, R_MAPPOPolicy=R_MAPPOPolicy,
                                 get_config=get_config, Box=Box, Discrete=Discrete)
