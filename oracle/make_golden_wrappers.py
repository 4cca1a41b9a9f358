#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/wrapper_cases.npz by running the REFERENCE's VecEnv wrappers
(onpolicy/envs/env_wrappers.py: Dummy / Subproc, plain / Share / Choose / ChooseSimple protocols) over the tiny
deterministic env of tests/fake_envs.py with seeded actions: every array a reset / step returns.

    python oracle/make_golden_wrappers.py
"""
import functools
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("DummyVecEnv", False, False), ("SubprocVecEnv", False, False), ("ShareDummyVecEnv", True, False),
         ("ShareSubprocVecEnv", True, False), ("ChooseDummyVecEnv", True, True), ("ChooseSubprocVecEnv", True, True),
         ("ChooseSimpleDummyVecEnv", False, True), ("ChooseSimpleSubprocVecEnv", False, True)]


def flatten(x, prefix, out):
    if isinstance(x, (tuple, list)) and not (len(x) and isinstance(x[0], dict)) and not isinstance(x, np.ndarray):
        for i, y in enumerate(x):
            flatten(y, prefix + "_%d" % i, out)
    else:
        arr = np.asarray(x)
        if arr.dtype != object:
            out[prefix] = arr


def main():
    ref_import.load_reference()
    envs = types.ModuleType("onpolicy.envs")
    envs.__path__ = [os.path.join(ref_import.REFERENCE_ROOT, "onpolicy", "envs")]
    sys.modules["onpolicy.envs"] = envs
    from onpolicy.envs import env_wrappers as W           # the reference's
    import fake_envs
    out = {}
    n = 3
    for name, share, choose in CASES:
        fns = [functools.partial(fake_envs.TinyEnv, seed=10 + i, share=share, choose=choose) for i in range(n)]
        venv = getattr(W, name)(fns)
        rng = np.random.default_rng(0)
        flatten(venv.reset(np.array([True, False, True])) if choose else venv.reset(), name + "_reset", out)
        for t in range(7):
            actions = rng.integers(0, 4, size=(n, 2, 1))
            flatten(venv.step(actions), name + "_step%d" % t, out)
            if choose and t == 3:
                flatten(venv.reset(np.array([False, True, True])), name + "_reset_mid", out)
        venv.close()
    np.savez_compressed(os.path.join(GOLD, "wrapper_cases.npz"), **out)
    print("wrapper_cases.npz: %d arrays, %d B" % (len(out), os.path.getsize(os.path.join(GOLD, "wrapper_cases.npz"))))


if __name__ == "__main__":
    main()
