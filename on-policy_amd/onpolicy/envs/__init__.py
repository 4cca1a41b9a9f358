"""Environment side of the package.  Provided here: the VecEnv wrappers (``env_wrappers``), duck-typed
spaces (``spaces``), a vectorised MPE ``simple_spread`` (``mpe.simple_spread``) and Hanabi as a batched native
stepper (``hanabi``).  The other simulators (SMAC, football, the other MPE scenarios) are out of scope (SURVEY.md
section 8).

To run against env packages that follow the reference's layout -- e.g. the reference's own
``onpolicy/envs`` directory -- list their directories in ``MAPPO_ENVS_PATH`` (``os.pathsep``-separated):
sub-modules that are not found here (``onpolicy.envs.starcraft2``, ``onpolicy.envs.football``, ...) are then
looked up there, under the same import names the reference's train scripts use.
"""
import os as _os


def _extend(path, sub=""):
    for root in _os.environ.get("MAPPO_ENVS_PATH", "").split(_os.pathsep):
        cand = _os.path.join(root, sub) if sub else root
        if root and _os.path.isdir(cand) and cand not in path:
            path.append(cand)


_extend(__path__)
