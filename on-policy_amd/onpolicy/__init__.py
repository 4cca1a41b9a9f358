"""MI355X-native drop-in for the MAPPO rollout-buffer / GAE / sampler / update path of
marlbenchmark/on-policy.  The module tree mirrors the reference's import paths
(``onpolicy.utils.shared_buffer``, ``onpolicy.algorithms.r_mappo...``, ``onpolicy.runner.shared...``)
so that the reference's training scripts run against it unchanged; the compute lives in
``libmappo_hip.so`` (on-policy_amd/csrc, C ABI in include/mappo_hip.h).

Unlike the reference's ``onpolicy/__init__.py`` nothing is imported eagerly here: importing the package must not
load the HIP library, start PyTorch or pull in the optional simulators (SMAC / football come from an external env
tree, see ``onpolicy/envs/__init__.py``).
"""
__version__ = "0.1.0"
