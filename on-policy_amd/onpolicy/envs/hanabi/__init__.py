"""Hanabi for config 5: a batched C++ stepper (csrc/hanabi_batch.cc, C ABI
include/hanabi_batch.h) behind the reference's env interface -- ``Hanabi_Env.HanabiEnv`` is the per-env class
the reference's train script builds, ``batch.HanabiBatchVecEnv`` steps all rollout threads in one native call."""
