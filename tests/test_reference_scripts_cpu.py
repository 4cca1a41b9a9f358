"""Drop-in check against the reference's own train scripts (scripts/train/*.py are the
boundary): every ``onpolicy.*`` name they import must resolve in this package, except the SMAC / football
simulators, which come from an external env tree (MAPPO_ENVS_PATH).  Skipped where the reference is not mounted
(the GPU box)."""
import ast
import importlib
import os

import pytest

REF_SCRIPTS = "/root/reference/onpolicy/scripts/train"
EXTERNAL = ("onpolicy.envs.starcraft2", "onpolicy.envs.football")
# flags of ours that the reference does not have (script -> names)
EXTRA_FLAGS = {"train_hanabi_forward.py": {"use_subproc_envs": False}, "train_mpe.py": {"use_device_env": False}}
# runner modules the reference's scripts name but the reference itself does not contain
ABSENT_IN_REFERENCE = ("onpolicy.runner.separated.hanabi_runner_forward", "onpolicy.runner.separated.football_runner")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SCRIPTS), reason="reference not mounted")


def _imports(path):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("onpolicy"):
            yield node.module, [a.name for a in node.names]
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.startswith("onpolicy"):
                    yield a.name, []


@pytest.mark.parametrize("script", ["train_mpe.py", "train_smac.py", "train_hanabi_forward.py", "train_football.py"])
def test_reference_train_script_imports_resolve(script):
    seen = 0
    for module, names in _imports(os.path.join(REF_SCRIPTS, script)):
        if module.startswith(EXTERNAL) or module in ABSENT_IN_REFERENCE:
            continue
        mod = importlib.import_module(module)
        for name in names:
            assert hasattr(mod, name), "%s: %s has no %s" % (script, module, name)
        seen += 1
    assert seen >= 3


def test_reference_flags_are_accepted_by_our_scripts():
    """The per-script flags of the reference (parse_args of each train script) exist here with the same
    defaults."""
    from onpolicy.config import get_config
    from onpolicy.scripts.train import train_mpe, train_smac, train_hanabi_forward, train_football
    for ours, script in ((train_mpe, "train_mpe.py"), (train_smac, "train_smac.py"),
                         (train_hanabi_forward, "train_hanabi_forward.py"), (train_football, "train_football.py")):
        src = open(os.path.join(REF_SCRIPTS, script)).read()
        tree = ast.parse(src)
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
        mod = ast.Module(body=[fn], type_ignores=[])
        ns = {}
        exec(compile(mod, script, "exec"), ns)                     # the reference's own parse_args, on our parser
        ref_args = ns["parse_args"]([], get_config())
        our_args = ours.parse_args([], get_config())
        extra = EXTRA_FLAGS.get(script, {})
        assert dict(vars(ref_args), **extra) == vars(our_args), script


def _run_reference_main(script, argv, tmp_path, monkeypatch):
    """Executes the reference's own ``main()`` (its source, compiled under a scratch ``__file__`` so that its run
    directory lands in tmp_path, not in /root/reference) with this package on the import path.  ``wandb`` and
    ``setproctitle`` (absent here; imported at the top of the script, unused with --use_wandb) are empty stand-ins."""
    import sys
    import types
    for name in ("wandb", "setproctitle"):
        if name not in sys.modules:
            stub = types.ModuleType(name)
            stub.setproctitle = lambda *_a, **_k: None
            monkeypatch.setitem(sys.modules, name, stub)
    fake = tmp_path / "onpolicy" / "scripts" / "train" / script
    fake.parent.mkdir(parents=True)
    src = open(os.path.join(REF_SCRIPTS, script)).read()
    ns = {"__file__": str(fake), "__name__": "reference_" + script[:-3]}
    exec(compile(src, str(fake), "exec"), ns)
    ns["main"](argv)
    return tmp_path / "onpolicy" / "scripts" / "results"


def test_reference_train_mpe_main_runs_against_this_package(tmp_path, monkeypatch):
    """The boundary, dynamically: the reference's scripts/train/train_mpe.py main() -- parser, env factory, runner
    selection, run(), post-processing -- drives THIS package's config / VecEnv / MPE env / runner / policy / trainer for
    two episodes on the CPU.  The rollout buffer of the product lives in HBM and has no CPU form, so the runner's buffer
    class is the oracle-backed host stand-in of the other CPU runner tests (test infrastructure); everything else is the
    product code the script imports."""
    import json
    import onpolicy.runner.shared.base_runner as base
    from host_buffer import HostSharedBuffer
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    results = _run_reference_main("train_mpe.py", [
        "--env_name", "MPE", "--algorithm_name", "mappo", "--experiment_name", "boundary", "--scenario_name",
        "simple_spread", "--num_agents", "3", "--num_landmarks", "3", "--seed", "1", "--n_training_threads", "1",
        "--n_rollout_threads", "2", "--num_mini_batch", "1", "--episode_length", "10", "--num_env_steps", "40",
        "--ppo_epoch", "2", "--use_ReLU", "--gain", "0.01", "--lr", "7e-4", "--critic_lr", "7e-4",
        "--use_wandb", "--cuda", "--log_interval", "1", "--save_interval", "1"], tmp_path, monkeypatch)
    run = results / "MPE" / "simple_spread" / "mappo" / "boundary" / "run1"
    assert (run / "models" / "actor.pt").exists() and (run / "models" / "critic.pt").exists()
    summary = json.load(open(run / "logs" / "summary.json"))
    assert any("average_episode_rewards" in k for k in summary), sorted(summary)[:5]


def test_reference_train_hanabi_main_runs_against_this_package(tmp_path, monkeypatch):
    """Same for scripts/train/train_hanabi_forward.py: the reference's main() builds ChooseDummyVecEnv over this package's
    HanabiEnv (the batched native stepper) and runs the turn-based forward runner for two episodes."""
    import json
    import onpolicy.runner.shared.base_runner as base
    from host_buffer import HostSharedBuffer
    monkeypatch.setattr(base, "SharedReplayBuffer", HostSharedBuffer)
    results = _run_reference_main("train_hanabi_forward.py", [
        "--env_name", "Hanabi", "--algorithm_name", "mappo", "--experiment_name", "boundary", "--hanabi_name",
        "Hanabi-Very-Small", "--num_agents", "2", "--seed", "1", "--n_training_threads", "1", "--n_rollout_threads", "1",
        "--n_eval_rollout_threads", "1", "--num_mini_batch", "1", "--episode_length", "12", "--num_env_steps", "24",
        "--ppo_epoch", "2", "--gain", "0.01", "--lr", "7e-4", "--critic_lr", "1e-3", "--hidden_size", "64",
        "--layer_N", "1", "--entropy_coef", "0.015", "--use_wandb", "--cuda", "--log_interval", "1",
        "--save_interval", "1"], tmp_path, monkeypatch)
    run = results / "Hanabi" / "Hanabi-Very-Small" / "mappo" / "boundary" / "run1"
    assert (run / "models" / "actor.pt").exists()
    assert json.load(open(run / "logs" / "summary.json")) is not None
