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
