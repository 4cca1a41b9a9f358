"""onpolicy.config.get_config against the reference's flag table (tests/golden/config_flags.json, dumped from the
reference's own parser by oracle/make_golden_config.py): same flags, defaults, types, choices and store_true /
store_false polarity.  Flags that only exist here are listed explicitly."""
import json
import os

from conftest import GOLD
from onpolicy.config import get_config

OURS_ONLY = {"buffer_device", "sampler_rng", "gae_exact", "gae_scan", "matrix_arithmetic"}


def _describe(parser):
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        out[a.dest] = dict(flags=sorted(a.option_strings), kind=type(a).__name__, default=a.default,
                           type=None if a.type is None else a.type.__name__,
                           choices=None if a.choices is None else list(a.choices), nargs=a.nargs)
    return out


def test_flag_table_equals_reference():
    ref = json.load(open(os.path.join(GOLD, "config_flags.json")))
    ours = _describe(get_config())
    assert set(ours) - set(ref) == OURS_ONLY
    assert set(ref) - set(ours) == set()
    for dest, spec in ref.items():
        assert ours[dest] == spec, (dest, ours[dest], spec)


def test_store_false_flags_invert_when_passed():
    """e.g. --use_ReLU / --use_valuenorm / --share_policy are store_false in the reference: passing them turns
    the feature OFF (SURVEY.md section 2, config.py:203-266)."""
    a = get_config().parse_known_args(["--use_ReLU", "--use_valuenorm", "--share_policy", "--use_popart"])[0]
    assert a.use_ReLU is False and a.use_valuenorm is False and a.share_policy is False and a.use_popart is True
