#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Dumps the REFERENCE's command-line contract (onpolicy/config.py get_config: every flag,
its default, type, choices and whether passing it sets True or False) to tests/golden/config_flags.json.

    python oracle/make_golden_config.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def describe(parser):
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        kind = type(a).__name__                      # _StoreAction / _StoreTrueAction / _StoreFalseAction
        out[a.dest] = dict(flags=sorted(a.option_strings), kind=kind, default=a.default,
                           type=None if a.type is None else a.type.__name__,
                           choices=None if a.choices is None else list(a.choices), nargs=a.nargs)
    return out


def main():
    ref = ref_import.load_reference()
    flags = describe(ref.get_config())
    with open(os.path.join(GOLD, "config_flags.json"), "w") as f:
        json.dump(flags, f, indent=1, sort_keys=True)
    print("config_flags.json: %d flags" % len(flags))


if __name__ == "__main__":
    main()
