"""Golden vectors for the command line (build container only): the reference's own argument parser
(``__main__.py``: ArgumentParserWithConfig + parse_args, compiled out of its syntax tree) run on a set of argv
lists; the parsed dictionaries go to ``cli.json``.

    python tests/golden/make_golden_cli.py
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
from typing import Any

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_host import REF, extract  # noqa: E402


class _Torch:
    class cuda:
        @staticmethod
        def is_available():
            return True


def main():
    ns = {"argparse": argparse, "json": json, "sys": sys, "Any": Any, "torch": _Torch}
    extract(os.path.join(REF, "__main__.py"), {"ArgumentParserWithConfig", "parse_args"}, ns)
    # classes are module-level ClassDef nodes: extract() only handles functions / assignments / methods, so add them
    import ast
    tree = ast.parse(open(os.path.join(REF, "__main__.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ArgumentParserWithConfig"]
    mod = ast.Module(body=cls, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, "ref_main", "exec"), ns)
    cases = {}
    with tempfile.TemporaryDirectory() as d:
        cfg = os.path.join(d, "cfg.json")
        json.dump({"output_size": [128, 96], "strategy": "all", "det_threshold": 0.7, "batch_size": 4,
                   "attr_groups": {"glasses": [6]}, "enh_threshold": 0.002}, open(cfg, "w"))
        argvs = {
            "minimal": ["-i", "imgs"],
            "everything": ["-i", "in", "-o", "out", "-s", "200", "300", "-f", "png", "-r", "800", "-ff", "0.8", "-st", "best",
                           "-p", "reflect", "-a", "-l", "lm.json", "-ag", '{"g": [6], "n": [-6]}', "-mg", '{"eyes": [4, 5]}',
                           "-dt", "0.5", "-et", "0.001", "-b", "16", "-n", "3", "-d", "cuda:1", "-cn"],
            "negative_thresholds": ["-i", "x", "-dt", "-1", "-et", "-0.5"],
            "single_sizes": ["-i", "x", "-s", "300", "-r", "640"],
            "config": ["-c", cfg, "-i", "x"],
            "config_overridden": ["-c", cfg, "-i", "x", "-st", "largest", "-b", "2"],
            "inplace": ["-i", "x", "-ci"],
        }
        for name, argv in argvs.items():
            sys.argv = ["face-crop-plus"] + argv
            kw = ns["parse_args"]()
            cases[name] = {"argv": [a if a != cfg else "<CFG>" for a in argv], "kwargs": kw}
        cases["_config_file"] = json.load(open(cfg))
        sys.argv = ["face-crop-plus"]
        try:
            ns["parse_args"]()
            cases["_no_input"] = "ok"
        except BaseException as e:
            cases["_no_input"] = type(e).__name__
    json.dump(cases, open(os.path.join(HERE, "cli.json"), "w"), indent=1, sort_keys=True)
    print("wrote cli.json:", sorted(cases))


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
