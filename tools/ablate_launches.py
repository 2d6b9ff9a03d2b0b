#!/usr/bin/env python3
"""Timing ablation of the bench step: the engine's launches whose name starts with one of the given prefixes are SKIPPED after their
first 64 calls (so every buffer they write holds sane values) -- wrong training, same kernels otherwise.  The step-time difference
bounds what folding those launches into their neighbours could buy.
usage: ablate_launches.py <prefix>[,<prefix>...] [bench.py args...]      e.g. ablate_launches.py sed_bn_train_finalize,sed_bn_bwd_finalize"""
import runpy
import sys

sys.path.insert(0, ".")
import sed_amd  # noqa: E402

skip = tuple(x for x in sys.argv[1].split(",") if x)
E = sed_amd.engine.CnnEngine
orig, calls = E._k, {}


def _k(self, name, fn, *args):
    if skip and name.startswith(skip):
        calls[name] = calls.get(name, 0) + 1
        if calls[name] > 64:
            return None
    return orig(self, name, fn, *args)


E._k = _k
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
