#!/usr/bin/env python3
"""Build experiment variants of the library next to the product one (gpurun_ab/lib_<name>.so), for same-box A/B runs
through PNDF_LIBRARY (tools/ab_bench.py).  Usage: python tools/build_variants.py name=-DFLAG[,-DFLAG2] ..."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as g  # noqa: E402

for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    out = os.path.join(REPO, "gpurun_ab", f"lib_{name}.so")
    g.build_library(out, flags=[f for f in flags.split(",") if f], tag="_" + name)
    print("built", out)
