#!/usr/bin/env python3
"""Builds an experimental variant of the library: python tools/build_variant.py NAME [-DTR_...=1 ...]
-> build_ab/libtr_NAME.so (git-ignored; travels to the GPU box with gpurun).  tools/ab_kernel.py times variants
against each other; the product build (__graft_entry__.build) refuses experimental flags."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
out = os.path.join(ROOT, "build_ab", f"libtr_{name}.so")
g.compile_library(out, ["-DTR_TUNING_ENV=1"] + extra, force=True)   # (variants read the tuning environment; the product does not)
print(out)
