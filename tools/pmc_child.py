#!/usr/bin/env python3
"""The workload of tools/pmc_groups.sh: N whole-frame launches of the 4K transmissive pass (cold inputs) with the library
of TR_AB_LIB (TR_ABLATE as the environment says), nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmission_renderer_amd import _lib
if os.environ.get("TR_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TR_AB_LIB"])
import torch
import bench
cfg = os.environ.get("TR_AB_CFG", "head")
kw = dict(head={}, allt=dict(all_transmissive=True), c3=dict(lights=4, roughness=0.25))[cfg]
wl = bench.PassWorkload(0, 3840, 2160, split=1, **kw)
for k in range(int(os.environ.get("TR_PMC_LAUNCHES", "12"))):
    wl.step(k)
torch.cuda.synchronize()
