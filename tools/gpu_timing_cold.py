"""Per-tile wait cycles of the 4K transmissive pass (a -DTR_TIMING=1 build: every wave adds up the cycles it waited for
the planes, the cluster lists and the taps) with COLD inputs (rotating input sets) and with one re-read set.
    python tools/build_variant.py timing -DTR_TIMING=1 && python tools/gpu_timing_cold.py build_ab/libtr_timing.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transmission_renderer_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])          # experiments only (the timing build)
import bench
wl = bench.PassWorkload(0, 3840, 2160, sets=4, split=1)
fn = wl.r.lib.tr_debug_read_timing
buf = (C.c_ulonglong * 12)()
for rotate in (True, False):
    wl.ramp(0.1)
    torch.cuda.synchronize(); fn(buf)
    n = 50
    for k in range(n): wl.step(k, rotate)
    torch.cuda.synchronize(); fn(buf)
    w0, w1, w2, total, tiles, waves = list(buf)[:6]
    waves &= 0xFFFFFFFF
    print(f"{'cold (4 rotating sets)' if rotate else 'one re-read set'}: per tile and wave (cycles): planes wait {w0 / tiles:.0f}, cluster lists {w1 / tiles:.0f}, "
          f"taps+LUT {w2 / tiles:.0f}, whole tile {total / tiles:.0f}; {tiles / waves:.2f} tiles per wave; ms per launch {wl.timed(50, rotate=rotate)[0]*1e3:.1f} us")
