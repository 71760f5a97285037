"""`LightClusterCoefficients::get_depth_slice` (shared-structs/src/lib.rs:43-63) is index work: bit-exact.

The library settles the slice against a table of depth thresholds built on the host with the reference's own fp32
operations (tr_depth_slice_thresholds).  CPU part: the table against the oracle.  GPU part (-m gpu): the device
function the shading passes use for their cluster lookup (tr_get_depth_slice) against the oracle on dense sweeps
around every threshold, random depths, the edge values, and other near/far/slice configurations."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from transmission_renderer_amd import _lib, wire

MAX_SLICES = 64

CONFIGS = [
    (wire.Z_NEAR, wire.Z_FAR, wire.NUM_DEPTH_SLICES),   # src/main.rs:60-62: 0.01, 500, 16
    (0.1, 100.0, 24),
    (0.05, 2000.0, 32),
    (1.0, 50.0, 8),
    (0.001, 10000.0, 64),
]


def _oracle_slices(coeffs, depths):
    L = oracle.load()
    return np.array([L.o_get_depth_slice(C.byref(coeffs), float(d)) for d in depths], dtype=np.uint32)


def _thresholds(coeffs):
    lib = C.CDLL(_lib.LIB_PATH)
    lib.tr_depth_slice_thresholds.restype = C.c_int32
    lib.tr_depth_slice_thresholds.argtypes = [C.POINTER(wire.LightClusterCoefficients), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    thr = (C.c_float * (MAX_SLICES + 2))()
    top = C.c_uint32()
    assert lib.tr_depth_slice_thresholds(C.byref(coeffs), thr, C.byref(top)) == 0
    return np.array(thr[:top.value + 2], dtype=np.float32), top.value


def _neighbours(x, n):
    """the n floats on each side of x (bit-pattern neighbours), x >= 0 finite"""
    b = np.float32(x).view(np.uint32).astype(np.int64)
    bits = np.clip(np.arange(b - n, b + n + 1), 0, 0x7F800000).astype(np.uint32)
    return bits.view(np.float32)


@pytest.mark.parametrize("near,far,slices", CONFIGS)
def test_threshold_table_against_the_oracle(near, far, slices):
    c = wire.LightClusterCoefficients.new(near, far, slices)
    thr, top = _thresholds(c)
    assert top == _oracle_slices(c, [0.0])[0] and top in (slices - 1, slices, slices + 1)
    assert np.isposinf(thr[0]) and thr[top + 1] == -1.0
    assert (np.diff(thr[1:top + 1]) <= 0).all()    # decreasing depths for increasing slices (far slices can share a
                                                   # threshold: 1 - d drops the low bits of a far depth)
    for k in range(1, top + 1):
        t = thr[k]
        up = np.nextafter(t, np.float32(np.inf), dtype=np.float32)
        s_at, s_up = _oracle_slices(c, [t, up])
        assert s_at >= k > s_up, (k, t, s_at, s_up)                  # the largest depth whose slice is >= k
    # slice(d) = #{k >= 1 : d <= thr[k]} on random depths, the whole [0, 1] range and beyond
    rng = np.random.default_rng(3)
    d = np.concatenate([rng.random(4000, dtype=np.float32), rng.random(2000, dtype=np.float32) ** 8,
                        np.float32([0.0, 1.0, 1.5, 1e-30, 1e-10, 3.0e38, np.inf])])
    want = _oracle_slices(c, d)
    got = (d[:, None] <= thr[None, 1:top + 1]).sum(axis=1)
    np.testing.assert_array_equal(got, want)


def test_unsupported_coefficients_are_refused():
    c = wire.LightClusterCoefficients.new(0.01, 500.0, 200)          # far-plane slice 200 > TR_MAX_DEPTH_SLICES
    lib = C.CDLL(_lib.LIB_PATH)
    thr = (C.c_float * (MAX_SLICES + 2))()
    top = C.c_uint32()
    lib.tr_depth_slice_thresholds.argtypes = [C.POINTER(wire.LightClusterCoefficients), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    assert lib.tr_depth_slice_thresholds(C.byref(c), thr, C.byref(top)) == 6   # TR_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("near,far,slices", CONFIGS)
def test_device_depth_slice_is_bit_exact(near, far, slices):
    torch = pytest.importorskip("torch")
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    c = wire.LightClusterCoefficients.new(near, far, slices)
    thr, top = _thresholds(c)
    rng = np.random.default_rng(17)
    parts = [_neighbours(t, 3000) for t in thr[1:top + 1]]          # every float within 3000 ulps of every threshold
    parts.append(rng.random(1 << 20, dtype=np.float32))              # uniform in [0, 1)
    parts.append(rng.random(1 << 19, dtype=np.float32) ** 6)         # towards the far plane, where 1 - d loses bits
    parts.append(np.float32(1.0) - rng.random(1 << 18, dtype=np.float32) ** 6 * np.float32(0.01))   # near plane
    # logarithmically spread bit patterns (every exponent), the edge values, NaN
    parts.append(rng.integers(0, 0x3F800001, 1 << 19, dtype=np.uint32).view(np.float32))
    parts.append(np.float32([0.0, 1.0, 1.0000001, 2.0, 1e-38, 1e-45, 3.0e38, np.inf, np.nan]))
    d = np.concatenate(parts)
    # shuffled: nearly every wave has a lane near a boundary and takes the table path; sorted: neighbouring lanes hold
    # neighbouring depths, so waves away from the thresholds take the estimate-only path
    d = np.concatenate([d, np.sort(d[np.isfinite(d)])])
    got = r.get_depth_slice(c, torch.from_numpy(d).to(r.device)).cpu().numpy().view(np.uint32)
    lib = oracle.load()
    want = np.empty(d.size, dtype=np.uint32)
    fn = lib.o_get_depth_slice
    cref = C.byref(c)
    for i, x in enumerate(d.tolist()):
        want[i] = fn(cref, x)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad.size, d[bad[:5]], got[bad[:5]], want[bad[:5]])
    r.close()
