"""The pieces of the sharded full pipeline's halo exchange on the one GPU of the test box (run with -m gpu): per-band
levels 1 / 2 (tr_generate_mips_band), the chain from level 3 (tr_generate_mips_from), the transmissive pass's tap-window
check (tr_set_tap_window) on a band whose out-of-window rows are POISONED, tr_exchange_halo through a one-rank RCCL
communicator, and the RGB8 tonemap a rank composites.  The N-rank exchange itself is covered over gloo on CPU
(tests/test_sharding_cpu.py: record_sharded(exchange="halo") bit for bit against the single-rank frame)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from transmission_renderer_amd import sharded, synthetic  # noqa: E402


@pytest.fixture(scope="module")
def renderer(ggx_lut):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    yield r
    r.close()


def _pyramid(r, w, h, seed=0):
    from transmission_renderer_amd.renderer import OpaquePyramid
    pyr = OpaquePyramid(w, h, r.device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    pyr.level(0).copy_((torch.rand((h, w, 4), generator=g) * 4.0).to(torch.float16).to(r.device))
    return pyr


@pytest.mark.parametrize("w,h,world", [(64, 48, 2), (256, 200, 3), (1920, 1080, 8), (3840, 2160, 8)])
def test_band_mips_then_chain_from_level_3_equal_the_whole_chain(renderer, w, h, world):
    r = renderer
    want = _pyramid(r, w, h)
    r.generate_mips(want)
    got = _pyramid(r, w, h)
    for l in range(1, got.levels):
        got.level(l).fill_(-3.0)
    for rank in range(world):                      # every "rank" in turn: levels 1 and 2 of its band only
        _, y0, y1 = sharded.band_rows(h, world, rank)
        r.generate_mips_band(got, y0, y1)
    torch.cuda.synchronize()
    for l in (1, 2):
        assert torch.equal(got.level(l).view(torch.int16), want.level(l).view(torch.int16)), l
    r.generate_mips_from(got, 3)
    torch.cuda.synchronize()
    assert torch.equal(got.texels.view(torch.int16), want.texels.view(torch.int16))


def test_band_mips_refusals(renderer):
    from transmission_renderer_amd import _lib
    pyr = _pyramid(renderer, 66, 48)               # a width that is not a multiple of 4: the boxes would leave the band
    with pytest.raises(_lib.TrError):
        renderer.generate_mips_band(pyr, 0, 24)
    pyr = _pyramid(renderer, 64, 48)
    with pytest.raises(_lib.TrError):
        renderer.generate_mips_band(pyr, 6, 24)    # not on a 4-row boundary


def _scene(r, w, h, thickness_scale):
    from test_gpu_parity import _upload_scene
    scene = synthetic.make_scene(w, h, num_point_lights=2)
    for m in scene["materials"]:
        m.thickness_factor *= thickness_scale
    _upload_scene(r, scene)
    return scene


@pytest.mark.parametrize("w,h,world,rank,halo,thickness_scale", [(640, 360, 3, 1, 16, 0.02), (1920, 1080, 4, 2, 64, 0.05),
                                                                 (640, 360, 2, 0, 24, 0.02)])
def test_tap_window_inside_poisoned_rows_outside(renderer, w, h, world, rank, halo, thickness_scale):
    """A band shaded with only the window's rows of levels 0 / 1 present (every other row NaN) equals the band of the
    whole-pyramid pass bit for bit when the reported excursion is 0."""
    from transmission_renderer_amd.renderer import GBufferPlanes
    r = renderer
    scene = _scene(r, w, h, thickness_scale)
    g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
    pyr = _pyramid(r, w, h, seed=3)
    r.generate_mips(pyr)
    want = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    _, y0, y1 = sharded.band_rows(h, world, rank)
    rect = (0, y0, w, y1)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, want, rect)
    lo, hi = max(y0 - halo, 0), min(y1 + halo, h)
    for level, (a, b) in ((0, (lo, hi)), (1, (lo // 2, hi // 2))):
        rows = pyr.level(level)
        rows[:a] = float("nan")
        rows[b:] = float("nan")
    got = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    r.set_tap_window(lo, hi)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, got, rect)
    excess = r.tap_window_excess()
    r.set_tap_window(0, 0)
    assert excess == 0, excess
    # (bit patterns: a poisoned row reaching a pixel would turn a finite value into NaN; the handful of pixels at the frame's
    #  left edge whose refracted ray leaves the clip volume are NaN in both, as in the reference: clip.w <= 0 is undefined)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.isfinite(want[y0:y1].float()).float().mean() > 0.999


def test_tap_window_reports_the_excursion(renderer):
    """The synthetic scene's thick volumes throw taps far from their pixels: a 16-row halo is reported as too small (and by
    how much); with the whole frame as the window nothing is reported."""
    from transmission_renderer_amd.renderer import GBufferPlanes
    r = renderer
    w, h = 640, 360
    scene = _scene(r, w, h, 1.0)
    g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
    pyr = _pyramid(r, w, h, seed=5)
    r.generate_mips(pyr)
    hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    _, y0, y1 = sharded.band_rows(h, 3, 1)
    r.set_tap_window(y0 - 16, y1 + 16)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr, (0, y0, w, y1))
    word = r.tap_window_excess_word()            # (the same word, left on the device: record_sharded(confirm="late") reduces it there)
    excess = r.tap_window_excess()
    assert 4 < excess < h, excess
    assert word.dtype == torch.int64 and word.is_cuda and int(word.item()) == excess
    r.set_tap_window(0, h)
    r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, hdr, (0, y0, w, y1))
    assert r.tap_window_excess() == 0
    # the compositor's verdict on the first launch's word: a fallback, the halo grows to what was needed + 25 %; the probe of a
    # halo that has been wide enough: three quarters of it
    comp = sharded.Compositor.__new__(sharded.Compositor)
    comp.halo_rows = 16
    assert not comp.halo_verdict(excess, 16, 16) and comp.halo_rows == int((16 + excess) * 1.25) + 4 and comp.halo_fallbacks == 1
    r.set_tap_window(0, 0)


def test_exchange_halo_one_rank_communicator_and_rgb8(renderer):
    r = renderer
    lib = r.lib
    ident = (C.c_uint8 * 128)()
    assert lib.tr_comm_unique_id(C.byref(ident)) == 0
    comm = C.c_void_p()
    assert lib.tr_comm_create(r._ctx, C.byref(ident), 1, 0, C.byref(comm)) == 0
    try:
        level = torch.rand((48, 64, 4), device=r.device).to(torch.float16)
        keep = level.clone()
        st = lib.tr_exchange_halo(r._ctx, comm, level.data_ptr(), 64 * 8, 48, 48, 8, torch.cuda.current_stream().cuda_stream)
        assert st == 0
        torch.cuda.synchronize()
        assert torch.equal(level, keep)            # one rank: nothing to exchange, nothing touched
    finally:
        lib.tr_comm_destroy(comm)
    hdr = (torch.rand((36, 64, 4), device=r.device) * 6.0).to(torch.float16)
    hdr[..., 3] = 1.0
    rgba = r.tonemap(hdr)
    rgb = r.tonemap_rgb8(hdr)
    bgr = r.tonemap_rgb8(hdr, bgra=True)
    torch.cuda.synchronize()
    assert torch.equal(rgb, rgba[..., :3]) and (rgba[..., 3] == 255).all()
    assert torch.equal(bgr, rgb.flip(-1))
