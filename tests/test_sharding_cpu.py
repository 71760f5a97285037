"""Row-band sharding on CPU: band arithmetic, the two all-gathers over gloo with world_size 2,
and the sharded pipeline order (opaque band -> gather level 0 -> mips -> transmissive band -> composite) driven
through the oracle so the exchange logic is checked end to end without a GPU."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from transmission_renderer_amd import sharded, synthetic, wire


def test_band_rows():
    # (rows_per_rank, y0, y1): ceil(H / N) rounded up to the 4-row wave tile, clipped to the frame
    assert [sharded.band_rows(2160, 8, r) for r in (0, 6, 7)] == [(272, 0, 272), (272, 1632, 1904), (272, 1904, 2160)]
    assert sharded.band_rows(2160, 4, 1) == (540, 540, 1080) and sharded.band_rows(2160, 1, 0) == (2160, 0, 2160)
    assert sharded.band_rect(3840, 2160, 4, 1) == (0, 540, 3840, 1080)
    assert sharded.padded_rows(2160, 8) == 2176 and sharded.padded_rows(2160, 2) == 2160
    for h in (1, 3, 42, 130, 1080, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            bands = [sharded.band_rows(h, n, r) for r in range(n)]
            rows = bands[0][0]
            assert rows % 4 == 0 and rows * n >= h and all(b[0] == rows for b in bands)
            assert bands[0][1] == 0 and bands[-1][2] == h                       # the bands tile the frame exactly
            assert all(bands[r][2] == bands[r + 1][1] for r in range(n - 1))
            assert all(b[1] == min(r * rows, h) for r, b in enumerate(bands))  # band r starts at its gather slot
    with pytest.raises(ValueError):
        sharded.band_rows(100, 4, 4)


def test_strips_of_rank():
    # strip s of the frame belongs to rank s % world; the last strip of the frame may be short
    assert sharded.strips_of_rank(2160, 64, 8, 0) == [(0, 64), (512, 576), (1024, 1088), (1536, 1600), (2048, 2112)]
    assert sharded.strips_of_rank(2160, 64, 8, 1)[-1] == (2112, 2160)             # 2160 = 33 * 64 + 48
    assert sharded.strips_of_rank(2160, 64, 8, 2)[-1] == (1664, 1728) and len(sharded.strips_of_rank(2160, 64, 8, 7)) == 4
    for h in (1, 5, 42, 130, 1080, 2160):
        for rows in (4, 16, 64):
            for n in (1, 2, 3, 8):
                got = sorted(s for r in range(n) for s in sharded.strips_of_rank(h, rows, n, r))
                assert got[0][0] == 0 and got[-1][1] == h and all(a[1] == b[0] for a, b in zip(got, got[1:]))
                assert all(y1 - y0 == rows for y0, y1 in got[:-1])
    with pytest.raises(ValueError):
        sharded.strips_of_rank(100, 16, 4, 4)


def test_synthetic_scene_is_deterministic_and_band_sliceable():
    a = synthetic.make_gbuffer(160, 90)
    b = synthetic.make_gbuffer(160, 90)
    for k in ("pos_depth", "nrm_scale", "uv", "material_id"):
        np.testing.assert_array_equal(a[k], b[k])
    band = synthetic.make_gbuffer(160, 90, rows=(30, 60))
    for k in ("pos_depth", "nrm_scale", "uv", "material_id"):
        np.testing.assert_array_equal(band[k], a[k][30:60])
    mats = synthetic.make_materials()
    digest = hashlib.sha256(b"".join(bytes(m) for m in mats)).hexdigest()
    assert digest == hashlib.sha256(b"".join(bytes(m) for m in synthetic.make_materials())).hexdigest()
    iors = [m.index_of_refraction for m in mats]
    assert iors[0] == 1.5 and iors[1] == 1.0 and any(np.isinf(m.attenuation_distance) for m in mats)
    assert len({int(x) for x in np.unique(a["material_id"])}) > 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OraclePyramid:
    """The pyramid as record_sharded sees it (level0_padded, level(k)) over the oracle's packed texel array.  Level 0 has
    its own padded storage (the gathers want equal bands); the packed array's copy of it is refreshed before a chain is
    built or sampled."""

    def __init__(self, w, h, rows):
        from oracle import oracle
        self.w, self.h = w, h
        self.level0 = torch.zeros((rows, w, 4), dtype=torch.float16)
        self.tex = oracle.new_pyramid(w, h, np.full((h, w, 4), 77.0, np.float16))      # (a sentinel: rows nobody delivered)
        self.levels, layout, _ = wire.pyramid_layout(w, h)
        self.offsets = [t[0] for t in layout]
        self._tex_t = torch.from_numpy(self.tex)                                        # shares memory

    def level0_padded(self):
        return self.level0

    def level(self, k):
        if k == 0:
            return self.level0[:self.h]
        wk, hk = max(self.w >> k, 1), max(self.h >> k, 1)
        off = int(self.offsets[k])
        return self._tex_t[off:off + wk * hk].view(hk, wk, 4)

    def sync_level0(self):
        self.tex[:self.w * self.h] = self.level0[:self.h].numpy().reshape(-1, 4)


class _OracleRenderer:
    """Stands in for TransmissionRenderer in sharded.record_sharded: the same three calls, answered by the CPU
    oracle on host tensors, so the REAL recorder (band arithmetic, exchange order, padded buffers) runs in the test."""

    def __init__(self, binding, materials=None):
        self.binding = binding
        self.materials = materials
        self.calls = []
        self.strips = None

    def set_strips(self, strip_rows, world, rank):
        self.strips = (strip_rows, world, rank) if strip_rows and world > 1 else None

    def _rects(self, rect):
        """What a whole-frame call shades: the rect, or with tr_set_strips in force this rank's strips of it."""
        if self.strips is None:
            return [rect]
        assert rect[1] == 0
        return [(rect[0], y0, rect[2], y1) for y0, y1 in sharded.strips_of_rank(rect[3], *self.strips)]

    def shade_opaque(self, g, uniforms, push, hdr, pyramid, rect):
        if self.strips is not None:
            self.calls.append(("opaque", rect))
            for r in self._rects(rect):
                self._shade_opaque(g, hdr, pyramid, r)
            return
        self._shade_opaque(g, hdr, pyramid, rect, log=True)

    def _shade_opaque(self, g, hdr, pyramid, rect, log=False):
        from oracle import oracle
        if log:
            self.calls.append(("opaque", rect))
        hdr16, _, mip0 = oracle.shade_opaque(self.binding, g, rect=rect)
        y0, y1 = rect[1], rect[3]
        hdr[y0:y1] = torch.from_numpy(hdr16[y0:y1])
        pyramid.level0[y0:y1] = torch.from_numpy(mip0[y0:y1])

    def generate_mips(self, pyramid):
        from oracle import oracle
        self.calls.append(("mips", None))
        pyramid.sync_level0()
        oracle.generate_mips(pyramid.w, pyramid.h, pyramid.tex)

    def generate_mips_band(self, pyramid, y0, y1):
        """tr_generate_mips_band: levels 1 and 2 of the band = levels 1 and 2 of the band's rows taken as an image of their
        own (sizes multiples of 4: exact 2x2 boxes)."""
        from oracle import oracle
        self.calls.append(("mips_band", (y0, y1)))
        assert pyramid.w % 4 == 0 and pyramid.h % 4 == 0 and y0 % 4 == 0 and (y1 % 4 == 0 or y1 == pyramid.h)
        if y1 == y0:
            return
        sub = oracle.new_pyramid(pyramid.w, y1 - y0, pyramid.level0[y0:y1].numpy())
        oracle.generate_mips(pyramid.w, y1 - y0, sub)
        offs = [t[0] for t in wire.pyramid_layout(pyramid.w, y1 - y0)[1]]
        for k in (1, 2):
            wk, hk = pyramid.w >> k, (y1 - y0) >> k
            pyramid.level(k)[y0 >> k:(y0 >> k) + hk] = torch.from_numpy(sub[int(offs[k]):int(offs[k]) + wk * hk].reshape(hk, wk, 4))

    def generate_mips_from(self, pyramid, first):
        """tr_generate_mips_from(3): levels 3.. = levels 1.. of the chain whose level 0 is level 2."""
        from oracle import oracle
        self.calls.append(("mips_from", first))
        assert first == 3
        w2, h2 = pyramid.w >> 2, pyramid.h >> 2
        sub = oracle.new_pyramid(w2, h2, pyramid.level(2).numpy())
        oracle.generate_mips(w2, h2, sub)
        levels, layout, _ = wire.pyramid_layout(w2, h2)
        offs = [t[0] for t in layout]
        assert levels == pyramid.levels - 2
        for k in range(1, levels):
            wk, hk = max(w2 >> k, 1), max(h2 >> k, 1)
            pyramid.level(k + 2)[:] = torch.from_numpy(sub[int(offs[k]):int(offs[k]) + wk * hk].reshape(hk, wk, 4))

    window = None
    excess = 0

    def set_tap_window(self, lo=0, hi=0):
        self.window = (lo, hi) if hi else None
        if hi:
            self.excess = 0

    def tap_window_excess(self):
        return self.excess

    device = "cpu"

    def tap_window_excess_word(self):
        return torch.tensor([self.excess], dtype=torch.int64)

    def _window_excess(self, g, push, pyramid, rect):
        """What the kernel's tap_window_excess reports, restated in numpy (float64): 1 + the level-0 rows by which the
        bilinear rows of a tap of level 0 / 1 lie outside the window."""
        lo, hi = self.window
        mats = self.materials
        w, h = pyramid.w, pyramid.h
        oy = g["origin_y"]
        ys = slice(rect[1] - oy, rect[3] - oy)
        pos = g["pos_depth"][ys, :, :3].astype(np.float64)
        n = g["nrm_scale"][ys, :, :3].astype(np.float64)
        scale = g["nrm_scale"][ys, :, 3].astype(np.float64)
        mid = g["material_id"][ys]
        ok = mid != wire.NOT_COVERED
        mid = np.where(ok, mid, 0)
        arr = lambda f: np.array([f(m) for m in mats], dtype=np.float64)[mid]       # noqa: E731
        ior, thick, tf, rough = (arr(lambda m: m.index_of_refraction), arr(lambda m: m.thickness_factor),
                                 arr(lambda m: m.transmission_factor), arr(lambda m: m.roughness_factor))
        eye = np.array(push.view_position[:3], dtype=np.float64)
        P = np.array(push.proj_view, dtype=np.float64).reshape(4, 4).T
        v = eye - pos
        v /= np.linalg.norm(v, axis=-1, keepdims=True)
        n = n / np.linalg.norm(n, axis=-1, keepdims=True)
        eta, nov = 1.0 / ior, (n * v).sum(-1)
        cn = -eta * nov + np.sqrt(np.maximum(1.0 - eta ** 2 * (1.0 - nov ** 2), 0.0))
        ex = pos + (-eta[..., None] * v - cn[..., None] * n) * (thick * scale)[..., None]
        c = np.concatenate([ex, np.ones(ex.shape[:-1] + (1,))], -1) @ P.T
        tv = (c[..., 1] / c[..., 3] + 1.0) / 2.0
        lod = np.clip(np.log2(np.float32(w)) * rough * np.clip(2.0 * ior - 2.0, 0.0, 1.0), 0.0, pyramid.levels - 1)
        l0 = np.floor(lod).astype(np.int64)
        worst = 0.0
        for k in (0, 1):                                                   # the lower and the upper level of the pair
            level = np.minimum(l0 + k, pyramid.levels - 1)
            for lv in (0, 1):
                sel = ok & (tf != 0.0) & (level == lv)
                if not sel.any():
                    continue
                hk = h >> lv
                t = np.clip(tv[sel] * hk - 0.5, 0.0, hk - 1.0)
                b = np.minimum(np.floor(t), max(hk - 2, 0))
                first, last = lo >> lv, (hi >> lv) - 1
                e = np.maximum(np.maximum(first - b, (b + 1.0) - last), 0.0) * (1 << lv)
                worst = max(worst, float(e.max()))
        return int(worst) + 1 if worst > 0.0 else 0

    def shade_transmission(self, g, uniforms, push, pyramid, hdr, rect):
        from oracle import oracle
        self.calls.append(("transmission", rect))
        if self.window is not None:
            self.excess = max(self.excess, self._window_excess(g, push, pyramid, rect))
        pyramid.sync_level0()
        frame = hdr[:pyramid.h].numpy()                                   # shares memory: shaded in place (LOAD)
        for r in self._rects(rect):
            oracle.shade_transmission(self.binding, g, pyramid.tex, hdr_f16=frame, rect=r)


def _worker(rank, world, port, w, h, out_dir, strip_rows=0, halo=0, thickness_scale=1.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from transmission_renderer_amd.png import read_png_rgba8
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lut = read_png_rgba8(os.path.join(root, "transmission_renderer_amd", "assets", "ggx_lut.png"))
        scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
        for m in scene["materials"]:
            m.thickness_factor *= thickness_scale
        binding = oracle.SceneBinding(scene, lut)
        if halo:                                                          # the halo exchange instead of the level-0 gather
            rows, y0, y1 = sharded.band_rows(h, world, rank)
            band = synthetic.make_gbuffer(w, h, rows=(y0, y1))
            hdr = torch.zeros((rows * world, w, 4), dtype=torch.float16)
            pyr = _OraclePyramid(w, h, rows * world)
            fake = _OracleRenderer(binding, scene["materials"])
            comp = sharded.Compositor(world, rank)
            comp.halo_rows = halo
            sharded.record_sharded(fake, band, band, scene["uniforms"], scene["push"], hdr, pyr, comp, exchange="halo")
            names = [c[0] for c in fake.calls]
            np.save(os.path.join(out_dir, f"frame_{rank}.npy"), hdr[:h].numpy())
            with open(os.path.join(out_dir, f"calls_{rank}.txt"), "w") as f:
                f.write(" ".join(names) + f"\n{comp.halo_fallbacks} {comp.halo_rows}\n")
            return
        if strip_rows:                                                    # rank-interleaved strips of whole-frame buffers
            g = synthetic.make_gbuffer(w, h)
            hdr = torch.full((h, w, 4), -7.0, dtype=torch.float16)        # (a sentinel no pass writes)
            pyr = _OraclePyramid(w, h, h)
            fake = _OracleRenderer(binding)
            comp = sharded.Compositor(world, rank)
            sharded.record_sharded_strips(fake, g, g, scene["uniforms"], scene["push"], hdr, pyr, comp, strip_rows=strip_rows,
                                          composite=True)
            assert fake.strips is None                                    # switched off again
            mine = sharded.strips_of_rank(h, strip_rows, world, rank)
            assert [c[0] for c in fake.calls] == (["opaque", "mips", "transmission"] if mine else ["mips"])
            np.save(os.path.join(out_dir, f"frame_{rank}.npy"), hdr.numpy())
            np.save(os.path.join(out_dir, f"mip0_{rank}.npy"), pyr.level0[:h].numpy())
            return
        rows, y0, y1 = sharded.band_rows(h, world, rank)
        band = synthetic.make_gbuffer(w, h, rows=(y0, y1))            # this rank's tile only
        hdr = torch.zeros((rows * world, w, 4), dtype=torch.float16)  # padded: equal bands for the gathers
        pyr = _OraclePyramid(w, h, rows * world)
        fake = _OracleRenderer(binding)
        comp = sharded.Compositor(world, rank)                        # torch.distributed (gloo) on host tensors
        assert comp.backend == "torch.distributed:gloo"
        sharded.record_sharded(fake, band, band, scene["uniforms"], scene["push"], hdr, pyr, comp)
        assert [c[0] for c in fake.calls] == ["opaque", "mips", "transmission"]
        assert fake.calls[0][1] == (0, y0, w, y1) == fake.calls[2][1]
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), hdr[:h].numpy())
        np.save(os.path.join(out_dir, f"mip0_{rank}.npy"), pyr.level0[:h].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("w,h,world", [(64, 40, 2), (48, 42, 2), (40, 30, 3)])
def test_record_sharded_over_gloo_matches_single_rank(tmp_path, ggx_lut, w, h, world):
    """sharded.record_sharded ITSELF on `world` gloo ranks (oracle-backed renderer): every rank ends with the
    single-rank frame, bit for bit; heights that do not divide into 4-row-aligned bands use the padded buffers."""
    from oracle import oracle
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame_{r}.npy") for r in range(world)]
    for f in frames[1:]:
        np.testing.assert_array_equal(frames[0].view(np.uint16), f.view(np.uint16))      # every rank has the frame

    scene = synthetic.make_scene(w, h, num_point_lights=2)
    binding = oracle.SceneBinding(scene, ggx_lut)
    hdr16, _, mip0 = oracle.shade_opaque(binding, scene["gbuffer"])
    np.testing.assert_array_equal(np.load(tmp_path / "mip0_0.npy").view(np.uint16), mip0.view(np.uint16))
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(binding, scene["gbuffer"], tex, hdr_f16=hdr16)
    np.testing.assert_array_equal(frames[0].view(np.uint16), hdr16.view(np.uint16))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("w,h,world,strip_rows", [(64, 40, 2, 8), (48, 42, 3, 4), (40, 10, 3, 8)])
def test_record_sharded_strips_over_gloo_matches_single_rank(tmp_path, ggx_lut, w, h, world, strip_rows):
    """sharded.record_sharded_strips on `world` gloo ranks: strip s shaded by rank s % world in place, level 0 and the
    frame exchanged strip by strip; every rank ends with the single-rank frame bit for bit (incl. a short last strip and
    a rank with no strip at all)."""
    from oracle import oracle
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path), strip_rows), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame_{r}.npy") for r in range(world)]
    for f in frames[1:]:
        np.testing.assert_array_equal(frames[0].view(np.uint16), f.view(np.uint16))
    scene = synthetic.make_scene(w, h, num_point_lights=2)
    binding = oracle.SceneBinding(scene, ggx_lut)
    hdr16, _, mip0 = oracle.shade_opaque(binding, scene["gbuffer"])
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"mip0_{r}.npy").view(np.uint16), mip0.view(np.uint16))
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(binding, scene["gbuffer"], tex, hdr_f16=hdr16)
    np.testing.assert_array_equal(frames[0].view(np.uint16), hdr16.view(np.uint16))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("w,h,world,halo,thickness_scale,fallback", [
    (64, 48, 2, 8, 0.02, False),      # thin volumes: the taps stay within 8 rows of their pixel
    (48, 72, 3, 12, 0.02, False),     # three bands of 24 rows
    (64, 48, 2, 4, 1.0, True),        # the synthetic scene's thick volumes throw taps across the frame: the frame is redone
])
def test_record_sharded_halo_exchange_over_gloo_matches_single_rank(tmp_path, ggx_lut, w, h, world, halo, thickness_scale, fallback):
    """record_sharded(exchange="halo") on gloo ranks (oracle-backed renderer): levels 1 / 2 per band, border rows of levels
    0 / 1 to the neighbours, level 2 to everyone, the chain from level 3 on replicated — every rank ends with the
    single-rank frame bit for bit; a frame whose taps leave the halo is detected, redone with the full gather, and the
    compositor's halo grows."""
    from oracle import oracle
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path), 0, halo, thickness_scale), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame_{r}.npy") for r in range(world)]
    for f in frames[1:]:
        np.testing.assert_array_equal(frames[0].view(np.uint16), f.view(np.uint16))
    scene = synthetic.make_scene(w, h, num_point_lights=2)
    for m in scene["materials"]:
        m.thickness_factor *= thickness_scale
    binding = oracle.SceneBinding(scene, ggx_lut)
    hdr16, _, mip0 = oracle.shade_opaque(binding, scene["gbuffer"])
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(binding, scene["gbuffer"], tex, hdr_f16=hdr16)
    np.testing.assert_array_equal(frames[0].view(np.uint16), hdr16.view(np.uint16))
    for r in range(world):
        calls, state = open(tmp_path / f"calls_{r}.txt").read().splitlines()
        fallbacks, halo_now = (int(x) for x in state.split())
        if fallback:
            assert calls.split() == ["opaque", "mips_band", "mips_from", "transmission", "mips", "transmission"]
            assert fallbacks == 1 and halo_now > halo
        else:
            assert calls.split() == ["opaque", "mips_band", "mips_from", "transmission"] and fallbacks == 0 and halo_now == halo


def _worker_late(rank, world, port, w, h, out_dir, halo, thickness_scale):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from transmission_renderer_amd.png import read_png_rgba8
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lut = read_png_rgba8(os.path.join(root, "transmission_renderer_amd", "assets", "ggx_lut.png"))
        scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
        for m in scene["materials"]:
            m.thickness_factor *= thickness_scale
        binding = oracle.SceneBinding(scene, lut)
        rows, y0, y1 = sharded.band_rows(h, world, rank)
        band = synthetic.make_gbuffer(w, h, rows=(y0, y1))
        comp = sharded.Compositor(world, rank)
        comp.halo_rows = halo
        log = []
        for frame in range(2):
            hdr = torch.zeros((rows * world, w, 4), dtype=torch.float16)
            pyr = _OraclePyramid(w, h, rows * world)
            fake = _OracleRenderer(binding, scene["materials"])
            sharded.record_sharded(fake, band, band, scene["uniforms"], scene["push"], hdr, pyr, comp, exchange="halo", confirm="late")
            log.append(f"{' '.join(c[0] for c in fake.calls)}|{comp.halo_fallbacks} {comp.halo_inexact_frames} {comp.halo_rows}")
        last_ok = comp.confirm_halo()
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), hdr[:h].numpy())
        with open(os.path.join(out_dir, f"late_{rank}.txt"), "w") as f:
            f.write("\n".join(log) + f"\n{int(last_ok)} {comp.halo_fallbacks} {comp.halo_inexact_frames}\n")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_record_sharded_halo_confirmed_a_frame_late_over_gloo(tmp_path, ggx_lut):
    """record_sharded(exchange="halo", confirm="late"): the excess word is all-reduced behind the passes and read when the
    next frame starts — no drain between a frame's passes and its composite.  Frame 1 (a 4-row halo under volumes that
    throw taps across the frame) is composited as it is and found inexact when frame 2 starts, which then exchanges more
    rows (here: everything, the gather) and is the single-rank frame bit for bit on every rank."""
    from oracle import oracle
    w, h, world, halo = 64, 48, 2, 4
    mp.spawn(_worker_late, args=(world, _free_port(), w, h, str(tmp_path), halo, 1.0), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame_{r}.npy") for r in range(world)]
    scene = synthetic.make_scene(w, h, num_point_lights=2)
    binding = oracle.SceneBinding(scene, ggx_lut)
    hdr16, _, mip0 = oracle.shade_opaque(binding, scene["gbuffer"])
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(binding, scene["gbuffer"], tex, hdr_f16=hdr16)
    for r in range(world):
        np.testing.assert_array_equal(frames[r].view(np.uint16), hdr16.view(np.uint16))
        first, second, end = open(tmp_path / f"late_{r}.txt").read().splitlines()
        calls, state = first.split("|")
        assert calls.split() == ["opaque", "mips_band", "mips_from", "transmission"]      # nothing redone, nothing read
        assert [int(x) for x in state.split()] == [0, 0, halo]
        calls, state = second.split("|")
        fallbacks, inexact, halo_now = (int(x) for x in state.split())
        assert fallbacks == 1 and inexact == 1 and halo_now > halo
        assert [int(x) for x in end.split()] == [1, 1, 1]


def test_halo_policy_grows_at_once_and_shrinks_by_probing():
    """Compositor's halo book-keeping without any rank: a tap that leaves the halo widens it at once; after
    halo_shrink_after clean frames one frame checks its taps against three quarters of the halo — the exchange shrinks when
    they fit, stays when they do not, and a probe frame whose taps left the whole halo is a fallback like any other."""
    comp = sharded.Compositor.__new__(sharded.Compositor)
    comp.halo_rows, comp.halo_shrink_after = 64, 3
    assert comp.halo_window_margin(64) == 64
    assert comp.halo_verdict(0, 64, 64) and comp.halo_clean_frames == 1
    assert not comp.halo_verdict(11, 64, 64)                       # the worst tap missed the window by 10 rows
    assert comp.halo_fallbacks == 1 and comp.halo_rows == int((64 + 11) * 1.25) + 4 and comp.halo_clean_frames == 0
    halo = 96
    comp.halo_rows = halo
    for _ in range(3):
        assert comp.halo_window_margin(halo) == halo and comp.halo_verdict(0, halo, halo)
    assert comp.halo_window_margin(halo) == 72                     # the probe: three quarters, a multiple of 4
    assert comp.halo_verdict(9, halo, 72) and comp.halo_rows == halo and comp.halo_clean_frames == 0   # 8 rows beyond 72: inside 96, no shrink
    for _ in range(3):
        assert comp.halo_verdict(0, halo, halo)
    assert comp.halo_verdict(0, halo, comp.halo_window_margin(halo)) and comp.halo_rows == 72            # they fit: the halo shrinks
    for _ in range(3):
        assert comp.halo_verdict(0, 72, 72)
    assert comp.halo_window_margin(72) == 52
    assert not comp.halo_verdict(40, 72, 52) and comp.halo_fallbacks == 2 and comp.halo_rows == int((52 + 40) * 1.25) + 4   # 39 rows beyond 52 > 72
    comp.halo_shrink_after = 0
    comp.halo_clean_frames = 100
    assert comp.halo_window_margin(64) == 64                       # (0: never probes)


def test_halo_rows_between():
    # reader 3 of 8 bands of 272 rows (4K) with a 64-row halo: the last 64 rows of band 2, the first 64 of band 4
    got = [sharded.halo_rows_between(2160, 272, 8, o, 3, 64) for o in range(8)]
    assert got == [(0, 0), (0, 0), (752, 816), (816, 1088), (1088, 1152), (0, 0), (0, 0), (0, 0)]
    # a halo larger than a band reaches into the second neighbour; the frame's ends clip it
    assert sharded.halo_rows_between(2160, 272, 8, 1, 3, 300) == (516, 544) and sharded.halo_rows_between(2160, 272, 8, 7, 6, 300) == (1904, 2160)
    # halo >= the level: every band to every reader (the all-gather)
    for o in range(3):
        assert sharded.halo_rows_between(40, 16, 3, o, 1, 40) == (o * 16, min((o + 1) * 16, 40))
    # bytes per link, 8K frame on 8 ranks, RGBA16F: level 0 + level 1 border rows + a band of level 2 vs a band of level 0
    halo, rows, w = 128, 540, 7680
    nb = (halo * w + (halo // 2) * (w // 2)) * 8 + (rows // 4) * (w // 4) * 8
    assert nb < 0.37 * rows * w * 8      # 11.9 MB to each neighbour (+ 2.1 MB of level 2 to the others) instead of 33.2 MB to all seven
