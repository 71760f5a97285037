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
    assert [sharded.band_rows(2160, 8, r) for r in (0, 7)] == [(0, 270), (1890, 2160)]
    assert sharded.band_rect(3840, 2160, 4, 1) == (0, 540, 3840, 1080)
    with pytest.raises(ValueError):
        sharded.band_rows(2161, 8, 0)


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


def _worker(rank, world, port, w, h, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from transmission_renderer_amd.png import read_png_rgba8
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lut = read_png_rgba8(os.path.join(root, "transmission_renderer_amd", "assets", "ggx_lut.png"))
        scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False)
        binding = oracle.SceneBinding(scene, lut)
        y0, y1 = sharded.band_rows(h, world, rank)
        band = synthetic.make_gbuffer(w, h, rows=(y0, y1))            # this rank's tile only

        # "main opaque" on the band; both attachments whole-frame, only the band written
        hdr16, _, mip0 = oracle.shade_opaque(binding, band)
        assert (mip0[:y0] == 0).all() and (mip0[y1:] == 0).all()
        mip0_t = torch.from_numpy(mip0)
        sharded.allgather_mip0(mip0_t, world)                          # exchange 1
        tex = oracle.new_pyramid(w, h, mip0_t.numpy())
        oracle.generate_mips(w, h, tex)                                # replicated
        oracle.shade_transmission(binding, band, tex, hdr_f16=hdr16)   # band, LOAD semantics
        frame = torch.from_numpy(hdr16)
        sharded.allgather_frame(frame, world)                          # exchange 2 (composite)
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_pipeline_matches_single_rank(tmp_path, ggx_lut):
    from oracle import oracle
    w, h, world = 64, 40, 2
    mp.spawn(_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    frames = [np.load(tmp_path / f"frame_{r}.npy") for r in range(world)]
    np.testing.assert_array_equal(frames[0].view(np.uint16), frames[1].view(np.uint16))  # every rank has the frame

    scene = synthetic.make_scene(w, h, num_point_lights=2)
    binding = oracle.SceneBinding(scene, ggx_lut)
    hdr16, _, mip0 = oracle.shade_opaque(binding, scene["gbuffer"])
    tex = oracle.new_pyramid(w, h, mip0)
    oracle.generate_mips(w, h, tex)
    oracle.shade_transmission(binding, scene["gbuffer"], tex, hdr_f16=hdr16)
    np.testing.assert_array_equal(frames[0].view(np.uint16), hdr16.view(np.uint16))
