"""tr_record_frame_timed (run with -m gpu): a GPU timestamp pair around every pass of a frame, under the zone names
of the reference's `record()` (src/main.rs:1643-2227, src/profiling.rs:134-236); the timed frame is the same frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

REFERENCE_ZONES = ["all commands", "frustum culling", "demultiplex draws compute shader", "depth pre pass", "main opaque",
                   "opaque framebuffer mipchain", "opaque transmissive objects", "tonemapping"]


def test_timed_frame_reports_the_reference_zones_and_the_same_pixels():
    from transmission_renderer_amd import meshes, synthetic, wire
    from transmission_renderer_amd.renderer import TransmissionRenderer
    w, h = 640, 360
    r = TransmissionRenderer(0)
    scene = synthetic.make_scene(w, h, num_point_lights=2, with_gbuffer=False, textured=True)
    r.upload_ggx_lut()
    r.upload_materials(scene["materials"])
    r.upload_textures(scene["textures"])
    r.upload_lights(scene["lights"])
    r.upload_geometry(meshes.make_mesh_scene(extra_instances=True))
    _, view = wire.default_camera()
    aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
    culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
    q = wire.view_rotation_inverse(view)
    work = r.new_frame_buffers(w, h)
    hdr, ldr = r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work)
    torch.cuda.synchronize()
    want_hdr, want_ldr = hdr.clone(), ldr.clone()
    hdr.zero_()
    hdr2, ldr2, zones = r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work, timed=True)
    assert torch.equal(hdr2, want_hdr) and torch.equal(ldr2, want_ldr)          # measuring does not change the frame
    assert list(zones)[0] == "all commands"
    for name in REFERENCE_ZONES:
        assert name in zones and zones[name] > 0.0, (name, zones)
    assert "assign lights to clusters" in zones
    inner = sum(v for k, v in zones.items() if k != "all commands")
    assert inner <= zones["all commands"] * 1.02 and zones["all commands"] < 50.0, zones
    # without a tonemap target there is no "tonemapping" zone
    _, none, zones2 = r.record_frame(scene["uniforms"], scene["push"], culling, view, q, aabbs, work, tonemap=False, timed=True)
    assert none is None and "tonemapping" not in zones2 and "main opaque" in zones2
    r.close()
