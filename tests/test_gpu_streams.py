"""One context driven from two HIP streams (run with -m gpu): a pass as two row bands on two streams
(bench.py's default step) must equal the whole-frame launch bit for bit — from a COLD context, where the
first band's call builds the lazily digested tables on its stream and the second band's call, on the other stream, must
wait for them (tr_shade.hip: tables_rebuilt / tables_acquire); and again after the tables are rebuilt on the other stream
while the first stream's launches may still read the old ones (tables_before_rebuild)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from transmission_renderer_amd import sharded, synthetic  # noqa: E402


def _context(ggx_lut, scene):
    from transmission_renderer_amd.renderer import TransmissionRenderer
    r = TransmissionRenderer(0)
    r.upload_ggx_lut(ggx_lut)
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    r.set_cluster_tables(torch.from_numpy(scene["cluster_counts"].view(np.int32)).to(r.device),
                         torch.from_numpy(scene["light_indices"].view(np.int32)).to(r.device))
    return r


@pytest.mark.parametrize("w,h", [(640, 360), (1920, 1080)])
def test_two_bands_on_two_streams_from_a_cold_context(ggx_lut, w, h):
    from transmission_renderer_amd.renderer import GBufferPlanes, OpaquePyramid
    if not torch.cuda.is_available():
        pytest.fail("no HIP device: the -m gpu tests must run on the GPU box")
    scene = synthetic.make_scene(w, h, num_point_lights=2)
    mip0 = torch.from_numpy(synthetic.make_opaque_mip0(w, h))

    def inputs(r):
        g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
        pyr = OpaquePyramid(w, h, r.device)
        pyr.level(0).copy_(mip0.to(r.device))
        return g, pyr

    # the reference: one context, one stream, one whole-frame launch
    r0 = _context(ggx_lut, scene)
    g, pyr = inputs(r0)
    r0.generate_mips(pyr)
    want = torch.zeros((h, w, 4), dtype=torch.float16, device=r0.device)
    r0.shade_transmission(g, scene["uniforms"], scene["push"], pyr, want)
    torch.cuda.synchronize()
    want = want.cpu()
    r0.close()

    bands = [(0, a, w, b) for a, b in (sharded.band_rows(h, 2, i)[1:] for i in range(2))]
    for trial in range(3):   # (a lost race does not lose every time)
        r = _context(ggx_lut, scene)       # cold: nothing digested, no level table, no tap records, no cluster x / y tables
        g, pyr = inputs(r)
        r.generate_mips(pyr)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
        for s, band in zip(streams, bands):
            with torch.cuda.stream(s):
                r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, got, band)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want), f"cold context, trial {trial}"

        # new materials uploaded on stream 1 while stream 0's launches are in flight, then both bands again
        other = synthetic.make_scene(w, h, num_point_lights=2, roughness_override=0.4)
        with torch.cuda.stream(streams[0]):
            for _ in range(4):
                r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, got, bands[0])
        with torch.cuda.stream(streams[1]):
            r.upload_materials(other["materials"])
            r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, got, bands[1])
        with torch.cuda.stream(streams[0]):
            r.shade_transmission(g, scene["uniforms"], scene["push"], pyr, got, bands[0])
        torch.cuda.synchronize()
        if trial == 0:
            ref = _context(ggx_lut, other)
            g2, pyr2 = inputs(ref)
            ref.generate_mips(pyr2)
            want2 = torch.zeros((h, w, 4), dtype=torch.float16, device=ref.device)
            ref.shade_transmission(g2, scene["uniforms"], scene["push"], pyr2, want2)
            torch.cuda.synchronize()
            want2 = want2.cpu()
            ref.close()
        assert torch.equal(got.cpu(), want2), f"materials re-uploaded across streams, trial {trial}"
        r.close()
