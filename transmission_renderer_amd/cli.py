"""glTF-in/frame-out command line of the reference, for the part of the pipeline this repository covers.

The reference's CLI (`Opt`, src/main.rs:65-91) is
    transmission-renderer [--scale f] [--roughness-override f] [--spotlights] ... <gltf_sample_model_name>
and renders to a window.  Here the frame goes to a file and the scene source is a synthetic TGB-v1 G-buffer
(`synthetic`): glTF import + rasterisation are SURVEY.md §8f row f3 and not built yet, so any other model name is
refused rather than approximated.  Everything downstream of the G-buffer is the real path:
cluster build -> main opaque -> mip chain -> transmissive pass -> tonemap -> PNG.

    python -m transmission_renderer_amd.cli synthetic --width 1920 --height 1080 --lights 2 --out frame.png
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="transmission_renderer_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("gltf_sample_model_name", help="'synthetic' (glTF import is not built yet)")
    ap.add_argument("-s", "--scale", type=float, default=1.0, help="model scale (flat model_scale of every fragment)")
    ap.add_argument("--roughness-override", type=float, default=None)
    ap.add_argument("--spotlights", action="store_true", help="add the reference's two spotlights (src/main.rs:455-476)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=2, help="point lights (the reference hard-codes 2)")
    ap.add_argument("--out", default="frame.png", help="tonemapped 8-bit sRGB PNG")
    ap.add_argument("--hdr-out", default=None, help="also save the RGBA16F HDR frame as .npy")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    if args.gltf_sample_model_name != "synthetic":
        print("only the 'synthetic' scene is available: glTF import / rasterisation (SURVEY.md 8f row f3) is not built",
              file=sys.stderr)
        return 2

    import torch
    from . import synthetic, wire
    from .png import write_png
    from .renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer

    w, h = args.width, args.height
    r = TransmissionRenderer(args.device)
    scene = synthetic.make_scene(w, h, num_point_lights=args.lights, roughness_override=args.roughness_override)
    if args.spotlights:
        scene["lights"] = scene["lights"] + wire.default_lights(spotlights=True)[2:]
    if args.scale != 1.0:
        scene["gbuffer"]["nrm_scale"][..., 3] *= np.float32(args.scale)
        for m in scene["materials"]:   # src/model_loading.rs:317: attenuation distance is pre-multiplied by the scale
            m.attenuation_distance = m.attenuation_distance * args.scale
    r.upload_ggx_lut()
    r.upload_materials(scene["materials"])
    r.upload_lights(scene["lights"])
    _, view = wire.default_camera()
    t0 = time.perf_counter()
    aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
    r.assign_lights_to_clusters(view, wire.view_rotation_inverse(view), aabbs)
    g = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
    pyr = OpaquePyramid(w, h, r.device)
    hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
    # the synthetic scene has one layer: it is shaded as opaque geometry first (the backdrop the refraction sees),
    # then as the transmissive layer in front of it
    r.record(g, g, scene["uniforms"], scene["push"], hdr, pyr)
    ldr = r.tonemap(hdr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    write_png(args.out, ldr.cpu().numpy())
    if args.hdr_out:
        np.save(args.hdr_out, hdr.cpu().numpy())
    print(f"{w}x{h}, sun + {len(scene['lights'])} lights: clusters + opaque + mips + transmission + tonemap in "
          f"{dt * 1e3:.2f} ms (first call, includes launch overheads) -> {args.out}")
    r.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
