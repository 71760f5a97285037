"""glTF-in/frame-out command line of the reference, on the MI355X path.

The reference's CLI (`Opt`, src/main.rs:65-91) is
    transmission-renderer [--scale f] [--roughness-override f] [--spotlights] [--rotate-model] [--external-model] <gltf_sample_model_name>
and renders to a window.  Here the frame goes to a file (the last of --frames N).  Scene sources:
    <path>.gltf / <path>.glb   a glTF 2.0 file — what `--external-model` gives the reference (the flag is accepted; a path is
                               recognised without it), placed like src/main.rs:364-368: translated to (0, 2, 0), scaled by
                               --scale; `--backdrop other.glb` puts a second scene behind it the way the reference always
                               loads Sponza (:342-351)
    <Name>                     a bare sample-model name, resolved like path_for_gltf_model (src/model_loading.rs:381-390):
                               <--sample-models-dir>/2.0/<Name>/glTF/<Name>.gltf (the Khronos checkout is not in this image)
    meshes                     the procedural mesh scene (transmission_renderer_amd/meshes.py), same pipeline
    synthetic                  a ready-made TGB-v1 G-buffer (the benchmark's input), no geometry stage
Pipeline (every stage on the GPU through libtr_shade.so):
    [frustum culling -> draw demultiplex -> vertex stage + rasteriser ->] cluster build -> main opaque ->
    mip chain -> transmissive pass -> tonemap -> PNG

    python -m transmission_renderer_amd.cli meshes --width 1920 --height 1080 --out frame.png
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np


def lottes_from_args(r, args):
    """tr_lottes_defaults with the --tonemap-* flags laid over it (saturation / cross-saturation follow the contrast unless
    given, as in the operator's own parameterisation)."""
    import ctypes as C
    from . import wire
    q = wire.LottesParams()
    r._check(r.lib.tr_lottes_defaults(C.byref(q)), "tr_lottes_defaults")
    for field in ("contrast", "shoulder", "hdr_max", "mid_in", "mid_out", "crosstalk"):
        v = getattr(args, "tonemap_" + field)
        if v is not None:
            setattr(q, field, float(v))
    q.saturation = float(args.tonemap_saturation) if args.tonemap_saturation is not None else q.contrast
    q.cross_saturation = float(args.tonemap_cross_saturation) if args.tonemap_cross_saturation is not None else q.contrast * 16.0
    return q


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="transmission_renderer_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("gltf_sample_model_name", help="a .gltf / .glb path, 'meshes' or 'synthetic'")
    ap.add_argument("-s", "--scale", type=float, default=1.0, help="model scale (src/main.rs:364-368)")
    ap.add_argument("--roughness-override", type=float, default=None)
    ap.add_argument("--spotlights", action="store_true", help="add the reference's two spotlights (src/main.rs:455-476); with "
                    "--frames N they turn by 0.01 rad per frame (:1243-1256), rewritten in place by tr_update_lights")
    ap.add_argument("--rotate-model", action="store_true",
                    help="the last instance of the model buffers turns about y by -0.0025 rad per frame (src/main.rs:920-921, "
                         "1258-1282), rewritten in place by tr_update_instances: nothing is re-uploaded or re-sized")
    ap.add_argument("--external-model", action="store_true",
                    help="the positional argument is a path, not a sample-model name (src/main.rs:353-357)")
    ap.add_argument("--sample-models-dir", default=os.environ.get("GLTF_SAMPLE_MODELS", "glTF-Sample-Models"),
                    help="where bare model names are resolved (src/model_loading.rs:381-390)")
    ap.add_argument("--frames", type=int, default=1, help="frames to record back to back (the reference's render loop); the "
                    "last one is written")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=2, help="point lights (the reference hard-codes 2)")
    ap.add_argument("--backdrop", default=None, metavar="GLTF",
                    help="a .gltf / .glb scene loaded FIRST with the identity transform and no roughness override, behind "
                         "the model — what the reference does with Sponza (src/main.rs:342-351); the refraction samples it")
    ap.add_argument("--timings", action="store_true",
                    help="render a second, timed frame and print the GPU time of every pass under the reference's "
                         "profiling zone names (src/main.rs:1643-2227)")
    # The reference bakes colstodian's LottesTonemapperParams::default() (src/main.rs:506-510); that crate is not vendored, so
    # its constants cannot be read off the reference: they are explicit inputs here.  The defaults are the values of the
    # operator's GDC 2016 presentation (tr_lottes_defaults) — ASSUMED, not pinned.
    tm = ap.add_argument_group("tonemap operator (Lottes; defaults assumed, see --help epilog)")
    tm.add_argument("--tonemap-contrast", type=float, default=None, help="default 1.6")
    tm.add_argument("--tonemap-shoulder", type=float, default=None, help="default 0.977")
    tm.add_argument("--tonemap-hdr-max", type=float, default=None, help="default 8.0")
    tm.add_argument("--tonemap-mid-in", type=float, default=None, help="default 0.18")
    tm.add_argument("--tonemap-mid-out", type=float, default=None, help="default 0.267")
    tm.add_argument("--tonemap-crosstalk", type=float, default=None, help="default 4.0")
    tm.add_argument("--tonemap-saturation", type=float, default=None, help="default: the contrast")
    tm.add_argument("--tonemap-cross-saturation", type=float, default=None, help="default: 16 x the contrast")
    ap.add_argument("--out", default="frame.png", help="tonemapped 8-bit sRGB PNG")
    ap.add_argument("--hdr-out", default=None, help="also save the RGBA16F HDR frame as .npy")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    name = args.gltf_sample_model_name
    is_file = args.external_model or name.lower().endswith((".gltf", ".glb"))
    if not is_file and name not in ("synthetic", "meshes"):
        # path_for_gltf_model, src/model_loading.rs:381-390
        resolved = os.path.join(args.sample_models_dir, "2.0", name, "glTF", name + ".gltf")
        if not os.path.exists(resolved):
            print(f"'{name}': give a .gltf / .glb path (--external-model), 'meshes' or 'synthetic', or point --sample-models-dir at a "
                  f"glTF-Sample-Models checkout ({resolved} does not exist)", file=sys.stderr)
            return 2
        name, is_file = resolved, True
    if args.frames < 1:
        print("--frames must be at least 1", file=sys.stderr)
        return 2
    if is_file and not os.path.exists(name):
        print(f"{name}: no such file", file=sys.stderr)
        return 2
    if args.backdrop is not None and (not is_file or not os.path.exists(args.backdrop)):
        print("--backdrop needs an existing .gltf / .glb file and a .gltf / .glb model in front of it", file=sys.stderr)
        return 2

    import torch
    from . import gltf, meshes, synthetic, wire
    from .png import write_png
    from .renderer import GBufferPlanes, OpaquePyramid, TransmissionRenderer

    w, h = args.width, args.height
    r = TransmissionRenderer(args.device)
    scene = synthetic.make_scene(w, h, num_point_lights=args.lights, roughness_override=args.roughness_override,
                                 with_gbuffer=(name == "synthetic"), textured=(name == "meshes"))
    if args.spotlights:
        scene["lights"] = scene["lights"] + wire.default_lights(spotlights=True)[2:]
    geometry = None
    if name == "synthetic":
        if args.scale != 1.0:
            scene["gbuffer"]["nrm_scale"][..., 3] *= np.float32(args.scale)
            for m in scene["materials"]:   # src/model_loading.rs:317: attenuation distance is pre-multiplied by the scale
                m.attenuation_distance = m.attenuation_distance * args.scale
    elif name == "meshes":
        geometry = meshes.make_mesh_scene(extra_instances=True)
        scene["materials"][2].alpha_clipping_cutoff = 0.75
        scene["materials"][7].alpha_clipping_cutoff = 0.6
    else:
        # src/main.rs:342-368: the backdrop scene (Sponza there) first, identity transform, no override; then the
        # model at (0, 2, 0) scaled by --scale into the same model buffers (materials and textures are appended)
        backdrop = gltf.load_gltf(args.backdrop) if args.backdrop else None
        loaded = gltf.load_gltf(name, scene=backdrop,
                                base_transform=meshes.Similarity(np.array([0.0, 2.0, 0.0], np.float32), args.scale),
                                roughness_override=args.roughness_override)
        geometry = loaded.geometry()
        scene["materials"] = loaded.materials or [wire.MaterialInfo.default()]
        scene["textures"] = loaded.textures
    r.upload_ggx_lut()
    r.upload_materials(scene["materials"])
    if scene.get("textures"):
        r.upload_textures(scene["textures"])
    r.upload_lights(scene["lights"])
    if geometry is not None:
        r.upload_geometry(geometry)
    _, view = wire.default_camera()
    lottes = lottes_from_args(r, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    aabbs = r.write_cluster_data(scene["uniforms"], wire.inverse_perspective(w, h), (w, h))
    if geometry is None:
        # the synthetic scene has one layer: it is shaded as opaque geometry first (the backdrop the refraction
        # sees), then as the transmissive layer in front of it
        r.assign_lights_to_clusters(view, wire.view_rotation_inverse(view), aabbs)
        opaque = transmissive = GBufferPlanes.from_numpy(scene["gbuffer"], r.device)
        pyr = OpaquePyramid(w, h, r.device)
        hdr = torch.zeros((h, w, 4), dtype=torch.float16, device=r.device)
        r.record(opaque, transmissive, scene["uniforms"], scene["push"], hdr, pyr)
        ldr = r.tonemap(hdr, r.baked_tonemap_params(lottes))
    else:
        # one native call per frame: culling, light assignment, demultiplex, rasteriser, opaque, mips, transmissive, tonemap
        culling = wire.CullingPushConstants.new(wire.perspective_matrix_reversed(w, h), view)
        work = r.new_frame_buffers(w, h)
        instances = np.ascontiguousarray(geometry["instances"], dtype=wire.INSTANCE_DTYPE).copy()
        model_rotation, spotlight_angle = 0.0, 0.0
        first_spot = len(scene["lights"]) - 2
        for frame in range(args.frames):
            # the reference's per-frame writes into its mapped buffers (src/main.rs:1243-1261, 1316-1322): sub-range updates
            if args.spotlights and frame > 0:
                spotlight_angle += 0.01
                spots = scene["lights"][first_spot:]
                for k, l in enumerate(spots):     # Light::set_spotlight_direction(Quat::from_rotation_y(angle [+ pi]) * Z)
                    a = np.float32(spotlight_angle + (np.pi if k else 0.0))
                    l.spotlight_direction_and_outer_angle[0] = float(np.sin(a))
                    l.spotlight_direction_and_outer_angle[1] = 0.0
                    l.spotlight_direction_and_outer_angle[2] = float(np.cos(a))
                r.update_lights(first_spot, spots)
            if args.rotate_model and frame > 0:
                model_rotation -= 0.0025
                half = np.float32(model_rotation) * np.float32(0.5)
                instances["rotation"][-1] = (0.0, np.sin(half), 0.0, np.cos(half))      # Quat::from_rotation_y
                r.update_instances(len(instances) - 1, instances[-1:])
            hdr, ldr = r.record_frame(scene["uniforms"], scene["push"], culling, view, wire.view_rotation_inverse(view), aabbs, work,
                                      lottes=lottes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if args.timings and geometry is not None:
        for _ in range(20):   # warm clocks
            r.record_frame(scene["uniforms"], scene["push"], culling, view, wire.view_rotation_inverse(view), aabbs, work)
        _, _, zones = r.record_frame(scene["uniforms"], scene["push"], culling, view, wire.view_rotation_inverse(view), aabbs,
                                     work, timed=True)
        for zone, ms in zones.items():
            print(f"  {zone:<36} {ms * 1e3:9.1f} us")
    write_png(args.out, ldr.cpu().numpy())
    if args.hdr_out:
        np.save(args.hdr_out, hdr.cpu().numpy())
    what = "clusters + " + ("" if geometry is None else "culling + rasteriser + ") + "opaque + mips + transmission + tonemap"
    extra = "" if geometry is None else f", {len(geometry['index']) // 3} triangles in {len(geometry['primitives'])} primitives"
    frames = f"{args.frames} frames of " if geometry is not None and args.frames > 1 else ""
    print(f"{w}x{h}, sun + {len(scene['lights'])} lights{extra}: {frames}{what} in {dt * 1e3:.2f} ms (first call, includes "
          f"launch overheads) -> {args.out}")
    r.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
