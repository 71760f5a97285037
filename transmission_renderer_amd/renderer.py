"""Host-side frame recorder over the C ABI: the compute-pipeline replacement of `record()`.

Mirrors the part of src/main.rs:1551-2263 that schedules the hot path —
  "main opaque" (fragment)  ->  "opaque framebuffer mipchain" (generate_mips)
  ->  "opaque transmissive objects" (fragment_transmission) —
with the attachment contract of src/render_passes.rs (RGBA16F colour targets, LOAD on the
transmission pass) and the zone names of src/profiling.rs as timing labels.

torch is plumbing only: it owns device memory (tensors) and the stream; every pixel is computed
by libtr_shade.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, wire
from .png import read_png_rgba8

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_ggx_lut() -> np.ndarray:
    """ggx_lut.png exactly as src/main.rs:295-330 uploads it: (1024, 1024, 4) uint8, row 0 first."""
    cache = os.path.join(ASSET_DIR, "ggx_lut.png")
    return read_png_rgba8(cache)


@dataclass
class GBufferPlanes:
    """TGB-v1 planes on the device (include/tr_shade.h tr_gbuffer)."""
    pos_depth: torch.Tensor    # (H, W, 4) float32
    nrm_scale: torch.Tensor    # (H, W, 4) float32
    uv: torch.Tensor           # (H, W, 2) float32
    material_id: torch.Tensor  # (H, W) int32 (bit pattern of u32; -1 = not covered)
    origin_x: int = 0          # frame position of plane element (0, 0): a rank may hold one tile only
    origin_y: int = 0

    @property
    def height(self) -> int:
        return int(self.pos_depth.shape[0])

    @property
    def width(self) -> int:
        return int(self.pos_depth.shape[1])

    def as_struct(self) -> wire.GBuffer:
        for t in (self.pos_depth, self.nrm_scale, self.uv, self.material_id):
            assert t.is_cuda and t.is_contiguous()
        assert self.pos_depth.dtype == torch.float32 and self.material_id.dtype == torch.int32
        return wire.GBuffer(self.pos_depth.data_ptr(), self.nrm_scale.data_ptr(), self.uv.data_ptr(),
                            self.material_id.data_ptr(), self.width, self.height, self.origin_x, self.origin_y)

    @classmethod
    def from_numpy(cls, g: dict, device) -> "GBufferPlanes":
        return cls(torch.from_numpy(g["pos_depth"]).to(device), torch.from_numpy(g["nrm_scale"]).to(device),
                   torch.from_numpy(g["uv"]).to(device),
                   torch.from_numpy(g["material_id"].view(np.int32)).to(device),
                   int(g.get("origin_x", 0)), int(g.get("origin_y", 0)))


class OpaquePyramid:
    """`opaque_sampled_hdr_framebuffer` (src/main.rs:383-402): RGBA16F, full mip chain, one allocation."""

    def __init__(self, width: int, height: int, device, level0_rows: Optional[int] = None):
        """level0_rows > height reserves extra rows behind level 0 (levels 1.. start later): a row-band sharded
        frame all-gathers level 0 in equal bands of sharded.padded_rows(height, world) rows in all."""
        lib = _lib.load()
        self.desc = wire.Pyramid()
        nbytes = C.c_size_t()
        st = lib.tr_pyramid_layout(width, height, C.byref(self.desc), C.byref(nbytes))
        if st != 0:
            raise _lib.TrError(st, "tr_pyramid_layout")
        self.width, self.height, self.levels = width, height, int(self.desc.levels)
        self.level0_rows = max(int(level0_rows or height), height)
        pad = (self.level0_rows - height) * width           # texels
        if pad:
            for l in range(1, self.levels):
                self.desc.level_offset[l] += pad
            if (nbytes.value // 8 + pad) * 8 > 0xFFFFFFFF:
                raise _lib.TrError(6, "padded pyramid exceeds 4 GiB")
        total = nbytes.value // 8 + pad
        # out_bytes includes one texel of tail padding (the sampler reads 16-byte texel pairs)
        self._storage = torch.zeros((total, 4), dtype=torch.float16, device=device)
        self.texels = self._storage[:total - 1]
        self.desc.texels = self._storage.data_ptr()

    def level(self, l: int) -> torch.Tensor:
        w, h = max(self.width >> l, 1), max(self.height >> l, 1)
        off = int(self.desc.level_offset[l])
        return self.texels[off:off + w * h].view(h, w, 4)

    def level0_padded(self) -> torch.Tensor:
        return self.texels[:self.width * self.level0_rows].view(self.level0_rows, self.width, 4)


class TransmissionRenderer:
    """One context = one GPU = one host thread (the reference's single queue, src/main.rs:243)."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("transmission_renderer_amd needs a HIP device: there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._ctx = C.c_void_p()
        st = self.lib.tr_context_create(device, C.byref(self._ctx))
        if st != 0:
            raise _lib.TrError(st, "tr_context_create")
        self._keep = {}

    # ---- plumbing
    def _check(self, st: int, where: str):
        if st != 0:
            raise _lib.TrError(st, where, self.lib.tr_last_hip_error(self._ctx))

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.tr_context_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables (descriptor sets 0 and 2 of the reference; src/descriptor_sets.rs:30-80, 149-176)
    def upload_materials(self, materials: Sequence[wire.MaterialInfo]):
        arr = wire.as_ctypes_array(list(materials), wire.MaterialInfo)
        self._check(self.lib.tr_upload_materials(self._ctx, arr, len(materials), self._stream()), "tr_upload_materials")

    def upload_lights(self, lights: Sequence[wire.Light]):
        arr = wire.as_ctypes_array(list(lights), wire.Light) if lights else None
        self._check(self.lib.tr_upload_lights(self._ctx, arr, len(lights), self._stream()), "tr_upload_lights")

    def update_lights(self, first: int, lights: Sequence[wire.Light]):
        """Rewrites lights [first, first + len(lights)) in place (tr_update_lights): the reference's per-frame write of its
        rotating spotlights (src/main.rs:1244-1256).  No allocation, no wait."""
        arr = wire.as_ctypes_array(list(lights), wire.Light)
        self._check(self.lib.tr_update_lights(self._ctx, int(first), len(lights), arr, self._stream()), "tr_update_lights")

    def update_instances(self, first: int, instances: np.ndarray):
        """Rewrites instances [first, first + len(instances)) of the uploaded geometry in place (tr_update_instances): the
        reference's per-frame write of the rotating model's instances (src/main.rs:1258-1261, 1316-1322)."""
        a = np.ascontiguousarray(instances, dtype=wire.INSTANCE_DTYPE)
        self._check(self.lib.tr_update_instances(self._ctx, int(first), len(a), a.ctypes.data, self._stream()), "tr_update_instances")

    def set_cluster_tables(self, counts: torch.Tensor, indices: torch.Tensor):
        assert counts.is_cuda and indices.is_cuda and counts.dtype == torch.int32 and indices.dtype == torch.int32
        assert indices.numel() == counts.numel() * wire.MAX_LIGHTS_PER_CLUSTER
        self._keep["clusters"] = (counts, indices)
        self._check(self.lib.tr_set_cluster_tables(self._ctx, counts.data_ptr(), indices.data_ptr(), counts.numel()),
                    "tr_set_cluster_tables")

    def upload_textures(self, textures: Sequence):
        """The bindless material textures (src/model_loading.rs:160-215): [(HxWx4 uint8 image, srgb), ...]; the
        index of an entry is the texture id materials refer to.  Mip chains are generated on the device."""
        imgs = [np.ascontiguousarray(img, dtype=np.uint8) for img, _ in textures]
        for img in imgs:
            assert img.ndim == 3 and img.shape[2] == 4
        descs = (wire.TextureDesc * max(len(imgs), 1))()
        for i, (img, (_, srgb)) in enumerate(zip(imgs, textures)):
            descs[i] = wire.TextureDesc(img.ctypes.data, img.shape[1], img.shape[0], 1 if srgb else 0, 0)
        self._check(self.lib.tr_upload_textures(self._ctx, descs if imgs else None, len(imgs), self._stream()),
                    "tr_upload_textures")

    def download_texture(self, index: int):
        """(layout, [level arrays]) of texture `index` as it sits in HBM, mip chain included."""
        lay = wire.TextureLayout()
        self._check(self.lib.tr_texture_get_layout(self._ctx, index, C.byref(lay)), "tr_texture_get_layout")
        buf = np.zeros((int(lay.total_texels), 4), dtype=np.uint8)
        self._check(self.lib.tr_download_texture(self._ctx, index, buf.ctypes.data, buf.nbytes, self._stream()),
                    "tr_download_texture")
        levels = []
        for l in range(int(lay.levels)):
            w, h = max(int(lay.width) >> l, 1), max(int(lay.height) >> l, 1)
            off = int(lay.level_offset[l])
            levels.append(buf[off:off + w * h].reshape(h, w, 4))
        return lay, levels

    # ---- geometry front end (model buffers -> the two TGB-v1 layers)
    def upload_geometry(self, geometry: dict):
        """geometry: position (N,3) f32, normal (N,3) f32, uv (N,2) f32, index (M,) u32, primitives / instances
        record arrays (wire.PRIMITIVE_DTYPE / INSTANCE_DTYPE), e.g. from meshes.ModelBuffers or gltf.load_gltf."""
        a = {k: np.ascontiguousarray(geometry[k], dtype=dt) for k, dt in
             (("position", np.float32), ("normal", np.float32), ("uv", np.float32), ("index", np.uint32),
              ("primitives", wire.PRIMITIVE_DTYPE), ("instances", wire.INSTANCE_DTYPE))}
        d = wire.GeometryDesc(a["position"].ctypes.data, a["normal"].ctypes.data, a["uv"].ctypes.data, len(a["position"]),
                              a["index"].ctypes.data, len(a["index"]), a["primitives"].ctypes.data, len(a["primitives"]),
                              a["instances"].ctypes.data, len(a["instances"]))
        self._check(self.lib.tr_upload_geometry(self._ctx, C.byref(d), self._stream()), "tr_upload_geometry")

    def new_layer(self, width: int, height: int) -> "GBufferPlanes":
        dev = self.device
        return GBufferPlanes(torch.empty((height, width, 4), dtype=torch.float32, device=dev),
                             torch.empty((height, width, 4), dtype=torch.float32, device=dev),
                             torch.empty((height, width, 2), dtype=torch.float32, device=dev),
                             torch.empty((height, width), dtype=torch.int32, device=dev))

    @staticmethod
    def _target(layer: "GBufferPlanes") -> wire.GBufferTarget:
        return wire.GBufferTarget(layer.pos_depth.data_ptr(), layer.nrm_scale.data_ptr(), layer.uv.data_ptr(),
                                  layer.material_id.data_ptr())

    def rasterize(self, draw_counts: torch.Tensor, draws, push: wire.PushConstants, opaque: "GBufferPlanes",
                  transmissive: "GBufferPlanes"):
        """The depth pre-passes + colour-pass rasterisation of src/main.rs:1900-2042 for explicit draw buffers."""
        ptrs = (C.c_void_p * 4)(*[d.data_ptr() for d in draws])
        to, tt = self._target(opaque), self._target(transmissive)
        self._check(self.lib.tr_rasterize(self._ctx, draw_counts.data_ptr(), C.byref(ptrs), C.byref(push), C.byref(to),
                                          C.byref(tt), self._stream()), "tr_rasterize")

    def draw_scene(self, culling: wire.CullingPushConstants, push: wire.PushConstants, opaque: "GBufferPlanes",
                   transmissive: "GBufferPlanes"):
        """Culling + demultiplex + rasterisation of the uploaded geometry into the two layers."""
        to, tt = self._target(opaque), self._target(transmissive)
        self._check(self.lib.tr_draw_scene(self._ctx, C.byref(culling), C.byref(push), C.byref(to), C.byref(tt),
                                           self._stream()), "tr_draw_scene")

    # ---- GPU culling (src/main.rs:1716-1763, 1811-1838)
    def frustum_culling(self, primitives: torch.Tensor, instances: torch.Tensor, push: wire.CullingPushConstants):
        """primitives / instances: uint8 device tensors holding PrimitiveInfo / Instance records.
        Returns instance_counts (int32 device tensor, one per primitive)."""
        assert primitives.is_cuda and instances.is_cuda and primitives.dtype == torch.uint8 and instances.dtype == torch.uint8
        n_prim = primitives.numel() // wire.PRIMITIVE_DTYPE.itemsize
        n_inst = instances.numel() // wire.INSTANCE_DTYPE.itemsize
        counts = torch.empty(n_prim, dtype=torch.int32, device=self.device)
        self._check(self.lib.tr_frustum_culling(self._ctx, primitives.data_ptr(), n_prim, instances.data_ptr(), n_inst,
                                                C.byref(push), counts.data_ptr(), self._stream()), "tr_frustum_culling")
        return counts

    def demultiplex_draws(self, primitives: torch.Tensor, instance_counts: torch.Tensor):
        """Returns (draw_counts int32[4], [4 uint8 tensors of tr_draw_command records, capacity = all primitives])."""
        n_prim = primitives.numel() // wire.PRIMITIVE_DTYPE.itemsize
        assert instance_counts.numel() == n_prim and instance_counts.dtype == torch.int32
        draw_counts = torch.empty(4, dtype=torch.int32, device=self.device)
        draws = [torch.zeros(n_prim * wire.DRAW_COMMAND_DTYPE.itemsize, dtype=torch.uint8, device=self.device) for _ in range(4)]
        ptrs = (C.c_void_p * 4)(*[d.data_ptr() for d in draws])
        self._check(self.lib.tr_demultiplex_draws(self._ctx, primitives.data_ptr(), n_prim, instance_counts.data_ptr(),
                                                  draw_counts.data_ptr(), C.byref(ptrs), self._stream()),
                    "tr_demultiplex_draws")
        return draw_counts, draws

    def upload_ggx_lut(self, rgba8: Optional[np.ndarray] = None):
        if rgba8 is None:
            rgba8 = load_ggx_lut()
        rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
        h, w, c = rgba8.shape
        assert c == 4
        self._check(self.lib.tr_upload_ggx_lut(self._ctx, rgba8.ctypes.data_as(C.c_void_p), w, h, self._stream()),
                    "tr_upload_ggx_lut")

    def get_depth_slice(self, coefficients: wire.LightClusterCoefficients, frag_depth: torch.Tensor) -> torch.Tensor:
        """`LightClusterCoefficients::get_depth_slice` (shared-structs/src/lib.rs:54-63) over a float32 device array;
        bit-exact with the reference's fp32 arithmetic (the passes' own cluster-lookup function)."""
        assert frag_depth.dtype == torch.float32 and frag_depth.is_contiguous() and frag_depth.device == self.device
        out = torch.empty(frag_depth.shape, dtype=torch.int32, device=self.device)
        self._check(self.lib.tr_get_depth_slice(self._ctx, C.byref(coefficients), frag_depth.data_ptr(), frag_depth.numel(),
                                                out.data_ptr(), self._stream()), "tr_get_depth_slice")
        return out

    # ---- clustered-light build
    def write_cluster_data(self, uniforms: wire.Uniforms, inverse_perspective: np.ndarray, screen_dimensions) -> torch.Tensor:
        """`write_cluster_data` (shader/src/lib.rs:519-594): (num_clusters, 8) float32 view-space AABBs."""
        n = int(uniforms.num_clusters[0]) * int(uniforms.num_clusters[1]) * \
            int(uniforms.light_clustering_coefficients.num_depth_slices)
        out = torch.empty((n, 8), dtype=torch.float32, device=self.device)
        ip = (C.c_float * 16)(*[float(x) for x in np.asarray(inverse_perspective, dtype=np.float32).reshape(-1)])
        sd = (C.c_uint32 * 2)(int(screen_dimensions[0]), int(screen_dimensions[1]))
        self._check(self.lib.tr_write_cluster_data(self._ctx, C.byref(uniforms), C.byref(ip), C.byref(sd),
                                                   out.data_ptr(), self._stream()), "tr_write_cluster_data")
        return out

    def assign_lights_to_clusters(self, view_matrix: np.ndarray, view_rotation: np.ndarray, aabbs: torch.Tensor,
                                  bind: bool = True):
        """`assign_lights_to_clusters` (shader/src/lib.rs:596-645) for the uploaded lights; returns
        (counts, indices) device tensors and (bind=True) makes them the shading passes' cluster tables."""
        n = int(aabbs.shape[0])
        counts = torch.zeros(n, dtype=torch.int32, device=self.device)
        indices = torch.zeros(n * wire.MAX_LIGHTS_PER_CLUSTER, dtype=torch.int32, device=self.device)
        vm = (C.c_float * 16)(*[float(x) for x in np.asarray(view_matrix, dtype=np.float32).reshape(-1)])
        q = (C.c_float * 4)(*[float(x) for x in np.asarray(view_rotation, dtype=np.float32).reshape(-1)])
        self._check(self.lib.tr_assign_lights_to_clusters(self._ctx, C.byref(vm), C.byref(q), aabbs.data_ptr(), n,
                                                          counts.data_ptr(), indices.data_ptr(), self._stream()),
                    "tr_assign_lights_to_clusters")
        if bind:
            self.set_cluster_tables(counts, indices)
        return counts, indices

    # ---- passes
    @staticmethod
    def _fmt(t: torch.Tensor) -> int:
        if t.dtype == torch.float16:
            return wire.FORMAT_RGBA16F
        if t.dtype == torch.float32:
            return wire.FORMAT_RGBA32F
        raise TypeError("colour targets are RGBA16F or RGBA32F")

    @staticmethod
    def _rect(g: GBufferPlanes, rect) -> wire.Rect:
        if rect is None:  # everything the planes cover
            return wire.Rect(g.origin_x, g.origin_y, g.origin_x + g.width, g.origin_y + g.height)
        return wire.Rect(*[int(v) for v in rect])

    @staticmethod
    def _check_target(hdr: torch.Tensor, push: wire.PushConstants):
        fw, fh = int(push.framebuffer_size[0]), int(push.framebuffer_size[1])
        # whole-frame pitch; a sharded frame's buffer carries padding rows behind the frame (sharded.padded_rows)
        assert hdr.is_cuda and hdr.is_contiguous() and hdr.numel() >= fw * fh * 4 and hdr.shape[-2] == fw, \
            "colour targets are whole-frame"

    def shade_opaque(self, g: GBufferPlanes, uniforms: wire.Uniforms, push: wire.PushConstants, hdr: torch.Tensor,
                     pyramid: Optional[OpaquePyramid] = None, rect=None):
        """"main opaque": `fragment` (shader/src/lib.rs:164-249) -> hdr and pyramid level 0."""
        gs = g.as_struct()
        self._check_target(hdr, push)
        mip0 = pyramid.texels.data_ptr() + int(pyramid.desc.level_offset[0]) * 8 if pyramid is not None else None
        self._check(self.lib.tr_shade_opaque(self._ctx, C.byref(gs), C.byref(uniforms), C.byref(push), hdr.data_ptr(),
                                             self._fmt(hdr), mip0, self._rect(g, rect), self._stream()),
                    "tr_shade_opaque")

    def shade_opaque_pyramid(self, g: GBufferPlanes, uniforms: wire.Uniforms, push: wire.PushConstants, hdr: torch.Tensor,
                             pyramid: OpaquePyramid, rect=None) -> int:
        """"main opaque" writing into the pyramid (tr_shade_opaque_pyramid): level 0 and — RGBA16F target, even frame sizes,
        rect on even pixels — level 1 from the pass's own quads.  Returns the level generate_mips_from continues from."""
        gs = g.as_struct()
        self._check_target(hdr, push)
        nxt = C.c_uint32()
        self._check(self.lib.tr_shade_opaque_pyramid(self._ctx, C.byref(gs), C.byref(uniforms), C.byref(push), hdr.data_ptr(),
                                                     self._fmt(hdr), C.byref(pyramid.desc), self._rect(g, rect), C.byref(nxt),
                                                     self._stream()), "tr_shade_opaque_pyramid")
        return int(nxt.value)

    def generate_mips(self, pyramid: OpaquePyramid):
        """"opaque framebuffer mipchain" (src/main.rs:2046-2064)."""
        self._check(self.lib.tr_generate_mips(self._ctx, C.byref(pyramid.desc), self._stream()), "tr_generate_mips")

    def generate_mips_from(self, pyramid: OpaquePyramid, first_level: int):
        """Levels first_level.. from level first_level - 1 (tr_generate_mips_from)."""
        self._check(self.lib.tr_generate_mips_from(self._ctx, C.byref(pyramid.desc), int(first_level), self._stream()),
                    "tr_generate_mips_from")

    def generate_mips_band(self, pyramid: OpaquePyramid, y0: int, y1: int):
        """Levels 1 and 2 of the rows [y0, y1) of level 0: a rank's band of a sharded frame (tr_generate_mips_band)."""
        self._check(self.lib.tr_generate_mips_band(self._ctx, C.byref(pyramid.desc), int(y0), int(y1), self._stream()),
                    "tr_generate_mips_band")

    def set_tap_window(self, row_lo: int = 0, row_hi: int = 0) -> None:
        """The pyramid rows (levels 0, 1) this rank holds; (0, 0): off.  tap_window_excess() after the transmissive pass."""
        if row_hi == 0:
            self._check(self.lib.tr_set_tap_window(self._ctx, 0, 0, None), "tr_set_tap_window")
            return
        if "tap_excess" not in self._keep:
            self._keep["tap_excess"] = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._keep["tap_excess"].zero_()
        self._check(self.lib.tr_set_tap_window(self._ctx, int(row_lo), int(row_hi), self._keep["tap_excess"].data_ptr()),
                    "tr_set_tap_window")

    def tap_window_excess(self) -> int:
        """0: every tap of the last transmissive pass lay inside the window; else 1 + the rows by which the worst one
        missed it (synchronises)."""
        return int(self._keep["tap_excess"].item()) if "tap_excess" in self._keep else 0

    def tap_window_excess_word(self) -> torch.Tensor:
        """The same word as a device tensor (int64, a copy enqueued on the current stream): nothing synchronises — for a
        caller that judges the frame later (sharded.record_sharded(confirm="late"))."""
        if "tap_excess" not in self._keep:
            return torch.zeros(1, dtype=torch.int64, device=self.device)
        return self._keep["tap_excess"].to(torch.int64)

    def shade_transmission(self, g: GBufferPlanes, uniforms: wire.Uniforms, push: wire.PushConstants,
                           pyramid: OpaquePyramid, hdr: torch.Tensor, rect=None):
        """"opaque transmissive objects": `fragment_transmission` (shader/src/lib.rs:37-162) over hdr (LOAD)."""
        gs = g.as_struct()
        self._check_target(hdr, push)
        self._check(self.lib.tr_shade_transmission(self._ctx, C.byref(gs), C.byref(uniforms), C.byref(push),
                                                   C.byref(pyramid.desc), hdr.data_ptr(), self._fmt(hdr),
                                                   self._rect(g, rect), self._stream()),
                    "tr_shade_transmission")

    def set_strips(self, strip_rows: int, world: int, rank: int) -> None:
        """Rank-interleaved strips (tr_set_strips): whole-frame shade_opaque / shade_transmission calls then shade this
        rank's strips only, in place; set_strips(0, 1, 0) turns it off."""
        self._check(self.lib.tr_set_strips(self._ctx, int(strip_rows), int(world), int(rank)), "tr_set_strips")

    def baked_tonemap_params(self, lottes: Optional[wire.LottesParams] = None) -> wire.TonemapParams:
        if lottes is None:
            lottes = wire.LottesParams()
            self._check(self.lib.tr_lottes_defaults(C.byref(lottes)), "tr_lottes_defaults")
        out = wire.TonemapParams()
        self._check(self.lib.tr_bake_lottes_params(C.byref(lottes), C.byref(out)), "tr_bake_lottes_params")
        return out

    def tonemap(self, hdr: torch.Tensor, params: Optional[wire.TonemapParams] = None, bgra: bool = False,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """"tonemapping": fragment_tonemap (shader/src/lib.rs:683-697) -> (H, W, 4) uint8, sRGB encoded (into `out` when
        given: e.g. a rank's rows of the whole 8-bit frame; the operator is pointwise, so a row band is a frame of its own)."""
        assert hdr.dtype == torch.float16 and hdr.is_cuda and hdr.is_contiguous() and hdr.shape[-1] == 4
        h, w = int(hdr.shape[0]), int(hdr.shape[1])
        params = params or self.baked_tonemap_params()
        if out is None:
            out = torch.empty((h, w, 4), dtype=torch.uint8, device=self.device)
        assert out.dtype == torch.uint8 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (h, w, 4)
        self._check(self.lib.tr_tonemap(self._ctx, hdr.data_ptr(), w, h, C.byref(params), out.data_ptr(), int(bgra),
                                        self._stream()), "tr_tonemap")
        return out

    def tonemap_rgb8(self, hdr: torch.Tensor, params: Optional[wire.TonemapParams] = None, bgra: bool = False,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same pixels as tonemap() without the constant alpha byte: (H, W, 3) uint8 (tr_tonemap_rgb8) — what a
        row-band sharded frame composites."""
        assert hdr.dtype == torch.float16 and hdr.is_cuda and hdr.is_contiguous() and hdr.shape[-1] == 4
        h, w = int(hdr.shape[0]), int(hdr.shape[1])
        params = params or self.baked_tonemap_params()
        if out is None:
            out = torch.empty((h, w, 3), dtype=torch.uint8, device=self.device)
        assert out.dtype == torch.uint8 and out.is_cuda and out.is_contiguous() and tuple(out.shape) == (h, w, 3)
        self._check(self.lib.tr_tonemap_rgb8(self._ctx, hdr.data_ptr(), w, h, C.byref(params), out.data_ptr(), int(bgra),
                                             self._stream()), "tr_tonemap_rgb8")
        return out

    def record_frame(self, uniforms: wire.Uniforms, push: wire.PushConstants, culling: wire.CullingPushConstants,
                     view_matrix: np.ndarray, view_rotation: np.ndarray, aabbs: torch.Tensor, work: dict, tonemap=True,
                     timed: bool = False, lottes: Optional[wire.LottesParams] = None):
        """One frame of the uploaded scene through tr_record_frame (the native recorder: culling, light assignment,
        demultiplex, rasteriser, opaque, mips, transmissive, tonemap).  `work` = new_frame_buffers(); returns
        (hdr, ldr or None) — and, with timed=True (tr_record_frame_timed, blocks), a dict {zone name: ms} under the
        reference's profiling zone names (src/main.rs:1643-2227)."""
        vm = (C.c_float * 16)(*[float(x) for x in np.asarray(view_matrix, dtype=np.float32).reshape(-1)])
        q = (C.c_float * 4)(*[float(x) for x in np.asarray(view_rotation, dtype=np.float32).reshape(-1)])
        params = self.baked_tonemap_params(lottes) if tonemap else None   # (lottes: the operator's eight constants; None = tr_lottes_defaults)
        self._keep["clusters"] = (work["counts"], work["indices"])
        d = wire.FrameDesc()
        d.push, d.uniforms, d.culling = C.pointer(push), C.pointer(uniforms), C.pointer(culling)
        d.view_matrix, d.view_rotation = C.cast(vm, C.POINTER(C.c_float)), C.cast(q, C.POINTER(C.c_float))
        d.cluster_aabbs, d.num_clusters = aabbs.data_ptr(), int(aabbs.shape[0])
        d.cluster_light_counts, d.light_indices = work["counts"].data_ptr(), work["indices"].data_ptr()
        d.opaque_layer, d.transmissive_layer = self._target(work["opaque"]), self._target(work["transmissive"])
        d.pyramid = work["pyramid"].desc
        d.hdr, d.hdr_format = work["hdr"].data_ptr(), self._fmt(work["hdr"])
        d.bgra = 0
        if tonemap:
            d.tonemap, d.ldr_out = C.pointer(params), work["ldr"].data_ptr()
        if timed:
            zones = (wire.FrameZone * 16)()
            n = C.c_uint32()
            self._check(self.lib.tr_record_frame_timed(self._ctx, C.byref(d), self._stream(), zones, 16, C.byref(n)),
                        "tr_record_frame_timed")
            times = {zones[i].name.decode(): float(zones[i].milliseconds) for i in range(n.value)}
            return work["hdr"], (work["ldr"] if tonemap else None), times
        self._check(self.lib.tr_record_frame(self._ctx, C.byref(d), self._stream()), "tr_record_frame")
        return work["hdr"], (work["ldr"] if tonemap else None)

    def new_frame_buffers(self, width: int, height: int, num_clusters: int = wire.NUM_CLUSTERS) -> dict:
        """The per-frame device buffers tr_record_frame works in (allocated once, reused every frame)."""
        dev = self.device
        return {"opaque": self.new_layer(width, height), "transmissive": self.new_layer(width, height),
                "pyramid": OpaquePyramid(width, height, dev),
                "hdr": torch.zeros((height, width, 4), dtype=torch.float16, device=dev),
                "ldr": torch.zeros((height, width, 4), dtype=torch.uint8, device=dev),
                "counts": torch.zeros(num_clusters, dtype=torch.int32, device=dev),
                "indices": torch.zeros(num_clusters * wire.MAX_LIGHTS_PER_CLUSTER, dtype=torch.int32, device=dev)}

    def record(self, opaque: GBufferPlanes, transmissive: GBufferPlanes, uniforms: wire.Uniforms,
               push: wire.PushConstants, hdr: torch.Tensor, pyramid: OpaquePyramid, rect=None):
        """The hot-path slice of `record()` in the reference's order (src/main.rs:1969-2124)."""
        whole = rect is None and opaque.origin_x == 0 and opaque.origin_y == 0 and opaque.width == int(push.framebuffer_size[0]) \
            and opaque.height == int(push.framebuffer_size[1])
        if whole and pyramid.desc.height == int(push.framebuffer_size[1]):
            # the opaque launch writes level 1 from its own quads where it can: the chain never reads level 0 back
            self.generate_mips_from(pyramid, self.shade_opaque_pyramid(opaque, uniforms, push, hdr, pyramid))
        else:
            self.shade_opaque(opaque, uniforms, push, hdr, pyramid, rect)
            self.generate_mips(pyramid)
        self.shade_transmission(transmissive, uniforms, push, pyramid, hdr, rect)
