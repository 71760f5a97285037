"""Synthetic "TGB-v1" G-buffer scenes: the benchmark / parity workload.

The reference renders glTF-Sample-Models through a rasteriser; neither the assets nor a
rasteriser exist here, so the shading passes are fed what the rasteriser would have produced:
per-pixel world position, interpolated normal, uv, flat material id, flat model scale and
frag_coord.z, for a surface that is consistent with the reference's own camera
(src/main.rs:39-54, 511-523), sun (:531-538), lights (:450-453) and cluster grid (:56-63).
Everything is a pure function of (width, height, seed): analytic fields per pixel, a
counter-based splitmix64 stream for the 16 materials.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import wire

f32 = np.float32
SEED = 0x7472616E736D6974  # "transmit"


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0) -> float:
        return lo + (hi - lo) * ((self.next_u64() >> 11) / float(1 << 53))


def make_materials(num=16, seed=SEED, roughness_override=None):
    """16 transmissive/volume materials spanning the parameter space of KHR_materials_{ior,transmission,
    volume,specular}; includes ior exactly 1.0 and 1.5, +INF attenuation distance,
    metallic 0/1/in-between, transmission_factor 0/0.5/1."""
    rng = SplitMix64(seed)
    mats = []
    for i in range(num):
        rough = rng.uniform(0.03, 1.0)
        ior = rng.uniform(1.0, 2.0)
        if i == 0:
            ior = 1.5   # glTF default; the dragon of DragonAttenuation
        if i == 1:
            ior = 1.0   # alpha_t = 0: the btdf lobe vanishes
        thickness = rng.uniform(0.0, 2.0)
        att_d = math.inf if (i % 3 == 2) else rng.uniform(0.05, 2.0)
        att_c = [rng.uniform(0.05, 1.0) for _ in range(3)]
        diffuse = [rng.uniform(0.05, 1.0) for _ in range(3)]
        tf = (1.0, 0.5, 1.0, 0.0)[i % 4] if i else 1.0
        metallic = (0.0, 0.0, rng.uniform(0.0, 1.0), 1.0)[(i // 2) % 4] if i else 0.0
        emissive = [0.0, 0.0, 0.0] if i != 5 else [0.05, 0.02, 0.0]
        spec = 1.0 if i % 5 else rng.uniform(0.3, 1.0)
        spec_c = [1.0, 1.0, 1.0] if i % 7 else [rng.uniform(0.5, 1.0) for _ in range(3)]
        if roughness_override is not None:
            rough = float(roughness_override)  # --roughness-override (src/model_loading.rs:294)
        mats.append(wire.MaterialInfo.default(
            metallic_factor=float(f32(metallic)), roughness_factor=float(f32(rough)),
            diffuse_factor=diffuse + [1.0], emissive_factor=emissive, index_of_refraction=float(f32(ior)),
            transmission_factor=float(f32(tf)), thickness_factor=float(f32(thickness)),
            attenuation_distance=att_d if math.isinf(att_d) else float(f32(att_d)), attenuation_colour=att_c,
            specular_factor=float(f32(spec)), specular_colour_factor=spec_c))
    return mats


def make_textures():
    """Eight procedural RGBA8 material textures (level 0) with the colour space the reference's loader would give
    them (src/model_loading.rs:233-291): sizes include non-square and odd ones.  Returns [(image, srgb), ...]."""
    def grid(w, h):
        x = (np.arange(w, dtype=np.float64) + 0.5)[None, :] / w
        y = (np.arange(h, dtype=np.float64) + 0.5)[:, None] / h
        return x, y

    def pack(*ch):
        return np.ascontiguousarray(np.stack([np.clip(np.broadcast_to(c, ch[0].shape) * 255.0 + 0.5, 0, 255).astype(np.uint8)
                                              for c in ch], axis=-1))
    out = []
    x, y = grid(64, 64)        # 0 diffuse (sRGB): stripes with alpha
    out.append((pack(0.5 + 0.5 * np.sin(x * 40) * np.ones_like(y), 0.3 + 0.6 * y * np.ones_like(x), 0.8 * x * y,
                     0.6 + 0.4 * np.cos(y * 9) * np.ones_like(x)), True))
    x, y = grid(128, 64)       # 1 metallic (b) / roughness (g), UNORM, non-square
    out.append((pack(0.0 * x * y, 0.15 + 0.8 * (0.5 + 0.5 * np.sin(x * 25 + y * 7)), 0.5 + 0.5 * np.cos(y * 13) * np.ones_like(x),
                     1.0 + 0.0 * x * y), False))
    x, y = grid(96, 96)        # 2 tangent-space normal map, UNORM
    nx, ny = 0.35 * np.sin(x * 30) * np.ones_like(y), 0.35 * np.cos(y * 26) * np.ones_like(x)
    nz = np.sqrt(np.clip(1 - nx * nx - ny * ny, 0, 1))
    out.append((pack(nx * 0.5 + 0.5, ny * 0.5 + 0.5, nz * 0.5 + 0.5, 1.0 + 0 * nx), False))
    x, y = grid(37, 21)        # 3 emissive (sRGB), odd sizes
    out.append((pack(0.2 * (x > 0.5) * np.ones_like(y), 0.1 * y * np.ones_like(x), 0.3 * x * (y < 0.4), 1.0 + 0 * x * y), True))
    x, y = grid(32, 32)        # 4 transmission (r), UNORM
    out.append((pack(0.4 + 0.6 * (((x * 4).astype(int) + (y * 4).astype(int)) % 2), 0.5 + 0 * x * y, 0.5 + 0 * x * y, 1.0 + 0 * x * y), False))
    x, y = grid(64, 16)        # 5 thickness (g), UNORM
    out.append((pack(0.0 * x * y, 0.3 + 0.7 * x * np.ones_like(y), 0.0 * x * y, 1.0 + 0 * x * y), False))
    x, y = grid(16, 16)        # 6 specular factor (a)
    out.append((pack(0.5 + 0 * x * y, 0.5 + 0 * x * y, 0.5 + 0 * x * y, 0.4 + 0.6 * x * y), False))
    x, y = grid(48, 80)        # 7 specular colour (sRGB)
    out.append((pack(0.6 + 0.4 * x * np.ones_like(y), 0.7 + 0.3 * y * np.ones_like(x), 0.9 - 0.4 * x * y, 1.0 + 0 * x * y), True))
    return out


def apply_textures(materials):
    """Gives five of the synthetic materials texture slots (ids into make_textures()): one with every slot the
    shaders read, the others with typical subsets."""
    T = wire.Textures
    sets = {2: dict(diffuse=0), 5: dict(metallic_roughness=1, normal_map=2), 7: dict(diffuse=0, emissive=3, transmission=4),
            9: dict(diffuse=0, metallic_roughness=1, normal_map=2, emissive=3, transmission=4, thickness=5, specular=6,
                    specular_colour=7),
            12: dict(thickness=5, specular=6, specular_colour=7)}
    for i, slots in sets.items():
        if i < len(materials):
            t = T(*([-1] * 9))
            for k, v in slots.items():
                setattr(t, k, v)
            materials[i].textures = t
    return materials


def make_lights(num_point_lights: int):
    """The reference's first point light (src/main.rs:451) for N=1, its two for N=2, plus two more for N=4."""
    pool = wire.default_lights() + [
        wire.Light.new_point((-1.5, 3.0, -1.0), (0.2, 0.3, 1.0), 4.0),
        wire.Light.new_point((1.5, 2.0, -2.5), (1.0, 1.0, 1.0), 3.0),
        wire.Light.new_point((0.0, 3.5, -3.0), (1.0, 0.6, 0.2), 6.0),
        wire.Light.new_point((-2.0, 1.0, -2.0), (0.3, 1.0, 0.4), 2.0),
    ]
    reps = (num_point_lights + len(pool) - 1) // len(pool) if num_point_lights else 0
    return (pool * max(reps, 1))[:num_point_lights]


def all_lights_cluster_tables(num_lights: int, num_clusters=wire.NUM_CLUSTERS):
    """Every cluster lists every light, in index order (what assign_lights_to_clusters produces when each
    light's falloff sphere covers the view volume, made deterministic)."""
    counts = np.full(num_clusters, num_lights, dtype=np.uint32)
    indices = np.zeros((num_clusters, wire.MAX_LIGHTS_PER_CLUSTER), dtype=np.uint32)
    indices[:, :num_lights] = np.arange(num_lights, dtype=np.uint32)[None, :]
    return counts, indices.reshape(-1)


def make_gbuffer(width: int, height: int, num_materials=16, coverage="full", rows=None):
    """TGB-v1 planes for rows [rows[0], rows[1]) (default: all) of a width x height frame, as numpy arrays."""
    eye, view = wire.default_camera()
    proj = wire.perspective_matrix_reversed(width, height)
    y0, y1 = (0, height) if rows is None else rows
    xs = (np.arange(width, dtype=np.float64) + 0.5)
    ys = (np.arange(y0, y1, dtype=np.float64) + 0.5)
    ndc_x = (xs / width * 2.0 - 1.0)[None, :]
    ndc_y = (ys / height * 2.0 - 1.0)[:, None]
    fx, fy = float(proj[0, 0]), float(-proj[1, 1])
    # view-space ray through the pixel centre and a wavy depth field 1.7..3.5 m in front of the camera
    dvx, dvy = ndc_x / fx, -ndc_y / fy
    zv = 2.6 + 0.6 * np.sin(7.0 * ndc_x + 1.3) * np.cos(5.0 * ndc_y) + 0.25 * np.sin(23.0 * ndc_x * ndc_y)
    vx, vy, vz = dvx * zv, dvy * zv, -zv
    # camera basis in world space (rows of the view matrix rotation)
    v = view.astype(np.float64)
    s = np.array([v[0, 0], v[1, 0], v[2, 0]])
    u = np.array([v[0, 1], v[1, 1], v[2, 1]])
    b = np.array([v[0, 2], v[1, 2], v[2, 2]])  # = -forward
    e = eye.astype(np.float64)
    pos = [e[k] + s[k] * vx + u[k] * vy + b[k] * vz for k in range(3)]
    a_, b_ = float(proj[2, 2]), float(proj[3, 2])
    depth = (a_ * vz + b_) / (-vz)  # clip.z / clip.w, reversed-Z: 1 at z_near, 0 at z_far
    # normal: towards the camera, bent by a two-octave ripple, left un-normalised like an interpolant
    inv = 1.0 / np.sqrt(vx * vx + vy * vy + vz * vz)
    tcx, tcy, tcz = -vx * inv, -vy * inv, -vz * inv  # view-space direction to the camera
    rx = 0.9 * np.sin(31.0 * ndc_x + 2.0 * ndc_y) + 0.3 * np.sin(97.0 * ndc_y)
    ry = 0.9 * np.cos(27.0 * ndc_y - 3.0 * ndc_x) + 0.3 * np.cos(89.0 * ndc_x)
    nvx, nvy, nvz = tcx + rx, tcy + ry, tcz + 0.0 * rx
    nlen = (0.75 + 0.25 * np.sin(11.0 * ndc_x + 5.0 * ndc_y))
    nrm = [(s[k] * nvx + u[k] * nvy + b[k] * nvz) * nlen for k in range(3)]

    h = y1 - y0
    pos_depth = np.empty((h, width, 4), dtype=f32)
    nrm_scale = np.empty((h, width, 4), dtype=f32)
    for k in range(3):
        pos_depth[..., k] = pos[k]
        nrm_scale[..., k] = nrm[k]
    pos_depth[..., 3] = depth
    uv = np.empty((h, width, 2), dtype=f32)
    uv[..., 0] = np.broadcast_to(xs[None, :] / width * 4.0, (h, width))
    uv[..., 1] = np.broadcast_to(ys[:, None] / height * 4.0, (h, width))

    # "objects": a 16 x 9 grid of cells with wavy borders; flat material id and model scale per cell
    cw, ch = width / 16.0, height / 9.0
    xw = xs[None, :] + 0.35 * cw * np.sin(ys[:, None] * (2.0 * np.pi / (3.1 * ch)))
    yw = ys[:, None] + 0.35 * ch * np.sin(xs[None, :] * (2.0 * np.pi / (2.7 * cw)))
    cx = np.floor(xw / cw).astype(np.int64)
    cy = np.floor(yw / ch).astype(np.int64)
    hsh = (cx * 73856093) ^ (cy * 19349663) ^ ((cx + cy) * 83492791)
    hsh = (hsh ^ (hsh >> 13)) & 0x7FFFFFFF
    material_id = (hsh % num_materials).astype(np.uint32)
    nrm_scale[..., 3] = np.array([1.0, 0.5, 2.0, 1.0], dtype=f32)[(hsh >> 8) % 4]
    if coverage == "holes":
        hole = (np.sin(9.0 * ndc_x) * np.sin(7.0 * ndc_y)) > 0.8
        material_id = np.where(hole, np.uint32(wire.NOT_COVERED), material_id).astype(np.uint32)
    elif coverage != "full":
        raise ValueError(coverage)
    return {"pos_depth": pos_depth, "nrm_scale": nrm_scale, "uv": uv, "material_id": np.ascontiguousarray(material_id),
            "width": width, "height": h, "origin_x": 0, "origin_y": y0, "frame_width": width, "frame_height": height}


def make_opaque_mip0(width: int, height: int) -> np.ndarray:
    """Procedural opaque-colour frame (checker x gradient x a few highlights, HDR in [0, 4]) as RGBA16F."""
    xs = (np.arange(width, dtype=np.float64) + 0.5)[None, :]
    ys = (np.arange(height, dtype=np.float64) + 0.5)[:, None]
    sq = max(width // 120, 2)
    checker = (((xs // sq).astype(np.int64) + (ys // sq).astype(np.int64)) & 1).astype(np.float64)
    g = 0.25 + 0.75 * checker
    r = g * (0.2 + 1.8 * xs / width)
    gch = g * (0.2 + 1.8 * ys / height)
    bch = g * (1.0 + 0.8 * np.sin(xs / width * 12.0) * np.cos(ys / height * 9.0))
    spot = np.exp(-(((xs / width - 0.3) ** 2 + (ys / height - 0.4) ** 2) * 900.0)) * 3.0
    spot = spot + np.exp(-(((xs / width - 0.72) ** 2 + (ys / height - 0.63) ** 2) * 2500.0)) * 2.0
    img = np.empty((height, width, 4), dtype=np.float16)
    img[..., 0] = np.clip(r + spot, 0.0, 4.0)
    img[..., 1] = np.clip(gch + spot, 0.0, 4.0)
    img[..., 2] = np.clip(bch + spot, 0.0, 4.0)
    img[..., 3] = 1.0
    return img


def make_scene(width: int, height: int, num_point_lights=1, seed=SEED, roughness_override=None, coverage="full",
               num_materials=16, with_gbuffer=True, textured=False):
    """Everything one frame needs, host side.  Returns a dict of numpy arrays + ctypes structs."""
    scene = {
        "width": width, "height": height,
        "materials": make_materials(num_materials, seed, roughness_override),
        "lights": make_lights(num_point_lights),
        "uniforms": wire.make_uniforms(width, height),
        "push": wire.make_push_constants(width, height),
    }
    if textured:
        scene["textures"] = make_textures()
        apply_textures(scene["materials"])
    scene["cluster_counts"], scene["light_indices"] = all_lights_cluster_tables(num_point_lights)
    if with_gbuffer:
        scene["gbuffer"] = make_gbuffer(width, height, num_materials, coverage)
    return scene
