"""Host-side mirror of the reference's wire structs and the host helpers that feed the hot path.

ctypes twins of include/tr_shade.h (which mirrors shared-structs/src/lib.rs byte for byte) plus
the few pieces of `src/main.rs` / `src/model_loading.rs` whose *values* reach the shading
kernels: the reversed-Z projection, the camera rig's initial pose, the sun, the light
constructors, the cluster coefficients, the material defaults and mip_levels_for_size.
All arithmetic is float32, in the reference's operation order.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

f32 = np.float32

# --------------------------------------------------------------------------- constants
Z_NEAR = f32(0.01)          # src/main.rs:56
Z_FAR = f32(500.0)          # src/main.rs:57
NUM_CLUSTERS_X = 24         # src/main.rs:60
NUM_CLUSTERS_Y = 16         # src/main.rs:61
NUM_DEPTH_SLICES = 16       # src/main.rs:62
NUM_CLUSTERS = NUM_CLUSTERS_X * NUM_CLUSTERS_Y * NUM_DEPTH_SLICES
MAX_LIGHTS_PER_CLUSTER = 128  # shared-structs/src/lib.rs:322
NOT_COVERED = 0xFFFFFFFF
FORMAT_RGBA16F = 0
FORMAT_RGBA32F = 1
FORMAT_RGBA8 = 2        # tr_allgather_frame only
FORMAT_RGB8 = 3         # ... without the constant alpha (tr_tonemap_rgb8's output)
MAX_MIP_LEVELS = 16


# --------------------------------------------------------------------------- wire structs
class PushConstants(C.Structure):  # shared-structs/src/lib.rs:8-16
    _fields_ = [
        ("proj_view", C.c_float * 16),
        ("view_position", C.c_float * 3),
        ("_pad0", C.c_float),
        ("framebuffer_size", C.c_uint32 * 2),
        ("acceleration_structure_address", C.c_uint64),
    ]


class LightClusterCoefficients(C.Structure):  # shared-structs/src/lib.rs:31-68
    _fields_ = [
        ("z_near", C.c_float),
        ("z_far", C.c_float),
        ("scale", C.c_float),
        ("bias", C.c_float),
        ("num_depth_slices", C.c_uint32),
        ("_pad", C.c_uint32 * 3),
    ]

    @classmethod
    def new(cls, z_near=Z_NEAR, z_far=Z_FAR, num_depth_slices=NUM_DEPTH_SLICES):
        """LightClusterCoefficients::new (shared-structs/src/lib.rs:44-52)."""
        z_near, z_far = f32(z_near), f32(z_far)
        ratio_log = f32(np.log2(f32(z_far / z_near)))
        scale = f32(f32(num_depth_slices) / ratio_log)
        bias = f32(-(f32(f32(num_depth_slices) * f32(np.log2(z_near))) / ratio_log))
        return cls(float(z_near), float(z_far), float(scale), float(bias), int(num_depth_slices))


class Uniforms(C.Structure):  # shared-structs/src/lib.rs:18-29
    _fields_ = [
        ("light_clustering_coefficients", LightClusterCoefficients),
        ("sun_dir", C.c_float * 3),
        ("_pad0", C.c_float),
        ("sun_intensity", C.c_float * 3),
        ("_pad1", C.c_float),
        ("cluster_size_in_pixels", C.c_float * 2),
        ("num_clusters", C.c_uint32 * 2),
        ("debug_clusters", C.c_uint32),
        ("ggx_lut_texture_index", C.c_uint32),
        ("_pad2", C.c_uint32 * 2),
    ]


class Textures(C.Structure):  # shared-structs/src/lib.rs:141-153
    _fields_ = [(n, C.c_int32) for n in (
        "diffuse", "metallic_roughness", "normal_map", "emissive", "occlusion",
        "transmission", "thickness", "specular", "specular_colour")]


class MaterialInfo(C.Structure):  # shared-structs/src/lib.rs:155-173
    _fields_ = [
        ("textures", Textures),
        ("metallic_factor", C.c_float),
        ("roughness_factor", C.c_float),
        ("alpha_clipping_cutoff", C.c_float),
        ("diffuse_factor", C.c_float * 4),
        ("emissive_factor", C.c_float * 3),
        ("_pad0", C.c_float),
        ("normal_map_scale", C.c_float),
        ("occlusion_strength", C.c_float),
        ("index_of_refraction", C.c_float),
        ("transmission_factor", C.c_float),
        ("thickness_factor", C.c_float),
        ("attenuation_distance", C.c_float),
        ("_pad1", C.c_float * 2),
        ("attenuation_colour", C.c_float * 3),
        ("_pad2", C.c_float),
        ("specular_factor", C.c_float),
        ("_pad3", C.c_float * 3),
        ("specular_colour_factor", C.c_float * 3),
        ("_pad4", C.c_float),
    ]

    @classmethod
    def default(cls, **overrides):
        """The glTF defaults `load_gltf` fills in (src/model_loading.rs:293-332)."""
        m = cls()
        m.textures = Textures(*([-1] * 9))
        m.metallic_factor = 1.0
        m.roughness_factor = 1.0
        m.alpha_clipping_cutoff = 0.5
        m.diffuse_factor = (C.c_float * 4)(1.0, 1.0, 1.0, 1.0)
        m.emissive_factor = (C.c_float * 3)(0.0, 0.0, 0.0)
        m.normal_map_scale = 0.0
        m.occlusion_strength = 1.0
        m.index_of_refraction = 1.5
        m.transmission_factor = 0.0
        m.thickness_factor = 0.0
        m.attenuation_distance = math.inf
        m.attenuation_colour = (C.c_float * 3)(1.0, 1.0, 1.0)
        m.specular_factor = 1.0
        m.specular_colour_factor = (C.c_float * 3)(1.0, 1.0, 1.0)
        for k, v in overrides.items():
            if isinstance(v, (tuple, list, np.ndarray)):
                arr = getattr(m, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(m, k, v)
        return m


class Light(C.Structure):  # shared-structs/src/lib.rs:70-139
    _fields_ = [
        ("position_and_spotlight_epsilon", C.c_float * 4),
        ("colour_emission_and_falloff_distance_sq", C.c_float * 4),
        ("spotlight_direction_and_outer_angle", C.c_float * 4),
    ]

    @classmethod
    def new_point(cls, position, colour, intensity):
        """Light::new_point (shared-structs/src/lib.rs:94-103)."""
        intensity = f32(intensity)
        falloff = f32(intensity / f32(0.05))
        c = [f32(f32(x) * intensity) for x in colour]
        return cls((C.c_float * 4)(*[float(f32(p)) for p in position], 0.0),
                   (C.c_float * 4)(*[float(x) for x in c], float(falloff)),
                   (C.c_float * 4)(0.0, 0.0, 0.0, 0.0))

    @classmethod
    def new_spot(cls, position, colour, intensity, direction, inner_angle_rad, outer_angle_rad):
        """Light::new_spot (shared-structs/src/lib.rs:105-123)."""
        intensity = f32(intensity)
        falloff = f32(intensity / f32(0.05))
        eps = f32(f32(np.cos(f32(inner_angle_rad))) - f32(np.cos(f32(outer_angle_rad))))
        c = [f32(f32(x) * intensity) for x in colour]
        return cls((C.c_float * 4)(*[float(f32(p)) for p in position], float(eps)),
                   (C.c_float * 4)(*[float(x) for x in c], float(falloff)),
                   (C.c_float * 4)(*[float(f32(d)) for d in direction], float(f32(outer_angle_rad))))


class ClusterAabb(C.Structure):  # shared-structs/src/lib.rs:282-288
    _fields_ = [("min", C.c_float * 3), ("_pad0", C.c_float), ("max", C.c_float * 3), ("_pad1", C.c_float)]


class TonemapParams(C.Structure):  # shader/src/tonemapping.rs:29-39 BakedLottesTonemapperParams
    _fields_ = [(n, C.c_float) for n in ("a", "b", "c", "d", "crosstalk", "saturation", "cross_saturation")]


class LottesParams(C.Structure):  # colstodian LottesTonemapperParams (un-vendored: explicit inputs)
    _fields_ = [(n, C.c_float) for n in ("contrast", "shoulder", "hdr_max", "mid_in", "mid_out", "crosstalk",
                                         "saturation", "cross_saturation")]


class GBuffer(C.Structure):  # include/tr_shade.h tr_gbuffer
    _fields_ = [
        ("pos_depth", C.c_void_p),
        ("nrm_scale", C.c_void_p),
        ("uv", C.c_void_p),
        ("material_id", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("origin_x", C.c_uint32),
        ("origin_y", C.c_uint32),
    ]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32)]


class Pyramid(C.Structure):  # include/tr_shade.h tr_pyramid
    _fields_ = [
        ("texels", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("levels", C.c_uint32),
        ("level_offset", C.c_uint32 * MAX_MIP_LEVELS),
    ]


# ---- geometry / draw records (shared-structs/src/lib.rs:238-281) as numpy record layouts
INSTANCE_DTYPE = np.dtype([("translation_and_scale", np.float32, 4), ("rotation", np.float32, 4),
                           ("primitive_id", np.uint32), ("material_id", np.uint32), ("_pad", np.uint32, 2)])
PRIMITIVE_DTYPE = np.dtype([("packed_bounding_sphere", np.float32, 4), ("draw_buffer_index", np.uint32),
                            ("index_count", np.uint32), ("first_index", np.uint32), ("first_instance", np.uint32)])
DRAW_COMMAND_DTYPE = np.dtype([("index_count", np.uint32), ("instance_count", np.uint32), ("first_index", np.uint32),
                               ("vertex_offset", np.int32), ("first_instance", np.uint32)])
assert INSTANCE_DTYPE.itemsize == 48 and PRIMITIVE_DTYPE.itemsize == 32 and DRAW_COMMAND_DTYPE.itemsize == 20

# ---- the glam-pbr API's records (include/tr_shade.h; glam-pbr/src/lib.rs:163-179, 200-205, 235-246, 438-441)
MATERIAL_PARAMS_DTYPE = np.dtype([("diffuse_colour", np.float32, 3), ("metallic", np.float32),
                                  ("perceptual_roughness", np.float32), ("index_of_refraction", np.float32),
                                  ("specular_colour", np.float32, 3), ("specular_factor", np.float32)])
BASIC_BRDF_PARAMS_DTYPE = np.dtype([("normal", np.float32, 3), ("light", np.float32, 3), ("light_intensity", np.float32, 3),
                                    ("view", np.float32, 3), ("material_params", MATERIAL_PARAMS_DTYPE)])
BRDF_RESULT_DTYPE = np.dtype([("diffuse", np.float32, 3), ("specular", np.float32, 3)])
TRANSMISSION_BTDF_PARAMS_DTYPE = np.dtype([("material_params", MATERIAL_PARAMS_DTYPE), ("normal", np.float32, 3),
                                           ("view", np.float32, 3), ("light", np.float32, 3)])
IBL_VOLUME_REFRACTION_PARAMS_DTYPE = np.dtype([
    ("material_params", MATERIAL_PARAMS_DTYPE), ("framebuffer_size_x", np.uint32), ("normal", np.float32, 3),
    ("view", np.float32, 3), ("proj_view_matrix", np.float32, 16), ("position", np.float32, 3), ("thickness", np.float32),
    ("model_scale", np.float32), ("attenuation_distance", np.float32), ("attenuation_colour", np.float32, 3)])
LIGHT_DIRECTION_DTYPE = np.dtype([("direction", np.float32, 3), ("distance", np.float32), ("attenuation", np.float32)])
assert (MATERIAL_PARAMS_DTYPE.itemsize, BASIC_BRDF_PARAMS_DTYPE.itemsize, BRDF_RESULT_DTYPE.itemsize,
        TRANSMISSION_BTDF_PARAMS_DTYPE.itemsize, IBL_VOLUME_REFRACTION_PARAMS_DTYPE.itemsize,
        LIGHT_DIRECTION_DTYPE.itemsize) == (40, 88, 24, 76, 168, 20)
NUM_DRAW_BUFFERS = 4


class CullingPushConstants(C.Structure):  # shared-structs/src/lib.rs:270-279
    _fields_ = [("view", C.c_float * 16), ("frustum_x_xz", C.c_float * 2), ("frustum_y_yz", C.c_float * 2),
                ("z_near", C.c_float), ("_pad", C.c_float * 3)]

    @classmethod
    def new(cls, perspective: np.ndarray, view: np.ndarray, z_near: float = Z_NEAR) -> "CullingPushConstants":
        """src/main.rs:1726-1746: frustum_x = (row3 + row0).truncate().normalize(), frustum_y likewise with row1.
        `perspective` and `view` are 4x4 float32 arrays indexed [column][row] like the rest of this module."""
        p = np.asarray(perspective, dtype=np.float32)

        def plane(i):
            v = (p[:3, 3] + p[:3, i]).astype(np.float32)
            return _normalize(v)
        fx, fy = plane(0), plane(1)
        out = cls()
        out.view = (C.c_float * 16)(*np.asarray(view, dtype=np.float32).reshape(-1))   # column-major
        out.frustum_x_xz = (C.c_float * 2)(float(fx[0]), float(fx[2]))
        out.frustum_y_yz = (C.c_float * 2)(float(fy[1]), float(fy[2]))
        out.z_near = float(z_near)
        return out


class GeometryDesc(C.Structure):  # include/tr_shade.h tr_geometry_desc
    _fields_ = [("position", C.c_void_p), ("normal", C.c_void_p), ("uv", C.c_void_p), ("num_vertices", C.c_uint32),
                ("index", C.c_void_p), ("num_indices", C.c_uint32), ("primitives", C.c_void_p),
                ("num_primitives", C.c_uint32), ("instances", C.c_void_p), ("num_instances", C.c_uint32)]


class GBufferTarget(C.Structure):  # include/tr_shade.h tr_gbuffer_target
    _fields_ = [("pos_depth", C.c_void_p), ("nrm_scale", C.c_void_p), ("uv", C.c_void_p), ("material_id", C.c_void_p)]


class FrameZone(C.Structure):  # include/tr_shade.h tr_frame_zone
    _fields_ = [("name", C.c_char_p), ("milliseconds", C.c_float), ("_pad", C.c_uint32)]


class FrameDesc(C.Structure):  # include/tr_shade.h tr_frame_desc
    _fields_ = [("push", C.POINTER(PushConstants)), ("uniforms", C.POINTER(Uniforms)),
                ("culling", C.POINTER(CullingPushConstants)), ("view_matrix", C.POINTER(C.c_float)),
                ("view_rotation", C.POINTER(C.c_float)), ("cluster_aabbs", C.c_void_p), ("num_clusters", C.c_uint32),
                ("_reserved", C.c_uint32), ("cluster_light_counts", C.c_void_p), ("light_indices", C.c_void_p),
                ("opaque_layer", GBufferTarget), ("transmissive_layer", GBufferTarget), ("pyramid", Pyramid),
                ("hdr", C.c_void_p), ("hdr_format", C.c_int32), ("bgra", C.c_int32),
                ("tonemap", C.POINTER(TonemapParams)), ("ldr_out", C.c_void_p)]


class TextureDesc(C.Structure):  # include/tr_shade.h tr_texture_desc
    _fields_ = [
        ("rgba8", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("srgb", C.c_uint32),
        ("_reserved", C.c_uint32),
    ]


class TextureLayout(C.Structure):  # include/tr_shade.h tr_texture_layout
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("levels", C.c_uint32),
        ("srgb", C.c_uint32),
        ("level_offset", C.c_uint32 * MAX_MIP_LEVELS),
        ("total_texels", C.c_uint32),
    ]


assert C.sizeof(PushConstants) == 96
assert C.sizeof(Uniforms) == 96 and Uniforms.sun_dir.offset == 32 and Uniforms.ggx_lut_texture_index.offset == 84
assert C.sizeof(MaterialInfo) == 160 and MaterialInfo.attenuation_colour.offset == 112
assert MaterialInfo.specular_colour_factor.offset == 144 and MaterialInfo.diffuse_factor.offset == 48
assert C.sizeof(Light) == 48 and C.sizeof(ClusterAabb) == 32


# --------------------------------------------------------------------------- host helpers
def mip_levels_for_size(width: int, height: int) -> int:
    """src/main.rs:2590-2592: (min(w,h) as f32).log2() as u32 + 1."""
    return int(f32(np.log2(f32(min(width, height))))) + 1


def pyramid_layout(width: int, height: int):
    """(levels, [(offset_texels, w, h)...], total_texels): tr_pyramid's packed layout."""
    levels = min(mip_levels_for_size(width, height), MAX_MIP_LEVELS)
    out, off = [], 0
    for l in range(levels):
        w, h = max(width >> l, 1), max(height >> l, 1)
        out.append((off, w, h))
        off += w * h
    return levels, out, off


def perspective_matrix_reversed(width: int, height: int) -> np.ndarray:
    """src/main.rs:39-54. Returns a 4x4 float32 array indexed [column][row] (glam column-major)."""
    aspect_ratio = f32(f32(width) / f32(height))
    vertical_fov = f32(f32(59.0) * f32(f32(np.pi) / f32(180.0)))
    focal_length = f32(f32(1.0) / f32(np.tan(f32(vertical_fov / f32(2.0)))))
    a = f32(Z_NEAR / f32(Z_FAR - Z_NEAR))
    b = f32(Z_FAR * a)
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = f32(focal_length / aspect_ratio)
    m[1, 1] = -focal_length
    m[2, 2] = a
    m[2, 3] = f32(-1.0)
    m[3, 2] = b
    return m


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    return (v * f32(f32(1.0) / f32(np.sqrt(f32(np.dot(v, v)))))).astype(f32)


def look_at_rh(eye, center, up) -> np.ndarray:
    """glam Mat4::look_at_rh as used at src/main.rs:519-523; [column][row] float32."""
    eye = np.asarray(eye, dtype=f32)
    f = _normalize(np.asarray(center, dtype=f32) - eye)
    s = _normalize(np.cross(f, np.asarray(up, dtype=f32)).astype(f32))
    u = np.cross(s, f).astype(f32)
    m = np.zeros((4, 4), dtype=f32)
    m[0] = (s[0], u[0], -f[0], 0.0)
    m[1] = (s[1], u[1], -f[1], 0.0)
    m[2] = (s[2], u[2], -f[2], 0.0)
    m[3] = (-np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye), 1.0)
    return m.astype(f32)


def mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Column-major [column][row] product a*b (float32)."""
    # element (col j, row i) = sum_k a[k][i] * b[j][k]
    return np.einsum("ki,jk->ji", a.astype(f32), b.astype(f32)).astype(f32)


def inverse_perspective(width: int, height: int) -> np.ndarray:
    """perspective_matrix.inverse() (src/main.rs:1506) in [column][row] storage; closed form of the matrix of
    perspective_matrix_reversed."""
    p = perspective_matrix_reversed(width, height).astype(np.float64)
    inv = np.linalg.inv(p.T).T     # p is stored [column][row]; invert the mathematical matrix
    return inv.astype(f32)


def view_rotation_inverse(view: np.ndarray) -> np.ndarray:
    """camera_rotation.inverse() (src/main.rs:1788) as a quaternion (x, y, z, w): the rotation part of the view
    matrix."""
    r = np.array([[view[c][r_] for c in range(3)] for r_ in range(3)], dtype=np.float64)
    w = np.sqrt(max(0.0, 1.0 + r[0, 0] + r[1, 1] + r[2, 2])) / 2.0
    return np.array([(r[2, 1] - r[1, 2]) / (4 * w), (r[0, 2] - r[2, 0]) / (4 * w), (r[1, 0] - r[0, 1]) / (4 * w), w],
                    dtype=f32)


def sun_as_normal(pitch=1.1, yaw=4.8) -> np.ndarray:
    """Sun::as_normal (src/main.rs:2715-2722) with the start-up pitch/yaw of src/main.rs:531-534."""
    pitch, yaw = f32(pitch), f32(yaw)
    return np.array([f32(np.cos(pitch)) * f32(np.sin(yaw)), f32(np.sin(pitch)),
                     f32(np.cos(pitch)) * f32(np.cos(yaw))], dtype=f32)


def default_camera():
    """Initial dolly rig (src/main.rs:511-523): position (0,3,1), pitch -15 deg, yaw 0.
    Returns (eye, view_matrix[column][row])."""
    eye = np.array([0.0, 3.0, 1.0], dtype=f32)
    pitch = f32(np.radians(-15.0))
    forward = np.array([0.0, np.sin(pitch), -np.cos(pitch)], dtype=f32)  # yaw 0 looks down -Z
    return eye, look_at_rh(eye, eye + forward, np.array([0.0, 1.0, 0.0], dtype=f32))


def make_push_constants(width: int, height: int, eye=None, view=None) -> PushConstants:
    """proj_view = perspective * view (src/main.rs:1194-1196) + view_position + framebuffer size."""
    if eye is None or view is None:
        eye, view = default_camera()
    pv = mat4_mul(perspective_matrix_reversed(width, height), view)
    pc = PushConstants()
    pc.proj_view = (C.c_float * 16)(*[float(x) for x in pv.reshape(-1)])
    pc.view_position = (C.c_float * 3)(*[float(x) for x in eye])
    pc.framebuffer_size = (C.c_uint32 * 2)(width, height)
    pc.acceleration_structure_address = 0
    return pc


def make_uniforms(width: int, height: int, sun_dir=None, sun_intensity=(3.0, 3.0, 3.0), debug_clusters=0) -> Uniforms:
    """The Uniforms block of src/main.rs:536-552."""
    u = Uniforms()
    u.light_clustering_coefficients = LightClusterCoefficients.new()
    sd = sun_as_normal() if sun_dir is None else np.asarray(sun_dir, dtype=f32)
    u.sun_dir = (C.c_float * 3)(*[float(x) for x in sd])
    u.sun_intensity = (C.c_float * 3)(*[float(f32(x)) for x in sun_intensity])
    u.cluster_size_in_pixels = (C.c_float * 2)(float(f32(width) / f32(NUM_CLUSTERS_X)),
                                               float(f32(height) / f32(NUM_CLUSTERS_Y)))
    u.num_clusters = (C.c_uint32 * 2)(NUM_CLUSTERS_X, NUM_CLUSTERS_Y)
    u.debug_clusters = int(debug_clusters)
    u.ggx_lut_texture_index = 0
    return u


def default_lights(spotlights: bool = False):
    """The hard-coded lights of src/main.rs:450-476."""
    lights = [Light.new_point((0.0, 0.8, 0.0), (1.0, 0.0, 0.0), 5.0),
              Light.new_point((8.0, 0.8, 0.0), (0.0, 1.0, 0.0), 10.0)]
    if spotlights:
        lights.append(Light.new_spot((0.0, 4.0, 0.0), (1.0, 1.0, 0.5), 50.0, (0.0, 0.0, 1.0), 0.7, 0.8))
        lights.append(Light.new_spot((0.0, 4.0, 0.0), (1.0, 1.0, 0.5), 50.0,
                                     (float(f32(np.sin(f32(np.pi)))), 0.0, float(f32(np.cos(f32(np.pi))))), 0.7, 0.8))
    return lights


def as_ctypes_array(items, ctype):
    arr = (ctype * len(items))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr
