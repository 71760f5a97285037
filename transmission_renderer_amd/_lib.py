"""ctypes binding of libtr_shade.so (include/tr_shade.h).

There is no fallback: if the HIP library has not been built (`python __graft_entry__.py` or
`make -C transmission_renderer_amd/csrc`) or cannot be loaded, importing the shading path raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtr_shade.so")

# Every symbol include/tr_shade.h declares (tests check the .so exports exactly these).
SYMBOLS = (
    "tr_abi_version", "tr_status_string", "tr_last_hip_error", "tr_context_create", "tr_context_destroy",
    "tr_pyramid_layout", "tr_upload_materials", "tr_upload_lights", "tr_set_cluster_tables",
    "tr_upload_ggx_lut", "tr_upload_textures", "tr_texture_get_layout", "tr_download_texture", "tr_frustum_culling", "tr_demultiplex_draws",
    "tr_upload_geometry", "tr_rasterize", "tr_draw_scene",
    "tr_write_cluster_data", "tr_assign_lights_to_clusters", "tr_shade_opaque",
    "tr_generate_mips", "tr_shade_transmission", "tr_lottes_defaults", "tr_bake_lottes_params", "tr_tonemap", "tr_record_frame", "tr_record_frame_timed",
    "tr_basic_brdf", "tr_transmission_btdf", "tr_ibl_volume_refraction", "tr_ibl_volume_refraction_requests", "tr_ibl_volume_refraction_resolve",
    "tr_light_direction_and_attenuation", "tr_d_ggx",
    "tr_v_smith_ggx_correlated", "tr_fresnel_schlick", "tr_compute_f0", "tr_get_depth_slice", "tr_depth_slice_thresholds",
    "tr_band_rows", "tr_comm_unique_id", "tr_comm_create", "tr_comm_from_nccl", "tr_comm_destroy", "tr_comm_last_error",
    "tr_allgather_frame", "tr_set_strips", "tr_strip_of_rank", "tr_allgather_strips",
    "tr_generate_mips_from", "tr_generate_mips_band", "tr_set_tap_window", "tr_exchange_halo", "tr_halo_rows", "tr_tonemap_rgb8",
    "tr_update_lights", "tr_update_instances", "tr_shade_opaque_pyramid", "tr_comm_query",
)

_lib = None


class TrError(RuntimeError):
    def __init__(self, status: int, where: str, hip_error: int = 0):
        self.status = status
        self.hip_error = hip_error
        msg = load().tr_status_string(status).decode()
        super().__init__(f"{where}: {msg} (status {status}, hipError {hip_error})")


def load() -> C.CDLL:
    """Loads libtr_shade.so once.  torch is imported first so both share one HIP runtime
    (torch bundles libamdhip64.so.7; the soname is resolved to the copy already in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "This package has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads the HIP runtime this process will use)
    except Exception:  # pragma: no cover - torch is plumbing, the library itself does not need it
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    lib.tr_abi_version.restype = u32
    lib.tr_status_string.restype = C.c_char_p
    lib.tr_status_string.argtypes = [i32]
    lib.tr_last_hip_error.restype = i32
    lib.tr_last_hip_error.argtypes = [vp]
    lib.tr_context_create.restype = i32
    lib.tr_context_create.argtypes = [i32, C.POINTER(vp)]
    lib.tr_context_destroy.restype = i32
    lib.tr_context_destroy.argtypes = [vp]
    lib.tr_pyramid_layout.restype = i32
    lib.tr_pyramid_layout.argtypes = [u32, u32, C.POINTER(wire.Pyramid), C.POINTER(C.c_size_t)]
    lib.tr_upload_materials.restype = i32
    lib.tr_upload_materials.argtypes = [vp, C.POINTER(wire.MaterialInfo), u32, vp]
    lib.tr_upload_lights.restype = i32
    lib.tr_upload_lights.argtypes = [vp, C.POINTER(wire.Light), u32, vp]
    lib.tr_set_cluster_tables.restype = i32
    lib.tr_set_cluster_tables.argtypes = [vp, vp, vp, u32]
    lib.tr_upload_ggx_lut.restype = i32
    lib.tr_upload_ggx_lut.argtypes = [vp, C.c_void_p, u32, u32, vp]
    lib.tr_upload_textures.restype = i32
    lib.tr_upload_textures.argtypes = [vp, C.POINTER(wire.TextureDesc), u32, vp]
    lib.tr_texture_get_layout.restype = i32
    lib.tr_texture_get_layout.argtypes = [vp, u32, C.POINTER(wire.TextureLayout)]
    lib.tr_download_texture.restype = i32
    lib.tr_download_texture.argtypes = [vp, u32, vp, C.c_size_t, vp]
    lib.tr_frustum_culling.restype = i32
    lib.tr_frustum_culling.argtypes = [vp, vp, u32, vp, u32, C.POINTER(wire.CullingPushConstants), vp, vp]
    lib.tr_demultiplex_draws.restype = i32
    lib.tr_demultiplex_draws.argtypes = [vp, vp, u32, vp, vp, C.POINTER(vp * 4), vp]
    lib.tr_upload_geometry.restype = i32
    lib.tr_upload_geometry.argtypes = [vp, C.POINTER(wire.GeometryDesc), vp]
    lib.tr_rasterize.restype = i32
    lib.tr_rasterize.argtypes = [vp, vp, C.POINTER(vp * 4), C.POINTER(wire.PushConstants), C.POINTER(wire.GBufferTarget),
                                 C.POINTER(wire.GBufferTarget), vp]
    lib.tr_draw_scene.restype = i32
    lib.tr_draw_scene.argtypes = [vp, C.POINTER(wire.CullingPushConstants), C.POINTER(wire.PushConstants),
                                  C.POINTER(wire.GBufferTarget), C.POINTER(wire.GBufferTarget), vp]
    lib.tr_write_cluster_data.restype = i32
    lib.tr_write_cluster_data.argtypes = [vp, C.POINTER(wire.Uniforms), C.POINTER(C.c_float * 16),
                                          C.POINTER(C.c_uint32 * 2), vp, vp]
    lib.tr_assign_lights_to_clusters.restype = i32
    lib.tr_assign_lights_to_clusters.argtypes = [vp, C.POINTER(C.c_float * 16), C.POINTER(C.c_float * 4), vp, u32, vp,
                                                 vp, vp]
    lib.tr_shade_opaque.restype = i32
    lib.tr_shade_opaque.argtypes = [vp, C.POINTER(wire.GBuffer), C.POINTER(wire.Uniforms),
                                    C.POINTER(wire.PushConstants), vp, i32, vp, wire.Rect, vp]
    lib.tr_generate_mips.restype = i32
    lib.tr_generate_mips.argtypes = [vp, C.POINTER(wire.Pyramid), vp]
    lib.tr_shade_transmission.restype = i32
    lib.tr_shade_transmission.argtypes = [vp, C.POINTER(wire.GBuffer), C.POINTER(wire.Uniforms),
                                          C.POINTER(wire.PushConstants), C.POINTER(wire.Pyramid), vp, i32,
                                          wire.Rect, vp]
    lib.tr_lottes_defaults.restype = i32
    lib.tr_lottes_defaults.argtypes = [C.POINTER(wire.LottesParams)]
    lib.tr_bake_lottes_params.restype = i32
    lib.tr_bake_lottes_params.argtypes = [C.POINTER(wire.LottesParams), C.POINTER(wire.TonemapParams)]
    lib.tr_tonemap.restype = i32
    lib.tr_tonemap.argtypes = [vp, vp, u32, u32, C.POINTER(wire.TonemapParams), vp, i32, vp]
    lib.tr_record_frame.restype = i32
    lib.tr_record_frame.argtypes = [vp, C.POINTER(wire.FrameDesc), vp]
    lib.tr_record_frame_timed.restype = i32
    lib.tr_record_frame_timed.argtypes = [vp, C.POINTER(wire.FrameDesc), vp, C.POINTER(wire.FrameZone), u32, C.POINTER(u32)]
    for name, nptr in (("tr_basic_brdf", 1), ("tr_transmission_btdf", 1), ("tr_light_direction_and_attenuation", 2),
                       ("tr_d_ggx", 2), ("tr_v_smith_ggx_correlated", 3), ("tr_fresnel_schlick", 3), ("tr_compute_f0", 3)):
        fn = getattr(lib, name)   # (ctx, <nptr input arrays>, count, out, stream)
        fn.restype = i32
        fn.argtypes = [vp] + [vp] * nptr + [u32, vp, vp]
    lib.tr_get_depth_slice.restype = i32
    lib.tr_get_depth_slice.argtypes = [vp, C.POINTER(wire.LightClusterCoefficients), vp, u32, vp, vp]
    lib.tr_depth_slice_thresholds.restype = i32
    lib.tr_depth_slice_thresholds.argtypes = [C.POINTER(wire.LightClusterCoefficients), C.POINTER(C.c_float), C.POINTER(u32)]
    lib.tr_band_rows.restype = i32
    lib.tr_band_rows.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    lib.tr_comm_unique_id.restype = i32
    lib.tr_comm_unique_id.argtypes = [C.POINTER(C.c_uint8 * 128)]
    lib.tr_comm_create.restype = i32
    lib.tr_comm_create.argtypes = [vp, C.POINTER(C.c_uint8 * 128), u32, u32, C.POINTER(vp)]
    lib.tr_comm_from_nccl.restype = i32
    lib.tr_comm_from_nccl.argtypes = [vp, u32, u32, C.POINTER(vp)]
    lib.tr_comm_destroy.restype = i32
    lib.tr_comm_destroy.argtypes = [vp]
    lib.tr_comm_last_error.restype = i32
    lib.tr_comm_last_error.argtypes = [vp]
    lib.tr_allgather_frame.restype = i32
    lib.tr_allgather_frame.argtypes = [vp, vp, vp, u32, u32, i32, vp]
    lib.tr_set_strips.restype = i32
    lib.tr_set_strips.argtypes = [vp, u32, u32, u32]
    lib.tr_strip_of_rank.restype = i32
    lib.tr_strip_of_rank.argtypes = [u32, u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.tr_allgather_strips.restype = i32
    lib.tr_allgather_strips.argtypes = [vp, vp, vp, u32, u32, u32, i32, vp]
    lib.tr_generate_mips_from.restype = i32
    lib.tr_generate_mips_from.argtypes = [vp, C.POINTER(wire.Pyramid), u32, vp]
    lib.tr_generate_mips_band.restype = i32
    lib.tr_generate_mips_band.argtypes = [vp, C.POINTER(wire.Pyramid), u32, u32, vp]
    lib.tr_set_tap_window.restype = i32
    lib.tr_set_tap_window.argtypes = [vp, u32, u32, vp]
    lib.tr_exchange_halo.restype = i32
    lib.tr_exchange_halo.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp]
    lib.tr_halo_rows.restype = i32
    lib.tr_halo_rows.argtypes = [u32, u32, u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.tr_tonemap_rgb8.restype = i32
    lib.tr_tonemap_rgb8.argtypes = [vp, vp, u32, u32, C.POINTER(wire.TonemapParams), vp, i32, vp]
    lib.tr_update_lights.restype = i32
    lib.tr_update_lights.argtypes = [vp, u32, u32, C.POINTER(wire.Light), vp]
    lib.tr_update_instances.restype = i32
    lib.tr_update_instances.argtypes = [vp, u32, u32, vp, vp]
    lib.tr_shade_opaque_pyramid.restype = i32
    lib.tr_shade_opaque_pyramid.argtypes = [vp, C.POINTER(wire.GBuffer), C.POINTER(wire.Uniforms), C.POINTER(wire.PushConstants), vp, i32,
                                            C.POINTER(wire.Pyramid), wire.Rect, C.POINTER(u32), vp]
    lib.tr_comm_query.restype = i32
    lib.tr_comm_query.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    lib.tr_ibl_volume_refraction.restype = i32
    lib.tr_ibl_volume_refraction.argtypes = [vp, vp, u32, C.POINTER(wire.Pyramid), vp, vp]
    lib.tr_ibl_volume_refraction_requests.restype = i32
    lib.tr_ibl_volume_refraction_requests.argtypes = [vp, vp, u32, vp, vp]
    lib.tr_ibl_volume_refraction_resolve.restype = i32
    lib.tr_ibl_volume_refraction_resolve.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    if lib.tr_abi_version() != 1:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.tr_abi_version()} != 1")
    _lib = lib
    return lib
