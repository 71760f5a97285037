"""The glam-pbr shading API (glam-pbr/src/lib.rs), batched on the MI355X.

The reference's `glam-pbr` crate exports pure per-sample functions — `basic_brdf(BasicBrdfParams) -> BrdfResult`
(:377), `transmission_btdf(MaterialParams, Normal, View, Light) -> Vec3` (:200), `ibl_volume_refraction(params,
framebuffer_sampler, ggx_lut_sampler) -> Vec3` (:292), `light_direction_and_attenuation` (:12), `d_ggx` (:101),
`v_smith_ggx_correlated` (:114), `fresnel_schlick` (:137), `compute_f0` (:454) — called by the shaders once per
pixel and light.  Here the same names take arrays: element i of the result is the reference function applied to
element i of the arguments, evaluated on the GPU by libtr_shade.so (tr_basic_brdf & co, include/tr_shade.h) with the
shading passes' own device code.  Arguments are numpy structured arrays (wire.*_DTYPE; uploaded for the call) or
CUDA tensors holding the same packed records; results are CUDA tensors.  There is no CPU path.

    api = GlamPbr(renderer)
    res = api.basic_brdf(params)            # params: wire.BASIC_BRDF_PARAMS_DTYPE[n] -> (n, 6) float32 (diffuse, specular)
"""
from __future__ import annotations

import numpy as np
import torch

from . import wire
from .renderer import OpaquePyramid, TransmissionRenderer


class GlamPbr:
    def __init__(self, renderer: TransmissionRenderer):
        self.r = renderer
        self.lib = renderer.lib

    # ---- plumbing
    def _records(self, a, dtype: np.dtype) -> torch.Tensor:
        """-> (n, itemsize / 4) float32 CUDA tensor of packed records (integer fields keep their bits)"""
        words = dtype.itemsize // 4
        if isinstance(a, torch.Tensor):
            assert a.is_cuda and a.is_contiguous() and a.dtype == torch.float32 and a.shape[-1] == words, (a.shape, words)
            return a.reshape(-1, words)
        a = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(a.reshape(-1).view(np.float32).reshape(-1, words).copy()).to(self.r.device)

    def _floats(self, a, width: int) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            assert a.is_cuda and a.is_contiguous() and a.dtype == torch.float32
            t = a
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.r.device)
        return t.reshape(-1, width) if width > 1 else t.reshape(-1)

    def _call(self, name: str, inputs, count: int, out: torch.Tensor):
        fn = getattr(self.lib, name)
        self.r._check(fn(self.r._ctx, *[t.data_ptr() for t in inputs], count, out.data_ptr(), self.r._stream()), name)
        return out

    # ---- the API
    def basic_brdf(self, params) -> torch.Tensor:
        """basic_brdf (:377-423): BasicBrdfParams[n] -> (n, 6): BrdfResult.diffuse, BrdfResult.specular"""
        p = self._records(params, wire.BASIC_BRDF_PARAMS_DTYPE)
        out = torch.empty((p.shape[0], 6), dtype=torch.float32, device=self.r.device)
        return self._call("tr_basic_brdf", [p], p.shape[0], out)

    def transmission_btdf(self, params) -> torch.Tensor:
        """transmission_btdf (:200-233): (material_params, normal, view, light)[n] -> (n, 3)"""
        p = self._records(params, wire.TRANSMISSION_BTDF_PARAMS_DTYPE)
        out = torch.empty((p.shape[0], 3), dtype=torch.float32, device=self.r.device)
        return self._call("tr_transmission_btdf", [p], p.shape[0], out)

    def ibl_volume_refraction(self, params, framebuffer: OpaquePyramid) -> torch.Tensor:
        """ibl_volume_refraction (:292-354): IblVolumeRefractionParams[n] -> (n, 3).  The framebuffer sampler is
        `framebuffer` through clamp_sampler, the GGX LUT sampler is the LUT uploaded to the renderer — what the
        reference's caller passes as closures (shader/src/lib.rs:126-138)."""
        p = self._records(params, wire.IBL_VOLUME_REFRACTION_PARAMS_DTYPE)
        out = torch.empty((p.shape[0], 3), dtype=torch.float32, device=self.r.device)
        import ctypes as C
        self.r._check(self.lib.tr_ibl_volume_refraction(self.r._ctx, p.data_ptr(), p.shape[0], C.byref(framebuffer.desc),
                                                        out.data_ptr(), self.r._stream()), "tr_ibl_volume_refraction")
        return out

    def ibl_volume_refraction_with(self, params, framebuffer_sampler, ggx_lut_sampler) -> torch.Tensor:
        """ibl_volume_refraction<FSamp, GSamp> (:292-299) with the CALLER'S closures: `framebuffer_sampler(uv (n, 2), lod (n,))
        -> (n, 3)` and `ggx_lut_sampler(normal_dot_view (n,), perceptual_roughness (n,)) -> (n, 2)` are called once, on
        device tensors, between the function's two halves (tr_ibl_volume_refraction_requests / _resolve)."""
        p = self._records(params, wire.IBL_VOLUME_REFRACTION_PARAMS_DTYPE)
        n = p.shape[0]
        req = torch.empty((n, 5), dtype=torch.float32, device=self.r.device)
        self.r._check(self.lib.tr_ibl_volume_refraction_requests(self.r._ctx, p.data_ptr(), n, req.data_ptr(), self.r._stream()),
                      "tr_ibl_volume_refraction_requests")
        rgb = framebuffer_sampler(req[:, 0:2], req[:, 2]).to(torch.float32).contiguous()
        ab = ggx_lut_sampler(req[:, 3], req[:, 4]).to(torch.float32).contiguous()
        assert rgb.shape == (n, 3) and ab.shape == (n, 2)
        out = torch.empty((n, 3), dtype=torch.float32, device=self.r.device)
        self.r._check(self.lib.tr_ibl_volume_refraction_resolve(self.r._ctx, p.data_ptr(), n, rgb.data_ptr(), ab.data_ptr(),
                                                                out.data_ptr(), self.r._stream()), "tr_ibl_volume_refraction_resolve")
        return out

    def light_direction_and_attenuation(self, fragment_position, light_position) -> torch.Tensor:
        """(:12-23): two (n, 3) arrays -> (n, 5): direction xyz, distance, attenuation"""
        f, l = self._floats(fragment_position, 3), self._floats(light_position, 3)
        assert f.shape == l.shape
        out = torch.empty((f.shape[0], 5), dtype=torch.float32, device=self.r.device)
        return self._call("tr_light_direction_and_attenuation", [f, l], f.shape[0], out)

    def d_ggx(self, normal_dot_halfway, actual_roughness) -> torch.Tensor:
        a, b = self._floats(normal_dot_halfway, 1), self._floats(actual_roughness, 1)
        assert a.shape == b.shape
        return self._call("tr_d_ggx", [a, b], a.shape[0], torch.empty_like(a))

    def v_smith_ggx_correlated(self, normal_dot_view, normal_dot_light, actual_roughness) -> torch.Tensor:
        a, b, c = (self._floats(x, 1) for x in (normal_dot_view, normal_dot_light, actual_roughness))
        assert a.shape == b.shape == c.shape
        return self._call("tr_v_smith_ggx_correlated", [a, b, c], a.shape[0], torch.empty_like(a))

    def fresnel_schlick(self, view_dot_halfway, f0, f90) -> torch.Tensor:
        a, b, c = self._floats(view_dot_halfway, 1), self._floats(f0, 3), self._floats(f90, 3)
        assert b.shape == c.shape == (a.shape[0], 3)
        return self._call("tr_fresnel_schlick", [a, b, c], a.shape[0], torch.empty_like(b))

    def compute_f0(self, metallic, index_of_refraction, diffuse_colour) -> torch.Tensor:
        a, b, c = self._floats(metallic, 1), self._floats(index_of_refraction, 1), self._floats(diffuse_colour, 3)
        assert a.shape == b.shape and c.shape == (a.shape[0], 3)
        return self._call("tr_compute_f0", [a, b, c], a.shape[0], torch.empty_like(c))
