"""The register allocations the design rests on, read from the built library's own gfx950 code object (no GPU, no
compile): DESIGN.md §3.1 — the headline kernel holds 8 waves per SIMD only at <= 64 VGPRs, and every structural change
of the pixel moves the allocation by +-2; scratch spills cost more than they buy (116 vs 93 us with 22 spills)."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from transmission_renderer_amd import _lib

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _code_object(path):
    """The gfx950 entry of the clang offload bundle in .hip_fatbin."""
    data = open(path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in the library"
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, size, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return data[i + o:i + o + size]
    raise AssertionError("no gfx950 code object in the library")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} is missing: run __graft_entry__.build() first")
    if not os.path.exists(READELF) or shutil.which("c++filt") is None:
        pytest.skip("llvm-readelf / c++filt not available")
    co = tmp_path_factory.mktemp("co") / "tr_shade_gfx950.co"
    co.write_bytes(_code_object(_lib.LIB_PATH))
    notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        field = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))   # noqa: E731
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("HIP_vector_type<unsigned int, 2u>", "uint2").replace("HIP_vector_type<float, 4u>", "float4")
        out[dem] = {"vgpr": field("vgpr_count"), "sgpr": field("sgpr_count"), "vgpr_spill": field("vgpr_spill_count"),
                    "sgpr_spill": field("sgpr_spill_count"), "scratch": field("private_segment_fixed_size"),
                    "lds": field("group_segment_fixed_size")}
    return out


def _shade(kernels, transmissive, out_t, tex, vis):
    key = f"void tr::shade_kernel<{'true' if transmissive else 'false'}, {out_t}, {tex}, {'true' if vis else 'false'}>(tr::tr_launch)"
    assert key in kernels, f"{key} not in the library: {sorted(k for k in kernels if 'shade_kernel' in k)}"
    return kernels[key]


def test_no_kernel_spills_or_uses_scratch(kernels):
    assert len(kernels) >= 40
    bad = {k: v for k, v in kernels.items() if v["vgpr_spill"] or v["sgpr_spill"] or v["scratch"]}
    assert not bad, bad


def test_headline_kernel_holds_eight_waves_per_simd(kernels):
    # gfx950: 512 VGPRs per SIMD lane, granule 8: <= 64 -> 8 waves, <= 72 -> 7, <= 80 -> 6, <= 96 -> 5, <= 128 -> 4
    assert _shade(kernels, True, "uint2", 0, False)["vgpr"] <= 64      # what bench.py times
    assert _shade(kernels, False, "uint2", 0, False)["vgpr"] <= 64
    assert _shade(kernels, False, "uint2", 0, True)["vgpr"] <= 64      # the frame recorder's untextured passes
    assert _shade(kernels, True, "uint2", 0, True)["vgpr"] <= 64       # (with an occupancy hint: TR_WAVES_ATTR)


def test_textured_launch_classes_keep_their_occupancy(kernels):
    for vis in (False, True):
        assert _shade(kernels, False, "uint2", 1, vis)["vgpr"] <= 72   # lite class, opaque: 7 waves
        assert _shade(kernels, True, "uint2", 1, vis)["vgpr"] <= 80    # lite class, transmissive: 6 waves
        for tex in (2, 3):                                             # every class incl. the full one (3: its usual glTF slot set only)
            for transmissive in (False, True):                         # 6 waves: tile inputs and late-read factors wait in LDS,
                k = _shade(kernels, transmissive, "uint2", tex, vis)   # whose footprint must leave room for 24 waves per CU
                assert k["vgpr"] <= 80, (tex, transmissive, vis, k)
                assert k["lds"] <= 5 * 1280, (tex, transmissive, vis, k)
        assert _shade(kernels, False, "uint2", 3, vis)["vgpr"] <= 80   # ... its base-colour + metallic-roughness + normal build: 6 waves
        assert _shade(kernels, True, "uint2", 3, vis)["vgpr"] <= 96    #     transmissive: 5 waves
        # five one-wave workgroups per SIMD = 20 per CU must fit the CU's 160 KB of LDS
        assert _shade(kernels, False, "uint2", 2, vis)["lds"] * 20 <= 160 * 1024
