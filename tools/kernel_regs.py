#!/usr/bin/env python3
"""Register / scratch use of the shading kernels of a library build: python tools/kernel_regs.py LIB.so [...]"""
import os, re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_object(path):
    data = open(path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, size, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return data[i + o:i + o + size]
    raise SystemExit("no gfx950 code object")


for lib in sys.argv[1:]:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(lib)); f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    print(lib)
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if "shade_kernel" not in name and "shade_lc" not in name:
            continue
        field = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))   # noqa: E731
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("HIP_vector_type<unsigned int, 2u>", "uint2").replace("HIP_vector_type<float, 4u>", "float4").replace("void tr::", "").replace("(tr::tr_launch)", "")
        print(f"  {dem:52s} vgpr {field('vgpr_count'):3d} sgpr {field('sgpr_count'):3d} spill {field('vgpr_spill_count')}/{field('sgpr_spill_count')} scratch {field('private_segment_fixed_size')}")
