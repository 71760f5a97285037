"""Screen-tile sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference is single-GPU (one VkQueue, src/main.rs:243); this is new surface.
Pixels of both shading passes are independent given replicated read-only inputs (tables, LUT,
opaque pyramid), so the frame is cut into contiguous row bands, one per rank, and no collective
is needed while shading.  Two exchanges exist, both all-gathers of whole row bands over RCCL/xGMI
("nccl" backend) — or gloo on CPU tensors in the tests:

  * `allgather_frame`: the final composite — every rank ends with the whole RGBA16F frame.
  * `allgather_mip0`: for the full opaque -> mips -> transmissive pipeline the transmissive pass
    samples the *whole* opaque pyramid at refracted coordinates, so level 0 has to be gathered
    (and the 10.67 B/px pyramid built redundantly per rank) between the two passes.

Bands are contiguous in memory (row-major frame), so both are in-place all_gather_into_tensor
calls with no packing kernel.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def band_rows(height: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [y0, y1) of `rank`.  Equal bands (height must divide): in-place all-gather needs equal counts.
    With textured materials a band must also hold whole 2x2 pixel quads (height / world even): the shading
    entry points refuse a rect that cuts quads (include/tr_shade.h tr_upload_textures)."""
    if height % world:
        raise ValueError(f"frame height {height} is not a multiple of world size {world}")
    rows = height // world
    return rank * rows, (rank + 1) * rows


def band_rect(width: int, height: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    y0, y1 = band_rows(height, world, rank)
    return 0, y0, width, y1


def _allgather_rows_inplace(frame: torch.Tensor, world: int, group=None) -> None:
    """frame: (H, W, C) contiguous, every rank has written its own band; afterwards all bands are everywhere."""
    assert frame.is_contiguous()
    rank = dist.get_rank(group)
    flat = frame.view(-1)
    if flat.numel() % world:
        raise ValueError("frame does not split into equal bands")
    n = flat.numel() // world
    mine = flat[rank * n:(rank + 1) * n]
    backend = dist.get_backend(group)
    if backend == "gloo":  # gloo has no in-place variant on views of the output: gather into a list of views
        outs = [flat[i * n:(i + 1) * n] for i in range(world)]
        dist.all_gather(outs, mine.clone(), group=group)
    else:
        dist.all_gather_into_tensor(flat, mine, group=group)


def allgather_frame(hdr: torch.Tensor, world: int, group=None) -> None:
    """Composite: all-gather the row bands of the RGBA16F/32F frame in place."""
    _allgather_rows_inplace(hdr, world, group)


def allgather_mip0(mip0: torch.Tensor, world: int, group=None) -> None:
    """Mid-frame exchange of the opaque colour (pyramid level 0) before generate_mips."""
    _allgather_rows_inplace(mip0, world, group)


def record_sharded(renderer, opaque, transmissive, uniforms, push, hdr, pyramid, world: int, rank: int, group=None,
                   composite: bool = True) -> None:
    """The hot-path slice of `record()` (src/main.rs:1969-2124) for one rank of a row-band sharded frame.

    `opaque` / `transmissive` are this rank's G-buffer tiles (origin_y = its first row).  Order:
    opaque band -> all-gather level 0 -> mip chain (replicated) -> transmissive band -> composite.
    """
    fw, fh = int(push.framebuffer_size[0]), int(push.framebuffer_size[1])
    rect = band_rect(fw, fh, world, rank)
    renderer.shade_opaque(opaque, uniforms, push, hdr, pyramid, rect)
    if world > 1:
        allgather_mip0(pyramid.level(0), world, group)
    renderer.generate_mips(pyramid)
    renderer.shade_transmission(transmissive, uniforms, push, pyramid, hdr, rect)
    if world > 1 and composite:
        allgather_frame(hdr, world, group)
