"""Screen-tile sharding across the GPUs of one node (one process per GPU).

The reference is single-GPU (one VkQueue, src/main.rs:243); this is new surface.
Pixels of both shading passes are independent given replicated read-only inputs (tables, LUT,
opaque pyramid), so the frame is cut into contiguous row bands, one per rank, and no collective
is needed while shading.  Two exchanges exist, both in-place all-gathers of whole row bands over RCCL/xGMI:

  * the composite: every rank ends with the whole RGBA16F frame;
  * level 0 of the opaque pyramid between the two passes (the transmissive pass samples the *whole*
    pyramid at refracted coordinates; the 10.67 B/px chain is then built redundantly per rank).

Band arithmetic is the library's (`tr_band_rows`, include/tr_shade.h): bands of ceil(H / N) rows rounded up to the
4-row wave tile, the last ones clipped to the frame; buffers that are gathered hold N * rows_per_rank rows (>= H).
On a GPU the gathers go through `tr_allgather_frame` — the library's own RCCL communicator, the entry point a
non-Python host calls too — and through torch.distributed otherwise (gloo on CPU tensors in the tests; also the
fallback when the library's communicator cannot be created).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib, wire


def band_rows(height: int, world: int, rank: int) -> Tuple[int, int, int]:
    """(rows_per_rank, y0, y1) of `rank`: tr_band_rows."""
    rows, y0, y1 = C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = _lib.load().tr_band_rows(int(height), int(world), int(rank), C.byref(rows), C.byref(y0), C.byref(y1))
    if st != 0:
        raise ValueError(f"tr_band_rows({height}, {world}, {rank}): status {st}")
    return rows.value, y0.value, y1.value


def band_rect(width: int, height: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    _, y0, y1 = band_rows(height, world, rank)
    return 0, y0, width, y1


def padded_rows(height: int, world: int) -> int:
    """Rows a gathered buffer must hold."""
    return band_rows(height, world, 0)[0] * world


def strips_of_rank(height: int, strip_rows: int, world: int, rank: int):
    """[(y0, y1), ...]: the strips of `rank` (tr_strip_of_rank): strip s of the frame belongs to rank s % world."""
    out, k = [], 0
    y0, y1 = C.c_uint32(), C.c_uint32()
    while True:
        st = _lib.load().tr_strip_of_rank(int(height), int(strip_rows), int(world), int(rank), k, C.byref(y0), C.byref(y1))
        if st != 0:
            raise ValueError(f"tr_strip_of_rank({height}, {strip_rows}, {world}, {rank}, {k}): status {st}")
        if y0.value == y1.value:
            return out
        out.append((y0.value, y1.value))
        k += 1


class Compositor:
    """In-place all-gather of equal row bands of a (rows_per_rank * world, W, C) device or host tensor."""

    def __init__(self, world: int, rank: int, renderer=None, group=None, prefer_library: bool = True,
                 single_rank_comm: bool = False):
        """single_rank_comm: make the library's communicator for world == 1 as well (the one-GPU rehearsal of the
        multi-GPU path: bench.py --rehearse-distributed)."""
        self.world, self.rank, self.group = world, rank, group
        self.renderer = renderer
        self._comm = C.c_void_p()
        self.backend = "none" if world == 1 else "torch.distributed:" + dist.get_backend(group)
        if (world > 1 or single_rank_comm) and renderer is not None and prefer_library:
            self._try_library_comm()

    def _try_library_comm(self):
        """The library's RCCL communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes."""
        lib = self.renderer.lib
        ident = (C.c_uint8 * 128)()
        ok = 1
        if self.rank == 0:
            ok = int(lib.tr_comm_unique_id(C.byref(ident)) == 0)
        box = [bytes(ident) if ok else None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        if box[0] is not None:
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            st = lib.tr_comm_create(self.renderer._ctx, C.byref(ident), self.world, self.rank, C.byref(self._comm))
            ok = int(st == 0)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.renderer.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)       # every rank or none
        if int(flag.item()) == 1:
            self.backend = "tr_allgather_frame (RCCL)"
        else:
            self.close()

    def close(self):
        if self._comm.value:
            self.renderer.lib.tr_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def allgather_rows(self, frame: torch.Tensor) -> None:
        """frame: (rows_per_rank * world, W, C) contiguous; every rank has written its own band."""
        if self.world == 1 and not self._comm.value:
            return
        assert frame.is_contiguous() and frame.shape[0] % self.world == 0, "gathered buffers hold world * rows_per_rank rows"
        rows = frame.shape[0] // self.world
        if self._comm.value:
            fmt = {torch.float16: wire.FORMAT_RGBA16F, torch.float32: wire.FORMAT_RGBA32F, torch.uint8: wire.FORMAT_RGBA8}[frame.dtype]
            assert frame.shape[2] == 4
            st = self.renderer.lib.tr_allgather_frame(self.renderer._ctx, self._comm, frame.data_ptr(), int(frame.shape[1]),
                                                      rows, fmt, torch.cuda.current_stream().cuda_stream)
            if st != 0:
                raise _lib.TrError(st, "tr_allgather_frame", self.renderer.lib.tr_comm_last_error(self._comm))
            return
        flat = frame.view(-1)
        n = flat.numel() // self.world
        mine = flat[self.rank * n:(self.rank + 1) * n]
        if dist.get_backend(self.group) == "gloo":   # gloo has no in-place variant on views of the output
            outs = [flat[i * n:(i + 1) * n] for i in range(self.world)]
            dist.all_gather(outs, mine.clone(), group=self.group)
        else:
            dist.all_gather_into_tensor(flat, mine, group=self.group)


def allgather_strips(comp: "Compositor", frame: torch.Tensor, strip_rows: int) -> None:
    """The composite of a strip-sharded frame: `frame` is (H, W, C), every rank has written its own strips in place.
    tr_allgather_strips on a GPU (one RCCL group of per-strip broadcasts); per-strip torch.distributed broadcasts
    otherwise (gloo on host tensors in the tests)."""
    if comp.world == 1 and not comp._comm.value:
        return
    assert frame.is_contiguous()
    h, w = int(frame.shape[0]), int(frame.shape[1])
    if comp._comm.value:
        fmt = {torch.float16: wire.FORMAT_RGBA16F, torch.float32: wire.FORMAT_RGBA32F, torch.uint8: wire.FORMAT_RGBA8}[frame.dtype]
        st = comp.renderer.lib.tr_allgather_strips(comp.renderer._ctx, comp._comm, frame.data_ptr(), w, h, int(strip_rows), fmt,
                                                   torch.cuda.current_stream().cuda_stream)
        if st != 0:
            raise _lib.TrError(st, "tr_allgather_strips", comp.renderer.lib.tr_comm_last_error(comp._comm))
        return
    works = []
    for s, y in enumerate(range(0, h, strip_rows)):
        works.append(dist.broadcast(frame[y:min(y + strip_rows, h)], src=dist.get_global_rank(comp.group, s % comp.world)
                                    if comp.group is not None else s % comp.world, group=comp.group, async_op=True))
    for wk in works:
        wk.wait()


def record_sharded_strips(renderer, opaque, transmissive, uniforms, push, hdr, pyramid, compositor: "Compositor",
                          strip_rows: int = 64, composite: bool = True) -> None:
    """record_sharded with rank-interleaved strips instead of one row band per rank (frames whose cost is uneven over
    the screen): `opaque` / `transmissive` cover the whole frame, `hdr` and `pyramid` are whole-frame buffers; every rank
    shades strips rank, rank + world, ... in place (one launch per pass), level 0 of the pyramid and the frame are
    exchanged strip by strip."""
    world, rank = compositor.world, compositor.rank
    fw, fh = int(push.framebuffer_size[0]), int(push.framebuffer_size[1])
    renderer.set_strips(strip_rows if world > 1 else 0, world, rank)
    try:
        mine = world == 1 or bool(strips_of_rank(fh, strip_rows, world, rank))
        if mine:
            renderer.shade_opaque(opaque, uniforms, push, hdr, pyramid, (0, 0, fw, fh))
        if world > 1:
            allgather_strips(compositor, pyramid.level(0), strip_rows)
        renderer.generate_mips(pyramid)
        if mine:
            renderer.shade_transmission(transmissive, uniforms, push, pyramid, hdr, (0, 0, fw, fh))
        if world > 1 and composite:
            allgather_strips(compositor, hdr, strip_rows)
    finally:
        renderer.set_strips(0, 1, 0)


def record_sharded(renderer, opaque, transmissive, uniforms, push, hdr, pyramid, compositor: Compositor,
                   composite: bool = True) -> None:
    """The hot-path slice of `record()` (src/main.rs:1969-2124) for one rank of a row-band sharded frame.

    `opaque` / `transmissive` are this rank's G-buffer tiles (origin_y = its first row); `hdr` holds
    padded_rows(H, world) rows and `pyramid` was made with level0_rows = the same.  Order:
    opaque band -> all-gather level 0 -> mip chain (replicated) -> transmissive band -> composite.
    """
    world, rank = compositor.world, compositor.rank
    fw, fh = int(push.framebuffer_size[0]), int(push.framebuffer_size[1])
    rect = band_rect(fw, fh, world, rank)
    if rect[3] > rect[1]:       # (a band can be empty when the height is far from a multiple of world * 4)
        renderer.shade_opaque(opaque, uniforms, push, hdr, pyramid, rect)
    if world > 1:
        compositor.allgather_rows(pyramid.level0_padded())
    renderer.generate_mips(pyramid)
    if rect[3] > rect[1]:
        renderer.shade_transmission(transmissive, uniforms, push, pyramid, hdr, rect)
    if world > 1 and composite:
        compositor.allgather_rows(hdr)
