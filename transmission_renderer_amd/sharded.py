"""Screen-tile sharding across the GPUs of one node (one process per GPU).

The reference is single-GPU (one VkQueue, src/main.rs:243); this is new surface.
Pixels of both shading passes are independent given replicated read-only inputs (tables, LUT,
opaque pyramid), so the frame is cut into contiguous row bands, one per rank, and no collective
is needed while shading.  Two exchanges exist, both in-place all-gathers of whole row bands over RCCL/xGMI:

  * the composite: every rank ends with the whole RGBA16F frame;
  * level 0 of the opaque pyramid between the two passes (the transmissive pass samples the *whole*
    pyramid at refracted coordinates; the 10.67 B/px chain is then built redundantly per rank).

The level-0 exchange need not be a gather of everything (round 4): a pixel's refraction taps land a bounded number of
rows from the pixel, and only taps of levels 0 and 1 need fine rows at all.  `record_sharded(exchange="halo")`: every rank
builds levels 1 and 2 of its own band (tr_generate_mips_band: exact 2x2 boxes that never leave the band), exchanges
`halo` border rows of levels 0 and 1 with the ranks that own them (tr_exchange_halo: ncclSend / ncclRecv), all-gathers
level 2 (1/16 of level 0's bytes) and builds levels 3.. from it.  The transmissive pass checks every tap of levels 0 / 1
against the rows the rank holds (tr_set_tap_window) and reports the largest excursion; when there is one the frame is
redone with the full gather and the compositor's halo grows to what was needed (bytes per link: DESIGN.md 4).

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


def halo_rows_between(total_rows: int, rows_per_rank: int, world: int, owner: int, reader: int, halo: int) -> Tuple[int, int]:
    """Rows [y0, y1) of `owner`'s band of a level that `reader` receives in a halo exchange (tr_halo_rows)."""
    y0, y1 = C.c_uint32(), C.c_uint32()
    st = _lib.load().tr_halo_rows(int(total_rows), int(rows_per_rank), int(world), int(owner), int(reader), int(halo), C.byref(y0), C.byref(y1))
    if st != 0:
        raise ValueError(f"tr_halo_rows: status {st}")
    return y0.value, y1.value


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
        # a GPU frame exchanged through a host-side backend (gloo: the one-GPU rehearsal of the N > 1 path, both ranks on one
        # device, which RCCL refuses) is staged through host memory by the fallbacks below
        self.host_staged = world > 1 and renderer is not None and str(getattr(renderer, "device", "cpu")) != "cpu" \
            and dist.get_backend(group) == "gloo"
        self._halo_pending = []         # confirm="late": (word, collective, event, halo, margin, frame, lag) of the frames not yet judged
        self.fell_back = False          # the library's RCCL communicator was wanted and could not be made
        self.rccl_ranks = None          # what RCCL itself reports for the library's communicator (tr_comm_query)
        if (world > 1 or single_rank_comm) and renderer is not None and prefer_library and not self.host_staged:
            self._try_library_comm()
            if not self._comm.value:
                self.fell_back = True
                self.backend += " (FALLBACK: the library's RCCL communicator could not be created)"
        if self.host_staged:
            self.backend += " (device tensors staged through host memory)"

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
            n, r = C.c_uint32(), C.c_uint32()
            if lib.tr_comm_query(self._comm, C.byref(n), C.byref(r)) == 0:
                self.rccl_ranks = int(n.value)
                self.backend += f", {n.value} ranks by ncclCommCount"
        else:
            self.close()

    def close(self):
        if self._comm.value:
            self.renderer.lib.tr_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    # ---- the halo exchange of the sharded full pipeline
    halo_rows = 0          # level-0 rows exchanged on either side of a band (0: not chosen yet -> the first frame gathers)
    halo_fallbacks = 0     # frames that had to be redone with the full gather (confirm="late": that were found inexact a frame later)
    halo_margin = 1.25     # growth factor over the largest excursion seen
    halo_shrink_after = 32     # clean frames in a row before a frame PROBES a smaller halo (0: the halo never shrinks)
    halo_shrink_factor = 0.75  # the probe: taps checked against this fraction of the halo while the whole halo is still exchanged
    halo_clean_frames = 0
    halo_inexact_frames = 0    # confirm="late" only: composited frames whose taps turned out to have left the halo
    _halo_frame = 0            # confirm="late": frames submitted so far

    def halo_window_margin(self, halo: int) -> int:
        """The rows around its band a rank's taps are checked against this frame: the halo, or — once in a while, after
        halo_shrink_after clean frames — a smaller one, to find out whether the exchange can shrink again."""
        if self.halo_shrink_after and self.halo_clean_frames >= self.halo_shrink_after:
            probe = max(4, int(halo * self.halo_shrink_factor) // 4 * 4)
            if probe < halo:
                return probe
        return halo

    def halo_verdict(self, excess: int, halo: int, margin: int) -> bool:
        """Book-keeping behind a frame's (all-rank) excess word — 0, or 1 + the rows by which the worst tap missed the window
        of `margin` rows.  True: the frame's taps stayed inside the `halo` rows that were exchanged."""
        if excess > 0 and excess - 1 > halo - margin:            # a tap left what was exchanged
            self.halo_fallbacks += 1
            self.halo_clean_frames = 0
            self.halo_rows = int((margin + excess) * self.halo_margin) + 4
            return False
        if margin < halo:                                        # a probe frame
            self.halo_clean_frames = 0
            if excess == 0:
                self.halo_rows = margin                          # (every tap fitted the smaller window)
            return True
        self.halo_clean_frames += 1
        return True

    def confirm_halo(self, drain: bool = True) -> bool:
        """confirm="late": judges the frames whose excess word is due.  A word that was reduced on the host (gloo, the CPU
        tests) is due when the next frame starts.  A word on the device was copied, behind its all-reduce, into pinned host
        memory with an event behind the copy: it is due TWO frames later — by then the copy has landed, so reading it waits
        for nothing (round 5 read the device word one frame late with .item(): a host wait for the frame in flight, every
        frame) — and the same frame on every rank, which the halo's size must be.  drain=True (the default: after a loop's
        last frame) judges everything outstanding.  True: every frame judged by this call was exact (or there was none)."""
        ok_all = True
        while self._halo_pending and (drain or self._halo_pending[0][5] + self._halo_pending[0][6] <= self._halo_frame + 1):
            word, work, event, halo, margin, _, _ = self._halo_pending.pop(0)
            if work is not None:
                work.wait()
            if event is not None:
                event.synchronize()
            ok = self.halo_verdict(int(word.item()), halo, margin)
            if not ok:
                self.halo_inexact_frames += 1
            ok_all = ok_all and ok
        return ok_all

    def _halo_late_submit(self, word: torch.Tensor, halo: int, margin: int) -> None:
        """Enqueues the all-rank maximum of a frame's excess word; nothing waits (see confirm_halo)."""
        self._halo_frame += 1
        if self.host_staged:
            word = word.cpu()          # (the rehearsal's host-side backend reduces host memory: this read waits for the passes)
        work = dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group, async_op=True) if self.world > 1 else None
        if word.is_cuda:
            if work is not None:
                work.wait()            # (orders the current STREAM behind the collective; the host goes on)
            host = torch.empty(1, dtype=word.dtype, pin_memory=True)
            host.copy_(word, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self._halo_pending.append((host, None, event, halo, margin, self._halo_frame, 2))
        else:
            self._halo_pending.append((word, work, None, halo, margin, self._halo_frame, 1))

    def exchange_halo(self, level_rows: torch.Tensor, rows_per_rank: int, halo: int) -> None:
        """level_rows: (total_rows, W, C) contiguous, a whole pyramid level of which this rank has written its band; on
        return it also holds `halo` rows on either side, received from their owners."""
        if self.world == 1:
            return
        total = int(level_rows.shape[0])
        row_bytes = int(level_rows.shape[1]) * int(level_rows.shape[2]) * level_rows.element_size()
        if self._comm.value:
            st = self.renderer.lib.tr_exchange_halo(self.renderer._ctx, self._comm, level_rows.data_ptr(), row_bytes, total,
                                                    int(rows_per_rank), int(halo), torch.cuda.current_stream().cuda_stream)
            if st != 0:
                raise _lib.TrError(st, "tr_exchange_halo", self.renderer.lib.tr_comm_last_error(self._comm))
            return
        ops, keep, landed = [], [], []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            a, b = halo_rows_between(total, rows_per_rank, self.world, self.rank, peer, halo)     # mine, for the peer
            if b > a:
                keep.append(level_rows[a:b].cpu() if self.host_staged else level_rows[a:b].clone())
                ops.append(dist.P2POp(dist.isend, keep[-1], peer, group=self.group))
            a, b = halo_rows_between(total, rows_per_rank, self.world, peer, self.rank, halo)     # the peer's, for me
            if b > a:
                if self.host_staged:
                    landed.append((a, b, torch.empty(level_rows[a:b].shape, dtype=level_rows.dtype)))
                    ops.append(dist.P2POp(dist.irecv, landed[-1][2], peer, group=self.group))
                else:
                    ops.append(dist.P2POp(dist.irecv, level_rows[a:b], peer, group=self.group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for a, b, host in landed:
            level_rows[a:b].copy_(host)

    def allgather_bands(self, level_rows: torch.Tensor, rows_per_rank: int) -> None:
        """In-place all-gather of a level whose bands are clipped to it (no padding rows behind it)."""
        self.exchange_halo(level_rows, rows_per_rank, int(level_rows.shape[0]))

    def max_over_ranks(self, value: int) -> int:
        if self.world == 1:
            return int(value)
        on_host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([int(value)], dtype=torch.int64, device="cpu" if on_host else self.renderer.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def allgather_rows(self, frame: torch.Tensor) -> None:
        """frame: (rows_per_rank * world, W, C) contiguous; every rank has written its own band."""
        if self.world == 1 and not self._comm.value:
            return
        assert frame.is_contiguous() and frame.shape[0] % self.world == 0, "gathered buffers hold world * rows_per_rank rows"
        rows = frame.shape[0] // self.world
        if self._comm.value:
            fmt = {torch.float16: wire.FORMAT_RGBA16F, torch.float32: wire.FORMAT_RGBA32F, torch.uint8: wire.FORMAT_RGBA8}[frame.dtype]
            if frame.dtype == torch.uint8 and frame.shape[2] == 3:
                fmt = wire.FORMAT_RGB8           # (the presented frame without its constant alpha: tonemap_rgb8)
            else:
                assert frame.shape[2] == 4
            st = self.renderer.lib.tr_allgather_frame(self.renderer._ctx, self._comm, frame.data_ptr(), int(frame.shape[1]),
                                                      rows, fmt, torch.cuda.current_stream().cuda_stream)
            if st != 0:
                raise _lib.TrError(st, "tr_allgather_frame", self.renderer.lib.tr_comm_last_error(self._comm))
            return
        flat = frame.view(-1)
        n = flat.numel() // self.world
        mine = flat[self.rank * n:(self.rank + 1) * n]
        if self.host_staged:
            outs = [torch.empty(n, dtype=flat.dtype) for _ in range(self.world)]
            dist.all_gather(outs, mine.cpu(), group=self.group)
            for i, o in enumerate(outs):
                if i != self.rank:
                    flat[i * n:(i + 1) * n].copy_(o)
        elif dist.get_backend(self.group) == "gloo":   # gloo has no in-place variant on views of the output
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
        channels = int(frame.shape[2])
        if frame.dtype == torch.uint8 and channels == 3:
            fmt = wire.FORMAT_RGB8          # (tonemap_rgb8's frames: three bytes per pixel, as allgather_rows)
        else:
            assert channels == 4, "a strip composite takes (H, W, 4) frames, or (H, W, 3) uint8 ones"
            fmt = {torch.float16: wire.FORMAT_RGBA16F, torch.float32: wire.FORMAT_RGBA32F, torch.uint8: wire.FORMAT_RGBA8}[frame.dtype]
        st = comp.renderer.lib.tr_allgather_strips(comp.renderer._ctx, comp._comm, frame.data_ptr(), w, h, int(strip_rows), fmt,
                                                   torch.cuda.current_stream().cuda_stream)
        if st != 0:
            raise _lib.TrError(st, "tr_allgather_strips", comp.renderer.lib.tr_comm_last_error(comp._comm))
        return
    works, landed = [], []
    for s, y in enumerate(range(0, h, strip_rows)):
        rows = frame[y:min(y + strip_rows, h)]
        buf = rows
        if comp.host_staged:                      # (see Compositor: a host-side backend under device tensors)
            buf = rows.cpu()
            if s % comp.world != comp.rank:
                landed.append((rows, buf))
        works.append(dist.broadcast(buf, src=dist.get_global_rank(comp.group, s % comp.world)
                                    if comp.group is not None else s % comp.world, group=comp.group, async_op=True))
    for wk in works:
        wk.wait()
    for rows, buf in landed:
        rows.copy_(buf)


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
                   composite: bool = True, exchange: str = "allgather", confirm: str = "now") -> None:
    """The hot-path slice of `record()` (src/main.rs:1969-2124) for one rank of a row-band sharded frame.

    `opaque` / `transmissive` are this rank's G-buffer tiles (origin_y = its first row); `hdr` holds
    padded_rows(H, world) rows and `pyramid` was made with level0_rows = the same.  Order:
    opaque band -> exchange of the opaque colour -> mip chain -> transmissive band -> composite.
    exchange = "allgather": all of level 0 to every rank, the chain replicated (round 3);
    exchange = "halo": border rows of levels 0 / 1 + all of level 2 (see the module docstring); needs frame sizes that are
    multiples of 4 (else it gathers) and falls back to the gather for a frame whose taps left the halo.
    confirm = "now": the frame's excess word is read (a device drain and a small all-reduce) before the composite, and a
    frame whose taps left the halo is redone — every frame returned is exact.  confirm = "late" (a renderer that hands out
    the word on the device: tap_window_excess_word): the word's all-reduce is enqueued, the composite follows at once, and
    the verdict is read later — a word reduced on the device two frames later, from pinned host memory behind an event, so
    that no host call of the frame loop waits for the GPU; a word reduced on the host when the next frame starts
    (compositor.confirm_halo) —: a frame whose taps left the halo has then been composited as it was —
    compositor.halo_inexact_frames counts them — and the following frames exchange more rows.  Either way a halo that has been wide enough for compositor.halo_shrink_after frames is probed:
    one frame checks its taps against three quarters of it, and the exchange shrinks if they fit.
    """
    world, rank = compositor.world, compositor.rank
    fw, fh = int(push.framebuffer_size[0]), int(push.framebuffer_size[1])
    rect = band_rect(fw, fh, world, rank)
    rows = band_rows(fh, world, rank)[0]
    mine = rect[3] > rect[1]     # (a band can be empty when the height is far from a multiple of world * 4)
    if mine:
        renderer.shade_opaque(opaque, uniforms, push, hdr, pyramid, rect)
    compositor.confirm_halo(drain=False)   # (confirm="late": the verdicts that are due; they may widen the halo used below)
    halo = compositor.halo_rows
    use_halo = (world > 1 and exchange == "halo" and fw % 4 == 0 and fh % 4 == 0 and pyramid.levels >= 4
                and 0 < halo and halo + rows < fh)     # (a halo that reaches every row is the gather)
    if use_halo:
        halo = min(-(-halo // 4) * 4, fh)
        margin = compositor.halo_window_margin(halo)   # (the same on every rank: the counters move with all-rank verdicts)
        # (late: the word stays on the device — a host-side backend could only reduce it after reading it)
        late = (confirm == "late" and hasattr(renderer, "tap_window_excess_word")
                and (world == 1 or str(renderer.device) == "cpu" or dist.get_backend(compositor.group) != "gloo" or compositor.host_staged))
        renderer.generate_mips_band(pyramid, rect[1], rect[3])                       # levels 1, 2 of the band
        compositor.exchange_halo(pyramid.level(0), rows, halo)
        compositor.exchange_halo(pyramid.level(1), rows // 2, halo // 2)
        compositor.allgather_bands(pyramid.level(2), rows // 4)
        renderer.generate_mips_from(pyramid, 3)
        excess, word = 0, None
        if mine:
            renderer.set_tap_window(max(rect[1] - margin, 0), min(rect[3] + margin, fh))
            try:
                renderer.shade_transmission(transmissive, uniforms, push, pyramid, hdr, rect)
            finally:
                if late:
                    word = renderer.tap_window_excess_word()     # (a device tensor: nothing is read here)
                else:
                    excess = renderer.tap_window_excess()
                renderer.set_tap_window(0, 0)
        if late:
            if word is None:
                word = torch.zeros(1, dtype=torch.int64, device=renderer.device)
            compositor._halo_late_submit(word, halo, margin)
            if composite:
                compositor.allgather_rows(hdr)
            return
        excess = compositor.max_over_ranks(excess)
        if compositor.halo_verdict(excess, halo, margin):
            if composite:
                compositor.allgather_rows(hdr)
            return
        # a tap left the halo: this frame is redone with the full gather, the next ones exchange what was needed
    elif world > 1 and exchange == "halo" and halo == 0:
        compositor.halo_rows = max(rows // 4, 4)         # (first frame: gather, then start from a quarter band)
    if world > 1:
        compositor.allgather_rows(pyramid.level0_padded())
    renderer.generate_mips(pyramid)
    if mine:
        renderer.shade_transmission(transmissive, uniforms, push, pyramid, hdr, rect)
    if world > 1 and composite:
        compositor.allgather_rows(hdr)
