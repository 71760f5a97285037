// tr_lc_kernels.h — the untextured plane pass as LOADER and CONSUMER waves (gfx950, wave64).
//
// shade_kernel (tr_kernels.h) gives every wave both roles: it requests its tile's planes, waits, shades, requests the taps,
// waits, stores.  Eight such in-order waves per SIMD alternate between the memory pipeline and the vector unit; by the
// counters of round 5 each CU's L1 holds as many misses as it can for half of a launch and is idle for the other half while
// the vector issue port is 0.82 busy.  Here the two roles are separate waves of one persistent workgroup:
//
//   * wave 0 of a workgroup — the LOADER — does nothing but send the next tiles' rows of `pos_depth`, `nrm_scale`,
//     `material_id` and the tile's cluster-table entries from memory into a ring of slots in the workgroup's LDS
//     (`global_load_lds`, non-temporal: no register is involved, the data never passes the vector unit), up to kAhead tiles in
//     flight behind its own counted `s_waitcnt vmcnt`, and publishes a tile by advancing a counter in LDS once the tile's
//     four transfers have landed;
//   * waves 1..15 — the CONSUMERS — draw a ticket from a counter in LDS, wait until that tile is published, read their
//     pixel's inputs from the slot (two ds_read_b128, three ds_read_b32), hand the slot back, and run the SAME shade_pixel as
//     shade_kernel: cluster list through the scalar unit, lights, taps, store.  A consumer never issues a plane load; what it
//     waits for in memory is its taps and the store's place in the queue.
//
// The ring: kSlots slots of 2560 B (1024 position + 1024 normal + 256 ids + 256 cluster entries); a slot carries the
// number of the tile it may next be filled with (`slot_turn`), which the consumer that has read tile t advances from t to
// t + kSlots — consumers finish out of order, a single count of consumed tiles would not say WHICH slots are free.
// LDS operations of one wave execute in order, so the hand-back store is behind the reads it follows without a wait.
// Two workgroups of sixteen waves per CU (64 registers, 77 KB of LDS each).
//
// Which tiles: the XCD's contiguous band of block tiles as in shade_kernel; workgroup g of the XCD's G takes block tiles g,
// g + G, g + 2G ... of the band (the XCD still sweeps its band front to back), four wave tiles each, in that order.
//
// Reference semantics: exactly shade_kernel<true, OutT, kTexNone, false>'s — fragment_transmission (shader/src/lib.rs:37-162),
// evaluate_lights_transmission (shader/src/lighting.rs:13-95), glam-pbr/src/lib.rs:200-354 — the pixel code is shared.
#pragma once

#include "tr_kernels.h"

namespace tr {

#ifndef TR_LC_WAVES
#define TR_LC_WAVES 16
#endif
#ifndef TR_LC_SLOTS
#define TR_LC_SLOTS 25
#endif
constexpr uint32_t kLcWaves = TR_LC_WAVES;         // waves of a workgroup: one loader, fifteen consumers
constexpr uint32_t kLcSlots = TR_LC_SLOTS;                 // ring slots: 64 000 B — the ring stays below 64 KB of LDS (what M0 of a transfer can address is not in the guides)
constexpr uint32_t kLcSlotBytes = 2560u;           // 1024 pos_depth + 1024 nrm_scale + 256 material_id + 256 cluster entries
constexpr uint32_t kLcOffNormal = 1024u, kLcOffIds = 2048u, kLcOffCluster = 2304u;
#ifndef TR_LC_AHEAD
#define TR_LC_AHEAD 8
#endif
constexpr uint32_t kLcAhead = TR_LC_AHEAD;         // tiles whose transfers the loader keeps in flight (4 transfers each: vmcnt <= 63)
constexpr uint32_t kLcSpinLimit = 1u << 21;       // polls of ~100 cycles: about a tenth of a second, five orders above a tile's time
static_assert(kLcAhead >= 1u && kLcAhead * 4u <= 60u && kLcAhead < kLcSlots, "loader run-ahead");

struct lc_control {
    uint32_t ticket;                // next tile of the workgroup's sequence to hand to a consumer
    uint32_t published;             // tiles [0, published) have landed in their slots
    uint32_t _pad[2];
    uint32_t slot_turn[kLcSlots + 2u];   // slot s may be filled with tile slot_turn[s] (s, s + kLcSlots, ...)
};

// One LDS-DMA transfer: every lane's `bytes` from (scalar base + the lane's 32-bit offset) to LDS at lds_base + lane * bytes.
// Inline assembly, so that the compiler's wait bookkeeping does not see it (it would drain every transfer in front of the
// next LDS access of the wave); M0 — the transfer's LDS base — is compiler-reserved and saved around the statement.
template <int BYTES>
__device__ __forceinline__ void lds_dma_nt(const void* base, uint32_t byte_offset, uint32_t lds_base) {
    uint32_t keep;
    if constexpr (BYTES == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(byte_offset), "s"(base), "s"(lds_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(byte_offset), "s"(base), "s"(lds_base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
}

template <typename OutT /* uint2 = RGBA16F, float4 = RGBA32F */>
__global__ __launch_bounds__(kLcWaves * 64u) __attribute__((amdgpu_waves_per_eu(8)))
void shade_lc_kernel(const tr_launch launch_by_value) {
    (void)launch_by_value;
    claunch* L = launder((claunch*)__builtin_amdgcn_kernarg_segment_ptr());
    __shared__ __attribute__((aligned(16))) unsigned char ring[kLcSlots * kLcSlotBytes];
    __shared__ lc_control ctl;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lx = lane & (kWaveTileW - 1u), ly = lane / kWaveTileW;

    if (threadIdx.x < kLcSlots) ctl.slot_turn[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0u) {
        ctl.ticket = 0u;
        ctl.published = 0u;
    }
    __syncthreads();

    // The XCD's band and this workgroup's share of it (see the file comment)
    const uint32_t ntiles = L->fp.tiles_x * L->fp.tiles_y;
    const uint32_t xcd = blockIdx.x & 7u, g = blockIdx.x >> 3, G = L->fp.j_step;   // (j_step: workgroups of the grid per XCD)
    const uint32_t per = ntiles >> 3, rem = ntiles & 7u;
    const uint32_t band_start = xcd * per + min(xcd, rem);
    const uint32_t band_len = per + (xcd < rem ? 1u : 0u);
    const uint32_t own_blocks = band_len > g ? (band_len - 1u - g) / G + 1u : 0u;
    const uint32_t own_tiles = own_blocks * 4u;
    // (scalar) tile t of the workgroup's sequence -> the wave tile's column / row in the rect
    struct tile_map {
        uint32_t tiles_x, tiles_x_magic, strip_tile_rows, strip_magic, strip_world, strip_rank;
    };
    auto tile_map_of = [](claunch* F) {
        return tile_map{F->fp.tiles_x, F->fp.tiles_x_magic, F->fp.strip_tile_rows, F->fp.strip_magic, F->fp.strip_world, F->fp.strip_rank};
    };
    auto tile_of = [&](const tile_map& M, uint32_t t, uint32_t& txi, uint32_t& tyi) {
        const uint32_t tile = band_start + (t >> 2) * G + g;
        tyi = __umulhi(tile, M.tiles_x_magic);
        txi = tile - tyi * M.tiles_x;
        if (txi >= M.tiles_x) {
            txi -= M.tiles_x;
            ++tyi;
        }
        if (M.strip_tile_rows != 0u) {   // (scalar) this rank's strips of a frame shared with other ranks, as in shade_kernel
            const uint32_t T = M.strip_tile_rows;
            uint32_t k = __umulhi(tyi, M.strip_magic), r = tyi - k * T;
            if (r >= T) {
                r -= T;
                ++k;
            }
            tyi = (k * M.strip_world + M.strip_rank) * T + r;
        }
        txi = txi * 4u + (t & 3u);
    };
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
    // (the ring's words as relaxed workgroup-scope atomics on the __shared__ objects themselves: a `volatile` pointer loses
    //  the LDS address space — flat loads with a vmcnt(0) behind each — and an empty asm with a memory clobber pins the order
    //  of the plain slot reads between them)
    auto lds_get = [](uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_put = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    if (wave == 0u) {
        // ------------------------------------------------------------------------------------------------ the loader
        // A tile must leave this loop every ~600 cycles (fifteen consumers at ~9 400 cycles a tile), so nothing in it waits
        // for memory: the launch constants are read once (the loader has the scalar registers: it never shades), the scalar
        // tile arithmetic runs once per block tile, and the slots' turn words are cached one per lane and read again — all
        // of them in one LDS access — only when the cached word says "not yet".  (First version: the kernarg segment and the
        // turn word read per tile, ~800 cycles of dependent round trips: the consumers starved, 97 us against 89.)
        __builtin_amdgcn_s_setprio(3);
        claunch* F = launder(L);
        const uint32_t rect_x0 = F->fp.rect_x0, rect_y0 = F->fp.rect_y0, x_last = F->fp.rect_x1 - 1u, y_last = F->fp.rect_y1 - 1u;
        const uint32_t g_width = F->fp.g_width, g_origin_x = F->fp.g_origin_x, g_origin_y = F->fp.g_origin_y;
        const void* const pos_depth = F->pos_depth;
        const void* const nrm_scale = F->nrm_scale;
        const void* const material_id = F->material_id;
        // cluster entries: lanes 0-15 the x table's of the tile's columns, lanes 16-19 the y table's of its rows (the other
        // lanes repeat them: every lane of a transfer delivers)
        // (both bases made opaque first: a select of two kernarg loads is otherwise turned into a per-lane load of the
        //  selected kernarg slot, and its vmcnt(0) drains every transfer in flight)
        const bool row_lane = (lane & 48u) == 16u;
        const char* const table = row_lane ? (const char*)launder(F->cluster_y_term) : (const char*)launder(F->cluster_x);
        const tile_map M = tile_map_of(F);
        uint32_t turns = lane;          // lane s: slot s's turn word as last read (slot_turn[s] starts as s)
        uint32_t slot = 0u, x0 = 0u, y0 = 0u;
        for (uint32_t t = 0u; t < own_tiles; ++t) {
            if ((t & 3u) == 0u) {       // (scalar) a new block tile: its four wave tiles follow side by side
                uint32_t txi, tyi;
                tile_of(M, t, txi, tyi);
                x0 = rect_x0 + txi * kWaveTileW;
                y0 = rect_y0 + tyi * kWaveTileH;
            } else {
                x0 += kWaveTileW;
            }
            // the pixels shade_kernel's load_inputs reads: out-of-rect lanes a clamped (valid) one
            const uint32_t cx = min(x0 + lx, x_last), cy = min(y0 + ly, y_last);
            const uint32_t gpix = mad24(cy - g_origin_y, g_width, cx - g_origin_x);
            const uint32_t table_at = row_lane ? min(y0 + (lane & 3u), y_last) : cx;
            // the slot is free once the consumer of tile t - kLcSlots has read it
            // (every spin is bounded: a protocol error must end as a wrong frame the parity tests catch, never as a hung GPU)
            uint32_t spins = 0u;
            while ((uint32_t)__builtin_amdgcn_readlane((int)turns, (int)slot) != t && ++spins < kLcSpinLimit) {
                if (spins > 1u) __builtin_amdgcn_s_sleep(1);
                turns = lds_get(&ctl.slot_turn[min(lane, kLcSlots - 1u)]);
            }
            if (spins >= kLcSpinLimit) break;   // (gives up: everything is "published", the consumers run out)
            const uint32_t at = ring_base + slot * kLcSlotBytes;
            lds_dma_nt<16>(pos_depth, gpix * 16u, at);
            lds_dma_nt<16>(nrm_scale, gpix * 16u, at + kLcOffNormal);
            lds_dma_nt<4>(material_id, gpix * 4u, at + kLcOffIds);
            {   // (the table differs by lane group: a per-lane 64-bit address, no scalar base)
                uint32_t keep;
                const char* p = table + (size_t)table_at * 4u;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p), "s"(at + kLcOffCluster) : "memory");
            }
            slot = slot + 1u == kLcSlots ? 0u : slot + 1u;
            if (t >= kLcAhead) {   // tile t - kLcAhead has landed when at most 4 kLcAhead transfers are still on their way
                wait_vmcnt<4 * kLcAhead>();
                lds_put(&ctl.published, t - kLcAhead + 1u);
            }
        }
        wait_vmcnt<0>();
        lds_put(&ctl.published, own_tiles);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- the consumers
#if TR_TIMING
    tr_timer timer = {{0ull, 0ull, 0ull, 0ull, 0ull}};
#endif
    while (true) {
        tile_phase<0>();
        uint32_t t = 0u;
        if (lane == 0u) t = __hip_atomic_fetch_add(&ctl.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= own_tiles) break;
        claunch* F = launder(L);
        uint32_t txi, tyi;
        tile_of(tile_map_of(F), t, txi, tyi);
        tile_regs cur;
        cur.px = F->fp.rect_x0 + txi * kWaveTileW + lx;
        cur.py = F->fp.rect_y0 + tyi * kWaveTileH + ly;
        const uint32_t slot = t % kLcSlots;
        uint32_t spins = 0u;
        while ((int32_t)(lds_get(&ctl.published) - t) <= 0 && ++spins < kLcSpinLimit) __builtin_amdgcn_s_sleep(1);
        if (spins >= kLcSpinLimit) break;
        {
            typedef float f4v __attribute__((ext_vector_type(4)));
            asm volatile("" ::: "memory");
            const unsigned char* const s = ring + slot * kLcSlotBytes;
            const f4v a = *reinterpret_cast<const f4v*>(s + lane * 16u);
            const f4v b = *reinterpret_cast<const f4v*>(s + kLcOffNormal + lane * 16u);
            cur.mat = *reinterpret_cast<const uint32_t*>(s + kLcOffIds + lane * 4u);
            cur.cluster_x = *reinterpret_cast<const uint32_t*>(s + kLcOffCluster + lx * 4u);
            cur.cluster_y_term = *reinterpret_cast<const uint32_t*>(s + kLcOffCluster + 64u + ly * 4u);
            cur.pd = float4{a.x, a.y, a.z, a.w};
            cur.ns = float4{b.x, b.y, b.z, b.w};
            asm volatile("" ::: "memory");
            // hand the slot back (behind the reads: a wave's LDS operations execute in order)
            if (lane == 0u) lds_put(&ctl.slot_turn[slot], t + kLcSlots);
        }
        claunch* S = launder(L);
        const bool inside = cur.px < S->fp.rect_x1 && cur.py < S->fp.rect_y1;
        const bool active = inside && cur.mat != TR_NOT_COVERED;
        const uint32_t key = inside ? cur.mat : TR_NOT_COVERED;
        f3 out = {0.f, 0.f, 0.f};
        uint64_t todo = ballot(key != TR_NOT_COVERED);
        cdmat* dmats = as_constant(S->dmats);
        if (todo) {
            const cluster_list cl = cluster_lookup(S, cur.pd.w, cur.cluster_x + cur.cluster_y_term, key != TR_NOT_COVERED);
            while (todo) {
                const int l0 = __ffsll((unsigned long long)todo) - 1;
                const uint32_t mk = (uint32_t)__builtin_amdgcn_readlane((int)key, l0);
                const uint32_t m0 = opaque(mk);
                const uint64_t group = ballot(key == mk);
                todo &= ~group;
                if (key == mk) out = shade_pixel<true>(L, dmats + m0, m0, cur.pd, cur.ns, lane, cl TR_PROBE_ARGS);
            }
        }
        // uncovered pixels keep the attachment (LOAD)
        if (active) {
            claunch* W = launder(L);
            const uint32_t pix = mad24(cur.py, W->fp.width, cur.px);
            if constexpr (sizeof(OutT) == 8) {
                const uint2 o = pack_rgba16f(out.x, out.y, out.z, 1.0f);
                typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(u2v{o.x, o.y}, reinterpret_cast<u2v*>(static_cast<char*>(W->hdr) + pix * 8u));
            } else {
                st<OutT>(W->hdr, pix * 16u, OutT{out.x, out.y, out.z, 1.0f});
            }
        }
    }
}

}  // namespace tr
