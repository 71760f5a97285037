// pattern_ceiling.hip — what MI355X can move in the ACCESS PATTERN of the 4K transmissive plane pass when nothing else competes:
// per 16x4 wave tile two non-temporal dwordx4 plane rows and a dword id row (36 B/px streamed once), T scattered 16-byte "taps"
// per pixel into an 88 MB RGBA16F pyramid at a bounded displacement from the pixel (the refraction scatter: level 0 / 1 texel
// pairs, two rows each), one 16-byte LUT-line load from a 4 MB table, an 8-byte non-temporal store per pixel; one-wave
// workgroups on the pass's own tile numbering (XCD bands, wave slot w takes tiles w, w + W, ...), W waves per SIMD chosen
// by an LDS allocation.  No shading arithmetic: a tile's loads are issued together, waited for once, folded with a few adds
// into the stored value (so that nothing is eliminated) — the least instruction stream that makes the requests.
//
// Reported per configuration: us per frame, TB/s on the 52 B/px the pass is priced on, the mean time a tile's loads are
// outstanding (s_memtime around the wait, converted with the measured tick rate), and Little's law read the other way:
//   bytes in flight per CU = (read bytes per frame / 256 CUs) / (frame time) x latency
// — the number the memory side sustains per CU for THIS pattern.  The last rows repeat the 8-wave measurement with a block
// of N dependent fma per pixel between the wait and the store (the arithmetic a real pass must overlap with the stream).
//
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/ubench/pattern_ceiling.hip -o /tmp/pattern_ceiling && /tmp/pattern_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

struct params {
    const f4v* pos; const f4v* nrm; const uint32_t* ids; const u4v* pyr; const f4v* lut; u2v* out;
    uint32_t width, height, tiles_x, ntiles, j_step, taps, fma, scatter;   // scatter: largest tap displacement in pixels
    // TAP_MODE (template parameter of pattern_kernel)
    //                        0: texel-aligned 16-byte pairs (the pass); 1: the same pairs forced to 16-byte alignment; 2: 8-byte loads, one texel each;
                           // 3: a level's taps of the whole tile as ONE coalesced load of the tile's texel window into LDS (the window from
                           //    the corner lanes' taps, every lane checked against it; a tile that does not fit gathers as in mode 0);
                           // 4: mode 0's requests folded onto 64 x 64 texels of each level (every tap a cache hit)
    // AHEAD (template parameter) 1: the NEXT tile's plane rows and ids are requested before this tile's loads are waited for (two tiles of a
                           // wave in flight: what more bytes in flight per CU are worth to the pattern itself)
    uint32_t tile_begin;   // first block tile of this launch's row band (two bands on two streams: bench.py's step)
    uint32_t order;        // how an XCD walks its band of block tiles: 0 row-major (the pass); k > 0: in super-columns k block tiles wide
                           // (row-major inside a super-column, the super-columns left to right)
    uint32_t field;        // the displacement field's direction: 0 both, 1 x only, 2 y only
    unsigned long long* wait_ticks; unsigned long long* waits;
};

// TAP_MODE / AHEAD are compile-time: each variant is its own lean kernel (a first version switched on them at run time and the
// extra live values of the window mode cost EVERY variant a wave per SIMD: 66 registers)
template <int LDS_BYTES, int TAP_MODE = 0, int AHEAD = 0>
__global__ __launch_bounds__(64) void pattern_kernel(const params p) {
    __shared__ unsigned char occupy[LDS_BYTES];      // limits the waves a CU holds
    __shared__ u4v win[TAP_MODE == 3 ? 128 : 1];      // TAP_MODE 3: the tile's texel windows of the two levels
    if (p.width == 0xFFFFFFFFu) occupy[threadIdx.x] = 1;
    const uint32_t lane = threadIdx.x, lx = lane & 15u, ly = lane >> 4;
    const uint32_t xcd = blockIdx.x & 7u, per = p.ntiles >> 3, rem = p.ntiles & 7u;
    const uint32_t band_start = xcd * per + (xcd < rem ? xcd : rem), band_len = per + (xcd < rem ? 1u : 0u);
    unsigned long long waited = 0, n = 0;
    auto pixel_of = [&](uint32_t j, uint32_t& px, uint32_t& py, uint32_t& txi, uint32_t& tyi) {
        uint32_t t = j >> 2;
        if (p.order) {   // (scalar) super-columns of `order` block tiles over the band's whole tile rows; what is left of the band row-major
            const uint32_t rows = band_len / p.tiles_x, cols = p.tiles_x / p.order, body = rows * cols * p.order;
            if (t < body) {
                const uint32_t per_col = rows * p.order, c = t / per_col, r = (t - c * per_col) / p.order, x = t - c * per_col - r * p.order;
                t = r * p.tiles_x + c * p.order + x;      // (order divides tiles_x)
            }
        }
        const uint32_t tile = p.tile_begin + band_start + t;
        tyi = tile / p.tiles_x;
        txi = (tile - tyi * p.tiles_x) * 4u + (j & 3u);
        px = min(txi * 16u + lx, p.width - 1u);
        py = min(tyi * 4u + ly, p.height - 1u);
    };
    f4v a_next = {0.f, 0.f, 0.f, 0.f}, b_next = {0.f, 0.f, 0.f, 0.f};
    uint32_t id_next = 0u;
    const uint32_t j0 = blockIdx.x >> 3;
    if (AHEAD && j0 < band_len * 4u) {
        uint32_t px, py, txi, tyi;
        pixel_of(j0, px, py, txi, tyi);
        const uint32_t pix = py * p.width + px;
        a_next = __builtin_nontemporal_load(p.pos + pix);
        b_next = __builtin_nontemporal_load(p.nrm + pix);
        id_next = p.ids[pix];
    }
    for (uint32_t j = j0; j < band_len * 4u; j += p.j_step) {
        uint32_t px, py, txi, tyi;
        pixel_of(j, px, py, txi, tyi);
        const uint32_t pix = py * p.width + px;
        f4v a, b;
        uint32_t id;
        if (AHEAD) {
            a = a_next; b = b_next; id = id_next;           // (requested a tile ago)
            if (j + p.j_step < band_len * 4u) {
                uint32_t qx, qy, tx2, ty2;
                pixel_of(j + p.j_step, qx, qy, tx2, ty2);
                const uint32_t q = qy * p.width + qx;
                a_next = __builtin_nontemporal_load(p.pos + q);
                b_next = __builtin_nontemporal_load(p.nrm + q);
                id_next = p.ids[q];
            }
        } else {
            a = __builtin_nontemporal_load(p.pos + pix);
            b = __builtin_nontemporal_load(p.nrm + pix);
            id = p.ids[pix];
        }
        // the taps: displaced from the pixel by a field that varies smoothly over the screen (neighbouring pixels refract
        // alike: adjacent lanes fetch adjacent texels, as in the pass; `scatter` is the field's amplitude in pixels) and jumps
        // at "material" borders every 96 pixels; pairs of rows like the sampler's (row, row + 1) of two levels
        // (integer arithmetic only, a handful of instructions: the addresses must not be what the kernel spends its time on — a
        //  first version formed them with sin / cos, divisions by 96 and 64-bit products and measured ITS OWN vector work)
        const int wob_x = (int)(((px >> 3) + (py >> 2)) & 63u) - 32, wob_y = (int)(((px >> 4) - (py >> 3)) & 63u) - 32;   // -32 .. 31, constant over 8 x 4 pixels
        const int sdx = p.field == 2u ? 0 : (wob_x * (int)p.scatter) >> 5, sdy = p.field == 1u ? 0 : (wob_y * (int)p.scatter) >> 5;
        u4v t[8];
        const uint32_t taps = p.taps;
        bool staged[2] = {false, false};
        u4v wreg[2] = {u4v{0, 0, 0, 0}, u4v{0, 0, 0, 0}};
        uint32_t w_at[2] = {0u, 0u}, w_pitch[2] = {0u, 0u}, w_lanes[2] = {0u, 0u};
        if (TAP_MODE == 3 && taps == 4u) {
            // per level: the window of texel pairs (16-byte chunks) that holds every lane's two rows of two texels, requested as
            // ONE coalesced load (a lane per chunk) beside the plane loads; spread to the lanes through LDS behind the common wait
#pragma unroll
            for (uint32_t level = 0; level < 2u; ++level) {
                const uint32_t lw = p.width >> level, lh = p.height >> level, base = level ? p.width * p.height : 0u;
                const uint32_t tx = (uint32_t)min(max((int)(px >> level) + (sdx >> level), 0), (int)lw - 2);
                const uint32_t ty = (uint32_t)min(max((int)(py >> level) + (sdy >> level), 0), (int)lh - 2);
                // the corner lanes' taps bound the window when the displacement is monotone over the tile; every lane is checked
                uint32_t cx[4], cy[4];
                const int corner[4] = {0, 15, 48, 63};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cx[c] = (uint32_t)__builtin_amdgcn_readlane((int)tx, corner[c]);
                    cy[c] = (uint32_t)__builtin_amdgcn_readlane((int)ty, corner[c]);
                }
                const uint32_t x_lo = min(min(cx[0], cx[1]), min(cx[2], cx[3])) & ~1u, x_hi = max(max(cx[0], cx[1]), max(cx[2], cx[3])) + 1u;
                const uint32_t y_lo = min(min(cy[0], cy[1]), min(cy[2], cy[3])), y_hi = max(max(cy[0], cy[1]), max(cy[2], cy[3])) + 1u;
                const uint32_t chunks = ((x_hi - x_lo) >> 1) + 1u, rows = y_hi - y_lo + 1u;
                const bool inside = tx >= x_lo && tx + 1u <= x_hi && ty >= y_lo && ty + 1u <= y_hi;
                const bool fits = chunks * rows <= 64u && __builtin_amdgcn_ballot_w64(!inside) == 0ull;   // (wave-uniform)
                if (fits) {
                    if (lane < chunks * rows) {
                        const uint32_t r = lane / chunks, c = lane - r * chunks;
                        const uint32_t gx = min(x_lo + 2u * c, lw - 2u), gy = min(y_lo + r, lh - 1u);
                        wreg[level] = *reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(p.pyr) + (base + gy * lw + gx) * 8u);
                    }
                    w_at[level] = (ty - y_lo) * chunks * 2u + (tx - x_lo);
                    w_pitch[level] = chunks * 2u;
                    w_lanes[level] = chunks * rows;
                    staged[level] = true;
                }
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (k < taps && !staged[k >> 1]) {
                const int dx = sdx, dy = sdy;
                const uint32_t level = k >> 1;    // taps 0,1: level 0 rows y, y + 1; taps 2,3: level 1
                const uint32_t lw = p.width >> level, lh = p.height >> level, base = level ? p.width * p.height : 0u;
                const uint32_t tx = (uint32_t)min(max((int)(px >> level) + (dx >> level), 0), (int)lw - 2);
                const uint32_t tyy = (uint32_t)min(max((int)(py >> level) + (dy >> level) + (int)(k & 1u), 0), (int)lh - 1);
                uint32_t texel = base + tyy * lw + (TAP_MODE == 1 ? (tx & ~1u) : tx);     // (the pyramid is < 4 GB: 32-bit byte offsets)
                if (TAP_MODE == 4)   // the same requests folded onto 64 x 64 texels of each level: every tap an L2 hit, no tap traffic from memory
                    texel = base + (tyy & 63u) * lw + (tx & 63u);
                const char* at = reinterpret_cast<const char*>(p.pyr) + texel * 8u;
                if (TAP_MODE == 5) {          // ONE texel (8 bytes) per lane and row: half the bytes through the L1, the same instructions and lines
                    const u2v lo = *reinterpret_cast<const u2v*>(at);
                    t[k] = u4v{lo.x, lo.y, 0u, 0u};
                } else if (TAP_MODE == 6) {   // 4 bytes per lane and row
                    t[k] = u4v{*reinterpret_cast<const uint32_t*>(at), 0u, 0u, 0u};
                } else if (TAP_MODE == 2) {
                    const u2v lo = *reinterpret_cast<const u2v*>(at), hi = *reinterpret_cast<const u2v*>(at + 8);
                    t[k] = u4v{lo.x, lo.y, hi.x, hi.y};
                } else {
                    typedef u4v u4v_a8 __attribute__((aligned(8)));
                    t[k] = *reinterpret_cast<const u4v_a8*>(at);
                }
            } else if (k >= taps) {
                t[k] = u4v{0, 0, 0, 0};
            }
        }
        // the LUT line: 258 entries of 16 bytes per material, indexed by n.v — smooth over the screen like the taps
        const f4v line = taps ? p.lut[(id & 15u) * 260u + (((px + py) >> 1) & 255u)] : f4v{0.f, 0.f, 0.f, 0.f};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (AHEAD) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // (everything but the next tile's three plane loads: they are the youngest)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        waited += __builtin_amdgcn_s_memtime() - t0;
        ++n;
        if (TAP_MODE == 3 && (staged[0] || staged[1])) {   // (wave-uniform) windows -> LDS -> every lane's two rows of two texels
#pragma unroll
            for (uint32_t level = 0; level < 2u; ++level)
                if (staged[level] && lane < w_lanes[level]) win[level * 64u + lane] = wreg[level];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave per workgroup: no barrier)
#pragma unroll
            for (uint32_t level = 0; level < 2u; ++level)
                if (staged[level]) {
                    const u2v* texels = reinterpret_cast<const u2v*>(win + level * 64u);
                    const uint32_t at = w_at[level], pitch = w_pitch[level];
                    const u2v q00 = texels[at], q10 = texels[at + 1u], q01 = texels[at + pitch], q11 = texels[at + pitch + 1u];
                    t[2u * level] = u4v{q00.x, q00.y, q10.x, q10.y};
                    t[2u * level + 1u] = u4v{q01.x, q01.y, q11.x, q11.y};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float x = a.x + b.x + line.x, y = a.y + b.y + line.y, z = a.z + b.z + a.w + b.w + line.z + line.w;
        uint32_t fold = id;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) fold += t[k].x ^ t[k].y ^ t[k].z ^ t[k].w;
        for (uint32_t k = 0; k < p.fma; ++k) {   // dependent chains, three per pixel
            x = __builtin_fmaf(x, 1.0001f, y);
            y = __builtin_fmaf(y, 0.9999f, z);
            z = __builtin_fmaf(z, 1.0002f, x);
        }
        __builtin_nontemporal_store(u2v{__float_as_uint(x + y) ^ fold, __float_as_uint(z)}, p.out + (size_t)(tyi * 4u + ly) * p.width + txi * 16u + lx);
    }
    if (lane == 0) {
        atomicAdd(p.wait_ticks + (blockIdx.x & 255u), waited);
        atomicAdd(p.waits + (blockIdx.x & 255u), n);
    }
}

static bool g_quiet = false;           // --brief: run() prints nothing, main() prints one JSON line
static double g_last_us = 0.0, g_last_in_flight_kb = 0.0, g_last_wait_us = 0.0;
template <int LDS_BYTES, int TAP_MODE = 0, int AHEAD = 0>
static void run(const char* name, params p, int waves_per_simd, int sets, void** pos, void** nrm, void** ids, void** pyr, void** out, double ticks_per_us,
                int bands = 1) {
    static hipStream_t streams[2] = {nullptr, nullptr};
    if (!streams[1]) {   // (two streams of their own: the legacy default stream would serialise with the other one)
        CHECK(hipStreamCreateWithFlags(&streams[0], hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&streams[1], hipStreamNonBlocking));
    }
    const uint32_t all_tiles = p.ntiles;
    // resident waves: 256 CUs x 4 SIMDs x waves; the grid is four rounds of them, like the pass's
    const uint32_t resident = 256u * 4u * (uint32_t)waves_per_simd;
    const uint32_t grid = resident * 4u;
    p.j_step = grid >> 3;
    CHECK(hipMemset(p.wait_ticks, 0, 256 * 8));
    CHECK(hipMemset(p.waits, 0, 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int warm = 1500, K = 200;    // (the clocks ramp over the first ~10 ms of load: ~0.15 s of launches before the timed ones)
    for (int k = 0; k < warm + K; ++k) {
        if (k == warm) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(p.wait_ticks, 0, 256 * 8));
            CHECK(hipMemset(p.waits, 0, 256 * 8));
            CHECK(hipEventRecord(e0, streams[0]));
        }
        const int s = k % sets;      // cold inputs: every launch another set (> 1.2 GB rotates through the 256 MiB cache)
        p.pos = (const f4v*)pos[s]; p.nrm = (const f4v*)nrm[s]; p.ids = (const uint32_t*)ids[s]; p.pyr = (const u4v*)pyr[s]; p.out = (u2v*)out[s];
        for (int b = 0; b < bands; ++b) {   // (no join between frames: band b of frame k + 1 follows band b of frame k on its stream)
            params q = p;
            const uint32_t rows = all_tiles / p.tiles_x, r0 = rows * b / bands, r1 = rows * (b + 1) / bands;
            q.tile_begin = r0 * p.tiles_x;
            q.ntiles = (r1 - r0) * p.tiles_x;
            q.j_step = (grid / bands) >> 3;      // (each band a grid of its own share of the wave slots' rounds)
            hipLaunchKernelGGL((pattern_kernel<LDS_BYTES, TAP_MODE, AHEAD>), dim3(grid / bands), dim3(64), 0, streams[b], q);
        }
    }
    if (bands > 1) {
        hipEvent_t j;
        CHECK(hipEventCreate(&j));
        CHECK(hipEventRecord(j, streams[1]));
        CHECK(hipStreamWaitEvent(streams[0], j, 0));
    }
    CHECK(hipEventRecord(e1, streams[0]));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / K;
    std::vector<unsigned long long> wt(256), wn(256);
    CHECK(hipMemcpy(wt.data(), p.wait_ticks, 256 * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(wn.data(), p.waits, 256 * 8, hipMemcpyDeviceToHost));
    double ticks = 0, cnt = 0;
    for (int i = 0; i < 256; ++i) { ticks += (double)wt[i]; cnt += (double)wn[i]; }
    const double px = (double)p.width * p.height;
    const double read_bytes = px * (36.0 + 8.0 * (p.taps ? 1.0 : 0.0));     // planes + one level-0-equivalent texel per pixel (the pass's figure)
    const double all_bytes = read_bytes + px * 8.0;
    const double lat_us = ticks / cnt / ticks_per_us;
    const double in_flight_per_cu = read_bytes / 256.0 / us * lat_us;
    g_last_us = us;
    g_last_in_flight_kb = in_flight_per_cu / 1024.0;
    g_last_wait_us = lat_us;
    if (g_quiet) return;
    if (p.taps == 0 || AHEAD)   // (the skeleton's loads are consumed straight behind the wait, and with a tile requested ahead the compiler's own counted wait precedes the timed one)
        printf("%-34s %7.1f us  %5.2f TB/s on %.0f B/px\n", name, us, all_bytes / us / 1e6, all_bytes / px);
    else
        printf("%-34s %7.1f us  %5.2f TB/s on %.0f B/px  wait %6.0f ticks = %5.2f us  reads in flight per CU %6.1f KB\n", name, us,
               all_bytes / us / 1e6, all_bytes / px, ticks / cnt, lat_us, in_flight_per_cu / 1024.0);
}

__global__ void tick_kernel(unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}

int main(int argc, char** argv) {
    const bool brief = argc > 1 && std::string(argv[1]) == "--brief";
    const bool explore = argc > 1 && std::string(argv[1]) == "--explore";
    g_quiet = brief;
    const uint32_t W = 3840, H = 2160;
    const size_t px = (size_t)W * H;
    const int sets = 4;
    void *pos[sets], *nrm[sets], *ids[sets], *pyr[sets], *out[sets];
    const size_t pyr_bytes = px * 8 * 4 / 3 + 4096;
    for (int s = 0; s < sets; ++s) {
        CHECK(hipMalloc(&pos[s], px * 16)); CHECK(hipMalloc(&nrm[s], px * 16)); CHECK(hipMalloc(&ids[s], px * 4));
        CHECK(hipMalloc(&pyr[s], pyr_bytes)); CHECK(hipMalloc(&out[s], px * 8));
        CHECK(hipMemset(pos[s], 0x11, px * 16)); CHECK(hipMemset(nrm[s], 0x22, px * 16)); CHECK(hipMemset(ids[s], 0x03, px * 4));
        CHECK(hipMemset(pyr[s], 0x3c, pyr_bytes));
    }
    params p{};
    void* lut;
    CHECK(hipMalloc(&lut, 16 * 16384 * 16));
    CHECK(hipMemset(lut, 0, 16 * 16384 * 16));
    p.lut = (const f4v*)lut;
    CHECK(hipMalloc(&p.wait_ticks, 256 * 8));
    CHECK(hipMalloc(&p.waits, 256 * 8));
    p.width = W; p.height = H; p.tiles_x = (W + 63) / 64; p.ntiles = p.tiles_x * ((H + 3) / 4);
    // s_memtime ticks per microsecond (against the constant 100 MHz clock)
    unsigned long long* d; unsigned long long h[3];
    CHECK(hipMalloc(&d, 24));
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, d);
    CHECK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    const double ticks_per_us = (double)h[0] / ((double)h[1] / 100.0);
    if (brief) {
        // bench.py's roofline.pattern_ceiling: the four figures the pass is held against, one JSON line
        double r[6], kb[2], wait[2];
        p.fma = 0;
        p.taps = 0; p.scatter = 48;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2); r[0] = g_last_us;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 1); r[1] = g_last_us;
        p.taps = 4;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2); r[2] = g_last_us; kb[0] = g_last_in_flight_kb; wait[0] = g_last_wait_us;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 1); r[3] = g_last_us;
        p.scatter = 0;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2); r[4] = g_last_us; kb[1] = g_last_in_flight_kb; wait[1] = g_last_wait_us;
        run<1024>("", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 1); r[5] = g_last_us;
        printf("{\"planes_and_store_only_us\": {\"two_bands\": %.1f, \"one_call\": %.1f}, "
               "\"pattern_taps_displaced_48px_us\": {\"two_bands\": %.1f, \"one_call\": %.1f, \"reads_in_flight_per_cu_kb\": %.1f, \"loads_outstanding_us\": %.2f}, "
               "\"pattern_taps_not_displaced_us\": {\"two_bands\": %.1f, \"one_call\": %.1f, \"reads_in_flight_per_cu_kb\": %.1f, \"loads_outstanding_us\": %.2f}}\n",
               r[0], r[1], r[2], r[3], kb[0], wait[0], r[4], r[5], kb[1], wait[1]);
        return 0;
    }
    printf("s_memtime: %.1f ticks per us\n", ticks_per_us);
    if (explore) {
        // where the taps' cost comes from: the direction of the displacement field, and the order an XCD walks its band in
        p.fma = 0; p.taps = 4;
        for (uint32_t field : {0u, 1u, 2u})
            for (uint32_t sc : {48u, 200u, 600u}) {
                p.field = field; p.scatter = sc;
                char nm[96];
                snprintf(nm, sizeof nm, "field %s, +-%u px", field == 0 ? "x and y" : field == 1 ? "x only" : "y only", sc);
                run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
            }
        p.field = 0; p.scatter = 48;
        // is a gather's cost its bytes through the L1 (64 B/clk per CU) or its instruction?
        for (uint32_t taps : {1u, 2u, 3u, 4u}) {
            p.taps = taps;
            char nm[96];
            snprintf(nm, sizeof nm, "%u taps of 16 bytes, all cache hits", taps);
            run<1024, 4>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
        }
        p.taps = 4;
        run<1024, 0>("4 taps of 16 bytes", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
        run<1024, 5>("4 taps of 8 bytes (one texel per lane and row)", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
        run<1024, 6>("4 taps of 4 bytes", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
        run<1024, 0>("4 taps of 16 bytes, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
        run<1024, 5>("4 taps of 8 bytes, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
        run<1024, 6>("4 taps of 4 bytes, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
        if (argc > 2) return 0;
        for (uint32_t order : {0u, 1u, 2u, 4u, 6u, 10u, 15u, 30u}) {
            if (order && p.tiles_x % order) continue;
            p.order = order;
            char nm[96];
            snprintf(nm, sizeof nm, "order %u, pattern, one call", order);
            run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
            snprintf(nm, sizeof nm, "order %u, pattern, two bands", order);
            run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
            p.taps = 0;
            snprintf(nm, sizeof nm, "order %u, planes + store only, two bands", order);
            run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
            p.taps = 4;
        }
        return 0;
    }

    p.scatter = 48; p.fma = 0;
    printf("-- planes + store only (the streaming skeleton)\n");
    p.taps = 0;
    run<160 * 1024 / 4>("1 wave per SIMD", p, 1, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<160 * 1024 / 8>("2 waves per SIMD", p, 2, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<160 * 1024 / 16>("4 waves per SIMD", p, 4, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024>("8 waves per SIMD", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    printf("-- the pass's pattern: planes + 4 scattered 16-byte taps + LUT line + store\n");
    p.taps = 4;
    run<160 * 1024 / 4>("1 wave per SIMD", p, 1, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<160 * 1024 / 8>("2 waves per SIMD", p, 2, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<160 * 1024 / 16>("4 waves per SIMD", p, 4, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024>("8 waves per SIMD", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    for (uint32_t sc : {0u, 8u, 200u}) {
        p.scatter = sc;
        char nm[64];
        snprintf(nm, sizeof nm, "8 waves, tap scatter +-%u px", sc);
        run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    }
    p.scatter = 48;
    printf("-- what in a tap costs: 8 waves per SIMD, one launch per frame\n");
    run<1024, 1>("pairs forced to 16-byte alignment", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024, 2>("two 8-byte loads per pair", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024, 3>("a level's taps as ONE window load + LDS", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024, 3>("... two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.scatter = 0;
    run<1024, 3>("... taps not displaced, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.scatter = 48;
    run<1024, 4>("taps folded onto 64x64 texels (all cache hits)", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024, 4>("... two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.taps = 2;
    run<1024>("level 0 taps only (2 of 4)", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    p.taps = 4;
    printf("-- every frame as two row bands on two streams (bench.py's step), 8 waves per SIMD\n");
    p.taps = 0;
    run<1024>("planes + store only, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.taps = 4;
    run<1024>("the pass's pattern, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.scatter = 0;
    run<1024>("... tap scatter +-0 px, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.scatter = 48;
    printf("-- two tiles of a wave in flight (the next tile's plane rows requested before this tile's wait), 8 waves per SIMD\n");
    p.taps = 0;
    run<1024, 0, 1>("planes + store only, one call", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    p.taps = 4;
    run<1024, 0, 1>("the pass's pattern, one call", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    run<1024, 0, 1>("the pass's pattern, two bands", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us, 2);
    p.fma = 64;
    run<1024, 0, 1>("... with 192 fma per pixel, one call", p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    p.fma = 0;
    printf("-- the same at 8 waves per SIMD with dependent fma per pixel behind the wait (3 chains x N)\n");
    for (uint32_t f : {16u, 32u, 64u, 96u, 128u}) {
        p.fma = f;
        char nm[64];
        snprintf(nm, sizeof nm, "8 waves, %u fma per pixel", 3 * f);
        run<1024>(nm, p, 8, sets, pos, nrm, ids, pyr, out, ticks_per_us);
    }
    return 0;
}
