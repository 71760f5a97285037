// Can a workgroup read a cache line, have ANOTHER XCD write part of it (write-through, agent scope) in the same launch, and
// then have an ordinary load on the first XCD return the new bytes?  (The question behind mip levels produced and consumed
// inside one launch: consecutive levels of a packed pyramid can share a 128-byte line.)
//   stage 1: workgroup r (XCD r) reads word 0 of its lines        — load kind: 0 plain, 1 agent-scope atomic (sc1), 2 nontemporal
//            then: 0 nothing, 1 `buffer_inv sc1`, 2 acquire fence at agent scope
//   stage 2: another XCD stores word 16 of those lines (agent-scope atomic store; or a plain write-back store, the positive
//            control: its bytes stay in the writer's L2), waits for the stores, signals
//   stage 3: the stage-1 workgroup itself (same CU: L1 + L2) and workgroup r + 8 (same XCD, another CU or not: L2) read word 16
//            with a plain load.  Stale = the value from before stage 2.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stale_line.hip -o /tmp/stale_line && /tmp/stale_line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr unsigned kMaxLines = 4096, kWordsPerLine = 32;

// (a `volatile` access is compiled to a system-scope one — sc0 sc1 — on this target: the plain load is spelled out)
__device__ inline unsigned plain_load(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void wait_for(unsigned* flag, unsigned target) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
}
__device__ inline void signal(unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void zero(unsigned* p, unsigned n) {
    for (unsigned i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) p[i] = 0u;
}

__global__ __launch_bounds__(64) void probe(unsigned* data, unsigned* flags, unsigned* out_same, unsigned* out_other, unsigned* sink,
                                            int load_kind, int inv_kind, int store_kind, unsigned* xcc_of_block, unsigned kLines) {
    const unsigned b = blockIdx.x, xcd = b & 7u;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_of_block[b] = id & 15u;
    }
    if (b < 8u) {
        unsigned acc = 0;
        for (unsigned l = xcd + 8u * threadIdx.x; l < kLines; l += 8u * 64u) {   // lines with l % 8 == xcd
            unsigned* w = data + (size_t)l * kWordsPerLine;
            if (load_kind == 0) acc += plain_load(w);
            else if (load_kind == 1) acc += __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else acc += __builtin_nontemporal_load(w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (inv_kind == 1) asm volatile("buffer_inv sc1" ::: "memory");
        else if (inv_kind == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (acc == 0xFFFFFFFFu) sink[0] = acc;
        signal(flags + 0);
        wait_for(flags + 0, 8u);
        // the lines read by XCD (xcd + 1 + k) % 8, k = 0..6 by line: this workgroup writes those whose writer it is
        if (store_kind != 2) {
            for (unsigned l = threadIdx.x; l < kLines; l += 64u) {
                const unsigned reader = l & 7u, writer = (reader + 1u + (l >> 3) % 7u) & 7u;
                if (writer != xcd) continue;
                if (store_kind == 1) __hip_atomic_store(data + (size_t)l * kWordsPerLine + 16u, l + 1000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else data[(size_t)l * kWordsPerLine + 16u] = l + 1000u;   // (write-back: stays in the writer's L2 — the positive control)
            }
        }
        if (store_kind == 2) wait_for(flags + 2, 8u);
        signal(flags + 1);
        wait_for(flags + 1, 8u);
        for (unsigned l = xcd + 8u * threadIdx.x; l < kLines; l += 8u * 64u) out_same[l] = plain_load(data + (size_t)l * kWordsPerLine + 16u);
    } else {
        if (store_kind == 2) {   // the positive control of the L1: the SAME XCD's other workgroup writes (plain stores into the shared L2)
            wait_for(flags + 0, 8u);
            for (unsigned l = xcd + 8u * threadIdx.x; l < kLines; l += 8u * 64u) data[(size_t)l * kWordsPerLine + 16u] = l + 1000u;
            signal(flags + 2);
        }
        wait_for(flags + 1, 8u);
        for (unsigned l = xcd + 8u * threadIdx.x; l < kLines; l += 8u * 64u) out_other[l] = plain_load(data + (size_t)l * kWordsPerLine + 16u);
    }
}

int main(int argc, char** argv) {
    const unsigned kLines = argc > 1 ? (unsigned)atoi(argv[1]) : kMaxLines;
    unsigned *data, *flags, *out_same, *out_other, *sink, *xcc;
    hipMalloc(&xcc, 64);
    hipMalloc(&data, kMaxLines * kWordsPerLine * 4);
    hipMalloc(&flags, 256);
    hipMalloc(&out_same, kMaxLines * 4);
    hipMalloc(&out_other, kMaxLines * 4);
    hipMalloc(&sink, 4);
    const char* loads[] = {"plain load", "agent-scope atomic load", "nontemporal load"};
    const char* invs[] = {"nothing", "buffer_inv sc1", "acquire fence (agent)"};
    std::vector<unsigned> a(kLines), b(kLines);
    for (int sk = 0; sk < 3; ++sk)
    for (int lk = 0; lk < 3; ++lk)
        for (int ik = 0; ik < 3; ++ik) {
            unsigned stale_same = 0, stale_other = 0, wrong = 0;
            for (int rep = 0; rep < 8; ++rep) {
                zero<<<256, 64>>>(data, kLines * kWordsPerLine);
                zero<<<1, 64>>>(flags, 64);
                zero<<<64, 64>>>(out_same, kLines);
                zero<<<64, 64>>>(out_other, kLines);
                probe<<<16, 64>>>(data, flags, out_same, out_other, sink, lk, ik, sk, xcc, kLines);
                if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
                hipMemcpy(a.data(), out_same, kLines * 4, hipMemcpyDeviceToHost);
                hipMemcpy(b.data(), out_other, kLines * 4, hipMemcpyDeviceToHost);
                for (unsigned l = 0; l < kLines; ++l) {
                    stale_same += a[l] == 0u;
                    stale_other += b[l] == 0u;
                    wrong += (a[l] != 0u && a[l] != l + 1000u) || (b[l] != 0u && b[l] != l + 1000u);
                }
            }
            printf("%s, stage 1 = %-24s then %-22s: stale on the same CU %5u, on the same XCD %5u of %u lines (8 runs); other values %u\n",
                   sk == 2 ? "same XCD, plain   " : sk ? "agent-scope stores" : "plain stores      ", loads[lk], invs[ik], stale_same, stale_other, kLines * 8, wrong);
        }
    unsigned ids[16];
    hipMemcpy(ids, xcc, 64, hipMemcpyDeviceToHost);
    printf("XCC of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", ids[i]);
    printf("\n");
    return 0;
}
