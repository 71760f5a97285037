// VALU issue-rate microbenchmark for gfx950: wave64 instructions per cycle per SIMD for the instruction
// classes the shading kernel is made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                         x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b); }
        if (KIND == 1) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
                         p0 = __builtin_elementwise_fma(p0, pb, pa); p1 = __builtin_elementwise_fma(p1, pb, pa); p2 = __builtin_elementwise_fma(p2, pb, pa); p3 = __builtin_elementwise_fma(p3, pb, pa); }
        if (KIND == 2) { x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                         x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7); }
        if (KIND == 3) { x0 = fmaxf(x0, a) ; x1 = fminf(x1, b); x2 = fmaxf(x2, a); x3 = fminf(x3, b); x4 = fmaxf(x4, b); x5 = fminf(x5, a); x6 = fmaxf(x6, b); x7 = fminf(x7, a);
                         asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 4) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b; }
        if (KIND == 5) { asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 6) { asm volatile("v_sqrt_f32 %0, %0\n v_rsq_f32 %1, %1\n v_exp_f32 %2, %2\n v_log_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_rsq_f32 %5, %5\n v_exp_f32 %6, %6\n v_log_f32 %7, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 7) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_f32_ubyte0 %7, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "vcc"); }
        if (KIND == 8) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 9) { asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %4\n v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 10) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "vcc"); }
        if (KIND == 11) { asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %3, %4, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %5, %6, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %6, %7, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %7, %0, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %0, %1, %7 op_sel_hi:[1,0,0]"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 12) { asm volatile("v_add_u32 %0, %0, %1\n v_lshlrev_b32 %1, 2, %1\n v_and_b32 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mad_u32_u24 %4, %4, %5, %6\n v_add_lshl_u32 %5, %5, %6, 3\n v_min_u32 %6, %6, %7\n v_add_u32 %7, %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 13) { asm volatile("v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %1, %8, %1, %2\n v_fma_f32 %2, %9, %2, %3\n v_fma_f32 %3, %9, %3, %4\n v_mul_f32 %4, 0x3b808081, %4\n v_mul_f32 %5, %8, %5\n v_fmac_f32 %6, %9, %7\n v_fmac_f32 %7, %8, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b)); }
        if (KIND == 14) { asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 15) { asm volatile("v_mul_f32 %0, %0, %1\n v_fmac_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_fmac_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 16) { asm volatile("v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %1, %8, %1, %2\n v_fma_f32 %2, %9, %2, %3\n v_fma_f32 %3, %9, %3, %4\n v_fma_f32 %4, %8, %4, %5\n v_fma_f32 %5, %8, %5, %6\n v_fma_f32 %6, %9, %6, %7\n v_fma_f32 %7, %9, %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b)); }
        if (KIND == 17) { asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_add_f32 %2, %9, %2\n v_add_f32 %3, %9, %3\n v_mul_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_max_f32 %6, %9, %6\n v_fmac_f32 %7, %9, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "s"(b)); }
        if (KIND == 18) { asm volatile("v_mul_f32 %0, 0x3b808081, %0\n v_mul_f32 %1, 0x3b808081, %1\n v_add_f32 %2, 0x3b808081, %2\n v_max_f32 %3, 0x34000000, %3\n v_mul_f32 %4, 0x3b808081, %4\n v_max_f32 %5, 0x34000000, %5\n v_add_f32 %6, 0x3b808081, %6\n v_mul_f32 %7, 0x3b808081, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 19) { asm volatile("v_mul_f32 %0, 2.0, %0\n v_add_f32 %1, 1.0, %1\n v_sub_f32 %2, 1.0, %2\n v_max_f32 %3, 0, %3\n v_mul_f32 %4, 0.5, %4\n v_fma_f32 %5, %5, 2.0, 2.0\n v_add_f32 %6, -1.0, %6\n v_mul_f32 %7, 4.0, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 20) { asm volatile("v_fma_f32 %0, -%0, %1, %2\n v_fma_f32 %1, %1, -%2, %3\n v_mul_f32_e64 %2, %2, -%3\n v_fma_f32 %3, %3, %4, -%5\n v_max_f32_e64 %4, -%4, -%5\n v_fma_f32 %5, -%5, %6, 1.0\n v_mul_f32_e64 %6, -%6, %7\n v_fma_f32 %7, |%7|, %0, %1"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (KIND == 21) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "vcc"); }
        if (KIND == 22) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "vcc"); }
        if (KIND == 23) { asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_f32_ubyte0 %7, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND> void run(const char* name, int blocks, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * N_ITER * 8;          // wave-instructions
    double per_simd_per_s = winstr / 1024.0 / (ms * 1e-3);    // 1024 SIMDs
    printf("%-28s blocks %5d  %.3f ms  %.2f Gwave-instr/s/SIMD -> %.2f cycles/instr @2.4GHz\n", name, blocks, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    for (int blocks : {8192}) {
        run<0>("v_fma_f32", blocks, d); run<1>("v_pk_fma_f32", blocks, d); run<4>("v_mul/add_f32", blocks, d); run<3>("v_max/min_f32", blocks, d);
        run<14>("v_fma_f32 (3 vgpr srcs)", blocks, d); run<15>("v_mul/fmac/add 2-src vgpr", blocks, d); run<13>("fma/mul with sgpr/literal", blocks, d);
        run<8>("v_mov_b32", blocks, d); run<9>("v_sub/max/min_f32", blocks, d); run<10>("v_cmp+v_cndmask", blocks, d); run<11>("v_fma_mix_f32", blocks, d); run<12>("int add/shift/and/mul24", blocks, d);
        run<16>("v_fma_f32 sgpr src", blocks, d); run<17>("VOP2 mul/add/sub/max sgpr src0", blocks, d); run<18>("VOP2 with 32-bit literal", blocks, d);
        run<19>("inline constants", blocks, d); run<20>("VOP3 neg/abs modifiers", blocks, d); run<21>("v_cndmask_b32", blocks, d); run<22>("v_cmp_lt_f32", blocks, d);
        run<23>("max3/floor/cvt", blocks, d);
        run<2>("v_rcp_f32", blocks, d); run<6>("sqrt/rsq/exp/log", blocks, d); run<5>("v_cvt_f32_f16", blocks, d); run<7>("cndmask/floor/cvt", blocks, d);
    }
    return 0;
}
