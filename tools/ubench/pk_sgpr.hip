// v_pk_fma_f32 / v_pk_mul_f32 with a scalar-register source on gfx950: does the packed form hide the scalar-operand
// penalty of the plain VALU (4.0 instead of 2.3 cycles per wave64 instruction)?
// Build: hipcc --offload-arch=gfx950 -O3 pk_sgpr.hip -o pk_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float t = threadIdx.x * 1e-3f;
    v2f p0 = {t, t + 1}, p1 = {t + 2, t + 3}, p2 = {t + 4, t + 5}, p3 = {t + 6, t + 7};
    v2f sa = {a, b};
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0)   // packed, all vector operands
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (KIND == 1)   // packed, one scalar pair source
            asm volatile("v_pk_fma_f32 %0, %4, %1, %2\n v_pk_fma_f32 %1, %4, %2, %3\n v_pk_fma_f32 %2, %4, %3, %0\n v_pk_fma_f32 %3, %4, %0, %1"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sa));
        if (KIND == 2)   // packed, scalar pair source, low half broadcast to both lanes
            asm volatile("v_pk_fma_f32 %0, %4, %1, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %2, %3 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %3, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %0, %1 op_sel_hi:[0,1,1]"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sa));
        if (KIND == 3)   // packed multiply with a scalar pair
            asm volatile("v_pk_mul_f32 %0, %4, %1\n v_pk_mul_f32 %1, %4, %2\n v_pk_mul_f32 %2, %4, %3\n v_pk_mul_f32 %3, %4, %0"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sa));
        if (KIND == 4)   // plain fma with a scalar source (reference: 4.0)
            asm volatile("v_fma_f32 %0, %4, %0, %1\n v_fma_f32 %1, %4, %1, %2\n v_fma_f32 %2, %5, %2, %3\n v_fma_f32 %3, %5, %3, %0"
                         : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x) : "s"(a), "s"(b));
        if (KIND == 6)   // plain fma, scalar-source and vector-only instructions alternating
            asm volatile("v_fma_f32 %0, %4, %0, %1\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %5, %2, %3\n v_fma_f32 %3, %3, %0, %1"
                         : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x) : "s"(a), "s"(b));
        if (KIND == 7)   // one scalar-source instruction in four
            asm volatile("v_fma_f32 %0, %4, %0, %1\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %0, %2, %3\n v_fma_f32 %3, %3, %0, %1"
                         : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x) : "s"(a), "s"(b));
        if (KIND == 8)   // vector-only
            asm volatile("v_fma_f32 %0, %1, %0, %1\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %0, %2, %3\n v_fma_f32 %3, %3, %0, %1"
                         : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x));
        if (KIND == 5)   // packed add
            asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND>
void run(const char* name, float* d, int flops_per_instr) {
    const int blocks = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = (double)blocks * 4 * N_ITER * 4;
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / instr;
    printf("%-44s %.3f ms  %.2f cycles per wave-instruction, %.2f per lane-op\n", name, ms, cyc, cyc / flops_per_instr);
}
int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    run<0>("v_pk_fma_f32 vgpr", d, 2); run<1>("v_pk_fma_f32 sgpr pair", d, 2); run<2>("v_pk_fma_f32 sgpr broadcast", d, 2);
    run<3>("v_pk_mul_f32 sgpr pair", d, 2); run<4>("v_fma_f32 sgpr", d, 1); run<5>("v_pk_add_f32 vgpr", d, 2);
    run<6>("v_fma_f32 sgpr / vgpr alternating", d, 1); run<7>("v_fma_f32 one sgpr in four", d, 1); run<8>("v_fma_f32 vgpr", d, 1);
    return 0;
}
