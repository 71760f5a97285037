// Does a wave64 VALU instruction cost less on gfx950 when half of its lanes are switched off?  (SIMD-32: a wave64
// instruction issues over two passes of 32 lanes.)  Build: hipcc --offload-arch=gfx950 -O3 exec_mask.hip -o exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
// MODE 0: all 64 lanes; 1: lanes 0..31 only; 2: lanes 32..63 only; 3: even lanes only; 4: lanes 0..15; 5: one lane
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    const unsigned lane = threadIdx.x & 63u;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32u : MODE == 2 ? lane >= 32u : MODE == 3 ? (lane & 1u) == 0u : MODE == 4 ? lane < 16u : lane == 5u;
    if (on) {
        for (int i = 0; i < N_ITER; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
void run(const char* name, float* d) {
    const int blocks = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = (double)blocks * 4 * N_ITER * 8;   // wave-instructions
    printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
}
int main() {
    float* d; hipMalloc(&d, 8192 * 256 * 4);
    run<0>("all 64 lanes", d); run<1>("lanes 0..31", d); run<2>("lanes 32..63", d); run<3>("even lanes", d); run<4>("lanes 0..15", d); run<5>("one lane", d);
    return 0;
}
