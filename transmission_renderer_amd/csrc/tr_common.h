// tr_common.h — what every gfx950 device header of this library shares: constants, the scalar-table address
// space, and the one-instruction math helpers.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/tr_shade.h"

namespace tr {

constexpr float kEpsilon = 1.1920929e-07f;  // core::f32::EPSILON (glam-pbr/src/lib.rs:95)
constexpr float kPi = 3.14159265358979323846f;
constexpr float kFrac1Pi = 0.318309886183790671538f;
constexpr float kLog2e = 1.44269504088896340736f;

typedef float v2f __attribute__((ext_vector_type(2)));  // one v_pk_*_f32 operand

// Tables that a whole wave reads at one address (material, lights, cluster lists, level geometry)
// are addressed through the constant address space: a uniform load from it is always issued on the
// scalar unit (s_load into SGPRs), also inside the material / cluster loops where the compiler
// cannot otherwise prove that no store clobbers them.  (Constant and global are the same memory.)
#define TR_CONSTANT __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const TR_CONSTANT T* as_constant(const T* p) {
    return (const TR_CONSTANT T*)(p);
}

// The lane's index in its wave, derived on the spot (two instructions, opaque to the optimiser: neither hoisted nor
// shared between uses) — for kernels that cannot afford a register for it across a long pixel.
__device__ __forceinline__ uint32_t wave_lane() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Makes a (uniform) pointer opaque to the optimiser: loads through the result cannot be hoisted above
// this point, so their live ranges start here.
template <class T>
__device__ __forceinline__ T* launder(T* p) {
#ifndef TR_NO_LAUNDER   // (tools/build_variant.py experiment: what the allocator does when every launch constant may be hoisted)
    asm volatile("" : "+s"(p));
#endif
    return p;
}

// Memory access as (uniform base pointer in scalar registers) + (32-bit byte offset per lane): the addressing form
// global_load / global_store take directly (saddr + voffset), so an access costs the one or two instructions that
// form the offset instead of 64-bit pointer arithmetic in vector registers.  Every buffer of the hot kernels is
// below 4 GiB (checked on the host).  Products of coordinates (< 2^24) use the full-rate 24-bit multiply-add.
template <class T>
__device__ __forceinline__ T ld(const void* base, uint32_t byte_offset) {
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_offset);
}
// Streamed-once data (the G-buffer planes): non-temporal, so it does not displace the tables and pyramid texels
// the same kernel keeps re-reading from L2.
template <class T>
__device__ __forceinline__ T ld_stream(const void* base, uint32_t byte_offset) {
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_offset));
}
template <class T>
__device__ __forceinline__ void st(void* base, uint32_t byte_offset, T value) {
    *reinterpret_cast<T*>(static_cast<char*>(base) + byte_offset) = value;
}
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
// The wave's lane mask of a predicate, straight from the compare (HIP's __ballot goes through v_cndmask + v_cmp;
// the builtin does too unless the predicate is ONE compare, so callers fold their conditions into a key first).
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// A scalar value the optimiser cannot relate to its source: inside `if (per_lane == s)` LLVM's equality propagation
// substitutes the per-lane value for the scalar `s`, and every table read indexed by it degrades to a per-lane vector
// load; indexing with opaque(s) instead keeps the reads on the scalar unit.
__device__ __forceinline__ uint32_t opaque(uint32_t s) {
    asm volatile("" : "+s"(s));
    return s;
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over every address space:
// it also waits for the wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)), a full memory round trip per barrier — the
// mip-chain kernels store a level to global memory and hand it to the next step through LDS, seven barriers deep.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------ small helpers
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_max(v2f a, float b) { return v2f{fmaxf(a.x, b), fmaxf(a.y, b)}; }

struct f3 {
    float x, y, z;
};

}  // namespace tr
