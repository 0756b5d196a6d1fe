// Helpers shared by the quad-major LDS-DMA kernels (fine_quad.hip, cascade_quad.hip).
#pragma once
#include "common.hpp"

namespace casmtr {

// one 4 KB chunk: 4 LDS-DMA instructions, lane-linear 1 KB each.  The source offsets o1..o3 are pre-biased by -1024, -2048, -3072
// (the immediate offset is added to BOTH addresses: tools/probes/glds_offset.hip) and every offset by +3072 against a base pointer
// that is 3072 bytes low, so that they stay non-negative.  M0 is neither saved nor restored: nothing else in these kernels uses it
// (tools/check_quad_isa.py).
__device__ __forceinline__ void glds_chunk(const float* base_m3072, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:3072"
                 :: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base_m3072), "s"(lds_dst) : "memory");
}
// the first n (1..3) instructions of a chunk
__device__ __forceinline__ void glds_chunk1(const float* base_m3072, unsigned o0, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(o0), "s"(base_m3072), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_chunk2(const float* base_m3072, unsigned o0, unsigned o1, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                 :: "v"(o0), "v"(o1), "s"(base_m3072), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_chunk3(const float* base_m3072, unsigned o0, unsigned o1, unsigned o2, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:2048"
                 :: "v"(o0), "v"(o1), "v"(o2), "s"(base_m3072), "s"(lds_dst) : "memory");
}
// wait until at most n (wave-uniform, 0..4) vector-memory operations are outstanding
__device__ __forceinline__ void glds_wait_dyn(int n) {
    if (n >= 4) glds_wait<4>();
    else if (n == 3) glds_wait<3>();
    else if (n == 2) glds_wait<2>();
    else if (n == 1) glds_wait<1>();
    else glds_wait<0>();
}

__device__ __forceinline__ unsigned row16_sum_u32(unsigned v) {
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    v += dpp_u32<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max_f32(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}

__device__ __forceinline__ void wave_lds_fence() {   // LDS writes of this wave are visible to its own later reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace casmtr
