// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CASMTR_WAVE 64
#define NEG_FILL (-1e9f)  // INF = 1e9 in coarse_matching.py:6 / cascade_matching.py:8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace casmtr {

// Order-preserving map float -> uint32 (total order of the finite floats; -0 < +0, irrelevant for us).
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// DPP controls (gfx9): quad_perm = 0x00..0xFF, row_shr:n = 0x110+n, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}

// All-lanes max / sum over each 16-lane DPP row (4 independent rows per wave).
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v));  // row_half_mirror
    v = max(v, dpp_u32<0x140>(v));  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum_f32(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}

// Whole-wave (64 lane) reductions, result valid in every lane.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = row16_max_u32(v);
    v = max(v, (unsigned)__shfl_xor((int)v, 16));
    v = max(v, (unsigned)__shfl_xor((int)v, 32));
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = row16_sum_f32(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    return ord2f(wave_max_u32(f2ord(v)));
}

__device__ __forceinline__ float div_scalar(float x, float s, float inv_s, int recip) {
    // torch CPU: true fp32 division; torch GPU kernels: x * fl32(1/s)  (see oracle/casmtr_oracle.c header)
    return recip ? __fmul_rn(x, inv_s) : __fdiv_rn(x, s);
}

// prof.hip
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
    ~ProfScope() { prof_end(id, s); }
};

}  // namespace casmtr

#define CASMTR_CHECK_LAUNCH()                         \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

#define CASMTR_ERR_UNSUPPORTED 1001  // shape outside what the kernel was built for (caller must not fall back to CPU)
