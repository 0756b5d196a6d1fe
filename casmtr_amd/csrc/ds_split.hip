// Dual-softmax similarity matrix on the f16 matrix pipe of gfx950, fp32-accurate, with exact index decisions.
//   CoarseMatching.forward: sim = einsum(feat_c0 / sqrt(C), feat_c1 / sqrt(C)) / T      src/model/functions/coarse_matching.py:59-63
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 vector rate, 1/16 of the f16 / bf16 MFMA rate, and the exact kernel in
// matching.hip already saturates it.  Here each operand row is normalised by a power of two so that its largest element lies
// in [512, 1024) and split into two f16 terms, a = hi + lo + r, |r| <= 2^-22 |a| (lo is a normal f16 for every element within
// 2^-13 of the row maximum; smaller elements round at 2^-25 absolute = 2^-34 of the row maximum).  v_mfma_f32_32x32x16_f16
// multiplies f16 exactly and accumulates in fp32:
//      a.b  ~  lo_a.hi_b + hi_a.lo_b + hi_a.hi_b          (dropped: lo.lo and the r terms, <= 3 x 2^-22 sum|a b|)
// i.e. 3/16 of the fp32 MFMA time.  Error budget against the oracle's chain (fmaf over c ascending of the 1/sqrt(C)-scaled
// operands, then / T), all relative to sum_c |a_c b_c| / (C T) <= |a_i| |b_j| / (C T):
//      the chain itself  (C + 3) 2^-24  (worst case, C = 256: 1.55e-5)      split 3 x 2^-22 = 7.2e-7
//      48 fp32 MFMA accumulations + 2 scalings  <= 50 x 2^-23 = 6.0e-6      => e_ij = 2^-15 |a_i| |b_j| / (C T)  (3.05e-5)
// An entry can be its row's exact maximum only if its approximate logit is >= max~ - 2 e_i (e_i with max_j |b_j|): those are
// the row's candidates (collected by pass 2: ds_flagged_kernel on the recomputed flagged segments, or ds_conf_kernel<true> where the
// matrix is stored and streamed); a row with one candidate has
// its argmax, a row with several re-evaluates them with the exact chain here (ds_fix_kernel) -- about 0.4 % of the rows on
// random features.  Sums, probabilities and confidences use the approximate logits (relative error of exp < 1e-4 x |a||b|/(C T) x 0.3).
#include <stdlib.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "ds_common.hpp"
#include "../../include/casmtr_hip.h"

namespace casmtr {

// =================================================================================================== operand images
// Pass 1 (ds_rownorm_kernel, wave per row, coalesced 1 KB reads): row maximum -> exponent, row norm.
// Pass 2 (ds_split_kernel, lane <-> row, 32 B of a row per lane and step): normalise, split, write the tile image
//   img[b][rb][ks = c/32][kg = (c/8)%4][part hi/lo][row 128][8 f16]   -- 16 KB per (row block, k-stage), every 1 KB of it one
// wave-instruction of the GEMM's LDS-DMA and, inside a kg plane, the lane-linear 16-B runs its ds_read_b128 fragment loads want.
// (1) one wave per row: exponent of the row maximum, epilogue factor, row norm, batch maximum of the norms
//     The factor carries the padding mask in its SIGN (negative = masked row): the GEMM epilogues read it from there.
__global__ __launch_bounds__(256) void ds_rownorm_kernel(const float* __restrict__ f, const uint8_t* __restrict__ mask, int N, int C,
                                                         float inv_sqrtC, float k0, int* __restrict__ ex, float* __restrict__ fac,
                                                         float* __restrict__ nrm, int Np) {
    const int lane = threadIdx.x & 63;
    const int gi = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;   // row of the padded range [0, Np)
    if (gi >= Np) return;
    float mx = 0.f, ss = 0.f;
    if (gi < N) {
        const float* p = f + ((size_t)b * N + gi) * C;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    }
    mx = wave_max_f32(mx);
    ss = wave_sum_f32(ss);
    // largest element -> [512, 1024): products < 2^20, 256-term sums < 2^28, f16 hi parts far from 65504
    const int e = (mx > 0.f && mx < INFINITY) ? ilogbf(mx) - 9 : 0;
    // |a| / sqrt(C), rounded up: 1.001 covers the fp32 summation (<= C 2^-24 relative) and the square root
    const float nr = gi < N ? sqrtf(ss) * inv_sqrtC * 1.001f : 0.f;
    if (lane == 0) {
        ex[(size_t)b * Np + gi] = e;
        const float fv = gi < N ? ldexpf(k0, e) : 0.f;
        fac[(size_t)b * Np + gi] = (mask && gi < N && mask[(size_t)b * N + gi] == 0) ? -fv : fv;
        nrm[(size_t)b * Np + gi] = nr;
    }
}

// (1b) workgroup per (pair, side): maximum of the row norms (86 k same-address atomicMax from pass 1 would serialise: 2 ms)
__global__ __launch_bounds__(256) void ds_nmax_kernel(const float* __restrict__ na, const float* __restrict__ nb, int NpA, int NpB,
                                                      unsigned* __restrict__ namax, unsigned* __restrict__ nbmax) {
    __shared__ float wm[4];
    const int b = blockIdx.x, side = blockIdx.y, Np = side ? NpB : NpA;
    const float* p = (side ? nb : na) + (size_t)b * Np;
    float m = 0.f;
    for (int i = threadIdx.x; i < Np; i += 256) m = fmaxf(m, p[i]);
    m = wave_max_f32(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) (side ? nbmax : namax)[b] = __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}

// (2) wave = 64 rows x one k-stage (32 channels = one 128-B line per row): coalesced reads (8 lanes share a row), transposition
//     through a wave-private LDS slab to lane <-> row, then normalise, split and write 1 KB runs of the tile image
__global__ __launch_bounds__(256) void ds_split_kernel(const float* __restrict__ f, int N, int C, const int* __restrict__ ex,
                                                       _Float16* __restrict__ img, int NRB) {
    __shared__ float slabs[4 * CASMTR_SLAB_FLOATS];
    const int rb = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = blockIdx.z * 2 + (wave >> 1), r0 = (wave & 1) * 64;
    if (ks >= C / 32) return;
    const float* base = f + (size_t)b * N * C + ks * 32;
    f32x4 x[8];
    wave_rows32_to_lanes(slabs + wave * CASMTR_SLAB_FLOATS, lane,
                         [&](int rr) { const int gi = rb * 128 + r0 + rr; return base + (size_t)(gi < N ? gi : N - 1) * C; }, x);
    const int gi = rb * 128 + r0 + lane;
    const int e = ex[(size_t)b * NRB * 128 + gi];
    char* out = reinterpret_cast<char*>(img) + (((size_t)b * NRB + rb) * (size_t)(C / 32) + ks) * 16384 + (r0 + lane) * 16;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
        h16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xn = gi < N ? ldexpf(x[2 * kg + (c >> 2)][c & 3], -e) : 0.f;
            const _Float16 h = (_Float16)xn;
            hi[c] = h;
            lo[c] = (_Float16)(xn - (float)h);
        }
        *reinterpret_cast<h16x8*>(out + kg * 4096) = hi;
        *reinterpret_cast<h16x8*>(out + kg * 4096 + 2048) = lo;
    }
}

// (1) + (1b) + (2) in ONE launch for both operands (round 6): the three passes above read every feature row twice and cost five launches
// (0.109 ms per 8-pair call).  Workgroup = 64 rows of one operand, wave w = the k-stages w, w + 4 (32 channels each): the rows arrive
// once (coalesced, transposed to lane <-> row through the wave's slab, 8 float4 per stage in registers), the waves exchange their
// partial maxima / sums of squares through LDS, then every wave normalises, splits and writes its stages of the tile image.  The batch
// maximum of the norms is an atomicMax per workgroup (positive floats order like their bit patterns; namax / nbmax are zeroed with the
// workspace).  Same exponent rule, same roundings: the image and the factors are ds_rownorm_kernel + ds_split_kernel's, bit for bit.
template <int KSW>   // k-stages per wave: C / 128
__global__ __launch_bounds__(256) void ds_prep_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const uint8_t* __restrict__ mask0,
                                                      const uint8_t* __restrict__ mask1, DsWs w, int L, int S, int C, float inv_sqrtC, float k0a,
                                                      int NIB, int NJB) {
    __shared__ float slabs[4 * CASMTR_SLAB_FLOATS];
    __shared__ float pmx[4][64], pss[4][64];
    const int side = blockIdx.z, b = blockIdx.y;
    const int N = side ? S : L, NRB = side ? NJB : NIB;
    if ((int)blockIdx.x >= 2 * NRB) return;
    const float* f = side ? f1 : f0;
    const uint8_t* mask = side ? mask1 : mask0;
    const int rb = blockIdx.x >> 1, r0 = (blockIdx.x & 1) * 64, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KS = C >> 5;
    f32x4 x[KSW][8];
    float mx = 0.f, ss = 0.f;
#pragma unroll
    for (int q = 0; q < KSW; ++q) {
        const int ks = wave + 4 * q;
        if (ks < KS) {
            const float* base = f + (size_t)b * N * C + ks * 32;
            wave_rows32_to_lanes(slabs + wave * CASMTR_SLAB_FLOATS, lane,
                                 [&](int rr) { const int gi = rb * 128 + r0 + rr; return base + (size_t)(gi < N ? gi : N - 1) * C; }, x[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = x[q][i];
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
    }
    pmx[wave][lane] = mx; pss[wave][lane] = ss;
    __syncthreads();
    const int gi = rb * 128 + r0 + lane;
    const bool inr = gi < N;
    const float m = inr ? fmaxf(fmaxf(pmx[0][lane], pmx[1][lane]), fmaxf(pmx[2][lane], pmx[3][lane])) : 0.f;
    const int e = (m > 0.f && m < INFINITY) ? ilogbf(m) - 9 : 0;   // largest element -> [512, 1024)
    if (wave == 0) {
        const float s2 = (pss[0][lane] + pss[1][lane]) + (pss[2][lane] + pss[3][lane]);
        const float nr = inr ? sqrtf(s2) * inv_sqrtC * 1.001f : 0.f;   // |a| / sqrt(C), rounded up (covers the fp32 summation and the root)
        const size_t o = (size_t)b * NRB * 128 + gi;
        (side ? w.exB : w.exA)[o] = e;
        const float fv = inr ? ldexpf(side ? 1.0f : k0a, e) : 0.f;
        (side ? w.fb : w.fa)[o] = (mask && inr && mask[(size_t)b * N + gi] == 0) ? -fv : fv;
        (side ? w.nb : w.na)[o] = nr;
        const float wm = wave_max_f32(nr);
        if (lane == 0) atomicMax((side ? w.nbmax : w.namax) + b, __float_as_uint(wm));
    }
    _Float16* img = side ? w.imgB : w.imgA;
#pragma unroll
    for (int q = 0; q < KSW; ++q) {
        const int ks = wave + 4 * q;
        if (ks < KS) {
            char* out = reinterpret_cast<char*>(img) + (((size_t)b * NRB + rb) * (size_t)KS + ks) * 16384 + (r0 + lane) * 16;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                h16x8 hi, lo;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float xn = inr ? ldexpf(x[q][2 * kg + (c >> 2)][c & 3], -e) : 0.f;
                    const _Float16 h = (_Float16)xn;
                    hi[c] = h;
                    lo[c] = (_Float16)(xn - (float)h);
                }
                *reinterpret_cast<h16x8*>(out + kg * 4096) = hi;
                *reinterpret_cast<h16x8*>(out + kg * 4096 + 2048) = lo;
            }
        }
    }
}

int ds_split_launch(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, const DsWs& w, int B, int L, int S,
                    int C, float temperature, int recip, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const float sqrtC = (float)sqrt((double)C);
    const float k0 = (float)(1.0 / ((double)C * (double)temperature));
    (void)recip;   // the operand pre-scaling mode only matters to the exact chain (ds_fix_kernel)
    const char* ev = getenv("CASMTR_DS_PREP");   // "3": the three-pass form (tests compare the two)
    if (!(ev && ev[0] == '3') && C <= 256 && (C & 31) == 0) {
        const dim3 grid(2 * (NIB > NJB ? NIB : NJB), B, 2);
        if (C <= 128) hipLaunchKernelGGL(ds_prep_kernel<1>, grid, dim3(256), 0, s, feat0, feat1, mask0, mask1, w, L, S, C, 1.0f / sqrtC, k0, NIB, NJB);
        else hipLaunchKernelGGL(ds_prep_kernel<2>, grid, dim3(256), 0, s, feat0, feat1, mask0, mask1, w, L, S, C, 1.0f / sqrtC, k0, NIB, NJB);
        CASMTR_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(ds_rownorm_kernel, dim3(NIB * 32, B), dim3(256), 0, s, feat0, mask0, L, C, 1.0f / sqrtC, k0, w.exA, w.fa, w.na, NIB * 128);
    hipLaunchKernelGGL(ds_rownorm_kernel, dim3(NJB * 32, B), dim3(256), 0, s, feat1, mask1, S, C, 1.0f / sqrtC, 1.0f, w.exB, w.fb, w.nb, NJB * 128);
    hipLaunchKernelGGL(ds_nmax_kernel, dim3(B, 2), dim3(256), 0, s, w.na, w.nb, NIB * 128, NJB * 128, w.namax, w.nbmax);
    hipLaunchKernelGGL(ds_split_kernel, dim3(NIB, B, (C + 63) / 64), dim3(256), 0, s, feat0, L, C, w.exA, w.imgA, NIB);
    hipLaunchKernelGGL(ds_split_kernel, dim3(NJB, B, (C + 63) / 64), dim3(256), 0, s, feat1, S, C, w.exB, w.imgB, NJB);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== GEMM
__device__ __forceinline__ const char* uniform_ptr(const char* p) {   // pins a wave-uniform address to an SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// 4 KB = 4 lane-linear LDS-DMA instructions behind one M0 write: the immediate offset advances the LDS destination and the
// source address together (tools/probes/glds_offset.hip).  M0 is not restored: nothing else in this kernel uses it.
__device__ __forceinline__ void glds_4k(const char* base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072"
                 :: "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void glds_2k(const char* base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024"
                 :: "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// one k16 stage of a wave's 64 x 64 tile: small terms first; every accumulator is touched again only after three other MFMAs.  The GEMM
// and the flagged-segment recompute (ds_flagged_kernel) share it: same products in the same order = bit-identical logits.
// SWAP (the transposed problem: operand "a" holds rows of B's image): the two cross products in the other order, so that an accumulator sees
// lo_A hi_B, hi_A lo_B, hi_A hi_B again.
template <bool SWAP = false>
__device__ __forceinline__ void ds16_mfma12(f32x16 (&acc)[2][2], const h16x8 (&ah)[2], const h16x8 (&al)[2], const h16x8 (&bh)[2],
                                            const h16x8 (&bl)[2]) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
            acc[ti][tj] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
            acc[ti][tj] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bh[tj], acc[ti][tj], 0, 0, 0);
}

// Epilogue of an interior tile (all 128 rows and 128 columns in range), straight from the accumulator registers (32x32 MFMA
// C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  Same results as ds_tile_epilogue<true, true>
// (ds_common.hpp, which edge tiles still use) with the per-element overhead removed: one uniform tile pointer + a 32-bit lane
// offset + a scalar row offset per store instead of 64-bit address arithmetic, no bounds predication, padding masks from the
// SIGN of the staged factors (facA / facB negative = masked row / column) instead of byte loads, ds_read_b128 transposes.
//   scratch: 4 x [32][68] wave-private slabs, then rowx[2][128][2], colx[2][128][2]
#define DS16_WL 68
template <bool MASKED, bool STORE>   // STORE: the similarity matrix and the 16-row-group column maxima go to memory (dense pass 2 / tests)
__device__ __forceinline__ void ds_split_epilogue_wave(f32x16 (&acc)[2][2], float* wl, float* rowx, float* colx, const float* facA,
                                                       const float* facB, float* __restrict__ sim, const DsWs& w, int b, int tI, int tJ,
                                                       int L, int S, int NIB, int wr, int wc) {
    // steps 1-3 for the 64 x 64 part (wr, wc) of the 128 x 128 tile (tI, tJ): wl = this wave's [32][68] slab, rowx [2 wc][128][2],
    // colx [2 wr][128][2] the tile's exchange areas, facA / facB the tile's 128 row / column factors
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 5, ln = lane & 31;
    // addresses: wave-uniform base (SGPR pair) + 32-bit byte offset (lane part + scalar row part): no 64-bit VALU arithmetic
    char* tile = const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(sim + ((size_t)b * L + tI * DS_BM + wr * 64) * S + tJ * DS_BN + wc * 64)));
    const unsigned lane_off = (unsigned)(4 * hi * S + ln) * 4u;
    const unsigned row_bytes = (unsigned)S * 4u;
    float fbv[2];
    bool cmask[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const float f = facB[wc * 64 + tj * 32 + ln];
        fbv[tj] = __builtin_fabsf(f);
        cmask[tj] = MASKED && __float_as_int(f) < 0;
    }
    // ---- 1. scale, mask, store; 2. column statistics on the fly.  Stores: SGPR row base + constant lane offset (no VALU address
    //         arithmetic).  (A slab-at-a-time variant fits 128 VGPRs = 4 workgroups per CU, but measured no faster than this one
    //         at 3: 1.74 / 1.91 ms unmasked / masked against 1.74 / 1.79.)
    float xv[2][2][16];
    float cmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const f32x4 f4[4] = {*reinterpret_cast<const f32x4*>(facA + wr * 64 + ti * 32 + 4 * hi),
                             *reinterpret_cast<const f32x4*>(facA + wr * 64 + ti * 32 + 8 + 4 * hi),
                             *reinterpret_cast<const f32x4*>(facA + wr * 64 + ti * 32 + 16 + 4 * hi),
                             *reinterpret_cast<const f32x4*>(facA + wr * 64 + ti * 32 + 24 + 4 * hi)};   // rows (q, 0..3)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            float g16 = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = f4[r >> 2][r & 3];
                float x = __fmul_rn(__fmul_rn(acc[ti][tj][r], __builtin_fabsf(f)), fbv[tj]);
                if (MASKED && (__float_as_int(f) < 0 || cmask[tj])) x = NEG_FILL;
                const char* rowbase = tile + (size_t)(ti * 32 + (r & 3) + 8 * (r >> 2)) * row_bytes;    // wave-uniform
                if constexpr (STORE) {
                    if (tj == 0) asm volatile("global_store_dword %0, %1, %2" :: "v"(lane_off), "v"(x), "s"(rowbase) : "memory");
                    else asm volatile("global_store_dword %0, %1, %2 offset:128" :: "v"(lane_off), "v"(x), "s"(rowbase) : "memory");
                }
                xv[ti][tj][r] = x;
                g16 = fmaxf(g16, x);
            }
            cmax[tj] = fmaxf(cmax[tj], g16);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const float m = fmaxf(cmax[tj], __shfl_xor(cmax[tj], 32));
        float sm = 0.f;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += __expf(xv[ti][tj][r] - m);
        sm += __shfl_xor(sm, 32);
        if (hi == 0) {
            float* o = colx + (wr * 128 + wc * 64 + tj * 32 + ln) * 2;
            o[0] = m; o[1] = sm;
        }
    }
    // ---- 3. rows: 32-row slabs through the wave-private LDS region, lane <-> (row ln, column half hi)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) wl[((r & 3) + 8 * (r >> 2) + 4 * hi) * DS16_WL + tj * 32 + ln] = xv[ti][tj][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f32x4* rp = reinterpret_cast<const f32x4*>(wl + ln * DS16_WL + hi * 32);
        f32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = rp[c];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) m = fmaxf(fmaxf(m, fmaxf(v[c].x, v[c].y)), fmaxf(v[c].z, v[c].w));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) sm += (__expf(v[c].x - m) + __expf(v[c].y - m)) + (__expf(v[c].z - m) + __expf(v[c].w - m));
        sm += __shfl_xor(sm, 32);
        if (hi == 0) {
            float* o = rowx + (wc * 128 + wr * 64 + ti * 32 + ln) * 2;
            o[0] = m; o[1] = sm;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// step 4: combine the two parts that share a row (wc = 0,1) / a column (wr = 0,1); role < 128: row `role`, else column role - 128
__device__ __forceinline__ void ds_split_epilogue_combine(const float* rowx, const float* colx, const DsWs& w, int b, int tI, int tJ, int L,
                                                          int S, int NJB, int NIB, int role) {
    const float* x0 = (role < 128 ? rowx : colx) + (role & 127) * 2;
    const float* x1 = x0 + 128 * 2;
    const float ma = x0[0], mb = x1[0];
    const float mm = fmaxf(ma, mb);
    const float tot = x0[1] * __expf(ma - mm) + x1[1] * __expf(mb - mm);
    if (role < 128) {
        const size_t o = ((size_t)b * NJB + tJ) * L + tI * DS_BM + role;
        w.rp_m[o] = mm; w.rp_s[o] = tot;
    } else {
        const size_t o = ((size_t)b * NIB + tI) * S + tJ * DS_BN + role - 128;
        w.cp_m[o] = mm; w.cp_s[o] = tot;
    }
}

//   scratch: 4 x [32][68] wave-private slabs, then rowx[2][128][2], colx[2][128][2]
template <bool MASKED, bool STORE>
__device__ __forceinline__ void ds_split_epilogue(f32x16 (&acc)[2][2], float* scratch, const float* facA, const float* facB,
                                                  float* __restrict__ sim, const DsWs& w, int b, int tI, int tJ, int L, int S,
                                                  int NJB, int NIB) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* rowx = scratch + 4 * 32 * DS16_WL;           // [2 wc][128 rows][2]
    float* colx = rowx + 2 * 128 * 2;                   // [2 wr][128 cols][2]
    ds_split_epilogue_wave<MASKED, STORE>(acc, scratch + wave * (32 * DS16_WL), rowx, colx, facA, facB, sim, w, b, tI, tJ, L, S, NIB, wave >> 1, wave & 1);
    __syncthreads();
    ds_split_epilogue_combine(rowx, colx, w, b, tI, tJ, L, S, NJB, NIB, (int)threadIdx.x);
}

// 128 x 128 block tile, 4 waves x (64 x 64), k-stages of 16 (8 KB of A image + 8 KB of B image = half a 16 KB image chunk),
// double buffered: the DMA of stage ks+1 is in flight while stage ks feeds 12 MFMAs per wave.  40 KB of LDS, <= 170 VGPRs -> 3
// workgroups per CU: other workgroups' MFMA loops run under a workgroup's epilogue (VALU / LDS / stores).
#define DS16_STAGE 16384
#define DS16_LDS3 (3 * DS16_STAGE + 2 * 128 * 4)                          // three operand stages (prefetch distance 2) + factors; the
                                                                          // epilogue scratch (4*32*68 + 2*2*128*2 floats) aliases the stages
// (Round 5 also carried a two-stage ring, a four-stage form with register-prefetched fragments, a 256 x 128 block kernel with 128 x 64
//  wave tiles and a slab-store epilogue, all bit-identical and none faster -- DESIGN.md 14.5; pruned in round 6.)
__global__ __launch_bounds__(256, 3) void ds_gemm16_kernel(const _Float16* __restrict__ imgA, const _Float16* __restrict__ imgB,
                                                           const float* __restrict__ fa, const float* __restrict__ fb,
                                                           const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                           float* __restrict__ sim, DsWs w, int L, int S, int KS, int NJB, int NIB, int store) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 3 stages x (A | B) / epilogue scratch, then facA[128] | facB[128]
    char* lds = reinterpret_cast<char*>(smem);
    float* facA = smem + (DS16_LDS3 - 2 * 128 * 4) / 4;
    float* facB = facA + 128;
    const int NSJ = (NJB + 7) >> 3;
    const int t = xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int st = t >> 6, wi = t & 63;
    const int tI = (st / NSJ) * 8 + (wi >> 3), tJ = (st % NSJ) * 8 + (wi & 7);
    if (tI >= NIB || tJ >= NJB) return;   // padding of the super-tile grid (whole workgroup exits: no barrier is skipped)
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    // factors of the tile's rows / columns; negative = the row / column is masked (padding), see ds_split_epilogue
    bool masked = false;
    if (tid < 128) {
        const int gi = tI * DS_BM + tid;
        const float f = fa[((size_t)b * NIB + tI) * 128 + tid];   // sign = padding mask (ds_rownorm_kernel)
        masked = mask0 && gi < L && __float_as_int(f) < 0;   // the SIGN BIT: -0.f (a row whose factor underflowed) is still masked
        facA[tid] = f;
    } else {
        const int gj = tJ * DS_BN + tid - 128;
        const float f = fb[((size_t)b * NJB + tJ) * 128 + tid - 128];
        masked = mask0 && gj < S && __float_as_int(f) < 0;
        facB[tid - 128] = f;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_src = uniform_ptr(reinterpret_cast<const char*>(imgA) + ((size_t)b * NIB + tI) * (size_t)KS * 8192);
    const char* b_src = uniform_ptr(reinterpret_cast<const char*>(imgB) + ((size_t)b * NJB + tJ) * (size_t)KS * 8192);
    const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_byte_addr(lds) + (unsigned)(wave * 2048)));
    const int any_masked = __syncthreads_or(masked);   // (barrier: the factor loads above have completed before any DMA is outstanding)
    glds_2k(a_src, voff, lds0);
    glds_2k(b_src, voff, lds0 + 8192);
    if (KS > 1) {
        glds_2k(a_src + 8192, voff, lds0 + DS16_STAGE);
        glds_2k(b_src + 8192, voff, lds0 + DS16_STAGE + 8192);
    }
    const int hi = lane >> 5, ln = lane & 31;
    // fragment (ti, part): plane (kg = hi, part), row wr*64 + ti*32 + ln
    const char* fa_base = lds + (hi * 2) * 2048 + (wr * 64 + ln) * 16;
    const char* fb_base = lds + 8192 + (hi * 2) * 2048 + (wc * 64 + ln) * 16;
    if (store & 2) __builtin_amdgcn_s_setprio(2);   // waves in the k-loop ahead of waves in an epilogue (default; CASMTR_DS_PRIO=0 / 2: off / the other way round):
                                                    // 1.574-1.582 against 1.592-1.611 ms, alternating on one box (tools/ds_prio.py)
    int buf = 0;
    for (int ks = 0; ks < KS; ++ks) {
        // stage ks has landed when only stage ks + 1 (4 instructions) is still in flight.  No __syncthreads here: its
        // s_waitcnt vmcnt(0) would drain the stage that was just prefetched
        if (ks + 1 < KS) glds_wait<4>(); else glds_wait<0>();
        lds_reads_done();
        __builtin_amdgcn_s_barrier();   // everyone's share of stage ks has landed; everyone is done reading the buffer that is refilled next
        asm volatile("" ::: "memory");
        if (ks + 2 < KS) {
            const int nb = buf >= 1 ? buf - 1 : 2;   // (ks + 2) % 3
            glds_2k(a_src + (size_t)(ks + 2) * 8192, voff, lds0 + (unsigned)(nb * DS16_STAGE));
            glds_2k(b_src + (size_t)(ks + 2) * 8192, voff, lds0 + (unsigned)(nb * DS16_STAGE) + 8192);
        }
        h16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const char* pa = fa_base + buf * DS16_STAGE + ti * 512;
            const char* pb = fb_base + buf * DS16_STAGE + ti * 512;
            ah[ti] = *reinterpret_cast<const h16x8*>(pa);
            al[ti] = *reinterpret_cast<const h16x8*>(pa + 2048);
            bh[ti] = *reinterpret_cast<const h16x8*>(pb);
            bl[ti] = *reinterpret_cast<const h16x8*>(pb + 2048);
        }
        ds16_mfma12(acc, ah, al, bh, bl);
        buf = buf == 2 ? 0 : buf + 1;
    }
    __syncthreads();   // every wave is done with the operand buffers: they become the epilogue's scratch
    if (store & 2) __builtin_amdgcn_s_setprio(0);
    if (store & 4) __builtin_amdgcn_s_setprio(2);   // experiment (CASMTR_DS_PRIO=2): the other way round
    const int prio_bits = store & 6;
    store &= 1;
    (void)prio_bits;
    if (tI * DS_BM + DS_BM <= L && tJ * DS_BN + DS_BN <= S) {
        if (store) {
            if (any_masked) ds_split_epilogue<true, true>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
            else ds_split_epilogue<false, true>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
        } else if (any_masked) ds_split_epilogue<true, false>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
        else ds_split_epilogue<false, false>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
    } else {   // edge tile: the general epilogue (bounds predication, masks from memory)
        if (tid < 128) facA[tid] = __builtin_fabsf(facA[tid]);
        else facB[tid - 128] = __builtin_fabsf(facB[tid - 128]);
        __syncthreads();
        if (store) ds_tile_epilogue<true, true, true>(acc, smem, facA, facB, mask0, mask1, sim, w, b, tI, tJ, L, S, 0.f, 0.f, NJB, NIB);
        else ds_tile_epilogue<true, true, false>(acc, smem, facA, facB, mask0, mask1, sim, w, b, tI, tJ, L, S, 0.f, 0.f, NJB, NIB);
    }
}

int ds_gemm16_launch(const uint8_t* mask0, const uint8_t* mask1, float* sim, const DsWs& w, int B, int L, int S, int C, int store, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const int ntiles = ((NJB + 7) / 8) * ((NIB + 7) / 8) * 64;
    // Three operand stages (prefetch distance 2): 1.78 -> 1.77 ms unmasked, 1.85 -> 1.78 ms with padding masks (round 4).  A persistent
    // variant with the epilogue software-pipelined under the next tile's k-stages (two accumulator sets, 2 workgroups per CU, 256 VGPRs
    // with spills) was built and measured at 2.35 ms and removed; so were round 5's wide-tile / four-stage / slab-store variants.
    // store == 0 (round 6, the sparse pass 2's configuration): the epilogue keeps only the softmax partials and segment maxima -- the
    // 4 L S B bytes of the similarity matrix (3.74 GB at 832 x 832, B = 8), of which pass 2 read 2 %, are never written; the flagged
    // segments are recomputed from the operand images (ds_flagged_launch).
    const size_t lds = DS16_LDS3;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    prof_symbol_args(CASMTR_PROF_DS_GEMM, "%s", store ? " (similarity matrix stored)" : " (statistics only: no matrix store)");
    const char* pe = getenv("CASMTR_DS_PRIO");
    const int prio = pe ? (atoi(pe) & 3) * 2 : 2;
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_DS_GEMM, ds_gemm16_kernel, dim3(ntiles, B), dim3(256), lds, s, w.imgA, w.imgB, w.fa, w.fb,
                        mask0, mask1, sim, w, L, S, C / 16, NJB, NIB, (store ? 1 : 0) | prio);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== pass 2 without the matrix
// Round 6.  Pass 2 only ever needed the few entries that matter: those at or above their row's / column's near-tie
// threshold (index candidates) and those whose confidence can exceed `thr` (conf = p01 p10 > thr needs p01 > thr, i.e.
// x > rmax + log(thr rsum) = tau).  Rounds 3-5 had the GEMM write the whole [B, L, S] matrix (3.74 GB per 8-pair call at 832 x 832:
// the largest write of the step) and a segment-sparse pass read back the ~2 % of it whose (row, 128-column) / (column, 128-row)
// segment maximum -- left behind by the GEMM epilogue in rp_m / cp_m -- says something can matter.  Now the matrix is not written at
// all (ds_gemm16_launch(store = 0)); the flagged segments are RECOMPUTED from the operand images:
//   ds_flagscan_kernel   thread per row / per column walks its segment maxima and appends itself to the list of every flagged
//                        (pair, column block) / (pair, row block);
//   ds_flagtiles_kernel  lists -> work items of <= 128 flagged lines;
//   ds_flagged_kernel    one 128 x 128 tile per item: the listed rows gathered from the fp32 features and split in the kernel (the bits
//                        of ds_split_kernel's image) against the block's contiguous image of B, ds_gemm16_kernel's MFMA sequence and
//                        scaling -- the logits are
//                        bit-identical to what the GEMM saw (the segment maxima it left are maxima of exactly these values); then
//                        the per-entry logic of rounds 3-5's sparse pass from the accumulator registers: candidate lists, confidences of the entries
//                        above tau, packed best-of-row / best-of-column atomics, borderline list.  Column items are the transposed
//                        problem (listed columns gathered out of B's image against a row block of A; the two cross products issued in
//                        swapped order so that every accumulator again sees the GEMM's sequence) and only collect index candidates:
//                        every entry with conf > thr sits in a flagged ROW segment.
// ~1.5 % of the GEMM's tiles at 832 x 832 (a row's maximum sits in one or two of its 85 segments).  Degenerate inputs (all segments
// flagged: duplicated or zero rows) make the lists long, not wrong; they overflow the candidate lists and take the exact passes anyway.
#define DS_FL_ROWS 128
#define DS_FL_QCAP 4000   // entries of a tile that pass their line's threshold (8 bytes each, in the 32 KB of the dead operand chunk)
#define DS_FL_CHUNK 8
__global__ __launch_bounds__(256) void ds_flagscan_kernel(DsWs w, int B, int L, int S, int NJB, int NIB, float thr, int exact_rows_only) {
    // thread <-> (line, chunk of 8 segments): the chunk's 8 maxima are in flight together; a wave's lanes share (pair, segment), so the
    // compiler's wave-aggregated atomic hands out the list slots
    const int RCH = (NJB + DS_FL_CHUNK - 1) / DS_FL_CHUNK, CCH = (NIB + DS_FL_CHUNK - 1) / DS_FL_CHUNK;
    const int RB = (L + 255) / 256, CB = (S + 255) / 256;
    int blk = blockIdx.x;
    if (blk < B * RCH * RB) {
        const int b = blk / (RCH * RB), ch = (blk / RB) % RCH, i = (blk % RB) * 256 + threadIdx.x;
        if (i >= L) return;
        const size_t o = (size_t)b * L + i;
        // exact path (matching.hip: ds_eflagged_kernel): only entries that can exceed thr matter (x > tau, strictly, as in ds_conf_kernel)
        const float rm = w.rmax[o], rs = w.rsum[o], tau = rm + __logf(thr * rs) - 1e-2f;
        const float lim = exact_rows_only ? __uint_as_float(__float_as_uint(tau) + (tau >= 0.f ? 1u : -1u)) : fminf(w.rthr[o], tau);
        float mv[DS_FL_CHUNK];
#pragma unroll
        for (int k = 0; k < DS_FL_CHUNK; ++k) mv[k] = w.rp_m[((size_t)b * NJB + min(ch * DS_FL_CHUNK + k, NJB - 1)) * L + i];
#pragma unroll
        for (int k = 0; k < DS_FL_CHUNK; ++k) {
            const int J = ch * DS_FL_CHUNK + k;
            if (J < NJB && mv[k] >= lim && mv[k] != NEG_FILL) {
                const int slot = atomicAdd(w.fl_rn + b * NJB + J, 1);
                w.fl_r[((size_t)b * NJB + J) * L + slot] = i;
            }
        }
        return;
    }
    blk -= B * RCH * RB;
    if (!exact_rows_only && blk < B * CCH * CB) {
        const int b = blk / (CCH * CB), ch = (blk / CB) % CCH, j = (blk % CB) * 256 + threadIdx.x;
        if (j >= S) return;
        const float ct = w.cthr[(size_t)b * S + j];
        float mv[DS_FL_CHUNK];
#pragma unroll
        for (int k = 0; k < DS_FL_CHUNK; ++k) mv[k] = w.cp_m[((size_t)b * NIB + min(ch * DS_FL_CHUNK + k, NIB - 1)) * S + j];
#pragma unroll
        for (int k = 0; k < DS_FL_CHUNK; ++k) {
            const int I = ch * DS_FL_CHUNK + k;
            if (I < NIB && mv[k] >= ct && mv[k] != NEG_FILL) {
                const int slot = atomicAdd(w.fl_cn + b * NIB + I, 1);
                w.fl_c[((size_t)b * NIB + I) * S + slot] = j;
            }
        }
    }
}

// work items: (list id << 8 | tile within the list), row lists first then column lists (id >= B * NJB); fl_tn[0] = their number
__global__ __launch_bounds__(256) void ds_flagtiles_kernel(DsWs w, int nrl, int ncl) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nrl + ncl) return;
    const int n = t < nrl ? w.fl_rn[t] : w.fl_cn[t - nrl];
    const int nt = (n + DS_FL_ROWS - 1) / DS_FL_ROWS;
    if (!nt) return;
    const int base = atomicAdd(w.fl_tn, nt);
    for (int k = 0; k < nt; ++k) w.fl_t[base + k] = (t << 8) | k;
}

__global__ __launch_bounds__(256, 2) void ds_flagged_kernel(const float* __restrict__ f0, const float* __restrict__ f1, DsWs w, int B, int L, int S,
                                                            int C, int NJB, int NIB, float thr, float kthr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // one 32-channel chunk of the gathered lines | of the block, then the per-line tables
    char* As = reinterpret_cast<char*>(smem);      // [kg 4][hi | lo][128 lines][8 f16]: two k16 stages of ds_gemm16_kernel's layout
    char* Bs = As + 16384;
    float* tab = smem + 2 * 16384 / 4;             // facA[128] | facB[128] | lim | rt | tau | rm | rinv | line[128] (int)
    float *facA = tab, *facB = tab + 128, *t_lim = tab + 256, *t_rt = tab + 384, *t_tau = tab + 512, *t_rm = tab + 640;
    float* t_rinv = tab + 768;
    int* line = reinterpret_cast<int*>(tab + 896);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 1, wc = wave & 1;
    const int hi = lane >> 5, ln = lane & 31;
    const int nitems = w.fl_tn[0], nrl = B * NJB;
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int e = w.fl_t[it], id = e >> 8, tile = e & 255;
        const bool cols = id >= nrl;                                   // wave-uniform (workgroup-uniform)
        // rows: lines = rows of A (imgA, L of them, thresholds rt / tau), block = column block blk of B;  cols: lines = columns (imgB)
        const int lid = cols ? id - nrl : id;
        const int NB_blk = cols ? NIB : NJB;                           // blocks per pair on the BLOCK side
        const int b = lid / NB_blk, blk = lid - b * NB_blk;
        const int NL = cols ? S : L;                                   // lines per pair, blocks per pair on the LINE side
        const int NB_line = cols ? NJB : NIB;
        const int n = (cols ? w.fl_cn : w.fl_rn)[lid] - tile * DS_FL_ROWS;
        const int* list = (cols ? w.fl_c : w.fl_r) + (size_t)lid * NL + tile * DS_FL_ROWS;
        const _Float16* img_blk = cols ? w.imgA : w.imgB;
        __syncthreads();   // the previous item's tables are no longer read
        if (tid < 128) {
            const int li = list[tid < n ? tid : 0];                    // short tiles repeat their first line (results discarded)
            line[tid] = li;
            const float f = (cols ? w.fb : w.fa)[(size_t)b * NB_line * 128 + li];
            facA[tid] = f;
            const size_t o = (size_t)b * NL + li;
            if (cols) t_lim[tid] = tid < n ? w.cthr[o] : INFINITY;
            else {
                const float rm = w.rmax[o], rs = w.rsum[o], rt = w.rthr[o], tau = rm + __logf(thr * rs) - 1e-2f;
                t_rt[tid] = rt; t_tau[tid] = tau; t_rm[tid] = rm; t_rinv[tid] = 1.0f / rs;
                t_lim[tid] = tid < n ? fminf(rt, tau) : INFINITY;
            }
        } else facB[tid - 128] = (cols ? w.fa : w.fb)[((size_t)b * NB_blk + blk) * 128 + tid - 128];
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // The listed lines come from the fp32 FEATURES (one contiguous C-float row each: a 32-channel chunk is one 128-byte line per row) and are
        // normalised and split on the way into LDS exactly as ds_split_kernel made the image (same exponent, ldexpf, f16 roundings: the same
        // bits).  Gathering them out of the tile image instead -- 64 16-byte pieces per row, each in a line shared with seven other rows --
        // moved 8x the bytes and cost 311 us per call (profiles/r06c_kernel_stats.csv).  The block side is its contiguous image.
        const int KS32 = C >> 5;
        const int lrow = tid >> 1, half = tid & 1;
        const int gl = line[lrow];
        const int ex = (cols ? w.exB : w.exA)[(size_t)b * NB_line * 128 + gl];
        const float* ap = (cols ? f1 : f0) + ((size_t)b * NL + gl) * C + half * 16;
        const char* bp = reinterpret_cast<const char*>(img_blk) + ((size_t)b * NB_blk + blk) * (size_t)KS32 * 16384 + tid * 16;
        char* const a_dst = As + (half * 2) * 4096 + lrow * 16;
        // the gathered rows come from anywhere in the pair's features (HBM / Infinity Cache latency): two chunks ahead; the block's image is
        // L2-resident after its first few items: one chunk ahead
        f32x4 av[4], avn[4];
        u32x4 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
            avn[i] = KS32 > 1 ? *reinterpret_cast<const f32x4*>(ap + 32 + 4 * i) : av[i];
            bv[i] = *reinterpret_cast<const u32x4*>(bp + 4096 * i);
        }
        const char* fa_base = As + hi * 4096 + (wr * 64 + ln) * 16;   // k16 sub-stage s: + s * 8192; tile ti: + ti * 512; lo part: + 2048
        const char* fb_base = Bs + hi * 4096 + (wc * 64 + ln) * 16;
        auto chunk = [&](int ks, f32x4 (&cur)[4]) {   // cur = chunk ks of the gathered rows; refilled with chunk ks + 2 once it is in LDS
            __syncthreads();   // previous chunk fully consumed
#pragma unroll
            for (int kgl = 0; kgl < 2; ++kgl) {
                h16x8 vh, vl;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float xn = ldexpf(cur[2 * kgl + (c >> 2)][c & 3], -ex);
                    const _Float16 h = (_Float16)xn;
                    vh[c] = h;
                    vl[c] = (_Float16)(xn - (float)h);
                }
                *reinterpret_cast<h16x8*>(a_dst + kgl * 4096) = vh;
                *reinterpret_cast<h16x8*>(a_dst + kgl * 4096 + 2048) = vl;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + 4096 * i + tid * 16) = bv[i];
            if (ks + 2 < KS32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(ap + (ks + 2) * 32 + 4 * i);
            }
            if (ks + 1 < KS32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const u32x4*>(bp + (size_t)(ks + 1) * 16384 + 4096 * i);
            }
            __syncthreads();
#pragma unroll
            for (int st = 0; st < 2; ++st) {   // the two k16 stages of the chunk, in the GEMM's order
                h16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const char* pa = fa_base + st * 8192 + ti * 512;
                    const char* pb = fb_base + st * 8192 + ti * 512;
                    ah[ti] = *reinterpret_cast<const h16x8*>(pa);
                    al[ti] = *reinterpret_cast<const h16x8*>(pa + 2048);
                    bh[ti] = *reinterpret_cast<const h16x8*>(pb);
                    bl[ti] = *reinterpret_cast<const h16x8*>(pb + 2048);
                }
                if (cols) ds16_mfma12<true>(acc, ah, al, bh, bl); else ds16_mfma12<false>(acc, ah, al, bh, bl);
            }
        };
        for (int ks = 0; ks < KS32; ks += 2) {
            chunk(ks, av);
            if (ks + 1 < KS32) chunk(ks + 1, avn);
        }
        // ---- the entries, from the accumulators (32x32 C/D layout: column ln, rows (r & 3) + 8 (r >> 2) + 4 hi of each block).  Almost every
        //      entry fails the line's threshold; the few that pass are queued in LDS (over the operand chunk, which is dead now) and handled
        //      by one thread each afterwards -- the candidate / confidence / borderline code exists once, not once per unrolled entry (the
        //      first version inlined it 64 times: 229 VGPRs, two workgroups per CU).
        __syncthreads();   // every wave is done with the last chunk
        int* qn = reinterpret_cast<int*>(As);                  // [0] = queued entries
        int2* queue = reinterpret_cast<int2*>(As + 16);        // (lr << 8 | lc, x bits)
        if (tid == 0) qn[0] = 0;
        __syncthreads();
        // (opaque copy of the lane id: otherwise the 64 packed (row, column) codes and table addresses of the entries are loop-invariant
        //  and get hoisted out of the item loop into 100+ VGPRs that stay live through the k-loop -- 345 spills at three workgroups per CU)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int hi = lane_o >> 5, ln = lane_o & 31;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int lc = wc * 64 + tj * 32 + ln;                     // position in the block
            const float fbs = facB[lc];
            const bool c_ok = blk * 128 + lc < (cols ? L : S) && __float_as_int(fbs) >= 0;   // in range and not padding
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                asm volatile("" ::: "memory");   // one 32 x 32 block at a time: without it all 64 table reads are hoisted (128 more live registers)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float fas = facA[lr];
                    // the GEMM's scaling order: (acc * |factor of A's row|) * |factor of B's column|
                    const float x = cols ? __fmul_rn(__fmul_rn(acc[ti][tj][r], __builtin_fabsf(fbs)), __builtin_fabsf(fas))
                                         : __fmul_rn(__fmul_rn(acc[ti][tj][r], __builtin_fabsf(fas)), __builtin_fabsf(fbs));
                    if (c_ok && x >= t_lim[lr] && __float_as_int(fas) >= 0) {
                        const int slot = atomicAdd(qn, 1);
                        if (slot < DS_FL_QCAP) queue[slot] = make_int2((lr << 8) | lc, __float_as_int(x));
                    }
                }
            }
        }
        __syncthreads();
        const int nq = qn[0];
        if (nq > DS_FL_QCAP && tid == 0) *w.ovf = 1;   // a tile full of (near-)equal entries: the exact passes decide the call
        const float keep = 1.0f - 2.0f * ds_conf_band(kthr, w.namax[b], w.nbmax[b]), cmin = 0.9f * thr;
        for (int q = tid; q < min(nq, DS_FL_QCAP); q += 256) {
            const int2 ent = queue[q];
            const int lr = ent.x >> 8, lc = ent.x & 255, gc = blk * 128 + lc, li = line[lr];
            const float x = __int_as_float(ent.y);
            if (cols) {
                const size_t co = (size_t)b * S + li;
                const int slot = atomicAdd(w.ccnt + co, 1);
                if (slot < DS_CAND_CAP) w.ccand[co * DS_CAND_CAP + slot] = gc; else *w.ovf = 1;
                continue;
            }
            const size_t o = (size_t)b * L + li;
            const int j = gc;
            if (x >= t_rt[lr]) {
                const int slot = atomicAdd(w.rcnt + o, 1);
                if (slot < DS_CAND_CAP) w.rcand[o * DS_CAND_CAP + slot] = j; else *w.ovf = 1;
            }
            if (x > t_tau[lr]) {
                const size_t co = (size_t)b * S + j;
                const float cf = (__expf(x - w.cmax[co]) * (1.0f / w.csum[co])) * (__expf(x - t_rm[lr]) * t_rinv[lr]);
                if (cf >= 0.f) {
                    const unsigned long long hk = (unsigned long long)__float_as_uint(cf) << 32;
                    const unsigned long long oldr = atomicMax(w.rbest + o, hk | (0xFFFFFFFFu - (unsigned)j));
                    const unsigned long long oldc = atomicMax(w.cbest + co, hk | (0xFFFFFFFFu - (unsigned)li));
                    // borderline entries (ds_xdecide_launch): the previous maximum and this entry within the error band of each other
                    if (cf > cmin) {
                        const float cr = __uint_as_float((unsigned)(oldr >> 32)), cc = __uint_as_float((unsigned)(oldc >> 32));
                        if (fminf(cf, cr) > cmin && fminf(cf, cr) >= fmaxf(cf, cr) * keep) {
                            ds_x_append(w, (int)o, j);
                            ds_x_append(w, (int)o, (int)(0xFFFFFFFFu - (unsigned)(oldr & 0xFFFFFFFFu)));
                        }
                        if (fminf(cf, cc) > cmin && fminf(cf, cc) >= fmaxf(cf, cc) * keep) {
                            ds_x_append(w, (int)o, j);
                            ds_x_append(w, b * L + (int)(0xFFFFFFFFu - (unsigned)(oldc & 0xFFFFFFFFu)), j);
                        }
                    }
                }
            }
        }
    }
}

int ds_flag_lists_launch(const DsWs& w, int B, int L, int S, float thr, int exact_rows_only, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const int nscan = B * ((NJB + DS_FL_CHUNK - 1) / DS_FL_CHUNK) * ((L + 255) / 256) + B * ((NIB + DS_FL_CHUNK - 1) / DS_FL_CHUNK) * ((S + 255) / 256);
    hipLaunchKernelGGL(ds_flagscan_kernel, dim3(nscan), dim3(256), 0, s, w, B, L, S, NJB, NIB, thr, exact_rows_only);
    hipLaunchKernelGGL(ds_flagtiles_kernel, dim3((B * (NJB + NIB) + 255) / 256), dim3(256), 0, s, w, B * NJB, B * NIB);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

int ds_flagged_launch(const float* feat0, const float* feat1, const DsWs& w, int B, int L, int S, int C, float thr, float kthr, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    if (const int r = ds_flag_lists_launch(w, B, L, S, thr, 0, s)) return r;
    constexpr size_t lds = 2 * 16384 + 1024 * 4;
    static int resident_tab[CASMTR_MAX_DEVICES] = {0};
    int resident = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_flagged_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (const int r = resident_workgroups(resident_tab, ds_flagged_kernel, 256, lds, &resident)) return r;
    hipLaunchKernelGGL(ds_flagged_kernel, dim3((unsigned)resident), dim3(256), lds, s, feat0, feat1, w, B, L, S, C, NJB, NIB, thr, kthr);
    CASMTR_CHECK_LAUNCH();
    if (getenv("CASMTR_DS_DEBUG")) {   // diagnostic only: synchronises
        int nt = 0;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&nt, w.fl_tn, sizeof nt, hipMemcpyDeviceToHost);
        int* cn = (int*)calloc((size_t)B * (NJB + NIB), sizeof(int));
        if (cn) {
            (void)hipMemcpy(cn, w.fl_rn, sizeof(int) * (size_t)B * NJB, hipMemcpyDeviceToHost);
            (void)hipMemcpy(cn + B * NJB, w.fl_cn, sizeof(int) * (size_t)B * NIB, hipMemcpyDeviceToHost);
            long long nr = 0, nc = 0;
            for (int i = 0; i < B * NJB; ++i) nr += cn[i];
            for (int i = 0; i < B * NIB; ++i) nc += cn[B * NJB + i];
            fprintf(stderr, "ds_flagged: %lld flagged row segments of %lld (%.2f %%), %lld column segments of %lld, %d work items of <= %d lines on %d workgroups\n",
                    nr, (long long)B * NJB * L, 100.0 * nr / ((double)B * NJB * L), nc, (long long)B * NIB * S, nt, DS_FL_ROWS, resident);
            free(cn);
        }
    }
    return 0;
}

// =================================================================================================== exact re-decision
// Thread per row (t < B*L) / per column: 0 candidates = fully masked -> index 0 (the first of the equal maxima); 1 -> that one;
// more -> the oracle's logit for each (fmaf chain over c ascending of the 1/sqrt(C)-scaled operands, then / T -- what
// v_mfma_f32_32x32x2_f32 computes in the exact kernel) and the first maximum.  Candidates are never masked entries.
template <bool RECIP>
__global__ __launch_bounds__(256) void ds_fix_kernel(const float* __restrict__ f0, const float* __restrict__ f1, DsWs w, int B, int L,
                                                     int S, int C, float sqrtC, float inv_sqrtC, float T, float invT,
                                                     int64_t* __restrict__ next_idx01, int64_t* __restrict__ next_idx10) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (L + S)) return;
    const bool col = t >= B * L;
    const int u = col ? t - B * L : t;
    const int N = col ? S : L;
    const int b = u / N, self = u % N;
    const int n = min((col ? w.ccnt : w.rcnt)[u], DS_CAND_CAP);
    const int* cand = (col ? w.ccand : w.rcand) + (size_t)u * DS_CAND_CAP;
    int best = 0;
    if (n == 1) best = cand[0];
    else if (n > 1) {
        float bx = -INFINITY;
        best = cand[0];
        for (int k = 0; k < n; ++k) {
            const int other = cand[k];
            const int i = col ? other : self, j = col ? self : other;
            const float* pa = f0 + ((size_t)b * L + i) * C;
            const float* pb = f1 + ((size_t)b * S + j) * C;
            float acc = 0.f;
            for (int c = 0; c < C; c += 4) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(pa + c), vb = *reinterpret_cast<const f32x4*>(pb + c);
                acc = __builtin_fmaf(div_scalar<RECIP>(va.x, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.x, sqrtC, inv_sqrtC), acc);
                acc = __builtin_fmaf(div_scalar<RECIP>(va.y, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.y, sqrtC, inv_sqrtC), acc);
                acc = __builtin_fmaf(div_scalar<RECIP>(va.z, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.z, sqrtC, inv_sqrtC), acc);
                acc = __builtin_fmaf(div_scalar<RECIP>(va.w, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.w, sqrtC, inv_sqrtC), acc);
            }
            const float x = div_scalar<RECIP>(acc, T, invT);
            if (x > bx || (x == bx && other < best)) { bx = x; best = other; }
        }
    }
    (col ? next_idx10 : next_idx01)[u] = best;
}

int ds_fix_launch(const float* feat0, const float* feat1, const DsWs& w, int B, int L, int S, int C, float temperature, int recip,
                  int64_t* next_idx01, int64_t* next_idx10, hipStream_t s) {
    const float sqrtC = (float)sqrt((double)C);
    const int total = B * (L + S);
    if (recip)
        hipLaunchKernelGGL(ds_fix_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, s, feat0, feat1, w, B, L, S, C, sqrtC,
                           1.0f / sqrtC, temperature, 1.0f / temperature, next_idx01, next_idx10);
    else
        hipLaunchKernelGGL(ds_fix_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, s, feat0, feat1, w, B, L, S, C, sqrtC,
                           1.0f / sqrtC, temperature, 1.0f / temperature, next_idx01, next_idx10);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== exact match list
// "Match list exact by construction" (VERDICT r04 item 3).  coarse_matching.py:116-132 decides a match from three float comparisons
// on conf = softmax_row * softmax_col: conf > thr, conf == its row's maximum, conf == its column's maximum.  The split path's
// confidences carry a relative error of up to `band` (ds_conf_band), so any of those comparisons can come out differently from the
// exact path's when the two sides are closer than that.  Pass 2 and ds_xnear_kernel put every entry for which that can happen on a
// list (typically a few dozen per batch: confidences within 0.5 % of thr or of a runner-up); for those entries
//   * the logit is recomputed with the oracle's fmaf chain (as ds_fix_kernel does for the argmax candidates),
//   * the softmax statistics of their row AND their column are recomputed from exact logits, in exactly the order the exact
//     kernels use (ds_tile_epilogue<RECIP, false>: 32-entry lane runs, lane pairs, wave pairs; ds_reduce_kernel: 128-wide blocks
//     ascending), so that max, sum and hence conf are BIT-IDENTICAL to casmtr_dual_softmax_fwd's,
//   * and the rows concerned take their decision (best column, conf > thr, mutual maximum) from those values.
// Entries not on the list are further than 2 * band from every decision boundary they take part in: the approximate comparison and
// the exact one agree.  Net effect: the (b, i, j) list equals the exact path's on every input; mconf of a re-decided row is the exact
// value, the other mconf values stay within the split's 1e-6.  Lists that overflow raise w.ovf -> the exact passes decide.

__device__ __forceinline__ void wave_lds_fence_() {   // this wave's LDS writes are visible to its own later reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (1) thread per row: the row's approximate best sits within the band of thr -> borderline
__global__ __launch_bounds__(256) void ds_xnear_kernel(DsWs w, int L, int total, float thr, float kthr) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / L;
    const float band = ds_conf_band(kthr, w.namax[b], w.nbmax[b]);
    // The borderline lists only take entries above cmin = 0.9 thr (pass 2), i.e. they assume 2 band < 0.1.  Large-norm features
    // (|a||b| / (C T) in the hundreds: band grows with namax nbmax) break that assumption: a runner-up below 0.9 thr could exactly
    // exceed thr or its line's best and would never be listed.  Such a pair hands every decision to the exact passes instead.
    if (band > DS_BAND_MAX) { *w.ovf = 1; return; }
    const unsigned long long key = w.rbest[t];
    if (!key) return;
    const float cf = __uint_as_float((unsigned)(key >> 32));
    if (fabsf(cf - thr) <= band * fmaxf(cf, thr)) ds_x_append(w, t, (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu)));
}

// (2) thread per listed entry: its row and its column need exact statistics (each line is claimed once, lists per pair)
__global__ __launch_bounds__(256) void ds_xclaim_kernel(DsWs w, int B, int L, int S) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(w.xcnt[0], DS_X_CAP);
    if (t >= n) return;
    const int ro = w.xent[2 * t], j = w.xent[2 * t + 1];
    const int b = ro / L, co = b * S + j;
    // byte flags inside 4-byte words: claim with an atomic OR on the containing word
    auto claim = [](unsigned char* flags, int idx) {
        unsigned* wp = reinterpret_cast<unsigned*>(flags) + (idx >> 2);
        const unsigned bit = 1u << ((idx & 3) * 8);
        return (atomicOr(wp, bit) & bit) == 0;
    };
    if (claim(w.rneed, ro)) {
        const int slot = atomicAdd(w.xln + b, 1);
        if (slot < DS_XL_PCAP) w.rlist[(size_t)b * DS_XL_PCAP + slot] = ro - b * L; else *w.ovf = 1;
    }
    if (claim(w.cneed, co)) {
        const int slot = atomicAdd(w.xln + B + b, 1);
        if (slot < DS_XL_PCAP) w.clist[(size_t)b * DS_XL_PCAP + slot] = j; else *w.ovf = 1;
    }
}

// (3) exact (max, sum exp) partials of the listed lines over one 128-wide block, in ds_tile_epilogue<RECIP, false>'s order.
// The logits are a small exact GEMM -- up to 32 listed lines x the block's 128 entries x C channels -- and run where the exact path
// runs its GEMM: v_mfma_f32_32x32x2_f32, whose k-sequence IS the oracle's c-ascending fmaf chain (products commute, so it does not
// matter on which side the line is).  Workgroup = (pair, side, group of up to 32 listed lines, four consecutive blocks); the lines'
// own rows sit once in LDS, pre-scaled (pitch 257 floats: operand A = a[line lane % 32][2 m + lane / 32] is a conflict-free
// ds_read2_b32 per two MFMAs); every wave takes one block: its 128 rows of the OTHER side pass 32 rows x 32 channels at a time through
// the wave's transposition slab (coalesced 128-byte rows, two steps in flight), lane (j, hi) picks channels 2 m + hi of row j as
// operand B.  32 steps x 16 MFMAs per unit = 16 us on one SIMD; ~1360 units per call.
// (History, all measured in the bench step with ~350 listed lines: a lane-per-row walk per line 1.4 ms (texture-address bound); a wave
// per line 0.7 ms (each pass re-reads the other side's 11 MB at a 1 KB stride); VALU chains, 8 or 32 lines per pass, 0.15-0.24 ms:
// one broadcast ds_read_b128 in front of every four FMAs, accumulators that spilled, v_pk_fma_f32 that hipcc packs two chains into.)
// Then four lanes per line replay what one LANE of the tile kernel reduces, from the logits parked in LDS, 8 lines at a time:
//   rows:    wave wc = half of the tile covers columns 64 wc .. 64 wc + 63; its lane (hi, row) scans columns 32 hi + c, c ascending;
//   columns: wave wr = half covers rows 64 wr .. 64 wr + 63; its lane (hi, col) scans rows 32 ti + (r & 3) + 8 (r >> 2) + 4 hi in
//            (ti, r) order --
// the lane pair (hi = 0, 1) shares its maximum and adds its sums, and the two halves combine as the tile kernel's step 4 does.
#define DS_XL_GB 32    // lines per workgroup unit = rows of the MFMA tile
#define DS_XL_SLP 257  // pitch of an own row in LDS (floats)
template <bool RECIP>
__global__ __launch_bounds__(256, 2) void ds_xstats_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                        const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, DsWs w,
                                                        int B, int L, int S, int C, float sqrtC, float inv_sqrtC, float T, float invT,
                                                        int NJB, int NIB) {
    constexpr int GB = DS_XL_GB, SLP = DS_XL_SLP, SLABF = 32 * 36;   // slab: 32 rows x 32 channels, rows padded to 36 floats
    static_assert(8 * 128 <= SLABF, "the logits of 8 lines alias the slab");
    extern __shared__ __attribute__((aligned(16))) float xs_smem[];
    float* sl = xs_smem;                                   // [GB][SLP] the lines' own rows, scaled by 1 / sqrt(C) (C <= 256); shared by the 4 waves
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert((GB * SLP) % 4 == 0, "slab alignment");
    float* slab = xs_smem + GB * SLP + wave * SLABF;       // wave-private transposition slab; later the logits of 8 lines
    const int hi = lane >> 5, jl = lane & 31;
    long long g = blockIdx.x;
    long long base = 0;
    // list lengths: one load per 64 lists (lane <-> list): the counters were written by device-scope atomics, every first read misses
    int cnts = 0;
    for (int sb = 0; sb < 2 * B; ++sb) {
        if ((sb & 63) == 0) cnts = sb + lane < 2 * B ? min(w.xln[sb + lane], DS_XL_PCAP) : 0;
        const bool col = sb >= B;
        const int b = col ? sb - B : sb;
        const int cnt = __shfl(cnts, sb & 63), nblk = col ? NIB : NJB, nq = (nblk + 3) >> 2;
        const long long nu = (long long)((cnt + GB - 1) / GB) * nq;
        for (; g < base + nu; g += gridDim.x) {
            const int grp = (int)((g - base) / nq), t = (int)((g - base) % nq) * 4 + wave;
            const int ng = min(GB, cnt - grp * GB);        // lines of this group
            const bool live = t < nblk;                    // this wave has a block
            const int* list = (col ? w.clist : w.rlist) + (size_t)b * DS_XL_PCAP + grp * GB;
            const int N = col ? S : L, M = col ? L : S;    // own side / other side
            const float* pown = (col ? f1 : f0) + (size_t)b * N * C;
            const float* pother = (col ? f0 : f1) + (size_t)b * M * C;
            const uint8_t* mown = mask0 ? (col ? mask1 : mask0) + (size_t)b * N : nullptr;
            const uint8_t* mother = mask0 ? (col ? mask0 : mask1) + (size_t)b * M : nullptr;
            // the group's line indices: ONE load (lane k <- entry k); wave v stages lines 8 v .. 8 v + 7 by independent loads
            const int mine = list[lane < ng ? lane : 0];
            {
                f32x4 rv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int self = __shfl(mine, 8 * wave + k);
                    rv[k] = lane * 4 < C ? *reinterpret_cast<const f32x4*>(pown + (size_t)self * C + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float* d = sl + (8 * wave + k) * SLP + lane * 4;
                    d[0] = div_scalar<RECIP>(rv[k].x, sqrtC, inv_sqrtC); d[1] = div_scalar<RECIP>(rv[k].y, sqrtC, inv_sqrtC);
                    d[2] = div_scalar<RECIP>(rv[k].z, sqrtC, inv_sqrtC); d[3] = div_scalar<RECIP>(rv[k].w, sqrtC, inv_sqrtC);
                }
            }
            const bool selfmask = mown && mown[mine] == 0;   // lane k: line k of the group is a padded row / column
            __syncthreads();
            if (live) {
                f32x16 acc[4];                               // sub-tile sub: D[line][32 sub + j]
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[sub][r] = 0.f;
                // step = 4 ks + sub: rows 32 sub .. 32 sub + 31 of the block, channels 32 ks .. 32 ks + 31 (4 loads per lane)
                auto load_rows = [&](int ks, int sub, f32x4 (&v)[4]) {
                    const int o0 = t * 128 + sub * 32;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        v[jj] = *reinterpret_cast<const f32x4*>(pother + (size_t)min(o0 + 8 * jj + (lane >> 3), M - 1) * C + ks * 32 + (lane & 7) * 4);
                };
                f32x4 v0[4], v1[4];                          // steps s + 1 and s + 2 in flight while step s runs
                load_rows(0, 0, v0);
                load_rows(0, 1, v1);
                const float* arow = sl + jl * SLP + hi;      // operand A: a[line lane % 32][2 m + lane / 32]
                const int nks = C / 32;
                for (int ks = 0; ks < nks; ++ks)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    f32x4 (&v)[4] = (sub & 1) ? v1 : v0;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(slab + (8 * jj + (lane >> 3)) * 36 + (lane & 7) * 4) = v[jj];
                    wave_lds_fence_();
                    f32x4 x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const f32x4*>(slab + jl * 36 + i * 4);
                    wave_lds_fence_();
                    {   // refill this ring slot with step s + 2
                        const int s2 = 4 * ks + sub + 2;
                        if (s2 < 4 * nks) load_rows(s2 >> 2, s2 & 3, v);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {            // channels 4 q .. 4 q + 3 of the chunk: MFMAs m = 2 q, 2 q + 1 (k = 2 m + hi)
                        const float a0 = arow[ks * 32 + 4 * q], a1 = arow[ks * 32 + 4 * q + 2];                 // one ds_read2_b32
                        const float b0 = div_scalar<RECIP>(hi ? x[q].y : x[q].x, sqrtC, inv_sqrtC);
                        const float b1 = div_scalar<RECIP>(hi ? x[q].w : x[q].z, sqrtC, inv_sqrtC);
                        acc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[sub], 0, 0, 0);
                        acc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[sub], 0, 0, 0);
                    }
                    asm volatile("" : "+v"(acc[sub]) :: "memory");   // the step's MFMAs stay in front of the next step's slab writes
                }
                // ---- the tile kernel's reduction, 8 lines at a time.  D layout: lane (j, hi), register r -> line (r & 3) + 8 (r >> 2) + 4 hi:
                // the lines of group gq are registers 4 gq .. 4 gq + 3 -> line-in-group rr + 4 hi; logits -> slab as [8][128]
#pragma unroll
                for (int gq = 0; gq < GB / 8; ++gq) {
                    if (8 * gq < ng) {
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub) {
                            const int o = t * 128 + sub * 32 + jl;
                            const bool om = mother && o < M && mother[o] == 0;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const int kl = rr + 4 * hi;     // line within the group
                                float vv = -INFINITY;           // outside the matrix: never wins a max, adds exp(-inf) = 0
                                const bool sm_ = __shfl((int)selfmask, 8 * gq + kl) != 0;
                                if (o < M) vv = (om || sm_) ? NEG_FILL : div_scalar<RECIP>(acc[sub][4 * gq + rr], T, invT);
                                slab[kl * 128 + sub * 32 + jl] = vv;
                            }
                        }
                        wave_lds_fence_();
                        const int k = lane >> 2, q = lane & 3, half = q >> 1, hh = q & 1;
                        const float* xk = slab + (k < 8 ? k : 0) * 128;
                        float xe[32];
#pragma unroll
                        for (int e = 0; e < 32; ++e) {
                            const int local = col ? ((e >> 4) * 32 + (e & 3) + 8 * ((e & 15) >> 2) + 4 * hh) : (32 * hh + e);
                            xe[e] = xk[half * 64 + local];
                        }
                        float m = -INFINITY;
#pragma unroll
                        for (int e = 0; e < 32; ++e) m = xe[e] > m ? xe[e] : m;
                        const float pm = __shfl_xor(m, 1);
                        m = pm > m ? pm : m;
                        float sm = 0.f;
#pragma unroll
                        for (int e = 0; e < 32; ++e) sm += __expf(xe[e] - m);
                        sm += __shfl_xor(sm, 1);
                        // step 4 of the tile epilogue: halves 0 and 1
                        const float mo = __shfl_xor(m, 2), so = __shfl_xor(sm, 2);
                        const float ma = half ? mo : m, mb = half ? m : mo, sa = half ? so : sm, sb2 = half ? sm : so;
                        const float mm = mb > ma ? mb : ma;
                        float tot = 0.f;
                        if (ma > -INFINITY) tot += sa * __expf(ma - mm);
                        if (mb > -INFINITY) tot += sb2 * __expf(mb - mm);
                        const int selfk = __shfl(mine, 8 * gq + (k < 8 ? k : 0));
                        if (q == 0 && k < 8 && 8 * gq + k < ng) {
                            const size_t oo = ((size_t)b * nblk + t) * N + selfk;
                            (col ? w.cp_m : w.rp_m)[oo] = mm;
                            (col ? w.cp_s : w.rp_s)[oo] = tot;
                        }
                        wave_lds_fence_();
                    }
                }
            }
            __syncthreads();   // sl is rewritten by the next unit
        }
        base += nu;
    }
}

// (4) wave per listed line: block partials -> (max, sum) exactly as ds_reduce_kernel (sum over the blocks in ascending order; the
// terms are formed by the lanes in parallel and added up in order by lane 0); next_conf = 1 / sum follows
__global__ __launch_bounds__(256) void ds_xreduce_kernel(DsWs w, int B, int L, int S, int NJB, int NIB, float* __restrict__ next_conf01,
                                                         float* __restrict__ next_conf10) {
    __shared__ float terms[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the per-pair list lengths: one load (lane <-> list), then only the listed lines are visited
    for (int sb0 = 0; sb0 < 2 * B; sb0 += 64) {
    const int mycnt = sb0 + lane < 2 * B ? min(w.xln[sb0 + lane], DS_XL_PCAP) : 0;
    for (int s2 = 0; s2 < min(64, 2 * B - sb0); ++s2) {
    const int sb = sb0 + s2, cnt = __shfl(mycnt, s2);
    const int nW = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    for (int k = (wg + nW - (sb * 64) % nW) % nW; k < cnt; k += nW) {   // list sb starts at wave 64 sb: the lists' lines run side by side
        const bool col = sb >= B;
        const int b = col ? sb - B : sb;
        const int self = (col ? w.clist : w.rlist)[(size_t)b * DS_XL_PCAP + k];
        const int N = col ? S : L, nblk = col ? NIB : NJB;
        const int line = b * N + self;
        const float* pm = (col ? w.cp_m : w.rp_m) + (size_t)b * nblk * N + self;
        const float* ps = (col ? w.cp_s : w.rp_s) + (size_t)b * nblk * N + self;
        float pmv[4], psv[4], m = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = lane + 64 * u;
            pmv[u] = kk < nblk ? pm[(size_t)kk * N] : -INFINITY;
            psv[u] = kk < nblk ? ps[(size_t)kk * N] : 0.f;
            m = fmaxf(m, pmv[u]);
        }
        m = wave_max_f32(m);
#pragma unroll
        for (int u = 0; u < 4; ++u) terms[wave][lane + 64 * u] = psv[u] * __expf(pmv[u] - m);
        wave_lds_fence_();
        if (lane == 0) {
            float s = 0.f;
            for (int kk = 0; kk < nblk; ++kk) s += terms[wave][kk];
            (col ? w.cmax : w.rmax)[line] = m;
            (col ? w.csum : w.rsum)[line] = s;
            (col ? next_conf10 : next_conf01)[line] = 1.0f / s;
        }
        wave_lds_fence_();
    }
    }
    }
}

// the oracle's logit of (row i of feat0, row j of feat1): fmaf chain over c ascending of the 1/sqrt(C)-scaled operands, then / T
template <bool RECIP>
__device__ __forceinline__ float ds_exact_logit(const float* __restrict__ pa, const float* __restrict__ pb, int C, float sqrtC,
                                                float inv_sqrtC, float T, float invT) {
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(pa + c), vb = *reinterpret_cast<const f32x4*>(pb + c);
        acc = __builtin_fmaf(div_scalar<RECIP>(va.x, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.x, sqrtC, inv_sqrtC), acc);
        acc = __builtin_fmaf(div_scalar<RECIP>(va.y, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.y, sqrtC, inv_sqrtC), acc);
        acc = __builtin_fmaf(div_scalar<RECIP>(va.z, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.z, sqrtC, inv_sqrtC), acc);
        acc = __builtin_fmaf(div_scalar<RECIP>(va.w, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.w, sqrtC, inv_sqrtC), acc);
    }
    return div_scalar<RECIP>(acc, T, invT);
}

// (5) wave per listed entry: exact confidence, the expression of ds_conf_kernel<false>.  The two feature rows come in by one coalesced
// load each and sit in LDS, pre-scaled; the chain itself is sequential (every lane runs it on broadcast reads).  (A thread per entry
// walking both rows 16 bytes at a time was 24 us of dependent memory latency.)
template <bool RECIP>
__global__ __launch_bounds__(256) void ds_xconf_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                       const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, DsWs w,
                                                       int L, int S, int C, float sqrtC, float inv_sqrtC, float T, float invT) {
    __shared__ __attribute__((aligned(16))) float rows[4][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(w.xcnt[0], DS_X_CAP);
    for (int t = blockIdx.x * 4 + wave; t < n; t += gridDim.x * 4) {
        const int ro = w.xent[2 * t], j = w.xent[2 * t + 1];
        const int b = ro / L;
        const size_t co = (size_t)b * S + j;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 va = lane * 4 < C ? *reinterpret_cast<const f32x4*>(f0 + (size_t)ro * C + lane * 4) : z;
        const f32x4 vb = lane * 4 < C ? *reinterpret_cast<const f32x4*>(f1 + co * C + lane * 4) : z;
        const float rm = w.rmax[ro], rs = w.rsum[ro], cm = w.cmax[co], cs = w.csum[co];
        const bool masked = mask0 && (mask0[ro] == 0 || mask1[co] == 0);
        *reinterpret_cast<f32x4*>(&rows[wave][0][lane * 4]) = (f32x4){div_scalar<RECIP>(va.x, sqrtC, inv_sqrtC), div_scalar<RECIP>(va.y, sqrtC, inv_sqrtC),
                                                                      div_scalar<RECIP>(va.z, sqrtC, inv_sqrtC), div_scalar<RECIP>(va.w, sqrtC, inv_sqrtC)};
        *reinterpret_cast<f32x4*>(&rows[wave][1][lane * 4]) = (f32x4){div_scalar<RECIP>(vb.x, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.y, sqrtC, inv_sqrtC),
                                                                      div_scalar<RECIP>(vb.z, sqrtC, inv_sqrtC), div_scalar<RECIP>(vb.w, sqrtC, inv_sqrtC)};
        wave_lds_fence_();
        float acc = 0.f;
        for (int c = 0; c < C; c += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&rows[wave][0][c]), b4 = *reinterpret_cast<const f32x4*>(&rows[wave][1][c]);
            acc = __builtin_fmaf(a4.x, b4.x, acc);
            acc = __builtin_fmaf(a4.y, b4.y, acc);
            acc = __builtin_fmaf(a4.z, b4.z, acc);
            acc = __builtin_fmaf(a4.w, b4.w, acc);
        }
        const float x = masked ? NEG_FILL : div_scalar<RECIP>(acc, T, invT);
        const float rinv = 1.0f / rs, cinv = 1.0f / cs;
        const float p01 = __expf(x - rm) * rinv;
        const float p10 = __expf(x - cm) * cinv;
        if (lane == 0) w.xcf[t] = p10 * p01;
        wave_lds_fence_();
    }
}

// (6) wave per listed entry e = (i, j): is e the exact best of row i (first column among equal values) and does it satisfy
// conf > thr and conf == column maximum?  Entries of row i / column j that are NOT listed lie below the band of the line's
// approximate best; so if that best itself is not listed, nothing listed can be the line's maximum.  The lanes share the walk over
// the list (a thread per entry walking the whole list was 37 us of serial LDS round trips on one CU).
__global__ __launch_bounds__(256) void ds_xdecide_kernel(DsWs w, int L, int S, float thr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(w.xcnt[0], DS_X_CAP);
    for (int t = blockIdx.x * 4 + wave; t < n; t += gridDim.x * 4) {
        const int ro = w.xent[2 * t], j = w.xent[2 * t + 1];
        const int b = ro / L;
        const size_t co = (size_t)b * S + j;
        const float cf = w.xcf[t];
        const int jbest = (int)(0xFFFFFFFFu - (unsigned)(w.rbest[ro] & 0xFFFFFFFFu));
        const int ibest = (int)(0xFFFFFFFFu - (unsigned)(w.cbest[co] & 0xFFFFFFFFu));
        bool row_listed = false, col_listed = false;   // the row's / the column's approximate best is on the list
        bool row_lose = false, col_lose = false;
        for (int k = lane; k < n; k += 64) {
            const int ro2 = w.xent[2 * k], j2 = w.xent[2 * k + 1];
            const float c2 = w.xcf[k];
            if (ro2 == ro) {
                row_listed |= j2 == jbest;
                row_lose |= c2 > cf || (c2 == cf && j2 < j);        // rbest semantics: maximal value, first column
            }
            if (j2 == j && ro2 >= b * L && ro2 < (b + 1) * L) {
                col_listed |= ro2 - b * L == ibest;
                col_lose |= c2 > cf;                                 // mutual maximum BY VALUE (coarse_matching.py:120-122)
            }
        }
        const bool rl = __ballot(row_listed) != 0ull, cl = __ballot(col_listed) != 0ull;
        const bool rlose = __ballot(row_lose) != 0ull, clo = __ballot(col_lose) != 0ull;
        if (!rl || rlose) continue;               // not this row's exact best (or the row is decided by an unlisted, clearly larger entry)
        if (lane == 0) {
            const bool ok = cf > thr && cl && !clo;
            w.rdec_j[ro] = j;                      // duplicates of e write the same values
            w.rdec_cf[ro] = cf;
            w.rdec[ro] = ok ? 3 : 1;
        }
    }
}

int ds_xdecide_launch(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, const DsWs& w, int B, int L,
                      int S, int C, float temperature, int recip, float thr, float* next_conf01, float* next_conf10, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const float sqrtC = (float)sqrt((double)C), kthr = 6.103515625e-05f / temperature;
    hipLaunchKernelGGL(ds_xnear_kernel, dim3((B * L + 255) / 256), dim3(256), 0, s, w, L, B * L, thr, kthr);
    hipLaunchKernelGGL(ds_xclaim_kernel, dim3(DS_X_CAP / 256), dim3(256), 0, s, w, B, L, S);
    CASMTR_CHECK_LAUNCH();
    // (pair, side, group of 32 listed lines, four blocks) units, one workgroup each, grid-strided: one round for the usual ~350 units
    const size_t xs_lds = sizeof(float) * (DS_XL_GB * DS_XL_SLP + 4 * 32 * 36);
    if (recip) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_xstats_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_lds);
        hipLaunchKernelGGL(ds_xstats_kernel<true>, dim3(512), dim3(256), xs_lds, s, feat0, feat1, mask0, mask1, w, B, L, S, C, sqrtC, 1.0f / sqrtC,
                           temperature, 1.0f / temperature, NJB, NIB);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_xstats_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_lds);
        hipLaunchKernelGGL(ds_xstats_kernel<false>, dim3(512), dim3(256), xs_lds, s, feat0, feat1, mask0, mask1, w, B, L, S, C, sqrtC, 1.0f / sqrtC,
                           temperature, 1.0f / temperature, NJB, NIB);
    }
    hipLaunchKernelGGL(ds_xreduce_kernel, dim3(256), dim3(256), 0, s, w, B, L, S, NJB, NIB, next_conf01, next_conf10);
    CASMTR_CHECK_LAUNCH();
    if (recip)
        hipLaunchKernelGGL(ds_xconf_kernel<true>, dim3(256), dim3(256), 0, s, feat0, feat1, mask0, mask1, w, L, S, C, sqrtC,
                           1.0f / sqrtC, temperature, 1.0f / temperature);
    else
        hipLaunchKernelGGL(ds_xconf_kernel<false>, dim3(256), dim3(256), 0, s, feat0, feat1, mask0, mask1, w, L, S, C, sqrtC,
                           1.0f / sqrtC, temperature, 1.0f / temperature);
    hipLaunchKernelGGL(ds_xdecide_kernel, dim3(256), dim3(256), 0, s, w, L, S, thr);
    CASMTR_CHECK_LAUNCH();
    if (getenv("CASMTR_DS_DEBUG")) {   // diagnostic only: synchronises
        int cnt[4] = {0, 0, 0, 0};
        int* ln = (int*)calloc((size_t)2 * B, sizeof(int));   // xln: [B] listed rows, then [B] listed columns
        if (!ln) return 0;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(cnt, w.xcnt, sizeof cnt, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ln, w.xln, sizeof(int) * (size_t)(2 * B), hipMemcpyDeviceToHost);
        int nr = 0, ncol = 0;
        for (int i = 0; i < B; ++i) { nr += ln[i]; ncol += ln[B + i]; }
        free(ln);
        fprintf(stderr, "ds_xdecide: %d borderline entries, %d rows + %d columns recomputed exactly (B = %d, L = %d, S = %d)\n", cnt[0], nr,
                ncol, B, L, S);
    }
    return 0;
}

}  // namespace casmtr
