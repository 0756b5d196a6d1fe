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
// the row's candidates (ds_conf_kernel<true> collects them while it streams the matrix anyway); a row with one candidate has
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

int ds_split_launch(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, const DsWs& w, int B, int L, int S,
                    int C, float temperature, int recip, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const float sqrtC = (float)sqrt((double)C);
    const float k0 = (float)(1.0 / ((double)C * (double)temperature));
    (void)recip;   // the operand pre-scaling mode only matters to the exact chain (ds_fix_kernel)
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

// Epilogue of an interior tile (all 128 rows and 128 columns in range), straight from the accumulator registers (32x32 MFMA
// C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  Same results as ds_tile_epilogue<true, true>
// (ds_common.hpp, which edge tiles still use) with the per-element overhead removed: one uniform tile pointer + a 32-bit lane
// offset + a scalar row offset per store instead of 64-bit address arithmetic, no bounds predication, padding masks from the
// SIGN of the staged factors (facA / facB negative = masked row / column) instead of byte loads, ds_read_b128 transposes.
//   scratch: 4 x [32][68] wave-private slabs, then rowx[2][128][2], colx[2][128][2]
#define DS16_WL 68
template <bool MASKED, bool WST = false>
__device__ __forceinline__ void ds_split_epilogue_wave(f32x16 (&acc)[2][2], float* wl, float* rowx, float* colx, const float* facA,
                                                       const float* facB, float* __restrict__ sim, const DsWs& w, int b, int tI, int tJ,
                                                       int L, int S, int NIB, int wr, int wc) {
    // steps 1-3 for the 64 x 64 part (wr, wc) of the 128 x 128 tile (tI, tJ): wl = this wave's [32][68] slab, rowx [2 wc][128][2],
    // colx [2 wr][128][2] the tile's exchange areas, facA / facB the tile's 128 row / column factors
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 5, ln = lane & 31;
    // addresses: wave-uniform base (SGPR pair) + 32-bit byte offset (lane part + scalar row part): no 64-bit VALU arithmetic
    char* tile = const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(sim + ((size_t)b * L + tI * DS_BM + wr * 64) * S + tJ * DS_BN + wc * 64)));
    char* grp = const_cast<char*>(uniform_ptr(reinterpret_cast<const char*>(w.cg_m + (((size_t)b * NIB + tI) * 8 + wr * 4) * S + tJ * DS_BN + wc * 64)));
    const unsigned lane_off = (unsigned)(4 * hi * S + ln) * 4u, grp_off = (unsigned)(hi * S + ln) * 4u;
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
                if constexpr (!WST) {
                    if (tj == 0) asm volatile("global_store_dword %0, %1, %2" :: "v"(lane_off), "v"(x), "s"(rowbase) : "memory");
                    else asm volatile("global_store_dword %0, %1, %2 offset:128" :: "v"(lane_off), "v"(x), "s"(rowbase) : "memory");
                }
                xv[ti][tj][r] = x;
                g16 = fmaxf(g16, x);
            }
            // maximum of this lane's 16-row group (wr, ti, hi): the sparse pass 2 reads 16 rows of a column, not 128
            const char* gb = grp + (size_t)(ti * 2) * row_bytes;
            if (tj == 0) asm volatile("global_store_dword %0, %1, %2" :: "v"(grp_off), "v"(g16), "s"(gb) : "memory");
            else asm volatile("global_store_dword %0, %1, %2 offset:128" :: "v"(grp_off), "v"(g16), "s"(gb) : "memory");
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
        if constexpr (WST) {   // the slab's rows as 256-byte runs: 8 stores of 1 KB instead of 32 of 256 B
            const float* sp = wl + (lane >> 4) * DS16_WL + (lane & 15) * 4;
            const unsigned so = (unsigned)((lane >> 4) * S + (lane & 15) * 4) * 4u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(sp + 4 * k * DS16_WL);
                const char* rowbase = tile + (size_t)(ti * 32 + 4 * k) * row_bytes;    // wave-uniform
                asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(so), "v"(q4), "s"(rowbase) : "memory");
            }
        }
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
template <bool MASKED, bool WST = false>
__device__ __forceinline__ void ds_split_epilogue(f32x16 (&acc)[2][2], float* scratch, const float* facA, const float* facB,
                                                  float* __restrict__ sim, const DsWs& w, int b, int tI, int tJ, int L, int S,
                                                  int NJB, int NIB) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* rowx = scratch + 4 * 32 * DS16_WL;           // [2 wc][128 rows][2]
    float* colx = rowx + 2 * 128 * 2;                   // [2 wr][128 cols][2]
    ds_split_epilogue_wave<MASKED, WST>(acc, scratch + wave * (32 * DS16_WL), rowx, colx, facA, facB, sim, w, b, tI, tJ, L, S, NIB, wave >> 1, wave & 1);
    __syncthreads();
    ds_split_epilogue_combine(rowx, colx, w, b, tI, tJ, L, S, NJB, NIB, (int)threadIdx.x);
}

// 128 x 128 block tile, 4 waves x (64 x 64), k-stages of 16 (8 KB of A image + 8 KB of B image = half a 16 KB image chunk),
// double buffered: the DMA of stage ks+1 is in flight while stage ks feeds 12 MFMAs per wave.  40 KB of LDS, <= 170 VGPRs -> 3
// workgroups per CU: other workgroups' MFMA loops run under a workgroup's epilogue (VALU / LDS / stores).
#define DS16_STAGE 16384
#define DS16_LDS (4 * 32 * 65 * 4 + 2 * 2 * 128 * 3 * 4 + 2 * 128 * 4)   // epilogue scratch (aliases the stages; the interior-tile
                                                                          // layout 4*32*68 + 2*2*128*2 floats is 512 B smaller) + factors
#define DS16_LDS3 (3 * DS16_STAGE + 2 * 128 * 4)                          // three operand stages (prefetch distance 2) + factors
template <int NSTG>   // operand stages in LDS: 2 (prefetch distance 1, 40 KB) or 3 (distance 2, 49 KB; still 3 workgroups per CU);
                      // 4 = three stages + the NEXT stage's fragments read into registers under the current stage's MFMAs
__global__ __launch_bounds__(256, 3) void ds_gemm16_kernel(const _Float16* __restrict__ imgA, const _Float16* __restrict__ imgB,
                                                           const float* __restrict__ fa, const float* __restrict__ fb,
                                                           const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                           float* __restrict__ sim, DsWs w, int L, int S, int KS, int NJB, int NIB, int skipI, int skipJ, int wst) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages x (A | B) / epilogue scratch, then facA[128] | facB[128]
    char* lds = reinterpret_cast<char*>(smem);
    constexpr bool PF = NSTG == 4;
    float* facA = smem + ((NSTG >= 3 ? DS16_LDS3 : DS16_LDS) - 2 * 128 * 4) / 4;
    float* facB = facA + 128;
    const int NSJ = (NJB + 7) >> 3;
    const int t = xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int st = t >> 6, wi = t & 63;
    const int tI = (st / NSJ) * 8 + (wi >> 3), tJ = (st % NSJ) * 8 + (wi & 7);
    if (tI >= NIB || tJ >= NJB) return;   // padding of the super-tile grid (whole workgroup exits: no barrier is skipped)
    if (tI < skipI && tJ < skipJ) return;   // the interior belongs to ds_gemm16w_kernel
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
    if (NSTG >= 3 && KS > 1) {
        glds_2k(a_src + 8192, voff, lds0 + DS16_STAGE);
        glds_2k(b_src + 8192, voff, lds0 + DS16_STAGE + 8192);
    }
    if (PF && KS > 2) {
        glds_2k(a_src + 2 * 8192, voff, lds0 + 2 * DS16_STAGE);
        glds_2k(b_src + 2 * 8192, voff, lds0 + 2 * DS16_STAGE + 8192);
    }
    const int hi = lane >> 5, ln = lane & 31;
    // fragment (ti, part): plane (kg = hi, part), row wr*64 + ti*32 + ln
    const char* fa_base = lds + (hi * 2) * 2048 + (wr * 64 + ln) * 16;
    const char* fb_base = lds + 8192 + (hi * 2) * 2048 + (wc * 64 + ln) * 16;
    if constexpr (PF) {
        // Stage ks is consumed from registers while stage ks + 1 is read from LDS, stage ks + 2 is landing and stage ks + 3 is issued
        // into the buffer stage ks was read from (everyone has read it: those reads were completed before this iteration's barrier).
        struct Frag { h16x8 ah[2], al[2], bh[2], bl[2]; };
        auto read = [&](Frag& f, int bufi) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const char* pa = fa_base + bufi * DS16_STAGE + ti * 512;
                const char* pb = fb_base + bufi * DS16_STAGE + ti * 512;
                f.ah[ti] = *reinterpret_cast<const h16x8*>(pa);
                f.al[ti] = *reinterpret_cast<const h16x8*>(pa + 2048);
                f.bh[ti] = *reinterpret_cast<const h16x8*>(pb);
                f.bl[ti] = *reinterpret_cast<const h16x8*>(pb + 2048);
            }
        };
        auto mfmas = [&](const Frag& f) {   // small terms first; every accumulator is touched again only after three other MFMAs
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ti], f.bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
        };
        int bcur = 0;   // buffer of stage ks
        auto step = [&](int ks, const Frag& cur, Frag& nxt) {
            const int bn = bcur == 2 ? 0 : bcur + 1;   // buffer of stage ks + 1
            if (ks + 1 < KS) {
                if (ks + 2 < KS) glds_wait<4>(); else glds_wait<0>();   // stage ks + 1 has landed when only stage ks + 2 is in flight
                lds_reads_done();
                __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave has read stage ks out of buffer bcur
                asm volatile("" ::: "memory");
                if (ks + 3 < KS) {
                    glds_2k(a_src + (size_t)(ks + 3) * 8192, voff, lds0 + (unsigned)(bcur * DS16_STAGE));
                    glds_2k(b_src + (size_t)(ks + 3) * 8192, voff, lds0 + (unsigned)(bcur * DS16_STAGE) + 8192);
                }
                read(nxt, bn);
            }
            mfmas(cur);
            bcur = bn;
        };
        Frag f0, f1;
        if (KS > 2) glds_wait<8>(); else if (KS > 1) glds_wait<4>(); else glds_wait<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read(f0, 0);
        for (int ks = 0; ks < KS; ks += 2) {
            step(ks, f0, f1);
            if (ks + 1 < KS) step(ks + 1, f1, f0);
        }
    } else {
    int buf = 0;
    for (int ks = 0; ks < KS; ++ks) {
        if (NSTG == 3) {
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
        } else {
            glds_wait<0>();
            __syncthreads();   // stage ks has landed for every wave; everyone is done reading the other buffer
            if (ks + 1 < KS) {
                glds_2k(a_src + (size_t)(ks + 1) * 8192, voff, lds0 + (unsigned)((buf ^ 1) * DS16_STAGE));
                glds_2k(b_src + (size_t)(ks + 1) * 8192, voff, lds0 + (unsigned)((buf ^ 1) * DS16_STAGE) + 8192);
            }
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
        // small terms first; every accumulator is touched again only after three other MFMAs
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bh[tj], acc[ti][tj], 0, 0, 0);
        buf = NSTG == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
    }
    }
    __syncthreads();   // every wave is done with the operand buffers: they become the epilogue's scratch
    if (tI * DS_BM + DS_BM <= L && tJ * DS_BN + DS_BN <= S) {
        if (wst) {
            if (any_masked) ds_split_epilogue<true, true>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
            else ds_split_epilogue<false, true>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
        } else if (any_masked) ds_split_epilogue<true>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
        else ds_split_epilogue<false>(acc, smem, facA, facB, sim, w, b, tI, tJ, L, S, NJB, NIB);
    } else {   // edge tile: the general epilogue (bounds predication, masks from memory)
        if (tid < 128) facA[tid] = __builtin_fabsf(facA[tid]);
        else facB[tid - 128] = __builtin_fabsf(facB[tid - 128]);
        __syncthreads();
        ds_tile_epilogue<true, true>(acc, smem, facA, facB, mask0, mask1, sim, w, b, tI, tJ, L, S, 0.f, 0.f, NJB, NIB);
    }
}


// The same GEMM with 128 x 64 WAVE tiles: block tile 256 x 128 = two vertically adjacent 128 x 128 tiles of the image layout, wave
// (wrr, wc) owns the 128 rows of tile 2 tI2 + wrr and 64 columns.  Per k-stage a wave reads 12 KB of operands for 24 MFMAs instead of
// 8 KB for 12: the 64 x 64 kernel's main loop runs at the LDS's 128 B/clk, this one has a third of that to
// spare.  Stage = 8 KB of each A tile + 8 KB of B = 24 KB, three stages (prefetch distance 2), 128 accumulator registers: two
// workgroups per CU.  Every accumulator sees the same MFMA sequence as in ds_gemm16_kernel and the epilogue is that kernel's, run
// per 64 x 64 part (each wave does its parts (0, wc) and (1, wc) of its tile one after the other): bit-identical results.  Only
// blocks whose 256 rows and 128 columns are all in range; ds_gemm16_kernel does the bottom and right strips.
#define DS16W_STAGE 24576
#define DS16W_LDS (3 * DS16W_STAGE + (256 + 128) * 4)
__global__ __launch_bounds__(256, 2) void ds_gemm16w_kernel(const _Float16* __restrict__ imgA, const _Float16* __restrict__ imgB,
                                                            const float* __restrict__ fa, const float* __restrict__ fb, int have_mask,
                                                            float* __restrict__ sim, DsWs w, int L, int S, int KS, int NJB, int NIB,
                                                            int NIB2, int NJBf, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 3 stages x (A0 | A1 | B) / epilogue scratch, then facA[256] | facB[128]
    char* lds = reinterpret_cast<char*>(smem);
    float* facA = smem + 3 * DS16W_STAGE / 4;
    float* facB = facA + 256;
    const int NSJ = (NJBf + 7) >> 3;
    const int t = xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int st = t >> 6, wi = t & 63;
    const int tI2 = (st / NSJ) * 8 + (wi >> 3), tJ = (st % NSJ) * 8 + (wi & 7);
    if (tI2 >= NIB2 || tJ >= NJBf) return;   // padding of the super-tile grid (whole workgroup exits: no barrier is skipped)
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wrr = wave >> 1, wc = wave & 1;
    bool masked = false;
    {
        const float f = fa[((size_t)b * NIB + 2 * tI2) * 128 + tid];   // the two tiles' factors are adjacent; sign = padding mask
        masked = have_mask && __float_as_int(f) < 0;
        facA[tid] = f;
        if (tid < 128) {
            const float g = fb[((size_t)b * NJB + tJ) * 128 + tid];
            masked = masked || (have_mask && __float_as_int(g) < 0);
            facB[tid] = g;
        }
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a0_src = uniform_ptr(reinterpret_cast<const char*>(imgA) + ((size_t)b * NIB + 2 * tI2) * (size_t)KS * 8192);
    const char* a1_src = a0_src + (size_t)KS * 8192;
    const char* b_src = uniform_ptr(reinterpret_cast<const char*>(imgB) + ((size_t)b * NJB + tJ) * (size_t)KS * 8192);
    const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_byte_addr(lds) + (unsigned)(wave * 2048)));
    const int any_masked = __syncthreads_or(masked);   // (barrier: the factor loads above have completed before any DMA is outstanding)
    auto issue = [&](int ks, int nb) {   // 6 DMA instructions per wave
        const unsigned d = lds0 + (unsigned)(nb * DS16W_STAGE);
        glds_2k(a0_src + (size_t)ks * 8192, voff, d);
        glds_2k(a1_src + (size_t)ks * 8192, voff, d + 8192);
        glds_2k(b_src + (size_t)ks * 8192, voff, d + 16384);
    };
    issue(0, 0);
    if (KS > 1) issue(1, 1);
    const int hi = lane >> 5, ln = lane & 31;
    // fragment (ti, part): plane (kg = hi, part), row ti*32 + ln of this wave's A tile / column wc*64 + tj*32 + ln
    const char* fa_base = lds + wrr * 8192 + (hi * 2) * 2048 + ln * 16;
    const char* fb_base = lds + 16384 + (hi * 2) * 2048 + (wc * 64 + ln) * 16;
    int buf = 0;
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) glds_wait<6>(); else glds_wait<0>();   // stage ks has landed when only stage ks + 1 is still in flight
        lds_reads_done();
        __builtin_amdgcn_s_barrier();   // everyone's share of stage ks has landed; everyone is done reading the buffer that is refilled next
        asm volatile("" ::: "memory");
        if (ks + 2 < KS) issue(ks + 2, buf >= 1 ? buf - 1 : 2);
        h16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const char* pa = fa_base + buf * DS16W_STAGE + ti * 512;
            ah[ti] = *reinterpret_cast<const h16x8*>(pa);
            al[ti] = *reinterpret_cast<const h16x8*>(pa + 2048);
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const char* pb = fb_base + buf * DS16W_STAGE + tj * 512;
            bh[tj] = *reinterpret_cast<const h16x8*>(pb);
            bl[tj] = *reinterpret_cast<const h16x8*>(pb + 2048);
        }
        if (dbg & 2) {   // timing experiment: no MFMAs
            acc[0][0][0] += (float)ah[0][0] + (float)al[1][1] + (float)ah[2][2] + (float)al[3][3] + (float)bh[0][0] + (float)bl[1][1];
            buf = buf == 2 ? 0 : buf + 1;
            continue;
        }
        // small terms first (the order of ds_gemm16_kernel for every accumulator)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bh[tj], acc[ti][tj], 0, 0, 0);
        buf = buf == 2 ? 0 : buf + 1;
    }
    __syncthreads();   // every wave is done with the operand buffers: they become the epilogue's scratch
    if (dbg & 1) {   // timing experiment: no epilogue
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) x += acc[i][j][r];
        if (x == 12345.f) sim[tid] = x;
        return;
    }
    float* wl = smem + wave * (32 * DS16_WL);
    float* xch = smem + 4 * 32 * DS16_WL;            // per tile: rowx [2 wc][128][2], colx [2 wr][128][2]
    float* rowx = xch + wrr * 1024, *colx = rowx + 512;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        f32x16 (&part)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2 * v]);
        if (any_masked) ds_split_epilogue_wave<true>(part, wl, rowx, colx, facA + wrr * 128, facB, sim, w, b, 2 * tI2 + wrr, tJ, L, S, NIB, v, wc);
        else ds_split_epilogue_wave<false>(part, wl, rowx, colx, facA + wrr * 128, facB, sim, w, b, 2 * tI2 + wrr, tJ, L, S, NIB, v, wc);
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) ds_split_epilogue_combine(xch + sub * 1024, xch + sub * 1024 + 512, w, b, 2 * tI2 + sub, tJ, L, S, NJB, NIB, tid);
}

int ds_gemm16_launch(const uint8_t* mask0, const uint8_t* mask1, float* sim, const DsWs& w, int B, int L, int S, int C, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const int ntiles = ((NJB + 7) / 8) * ((NIB + 7) / 8) * 64;
    // Default: three operand stages (prefetch distance 2): 1.78 -> 1.77 ms unmasked, 1.85 -> 1.78 ms with padding masks (round 4).  The
    // kernel is bound by LDS bandwidth in its main loop (DESIGN.md sections 11, 12), not by the DMA latency, so this is all a deeper
    // ring buys.  A persistent variant with the epilogue software-pipelined under the next tile's k-stages (two accumulator sets, 2
    // workgroups per CU, 256 VGPRs with spills) was built and measured at 2.35 ms: the epilogue's slab traffic lands on the same
    // saturated LDS; it was removed again.
    // Interior by ds_gemm16w_kernel (128 x 64 wave tiles) where at least one 256 x 128 block is whole, the bottom / right strips by
    // ds_gemm16_kernel (CASMTR_DS_GEMM16_WIDE=0: everything by the latter; CASMTR_DS_GEMM16_STAGES=2: its two-stage form)
    int skipI = 0, skipJ = 0;
    const char* evs = getenv("CASMTR_DS_GEMM16_WST");
    const int wst = evs && evs[0] == '1';
    const char* evw = getenv("CASMTR_DS_GEMM16_WIDE");
    const int NIB2 = L / 256, NJBf = S / DS_BN;
    if (evw && evw[0] == '1' && NIB2 > 0 && NJBf > 0) {
        const int nt = ((NJBf + 7) / 8) * ((NIB2 + 7) / 8) * 64;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm16w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DS16W_LDS);
        CASMTR_LAUNCH_TIMED(CASMTR_PROF_DS_GEMM, ds_gemm16w_kernel, dim3(nt, B), dim3(256), DS16W_LDS, s, w.imgA, w.imgB, w.fa, w.fb, mask0 ? 1 : 0,
                            sim, w, L, S, C / 16, NJB, NIB, NIB2, NJBf, g_debug_flags >> 12);
        skipI = 2 * NIB2; skipJ = NJBf;
        if (skipI >= NIB && skipJ >= NJB) { CASMTR_CHECK_LAUNCH(); return 0; }
    }
    const char* ev3 = getenv("CASMTR_DS_GEMM16_STAGES");
    if (ev3 && ev3[0] == '4') {
        const size_t lds = DS16_LDS3;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CASMTR_LAUNCH_TIMED(skipI ? CASMTR_PROF_DS_GEMM_EDGE : CASMTR_PROF_DS_GEMM, ds_gemm16_kernel<4>, dim3(ntiles, B), dim3(256), lds, s, w.imgA, w.imgB, w.fa, w.fb,
                            mask0, mask1, sim, w, L, S, C / 16, NJB, NIB, skipI, skipJ, wst);
    } else if (ev3 && ev3[0] == '2') {
        const size_t lds = DS16_LDS;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CASMTR_LAUNCH_TIMED(skipI ? CASMTR_PROF_DS_GEMM_EDGE : CASMTR_PROF_DS_GEMM, ds_gemm16_kernel<2>, dim3(ntiles, B), dim3(256), lds, s, w.imgA, w.imgB, w.fa, w.fb,
                            mask0, mask1, sim, w, L, S, C / 16, NJB, NIB, skipI, skipJ, wst);
    } else {
        const size_t lds = DS16_LDS3;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CASMTR_LAUNCH_TIMED(skipI ? CASMTR_PROF_DS_GEMM_EDGE : CASMTR_PROF_DS_GEMM, ds_gemm16_kernel<3>, dim3(ntiles, B), dim3(256), lds, s, w.imgA, w.imgB, w.fa, w.fb,
                            mask0, mask1, sim, w, L, S, C / 16, NJB, NIB, skipI, skipJ, wst);
    }
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== sparse pass 2
// The dense pass 2 (ds_conf_kernel) streams the whole matrix to find the few entries that matter: those at or above their row's /
// column's near-tie threshold (index candidates) and those whose confidence can exceed `thr` (conf = p01 p10 > thr needs
// p01 > thr, i.e. x > rmax + log(thr rsum) = tau).  The GEMM epilogue already left the maximum of every (row, 128-column block)
// and (column, 128-row block) segment in rp_m / cp_m, and a segment whose maximum is below the threshold holds no such entry --
// on matching features that is all but one or two of a row's 85 segments.  One wave per 64 rows (lane <-> row while the
// segment maxima stream past, coalesced); every flagged (row, block) segment is then read by the whole wave (512 B).  Column
// segments (128 rows x one column, strided) only look for index candidates: every entry with conf > thr sits in a flagged ROW
// segment, which also feeds the column-best atomics.  Same candidate lists and best-of-row / best-of-column keys as the dense pass.
#define DS_SP_CHUNK 8
__global__ __launch_bounds__(256) void ds_sparse_kernel(const float* __restrict__ sim, DsWs w, int B, int L, int S, int NJB, int NIB,
                                                        float thr, float kthr) {
    const int lane = threadIdx.x & 63;
    const int gw0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int RG = (L + 63) / 64, CG = (S + 63) / 64;
    const int RCH = (NJB + DS_SP_CHUNK - 1) / DS_SP_CHUNK, CCH = (NIB + DS_SP_CHUNK - 1) / DS_SP_CHUNK;   // block chunks per wave
    if (gw0 < B * RG * RCH) {
        const int gw = gw0 / RCH, t0 = (gw0 % RCH) * DS_SP_CHUNK;
        const int b = gw / RG, g = gw % RG, i = g * 64 + lane;
        const bool ok = i < L;
        const size_t ro = (size_t)b * L + (ok ? i : L - 1);
        const float rm = w.rmax[ro], rs = w.rsum[ro], rt = w.rthr[ro];
        const float tau = rm + __logf(thr * rs) - 1e-2f, rinv = 1.0f / rs;
        const float keep = 1.0f - 2.0f * ds_conf_band(kthr, w.namax[b], w.nbmax[b]), cmin = 0.9f * thr;
        const float lim = fminf(rt, tau);
        float mv[DS_SP_CHUNK];
#pragma unroll
        for (int k = 0; k < DS_SP_CHUNK; ++k)   // all loads in flight before the first use
            mv[k] = w.rp_m[((size_t)b * NJB + min(t0 + k, NJB - 1)) * L + (ok ? i : L - 1)];
#pragma unroll
        for (int k = 0; k < DS_SP_CHUNK; ++k) {
            const int tJ = t0 + k;
            const float m = mv[k];
            unsigned long long bal = __ballot(ok && tJ < NJB && m >= lim && m != NEG_FILL);
            while (bal) {
                // up to four flagged segments per round: their 2 x 256-byte reads are all in flight before the first is looked at (a
                // wave meets ~9 flagged segments; one dependent HBM round trip each was what the kernel's 150 us consisted of)
                int ls[4], nl = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (bal) { ls[q] = __ffsll((long long)bal) - 1; bal &= bal - 1; ++nl; } else ls[q] = 0;
                }
                float xs[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* p = sim + ((size_t)b * L + g * 64 + ls[q]) * S;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int j = tJ * DS_BN + lane + 64 * u;
                        xs[q][u] = (q < nl && j < S) ? p[j] : NEG_FILL;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nl) break;   // wave-uniform
                    const int l = ls[q];
                    const int r = g * 64 + l;
                    const float rm_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm), l));
                    const float rt_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rt), l));
                    const float tau_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tau), l));
                    const float rinv_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rinv), l));
                    const size_t o = (size_t)b * L + r;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int j = tJ * DS_BN + lane + 64 * u;
                        const float x = xs[q][u];
                        if (x == NEG_FILL) continue;   // also: column out of range / no segment in this slot
                        if (x >= rt_l) {
                            const int slot = atomicAdd(w.rcnt + o, 1);
                            if (slot < DS_CAND_CAP) w.rcand[o * DS_CAND_CAP + slot] = j; else *w.ovf = 1;
                        }
                        if (x > tau_l) {
                            const size_t co = (size_t)b * S + j;
                            const float cf = (__expf(x - w.cmax[co]) * (1.0f / w.csum[co])) * (__expf(x - rm_l) * rinv_l);
                            if (cf >= 0.f) {
                                const unsigned long long hi = (unsigned long long)__float_as_uint(cf) << 32;
                                const unsigned long long oldr = atomicMax(w.rbest + o, hi | (0xFFFFFFFFu - (unsigned)j));
                                const unsigned long long oldc = atomicMax(w.cbest + co, hi | (0xFFFFFFFFu - (unsigned)r));
                                // Borderline entries (see ds_xdecide_launch): the previous maximum `old` and this entry are within the
                                // error band of each other -> neither ordering is certain -> both go on the list for exact
                                // re-decision.  Every entry within the band of the FINAL maximum is caught this way: it either
                                // meets the final maximum as `old`, or is met as `old` by the chain of later maxima that ends there.
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
        }
    } else if (gw0 < B * RG * RCH + B * CG * CCH) {
        const int gc0 = gw0 - B * RG * RCH;
        const int gc = gc0 / CCH, t0 = (gc0 % CCH) * DS_SP_CHUNK;
        const int b = gc / CG, g = gc % CG, j = g * 64 + lane;
        const bool ok = j < S;
        const float ct = w.cthr[(size_t)b * S + (ok ? j : S - 1)];
        float mv[DS_SP_CHUNK];
#pragma unroll
        for (int k = 0; k < DS_SP_CHUNK; ++k)
            mv[k] = w.cp_m[((size_t)b * NIB + min(t0 + k, NIB - 1)) * S + (ok ? j : S - 1)];
#pragma unroll
        for (int k = 0; k < DS_SP_CHUNK; ++k) {
            const int tI = t0 + k;
            const float m = mv[k];
            unsigned long long bal = __ballot(ok && tI < NIB && m >= ct && m != NEG_FILL);
            while (bal) {   // four flagged column segments per round, their group maxima and then their entries in flight together
                int ls[4], nl = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (bal) { ls[q] = __ffsll((long long)bal) - 1; bal &= bal - 1; ++nl; } else ls[q] = 0;
                }
                float gm[4][2], xs[4][2], ctl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = g * 64 + ls[q];
                    ctl[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ct), ls[q]));
#pragma unroll
                    for (int u = 0; u < 2; ++u)   // row lane + 64 u of the block belongs to the 16-row group (wr = u, ti = lane >> 5, hi = (lane >> 2) & 1)
                        gm[q][u] = q < nl ? w.cg_m[(((size_t)b * NIB + tI) * 8 + u * 4 + (lane >> 5) * 2 + ((lane >> 2) & 1)) * S + col] : NEG_FILL;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = g * 64 + ls[q];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int i = tI * DS_BM + lane + 64 * u;
                        xs[q][u] = (q < nl && i < L && gm[q][u] >= ctl[q]) ? sim[((size_t)b * L + i) * S + col] : NEG_FILL;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nl) break;   // wave-uniform
                    const size_t co = (size_t)b * S + g * 64 + ls[q];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int i = tI * DS_BM + lane + 64 * u;
                        const float x = xs[q][u];
                        if (x == NEG_FILL || !(x >= ctl[q])) continue;
                        const int slot = atomicAdd(w.ccnt + co, 1);
                        if (slot < DS_CAND_CAP) w.ccand[co * DS_CAND_CAP + slot] = i; else *w.ovf = 1;
                    }
                }
            }
        }
    }
}

int ds_sparse_launch(const float* sim, const DsWs& w, int B, int L, int S, float thr, float kthr, hipStream_t s) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const int waves = B * ((L + 63) / 64) * ((NJB + DS_SP_CHUNK - 1) / DS_SP_CHUNK) + B * ((S + 63) / 64) * ((NIB + DS_SP_CHUNK - 1) / DS_SP_CHUNK);
    hipLaunchKernelGGL(ds_sparse_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, sim, w, B, L, S, NJB, NIB, thr, kthr);
    CASMTR_CHECK_LAUNCH();
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
