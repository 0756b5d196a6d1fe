// Matching kernels for gfx950.
//   casmtr_dual_softmax_fwd   CoarseMatching.forward + get_coarse_match    src/model/functions/coarse_matching.py:40-153
//   casmtr_window_match_fwd   CascadeMatching.forward (one direction)      src/model/functions/cascade_matching.py:63-161
//   casmtr_nms_select_fwd     CascadeMatching.get_coarse_match (inference) cascade_matching.py:170-261,317-331
//                             + PostProcess.apply None/'maxpool_nms'       post_processing.py:41-44,111-121
//                             + mask_window_border[_with_padding]          cascade_functions.py:120-172
// Arithmetic for everything that feeds an index is the oracle's: operands pre-scaled by 1/sqrt(C) (division or
// reciprocal multiply, `recip`), fp32 fmaf chain over c ascending (v_mfma_f32_32x32x2_f32 is exactly that chain),
// result scaled by 1/T, masked entries = -1e9, argmax = first maximum of the LOGITS.
#include <stdlib.h>
#include "common.hpp"
#include "ds_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

// =================================================================================================== dual softmax
// Pass 1: sim = (f0/sqrtC) . (f1/sqrtC)^T / T on the fp32 matrix cores, 128x128 block tile, 4 waves x (64x64),
// BK = 32.  The epilogue parks the tile in LDS, derives per-row / per-column (max, first argmax, sum exp) partials
// for this block, and streams the tile to HBM (it is re-read once by pass 2; 468 MB/pair at 832x832 is cheaper
// to move at 8 TB/s than to recompute at 157 TFLOP/s).
extern "C" size_t casmtr_dual_softmax_ws_bytes(int B, int L, int S) { return ds_carve(nullptr, nullptr, B, L, S, 0); }
extern "C" size_t casmtr_dual_softmax_split_ws_bytes(int B, int L, int S, int C) { return ds_carve(nullptr, nullptr, B, L, S, C); }

template <bool RECIP>
__device__ __forceinline__ void ds_gemm_tile(int bx, int nbx, const float* __restrict__ f0, const float* __restrict__ f1,
                                             const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                             float* __restrict__ sim, const DsWs& w, int L, int S, int C, float sqrtC,
                                             float inv_sqrtC, float T, float invT, int NJB, int NIB, int store) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // As[128][33] | Bs[128][33], then epilogue scratch
    float (*As)[33] = reinterpret_cast<float (*)[33]>(smem);
    float (*Bs)[33] = reinterpret_cast<float (*)[33]>(smem + 128 * 33);
    // Tile order: 8x8 super-tiles (8 A panels + 8 B panels = 2.1 MB, L2-resident), one contiguous run of super-tiles per
    // XCD, so that operand panels are fetched from the fabric once per super-tile instead of once per tile.  Speed only.
    const int NSJ = (NJB + 7) >> 3;
    const int t = xcd_chunk_remap(bx, nbx);
    const int st = t >> 6, wi = t & 63;
    const int tI = (st / NSJ) * 8 + (wi >> 3), tJ = (st % NSJ) * 8 + (wi & 7);
    if (tI >= NIB || tJ >= NJB) return;   // padding of the super-tile grid (whole workgroup exits: no barrier is skipped)
    const int b = blockIdx.y, i0 = tI * DS_BM, j0 = tJ * DS_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid >> 1, lc0 = (tid & 1) * 16;
    const float* ap = f0 + ((size_t)b * L + (i0 + lrow < L ? i0 + lrow : L - 1)) * C + lc0;
    const float* bp = f1 + ((size_t)b * S + (j0 + lrow < S ? j0 + lrow : S - 1)) * C + lc0;
    f32x4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
        bv[i] = *reinterpret_cast<const f32x4*>(bp + 4 * i);
    }
    for (int k0 = 0; k0 < C; k0 += DS_BK) {
        __syncthreads();  // previous k-tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                As[lrow][lc0 + 4 * i + c] = div_scalar<RECIP>(av[i][c], sqrtC, inv_sqrtC);
                Bs[lrow][lc0 + 4 * i + c] = div_scalar<RECIP>(bv[i][c], sqrtC, inv_sqrtC);
            }
        if (k0 + DS_BK < C) {  // prefetch the next k-tile into registers; it lands while the MFMAs below run
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = *reinterpret_cast<const f32x4*>(ap + k0 + DS_BK + 4 * i);
                bv[i] = *reinterpret_cast<const f32x4*>(bp + k0 + DS_BK + 4 * i);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DS_BK / 2; ++kk) {
            const int kc = 2 * kk + (lane >> 5), rr = lane & 31;
            const float a0 = As[wr * 64 + rr][kc], a1 = As[wr * 64 + 32 + rr][kc];
            const float b0 = Bs[wc * 64 + rr][kc], b1 = Bs[wc * 64 + 32 + rr][kc];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done reading As/Bs: the region becomes the epilogue's scratch
    if (store) ds_tile_epilogue<RECIP, false, true>(acc, smem, nullptr, nullptr, mask0, mask1, sim, w, b, tI, tJ, L, S, T, invT, NJB, NIB);
    else ds_tile_epilogue<RECIP, false, false>(acc, smem, nullptr, nullptr, mask0, mask1, sim, w, b, tI, tJ, L, S, T, invT, NJB, NIB);
}

// One workgroup per tile.  As the split path's fallback (guard != nullptr; runs only when its candidate lists overflowed) the grid
// is small and every workgroup walks `ntiles` tiles: the usual case -- flag clear -- is then a few thousand immediate exits.
template <bool RECIP>
__global__ __launch_bounds__(256, 3) void ds_gemm_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                         const uint8_t* __restrict__ mask0,
                                                         const uint8_t* __restrict__ mask1, float* __restrict__ sim,
                                                         DsWs w, int L, int S, int C, float sqrtC, float inv_sqrtC,
                                                         float T, float invT, int NJB, int NIB, const int* __restrict__ guard,
                                                         int ntiles, int store) {
    if (!guard) {
        ds_gemm_tile<RECIP>(blockIdx.x, gridDim.x, f0, f1, mask0, mask1, sim, w, L, S, C, sqrtC, inv_sqrtC, T, invT, NJB, NIB, store);
        return;
    }
    if (*guard == 0) return;
    for (int bx = blockIdx.x; bx < ntiles; bx += gridDim.x) {
        ds_gemm_tile<RECIP>(bx, ntiles, f0, f1, mask0, mask1, sim, w, L, S, C, sqrtC, inv_sqrtC, T, invT, NJB, NIB, 1);
        __syncthreads();   // the epilogue's scratch becomes the next tile's operand buffers
    }
}

// combine block partials -> per row/col (max, sum, first argmax); next_conf = softmax value at the argmax = 1/sum.
// Split path (pa == nullptr): no argmax here; instead the near-tie threshold of the row / column: every entry whose exact logit
// can equal the exact maximum has an approximate logit >= max~ - 2 e, e = 2^-15 |a_i| max_j |b_j| / (C T) (ds_split.hip).
struct DsReduceSide {   // one direction of ds_reduce_kernel: rows (blocks = column blocks) or columns
    const float *pm, *ps;
    const int* pa;
    int nblk, N, total;
    float *omax, *osum;
    int64_t* oidx;
    float* oconf;
    const float* nrm;
    int npad;
    const unsigned* other_max;
    float* othr;
};
// both directions in one launch (round 6: two launches of ~20 us each, plus two more that exit at once in the guarded fallback)
__global__ __launch_bounds__(256) void ds_reduce_kernel(const DsReduceSide r0, const DsReduceSide r1, float kthr, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;
    const int nb0 = (r0.total + 255) / 256;
    const bool second = (int)blockIdx.x >= nb0;
    const DsReduceSide& R = second ? r1 : r0;
    const float *pm = R.pm, *ps = R.ps;
    const int* pa = R.pa;
    const int nblk = R.nblk, N = R.N, total = R.total, npad = R.npad;
    float *omax = R.omax, *osum = R.osum, *oconf = R.oconf, *othr = R.othr;
    int64_t* oidx = R.oidx;
    const float* nrm = R.nrm;
    const unsigned* other_max = R.other_max;
    const int t = ((int)blockIdx.x - (second ? nb0 : 0)) * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / N, i = t % N;
    const size_t base = (size_t)b * nblk * N + i;
    // only ~1.3 workgroups per CU: the loads of 8 partials are issued together, the argmax is fetched once at the end
    float m = pm[base]; int kb = 0;
#pragma unroll 8
    for (int k = 1; k < nblk; ++k) {
        const float x = pm[base + (size_t)k * N];
        if (x > m) { m = x; kb = k; }   // strict: the first block holding the maximum wins
    }
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < nblk; ++k) s += ps[base + (size_t)k * N] * __expf(pm[base + (size_t)k * N] - m);
    omax[t] = m; osum[t] = s; oconf[t] = 1.0f / s;
    if (pa) oidx[t] = pa[base + (size_t)kb * N];
    else {
        // fully masked row (all NEG_FILL): no candidates, the fix-up pass answers 0 = the first maximum
        const float e2 = kthr * nrm[(size_t)b * npad + i] * __uint_as_float(other_max[b]);
        othr[t] = (m == NEG_FILL) ? INFINITY : m - e2 - fabsf(m) * 9.5367431640625e-7f;
    }
}

// Pass 2: conf = softmax10 * softmax01 (coarse_matching.py:66-68), best-of-row / best-of-column (value, first index)
// through packed 64-bit atomicMax; optionally overwrites sim with conf (the reference's data['stage_8c']['conf_matrix']).
// Streaming kernel: a wave owns 256 consecutive columns (one float4 per lane) of a 64-row strip; row statistics come in
// through wave-uniform scalar loads, the column best lives in registers, the row best is a DPP wave reduction -> one
// atomic per (row, wave).  No LDS, full occupancy, 1 KiB coalesced reads.
#define DSC_ROWS 64
// When conf_matrix is not materialised, only entries that can exceed `thr` matter downstream (conf > thr is tested first, and
// a row / column maximum that takes part in a match is attained by such an entry): conf = p01 * p10 > thr needs p01 > thr,
// i.e. sim > rmax + log(thr * rsum).  A (row, wave) pair none of whose 256 entries passes that test (minus a slack far above
// the rounding of __expf) is skipped after 4 compares and a ballot -- almost all of them: the pass becomes a pure read.
// CAND (split path): entries at or above the near-tie threshold of their row / column are appended to that row's / column's
// candidate list (almost always exactly one: the maximum itself).  Masked entries never are.
template <bool CAND>
__global__ __launch_bounds__(256) void ds_conf_kernel(float* __restrict__ sim, DsWs w, int L, int S, int want_conf, float thr,
                                                      const int* __restrict__ guard, float kthr) {
    if (guard && *guard == 0) return;
    const int b = blockIdx.z, i0 = blockIdx.y * DSC_ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.x * 1024 + wave * 256 + lane * 4;
    if (blockIdx.x * 1024 + wave * 256 >= S) return;
    const bool vec = (S & 3) == 0 && j + 3 < S;
    const int nr = min(DSC_ROWS, L - i0);
    float cm[4], cinv[4], best[4], cth[4];
    int bi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int jj = min(j + u, S - 1);
        cth[u] = CAND ? w.cthr[(size_t)b * S + jj] : 0.f;
        cm[u] = w.cmax[(size_t)b * S + jj];
        cinv[u] = 1.0f / w.csum[(size_t)b * S + jj];
        best[u] = -1.f; bi[u] = 0;
    }
    const float* rmax = w.rmax + (size_t)b * L + i0;   // wave-uniform
    const float* rsum = w.rsum + (size_t)b * L + i0;
    const float* rthr = CAND ? w.rthr + (size_t)b * L + i0 : nullptr;
    // CAND: borderline entries for the exact re-decision of the match list (ds_split.hip: ds_xdecide_launch; same rule as ds_flagged_kernel)
    const float keep = CAND ? 1.0f - 2.0f * ds_conf_band(kthr, w.namax[b], w.nbmax[b]) : 0.f, cmin = 0.9f * thr;
    float* base = sim + ((size_t)b * L + i0) * S + j;
    constexpr int RU = 4;  // rows in flight per lane
    for (int r0 = 0; r0 < nr; r0 += RU) {
        float x[RU][4];
#pragma unroll
        for (int q = 0; q < RU; ++q) {
            const int r = min(r0 + q, nr - 1);
            if (vec) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(base + (size_t)r * S);
                x[q][0] = t.x; x[q][1] = t.y; x[q][2] = t.z; x[q][3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) x[q][u] = (j + u < S) ? base[(size_t)r * S + u] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < RU; ++q) {
            const int r = r0 + q;
            if (r >= nr) break;  // wave-uniform
            const float rm = rmax[r], rs = rsum[r], rinv = 1.0f / rs;
            if (CAND) {
                const float rt = rthr[r];   // wave-uniform
                const bool anyc = x[q][0] >= rt || x[q][1] >= rt || x[q][2] >= rt || x[q][3] >= rt ||
                                  x[q][0] >= cth[0] || x[q][1] >= cth[1] || x[q][2] >= cth[2] || x[q][3] >= cth[3];
                if (__ballot(anyc) != 0ull) {   // rare: about one (row, wave) in 43 holds its row's maximum
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u >= S || x[q][u] == NEG_FILL) continue;
                        if (x[q][u] >= rt) {
                            const size_t o = (size_t)b * L + i0 + r;
                            const int slot = atomicAdd(w.rcnt + o, 1);
                            if (slot < DS_CAND_CAP) w.rcand[o * DS_CAND_CAP + slot] = j + u; else *w.ovf = 1;
                        }
                        if (x[q][u] >= cth[u]) {
                            const size_t o = (size_t)b * S + j + u;
                            const int slot = atomicAdd(w.ccnt + o, 1);
                            if (slot < DS_CAND_CAP) w.ccand[o * DS_CAND_CAP + slot] = i0 + r; else *w.ovf = 1;
                        }
                    }
                }
            }
            if (!want_conf && thr > 0.f) {
                const float tau = rm + __logf(thr * rs) - 1e-2f;   // wave-uniform
                const bool any = x[q][0] > tau || x[q][1] > tau || x[q][2] > tau || x[q][3] > tau;
                if (__ballot(any) == 0ull) continue;
            }
            float cf[4];
            float rbest = -1.f; int rj = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float p01 = __expf(x[q][u] - rm) * rinv;
                const float p10 = __expf(x[q][u] - cm[u]) * cinv[u];
                cf[u] = (j + u < S) ? p10 * p01 : -1.f;
                if (CAND && fminf(cf[u], best[u]) > cmin && fminf(cf[u], best[u]) >= fmaxf(cf[u], best[u]) * keep) {
                    ds_x_append(w, b * L + i0 + r, j + u);     // this entry and the column's best so far cannot be ordered for certain
                    ds_x_append(w, b * L + bi[u], j + u);
                }
                if (cf[u] > best[u]) { best[u] = cf[u]; bi[u] = i0 + r; }
                if (cf[u] > rbest) { rbest = cf[u]; rj = j + u; }
            }
            if (want_conf) {
                if (vec) *reinterpret_cast<f32x4*>(base + (size_t)r * S) = (f32x4){cf[0], cf[1], cf[2], cf[3]};
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (j + u < S) base[(size_t)r * S + u] = cf[u];
                }
            }
            // row best over the wave's 256 columns: max value, then the first lane holding it (lanes are column-ordered)
            const unsigned kb = rbest >= 0.f ? __float_as_uint(rbest) : 0u;
            const unsigned wm = wave_max_u32(kb);
            const unsigned long long bal = __ballot(kb == wm && rbest >= 0.f);
            if (CAND) {   // more than one entry of this (row, wave) within the band of the wave's best: all of them are borderline
                const float wmf = __uint_as_float(wm);
                bool nb[4];
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    nb[u] = cf[u] > cmin && cf[u] >= wmf * keep;
                    cnt += __popcll(__ballot(nb[u]));
                }
                if (cnt > 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (nb[u]) ds_x_append(w, b * L + i0 + r, j + u);
                }
            }
            if (bal && lane == __ffsll((long long)bal) - 1) {
                const unsigned long long key = ((unsigned long long)wm << 32) | (0xFFFFFFFFu - (unsigned)rj);
                const unsigned long long old = atomicMax(w.rbest + (size_t)b * L + i0 + r, key);
                if (CAND) {
                    const float cw = __uint_as_float(wm), co = __uint_as_float((unsigned)(old >> 32));
                    if (fminf(cw, co) > cmin && fminf(cw, co) >= fmaxf(cw, co) * keep) {
                        ds_x_append(w, b * L + i0 + r, rj);
                        ds_x_append(w, b * L + i0 + r, (int)(0xFFFFFFFFu - (unsigned)(old & 0xFFFFFFFFu)));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (j + u < S && best[u] >= 0.f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(best[u]) << 32) | (0xFFFFFFFFu - (unsigned)bi[u]);
            const unsigned long long old = atomicMax(w.cbest + (size_t)b * S + j + u, key);
            if (CAND) {
                const float co = __uint_as_float((unsigned)(old >> 32));
                if (fminf(best[u], co) > cmin && fminf(best[u], co) >= fmaxf(best[u], co) * keep) {
                    ds_x_append(w, b * L + bi[u], j + u);
                    ds_x_append(w, b * L + (int)(0xFFFFFFFFu - (unsigned)(old & 0xFFFFFFFFu)), j + u);
                }
            }
        }
}

// Pass 2 of the exact path WITHOUT the matrix (round 6; the split path's ds_flagged_kernel has the reasoning).  When conf_matrix is not
// requested, casmtr_dual_softmax_fwd no longer writes the [B, L, S] matrix: only entries with x > tau = rmax + log(thr rsum) can reach
// conf > thr, they sit in the (row, 128-column) segments whose maximum -- left in rp_m by the GEMM epilogue -- exceeds tau, and those
// segments are recomputed here: the listed rows gathered from feat_c0 against the block's rows of feat_c1 with ds_gemm_tile's operand
// scaling, k-loop and MFMA sequence (the same fmaf chain: bit-identical logits), then ds_conf_kernel's confidence arithmetic and packed
// best-of-row / best-of-column atomics for the entries above tau.  Entries at or below tau have conf < thr: they can neither be
// selected nor be the row / column maximum of a selected entry, so the match list is the dense pass's.
#define DSE_QCAP 4000
template <bool RECIP>
__global__ __launch_bounds__(256, 2) void ds_eflagged_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                             const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, DsWs w,
                                                             int B, int L, int S, int C, float sqrtC, float inv_sqrtC, float T, float invT,
                                                             float thr, int NJB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // As[128][33] | Bs[128][33] (later: the queue), then the per-row tables
    float (*As)[33] = reinterpret_cast<float (*)[33]>(smem);
    float (*Bs)[33] = reinterpret_cast<float (*)[33]>(smem + 128 * 33);
    float* t_tau = smem + 2 * 128 * 33;
    float *t_rm = t_tau + 128, *t_rinv = t_tau + 256;
    int* line = reinterpret_cast<int*>(t_tau + 384);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int nitems = w.fl_tn[0];
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int e = w.fl_t[it], lid = e >> 8, tile = e & 255;      // row lists only (ds_flagscan_kernel, exact_rows_only)
        const int b = lid / NJB, blk = lid - b * NJB, j0 = blk * DS_BN;
        const int n = w.fl_rn[lid] - tile * 128;
        const int* list = w.fl_r + (size_t)lid * L + tile * 128;
        __syncthreads();   // the previous item's tables and queue are no longer read
        if (tid < 128) {
            const int li = list[tid < n ? tid : 0];
            line[tid] = li;
            const size_t o = (size_t)b * L + li;
            const float rm = w.rmax[o], rs = w.rsum[o];
            t_tau[tid] = tid < n ? rm + __logf(thr * rs) - 1e-2f : INFINITY;
            t_rm[tid] = rm; t_rinv[tid] = 1.0f / rs;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int lrow = tid >> 1, lc0 = (tid & 1) * 16;
        const float* ap = f0 + ((size_t)b * L + line[lrow]) * C + lc0;
        const float* bp = f1 + ((size_t)b * S + (j0 + lrow < S ? j0 + lrow : S - 1)) * C + lc0;
        f32x4 av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
            bv[i] = *reinterpret_cast<const f32x4*>(bp + 4 * i);
        }
        for (int k0 = 0; k0 < C; k0 += DS_BK) {   // ds_gemm_tile's k-loop
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    As[lrow][lc0 + 4 * i + c] = div_scalar<RECIP>(av[i][c], sqrtC, inv_sqrtC);
                    Bs[lrow][lc0 + 4 * i + c] = div_scalar<RECIP>(bv[i][c], sqrtC, inv_sqrtC);
                }
            if (k0 + DS_BK < C) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[i] = *reinterpret_cast<const f32x4*>(ap + k0 + DS_BK + 4 * i);
                    bv[i] = *reinterpret_cast<const f32x4*>(bp + k0 + DS_BK + 4 * i);
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < DS_BK / 2; ++kk) {
                const int kc = 2 * kk + (lane >> 5), rr = lane & 31;
                const float a0 = As[wr * 64 + rr][kc], a1 = As[wr * 64 + 32 + rr][kc];
                const float b0 = Bs[wc * 64 + rr][kc], b1 = Bs[wc * 64 + 32 + rr][kc];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();   // operand tiles dead: the region becomes the queue of entries above tau
        int* qn = reinterpret_cast<int*>(smem);
        int2* queue = reinterpret_cast<int2*>(smem + 4);
        if (tid == 0) qn[0] = 0;
        __syncthreads();
        int lane_o = lane;   // (opaque: keeps the entries' 64 (row, column) codes from being hoisted out of the item loop, see ds_flagged_kernel)
        asm volatile("" : "+v"(lane_o));
        const int hi = lane_o >> 5, ln = lane_o & 31;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int lc = wc * 64 + tj * 32 + ln, gj = j0 + lc;
            const bool c_ok = gj < S && (!mask1 || mask1[(size_t)b * S + (gj < S ? gj : S - 1)] != 0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float x = div_scalar<RECIP>(acc[ti][tj][r], T, invT);
                    if (c_ok && x > t_tau[lr]) {
                        const int slot = atomicAdd(qn, 1);
                        if (slot < DSE_QCAP) queue[slot] = make_int2((lr << 8) | lc, __float_as_int(x));
                    }
                }
            }
        }
        __syncthreads();
        const int nq = min(qn[0], DSE_QCAP);   // (a tile with more than DSE_QCAP entries above tau: thr * rsum tiny -- the caller keeps such thresholds on the dense pass)
        for (int q = tid; q < nq; q += 256) {
            const int2 ent = queue[q];
            const int lr = ent.x >> 8, j = j0 + (ent.x & 255), li = line[lr];
            if (mask0 && mask0[(size_t)b * L + li] == 0) continue;     // padding: the GEMM's entry is NEG_FILL there
            const float x = __int_as_float(ent.y);
            const size_t o = (size_t)b * L + li, co = (size_t)b * S + j;
            const float p01 = __expf(x - t_rm[lr]) * t_rinv[lr];
            const float p10 = __expf(x - w.cmax[co]) * (1.0f / w.csum[co]);
            const float cf = p10 * p01;                                  // ds_conf_kernel's arithmetic, operation for operation
            if (cf >= 0.f) {
                const unsigned long long hk = (unsigned long long)__float_as_uint(cf) << 32;
                atomicMax(w.rbest + o, hk | (0xFFFFFFFFu - (unsigned)j));
                atomicMax(w.cbest + co, hk | (0xFFFFFFFFu - (unsigned)li));
            }
        }
    }
}

// coarse_matching.py:116-132: conf > thr, border removal, mutual maximum BY VALUE, first j per row.
__global__ __launch_bounds__(256) void ds_flag_kernel(DsWs w, float thr, int border_rm, const int32_t* __restrict__ valid_hw,
                                                      int h0c, int w0c, int h1c, int w1c, int L, int S, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / L, i = t % L;
    const unsigned long long key = w.rbest[t];
    float cf = __uint_as_float((unsigned)(key >> 32));
    int j = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    bool ok = cf > thr;
    if (ok) ok = (unsigned)(w.cbest[(size_t)b * S + j] >> 32) == (unsigned)(key >> 32);
    if (w.rdec && (w.rdec[t] & 1)) {   // split path: this row's (best column, conf > thr, mutual maximum) was re-decided exactly
        ok = (w.rdec[t] & 2) != 0;
        j = w.rdec_j[t];
        cf = w.rdec_cf[t];
    }
    if (ok && border_rm > 0) {
        const int vh0 = valid_hw ? valid_hw[b * 4 + 0] : h0c, vw0 = valid_hw ? valid_hw[b * 4 + 1] : w0c;
        const int vh1 = valid_hw ? valid_hw[b * 4 + 2] : h1c, vw1 = valid_hw ? valid_hw[b * 4 + 3] : w1c;
        const int y0 = i / w0c, x0 = i % w0c, y1 = j / w1c, x1 = j % w1c;
        if (y0 < border_rm || x0 < border_rm || y0 >= vh0 - border_rm || x0 >= vw0 - border_rm) ok = false;
        if (y1 < border_rm || x1 < border_rm || y1 >= vh1 - border_rm || x1 >= vw1 - border_rm) ok = false;
    }
    w.flags[t] = ok ? 1 : 0;
    w.jsel[t] = j;
    w.csel[t] = cf;
}

// =================================================================================================== ordered compaction
// flags[total] -> (b,i) ordered lists, as torch.where() would return them.  3 small kernels: per-1024 counts,
// single-workgroup exclusive scan, ordered write.  `keep_one`: the reference's "mask[:, 0] = True" when nothing
// survived in the whole batch (cascade_matching.py:254-255).
__global__ __launch_bounds__(1024) void compact_count_kernel(const unsigned char* __restrict__ flags, int total,
                                                             int* __restrict__ blk) {
    __shared__ int wsum[16];
    const int t = blockIdx.x * 1024 + threadIdx.x;
    const bool f = t < total && flags[t];
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += wsum[i];
        blk[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void compact_scan_kernel(int* __restrict__ blk, int nblk, int64_t* __restrict__ n_out,
                                                            int keep_one, int B) {
    // sequential chunks of 1024 blocks; nblk is a few hundred at most for this workload
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blk[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nblk) blk[i] = carry + part[threadIdx.x] - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blk[nblk] = carry;  // total
        *n_out = (carry == 0 && keep_one) ? B : carry;
    }
}

__global__ __launch_bounds__(1024) void compact_write_kernel(const unsigned char* __restrict__ flags,
                                                             const int64_t* __restrict__ jsrc,
                                                             const float* __restrict__ csrc, const int* __restrict__ blk,
                                                             int nblk, int total, int N, int keep_one,
                                                             int64_t* __restrict__ b_ids, int64_t* __restrict__ i_ids,
                                                             int64_t* __restrict__ j_ids, float* __restrict__ mconf) {
    __shared__ int wsum[16];
    const int t = blockIdx.x * 1024 + threadIdx.x;
    if (blk[nblk] == 0) {
        if (keep_one && t < total && (t % N) == 0) {
            const int b = t / N;
            b_ids[b] = b; i_ids[b] = 0; j_ids[b] = jsrc[t]; mconf[b] = csrc[t];
        }
        return;
    }
    const bool f = t < total && flags[t];
    const unsigned long long bal = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = blk[blockIdx.x];
    for (int i = 0; i < wv; ++i) off += wsum[i];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (f) {
        b_ids[off] = t / N; i_ids[off] = t % N; j_ids[off] = jsrc[t]; mconf[off] = csrc[t];
    }
}

static int run_compaction(const unsigned char* flags, const int64_t* jsrc, const float* csrc, int* blk, int total, int N,
                          int keep_one, int B, int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                          int64_t* n_matches, hipStream_t s) {
    const int nblk = (total + 1023) / 1024;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblk), dim3(1024), 0, s, flags, total, blk);
    CASMTR_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, blk, nblk, n_matches, keep_one, B);
    CASMTR_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_write_kernel, dim3(nblk), dim3(1024), 0, s, flags, jsrc, csrc, blk, nblk, total, N,
                       keep_one, b_ids, i_ids, j_ids, mconf);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// guarded zero fill (the fallback sequence cannot use hipMemsetAsync: it must be a no-op when the guard is clear)
__global__ __launch_bounds__(256) void ds_zero_kernel(unsigned long long* __restrict__ p, size_t n, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) p[t] = 0ull;
}

// exact pass 1 + statistics (+ pass 2 when `with_conf`); guard: device flag, null = run
static int ds_exact_passes(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, float temperature,
                           int recip, float thr, int want_conf, float* sim_ws, const DsWs& w, int64_t* next_idx01,
                           float* next_conf01, int64_t* next_idx10, float* next_conf10, int B, int L, int S, int C,
                           const int* guard, hipStream_t s, int store = 1) {
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const float sqrtC = (float)sqrt((double)C);
    const size_t gemm_lds = sizeof(float) * (4 * 32 * 65 + 2 * 2 * 128 * 3);  // >= the 2 x [128][33] operand tiles
    // per-device attribute: set on every call (cheap), never cached in a process-wide flag
    if (recip)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds);
    else
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_gemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds);
    {
        ProfScope ps(guard ? -1 : CASMTR_PROF_DS_GEMM, s, recip ? "ds_gemm_kernel<true>" : "ds_gemm_kernel<false>");
        const int ntiles = ((NJB + 7) / 8) * ((NIB + 7) / 8) * 64;
        const int gx = guard ? min(ntiles, 768) : ntiles;
        if (recip)
            hipLaunchKernelGGL(ds_gemm_kernel<true>, dim3(gx, B), dim3(256), gemm_lds, s, feat0, feat1, mask0, mask1, sim_ws,
                               w, L, S, C, sqrtC, 1.0f / sqrtC, temperature, 1.0f / temperature, NJB, NIB, guard, ntiles, store);
        else
            hipLaunchKernelGGL(ds_gemm_kernel<false>, dim3(gx, B), dim3(256), gemm_lds, s, feat0, feat1, mask0, mask1, sim_ws,
                               w, L, S, C, sqrtC, 1.0f / sqrtC, temperature, 1.0f / temperature, NJB, NIB, guard, ntiles, store);
    }
    CASMTR_CHECK_LAUNCH();
    {
        ProfScope ps(guard ? -1 : CASMTR_PROF_DS_REDUCE, s, "ds_reduce_kernel (rows + columns)");
        const DsReduceSide rr{w.rp_m, w.rp_s, w.rp_a, NJB, L, B * L, w.rmax, w.rsum, next_idx01, next_conf01, nullptr, 0, nullptr, nullptr};
        const DsReduceSide rc{w.cp_m, w.cp_s, w.cp_a, NIB, S, B * S, w.cmax, w.csum, next_idx10, next_conf10, nullptr, 0, nullptr, nullptr};
        hipLaunchKernelGGL(ds_reduce_kernel, dim3((B * L + 255) / 256 + (B * S + 255) / 256), dim3(256), 0, s, rr, rc, 0.f, guard);
    }
    CASMTR_CHECK_LAUNCH();
    if (guard) {   // rbest / cbest (adjacent in the workspace) hold the split pass's (invalid) results
        hipLaunchKernelGGL(ds_zero_kernel, dim3(256), dim3(256), 0, s, w.rbest,
                           (size_t)((w.cbest + (size_t)B * S) - w.rbest), guard);
        CASMTR_CHECK_LAUNCH();
    }
    if (!store) {   // the matrix was not written: pass 2 on the recomputed flagged segments (guard == nullptr here)
        ProfScope ps(CASMTR_PROF_DS_CONF, s, "ds_flagscan_kernel + ds_flagtiles_kernel + ds_eflagged_kernel (flagged segments recomputed, fp32 chain)");
        if (const int r = ds_flag_lists_launch(w, B, L, S, thr, 1, s)) return r;
        constexpr size_t lds = sizeof(float) * (2 * 128 * 33 + 512);
        static int resident_tab[2][CASMTR_MAX_DEVICES] = {{0}, {0}};
        int resident = 0;
        if (recip) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_eflagged_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (const int r = resident_workgroups(resident_tab[1], ds_eflagged_kernel<true>, 256, lds, &resident)) return r;
            hipLaunchKernelGGL(ds_eflagged_kernel<true>, dim3((unsigned)resident), dim3(256), lds, s, feat0, feat1, mask0, mask1, w, B, L, S, C, sqrtC,
                               1.0f / sqrtC, temperature, 1.0f / temperature, thr, NJB);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ds_eflagged_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (const int r = resident_workgroups(resident_tab[0], ds_eflagged_kernel<false>, 256, lds, &resident)) return r;
            hipLaunchKernelGGL(ds_eflagged_kernel<false>, dim3((unsigned)resident), dim3(256), lds, s, feat0, feat1, mask0, mask1, w, B, L, S, C, sqrtC,
                               1.0f / sqrtC, temperature, 1.0f / temperature, thr, NJB);
        }
        CASMTR_CHECK_LAUNCH();
        return 0;
    }
    {
        ProfScope ps(guard ? -1 : CASMTR_PROF_DS_CONF, s, "ds_conf_kernel<false>");
        hipLaunchKernelGGL(ds_conf_kernel<false>, dim3((S + 1023) / 1024, (L + DSC_ROWS - 1) / DSC_ROWS, B), dim3(256), 0, s, sim_ws,
                           w, L, S, want_conf, thr, guard, 0.f);
    }
    CASMTR_CHECK_LAUNCH();
    return 0;
}

static int ds_select(const DsWs& w, float thr, int border_rm, const int32_t* valid_hw, int h0c, int w0c, int h1c, int w1c,
                     int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches, int B, int L, int S,
                     hipStream_t s) {
    ProfScope ps(CASMTR_PROF_DS_SELECT, s, "ds_flag_kernel + compaction (count / scan / write)");
    hipLaunchKernelGGL(ds_flag_kernel, dim3((B * L + 255) / 256), dim3(256), 0, s, w, thr, border_rm, valid_hw, h0c, w0c,
                       h1c, w1c, L, S, B * L);
    CASMTR_CHECK_LAUNCH();
    return run_compaction(w.flags, w.jsel, w.csel, w.blk, B * L, L, 0, B, b_ids, i_ids, j_ids, mconf, n_matches, s);
}

extern "C" int casmtr_dual_softmax_fwd(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                                       float temperature, int recip, float thr, int border_rm, const int32_t* valid_hw,
                                       int h0c, int w0c, int h1c, int w1c, int want_conf, float* sim_ws, void* stats_ws,
                                       int64_t* next_idx01, float* next_conf01, int64_t* next_idx10, float* next_conf10,
                                       int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches,
                                       int B, int L, int S, int C, casmtr_stream_t stream) {
    if (C % DS_BK != 0 || (mask0 == nullptr) != (mask1 == nullptr)) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0 || S <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DsWs w{};
    ds_carve(&w, reinterpret_cast<char*>(stats_ws), B, L, S, 0);
    // a kernel, not hipMemsetAsync: inside a captured HIP graph (casmtr_amd/graph.py) a memset node is not reliably ordered against
    // the kernels around it on this ROCm stack (replays faulted after tens of steps: counters read before they were cleared)
    hipLaunchKernelGGL(ds_zero_kernel, dim3(256), dim3(256), 0, s, w.rbest,
                       (size_t)(w.zero_end - reinterpret_cast<char*>(w.rbest)) / sizeof(unsigned long long), (const int*)nullptr);
    CASMTR_CHECK_LAUNCH();
    // want_conf: 0 = neither conf_matrix nor the similarity matrix is written (round 6: pass 2 recomputes the segments that can hold a
    // match; thr < 1e-3 keeps the streaming pass, whose log(thr rsum) test it replaces), 1 = conf_matrix in sim_ws, 2 = the similarity
    // matrix stays in sim_ws (tests; the python layer's `want_sim`)
    const int store = want_conf != 0 || thr < 1e-3f;
    int rc = ds_exact_passes(feat0, feat1, mask0, mask1, temperature, recip, thr, want_conf == 1, sim_ws, w, next_idx01, next_conf01,
                             next_idx10, next_conf10, B, L, S, C, nullptr, s, store);
    if (rc) return rc;
    return ds_select(w, thr, border_rm, valid_hw, h0c, w0c, h1c, w1c, b_ids, i_ids, j_ids, mconf, n_matches, B, L, S, s);
}

// Same contract, stats_ws sized by casmtr_dual_softmax_split_ws_bytes.  The similarity matrix comes from the f16 matrix pipe
// (16x the fp32 MFMA rate) as three products of a two-term f16 split of the row-normalised operands: error below 2^-15 |a||b|/(C T)
// including the fp32 chain's own rounding, so the statistics (sums, confidences) agree with the exact path far inside the 1e-4
// softmax tolerance.  Everything that decides an INDEX is then re-decided exactly: each entry within twice that bound of its row /
// column maximum is a candidate, and rows / columns with more than one candidate recompute those logits with the oracle's fp32 fmaf
// chain and take the first maximum (ds_split.hip).  More than DS_CAND_CAP candidates anywhere (degenerate inputs: duplicated or
// all-zero feature rows) -> the exact passes run after all, behind a device-side flag: no host synchronisation either way.
extern "C" int casmtr_dual_softmax_split_fwd(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1,
                                             float temperature, int recip, float thr, int border_rm, const int32_t* valid_hw,
                                             int h0c, int w0c, int h1c, int w1c, int want_conf, float* sim_ws, void* stats_ws,
                                             int64_t* next_idx01, float* next_conf01, int64_t* next_idx10, float* next_conf10,
                                             int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches,
                                             int B, int L, int S, int C, casmtr_stream_t stream) {
    if (C % DS_BK != 0 || (mask0 == nullptr) != (mask1 == nullptr)) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0 || S <= 0) return 0;
    if (C > 256 || (C & 63) || L > 32768 || S > 32768)   // the exact re-decision stages a feature row of <= 256 channels in LDS and <= 256 block partials per
                   // line (every shipped config: C = 256, 10 816 tokens): wider features / larger grids take
                   // the all-fp32 path, whose workspace is a prefix of this one
        return casmtr_dual_softmax_fwd(feat0, feat1, mask0, mask1, temperature, recip, thr, border_rm, valid_hw, h0c, w0c, h1c, w1c, want_conf == 1,
                                       sim_ws, stats_ws, next_idx01, next_conf01, next_idx10, next_conf10, b_ids, i_ids, j_ids, mconf, n_matches,
                                       B, L, S, C, stream);
    hipStream_t s = (hipStream_t)stream;
    DsWs w{};
    ds_carve(&w, reinterpret_cast<char*>(stats_ws), B, L, S, C);
    const int NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    // a kernel, not hipMemsetAsync: inside a captured HIP graph (casmtr_amd/graph.py) a memset node is not reliably ordered against
    // the kernels around it on this ROCm stack (replays faulted after tens of steps: counters read before they were cleared)
    hipLaunchKernelGGL(ds_zero_kernel, dim3(256), dim3(256), 0, s, w.rbest,
                       (size_t)(w.zero_end - reinterpret_cast<char*>(w.rbest)) / sizeof(unsigned long long), (const int*)nullptr);
    CASMTR_CHECK_LAUNCH();
    int rc;
    const bool dense = want_conf == 1 || thr < 1e-3f;   // pass 2 streams the stored matrix (needed to write conf_matrix; log(thr rsum) undefined)
    {
        ProfScope ps(CASMTR_PROF_DS_SPLIT, s, "ds_prep_kernel (row exponents, norms, f16 split images of both operands)");
        rc = ds_split_launch(feat0, feat1, mask0, mask1, w, B, L, S, C, temperature, recip, s);
    }
    if (rc) return rc;
    {
        // the matrix itself is only written where something reads it: conf_matrix output / the dense pass 2 (want_conf == 1, thr < 1e-3)
        // or a caller that asked for the similarity matrix (want_conf == 2: tests)
        rc = ds_gemm16_launch(mask0, mask1, sim_ws, w, B, L, S, C, dense || want_conf == 2, s);   // timed inside (events attached to the dispatch)
    }
    if (rc) return rc;
    const float kthr = 6.103515625e-05f / temperature;   // 2 e = 2^-14 |a_i|/sqrtC max|b_j|/sqrtC / T
    {
        ProfScope ps(CASMTR_PROF_DS_REDUCE, s, "ds_reduce_kernel (rows + columns)");
        const DsReduceSide rr{w.rp_m, w.rp_s, nullptr, NJB, L, B * L, w.rmax, w.rsum, next_idx01, next_conf01, w.na, NIB * DS_BM, w.nbmax, w.rthr};
        const DsReduceSide rc{w.cp_m, w.cp_s, nullptr, NIB, S, B * S, w.cmax, w.csum, next_idx10, next_conf10, w.nb, NJB * DS_BN, w.namax, w.cthr};
        hipLaunchKernelGGL(ds_reduce_kernel, dim3((B * L + 255) / 256 + (B * S + 255) / 256), dim3(256), 0, s, rr, rc, kthr, nullptr);
        CASMTR_CHECK_LAUNCH();
    }
    {
        ProfScope ps(CASMTR_PROF_DS_CONF, s, !dense ? "ds_flagscan_kernel + ds_flagtiles_kernel + ds_flagged_kernel (flagged segments recomputed)" : "ds_conf_kernel<true>");
        if (!dense) {   // pass 2 on the flagged segments only, recomputed from the operand images (no matrix in memory)
            rc = ds_flagged_launch(feat0, feat1, w, B, L, S, C, thr, kthr, s);
            if (rc) return rc;
        } else
            hipLaunchKernelGGL(ds_conf_kernel<true>, dim3((S + 1023) / 1024, (L + DSC_ROWS - 1) / DSC_ROWS, B), dim3(256), 0, s, sim_ws,
                               w, L, S, want_conf == 1, thr, nullptr, kthr);
    }
    CASMTR_CHECK_LAUNCH();
    {
        ProfScope ps(CASMTR_PROF_DS_FIX, s, "ds_fix_kernel + guarded exact-pass launches (exit at once unless a candidate list overflowed)");
        rc = ds_fix_launch(feat0, feat1, w, B, L, S, C, temperature, recip, next_idx01, next_idx10, s);
        if (rc) return rc;
        // match list exact by construction: every entry whose approximate confidence cannot be ordered for certain against thr, its
        // row's or its column's runner-up gets its logit, its row's and its column's softmax statistics from the exact chain, in
        // the exact kernels' own summation order, and the rows concerned are decided from those values (ds_split.hip)
        rc = ds_xdecide_launch(feat0, feat1, mask0, mask1, w, B, L, S, C, temperature, recip, thr, next_conf01, next_conf10, s);
        if (rc) return rc;
        // a list overflowed: the exact passes below decide everything; the re-decisions made from the (truncated) lists are dropped
        hipLaunchKernelGGL(ds_zero_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(w.rdec),
                           ((size_t)B * L + 7) / 8, (const int*)w.ovf);
        rc = ds_exact_passes(feat0, feat1, mask0, mask1, temperature, recip, thr, want_conf == 1, sim_ws, w, next_idx01, next_conf01,
                             next_idx10, next_conf10, B, L, S, C, w.ovf, s);
    }
    if (rc) return rc;
    return ds_select(w, thr, border_rm, valid_hw, h0c, w0c, h1c, w1c, b_ids, i_ids, j_ids, mconf, n_matches, B, L, S, s);
}

// =================================================================================================== window match
// One wave per query token, 4 tokens per workgroup, no block-level synchronisation.  The C channels are walked in
// chunks of 32 (one 128-B line per candidate row): each chunk of the K candidate rows goes through the wave's slab
// (coalesced loads, wave_rows32_to_lanes), every lane normalises its own row chunk and extends its fmaf chain
// (c ascending across chunks, so the result is the reference's sequential chain).  The query chunk comes in through
// wave-uniform scalar loads.  lane <-> candidates k = lane and 64 + lane (K <= 128).
template <int C, bool RECIP>
__global__ __launch_bounds__(256) void window_match_kernel(const float* __restrict__ fq, const float* __restrict__ fk,
                                                           const int64_t* __restrict__ idx,
                                                           const uint8_t* __restrict__ mq, const uint8_t* __restrict__ mk,
                                                           float sqrtC, float inv_sqrtC, float T, float invT,
                                                           float* __restrict__ conf, float* __restrict__ next_conf,
                                                           int64_t* __restrict__ next_idx, int N, int M, int K,
                                                           int nblocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* slab = smem + wave * (CASMTR_SLAB_FLOATS + 128);
    int* cand = reinterpret_cast<int*>(slab + CASMTR_SLAB_FLOATS);  // [128]
    const int b = blockIdx.y;
    const int n = xcd_chunk_remap(blockIdx.x, nblocks) * 4 + wave;   // neighbouring tokens (overlapping windows) share an L2
    if (n >= N) return;
    const int64_t* ip = idx + ((size_t)b * N + n) * K;
    const int c0 = lane < K ? (int)ip[lane] : 0;
    const int c1 = 64 + lane < K ? (int)ip[64 + lane] : 0;
    cand[lane] = c0;
    cand[64 + lane] = c1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const cfloat_p qp = as_const(fq + ((size_t)b * N + n) * C);  // wave-uniform -> scalar loads
    const float* kb = fk + (size_t)b * M * C;
    float acc[2] = {0.f, 0.f};
    // (a register double-buffered version of this loop -- loads of step s+1 issued before step s is consumed -- was
    //  measured 3x SLOWER: it spills at the 4-waves/SIMD budget and occupancy, not prefetch depth, is what hides latency here)
#pragma unroll
    for (int ch = 0; ch < C / 32; ++ch) {
        float qn[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) qn[i] = div_scalar<RECIP>(qp[ch * 32 + i], sqrtC, inv_sqrtC);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p * 64 < K) {  // wave-uniform
                f32x4 kr[8];
                wave_rows32_to_lanes(slab, lane, [&](int r) { return kb + (size_t)cand[min(p * 64 + r, K - 1)] * C + ch * 32; }, kr);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[p] = __builtin_fmaf(qn[4 * i + 0], div_scalar<RECIP>(kr[i].x, sqrtC, inv_sqrtC), acc[p]);
                    acc[p] = __builtin_fmaf(qn[4 * i + 1], div_scalar<RECIP>(kr[i].y, sqrtC, inv_sqrtC), acc[p]);
                    acc[p] = __builtin_fmaf(qn[4 * i + 2], div_scalar<RECIP>(kr[i].z, sqrtC, inv_sqrtC), acc[p]);
                    acc[p] = __builtin_fmaf(qn[4 * i + 3], div_scalar<RECIP>(kr[i].w, sqrtC, inv_sqrtC), acc[p]);
                }
            }
        }
    }
    const int mqv = mq ? mq[(size_t)b * N + n] : 1;
    float x[2];
    unsigned key[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int k = p * 64 + lane;
        x[p] = 0.f; key[p] = 0u;
        if (k < K) {
            float v = div_scalar<RECIP>(acc[p], T, invT);
            if (mq && !(mqv && mk[(size_t)b * M + (p ? c1 : c0)])) v = NEG_FILL;
            x[p] = v; key[p] = f2ord(v);
        }
    }
    const unsigned wm = wave_max_u32(max(key[0], key[1]));
    const float m = ord2f(wm);
    float e0, e1;
    window_softmax2(x[0], x[1], m, lane < K, 64 + lane < K, e0, e1);
    if (conf) {
        if (lane < K) conf[((size_t)b * N + n) * K + lane] = e0;
        if (64 + lane < K) conf[((size_t)b * N + n) * K + 64 + lane] = e1;
    }
    const unsigned long long b0 = __ballot(key[0] == wm && lane < K);
    const unsigned long long b1 = __ballot(key[1] == wm && 64 + lane < K);
    const int am = b0 ? (__ffsll((long long)b0) - 1) : (64 + __ffsll((long long)b1) - 1);
    if (lane == (am & 63)) {
        next_conf[(size_t)b * N + n] = am < 64 ? e0 : e1;
        next_idx[(size_t)b * N + n] = am < 64 ? c0 : c1;
    }
}

// Quad variant: one WAVE per quad of query tokens (the 4 children of a coarse cell), 4 quads per workgroup, no block-level
// synchronisation.  CascadeQTAttB hands every child the same window list (modules/quadtree_attention.py:450): lane <->
// candidates k = lane and 64 + lane; each lane reads its candidate row's 32-channel chunk straight into registers
// (8 x dwordx4), normalises it once and extends the chains of all 4 children with it, so a key row is fetched and normalised
// once per quad.  The kernel verifies that the 4 index rows really are identical; if not, each child walks its own rows
// (same results).  The 4 normalised queries sit in 2 KB of wave-private LDS and come back as broadcast reads.  Same
// arithmetic as above: operands pre-scaled, c-ascending fmaf chain.  (An LDS-staged version -- key tile staged 32 channels at
// a time, double buffered, wave <-> child -- needed 8 block barriers per quad and ran at 0.82 ms per launch against 0.56.)
template <int C, bool RECIP>
__global__ __launch_bounds__(256) void window_match_quad_kernel(const float* __restrict__ fq, const float* __restrict__ fk,
                                                                const int64_t* __restrict__ idx,
                                                                const uint8_t* __restrict__ mq, const uint8_t* __restrict__ mk,
                                                                float sqrtC, float inv_sqrtC, float T, float invT,
                                                                float* __restrict__ conf, float* __restrict__ next_conf,
                                                                int64_t* __restrict__ next_idx, int N, int M, int K, int h,
                                                                int w, int nquads, const int64_t* __restrict__ topk_pos,
                                                                int w1, int dil) {
    // topk_pos != nullptr: implicit windows (casmtr_window_match_pos_fwd) -- idx is not read; the candidate list of the quad is
    // expanded from topk_pos [B,nquads,K/4,2] exactly as CascadeQTAttB does (modules/quadtree_attention.py:419-450)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* qn = smem + wave * 4 * C;                              // [4 children][C] normalised queries
    const int b = blockIdx.y;
    const int quad = xcd_chunk_remap(blockIdx.x, gridDim.x) * 4 + wave;   // neighbouring quads (overlapping windows) share an L2
    if (quad >= nquads) return;
    const int wq = w >> 1, qy = quad / wq, qx = quad % wq;
    int tok[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) tok[f] = (2 * qy + (f >> 1)) * w + 2 * qx + (f & 1);
#pragma unroll
    for (int f = 0; f < 4; ++f)
        for (int c = lane; c < C; c += 64) qn[f * C + c] = div_scalar<RECIP>(fq[((size_t)b * N + tok[f]) * C + c], sqrtC, inv_sqrtC);
    int ci[2][4];
    bool lsame = true;
    if (topk_pos) {
        const int64_t* pos = topk_pos + ((size_t)b * nquads + quad) * (K >> 2) * 2;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = p * 64 + lane;
            int id = 0;
            if (k < K) {
                const int e = k >> 2, t = k & 3;
                long long v = (pos[2 * e] * 2 + (t >> 1) * dil) * w1 + pos[2 * e + 1] * 2 + (t & 1) * dil;
                v = v < 0 ? 0 : (v > (long long)M - 1 ? (long long)M - 1 : v);   // torch.clamp, :429
                id = (int)v;
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) ci[p][f] = id;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int k = p * 64 + lane;
                ci[p][f] = k < K ? (int)idx[((size_t)b * N + tok[f]) * K + k] : 0;
                lsame = lsame && ci[p][f] == ci[p][0];
            }
    }
    const bool same = __ballot(!lsame) == 0ull;                  // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* kb = fk + (size_t)b * M * C;
    float acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[p][f] = 0.f;
    const int rounds = same ? 1 : 4;
    for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
        for (int ch = 0; ch < C / 32; ++ch) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p * 64 < K) {  // wave-uniform
                    int row = ci[p][0];
                    if (!same) row = rd == 1 ? ci[p][1] : (rd == 2 ? ci[p][2] : (rd == 3 ? ci[p][3] : row));
                    const f32x4* kp = reinterpret_cast<const f32x4*>(kb + (size_t)row * C + ch * 32);
                    f32x4 kr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) kr[i] = kp[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        kr[i].x = div_scalar<RECIP>(kr[i].x, sqrtC, inv_sqrtC); kr[i].y = div_scalar<RECIP>(kr[i].y, sqrtC, inv_sqrtC);
                        kr[i].z = div_scalar<RECIP>(kr[i].z, sqrtC, inv_sqrtC); kr[i].w = div_scalar<RECIP>(kr[i].w, sqrtC, inv_sqrtC);
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if (same || f == rd) {
                            const f32x4* qp = reinterpret_cast<const f32x4*>(qn + f * C + ch * 32);   // broadcast reads
                            float a = acc[p][f];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const f32x4 qv = qp[i];
                                a = __builtin_fmaf(qv.x, kr[i].x, a);
                                a = __builtin_fmaf(qv.y, kr[i].y, a);
                                a = __builtin_fmaf(qv.z, kr[i].z, a);
                                a = __builtin_fmaf(qv.w, kr[i].w, a);
                            }
                            acc[p][f] = a;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int n = tok[f];
        const int c0 = ci[0][f], c1 = ci[1][f];
        const int mqv = mq ? mq[(size_t)b * N + n] : 1;
        float x[2];
        unsigned key[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = p * 64 + lane;
            x[p] = 0.f; key[p] = 0u;
            if (k < K) {
                float v = div_scalar<RECIP>(acc[p][f], T, invT);
                if (mq && !(mqv && mk[(size_t)b * M + (p ? c1 : c0)])) v = NEG_FILL;
                x[p] = v; key[p] = f2ord(v);
            }
        }
        const unsigned wm = wave_max_u32(max(key[0], key[1]));
        const float m = ord2f(wm);
        float e0, e1;
        window_softmax2(x[0], x[1], m, lane < K, 64 + lane < K, e0, e1);
        if (conf) {
            if (lane < K) conf[((size_t)b * N + n) * K + lane] = e0;
            if (64 + lane < K) conf[((size_t)b * N + n) * K + 64 + lane] = e1;
        }
        const unsigned long long b0 = __ballot(key[0] == wm && lane < K);
        const unsigned long long b1 = __ballot(key[1] == wm && 64 + lane < K);
        const int am = b0 ? (__ffsll((long long)b0) - 1) : (64 + __ffsll((long long)b1) - 1);
        if (lane == (am & 63)) {
            next_conf[(size_t)b * N + n] = am < 64 ? e0 : e1;
            next_idx[(size_t)b * N + n] = am < 64 ? c0 : c1;
        }
    }
}

template <int C, bool RECIP>
static int launch_window_match_r(const float* fq, const float* fk, const int64_t* idx, const uint8_t* mq, const uint8_t* mk,
                                 float T, float* conf, float* next_conf, int64_t* next_idx, int B, int N, int M, int K, int h,
                                 int w, hipStream_t s) {
    const float sqrtC = (float)sqrt((double)C);
    ProfScope ps(CASMTR_PROF_WINDOW_MATCH, s, "window_match_quad_kernel / window_match_kernel (explicit index list)");
    if (h > 0 && w > 0 && (h % 2 == 0) && (w % 2 == 0) && h * w == N) {
        const int nquads = (h / 2) * (w / 2);
        hipLaunchKernelGGL((window_match_quad_kernel<C, RECIP>), dim3((nquads + 3) / 4, B), dim3(256), sizeof(float) * 16 * C, s, fq,
                           fk, idx, mq, mk, sqrtC, 1.0f / sqrtC, T, 1.0f / T, conf, next_conf, next_idx, N, M, K, h, w, nquads,
                           (const int64_t*)nullptr, 0, 1);
    } else {
        const size_t lds = sizeof(float) * 4 * (CASMTR_SLAB_FLOATS + 128);
        const int nblocks = (N + 3) / 4;
        hipLaunchKernelGGL((window_match_kernel<C, RECIP>), dim3(nblocks, B), dim3(256), lds, s, fq, fk, idx, mq, mk, sqrtC,
                           1.0f / sqrtC, T, 1.0f / T, conf, next_conf, next_idx, N, M, K, nblocks);
    }
    CASMTR_CHECK_LAUNCH();
    return 0;
}

template <int C>
static int launch_window_match(const float* fq, const float* fk, const int64_t* idx, const uint8_t* mq, const uint8_t* mk,
                               float T, int recip, float* conf, float* next_conf, int64_t* next_idx, int B, int N, int M,
                               int K, int h, int w, hipStream_t s) {
    return recip ? launch_window_match_r<C, true>(fq, fk, idx, mq, mk, T, conf, next_conf, next_idx, B, N, M, K, h, w, s)
                 : launch_window_match_r<C, false>(fq, fk, idx, mq, mk, T, conf, next_conf, next_idx, B, N, M, K, h, w, s);
}

extern "C" int casmtr_window_match_fwd(const float* feat_q, const float* feat_k, const int64_t* idx, const uint8_t* mask_q,
                                       const uint8_t* mask_k, float temperature, int recip, float* conf, float* next_conf,
                                       int64_t* next_idx, int B, int N, int M, int K, int C, int h, int w,
                                       casmtr_stream_t stream) {
    if (K > 128 || K <= 0 || (mask_q == nullptr) != (mask_k == nullptr)) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || N <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (C == 256) return launch_window_match<256>(feat_q, feat_k, idx, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx, B, N, M, K, h, w, s);
    if (C == 128) return launch_window_match<128>(feat_q, feat_k, idx, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx, B, N, M, K, h, w, s);
    if (C == 64) return launch_window_match<64>(feat_q, feat_k, idx, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx, B, N, M, K, h, w, s);
    if (C == 32) return launch_window_match<32>(feat_q, feat_k, idx, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx, B, N, M, K, h, w, s);
    return CASMTR_ERR_UNSUPPORTED;
}

// implicit windows on the wave-per-quad kernel (window_dma.hip dispatches here by default)
template <int C, bool RECIP>
static int launch_wm_quad_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                              float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW,
                              int dil, hipStream_t s) {
    const float sqrtC = (float)sqrt((double)C);
    const int nquads = (h0 / 2) * (w0 / 2);
    ProfScope ps(CASMTR_PROF_WINDOW_MATCH, s, "window_match_quad_kernel (implicit windows)");
    hipLaunchKernelGGL((window_match_quad_kernel<C, RECIP>), dim3((nquads + 3) / 4, B), dim3(256), sizeof(float) * 16 * C, s, fq, fk,
                       (const int64_t*)nullptr, mq, mk, sqrtC, 1.0f / sqrtC, T, 1.0f / T, conf, next_conf, next_idx, h0 * w0, h1 * w1,
                       4 * KW, h0, w0, nquads, tp, w1, dil);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

int casmtr_window_match_quad_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                                 int recip, float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1,
                                 int KW, int C, int dil, hipStream_t s) {
#define WMQ(CC)                                                                                                                    \
    if (C == CC)                                                                                                                   \
        return recip ? launch_wm_quad_pos<CC, true>(fq, fk, tp, mq, mk, T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, s) \
                     : launch_wm_quad_pos<CC, false>(fq, fk, tp, mq, mk, T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, s);
    WMQ(256) WMQ(128) WMQ(64) WMQ(32)
#undef WMQ
    return CASMTR_ERR_UNSUPPORTED;
}

// =================================================================================================== NMS + selection
__global__ __launch_bounds__(256) void nms_flag_kernel(const float* __restrict__ conf, const int64_t* __restrict__ idx01,
                                                       const int64_t* __restrict__ idx10, int nms_window, float test_thr,
                                                       const float* __restrict__ pre0, int hp0, int wp0, float pt0,
                                                       const float* __restrict__ pre1, int hp1, int wp1, float pt1,
                                                       int border_rm, const int32_t* __restrict__ valid_hw,
                                                       int double_check, unsigned char* __restrict__ keep, int B, int H0,
                                                       int W0, int H1, int W1, const unsigned char* __restrict__ extra_keep) {
    const int N = H0 * W0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N) return;
    const int b = t / N, i = t % N, y = i / W0, x = i % W0;
    const float* cf = conf + (size_t)b * N;
    const float me = cf[i];
    bool k = true;
    if (nms_window > 0) {
        // F.max_pool2d(return_indices=True): first maximum in row-major window scan wins (strict >), NaN propagates
        const int r = nms_window / 2;
        float best = 0.f; int bi = -1;
        for (int yy = max(y - r, 0); yy <= min(y + r, H0 - 1); ++yy)
            for (int xx = max(x - r, 0); xx <= min(x + r, W0 - 1); ++xx) {
                const float v = cf[yy * W0 + xx];
                if (bi < 0 || v > best || v != v) { best = v; bi = yy * W0 + xx; }
            }
        k = (bi == i);
    }
    if (!(me > test_thr)) k = false;
    if (extra_keep && !extra_keep[t]) k = false;   // PostProcess methods other than None / maxpool_nms, post_config rt / rd
    if (pre0) {
        int sy = (int)floorf((float)y * ((float)hp0 / (float)H0)); sy = min(sy, hp0 - 1);
        int sx = (int)floorf((float)x * ((float)wp0 / (float)W0)); sx = min(sx, wp0 - 1);
        if (pre0[(size_t)b * hp0 * wp0 + sy * wp0 + sx] <= pt0) k = false;
    }
    if (pre1) {
        int sy = (int)floorf((float)y * ((float)hp1 / (float)H0)); sy = min(sy, hp1 - 1);
        int sx = (int)floorf((float)x * ((float)wp1 / (float)W0)); sx = min(sx, wp1 - 1);
        if (pre1[(size_t)b * hp1 * wp1 + sy * wp1 + sx] <= pt1) k = false;
    }
    const int64_t j = idx01[t];
    if (border_rm > 0) {
        const int vh0 = valid_hw ? valid_hw[b * 4 + 0] : H0, vw0 = valid_hw ? valid_hw[b * 4 + 1] : W0;
        const int vh1 = valid_hw ? valid_hw[b * 4 + 2] : H1, vw1 = valid_hw ? valid_hw[b * 4 + 3] : W1;
        if (y < border_rm || x < border_rm || y >= vh0 - border_rm || x >= vw0 - border_rm) k = false;
        const int ty = (int)(j / W1), tx = (int)(j % W1);
        if (tx < border_rm || tx > vw1 - border_rm || ty < border_rm || ty > vh1 - border_rm) k = false;  // :137-138
    }
    if (double_check && idx10[(size_t)b * H1 * W1 + j] != i) k = false;
    keep[t] = k ? 1 : 0;
}

extern "C" size_t casmtr_nms_select_ws_bytes(int B, int H0, int W0) {
    const size_t total = (size_t)B * H0 * W0;
    return align256(total) + align256(sizeof(int) * ((total + 1023) / 1024 + 8));
}

extern "C" int casmtr_nms_select_fwd(const float* next_conf01, const int64_t* next_idx01, const int64_t* next_idx10,
                                     int nms_window, float test_thr, const float* pre_conf0, int hp0, int wp0,
                                     float pre_thr0, const float* pre_conf1, int hp1, int wp1, float pre_thr1,
                                     int border_rm, const int32_t* valid_hw, int double_check, void* ws, int64_t* b_ids,
                                     int64_t* i_ids, int64_t* j_ids, float* mconf, int64_t* n_matches, int B, int H0,
                                     int W0, int H1, int W1, const uint8_t* extra_keep, casmtr_stream_t stream) {
    if (B <= 0 || H0 <= 0 || W0 <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int total = B * H0 * W0;
    unsigned char* keep = reinterpret_cast<unsigned char*>(ws);
    int* blk = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + align256((size_t)total));
    ProfScope ps(CASMTR_PROF_NMS_SELECT, s, "nms_flag_kernel + compaction (count / scan / write)");
    hipLaunchKernelGGL(nms_flag_kernel, dim3((total + 255) / 256), dim3(256), 0, s, next_conf01, next_idx01, next_idx10,
                       nms_window, test_thr, pre_conf0, hp0, wp0, pre_thr0, pre_conf1, hp1, wp1, pre_thr1, border_rm,
                       valid_hw, double_check, keep, B, H0, W0, H1, W1, extra_keep);
    CASMTR_CHECK_LAUNCH();
    return run_compaction(keep, next_idx01, next_conf01, blk, total, H0 * W0, 1, B, b_ids, i_ids, j_ids, mconf, n_matches, s);
}
