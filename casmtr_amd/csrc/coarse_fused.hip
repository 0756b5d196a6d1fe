// QTAttB.process_coarse_level (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:161-178) as ONE kernel:
// dense QK^T -> softmax over the keys -> top-k -> A.V for a 16-row query tile of one (pair, head), the [16][S] logits /
// probabilities tile living in LDS.  The round-1 path (coarse_logits_kernel / coarse_row_kernel / coarse_av_kernel, qta_fused.hip)
// round-trips a [B,H,L,S_pad] workspace through HBM four times per call (122 MB written, rewritten and read twice at 26x26, B = 8).
//   phase A: logits = temp * Q K^T with v_mfma_f32_16x16x4_f32 -- an exact k-ascending fmaf chain (tools/probes/
//            mfma16x16x4_layout.hip: bit-identical to fmaf), so the values the top-k selects on equal the oracle's sequential
//            chain; a wave takes every 4th block of 16 keys, operands straight from global memory (a (pair, head)'s keys and
//            values are 86 KB each and stay in L2 for its 43 tiles);
//   phase B: one wave per row, the row in registers (lane <-> key, 64 apart): softmax, top-k by threshold + 64-lane bitonic
//            sort (the selection of coarse_row_kernel, same total order (logit desc, position asc)), probabilities back to LDS;
//   phase C: message = P V with the same MFMA shape, the key range split over the 4 waves, partial sums folded through LDS;
//            final = message * weight[0] fused into the store (:274).
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

template <int EMAX>
__global__ __launch_bounds__(256) void coarse_fused_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float temp, int topk, float w_level,
                                                           float* __restrict__ message, float* __restrict__ acc_out,
                                                           float* __restrict__ topk_score, int64_t* __restrict__ topk_idx, int L,
                                                           int S, int H, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Spad = (S + 63) / 64 * 64, RS = Spad + 2;   // row stride = 2 mod 32: the phase-C reads (lane <-> (row, k)) are conflict-free
    float* Sl = smem;                                      // [16][RS] logits, then probabilities; later the partial sums
    const int tile = max(16 * RS, 4 * 16 * 32);            // the tile doubles as the [4][16][32] reduction buffer of phase C
    unsigned* cbuf = reinterpret_cast<unsigned*>(smem + tile);   // [4 waves][2][64] top-k compaction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, h = bh % H, HD = H * 32;
    const int l0 = blockIdx.x * 16;
    const int mn = lane & 15, kk = lane >> 4;              // MFMA operand coordinates: row / column within the block, k within the step
    const float* qb = q + (size_t)b * L * HD + h * 32;
    const float* kb = k + (size_t)b * S * HD + h * 32;
    const float* vb = v + (size_t)b * S * HD + h * 32;

    // ---- phase A
    {
        float qa[8];
        const float* qr = qb + (size_t)min(l0 + mn, L - 1) * HD + kk;
#pragma unroll
        for (int j = 0; j < 8; ++j) qa[j] = qr[4 * j];
        // software pipeline: the next block's key operands are fetched before the current block's dependent MFMA chain runs
        const int nblk = Spad / 16;
        float kv[8], kn[8];
        auto fetch = [&](int blk, float (&dst)[8]) {
            const float* kr = kb + (size_t)min(blk * 16 + mn, S - 1) * HD + kk;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = kr[4 * j];
        };
        if (wave < nblk) fetch(wave, kv);
        for (int blk = wave; blk < nblk; blk += 4) {
            if (blk + 4 < nblk) fetch(blk + 4, kn);
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[j], kv[j], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Sl[(4 * kk + r) * RS + blk * 16 + mn] = temp * acc[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) kv[j] = kn[j];
        }
    }
    __syncthreads();
    if (dbg & 1) return;

    // ---- phase B: rows wave, wave + 4, ...
    for (int rr = wave; rr < ((dbg & 2) ? 0 : 16); rr += 4) {
        const int l = l0 + rr;
        if (l >= L) break;   // wave-uniform
        float* row = Sl + rr * RS;
        float lv[EMAX];
        unsigned key[EMAX];
        unsigned lm = 0;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            const int kx = e * 64 + lane;
            lv[e] = (kx < S) ? row[kx] : 0.f;
            key[e] = (kx < S) ? f2ord(lv[e]) : 0u;
            lm = max(lm, key[e]);
        }
        const float m = ord2f(wave_max_u32(lm));
        float ps[EMAX];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            ps[e] = (e * 64 + lane < S) ? expf(lv[e] - m) : 0.f;
            s += ps[e];
        }
        s = wave_sum_f32(s);
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            ps[e] = ps[e] / s;
            if (e * 64 + lane < Spad) row[e * 64 + lane] = ps[e];  // zero in the [S,Spad) padding
        }
        // top-k: (logit desc, position asc) is a total order, so the answer is the sorted prefix of ANY superset of the k best.
        // theta := the k-th largest of the 64 per-lane maxima; the elements >= theta (at most 64 unless the row is pathologically
        // concentrated in a few lanes) are compacted in position order, sorted across the wave, the first k lanes store.
        bool done = false;
        if (topk <= 64) {
            unsigned cur = 0;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) cur = max(cur, key[e]);
                unsigned srt = cur;
            static_for<1, 7>([&](auto kq_) {   // 64-lane bitonic sort, descending: 21 compare-exchange steps
                constexpr int kq = 1 << decltype(kq_)::value;
                static_for<0, decltype(kq_)::value>([&](auto j_) {
                    constexpr int j = kq >> (1 + decltype(j_)::value);
                    const unsigned o = wave_xor_u32<j>(srt);
                    const bool keep_max = ((lane & kq) == 0) == ((lane & j) == 0);
                    srt = keep_max ? max(srt, o) : min(srt, o);
                });
            });
            const unsigned theta = (unsigned)__builtin_amdgcn_readlane((int)srt, topk - 1);
            int cnt = 0;
            if (theta != 0u) {
#pragma unroll
                for (int e = 0; e < EMAX; ++e) cnt += __popcll(__ballot(key[e] >= theta));
            }
            if (theta != 0u && cnt <= 64) {   // wave-uniform
                unsigned* ck = cbuf + wave * 128;
                unsigned* cp = ck + 64;
                int base = 0;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    const bool f = key[e] >= theta;
                    const unsigned long long bal = __ballot(f);
                    if (f) {
                        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                        ck[slot] = key[e];
                        cp[slot] = (unsigned)(e * 64 + lane);
                    }
                    base += __popcll(bal);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                unsigned sk = lane < cnt ? ck[lane] : 0u;
                unsigned sp = lane < cnt ? cp[lane] : 0xFFFFFFFFu;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                static_for<1, 7>([&](auto kq_) {   // bitonic sort on (key desc, position asc)
                    constexpr int kq = 1 << decltype(kq_)::value;
                    static_for<0, decltype(kq_)::value>([&](auto j_) {
                        constexpr int j = kq >> (1 + decltype(j_)::value);
                        const unsigned ok = wave_xor_u32<j>(sk);
                        const unsigned op = wave_xor_u32<j>(sp);
                        const bool other_first = ok > sk || (ok == sk && op < sp);   // does the partner precede me in the order?
                        const bool want_first = ((lane & kq) == 0) == ((lane & j) == 0);
                        const bool take = want_first == other_first;
                        sk = take ? ok : sk;
                        sp = take ? op : sp;
                    });
                });
                if (lane < topk) {
                    const size_t o = (((size_t)b * L + l) * topk + lane) * H + h;
                    topk_idx[o] = sp;
                    topk_score[o] = expf(ord2f(sk) - m) / s;
                }
                done = true;
            }
        }
        if (!done) {   // iterative wave argmax (ties beyond 64 survivors, k > 64)
            for (int t = 0; t < topk; ++t) {
                unsigned cur = 0;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) cur = max(cur, key[e]);
                const unsigned wm = wave_max_u32(cur);
                bool found = false;  // wave-uniform
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    if (!found) {
                        const unsigned long long bal = __ballot(key[e] == wm);
                        if (bal) {
                            found = true;
                            const int src = __ffsll((long long)bal) - 1;  // smallest lane at the smallest e -> smallest position
                            if (lane == src) {
                                const size_t o = (((size_t)b * L + l) * topk + t) * H + h;
                                topk_idx[o] = e * 64 + src;
                                topk_score[o] = ps[e];
                                key[e] = 0u;
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (dbg & 4) return;

    // ---- phase C: wave w takes the k-steps w, w+4, ... (4 keys each) for both 16-column halves of D
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    {
        // rounds of U k-steps (4 keys each): the value operands of round r+1 are fetched while the MFMAs of round r run
        constexpr int U = 6;
        const float* pr = Sl + mn * RS + kk;
        const int nstep = Spad / 4;   // multiple of 16
        float va[U][2], vn[U][2];
        auto fetch = [&](int j0, float (&dst)[U][2]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * u;
                const float* vr = vb + (size_t)min(4 * j + kk, S - 1) * HD + mn;   // P is 0 beyond S (and steps beyond nstep are skipped)
                dst[u][0] = vr[0]; dst[u][1] = vr[16];
            }
        };
        fetch(wave, va);
        for (int j0 = wave; j0 < nstep; j0 += 4 * U) {
            if (j0 + 4 * U < nstep) fetch(j0 + 4 * U, vn);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 4 * u;
                if (j < nstep) {   // wave-uniform
                    const float pa = pr[4 * j];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, va[u][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, va[u][1], acc[1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { va[u][0] = vn[u][0]; va[u][1] = vn[u][1]; }
        }
    }
    __syncthreads();   // every wave is done reading the probabilities: the tile becomes the reduction buffer
    float* red = Sl;   // [4 waves][16 rows][32]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kk + r) * 32 + nb * 16 + mn] = acc[nb][r];
    __syncthreads();
    if (tid < 128) {
        const int rr = tid >> 3, d4 = (tid & 7) * 4, l = l0 + rr;
        if (l < L) {
            f32x4 tot = *reinterpret_cast<const f32x4*>(red + rr * 32 + d4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(red + (w * 16 + rr) * 32 + d4);
                tot.x += p.x; tot.y += p.y; tot.z += p.z; tot.w += p.w;
            }
            const size_t o = ((size_t)b * L + l) * HD + h * 32 + d4;
            if (message) *reinterpret_cast<f32x4*>(message + o) = tot;
            if (acc_out) *reinterpret_cast<f32x4*>(acc_out + o) = (f32x4){tot.x * w_level, tot.y * w_level, tot.z * w_level, tot.w * w_level};
        }
    }
}

// -> CASMTR_ERR_UNSUPPORTED when the shape is outside this kernel (the caller then uses the three-kernel path)
int casmtr_qta_coarse_level_fused(const float* q, const float* k, const float* v, float temp, int topk, float w_level, float* message,
                                  float* acc_out, float* topk_score, int64_t* topk_idx, int B, int L, int S, int H, hipStream_t s) {
    const int Spad = (S + 63) / 64 * 64, E = Spad / 64;
    if (E > 16 || S < 1) return CASMTR_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((16 * (Spad + 2) > 2048 ? 16 * (Spad + 2) : 2048) + 4 * 128);
    const dim3 grid((L + 15) / 16, B * H);
    ProfScope ps(CASMTR_PROF_COARSE_FUSED, s, "coarse_fused_kernel");
#define CF_CASE(EE)                                                                                                                     \
    if (E <= EE) {                                                                                                                      \
        if (lds > 48 * 1024)                                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coarse_fused_kernel<EE>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                                        \
        hipLaunchKernelGGL(coarse_fused_kernel<EE>, grid, dim3(256), lds, s, q, k, v, temp, topk, w_level, message, acc_out, topk_score,  \
                           topk_idx, L, S, H, g_debug_flags);                                                                                          \
        CASMTR_CHECK_LAUNCH();                                                                                                          \
        return 0;                                                                                                                       \
    }
    CF_CASE(4)
    CF_CASE(8)
    CF_CASE(12)
    CF_CASE(16)
#undef CF_CASE
    return CASMTR_ERR_UNSUPPORTED;
}
