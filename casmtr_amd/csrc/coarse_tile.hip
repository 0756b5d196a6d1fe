// QTAttB.process_coarse_level (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:161-178) as ONE kernel,
// round 4: dense QK^T -> softmax over the keys -> top-k -> A.V with the logits BORN in the layout the selection works in.
//
// The round-1 path (coarse_logits / coarse_row / coarse_av, qta_fused.hip) moves a [B,H,L,S_pad] workspace through the memory system
// three times and spends ~800 VALU instructions per row on the selection (two 64-lane bitonic sorts, exact expf + division per
// element); the first fused attempt (round 2, pruned in round 5) kept a [16][S] tile in LDS and paid for it with strided operand loads and one
// row per wave at a time.  Here a workgroup owns 16 query rows of one (pair, head); wave w owns rows 4w .. 4w+3 and never shares them:
//   QK^T   lane <-> key: v_mfma_f32_4x4x1 with A = q[row l%4][d] and B = this lane's key row element d gives every lane the 4 rows'
//          logits of ITS key -- for key block e (64 keys) that is x[e][0..3], i.e. after the pass over all blocks a wave holds 4 whole
//          rows as x[e][f] with key = 64 e + lane: exactly coarse_row_kernel's register layout, with no transposition and no
//          workspace.  The d-sum is the sequential d-ascending fmaf chain of the oracle (tools/probes/mfma4x4_layout.hip), so every
//          index derived from the logits is bit-identical.  Key blocks (64 rows x 128 B) arrive by LDS-DMA through a 4-slot ring
//          shared by the 4 waves (each issues a quarter of a block, one s_barrier per block, prefetch distance 3).
//   top-k  per row, (logit desc, position asc): theta = a threshold with topk .. topk+4 of the 64 per-lane maxima at or above it
//          (scalar bisection on ballots: ~6 rounds of 1 VALU + 6 SALU instead of a 21-step wave sort), the <= 64 elements >= theta are
//          compacted in position order, packed as ((key - theta + 1) << 6) | (63 - slot) -- EXACT, no truncation: the survivors' ordered
//          keys span less than 2^26 -- and sorted with ONE 32-bit 64-lane bitonic network.  Rows whose survivors do not fit
//          (more than 64, or a key range of 2^26 and more) take the exact (key, position) pair sort / the iterated wave argmax.
//   A.V    block <-> key orientation: MFMA block b = lane / 4 takes key 16 t + b, A = p[row l%4][key], B = V[key][8 j + g] (two
//          ds_read_b128 per 8 MFMAs instead of one ds_read_b32 per MFMA), 8 independent accumulators (g), partial sums over b folded
//          at the end (2 DPP steps + 2 KB of LDS).  Probabilities are exp2(x log2e - max log2e) unnormalised, 1 / sum applied to the
//          message (1e-4 tolerance, no index depends on them); value blocks come through the same ring.
// Outputs: message / acc_out = message * weight[0] (:274) token-major, topk_tab [B,H,L,topk] int32 for the finer level (fine_quad.hip),
// on request the reference's topk_score / topk_idx [B,L,topk,H] (:170-175).  Everything is stored at the end of the workgroup, behind
// the last DMA wait (vmcnt counts stores too).
#include <stdio.h>
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct CoarseTArgs {
    const float* q;        // [B,L,H,32]
    const float* k;        // [B,S,H,32]
    const float* v;        // [B,S,H,32]
    float* message;        // nullable [B,L,H,32]
    float* acc_out;        // nullable [B,L,H,32]
    float* topk_score;     // nullable [B,L,topk,H]
    int64_t* topk_idx;     // nullable [B,L,topk,H]
    int32_t* topk_tab;     // nullable [B,H,L,topk]
    float temp, w_level;
    int topk, B, L, S, H, ntiles, BH, dbg;
};

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

// lanes l with ((l & KQ) == 0) == ((l & J) == 0): the lanes of a bitonic step (block KQ, distance J) that keep the maximum
template <int KQ, int J>
constexpr unsigned long long keep_max_mask() {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & KQ) == 0) == ((l & J) == 0)) m |= 1ull << l;
    return m;
}
// mask bit of the lane ? b : a, the mask a compile-time constant in an SGPR pair (a lane-derived condition costs 2-3 VALU per use,
// or 42 SGPRs if the compiler hoists all 21 of them)
__device__ __forceinline__ unsigned lane_select(unsigned a, unsigned b, unsigned long long mask) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
// 64-lane bitonic sort, descending (lane 0 = largest): 21 compare-exchange steps.  max / min of a lane and its partner are
// symmetric, so a step is max(v, partner) and min(v, partner) -- the partner fetched by a DPP operand (two single-use DPP moves: the
// compiler folds each into its v_max / v_min) or one gfx950 row / half swap whose two results ARE the pair -- and one select.
template <int KQ, int J>
__device__ __forceinline__ unsigned sort_step(unsigned v) {
    unsigned mx, mn;
    if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        mx = max(r[0], r[1]); mn = min(r[0], r[1]);
    } else if constexpr (J == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        mx = max(r[0], r[1]); mn = min(r[0], r[1]);
    } else {
        mx = max(v, wave_xor_u32<J>(v));
        mn = min(v, wave_xor_u32<J>(v));
    }
    return lane_select(mn, mx, keep_max_mask<KQ, J>());
}
__device__ __forceinline__ unsigned wave_sort_desc_u32(unsigned srt) {
    static_for<1, 7>([&](auto kq_) {
        constexpr int kq = 1 << decltype(kq_)::value;
        static_for<0, decltype(kq_)::value>([&](auto j_) {
            constexpr int j = kq >> (1 + decltype(j_)::value);
            srt = sort_step<kq, j>(srt);
        });
    });
    return srt;
}

#define CT_LOG2E 1.44269504088896340736f

// Top-k of one row held as key[e] in lane (position 64 e + lane; 0 = no element).  cb: this row's 64 x (key, position) compaction
// buffer in LDS.  Afterwards lane t < topk holds the t-th element of the (key desc, position asc) order in (rkey, rpos).
template <int EMAX>
__device__ __forceinline__ void row_topk(unsigned (&key)[EMAX], int nblk, int topk, int lane, uint2* cb, unsigned& rkey, int& rpos) {
    unsigned lm = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) lm = max(lm, key[e]);
    const unsigned wm = wave_max_u32(lm);
    // ---- threshold: topk <= #(lane maxima >= theta) <= topk + 4 where the maxima allow it (ties can force more)
    unsigned lo = max(wave_min_u32(lm), 1u), hi = wm + 1u, theta = lo;   // #(lm >= lo) >= topk: at least topk lanes hold an element (S >= topk)
    int cl = __popcll(__ballot(lm >= lo));
    for (int it = 0; it < 34 && cl > topk + 4 && hi - lo > 1u; ++it) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int c = __popcll(__ballot(lm >= mid));
        if (c >= topk) { lo = mid; cl = c; } else hi = mid;
    }
    theta = lo;
    // ---- compaction in position order
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        if (e < nblk) {
            const bool f = key[e] >= theta;
            const unsigned long long bal = __ballot(f);
            const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, (unsigned)cnt));
            if (f && slot < 64) cb[slot] = make_uint2(key[e], (unsigned)(e * 64 + lane));
            cnt += __popcll(bal);
        }
    }
    wave_lds_fence();
    rkey = 0u; rpos = 0;
    if (cnt <= 64) {   // wave-uniform
        const uint2 mine = lane < cnt ? cb[lane] : make_uint2(0u, 0xFFFFFFFFu);
        if (wm - theta < (1u << 26) - 1u) {
            // exact 32-bit packing: (key - theta + 1) needs 26 bits, the slot (= position rank among the survivors) 6
            unsigned pk = lane < cnt ? (((mine.x - theta + 1u) << 6) | (unsigned)(63 - lane)) : 0u;
            pk = wave_sort_desc_u32(pk);
            const int s = 63 - (int)(pk & 63u);
            rkey = theta + (pk >> 6) - 1u;
            rpos = (int)cb[lane < topk ? s : 0].y;
        } else {
            // survivors spread over 2^26 and more ordered keys (many binades): (key, position) pair sort
            unsigned sk = mine.x, sp = mine.y;
            static_for<1, 7>([&](auto kq_) {
                constexpr int kq = 1 << decltype(kq_)::value;
                static_for<0, decltype(kq_)::value>([&](auto j_) {
                    constexpr int j = kq >> (1 + decltype(j_)::value);
                    const unsigned ok = wave_xor_u32<j>(sk);
                    const unsigned op = wave_xor_u32<j>(sp);
                    const bool other_first = ok > sk || (ok == sk && op < sp);
                    const bool want_first = ((lane & kq) == 0) == ((lane & j) == 0);
                    const bool take = want_first == other_first;
                    sk = take ? ok : sk;
                    sp = take ? op : sp;
                });
            });
            rkey = sk; rpos = (int)sp;
        }
    } else {
        // more than 64 elements at or above the threshold (heavy ties): iterated wave argmax, smallest position first
        for (int t = 0; t < topk; ++t) {
            unsigned cur = 0;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) cur = max(cur, key[e]);
            const unsigned wmx = wave_max_u32(cur);
            bool found = false;
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                if (!found) {
                    const unsigned long long bal = __ballot(key[e] == wmx);
                    if (bal) {
                        found = true;
                        const int src = __ffsll((long long)bal) - 1;
                        if (lane == src) key[e] = 0u;
                        if (lane == t) { rkey = wmx; rpos = e * 64 + src; }
                    }
                }
            }
        }
    }
    wave_lds_fence();   // the buffer may be reused
}

template <int EMAX, int NW, int NS>   // NS ring slots (prefetch distance NS - 1 blocks); NW waves (4 rows each) share the key / value ring: NW = 8 -> 32-row tiles, half the DMA volume of NW = 4
__global__ __launch_bounds__(64 * NW, (EMAX <= 11 ? 4 : 3)) void coarse_tile_kernel(const CoarseTArgs a) {
    static_assert(NW == 4 || NW == 8, "a block is 8 DMA instructions");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RING_FLOATS = NS * 2048;         // NS slots x 64 rows x 128 B
    constexpr int WAVE_FLOATS = 512;               // per-wave scratch: 4 x 512 B compaction buffers | 2 x 1 KB probability blocks | reduction
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- tile: XCD x serves the (pair, head) slices x, x + 8, ... (their keys and values stay in its L2 for all their tiles)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int bh = (slot_id / a.ntiles) * 8 + xcd, tile = slot_id % a.ntiles;
    if (bh >= a.BH) return;
    const int H = a.H, HD = H * 32, L = a.L, S = a.S, b = bh / H, h = bh % H;
    const int nblk = (S + 63) >> 6;
    const int l0 = tile * (4 * NW) + 4 * w;
    const int nrows = min(max(L - l0, 0), 4);      // this wave's valid rows (wave-uniform)
    float* ring = smem;
    float* scr = smem + RING_FLOATS + w * WAVE_FLOATS;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const float* kb = a.k + (size_t)b * S * HD + h * 32 - 768;   // 3072 bytes low (quad_common.hpp: glds_chunk2)
    const float* vb = a.v + (size_t)b * S * HD + h * 32 - 768;
    // ---- DMA: block g < nblk = keys 64 g .., block nblk + g = values 64 g ..; this wave moves 64 / NW rows (8 / NW instructions).
    // Byte offset of a lane's 16 bytes = min(row, S - 1) * row pitch + swizzled unit: the row part advances by a constant per block
    // and is clamped on the BYTE offset, so an issue costs 3 VALU per instruction (no per-block multiply).
    const int r0 = (64 / NW) * w + (lane >> 3), r1 = r0 + 8, pu = lane & 7;
    const unsigned pitch = (unsigned)HD * 4u, last_off = (unsigned)(S - 1) * pitch;
    const unsigned rb0 = (unsigned)r0 * pitch, rb1 = (unsigned)r1 * pitch;
    // keys: lane <-> row reads of 8 units, unit swizzle (row >> 1) & 7; values: 4 lanes per row read 2 units each, swizzle (row >> 1) & 1
    const unsigned ku0 = (unsigned)((pu ^ ((r0 >> 1) & 7)) * 16 + 3072), ku1 = (unsigned)((pu ^ ((r1 >> 1) & 7)) * 16 + 3072 - 1024);
    const unsigned vu0 = (unsigned)((pu ^ ((r0 >> 1) & 1)) * 16 + 3072), vu1 = (unsigned)((pu ^ ((r1 >> 1) & 1)) * 16 + 3072 - 1024);
    auto issue = [&](int g) {
        if (g >= 2 * nblk || (a.dbg & 4)) return;
        const bool isv = g >= nblk;
        const unsigned blk_off = (unsigned)(isv ? g - nblk : g) * 64u * pitch;   // scalar
        const unsigned o0 = min(rb0 + blk_off, last_off) + (isv ? vu0 : ku0);
        const unsigned dst = ring_lds + (unsigned)((g % NS) * 8192 + w * (8192 / NW));
        if constexpr (NW == 4) {
            const unsigned o1 = min(rb1 + blk_off, last_off) + (isv ? vu1 : ku1);
            glds_chunk2(isv ? vb : kb, o0, o1, dst);
        } else {
            glds_chunk1(isv ? vb : kb, o0, dst);
        }
    };
    constexpr int NI = 8 / NW;   // DMA instructions per block and wave
    // block g has landed when at most the later blocks' instructions (min(NS - 2, 2 nblk - 1 - g) blocks) are outstanding
    auto wait_block = [&](int g) {
        const int later = min(NS - 2, 2 * nblk - 1 - g);
        if (later >= 4) glds_wait<4 * NI>();
        else if (later == 3) glds_wait<3 * NI>();
        else if (later == 2) glds_wait<2 * NI>();
        else if (later == 1) glds_wait<NI>();
        else glds_wait<0>();
    };
    static_for<0, NS - 1>([&](auto gc) { issue(decltype(gc)::value); });
    // ---- queries: lane l holds q[row l % 4][0 .. 31].  The empty asm consumes them HERE: the compiler waits for these loads now
    // (vmcnt(0), which also covers the first blocks) instead of inside the block loop, where its vmcnt(0) -- it cannot see the DMA
    // instructions -- would drain every prefetched block in every iteration.
    f32x4 qa[8];
    {
        const float* qr = a.q + (((size_t)b * L + min(l0 + (lane & 3), L - 1)) * H + h) * 32;
#pragma unroll
        for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qr + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(qa[u]));
    }
    const unsigned krow = (unsigned)(lane * 128), kswz = (unsigned)((lane >> 1) & 7);   // this lane's key row inside a slot, its unit swizzle

    // ================================================================== QK^T
    // Keys beyond S (last block) get the logit -FLT_MAX: probability exactly 0, ordered key below every real logit's, so the
    // selection needs no per-element validity test (they could only be selected if fewer than topk keys existed: rejected on the host).
    f32x4 x[EMAX];
    static_for<0, EMAX>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        x[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (e < nblk) {   // wave-uniform
            wait_block(e);
            if (!(a.dbg & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // slot (e + NS - 1) % NS was read in iteration e - 1; every wave's reads had landed in registers (its MFMAs consumed them)
            // before it arrived at this barrier
            issue(e + NS - 1);
            if (nrows > 0) {
                const char* sb = reinterpret_cast<const char*>(ring) + (e % NS) * 8192 + krow;
                f32x4 kr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(sb + (((unsigned)u ^ kswz) << 4));
                f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c, 0, 0, 0);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) x[e][f] = a.temp * c[f];
                if (64 * e + 64 > S) {   // the last block (wave-uniform)
                    const bool valid = lane < S - 64 * e;
#pragma unroll
                    for (int f = 0; f < 4; ++f) x[e][f] = valid ? x[e][f] : -3.402823466e38f;
                }
            }
        }
    });

    // ================================================================== softmax statistics + top-k, row by row (registers only)
    float rinv[4] = {0.f, 0.f, 0.f, 0.f};
    float rsc[4] = {0.f, 0.f, 0.f, 0.f};
    int ridx[4] = {0, 0, 0, 0};
    if (!(a.dbg & 1)) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < nrows) {   // wave-uniform
                unsigned key[EMAX];
                unsigned lm = 0;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    key[e] = e < nblk ? f2ord(x[e][f]) : 0u;
                    lm = max(lm, key[e]);
                }
                const float m = ord2f(wave_max_u32(lm));
                const float mb = m * CT_LOG2E;
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    const float p = e < nblk ? __builtin_amdgcn_exp2f(__builtin_fmaf(x[e][f], CT_LOG2E, -mb)) : 0.f;
                    x[e][f] = p;   // the logit lives on in key[e]
                    sum += p;
                }
                sum = wave_sum_f32(sum);
                rinv[f] = __builtin_amdgcn_rcpf(sum);
                unsigned rkey; int rpos;
                row_topk<EMAX>(key, nblk, a.topk, lane, reinterpret_cast<uint2*>(scr) + f * 64, rkey, rpos);
                ridx[f] = rpos;
                rsc[f] = __builtin_amdgcn_exp2f(__builtin_fmaf(ord2f(rkey), CT_LOG2E, -mb)) * rinv[f];
            }
        }
    } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) rinv[f] = 1.f;
    }

    // ================================================================== A.V
    f32x4 acc[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bq = lane >> 2, j = lane & 3;
    // value rows 16 t + bq, logical units 2 j, 2 j + 1 (d = 8 j ..), physical unit = logical ^ ((row >> 1) & 1)
    const unsigned vrd0 = (unsigned)(bq * 128 + (((2 * j) ^ ((bq >> 1) & 1)) * 16)), vrd1 = (unsigned)(bq * 128 + (((2 * j + 1) ^ ((bq >> 1) & 1)) * 16));
    static_for<0, EMAX>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if (e < nblk) {   // wave-uniform
            const int g = nblk + e;
            wait_block(g);
            if (!(a.dbg & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(g + NS - 1);   // into the slot read in the previous iteration (see the QK^T loop)
            float* Pb = scr + (e & 1) * 256;
            *reinterpret_cast<f32x4*>(Pb + lane * 4) = x[e];   // key = lane: its 4 rows' probabilities
            wave_lds_fence();
            float pa[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pa[t] = Pb[64 * t + lane];   // lane (bq, i = j): p[row i][key 16 t + bq]
            const char* sb = reinterpret_cast<const char*>(ring) + (g % NS) * 8192;
            f32x4 vv[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                vv[t][0] = *reinterpret_cast<const f32x4*>(sb + t * 2048 + vrd0);
                vv[t][1] = *reinterpret_cast<const f32x4*>(sb + t * 2048 + vrd1);
            }
            if (nrows > 0 && !(a.dbg & 2)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg)
                        acc[gg] = __builtin_amdgcn_mfma_f32_4x4x1f32(pa[t], vv[t][gg >> 2][gg & 3], acc[gg], 0, 0, 0);
                }
            }
        }
    });
    // every DMA has landed (the last wait above was vmcnt(0)); no barrier below: the scratch is wave-private
    if (nrows == 0) return;
    // ---- fold the 16 key classes: lane (bq, j) holds the partial sums of rows 0..3, d = 8 j + g over the keys == bq (mod 16)
    {
        float* red = scr;   // [4 DPP rows][4 rows][32 d]
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = acc[g][i];
                y += dpp_zfill_f32<0x104>(y);   // row_shl:4
                y += dpp_zfill_f32<0x108>(y);   // row_shl:8: lanes l % 16 < 4 hold the sum over their DPP row's 4 key classes
                acc[g][i] = y;
            }
        wave_lds_fence();   // the probability blocks are dead
        if ((lane & 15) < 4) {
            const int R = lane >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(red + (R * 4 + i) * 32 + 8 * j) = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
                *reinterpret_cast<f32x4*>(red + (R * 4 + i) * 32 + 8 * j + 4) = (f32x4){acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
            }
        }
        wave_lds_fence();
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int i = lane >> 4, dd = 2 * (lane & 15);
        f32x2 tot = *reinterpret_cast<const f32x2*>(red + i * 32 + dd);
#pragma unroll
        for (int R = 1; R < 4; ++R) {
            const f32x2 p = *reinterpret_cast<const f32x2*>(red + (R * 4 + i) * 32 + dd);
            tot.x += p.x; tot.y += p.y;
        }
        const float ri = i == 0 ? rinv[0] : i == 1 ? rinv[1] : i == 2 ? rinv[2] : rinv[3];
        tot.x *= ri; tot.y *= ri;
        if (i < nrows) {
            const size_t o = (((size_t)b * L + l0 + i) * H + h) * 32 + dd;
            if (a.message) *reinterpret_cast<f32x2*>(a.message + o) = tot;
            if (a.acc_out) *reinterpret_cast<f32x2*>(a.acc_out + o) = (f32x2){tot.x * a.w_level, tot.y * a.w_level};   // :274
        }
    }
    // ---- the selection's results: lane t < topk of row f
    if (lane < a.topk) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < nrows) {
                const int l = l0 + f;
                if (a.topk_tab) a.topk_tab[((size_t)bh * L + l) * a.topk + lane] = ridx[f];
                const size_t o = (((size_t)b * L + l) * a.topk + lane) * H + h;
                if (a.topk_idx) a.topk_idx[o] = ridx[f];
                if (a.topk_score) a.topk_score[o] = rsc[f];
            }
        }
    }
}

// -> CASMTR_ERR_UNSUPPORTED when the shape is outside this kernel (the caller then uses the three-kernel path)
int casmtr_qta_coarse_level_tile(const float* q, const float* k, const float* v, float temp, int topk, float w_level, float* message,
                                 float* acc_out, float* topk_score, int64_t* topk_idx, int32_t* topk_tab, int B, int L, int S, int H,
                                 hipStream_t s) {
    const int E = (S + 63) / 64;
    if (E > 16 || S < 1 || topk < 1 || topk > 60 || topk > S || H < 1 || (long long)S * H * 128 >= (1ll << 31)) return CASMTR_ERR_UNSUPPORTED;
    CoarseTArgs a{};
    a.q = q; a.k = k; a.v = v; a.message = message; a.acc_out = acc_out; a.topk_score = topk_score; a.topk_idx = topk_idx; a.topk_tab = topk_tab;
    // measured at 26x26, B = 8 (tools/coarse_tile_sweep.py): 4 waves x 4 slots 103 us, 4 x 3 115, 8 waves x {3,4,6} 105-113 (the 8-wave
    // barrier costs what the halved DMA volume saves), 4 x 6 141 (2 workgroups per CU).  Instantiated: 4 waves x {3, 4} slots.
    const int nw = 4;
    int ns = 4;
    { const char* ev = getenv("CASMTR_CT_SLOTS"); if (ev && atoi(ev) == 3) ns = 3; }   // measurement knob
    a.temp = temp; a.w_level = w_level; a.topk = topk; a.B = B; a.L = L; a.S = S; a.H = H; a.ntiles = (L + 4 * nw - 1) / (4 * nw); a.BH = B * H;
    a.dbg = g_debug_flags >> 8;   // CASMTR debug flags 256 / 512 / 1024: skip the selection / the A.V arithmetic / the DMA (timing experiments)
    const unsigned grid = (unsigned)((a.BH + 7) / 8 * 8 * a.ntiles);
#define CT_LAUNCH(EE, NWW, NSS)                                                                                                      \
    {                                                                                                                                \
        const size_t lds = sizeof(float) * (NSS * 2048 + NWW * 512);                                                                 \
        if (lds > 48 * 1024)                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coarse_tile_kernel<EE, NWW, NSS>),                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                         \
        prof_symbol_args(CASMTR_PROF_COARSE_FUSED, "<%d,%d,%d>", EE, NWW, NSS);                                                     \
        CASMTR_LAUNCH_TIMED(CASMTR_PROF_COARSE_FUSED, (coarse_tile_kernel<EE, NWW, NSS>), dim3(grid), dim3(64 * NWW), lds, s, a);    \
    }
#define CT_CASE(EE)                                                  \
    if (E <= EE) {                                                   \
        if (ns == 4) CT_LAUNCH(EE, 4, 4)                             \
        else CT_LAUNCH(EE, 4, 3)                                     \
        CASMTR_CHECK_LAUNCH();                                       \
        return 0;                                                    \
    }
    CT_CASE(4)
    CT_CASE(5)
    CT_CASE(8)
    CT_CASE(11)
    CT_CASE(16)
#undef CT_CASE
#undef CT_LAUNCH
    return CASMTR_ERR_UNSUPPORTED;
}
