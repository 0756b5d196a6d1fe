// QTAttB.process_fine_level + its share of the message merge (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/
// quadtree_attention.py:180-229,262-284) on QUAD-MAJOR operands: the round-3 fine-level kernel.
//
// Layout ("quad-major per head", produced by casmtr_nchw_to_quads_multi straight from the module's NCHW pyramids, at the cost the
// token-major conversion had):  x_qm[b][h][Q][c][d], Q = (r/2)*(w/2) + (c/2) the quad of token (r, c), c = (r&1)*2 + (c&1) its child
// slot, d < 32.  A parent selected at the previous level IS a quad index of this level's key grid, so
//   * the 4 children of a parent (:193-199) are ONE contiguous 512-byte run -- no (row, col) decomposition, no integer division per
//     candidate, 4 lines per address instead of 1 (token-major: 4 separate 128-byte rows in two image rows, 1 KB apart per token);
//   * a (pair, head) slice of K or V is one contiguous 1.4 MB block (head <-> XCD affinity keeps it in that XCD's L2);
//   * the 4 queries of an item are one contiguous 512-byte run.
// The previous level's top-k arrives as a compact int32 table parents[b][h][quad][Kp] (written by the previous level's kernel next
// to -- or instead of -- the reference's int64 [B,L,K,H] tensor, whose 8-byte entries at a 64-byte stride made every XCD fetch all
// eight heads' lines: 8x over-fetch of the index stream).
//
// Structure: persistent single-wave workgroups, ~10 KB of LDS each (14-16 waves per CU; the round-2 kernel: 8), one (quad, head) item
// per wave at a time.  K and V rows come in by LDS-DMA in 4 KB chunks (8 parents = 32 candidate rows = 4 wave-instructions behind ONE
// M0 write: the instruction's immediate offset advances the LDS destination and the source together, tools/probes/glds_offset.hip)
// through a two-slot ring:
//   K pass (64 candidates = both slots): lane <-> candidate, 32 v_mfma_f32_4x4x1_16B_f32 = the exact d-ascending fmaf chain of the
//     oracle (tools/probes/mfma4x4_layout.hip), so the logits and every index derived from them are bit-identical;
//   softmax: one series (child) per 16-lane DPP row;
//   top-k (levels that feed a finer one): each lane sorts its 8 packed keys (25 high bits of the ordered logit | 127 - position: ONE
//     32-bit compare-exchange = v_max_u32 + v_min_u32), then a 4-round bitonic merge tree across the row keeps the 16 best: after it
//     every lane pair holds the sorted top-16, lane j reads rank (j&1)*8 + j/2 out of its own registers and the row stores the list
//     with ONE instruction per output tensor (round 2: 16 rounds of row-argmax with a divergent one-lane store each, ~50 VALU per
//     extracted element).  The packed order equals the exact (logit desc, position asc) order unless two of the 17 best share their 25
//     high bits; that is checked (adjacent ranks + a count of elements >= the last rank's bucket) and such waves -- exact ties, logits
//     closer than 2^-16 relative -- redo the selection with the exact iterated argmax;
//   V chunks (32 rows): 16 v_mfma_f32_4x4x1 each (two rows per instruction), probabilities as operand A from 4 ds_read_b128.
// Results are in raster order of the h0 x w0 grid; final = final[parent] + message * weight (:277-281) fused into the store.
//
// Front end (round 4): the next item's queries, parent list and final[parent] row also arrive by LDS-DMA (4 x 256-byte
// global_load_lds_dword into a double-buffered 1 KB staging area, issued at the START of the current item, consumed under its last
// V chunks).  Round 3 fetched them with ordinary global loads into registers; the compiler cannot see the DMA instructions, so its
// own `s_waitcnt vmcnt(0)` for those loads -- placed right behind them, where it copies the values into the loop-carried registers --
// drained the whole DMA ring two or three times per item.  Now the loop contains no
// compiler-visible vector load at all (tools/check_quad_isa.py checks that).
#include <stdio.h>
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct FineQArgs {
    const float* q;          // [B,H,Lq0,4,32]
    const float* key;        // [B,H,Lq1,4,32]
    const float* value;      // [B,H,Lq1,4,32]
    const int32_t* parents;  // [B,H,Lq0,Kp] quad index on the key grid (= the previous level's absolute top-k index)
    const float* acc_in;     // nullable [B,Lq0,H*32]
    float* message;          // nullable [B,L,H*32]
    float* acc_out;          // nullable [B,L,H*32]
    int32_t* topk_tab;       // nullable [B,H,L,topk]: this level's top-k as the next level's parents table
    float* topk_score;       // nullable [B,L,topk,H]
    int64_t* topk_idx;       // nullable [B,L,topk,H]
    float temp, w_level;
    int topk, B, h0, w0, h1, w1, H, Kp, nquads, lq1;
    unsigned div_magic;      // ceil(2^32 / (w1/2)): p / (w1/2) == umulhi(p, div_magic) for p < 2^22 (0: w1/2 == 1)
    int* ctr;                // nullable: work_counters() -- items beyond a wave's first two are claimed (dynamic schedule), not dealt out
    int claim;               // items per claim (consecutive): one L2 atomic on one address costs ~16 ns, the finest level hands out an item
                             // every 11 ns per XCD -- with one item per claim the launch took 364 instead of 235 us (profiles/r06_dyn_ab.txt)
    unsigned magic_chunk, magic_last, magic_wq;   // n / d == umulhi(n, magic) for d = items per XCD chunk (all but the last / the last chunk) and
                                                  // d = w0 / 2; 0: d == 1
    int xflags;              // experiment switches (CASMTR_FQ_FLAGS): 1 nt loads of the side streams, 2 nt stores, 4 sc1 stores, 8 one K/V slice for all pairs, 16 staggered start
};

__device__ __forceinline__ void ce_desc(unsigned& a, unsigned& b) {   // compare-exchange: a >= b afterwards
    const unsigned hi = max(a, b), lo = min(a, b);
    a = hi; b = lo;
}
// descending sort of 8 registers (optimal 19-comparator network) and of a bitonic 8-sequence (12 comparators)
__device__ __forceinline__ void sort8_desc(unsigned (&s)[8]) {
    ce_desc(s[0], s[1]); ce_desc(s[2], s[3]); ce_desc(s[4], s[5]); ce_desc(s[6], s[7]);
    ce_desc(s[0], s[2]); ce_desc(s[1], s[3]); ce_desc(s[4], s[6]); ce_desc(s[5], s[7]);
    ce_desc(s[1], s[2]); ce_desc(s[5], s[6]); ce_desc(s[0], s[4]); ce_desc(s[3], s[7]);
    ce_desc(s[1], s[5]); ce_desc(s[2], s[6]);
    ce_desc(s[1], s[4]); ce_desc(s[3], s[6]);
    ce_desc(s[2], s[4]); ce_desc(s[3], s[5]);
    ce_desc(s[3], s[4]);
}
__device__ __forceinline__ void bitonic8_desc(unsigned (&s)[8]) {
    ce_desc(s[0], s[4]); ce_desc(s[1], s[5]); ce_desc(s[2], s[6]); ce_desc(s[3], s[7]);
    ce_desc(s[0], s[2]); ce_desc(s[1], s[3]); ce_desc(s[4], s[6]); ce_desc(s[5], s[7]);
    ce_desc(s[0], s[1]); ce_desc(s[2], s[3]); ce_desc(s[4], s[5]); ce_desc(s[6], s[7]);
}
// One round of the merge tree.  Before: every lane pair (even, odd) of a group of G lanes holds the sorted top-16 of the group's
// elements (even lane: ranks 0-7 in s[0..7], odd lane: ranks 8-15); CTRL mirrors the 2G-lane group (lane k <- lane 2G-1-k), so an even
// lane reads the OTHER group's ranks 15-i and an odd lane its ranks 7-i: max() of the two is the bitonic top-16 of both groups,
// which one cross-lane step (distance 8: even keeps max, odd keeps min) and an in-lane bitonic sort put in order.  After: the same
// invariant for groups of 2G lanes.
template <int CTRL>
__device__ __forceinline__ void merge_round(unsigned (&s)[8], bool odd) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = max(s[i], dpp_u32<CTRL>(s[7 - i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned t = dpp_u32<0xB1>(x[i]);
        s[i] = odd ? min(x[i], t) : max(x[i], t);
    }
    bitonic8_desc(s);
}

// Softmax over an item's candidates (+ the top-k selection of levels that feed a finer one), one series (child) per 16-lane row.
// In: lg[p][f] = logit of candidate 64 p + lane for child f.  Out: P[child][parity][m] in Pld (operand A of the V chunks); EXACT: lane
// (f, j) gets rank (j&1)*8 + j/2 of child f: its probability (out_sc) and absolute key index (out_idx).  t2 = the item's staged parent list.
template <int NPASS, bool EXACT, bool FULL>
__device__ __forceinline__ void softmax_select(const FineQArgs& a, const f32x4 (&lg)[NPASS], float* Pld, const int* t2, int lane, int K,
                                               int w1p, float& out_sc, int& out_idx) {
    constexpr int KMAX = 64 * NPASS;
    constexpr int E = KMAX / 16;
    constexpr int KS = KMAX + 4;
    constexpr int PST = 32 * NPASS + 4;
    const int f = lane >> 4, j = lane & 15;
    float* Sld = Pld;   // [4][KS]
#pragma unroll
    for (int pp = 0; pp < NPASS; ++pp)
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) Sld[ff * KS + 64 * pp + lane] = lg[pp][ff];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float lv[E];
    unsigned xk[E];
    const f32x4* sp = reinterpret_cast<const f32x4*>(Sld + f * KS + j * E);
#pragma unroll
    for (int e4 = 0; e4 < E / 4; ++e4) {
        const f32x4 v = sp[e4];
        lv[4 * e4 + 0] = v.x; lv[4 * e4 + 1] = v.y; lv[4 * e4 + 2] = v.z; lv[4 * e4 + 3] = v.w;
    }
    float m;
    if constexpr (EXACT) {
        unsigned lm = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            xk[e] = (FULL || j * E + e < K) ? f2ord(lv[e]) : 0u;
            lm = max(lm, xk[e]);
        }
        m = ord2f(row16_max_u32(lm));
    } else {   // no selection: the maximum only centres the exponentials
        float fm = -3.0e38f;
#pragma unroll
        for (int e = 0; e < E; ++e) fm = (FULL || j * E + e < K) ? fmaxf(fm, lv[e]) : fm;
        fm = fmaxf(fm, dpp_f32<0xB1>(fm));
        fm = fmaxf(fm, dpp_f32<0x4E>(fm));
        fm = fmaxf(fm, dpp_f32<0x141>(fm));
        m = fmaxf(fm, dpp_f32<0x140>(fm));
    }
    float ps[E];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ps[e] = (FULL || j * E + e < K) ? __expf(lv[e] - m) : 0.f;
        sum += ps[e];
    }
    sum = __builtin_amdgcn_rcpf(row16_sum_f32(sum));   // 1 ulp: the probabilities carry a 1e-4 tolerance, no index depends on them
#pragma unroll
    for (int e = 0; e < E; ++e) ps[e] = ps[e] * sum;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // every lane has its logits: the buffer becomes P
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // candidate j*E + e -> P[f][e & 1][(j*E + e) >> 1]
    if constexpr (E == 4) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(Pld + (f * 2 + 0) * PST + 2 * j) = (f32x2){ps[0], ps[2]};
        *reinterpret_cast<f32x2*>(Pld + (f * 2 + 1) * PST + 2 * j) = (f32x2){ps[1], ps[3]};
    } else {
        *reinterpret_cast<f32x4*>(Pld + (f * 2 + 0) * PST + 4 * j) = (f32x4){ps[0], ps[2], ps[4], ps[6]};
        *reinterpret_cast<f32x4*>(Pld + (f * 2 + 1) * PST + 4 * j) = (f32x4){ps[1], ps[3], ps[5], ps[7]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (EXACT) {
        // ---- selection on the logits, (logit desc, position asc); lane j ends up with rank (j&1)*8 + j/2
        const bool odd = j & 1;
        const int rank = (j & 1) * 8 + (j >> 1), topm1 = a.topk - 1;
        unsigned s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            s[e] = (e < E && xk[e < E ? e : 0] != 0u) ? ((xk[e < E ? e : 0] & ~127u) | (unsigned)(127 - (j * E + e))) : 0u;
        sort8_desc(s);
        unsigned loc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) loc[e] = s[e];
        {   // round 1: the pair's 16 elements, sorted across (even, odd)
            unsigned x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned tt = dpp_u32<0xB1>(s[7 - i]);
                x[i] = odd ? min(s[i], tt) : max(s[i], tt);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = x[i];
            bitonic8_desc(s);
        }
        merge_round<0x1B>(s, odd);    // quad_perm [3,2,1,0]
        merge_round<0x141>(s, odd);   // row_half_mirror
        merge_round<0x140>(s, odd);   // row_mirror
        unsigned mine = s[0];
        {
            const int sel = j >> 1;
            const unsigned m01 = (sel & 1) ? s[1] : s[0], m23 = (sel & 1) ? s[3] : s[2];
            const unsigned m45 = (sel & 1) ? s[5] : s[4], m67 = (sel & 1) ? s[7] : s[6];
            const unsigned m03 = (sel & 2) ? m23 : m01, m47 = (sel & 2) ? m67 : m45;
            mine = (sel & 4) ? m47 : m03;
        }
        // is the packed order the exact order for the first topk ranks?  (see the file header)
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 7; ++i) bad |= ((s[i] ^ s[i + 1]) < 128u) && ((odd ? 8 : 0) + i < topm1);
        {
            const unsigned tt = dpp_u32<0xB1>(s[0]);
            bad |= !odd && ((s[7] ^ tt) < 128u) && (7 < topm1);
        }
        const int jstar = (topm1 & 7) * 2 + (topm1 >> 3);
        const unsigned thr = row16_max_u32(j == jstar ? mine : 0u) & ~127u;
        unsigned cntge = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) cntge += (loc[e] >= thr) ? 1u : 0u;
        bad |= row16_sum_u32(cntge) != (unsigned)a.topk;
        int my_pos = 127 - (int)(mine & 127u);
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            // exact path: iterated row argmax on the full 32-bit keys, first position by ballot / ffs
            unsigned key[E];
#pragma unroll
            for (int e = 0; e < E; ++e) key[e] = xk[e];
            for (int tk = 0; tk < a.topk; ++tk) {
                unsigned cur = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) cur = max(cur, key[e]);
                const unsigned rm = row16_max_u32(cur);
                const unsigned long long bal = __ballot(cur == rm);
                const unsigned bits = (unsigned)(bal >> (f * 16)) & 0xFFFFu;
                const int wj = __ffs(bits) - 1;   // first lane of the row holding the maximum -> smallest position
                unsigned kp1 = 0;
                if (j == wj) {
                    bool done = false;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool hit = !done && key[e] == rm;
                        if (hit) { kp1 = (unsigned)(j * E + e + 1); key[e] = 0u; done = true; }
                    }
                }
                const unsigned wp1 = row16_max_u32(kp1);
                if (rank == tk) my_pos = (int)wp1 - 1;
            }
        }
        if (rank >= a.topk) my_pos = 0;
        const int c = my_pos;
        out_sc = Pld[(f * 2 + (c & 1)) * PST + (c >> 1)];
        const int par = t2[c >> 2];   // candidate c = 4 * parent slot + child
        const int qy1 = a.div_magic ? (int)__umulhi((unsigned)par, a.div_magic) : par;
        const int qx1 = par - qy1 * w1p;
        out_idx = (2 * qy1 + ((c >> 1) & 1)) * a.w1 + 2 * qx1 + (c & 1);   // absolute index on the h1 x w1 grid (:224)
    }
}

template <int NPASS, bool EXACT, bool FULL>   // EXACT: the level feeds a finer one (top-k requested): bit-exact sequential d-chain;
                                              // FULL: the lists have exactly 64 * NPASS candidates (every shipped config): no per-element bound test
__global__ __launch_bounds__(128, (NPASS == 1 ? 4 : 3)) void fine_quad_kernel(const FineQArgs a) {
    constexpr int KMAX = 64 * NPASS;
    constexpr int KS = KMAX + 4;          // row stride of the logits transposition buffer
    constexpr int PST = 32 * NPASS + 4;   // stride of one (child, parity) run of probabilities
    constexpr int P_FLOATS = 8 * PST;
    constexpr int NV = 2 * NPASS;         // V chunks per item
    static_assert(P_FLOATS >= 4 * KS, "the transposition buffer aliases the probabilities");
    constexpr int STG = 256;              // staging buffer of one item: q [4][32] (16-byte units XOR-swizzled) | parents [32] | final[parent] [32] | unused
    constexpr int WAVE_FLOATS = 2048 + P_FLOATS + 2 * STG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // two INDEPENDENT waves per workgroup (no block barrier anywhere): LDS is granted in coarse granules, and one wave's ~11 KB
    // rounded up alone leaves room for fewer single-wave workgroups per CU than two together
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ring = smem + wave * WAVE_FLOATS;                // 2 slots x 32 rows x 128 B (XOR-swizzled 16-byte units)
    float* Pld = ring + 2048;                               // P[child][parity][m] (candidate 2m + parity); first the [4][KS] logits
    float* stg = Pld + P_FLOATS;                            // [2][STG]
    const int H = a.H, HD = H * 32, Kp = a.Kp, K = 4 * Kp;
    const int L = a.h0 * a.w0, wq = a.w0 >> 1, Lq = a.nquads, w1p = a.w1 >> 1;
    // ---- work list: XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = cnt > 0 ? a.B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    const int t = (blockIdx.x >> 3) * 2 + wave;
    int* const ctr = a.ctr ? a.ctr + xcd * WORK_XCD_INTS : nullptr;
    if (g >= G || t >= total) {
        if (ctr && lane == 0) work_leave(ctr, stride);
        return;
    }
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const int un = lane & 7;
    // DMA source offset inside a parent's 512-byte run.  Row r = 8 j + lane / 8 of a pass (j = DMA instruction 0..7); physical unit
    // `un` of row r receives logical unit un ^ ((r >> 1) & 7): lane <-> candidate ds_read_b128 of the K pass is then conflict-free.
    unsigned cK[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        cK[j] = (unsigned)(((lane >> 3) & 3) * 128 + ((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072 - j * 1024);
    unsigned rd[8];    // K pass: byte offset of logical unit u of this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    unsigned va[8];    // V chunk: byte offset of V[row 2 mm + lane/32][d = lane%32] for mm % 8 == x, minus mm * 256
#pragma unroll
    for (int x = 0; x < 8; ++x) va[x] = (unsigned)((lane >> 5) * 128 + ((((lane & 31) >> 2) ^ x) * 16) + (lane & 3) * 4);
    const float* pa = Pld + ((lane & 3) * 2 + (lane >> 5)) * PST;   // operand A of the V chunks: P[child lane%4][parity lane/32][.]

    // ---- per-item front end, one item ahead: global -> LDS staging (prefetch, DMA), staging -> DMA offsets (stage_in)
    struct Item { int b, quad, l00; };                  // pair, quad, first child's token (child f -> l00 + (f>>1)*w0 + (f&1))
    // Item tt of this XCD's list: pair tt / cnt, quad g * chunk + tt % cnt.  The wave's first two items are static (its index, + stride);
    // from the third on they are either dealt out with that stride (ctr == nullptr) or CLAIMED from the XCD's counter: the waves of an
    // XCD then work on one compact front of consecutive items whatever their individual speeds (cascade_quad.hip has the measurements).
    const unsigned mcnt = g == G - 1 ? a.magic_last : a.magic_chunk;
    auto item_of = [&](int tt, Item& it) {
        const unsigned ut = (unsigned)__builtin_amdgcn_readfirstlane(tt);
        const unsigned b = mcnt ? __umulhi(ut, mcnt) : ut;
        const unsigned quad = (unsigned)(g * chunk) + (ut - b * (unsigned)cnt);
        const unsigned cy = a.magic_wq ? __umulhi(quad, a.magic_wq) : quad, cx = quad - cy * (unsigned)wq;
        it.b = (int)b; it.quad = (int)quad; it.l00 = (int)(2 * cy * (unsigned)a.w0 + 2 * cx);
    };
    int tn = t + stride;   // the item after it_nx
    int rem = 0;           // items left in the claimed run behind tn
    if (a.xflags & 16) {   // timing experiment: the waves of a CU start up to one item period apart
        const int k = ((int)(blockIdx.x >> 3) % 5) * 2 + wave;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(11);
    }
    const unsigned stg_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(stg));
    // The item's whole front end arrives by ONE global_load_lds_dwordx4 (round 5; four global_load_lds_dword before: an LDS-DMA
    // wave-instruction costs the CU's texture-address path ~19 cycles whatever its width, tools/probes/dma_width.hip, so the four narrow
    // ones were a sixth of an item's address-path time).  Lane l fills the 16-byte staging unit l:
    //   units  0..31  q: child r = l / 8, physical unit l % 8 <- logical unit (l % 8) ^ (r >> 1) (conflict-free A-operand reads)
    //   units 32..39  parents[0 .. 31] in list order (lists shorter than 32: the last 16-byte unit of the list again)
    //   units 40..47  final[parent] row of this head (acc_in), 32 floats
    //   units 48..63  unused (they re-read the lane-47 source)
    // The three sources are different tensors, so the instruction takes a 64-bit address per lane:
    //   address = lane base + qd * mulq + bq * mula,  qd = (b * H + h) * Lq + quad,  bq = b * Lq + quad.
    unsigned long long sbase;
    unsigned mulq, mula;
    {
        const int u = lane < 48 ? lane : 47;
        if (u < 32) {
            const int r = u >> 3, pu = u & 7;
            sbase = (unsigned long long)a.q + (unsigned)(r * 128 + ((pu ^ (r >> 1)) * 16));
            mulq = 512u; mula = 0u;
        } else if ((u < 40 || !a.acc_in) && !(Kp & 3)) {
            sbase = (unsigned long long)a.parents + (unsigned)(min((u - 32) & 7, Kp / 4 - 1) * 16);
            mulq = (unsigned)(Kp * 4); mula = 0u;
        } else if (u < 40 || !a.acc_in) {          // ragged lists (Kp % 4 != 0) come by their own dword instruction below: nothing to read here
            sbase = (unsigned long long)a.q;
            mulq = 512u; mula = 0u;
        } else {
            sbase = (unsigned long long)a.acc_in + (unsigned)(h * 128 + (u - 40) * 16);
            mulq = 0u; mula = (unsigned)(HD * 4);
        }
    }
    auto prefetch = [&](const Item& it, int buf) {   // 1 DMA instruction
        const unsigned qd = (unsigned)((it.b * H + h) * Lq + it.quad), bq = (unsigned)(it.b * Lq + it.quad);   // wave-uniform
        const unsigned long long addr = sbase + (unsigned long long)qd * mulq + (unsigned long long)bq * mula;
        const unsigned dst = stg_lds + (unsigned)(buf * STG * 4);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(addr), "s"(dst) : "memory");
        if ((Kp & 3) && lane < 32) {   // a 16-byte read of the last list piece would run past the table's end: 4-byte pieces, lanes 0-31 only
            const unsigned off = qd * (unsigned)(Kp * 4) + (unsigned)(min(lane, Kp - 1) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(off), "s"(a.parents), "s"(dst + 512u) : "memory");
        }
    };
    unsigned voff[NPASS][8];   // DMA instruction j of pass p: source offset of this lane's 16 bytes (rows 64p + 8j + lane/8)
    auto stage_in = [&](int buf) {   // the staged parent list -> this lane's DMA offsets
        // DMA instruction jj = 4 i + x of the item covers parent slots 2 jj (lanes 0-31) and 2 jj + 1 (lanes 32-63)
        const int* t2 = reinterpret_cast<const int*>(stg + buf * STG + 128);
        const bool hh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2 * NPASS; ++i) {
            const int4 pa4 = *reinterpret_cast<const int4*>(t2 + 8 * i), pb4 = *reinterpret_cast<const int4*>(t2 + 8 * i + 4);
            voff[i >> 1][(i & 1) * 4 + 0] = ((unsigned)(hh ? pa4.y : pa4.x) << 9) + cK[0];
            voff[i >> 1][(i & 1) * 4 + 1] = ((unsigned)(hh ? pa4.w : pa4.z) << 9) + cK[1];
            voff[i >> 1][(i & 1) * 4 + 2] = ((unsigned)(hh ? pb4.y : pb4.x) << 9) + cK[2];
            voff[i >> 1][(i & 1) * 4 + 3] = ((unsigned)(hh ? pb4.w : pb4.z) << 9) + cK[3];
        }
    };
    // chunk c of pass p (rows 64p + 32c ..) of K (isv = 0) or V (isv = 1) of pair b -> ring slot c
    // (xflags & 8, timing experiment: every pair gathers from pair 0's slices -- no slice transitions in the L2; results are garbage)
    const size_t pair_pitch = (a.xflags & 8) ? 0 : (size_t)H * a.lq1 * 128;             // floats between two pairs' slices
    const float* const k0 = a.key + (size_t)h * a.lq1 * 128 - 768;                       // this head's slice of pair 0, 3072 bytes low
    const float* const v0 = a.value + (size_t)h * a.lq1 * 128 - 768;
    auto issue = [&](int isv, auto pc, auto cc, int b) {
        constexpr int p = decltype(pc)::value, c = decltype(cc)::value;
        const float* base = (isv ? v0 : k0) + (size_t)b * pair_pitch;   // wave-uniform
        glds_chunk(base, voff[p][4 * c + 0], voff[p][4 * c + 1], voff[p][4 * c + 2], voff[p][4 * c + 3], ring_lds + (unsigned)(c * 4096));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // (Round 5, measured and removed: "slice warming".  A wave's first item of a new pair finds that pair's K / V slice cold in the L2 and
    //  pays its K and its V round trip in sequence; with one slice for all pairs the finest level takes 169 instead of 191-204 us
    //  (tools/fq_samepair.py, CASMTR_FQ_FLAGS=8).  Requesting the V rows of such an item together with its K rows -- the same DMA
    //  instructions pointed at a 1 KB sink in LDS -- made the step SLOWER: finest level 2.29 -> 2.38 ms, middle level 1.30 -> 1.42 ms per
    //  step in alternating runs on one box.  The extra 8-16 instructions sit in front of the K rows in the in-order return queue.)

    Item it_cur{}, it_nx{};
    item_of(t, it_cur);
    prefetch(it_cur, 0);
    glds_wait<0>();
    stage_in(0);
    lds_reads_done();
    int cbuf = 0;   // staging buffer of the current item
    float acc_cur = 0.f;
    // results of the previous item: stored right behind the next item's first DMA wait (vmcnt counts stores too, in order: a store
    // issued just in front of a wait stalls the wave for its whole round trip)
    f32x4 pend = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pend_acc = 0.f, pend_sc = 0.f;
    int pend_b = 0, pend_l00 = 0, pend_idx = 0;
    bool have_pend = false;
    // What only the stores of an item's results use -- five output pointers, L, HD, w0, top-k, the level weight -- lives in VGPRs
    // (opaque copies: the compiler cannot tell that they are wave-uniform).  In SGPRs they pushed the EXACT instances over the 102
    // scalar registers: ~107 v_readlane reloads + ~109 hazard s_nop per item in the middle level's loop (a seventh of its instructions).
    // (Only where the scalar file overflows and vector registers are to spare: the 128-candidate instances with top-k, 3 waves per SIMD.
    //  The 64-candidate ones run at 4 waves per SIMD = 128 VGPRs and would spill vector registers instead.)
    constexpr bool IN_VGPR = EXACT && NPASS == 2;
    auto vg = [](auto x) { if constexpr (IN_VGPR) asm volatile("" : "+v"(x)); return x; };
    auto vgp = [&](auto* ptr) {
        if constexpr (!IN_VGPR) return ptr;
        else {
            unsigned lo = (unsigned)(unsigned long long)ptr, hi2 = (unsigned)((unsigned long long)ptr >> 32);
            asm volatile("" : "+v"(lo), "+v"(hi2));
            return reinterpret_cast<decltype(ptr)>(((unsigned long long)hi2 << 32) | lo);
        }
    };
    float* const o_message = vgp(a.message);
    float* const o_acc = vgp(a.acc_out);
    int32_t* const o_tab = vgp(a.topk_tab);
    float* const o_score = vgp(a.topk_score);
    int64_t* const o_idx = vgp(a.topk_idx);
    const int vL = vg(L), vHD = vg(HD), vw0 = vg(a.w0), vtopk = vg(a.topk), vH = vg(H), vh = vg(h);
    const float vwl = vg(a.w_level);
    auto flush = [&]() {
        if (have_pend && !(a.xflags & 32)) {   // (32: timing experiment -- no result stores at all)
            const int hi = lane >> 5;
            const float vA = hi ? pend[2] : pend[0], vB = hi ? pend[3] : pend[1];
            const size_t o = ((size_t)pend_b * vL + pend_l00 + hi * vw0) * vHD + vh * 32 + (lane & 31);
            if (a.message) { o_message[o] = vA; o_message[o + vHD] = vB; }
            if (a.acc_out) {   // separate multiply and add (:277-281)
                const float rA = pend_acc + vA * vwl, rB = pend_acc + vB * vwl;
                if (a.xflags & 2) {
                    __builtin_nontemporal_store(rA, o_acc + o);
                    __builtin_nontemporal_store(rB, o_acc + o + vHD);
                } else if (a.xflags & 4) {
                    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(o_acc + o), "v"(rA) : "memory");
                    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(o_acc + o + vHD), "v"(rB) : "memory");
                } else {
                    o_acc[o] = rA;
                    o_acc[o + vHD] = rB;
                }
            }
            if constexpr (EXACT) {
                const int f = lane >> 4, j = lane & 15, rank = (j & 1) * 8 + (j >> 1);
                if (rank < vtopk) {
                    const size_t lt = (size_t)pend_l00 + (f >> 1) * vw0 + (f & 1);          // token within the pair
                    const size_t lf = (size_t)pend_b * vL + lt;
                    if (a.topk_tab) o_tab[(((size_t)pend_b * vH + vh) * vL + lt) * vtopk + rank] = pend_idx;
                    if (a.topk_idx) o_idx[(lf * vtopk + rank) * vH + vh] = pend_idx;
                    if (a.topk_score) o_score[(lf * vtopk + rank) * vH + vh] = pend_sc;
                }
            }
        }
        have_pend = false;
    };
    issue(0, I0{}, I0{}, it_cur.b);
    issue(0, I0{}, I1{}, it_cur.b);
    bool more = tn < total;
    if (more) item_of(tn, it_nx);
    for (;;) {
        const int b = it_cur.b, l00 = it_cur.l00, bn = it_nx.b;
        int claim_ret = 0, claimed = 0;   // this item's claim (the item after it_nx): issued in K pass 0, collected behind the first V wait
        const float* qs = stg + cbuf * STG;
        const int* t2 = reinterpret_cast<const int*>(qs + 128);
        // ================================================================== K passes: logits of candidates 64p .. 64p+63
        f32x4 lg[NPASS];
        static_for<0, NPASS>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            glds_wait<0>();
            if constexpr (p == 0) {
                flush();
                if (more) prefetch(it_nx, cbuf ^ 1);   // lands under this item's K pass and softmax; consumed at its chunk NV - 2
                work_claim_issue(ctr, more && ctr != nullptr && rem == 0, claim_ret);   // (unconditional statement; EXEC = 0 unless a run ends)
                acc_cur = a.acc_in ? qs[160 + (lane & 31)] : 0.f;   // final[parent] of the item for d = lane % 32 (:277)
            }
            f32x4 qa[8], kr[8];   // operand A: lane l holds q[child l%4][d]; operand B: this lane's candidate row
#pragma unroll
            for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qs + (lane & 3) * 32 + ((u ^ ((lane & 3) >> 1)) * 4));
#pragma unroll
            for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ring) + rd[u]);
            lds_reads_done();
            if constexpr (p + 1 < NPASS) {   // the ring is free again: next pass of K
                issue(0, std::integral_constant<int, (p + 1 < NPASS ? p + 1 : 0)>{}, I0{}, b);
                issue(0, std::integral_constant<int, (p + 1 < NPASS ? p + 1 : 0)>{}, I1{}, b);
            } else {                         // ... or the first two chunks of V
                issue(1, I0{}, I0{}, b);
                issue(1, I0{}, I1{}, b);
            }
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (EXACT) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c, 0, 0, 0);
                }
            } else {
                // no index depends on these logits (finest level): four interleaved partial d-chains instead of one sequential chain
                f32x4 c4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) c4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c4[0], 0, 0, 0);
                    c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c4[1], 0, 0, 0);
                    c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c4[2], 0, 0, 0);
                    c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c4[3], 0, 0, 0);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) c[f] = (c4[0][f] + c4[1][f]) + (c4[2][f] + c4[3][f]);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) lg[p][f] = a.temp * c[f];
            asm volatile("" : "+v"(lg[p]));   // keep the pass's arithmetic inside the pass
        });
        // ================================================================== softmax (+ top-k), one series (child) per 16-lane row
        // (round 5: a softmax straight from the K pass's lane <-> candidate layout -- whole-wave reductions by DPP + permlane swaps, no
        //  transposition through LDS, one fence instead of three -- measured 215 against 187 us per launch at the finest level: the
        //  eight 64-lane reductions are a longer dependent chain than the LDS round trips they replace)
        softmax_select<NPASS, EXACT, FULL>(a, lg, Pld, t2, lane, K, w1p, pend_sc, pend_idx);
        // ================================================================== V chunks; then the next item's K pass 0
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<0, NV>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            // in flight behind chunk c: chunk c + 1 (or the next item's first K chunk) = 4 instructions; the staging DMA of the next
            // item is older than every V chunk, so it has landed behind the first of these waits
            if (c + 1 < NV || more) glds_wait<4>(); else glds_wait<0>();
            if constexpr (c == 0) claimed = work_claimed(claim_ret);   // the claim is older than the eight instructions issued behind it
            if constexpr (c == NV - 2) {
                if (more) stage_in(cbuf ^ 1);   // every chunk of this item has been issued: the offsets become the next item's
            }
            f32x4 pv[4];   // operand A of MFMA mm: P[child lane%4][parity lane/32][16 c + mm]
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = *reinterpret_cast<const f32x4*>(pa + 16 * c + 4 * i);
            float vb[16];
            const char* sb = reinterpret_cast<const char*>(ring) + (c & 1) * 4096;
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) vb[mm] = *reinterpret_cast<const float*>(sb + va[mm & 7] + mm * 256);
            lds_reads_done();
            // the slot is free: chunk c + 2 of V, or the next item's K pass 0
            if constexpr (c + 2 < NV) {
                issue(1, std::integral_constant<int, ((c + 2) >> 1)>{}, std::integral_constant<int, (c & 1)>{}, b);
            } else {
                if (more) issue(0, I0{}, std::integral_constant<int, (c & 1)>{}, bn);
            }
#pragma unroll
            for (int mm = 0; mm < 16; ++mm)
                acc[mm & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(pv[mm >> 2][mm & 3], vb[mm], acc[mm & 3], 0, 0, 0);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        });
        {
            f32x4 tot;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = (acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c]);
                const unsigned xi = __float_as_uint(x);
                const auto sw = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // lanes l and l ^ 32
                tot[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            pend = tot; pend_acc = acc_cur; pend_b = b; pend_l00 = l00; have_pend = true;
        }
        if (!more) break;
        it_cur = it_nx; cbuf ^= 1;
        if (!ctr) tn += stride;
        else if (rem > 0) { ++tn; --rem; }
        else { tn = 2 * stride + claimed * a.claim; rem = a.claim - 1; }   // counter value c = items 2 stride + c claim .. + claim - 1
        more = tn < total;
        if (more) item_of(tn, it_nx);
    }
    glds_wait<0>();
    flush();
    if (ctr && lane == 0) work_leave(ctr, stride);
}

template <int NPASS, bool EXACT, bool FULL>
static int launch_fine_quad(const FineQArgs& a, hipStream_t s) {
    constexpr size_t lds = 2 * sizeof(float) * (2048 + 8 * (32 * NPASS + 4) + 2 * 256);
    // persistent grid: exactly the workgroups that are resident at once
    static int resident[CASMTR_MAX_DEVICES] = {0};
    int res = 0;
    if (const int r = resident_workgroups(resident, fine_quad_kernel<NPASS, EXACT, FULL>, 128, lds, &res)) return r;
    // Waves per XCD.  An XCD walks the pairs one after the other, its 4 MB L2 holding one (pair, head) slice of K and V at a time.
    // While the first waves are already on the next pair and the last ones still on this one, two slices compete for the cache; the
    // share of the time spent like that is (waves in flight) / (items per pair).  Measured on the finest CasMTR-4c level (2.8 MB
    // per slice, 2704 items per pair and XCD; TCC_MISS per launch / us per launch): 452 waves 6.1-9.4 M / 220, 320 waves 4.4 M / 192,
    // 256 waves 3.5 M / 200 (3.0 M are compulsory).  So when two slices do not fit the L2 together, the wave count is capped at an
    // eighth of a pair's items (and chosen so that every wave makes the same number of items per pair: no wave runs a round ahead).
    // Small slices (the middle level: 0.7 MB) run at full residency.
    const int G = 8 / a.H;                                             // XCDs sharing a head split the pair's quads
    const long long per_pair = (a.nquads + G - 1) / G;
    long long wpx = (long long)res / 8 * 2;                            // resident waves per XCD
    static int cu_count[CASMTR_MAX_DEVICES] = {0};
    int ncu = 0;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CASMTR_MAX_DEVICES) return CASMTR_ERR_UNSUPPORTED;
        ncu = __atomic_load_n(&cu_count[dev], __ATOMIC_RELAXED);
        if (!ncu) {
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return CASMTR_ERR_UNSUPPORTED;
            __atomic_store_n(&cu_count[dev], ncu, __ATOMIC_RELAXED);
        }
    }
    const long long even = ncu >= 8 ? 2ll * (ncu / 8) : 2;             // one more workgroup (2 waves) on every CU of an XCD
    const char* ev = getenv("CASMTR_FQ_WAVES_PER_XCD");                // measurement knob (tools/fq_sweep.py)
    if (ev && atoi(ev) > 0) wpx = atoi(ev) < wpx ? atoi(ev) : wpx;
    else if (2ull * 2 * a.lq1 * 512 > 3ull << 20) {                    // two K + V slices against 3 of the L2's 4 MB
        long long rounds = (per_pair + wpx - 1) / wpx;
        if (rounds < 8) rounds = 8;
        wpx = (per_pair + rounds - 1) / rounds;
        // ... and the same number of workgroups on every CU: the items are dealt out statically, so the CUs that hold one workgroup
        // more run all their waves slower and finish last (round 4, 104x104, B = 8: 338 waves per XCD 226 us, 320 = 5 workgroups on
        // each of the 32 CUs 193 us, 256 192 us, 352 221 us, 384 218 us)
        if (wpx > even) wpx = wpx / even * even;
    }
    if (wpx > per_pair * a.B) wpx = per_pair * a.B;
    const long long blocks = (wpx + 1) / 2 * 8;
    if (getenv("CASMTR_FQ_DEBUG")) fprintf(stderr, "fine_quad<%d,%d>: %zu B LDS per workgroup, %d resident workgroups, launching %lld\n", NPASS, (int)EXACT, lds, res, blocks);
    prof_symbol_args(NPASS == 1 ? CASMTR_PROF_QTA_FINE : CASMTR_PROF_QTA_FINE2, "<%d,%s,%s>", NPASS, EXACT ? "true" : "false", FULL ? "true" : "false");
    CASMTR_LAUNCH_TIMED(NPASS == 1 ? CASMTR_PROF_QTA_FINE : CASMTR_PROF_QTA_FINE2, (fine_quad_kernel<NPASS, EXACT, FULL>), dim3((unsigned)blocks),
                        dim3(128), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" int casmtr_qta_fine_level_quad_fwd(const float* q, const float* key, const float* value, const int32_t* parents, float temp,
                                              int topk, float w_level, const float* acc_in, float* message, float* acc_out,
                                              int32_t* topk_tab, float* topk_score, int64_t* topk_idx, int B, int h0, int w0, int h1,
                                              int w1, int H, int D, int Kp, casmtr_stream_t stream) {
    const int K = 4 * Kp;
    if (D != 32 || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1) || Kp < 1 || Kp > 32 || topk > K || topk > 16 ||
        (H != 8 && H != 4 && H != 2 && H != 1) || (long long)(h1 / 2) * (w1 / 2) >= (1 << 22) ||
        (long long)(h1 / 2) * (w1 / 2) * (w1 / 2) >= (1ll << 32))   // umulhi(p, ceil(2^32 / d)) == p / d needs p * d < 2^32 (p < lq1, d = w1 / 2)
        return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || h0 <= 0 || w0 <= 0) return 0;
    {   // the staging DMA addresses q / parents / acc_in with 32-bit byte offsets from the tensor base
        const long long lq0 = (long long)(h0 / 2) * (w0 / 2);
        if ((long long)B * H * lq0 * 512 >= (1ll << 32) || (long long)B * H * lq0 * Kp * 4 >= (1ll << 32)) return CASMTR_ERR_UNSUPPORTED;
    }
    FineQArgs a{};
    a.q = q; a.key = key; a.value = value; a.parents = parents; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out;
    a.topk_tab = topk_tab; a.topk_score = topk_score; a.topk_idx = topk_idx; a.temp = temp; a.w_level = w_level; a.topk = topk;
    a.B = B; a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.H = H; a.Kp = Kp; a.nquads = (h0 / 2) * (w0 / 2); a.lq1 = (h1 / 2) * (w1 / 2);
    const unsigned d = (unsigned)(w1 / 2);
    a.div_magic = d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u;
    { const char* ev = getenv("CASMTR_FQ_FLAGS"); a.xflags = ev ? atoi(ev) : 0; }
    {   // item index -> (pair, quad) and quad -> (row, column) by multiply-high: exact for every index the kernel can meet (else: static schedule)
        const int G = 8 / H, chunk = (a.nquads + G - 1) / G, last = a.nquads - (G - 1) * chunk;
        const unsigned long long nmax = (unsigned long long)B * (unsigned long long)(chunk > 0 ? chunk : 1) + 2;
        auto magic = [](unsigned d, unsigned long long nmax_, unsigned* m) {   // n / d == umulhi(n, m) for n <= nmax_ (m = 0: d == 1)
            if (d <= 1) { *m = 0; return true; }
            const unsigned long long mm = (0x100000000ull + d - 1) / d, e = mm * d - 0x100000000ull;
            if (mm > 0xFFFFFFFFull || nmax_ * e >= 0x100000000ull) return false;
            *m = (unsigned)mm;
            return true;
        };
        const bool ok = magic((unsigned)chunk, nmax, &a.magic_chunk) && magic((unsigned)(last > 0 ? last : 1), nmax, &a.magic_last) &&
                        magic((unsigned)(w0 / 2), (unsigned long long)a.nquads + 1, &a.magic_wq);
        if (!ok) return CASMTR_ERR_UNSUPPORTED;   // (grids of more than ~10^5 quads per XCD chunk: no shipped configuration)
        const char* ev = getenv("CASMTR_FQ_DYNAMIC");
        a.ctr = (ev && ev[0] == '0') ? nullptr : work_counters((hipStream_t)stream);
        const char* ec = getenv("CASMTR_FQ_CLAIM");
        // isolated launches, B = 8 (profiles/r06_dyn_ab.txt): finest level static 211-236 us, 1 / 2 / 4 / 8 / 16 items per claim 360 / 225 /
        // 208-220 / 215-220 / 232-262; middle level static 112-126, 118 / 108-111 / 111-120 / 124-132 / 164-175
        a.claim = ec && atoi(ec) > 0 ? atoi(ec) : (K <= 64 ? 4 : 2);
    }
    hipStream_t s = (hipStream_t)stream;
    const bool full = K == 64 || K == 128;
    if (topk > 0) {
        if (K <= 64) return full ? launch_fine_quad<1, true, true>(a, s) : launch_fine_quad<1, true, false>(a, s);
        return full ? launch_fine_quad<2, true, true>(a, s) : launch_fine_quad<2, true, false>(a, s);
    }
    if (K <= 64) return full ? launch_fine_quad<1, false, true>(a, s) : launch_fine_quad<1, false, false>(a, s);
    return full ? launch_fine_quad<2, false, true>(a, s) : launch_fine_quad<2, false, false>(a, s);
}
