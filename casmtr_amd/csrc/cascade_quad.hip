// CascadeQTAttB.forward (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:400-452) on QUAD-MAJOR operands,
// two query quads per work item where their windows allow it: the round-3 cascade attention kernel (5 x 5 windows, dilation 1).
//
// A query quad attends to the 4 children of each of its 25 window cells (:419-429).  CascadeFeatureTransformer.get_window_warp_idx
// (src/model/modules/transformer.py:416-440) makes those cells a 5 x 5 block of the coarse grid around the quad's coarse match, and
// coarse matches of neighbouring cells mostly move together, so the windows of the horizontally adjacent quads (2m, 2m+1) are usually
// the same block or one column apart.  Such a pair shares ONE 5 x 6 box: 30 cells = 120 candidate rows are gathered once (LDS-DMA,
// 512-byte runs of the quad-major layout: a box row is one contiguous 3 KB piece per head) and both quads' 8 queries run against
// them -- 42 % fewer gathered bytes and half the per-item pipeline latency per quad than one item per quad (the round-2 kernel,
// cascade_dma.hip, which is at the L2 -> LDS gather ceiling of its 128-byte token-major rows).  Each quad masks the box columns
// outside its own window; probabilities of masked candidates are exactly 0.  Pairs that cannot share (different rows, columns further
// apart, irregular position lists, last quad of an odd row) run as two single-quad sub-items with their own 25 cells, same code.
// No index depends on these logits (upsampled_idx is the window list itself), so the d-sum uses four interleaved partial chains
// and the result carries the 1e-4 softmax tolerance; rel_pos (indoor model) is added to the logits as in :438-441.
//
// Round 4: a wave can walk several heads of a sub-item before it moves on (unit = (sub-item, head); CASMTR_CQ_HEADS_PER_WAVE).  The window
// lists are shared by the heads (:419-429), so the per-item front end -- 2 x 25 int64 positions, the regularity test, the sharing decision,
// masks, 16 DMA offsets -- then runs once per sub-item instead of once per (sub-item, head): about a quarter of the kernel's VALU / SALU
// work at H = 4.  Only the queries differ between the heads of a sub-item: they arrive by one LDS-DMA instruction per unit (both quads'
// 512 bytes) in a double-buffered staging area.  It pays on isolated launches with smooth windows and not inside the real step (see
// the launcher), so the default stays one head per wave = XCD <-> head.
//
// Pipeline per unit (one head of one quad pair): fine_quad.hip's two-pass schedule -- K chunks (4 KB =
// 8 cells) through a two-slot ring, lane <-> candidate v_mfma_f32_4x4x1 logits for slot 0 and slot 1 against the SAME staged rows,
// softmax with one series per 16-lane DPP row, V chunks with the probabilities as operand A, next sub-item's K under the last V chunks.
#include <stdio.h>
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct CasQArgs {
    const float* q;        // [B,H,Lq0,4,32]
    const float* key;      // [B,H,Lq1,4,32]
    const float* value;    // [B,H,Lq1,4,32]
    const int64_t* tp;     // [B,Lq0,25,2] (row, col) on the (h1/2) x (w1/2) grid
    const float* rel;      // nullable [B,H,L,100]
    float* message;        // [B,L,H*32]
    float temp;
    int B, h0, w0, h1, w1, H, nquads, lq1, npr, nitems;   // npr = pair items per quad row, nitems = pair items per image pair
    int hw;                // heads a wave walks per sub-item (divides H): the XCDs split into H / hw head groups x 8 hw / H item chunks
    int colmajor;          // item order inside an XCD's chunk: 1 = down the columns of quad pairs (consecutive items of a wave's neighbours
                           // share 4 of their 5 window rows), 0 = along the rows
    int* ctr;              // nullable: work_counters() -- items beyond a wave's first two are CLAIMED (atomic counter per XCD) instead of dealt
                           // out with a fixed stride: the waves of an XCD then always work on one compact front of ~2 x (waves per XCD)
                           // consecutive items (the window boxes of a front share the L2), and a wave that drew expensive items (incoherent
                           // windows: 25.6 KB straight from HBM) simply takes fewer
    int claim;             // consecutive items per claim (an L2 atomic on one address costs ~16 ns: see fine_quad.hip)
};

struct Sub {   // one sub-item: the candidate rows of `ncells` cells against nq query quads (all wave-uniform)
    int b, l00_0, nq, ncells;   // slot 1's quad is the right-hand neighbour: first token l00_0 + 2
    unsigned mask0, mask1;   // bit e: cell e belongs to slot 0's / slot 1's window
    int quadA, hasB, qslot;  // the item's left quad, whether it has a right neighbour, staging slot of query slot 0 (1: the right quad runs alone)
};

template <bool HAS_REL>
__global__ __launch_bounds__(128, (HAS_REL ? 2 : 3)) void cascade_quad_kernel(const CasQArgs a) {
    constexpr int KW = 25, KS = 128 + 4, PST = 64 + 4, SLOT_FLOATS = 8 * PST;   // per query slot: P[child][parity][64] / logits [4][KS]
    static_assert(SLOT_FLOATS >= 4 * KS, "the transposition buffer aliases the probabilities");
    constexpr int QBUF = 256;             // one unit's queries: [2 query slots][4 children][32 d], 16-byte units XOR-swizzled by (child >> 1)
    constexpr int WAVE_FLOATS = 2048 + 2 * SLOT_FLOATS + 2 * QBUF + 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ring = smem + wave * WAVE_FLOATS;                // 2 slots x 32 rows x 128 B (XOR-swizzled 16-byte units)
    float* Pld = ring + 2048;                               // [2 query slots][SLOT_FLOATS]
    float* qst = Pld + 2 * SLOT_FLOATS;                     // [2 units][QBUF]: queries of the current / the next unit (LDS-DMA)
    int* t2 = reinterpret_cast<int*>(qst + 2 * QBUF);               // t2[parity * 16 + j] = cell 2j + parity
    const int H = a.H, HD = H * 32, L = a.h0 * a.w0, wq = a.w0 >> 1, Lq = a.nquads, h1p = a.h1 >> 1, w1p = a.w1 >> 1;
    // ---- work list: XCD x -> head group x % (H / hw) (heads h0 .. h0 + hw - 1), the XCDs sharing a group split every image pair's quad
    //      pairs into contiguous chunks; a wave walks the hw heads of a sub-item before it moves on
    const int HW = a.hw, ngrp = H / HW, xcd = blockIdx.x & 7, h0 = (xcd % ngrp) * HW, G = 8 / ngrp, g = xcd / ngrp;
    const int chunk = (a.nitems + G - 1) / G, cnt = min(chunk, a.nitems - g * chunk);
    const int total = cnt > 0 ? a.B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    int t = (blockIdx.x >> 3) * 2 + wave;
    int* const ctr = a.ctr ? a.ctr + xcd * WORK_XCD_INTS : nullptr;
    // dynamic claiming: every wave of this XCD's share of the grid reports in once when it leaves; the last one re-zeroes the pair
    auto leave = [&]() {
        if (ctr && lane == 0) work_leave(ctr, stride);
    };
    if (g >= G || t >= total) { leave(); return; }
    bool claim_pending = false;   // the item after the last prefetch has been claimed and not collected yet (then t == total)
    int rem = 0;                  // items left in the claimed run behind t
    for (int i = lane; i < 2048; i += 64) ring[i] = 0.f;    // rows of a short last chunk that no DMA ever wrote must be finite
    lds_reads_done();
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const int un = lane & 7;
    unsigned cK[4];   // DMA source offset inside a cell's 512-byte run (row 8j + lane/8 of a pass, physical unit un <- logical un ^ ((row>>1)&7))
#pragma unroll
    for (int j = 0; j < 4; ++j)
        cK[j] = (unsigned)(((lane >> 3) & 3) * 128 + ((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072 - j * 1024);
    unsigned rd[8];   // K pass: byte offset of logical unit u of this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    unsigned va[8];   // V chunk: byte offset of V[row 2 mm + lane/32][d = lane%32] for mm % 8 == x, minus mm * 256
#pragma unroll
    for (int x = 0; x < 8; ++x) va[x] = (unsigned)((lane >> 5) * 128 + ((((lane & 31) >> 2) ^ x) * 16) + (lane & 3) * 4);
    const int paoff = ((lane & 3) * 2 + (lane >> 5)) * PST;   // operand A of the V chunks: P[child lane%4][parity lane/32][.]
    const size_t head_pitch = (size_t)a.lq1 * 128;                      // floats between two (pair, head) slices
    const float* const k0 = a.key - 768;                               // 3072 bytes low
    const float* const v0 = a.value - 768;
    const unsigned qst_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(qst));
    // query staging: lane l -> slot l / 32, child (l / 8) % 4, physical unit l % 8 <- logical unit (l % 8) ^ (child >> 1)
    const unsigned qsrc = (unsigned)(((lane >> 3) & 3) * 128 + (((lane & 7) ^ (((lane >> 3) & 3) >> 1)) * 16));

    // ---- item-level prefetch registers: window positions (lane e < 25: quad A, lane 32 + e: quad B) and queries (halves likewise)
    int pf_r = 0, pf_c = 0;
    int pf_b = 0, pf_quadA = 0, pf_l00A = 0;
    bool pf_hasB = false;
    bool regs_full = false, pendB = false;
    auto prefetch = [&](int tt) {
        const int b = tt / cnt, it = g * chunk + tt % cnt;
        const int hq = a.nitems / a.npr;
        const int qy = a.colmajor ? it % hq : it / a.npr, m = a.colmajor ? it / hq : it % a.npr;
        pf_b = b; pf_quadA = qy * wq + 2 * m; pf_l00A = 2 * qy * a.w0 + 4 * m; pf_hasB = 2 * m + 1 < wq;
        const int e = lane & 31, quad = pf_quadA + (lane >> 5);
        if (lane < 32 || pf_hasB) {
            if (e < KW) {
                const longlong2 rc = *reinterpret_cast<const longlong2*>(a.tp + (((size_t)b * Lq + quad) * KW + e) * 2);
                pf_r = (int)rc.x; pf_c = (int)rc.y;
            }
        }
        regs_full = true;
    };
    unsigned voff[2][8];
    Sub sub_cur{}, sub_nx{};
    // registers -> the next sub-item: mode decision, cells + queries to LDS, DMA offsets
    auto stage_in = [&]() {
        const int e = lane & 31;
        const int cl = min(max(pf_r, 0), h1p - 1) * w1p + min(max(pf_c, 0), w1p - 1);   // this lane's cell (valid for e < 25)
        int cellv;
        if (pendB) {        // second half of a pair that could not share: quad B alone
            const int last = __builtin_amdgcn_readlane(cl, 32 + KW - 1);
            cellv = e < KW ? cl : last;
            if (lane >= 32) t2[(e & 1) * 16 + (e >> 1)] = cellv;
            sub_nx = Sub{pf_b, pf_l00A + 2, 1, KW, (1u << KW) - 1u, 0u, pf_quadA, 1, 1};
            pendB = false; regs_full = false;
        } else {
            const int oyA = __builtin_amdgcn_readlane(pf_r, 0), oxA = __builtin_amdgcn_readlane(pf_c, 0);
            const int oyB = __builtin_amdgcn_readlane(pf_r, 32), oxB = __builtin_amdgcn_readlane(pf_c, 32);
            const int oy = lane < 32 ? oyA : oyB, ox = lane < 32 ? oxA : oxB;
            const bool ok = e >= KW || (pf_r == oy + e / 5 && pf_c == ox + e % 5);
            const unsigned long long bal = __ballot(ok);
            const bool regA = (unsigned)bal == 0xFFFFFFFFu && oyA >= 0 && oyA + 5 <= h1p && oxA >= 0 && oxA + 5 <= w1p;
            const bool regB = (unsigned)(bal >> 32) == 0xFFFFFFFFu && oyB >= 0 && oyB + 5 <= h1p && oxB >= 0 && oxB + 5 <= w1p;
            const int dx = oxA > oxB ? oxA - oxB : oxB - oxA;
            if (pf_hasB && regA && regB && oyA == oyB && dx <= 1) {   // one 5 x (5 + dx) box for both quads
                const int bx0 = min(oxA, oxB), bw = 5 + dx, nc = 5 * bw;
                const int ee = e < nc ? e : nc - 1;
                cellv = (oyA + ee / bw) * w1p + bx0 + ee % bw;
                if (lane < 32) t2[(e & 1) * 16 + (e >> 1)] = cellv;
                unsigned mA = 0, mB = 0;
#pragma unroll
                for (int r = 0; r < 5; ++r) { mA |= 0x1Fu << (bw * r + (oxA - bx0)); mB |= 0x1Fu << (bw * r + (oxB - bx0)); }
                sub_nx = Sub{pf_b, pf_l00A, 2, nc, mA, mB, pf_quadA, 1, 0};
                regs_full = false;
            } else {                                                   // quad A alone now; quad B (if any) as the next sub-item
                const int last = __builtin_amdgcn_readlane(cl, KW - 1);
                cellv = e < KW ? cl : last;
                if (lane < 32) t2[(e & 1) * 16 + (e >> 1)] = cellv;
                sub_nx = Sub{pf_b, pf_l00A, 1, KW, (1u << KW) - 1u, 0u, pf_quadA, (int)pf_hasB, 0};
                pendB = pf_hasB; regs_full = pf_hasB;
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int4 p4 = *reinterpret_cast<const int4*>(t2 + (lane >> 5) * 16 + 4 * i);
            voff[i >> 1][(i & 1) * 4 + 0] = ((unsigned)p4.x << 9) + cK[0];
            voff[i >> 1][(i & 1) * 4 + 1] = ((unsigned)p4.y << 9) + cK[1];
            voff[i >> 1][(i & 1) * 4 + 2] = ((unsigned)p4.z << 9) + cK[2];
            voff[i >> 1][(i & 1) * 4 + 3] = ((unsigned)p4.w << 9) + cK[3];
        }
    };
    // chunk c (cells 8c .. 8c+7) of pass p of K (isv = 0) or V (isv = 1) -> ring slot c; the last chunk (cells 24 ..) is short
    auto issue = [&](int isv, auto pc, auto cc, const Sub& s, int hh) {
        constexpr int p = decltype(pc)::value, c = decltype(cc)::value;
        const int sb = __builtin_amdgcn_readfirstlane(s.b);               // wave-uniform by construction; tell the compiler
        const int hu = __builtin_amdgcn_readfirstlane(hh);
        const float* base = (isv ? v0 : k0) + ((size_t)sb * H + hu) * head_pitch;
        const unsigned dst = ring_lds + (unsigned)(c * 4096);
        if constexpr (p == 1 && c == 1) {
            const int n3 = __builtin_amdgcn_readfirstlane((s.ncells - 24 + 1) >> 1);
            if (n3 >= 4) glds_chunk(base, voff[p][4], voff[p][5], voff[p][6], voff[p][7], dst);
            else if (n3 == 3) glds_chunk3(base, voff[p][4], voff[p][5], voff[p][6], dst);
            else if (n3 == 2) glds_chunk2(base, voff[p][4], voff[p][5], dst);
            else glds_chunk1(base, voff[p][4], dst);
        } else {
            glds_chunk(base, voff[p][4 * c + 0], voff[p][4 * c + 1], voff[p][4 * c + 2], voff[p][4 * c + 3], dst);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // the queries of both quads of item (b, quadA) for head hh -> staging buffer `buf` (one instruction: lanes 0-31 the left quad,
    // 32-63 the right one, or the left one again when there is none)
    auto issue_q = [&](int b, int quadA, int hasB, int hh, int buf) {
        const int row0 = __builtin_amdgcn_readfirstlane((b * H + hh) * Lq + quadA), hb = __builtin_amdgcn_readfirstlane(hasB);
        const unsigned off = (unsigned)((row0 + ((lane >> 5) & hb)) * 512) + qsrc + 3072u;
        glds_chunk1(a.q - 768, off, qst_lds + (unsigned)(buf * QBUF * 4));
    };
    prefetch(t);
    t += stride;
    stage_in();
    sub_cur = sub_nx;
    if (!regs_full && t < total) {
        prefetch(t);
        if (ctr) {   // the first claim is waited for on the spot (once per wave); items 0 .. 2 stride - 1 are the waves' static first two
            int r0;
            work_claim_issue(ctr, true, r0);
            glds_wait<0>();
            t = work_claimed(r0) * a.claim + 2 * stride;
            rem = a.claim - 1;
        } else t += stride;
    }
    int hcur = h0, ubuf = 0;   // head of the current unit, its query staging buffer
    issue_q(sub_cur.b, sub_cur.quadA, sub_cur.hasB, h0, 0);
    // results of the previous unit: stored right behind the next one's first DMA wait
    f32x4 pend[2];
    Sub pend_sub{};
    int pend_h = 0;
    bool have_pend = false;
    auto flush = [&]() {
        if (have_pend) {
            const int hi = lane >> 5;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl < pend_sub.nq) {
                    const float vA = hi ? pend[sl][2] : pend[sl][0], vB = hi ? pend[sl][3] : pend[sl][1];
                    const size_t o = ((size_t)pend_sub.b * L + (pend_sub.l00_0 + 2 * sl) + hi * a.w0) * HD + pend_h * 32 + (lane & 31);
                    a.message[o] = vA;
                    a.message[o + HD] = vB;
                }
            }
        }
        have_pend = false;
    };
    issue(0, I0{}, I0{}, sub_cur, h0);
    issue(0, I0{}, I1{}, sub_cur, h0);
    for (;;) {
        const Sub s = sub_cur;
        const int h = hcur;
        const bool same = h + 1 < h0 + HW;                 // the next unit is the next head of this sub-item: same cells, masks, offsets
        const bool more_sub = regs_full || pendB;          // another sub-item follows (its item's identity is in the prefetch registers)
        const bool more = same || more_sub;
        const int hn = same ? h + 1 : h0;
        const float* qcur = qst + ubuf * QBUF + s.qslot * 128 + (lane & 3) * 32;
        const int qsw = (lane & 3) >> 1;
        const int n3 = (s.ncells - 24 + 1) >> 1;          // DMA instructions of the short last chunk
        int claim_ret = 0;                                // this unit's claim (work_claim_issue under chunk 2, collected behind chunk 3's wait)
        const int nrow3 = 4 * s.ncells - 96;              // its valid rows
        // ================================================================== K passes: logits of rows 64p .. 64p+63 for both query slots
        float relv[HAS_REL ? 2 : 1][2][4];
        static_for<0, 2>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            glds_wait<0>();
            if constexpr (p == 0) {
                flush();
                // the next unit's queries: a whole unit ahead of their use (they come from HBM; everything issued later waits behind them)
                if (same) issue_q(s.b, s.quadA, s.hasB, hn, ubuf ^ 1);
                else if (more_sub) issue_q(pf_b, pf_quadA, (int)pf_hasB, h0, ubuf ^ 1);
                if constexpr (HAS_REL) {   // :438-441; candidate index within the quad's own list = rank of the cell in its window * 4 + child
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            const int k = 64 * pp + lane, cell = k >> 2;
                            const unsigned mk = sl ? s.mask1 : s.mask0;
                            const bool valid = sl < s.nq && ((mk >> cell) & 1u);
                            const int kq = __popc(mk & ((1u << cell) - 1u)) * 4 + (k & 3);
                            const int l00 = s.l00_0 + 2 * sl;
#pragma unroll
                            for (int f = 0; f < 4; ++f)
                                relv[HAS_REL ? sl : 0][pp][f] =
                                    valid ? a.rel[(((size_t)s.b * H + h) * L + l00 + (f >> 1) * a.w0 + (f & 1)) * (4 * KW) + kq] : 0.f;
                        }
                }
            }
            f32x4 kr[8];   // operand B: this lane's candidate row
#pragma unroll
            for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ring) + rd[u]);
            lds_reads_done();
            if constexpr (p == 0) {          // the ring is free again: K pass 1
                issue(0, I1{}, I0{}, s, h);
                issue(0, I1{}, I1{}, s, h);
            } else {                         // ... or the first two chunks of V
                issue(1, I0{}, I0{}, s, h);
                issue(1, I0{}, I1{}, s, h);
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl == 1 && s.nq < 2) break;     // wave-uniform
                f32x4 c4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) c4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {   // operand A in two halves of 16 d (16 VGPRs instead of 32): lane l holds q[slot][child l%4][d]
                    f32x4 qa[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qcur + sl * 128 + (((4 * hf + u) ^ qsw) * 4));
                    lds_reads_done();
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[4 * hf + u].x, c4[0], 0, 0, 0);
                        c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[4 * hf + u].y, c4[1], 0, 0, 0);
                        c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[4 * hf + u].z, c4[2], 0, 0, 0);
                        c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[4 * hf + u].w, c4[3], 0, 0, 0);
                    }
                    asm volatile("" : "+v"(c4[0]), "+v"(c4[1]), "+v"(c4[2]), "+v"(c4[3]));
                }
                // logits -> the slot's transposition buffer [4 children][KS] (the probabilities of the previous sub-item are dead:
                // its last V chunk has been consumed)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float x = a.temp * ((c4[0][f] + c4[1][f]) + (c4[2][f] + c4[3][f]));
                    if constexpr (HAS_REL) x = x + relv[HAS_REL ? sl : 0][p][f];
                    Pld[sl * SLOT_FLOATS + f * KS + 64 * p + lane] = x;
                }
            }
        });
        // ================================================================== softmax per (slot, child): one series per 16-lane row
        {
            const int f = lane >> 4, j = lane & 15;
            wave_lds_fence();
            float ps[2][8];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl < s.nq) {
                    const unsigned mk = sl ? s.mask1 : s.mask0;
                    const bool vlo = (mk >> (2 * j)) & 1u, vhi = (mk >> (2 * j + 1)) & 1u;   // candidates 8j .. 8j+3 | 8j+4 .. 8j+7
                    const f32x4* sp = reinterpret_cast<const f32x4*>(Pld + sl * SLOT_FLOATS + f * KS + j * 8);
                    const f32x4 x0 = sp[0], x1 = sp[1];
                    float fm = -3.0e38f;
                    if (vlo) fm = fmaxf(fmaxf(x0.x, x0.y), fmaxf(x0.z, x0.w));
                    if (vhi) fm = fmaxf(fm, fmaxf(fmaxf(x1.x, x1.y), fmaxf(x1.z, x1.w)));
                    const float m = row16_max_f32(fm);
                    float sum = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ps[sl][e] = vlo ? __expf(x0[e] - m) : 0.f; sum += ps[sl][e]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ps[sl][4 + e] = vhi ? __expf(x1[e] - m) : 0.f; sum += ps[sl][4 + e]; }
                    sum = __builtin_amdgcn_rcpf(row16_sum_f32(sum));
#pragma unroll
                    for (int e = 0; e < 8; ++e) ps[sl][e] *= sum;
                }
            }
            wave_lds_fence();   // every lane has its logits: the buffers become P (candidate 8j + e -> P[f][e & 1][4j + e/2])
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl < s.nq) {
                    float* P = Pld + sl * SLOT_FLOATS;
                    *reinterpret_cast<f32x4*>(P + (f * 2 + 0) * PST + 4 * j) = (f32x4){ps[sl][0], ps[sl][2], ps[sl][4], ps[sl][6]};
                    *reinterpret_cast<f32x4*>(P + (f * 2 + 1) * PST + 4 * j) = (f32x4){ps[sl][1], ps[sl][3], ps[sl][5], ps[sl][7]};
                }
            }
            wave_lds_fence();
        }
        // ================================================================== V chunks; then the next sub-item's K pass 0
        f32x4 acc[2][2];   // two interleaved chains per slot
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c == 2) {
                if (!same && more_sub) stage_in();   // last head, every chunk issued: the offsets become the next sub-item's
            }
            // in flight behind chunk c: chunk c + 1 (the short one behind chunk 2), or the next sub-item's first K chunk
            if constexpr (c == 2) glds_wait_dyn(n3);
            else if constexpr (c == 3) { if (more) glds_wait<4>(); else glds_wait<0>(); }
            else glds_wait<4>();
            if constexpr (c == 3) {   // the claim issued under chunk 2 is older than the (at most four) instructions still in flight
                const int v = work_claimed(claim_ret);
                if (claim_pending) { t = v * a.claim + 2 * stride; rem = a.claim - 1; claim_pending = false; }
            }
            if constexpr (c == 2) {
                const bool pf = !same && more_sub && !regs_full && t < total;
                if (pf) prefetch(t);   // issued behind the wait
                // ... and the item after it is claimed now, a whole unit before it is needed (dynamic schedule; unconditional statement)
                work_claim_issue(ctr, pf && ctr != nullptr && rem == 0, claim_ret);
                if (pf) {
                    if (!ctr) t += stride;
                    else if (rem > 0) { ++t; --rem; }
                    else { claim_pending = true; t = total; }
                }
            }
            const int nmm = c < 3 ? 16 : (nrow3 >> 1);   // valid row pairs of this chunk
            float vb[16];
            const char* sb = reinterpret_cast<const char*>(ring) + (c & 1) * 4096;
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) vb[mm] = *reinterpret_cast<const float*>(sb + va[mm & 7] + mm * 256);
            f32x4 pv[2][4];   // operand A of MFMA mm: P[slot][child lane%4][parity lane/32][16 c + mm]
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (sl < s.nq) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) pv[sl][i] = *reinterpret_cast<const f32x4*>(Pld + sl * SLOT_FLOATS + paoff + 16 * c + 4 * i);
                }
            lds_reads_done();
            // the slot is free: the next V chunk, or the next sub-item's K pass 0
            if constexpr (c == 0) issue(1, I1{}, I0{}, s, h);
            else if constexpr (c == 1) issue(1, I1{}, I1{}, s, h);
            else if constexpr (c == 2) { if (more) issue(0, I0{}, I0{}, same ? s : sub_nx, hn); }
            else { if (more) issue(0, I0{}, I1{}, same ? s : sub_nx, hn); }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl < s.nq) {
#pragma unroll
                    for (int m4 = 0; m4 < 4; ++m4) {
                        if (4 * m4 < nmm) {   // wave-uniform; rows beyond the last cell were never loaded
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int mm = 4 * m4 + i;
                                // a partially valid group (25 cells: rows 96..99 = mm 0, 1): the other rows carry probability exactly 0
                                acc[sl][i & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(pv[sl][m4][i], vb[mm], acc[sl][i & 1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        });
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl < s.nq) {
                f32x4 tot;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = acc[sl][0][c] + acc[sl][1][c];
                    const unsigned xi = __float_as_uint(x);
                    const auto sw = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // lanes l and l ^ 32
                    tot[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
                pend[sl] = tot;
            }
        }
        pend_sub = s; pend_h = h; have_pend = true;
        if (!more) break;
        if (!same) sub_cur = sub_nx;
        hcur = hn; ubuf ^= 1;
    }
    glds_wait<0>();
    flush();
    leave();
}

template <bool HAS_REL>
static int launch_cas_quad(const CasQArgs& a, hipStream_t s) {
    constexpr size_t lds = 2 * sizeof(float) * (2048 + 2 * 8 * 68 + 2 * 256 + 32);
    static int resident[CASMTR_MAX_DEVICES] = {0};
    int res = 0;
    if (const int r = resident_workgroups(resident, cascade_quad_kernel<HAS_REL>, 128, lds, &res)) return r;
    const int G = 8 / (a.H / a.hw);
    const long long per_pair = (a.nitems + G - 1) / G;
    long long wpx = (long long)res / 8 * 2;                            // resident waves per XCD
    const char* ev = getenv("CASMTR_CQ_WAVES_PER_XCD");                // measurement knob
    if (ev && atoi(ev) > 0) wpx = atoi(ev) < wpx ? atoi(ev) : wpx;
    if (wpx > per_pair * a.B) wpx = per_pair * a.B;
    const long long blocks = (wpx + 1) / 2 * 8;
    if (getenv("CASMTR_FQ_DEBUG")) fprintf(stderr, "cascade_quad<%d>: %zu B LDS per workgroup, %d resident workgroups, launching %lld\n", (int)HAS_REL, lds, res, blocks);
    prof_symbol_args(CASMTR_PROF_CASCADE_ATTN, "<%s>", HAS_REL ? "true" : "false");
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_CASCADE_ATTN, (cascade_quad_kernel<HAS_REL>), dim3((unsigned)blocks), dim3(128), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" int casmtr_cascade_attn_quad_fwd(const float* q, const float* key, const float* value, const int64_t* topk_pos,
                                            const float* rel_pos, float temp, float* message, int B, int h0, int w0, int h1, int w1,
                                            int nhead, int D, int KW, casmtr_stream_t stream) {
    if (D != 32 || KW != 25 || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1) || h1 < 10 || w1 < 10 ||
        (nhead != 8 && nhead != 4 && nhead != 2 && nhead != 1) || (long long)(h1 / 2) * (w1 / 2) >= (1 << 22))
        return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || h0 <= 0 || w0 <= 0) return 0;
    CasQArgs a{};
    a.q = q; a.key = key; a.value = value; a.tp = topk_pos; a.rel = rel_pos; a.message = message; a.temp = temp;
    a.B = B; a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.H = nhead; a.nquads = (h0 / 2) * (w0 / 2); a.lq1 = (h1 / 2) * (w1 / 2);
    a.npr = (w0 / 2 + 1) / 2; a.nitems = (h0 / 2) * a.npr;
    // item order inside an XCD's chunk.  With dynamic claiming (round 6) along the rows: 480 against 580 us per launch on smooth windows, the
    // same on the bench's mix (profiles/r06_cq_fields.txt); the static schedule of rounds 3-5 measured the two orders alike.
    { const char* ev = getenv("CASMTR_CQ_ORDER"); a.colmajor = ev ? ev[0] == 'c' : 0; }
    // heads per wave: the front end (positions, sharing decision, masks, DMA offsets) is amortised over them, but an XCD's L2 then
    // holds that many heads' window neighbourhoods at once (~1.3 MB each at 208 x 208 with 320 waves per XCD).  Measured (round 4):
    // isolated launches on perfectly smooth windows 519 -> 466 us with all 4 heads per wave; inside the bench step (windows from the
    // real coarse matches, 8-15 % of them incoherent) 12.80 / 12.81 / 12.85 ms per step at 1 / 2 / 4 heads per wave, CasMTR-2c 21.4 /
    // 22.0 / 22.0 ms: there the kernel is bound by its memory traffic, not by its instruction stream.  Default: one head per wave
    // (XCD <-> head, the round-3 mapping).
    a.hw = 1;
    { const char* ev = getenv("CASMTR_CQ_DYNAMIC"); a.ctr = (ev && ev[0] == '0') ? nullptr : work_counters((hipStream_t)stream); }
    { const char* ev = getenv("CASMTR_CQ_CLAIM"); a.claim = ev && atoi(ev) > 0 ? atoi(ev) : 1; }   // 1 / 2 / 4 / 8: 495 / 503 / 511 / 540 us (12 % random windows)
    { const char* ev = getenv("CASMTR_CQ_HEADS_PER_WAVE"); const int v = ev ? atoi(ev) : 0; if (v >= 1 && v <= nhead && nhead % v == 0 && (8 % (nhead / v)) == 0) a.hw = v; }
    return rel_pos ? launch_cas_quad<true>(a, (hipStream_t)stream) : launch_cas_quad<false>(a, (hipStream_t)stream);
}
