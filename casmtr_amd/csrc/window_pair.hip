// CascadeMatching.forward on implicit 5 x 5 windows, TWO query quads per work item where their windows allow it
// (src/model/functions/cascade_matching.py:63-161; window lists: src/model/modules/transformer.py:416-440).
//
// window_match_pos_kernel (window_dma.hip) gathers a quad's 100 candidate rows (512 B each at C = 128) in eight 8 KB stages and
// spends a DMA round trip per stage: it runs at two waves per SIMD with one stage of prefetch, and a stage's arithmetic (one
// dependent chain of 32 v_mfma_f32_4x4x1) is a fraction of that round trip.  The windows of the horizontally adjacent quads
// (2m, 2m+1) are the same 5 x 5 block of coarse cells or one column apart for 85-92 % of the pairs of every bench configuration
// (tools/window_coherence.py; the observation behind cascade_quad.hip).  Such a pair shares ONE 5 x (5 + dx) box: 25 or 30 cells =
// 100 or 120 candidate rows are gathered once and both quads' 8 queries run against them -- 40 % fewer gathered bytes per quad, and
// two INDEPENDENT accumulator chains per stage, which fill each other's MFMA dependency bubbles.  Each quad masks the box cells
// outside its own window; its candidate order (cell-major, child-minor: the order of upsampled_idx, quadtree_attention.py:419-429)
// is the box order restricted to its cells; before the softmax each quad's logits are pulled into that order (cross-lane reads), so
// the wave reductions and the conf_matrix [B,N,100] stores are those of the single-quad kernel.  Pairs that cannot share (different rows, further apart, irregular or clamped
// position lists, last quad of an odd row) run as two single-quad sub-items with their own 25 cells through the same code.
//
// Arithmetic is window_match_pos_kernel's, i.e. the oracle's: operands pre-scaled by 1/sqrt(C) (reciprocal multiply or division;
// nothing but the queries when sqrt(C) is a power of two), fp32 fmaf chain over c ascending on v_mfma_f32_4x4x1 (lane <->
// candidate), / T, masked entries -1e9, argmax = first maximum of the logits, window_softmax2 for the probabilities.  Outputs are
// bit-equal to window_match_pos_kernel's (tests/test_gpu_ops.py::test_window_match_pair_kernel).
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct WPairArgs {
    const float* fq;        // [B, h0*w0, C]
    const float* fk;        // [B, h1*w1, C]
    const int64_t* tp;      // [B, (h0/2)*(w0/2), 25, 2] (row, col) on the (h1/2) x (w1/2) grid
    const uint8_t* mq;      // nullable [B, h0*w0]
    const uint8_t* mk;      // nullable [B, h1*w1]
    float* conf;            // nullable [B, h0*w0, 100]
    float* next_conf;       // [B, h0*w0]
    int64_t* next_idx;      // [B, h0*w0]
    float sqrtC, inv_sqrtC, T, invT;
    int B, h0, w0, h1, w1, nquads, npr, nitems;   // npr = pair items per quad row, nitems = pair items per image pair
    int* ctr;               // nullable: work_counters() -- items beyond a wave's first two are claimed, not dealt out (see cascade_quad.hip)
    int claim;              // consecutive items per claim
};

struct WSub {   // one sub-item (all wave-uniform): `ncells` box cells against nq query quads; slot 1's quad is the right-hand neighbour
    int b, l00_0, nq, ncells, qslot;   // qslot: query staging slot of this sub-item's slot 0 (1: the right quad runs alone)
    unsigned mask0, mask1;             // bit e: box cell e belongs to slot 0's / slot 1's window
};

template <int C, bool RECIP>
__global__ __launch_bounds__(128, 2) void window_match_pair_kernel(const WPairArgs a) {
    constexpr int KW = 25, K = 100, NCH = C / 32, NS = 2 * NCH;
    constexpr bool P2 = C == 16 || C == 64 || C == 256;   // sqrt(C) a power of two: see window_dma.hip
    constexpr int QS = C + 4;                             // query row stride (the 4 children's broadcast reads in different banks)
    constexpr int NQI = C / 32;                           // loads of 256 floats that bring both quads' 8 query rows
    constexpr int WAVE_FLOATS = 8 * QS + 32 + 7 * 256 + 8 * 256;   // queries [2][4][QS] | cell bases [32] | odd stages (<= 56 rows) | even stages (64 rows)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* qn = smem + wave * WAVE_FLOATS;
    int* t2 = reinterpret_cast<int*>(qn + 8 * QS);
    float* buf1 = reinterpret_cast<float*>(t2 + 32);      // pass 1 (rows 64 ..): lanes beyond its last row read into buf0 -- finite, unused
    float* buf0 = buf1 + 7 * 256;
    const int N = a.h0 * a.w0, S = a.h1 * a.w1, wq = a.w0 >> 1, h1p = a.h1 >> 1, w1p = a.w1 >> 1, w1 = a.w1;
    const int xcd = blockIdx.x & 7, chunk = (a.nitems + 7) >> 3;
    const int cnt = min(chunk, a.nitems - xcd * chunk);
    const int total = cnt > 0 ? a.B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    int t = (blockIdx.x >> 3) * 2 + wave;
    int* const ctr = a.ctr ? a.ctr + xcd * WORK_XCD_INTS : nullptr;
    if (t >= total) {
        if (ctr && lane == 0) work_leave(ctr, stride);
        return;
    }
    bool claim_pending = false;   // the item after the last prefetch has been claimed and not collected yet (then t == total)
    int rem = 0;                  // items left in the claimed run behind t
    const unsigned buf0_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf0));
    const unsigned buf1_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf1));
    const int sl8 = lane >> 3, un = lane & 7;
    unsigned swzb[4];   // DMA instruction j: rows 8 j + lane / 8, unit un <- logical unit un ^ ((row >> 1) & 7); pre-biased for glds_chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) swzb[j] = (unsigned)((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072u - 1024u * j;
    unsigned rd[8];     // read side: byte offset of logical unit u in this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));

    // ---- item-level prefetch registers: window positions (lane e < 25: quad A, lane 32 + e: quad B), both quads' queries, query masks
    int pf_r = 0, pf_c = 0, pf_b = 0, pf_l00A = 0, pf_mq = 1;
    bool pf_hasB = false, regs_full = false, pendB = false;
    f32x4 pf_q[NQI];
    auto prefetch = [&](int tt) {
        const int b = tt / cnt, it = xcd * chunk + tt % cnt;
        const int qy = it / a.npr, m = it % a.npr;
        const int quadA = qy * wq + 2 * m;
        pf_b = b; pf_l00A = 2 * qy * a.w0 + 4 * m; pf_hasB = 2 * m + 1 < wq;
        const int e = lane & 31;
        if ((lane < 32 || pf_hasB) && e < KW) {
            const longlong2 rc = *reinterpret_cast<const longlong2*>(a.tp + (((size_t)b * a.nquads + quadA + (lane >> 5)) * KW + e) * 2);
            pf_r = (int)rc.x; pf_c = (int)rc.y;
        }
#pragma unroll
        for (int i = 0; i < NQI; ++i) {   // float i*256 + 4*lane of the 8 rows [slot][child][C]; without a right neighbour slot 1 re-reads slot 0
            const int el = i * 256 + lane * 4, r = el / C, col = el % C, sl = pf_hasB ? (r >> 2) : 0, f = r & 3;
            pf_q[i] = *reinterpret_cast<const f32x4*>(a.fq + ((size_t)b * N + pf_l00A + 2 * sl + (f >> 1) * a.w0 + (f & 1)) * C + col);
        }
        if (a.mq && lane < 8) {
            const int sl = pf_hasB ? (lane >> 2) : 0, f = lane & 3;
            pf_mq = a.mq[(size_t)b * N + pf_l00A + 2 * sl + (f >> 1) * a.w0 + (f & 1)];
        }
        regs_full = true;
    };
    f32x4 q_nx[NQI];
    int mq_nx = 1;
    auto put_queries = [&]() {
#pragma unroll
        for (int i = 0; i < NQI; ++i) {
            const int el = i * 256 + lane * 4, r = el / C, col = el % C;
            *reinterpret_cast<f32x4*>(qn + r * QS + col) = q_nx[i];
        }
        wave_lds_fence();
    };
    unsigned rowb[2][8];
    int cnd_nx[2] = {0, 0}, mk_nx[2] = {1, 1};
    WSub sub_nx{};
    // registers -> the next sub-item: sharing decision, cell bases to LDS, this lane's candidates, DMA row offsets, masks
    auto stage_in = [&]() {
        const int e = lane & 31;
        const int cb = (pf_r * 2) * w1 + pf_c * 2;   // first child of this lane's window cell on the fine grid (valid for e < 25)
        if (pendB) {            // second half of a pair that could not share: quad B alone, its own 25 cells in list order
            if (lane >= 32 && e < KW) t2[e] = cb;
            sub_nx = WSub{pf_b, pf_l00A + 2, 1, KW, 1, (1u << KW) - 1u, 0u};
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
            if (pf_hasB && regA && regB && oyA == oyB && dx <= 1) {   // one 5 x (5 + dx) box for both quads, row-major
                const int bx0 = min(oxA, oxB), bw = 5 + dx, nc = 5 * bw;
                if (lane < 32 && e < nc) t2[e] = (2 * (oyA + e / bw)) * w1 + 2 * (bx0 + e % bw);
                unsigned mA = 0, mB = 0;
#pragma unroll
                for (int r = 0; r < 5; ++r) { mA |= 0x1Fu << (bw * r + (oxA - bx0)); mB |= 0x1Fu << (bw * r + (oxB - bx0)); }
                sub_nx = WSub{pf_b, pf_l00A, 2, nc, 0, mA, mB};
                regs_full = false;
            } else {                                                   // quad A alone now; quad B (if any) as the next sub-item
                if (lane < 32 && e < KW) t2[e] = cb;
                sub_nx = WSub{pf_b, pf_l00A, 1, KW, 0, (1u << KW) - 1u, 0u};
                pendB = pf_hasB; regs_full = pf_hasB;
            }
            // both quads' queries, pre-scaled, wait in registers until the current sub-item's last stage has read qn (put_queries)
#pragma unroll
            for (int i = 0; i < NQI; ++i) {
                f32x4 v = pf_q[i];
                if constexpr (P2) {
                    const float inv_C = a.inv_sqrtC * a.inv_sqrtC;   // exact: a power of two
                    v.x *= inv_C; v.y *= inv_C; v.z *= inv_C; v.w *= inv_C;
                } else {
                    v.x = div_scalar<RECIP>(v.x, a.sqrtC, a.inv_sqrtC); v.y = div_scalar<RECIP>(v.y, a.sqrtC, a.inv_sqrtC);
                    v.z = div_scalar<RECIP>(v.z, a.sqrtC, a.inv_sqrtC); v.w = div_scalar<RECIP>(v.w, a.sqrtC, a.inv_sqrtC);
                }
                q_nx[i] = v;
            }
            mq_nx = pf_mq;
        }
        wave_lds_fence();
        // candidate k = 64 p + lane: cell k / 4 of the box (padded with the last cell's last child), child k % 4 -> row + child / 2,
        // col + child % 2 on the fine grid, clamped like torch.clamp on the flattened index (:429)
        const int ncl = sub_nx.ncells;
        const int coff = ((lane & 3) >> 1) * w1 + (lane & 1), coff3 = w1 + 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = 64 * p + lane;
            const int id = t2[min(k >> 2, ncl - 1)] + (k < 4 * ncl ? coff : coff3);
            cnd_nx[p] = id < 0 ? 0 : (id > S - 1 ? S - 1 : id);
        }
        // row 8 j + lane / 8 of DMA instruction j is the candidate of lane 8 j + lane / 8: one cross-lane read per instruction
        const int rb[2] = {cnd_nx[0] * (C * 4), cnd_nx[1] * (C * 4)};
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < (p == 0 ? 8 : 7); ++j)
                rowb[p][j] = (unsigned)__builtin_amdgcn_ds_bpermute((8 * j + sl8) * 4, rb[p]) + swzb[j & 3];
        if (a.mk) {
            mk_nx[0] = a.mk[(size_t)sub_nx.b * S + cnd_nx[0]];
            mk_nx[1] = a.mk[(size_t)sub_nx.b * S + cnd_nx[1]];
        }
    };
    // stage (ch, p): 32 channels of the rows of pass p -> its buffer; pass 1 has 5 (25 cells) or 7 (30 cells) instructions
    auto issue = [&](auto sc, const WSub& s) {
        constexpr int st = decltype(sc)::value;
        constexpr int ch = st >> 1, p = st & 1;
        const int sb = __builtin_amdgcn_readfirstlane(s.b);
        const float* base = a.fk + (size_t)sb * S * C + ch * 32 - 768;   // 3072 bytes low (swzb)
        if constexpr (p == 0) {
            glds_chunk(base, rowb[0][0], rowb[0][1], rowb[0][2], rowb[0][3], buf0_lds);
            glds_chunk(base, rowb[0][4], rowb[0][5], rowb[0][6], rowb[0][7], buf0_lds + 4096);
        } else {
            glds_chunk(base, rowb[1][0], rowb[1][1], rowb[1][2], rowb[1][3], buf1_lds);
            if (s.ncells > KW) glds_chunk3(base, rowb[1][4], rowb[1][5], rowb[1][6], buf1_lds + 4096);
            else glds_chunk1(base, rowb[1][4], buf1_lds + 4096);
        }
    };
    auto wait_for = [&](int younger) {   // at most `younger` (5, 7 or 8) vector-memory operations may still be in flight
        if (younger >= 8) glds_wait<8>();
        else if (younger == 7) glds_wait<7>();
        else glds_wait<5>();
    };
    // results of the previous sub-item (own candidate order: lane l holds candidates l and 64 + l of each quad's list), stored one stage
    // late (right behind a DMA wait: vmcnt counts stores too, in order)
    float pe[2][4][2], pnc[2][4];
    int pam[2][4], pcnd[2][2] = {{0, 0}, {0, 0}};
    WSub pend_sub{};
    bool have_pend = false;
    auto flush = [&]() {
        if (have_pend) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl < pend_sub.nq) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const size_t n = (size_t)pend_sub.b * N + pend_sub.l00_0 + 2 * sl + (f >> 1) * a.w0 + (f & 1);
                        if (a.conf) {
                            a.conf[n * K + lane] = pe[sl][f][0];
                            if (64 + lane < K) a.conf[n * K + 64 + lane] = pe[sl][f][1];
                        }
                        if (lane == (pam[sl][f] & 63)) {
                            a.next_conf[n] = pnc[sl][f];
                            a.next_idx[n] = pam[sl][f] < 64 ? pcnd[sl][0] : pcnd[sl][1];
                        }
                    }
                }
            }
        }
        have_pend = false;
    };
    // own candidate k = 64 p + lane of a quad: window cell k / 4 = (cell / 5, cell % 5)
    int own_r[2], own_c[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) { const int cell = min((64 * p + lane) >> 2, KW - 1); own_r[p] = cell / 5; own_c[p] = cell % 5; }

    WSub sub_cur{};
    prefetch(t);
    t += stride;
    stage_in();
    sub_cur = sub_nx;
    put_queries();
    int cnd[2] = {cnd_nx[0], cnd_nx[1]}, mkv[2] = {mk_nx[0], mk_nx[1]}, mqv = mq_nx;
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
    issue(std::integral_constant<int, 0>{}, sub_cur);
    for (;;) {
        const WSub s = sub_cur;
        const bool more = regs_full || pendB;          // another sub-item follows (its item's identity is in the prefetch registers)
        const int n1 = s.ncells > KW ? 7 : 5;
        int claim_ret = 0;   // this sub-item's claim: issued under its last stage, collected at its end
        const float* qp = qn + s.qslot * 4 * QS + (lane & 3) * QS;
        f32x4 acc[2][2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[sl][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<0, NS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            constexpr int ch = st >> 1, p = st & 1;
            if constexpr (st == NS - 1) {
                if (more) stage_in();   // the next sub-item's front end, then its stage 0, under this sub-item's last stage
            }
            lds_reads_done();
            if constexpr (st + 1 < NS) {
                issue(std::integral_constant<int, (st + 1 < NS ? st + 1 : 0)>{}, s);
                wait_for(p == 0 ? n1 : 8);   // the stage just issued is pass 1 (n1 instructions) behind a pass 0, and vice versa
            } else {
                if (more) { issue(std::integral_constant<int, 0>{}, sub_nx); wait_for(8); }
                else glds_wait<0>();
            }
            if constexpr (st == 0) flush();
            if constexpr (st == NS - 1) {
                const bool pf = more && !regs_full && t < total;
                if (pf) prefetch(t);   // behind the wait: a whole sub-item ahead of stage_in
                work_claim_issue(ctr, pf && ctr != nullptr && rem == 0, claim_ret);   // dynamic schedule: the next run of items (unconditional statement)
                if (pf) {
                    if (!ctr) t += stride;
                    else if (rem > 0) { ++t; --rem; }
                    else { claim_pending = true; t = total; }
                }
            }
            const char* bp = reinterpret_cast<const char*>(p ? buf1 : buf0);
            f32x4 kr[8];          // operand B: this lane's candidate row chunk
#pragma unroll
            for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(bp + rd[u]);
            f32x4 qa[2][8];       // operand A: lane l holds q[slot][child l % 4][c]
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (sl < s.nq) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) qa[sl][u] = *reinterpret_cast<const f32x4*>(qp + sl * 4 * QS + ch * 32 + 4 * u);
                }
            lds_reads_done();
            if constexpr (!P2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    kr[u].x = div_scalar<RECIP>(kr[u].x, a.sqrtC, a.inv_sqrtC); kr[u].y = div_scalar<RECIP>(kr[u].y, a.sqrtC, a.inv_sqrtC);
                    kr[u].z = div_scalar<RECIP>(kr[u].z, a.sqrtC, a.inv_sqrtC); kr[u].w = div_scalar<RECIP>(kr[u].w, a.sqrtC, a.inv_sqrtC);
                }
            }
            if (s.nq == 2) {   // two independent chains, interleaved
                f32x4 a0 = acc[0][p], a1 = acc[1][p];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].x, kr[u].x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[1][u].x, kr[u].x, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].y, kr[u].y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[1][u].y, kr[u].y, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].z, kr[u].z, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[1][u].z, kr[u].z, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].w, kr[u].w, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[1][u].w, kr[u].w, a1, 0, 0, 0);
                }
                acc[0][p] = a0; acc[1][p] = a1;
            } else {
                f32x4 a0 = acc[0][p];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].x, kr[u].x, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].y, kr[u].y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].z, kr[u].z, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[0][u].w, kr[u].w, a0, 0, 0, 0);
                }
                acc[0][p] = a0;
            }
            asm volatile("" : "+v"(acc[0][p]), "+v"(acc[1][p]));   // keep the stage's arithmetic inside the stage
        });
        // softmax over each quad's own 100 candidates, first argmax of the logits (cascade_matching.py:119-149); stores deferred to flush().
        // A shared box holds the candidates in BOX order; each quad's logits (and candidate ids, key masks) are first pulled into its own
        // list order -- lane l <- own candidates l and 64 + l, exactly window_match_pos_kernel's register layout -- so that the wave
        // reductions add the same numbers in the same order and the probabilities come out bit-identical to the single-quad kernel's.
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl < s.nq) {
                float xo[2][4];
                int cndo[2], mko[2];
                if (s.nq == 2) {
                    const unsigned mk_ = sl ? s.mask1 : s.mask0;
                    const int off = __builtin_ctz(mk_), bw = s.ncells / 5;   // the window's first column inside the box, box width
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int kb = ((own_r[p] * bw + own_c[p] + off) << 2) | (lane & 3);   // box index of own candidate 64 p + lane
                        const int addr = (kb & 63) << 2;
                        const bool hi = kb >= 64;
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const int v0 = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(acc[sl][0][f]));
                            const int v1 = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(acc[sl][1][f]));
                            xo[p][f] = __int_as_float(hi ? v1 : v0);
                        }
                        const int c0 = __builtin_amdgcn_ds_bpermute(addr, cnd[0]), c1 = __builtin_amdgcn_ds_bpermute(addr, cnd[1]);
                        cndo[p] = hi ? c1 : c0;
                        const int m0 = __builtin_amdgcn_ds_bpermute(addr, mkv[0]), m1 = __builtin_amdgcn_ds_bpermute(addr, mkv[1]);
                        mko[p] = hi ? m1 : m0;
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) xo[p][f] = acc[0][p][f];
                        cndo[p] = cnd[p]; mko[p] = mkv[p];
                    }
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int mqf = __builtin_amdgcn_readlane(mqv, (s.qslot + sl) * 4 + f);
                    float x[2] = {0.f, 0.f};
                    unsigned key[2] = {0u, 0u};
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (64 * p + lane < K) {
                            float v = div_scalar<RECIP>(xo[p][f], a.T, a.invT);
                            if (a.mq && !(mqf && mko[p])) v = NEG_FILL;
                            x[p] = v; key[p] = f2ord(v);
                        }
                    }
                    const unsigned wm = wave_max_u32(max(key[0], key[1]));
                    const float m = ord2f(wm);
                    float e0, e1;
                    window_softmax2(x[0], x[1], m, lane < K, 64 + lane < K, e0, e1);
                    const unsigned long long b0 = __ballot(key[0] == wm && lane < K);
                    const unsigned long long b1 = __ballot(key[1] == wm && 64 + lane < K);
                    const int am = b0 ? (__ffsll((long long)b0) - 1) : (64 + __ffsll((long long)b1) - 1);
                    pe[sl][f][0] = e0; pe[sl][f][1] = e1; pam[sl][f] = am; pnc[sl][f] = am < 64 ? e0 : e1;
                }
                pcnd[sl][0] = cndo[0]; pcnd[sl][1] = cndo[1];
            }
        }
        pend_sub = s; have_pend = true;
        lds_reads_done();
        if (claim_pending) {
            // the claim must be back before the loop's back edge (the compiler may copy loop-carried registers there).  Everything older --
            // the next sub-item's stage 0, this prefetch's loads -- was issued before the softmax above and is needed at the top anyway.
            glds_wait<0>();
            t = work_claimed(claim_ret) * a.claim + 2 * stride;
            rem = a.claim - 1;
            claim_pending = false;
        }
        if (!more) break;
        put_queries();
        cnd[0] = cnd_nx[0]; cnd[1] = cnd_nx[1]; mkv[0] = mk_nx[0]; mkv[1] = mk_nx[1]; mqv = mq_nx;
        sub_cur = sub_nx;
    }
    glds_wait<0>();
    flush();
    if (ctr && lane == 0) work_leave(ctr, stride);
}

template <int C, bool RECIP>
static int launch_wm_pair(const WPairArgs& a, hipStream_t s) {
    constexpr size_t lds = sizeof(float) * 2 * (8 * (C + 4) + 32 + 7 * 256 + 8 * 256);
    static int resident_tab[CASMTR_MAX_DEVICES] = {0};   // persistent grid: exactly the workgroups that are resident at once
    int resident = 0;
    if (const int r = resident_workgroups(resident_tab, window_match_pair_kernel<C, RECIP>, 128, lds, &resident)) return r;
    const long long work = (long long)a.B * a.nitems;
    long long blocks = resident;
    if (blocks > (work + 1) / 2) blocks = ((work + 1) / 2 + 7) / 8 * 8;
    prof_symbol_args(CASMTR_PROF_WINDOW_MATCH, "<%d,%s>", C, RECIP ? "true" : "false");
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_WINDOW_MATCH, (window_match_pair_kernel<C, RECIP>), dim3((unsigned)blocks), dim3(128), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// -> CASMTR_ERR_UNSUPPORTED for shapes the pair kernel does not cover (the caller runs window_match_pos_kernel)
int casmtr_window_match_pair(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T, int recip,
                             float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW, int C, int dil,
                             hipStream_t s) {
    // the division mode (recip == 0, the reference's CPU arithmetic) would need more registers than two waves per SIMD leave without
    // spilling (a scratch access between a DMA issue and its hand-counted wait is not allowed): it stays on the single-quad kernel
    if (!recip || KW != 25 || dil != 1 || (C != 128 && C != 64) || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1) || h1 < 10 || w1 < 10 ||
        (long long)h1 * w1 * C * 4 >= (1ll << 32))   // row byte offsets are 32-bit
        return CASMTR_ERR_UNSUPPORTED;
    WPairArgs a{};
    a.fq = fq; a.fk = fk; a.tp = tp; a.mq = mq; a.mk = mk; a.conf = conf; a.next_conf = next_conf; a.next_idx = next_idx;
    a.sqrtC = (float)sqrt((double)C); a.inv_sqrtC = 1.0f / a.sqrtC; a.T = T; a.invT = 1.0f / T;
    a.B = B; a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.nquads = (h0 / 2) * (w0 / 2);
    a.npr = (w0 / 2 + 1) / 2; a.nitems = (h0 / 2) * a.npr;
    { const char* ev = getenv("CASMTR_WP_DYNAMIC"); a.ctr = (ev && ev[0] == '0') ? nullptr : work_counters(s); }
    { const char* ev = getenv("CASMTR_WP_CLAIM"); a.claim = ev && atoi(ev) > 0 ? atoi(ev) : 1; }
    return C == 128 ? launch_wm_pair<128, true>(a, s) : launch_wm_pair<64, true>(a, s);
}
