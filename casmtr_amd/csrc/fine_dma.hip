// QTAttB.process_fine_level + its share of the message merge (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/
// quadtree_attention.py:180-229,262-284) for gfx950: persistent, wave-per-(quad, head), LDS-DMA + matrix-core streaming kernel.
//
// Work item = one head of one quad of query tokens (the 4 children share the head's candidate list: 4 children of each of the
// Kp parents the previous level selected for this head).  A WAVE takes an item from the candidate list to the stored message /
// top-k, then moves on; workgroups are 2 independent waves, no block barrier.  Stages per item: K(pass 0..), V(pass 0..), 64
// candidate rows x 128 B (D = 32 floats of the head = one cache line) per stage, brought in by LDS-DMA (8 wave-instructions, 8
// lanes per line, no VGPRs in flight) into one of two wave-private 8 KB buffers, always one stage ahead of the arithmetic -- the
// pipeline runs across items: stage 0 of the next item is issued under the last stage of the current one.
//   K stage : lane <-> candidate; the row comes back with 8 conflict-free ds_read_b128 (source-side XOR swizzle) and goes through
//             32 v_mfma_f32_4x4x1_16B_f32 (operand A: the 4 children's q[d]; operand B: the lane-per-candidate rows): one
//             instruction per d, a d-sequence of them is the exact d-ascending fmaf chain (tools/probes/mfma4x4_layout.hip), so
//             the logits -- and therefore the top-k indices -- are bit-identical to the oracle's sequential chain;
//   select  : logits -> LDS (the stage's own, already consumed buffer) -> one series per 16-lane DPP row: softmax and top-k by
//             iterated row argmax (first position by ballot / ffs), exactly the selection of quad_attn_kernel<H,KMAX,0>;
//   V stage : one v_mfma_f32_4x4x1 per PAIR of value rows (blocks 0-7: row 2m, blocks 8-15: row 2m+1; operand A = the 4
//             children's probabilities of the row, operand B = the two staged rows as they lie in LDS), lane l accumulates
//             message[child r][d = l % 32] in register r;  final = final[parent] + message * weight fused into the store (:277-281).
// Work order: XCD x (= blockIdx % 8, observed placement; speed only) takes head x % H, so the rows an XCD gathers from are one
// head's slice of a pair's keys and values (2.8 MB at 104x104) and stay resident in its 4 MB L2; the round-1 kernel gathered all
// heads of a quad from every XCD (22 MB working set: 47 % L2 hit rate, 6x the algorithmic bytes over the fabric).
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct FineArgs {
    const float* q;        // [B,L,H*32]
    const float* key;      // [B,S,H*32]
    const float* value;    // [B,S,H*32]
    const int64_t* pidx;   // [B,Lq,Kp,H] previous level's top-k (absolute index on the (h1/2) x (w1/2) grid)
    const float* acc_in;   // nullable [B,Lq,H*32]
    float* message;        // nullable [B,L,H*32]
    float* acc_out;        // nullable [B,L,H*32]
    float* topk_score;     // [B,L,topk,H]
    int64_t* topk_idx;     // [B,L,topk,H]
    float temp, w_level;
    int topk, B, h0, w0, h1, w1, H, Kp, nquads, dbg;
};

template <int NPASS, bool EXACT>   // EXACT: the logits feed a top-k (bit-exact sequential d-chain); otherwise only a softmax
__global__ __launch_bounds__(128, 2) void fine_level_dma_kernel(const FineArgs a) {
    constexpr int KMAX = 64 * NPASS, NS = 2 * NPASS;
    constexpr int E = KMAX / 16;          // elements per lane in the 16-lane-row softmax / top-k
    constexpr int KS = KMAX + 4;          // logits row stride (floats)
    constexpr int WAVE_FLOATS = KMAX * 4 + 2 * 128 + 2 * 32 + 2 * 2048;   // probabilities [KMAX][4] | q [2][4][32] | parents [2][32] | 2 x [64][32]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Ald = smem + wave * WAVE_FLOATS;
    float* qs = Ald + KMAX * 4;                               // [parity][4 children][32]
    int* ptab = reinterpret_cast<int*>(qs + 2 * 128);         // [parity][Kp] first-child index r*w1 + c of every parent
    float* buf = reinterpret_cast<float*>(ptab + 2 * 32);
    const int H = a.H, HD = H * 32, Kp = a.Kp, K = 4 * Kp;
    const int L = a.h0 * a.w0, S = a.h1 * a.w1, wq = a.w0 >> 1, Lq = a.nquads, w1p = a.w1 >> 1;
    // ---- work list: XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = cnt > 0 ? a.B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    const int t = (blockIdx.x >> 3) * 2 + wave;
    if (g >= G || t >= total) return;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf));
    const int sl = lane >> 3, un = lane & 7;                  // DMA: row within the instruction's 8, 16-byte unit of the 128-byte row
    const unsigned swz[2] = {(unsigned)((un ^ (lane >> 4)) * 16), (unsigned)((un ^ (4 + (lane >> 4))) * 16)};   // K stages, DMA instr j even / odd
    unsigned rd[8];                                           // K stages: byte offset of logical unit u in this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    const bool no_dma = a.dbg & CASMTR_DBG_NO_DMA, no_math = a.dbg & CASMTR_DBG_NO_MATH;
    const float* kh = a.key + h * 32;
    const float* vh = a.value + h * 32;

    // ---- per-item front end, run one item ahead: global -> registers (prefetch), registers -> LDS + DMA row offsets (stage_in)
    long long pf_p = 0;
    f32x4 pf_q = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pf_acc = 0.f, acc_cur = 0.f, acc_nx = 0.f;      // final[parent] of the item (lane d < 32), :277
    // the wave's items in order, without a division per item: a cursor (pair, quad within the chunk, quad row / column) that
    // advances by the number of waves of the XCD
    struct Item { int b, quad, l00; };   // pair, quad, first child's token index (child f -> l00 + (f>>1)*w0 + (f&1))
    int cb = t / cnt, cq = t % cnt, cy = (g * chunk + cq) / wq, cx = (g * chunk + cq) % wq;
    const int sy = stride / wq, sx = stride % wq;
    auto take = [&](Item& it) {   // -> false when the wave's list is exhausted
        if (cb >= a.B) return false;
        it.b = cb; it.quad = g * chunk + cq; it.l00 = 2 * cy * a.w0 + 2 * cx;
        cq += stride;
        if (cq >= cnt) {
            while (cq >= cnt) { cq -= cnt; ++cb; }
            cy = (g * chunk + cq) / wq; cx = (g * chunk + cq) % wq;
        } else {
            cy += sy; cx += sx;
            if (cx >= wq) { cx -= wq; ++cy; }
        }
        return true;
    };
    auto prefetch = [&](const Item& it) {
        if (lane < Kp) pf_p = a.pidx[(((size_t)it.b * Lq + it.quad) * Kp + lane) * H + h];
        if (lane < 32) {
            const int f = lane >> 3, lf = it.l00 + (f >> 1) * a.w0 + (f & 1);
            pf_q = *reinterpret_cast<const f32x4*>(a.q + ((size_t)it.b * L + lf) * HD + h * 32 + un * 4);
            if (a.acc_in) pf_acc = a.acc_in[((size_t)it.b * Lq + it.quad) * HD + h * 32 + lane];
        }
    };
    unsigned rowb[NPASS][8];   // DMA instruction j of pass p moves candidate rows 64p + 8j .. + 7: lane -> row 64p + 8j + lane/8
    auto stage_in = [&](int par) {   // consumes the prefetch registers of the item whose parity is `par`
        if (lane < Kp) {
            const int p = (int)pf_p;
            ptab[par * 32 + lane] = (p / w1p) * 2 * a.w1 + (p % w1p) * 2;   // :193-199, children (+0,+0),(+0,+1),(+1,+0),(+1,+1)
        }
        if (lane < 32) *reinterpret_cast<f32x4*>(qs + par * 128 + lane * 4) = pf_q;
        acc_nx = pf_acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int k = 64 * p + 8 * j + sl;
                k = k < K ? k : K - 1;
                rowb[p][j] = (unsigned)(ptab[par * 32 + (k >> 2)] + ((k >> 1) & 1) * a.w1 + (k & 1)) * (unsigned)(HD * 4);
            }
    };
    auto issue = [&](auto sc, int b) {
        constexpr int s = decltype(sc)::value;
        constexpr int isv = s / NPASS, p = s % NPASS;
        const float* base = (isv ? vh : kh) + (size_t)b * S * HD;   // wave-uniform: scalar arithmetic
#pragma unroll
        for (int j = 0; j < 8; ++j)
            glds16(base, rowb[p][j] + (isv ? (unsigned)(un * 16) : swz[j & 1]), buf_lds + (unsigned)((s & 1) * 8192 + j * 1024));
    };

    Item it_cur{}, it_nx{}, it_pf{};
    take(it_cur);
    prefetch(it_cur);
    stage_in(0);
    acc_cur = acc_nx;
    int par = 0;
    // results of the previous item: stored one stage late, right behind a DMA wait, so that the stores have a whole stage to
    // retire before the next wait (vmcnt counts them too, in order: a store issued just in front of a wait stalls the wave)
    f32x4 pend = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pend_acc = 0.f;
    int pend_b = 0, pend_l00 = 0;
    bool have_pend = false;
    auto flush = [&]() {
        if (have_pend && lane < 32) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const size_t o = ((size_t)pend_b * L + pend_l00 + (f >> 1) * a.w0 + (f & 1)) * HD + h * 32 + lane;
                if (a.message) a.message[o] = pend[f];
                if (a.acc_out) a.acc_out[o] = pend_acc + pend[f] * a.w_level;   // separate multiply and add (:277-281)
            }
        }
        have_pend = false;
    };
    if (!no_dma) issue(std::integral_constant<int, 0>{}, it_cur.b);
    bool more = take(it_nx), has_pf = false;
    if (more) prefetch(it_nx);
    for (;; par ^= 1) {
        const int b = it_cur.b, l00 = it_cur.l00, bn = it_nx.b;
        const float* qsp = qs + par * 128;
        const int* ptp = ptab + par * 32;
        f32x4 lg[NPASS];
        f32x4 acc[4];
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int isv = s / NPASS, p = s % NPASS;
            if constexpr (s == NS - 1) {
                // the next item's front end, then its stage 0 -- under this item's last stage
                if (more) stage_in(par ^ 1);
            }
            if (!no_dma) {
                if constexpr (s + 1 < NS) {
                    lds_reads_done();
                    issue(std::integral_constant<int, s + 1>{}, b);
                    glds_wait<8>();
                } else {
                    lds_reads_done();
                    if (more) { issue(std::integral_constant<int, 0>{}, bn); glds_wait<8>(); }
                    else glds_wait<0>();
                }
            }
            if constexpr (s == 0) flush();
            if constexpr (s == NS - 1) {
                // the item after next: its loads are issued behind the wait above (a load issued in front of a DMA batch has to
                // land before the wait that follows the batch) and have a whole item's time before stage_in consumes them
                has_pf = more && take(it_pf);
                if (has_pf) prefetch(it_pf);
            }
            const char* bp = reinterpret_cast<const char*>(buf) + (s & 1) * 8192;
            if (no_math) return;
            if constexpr (!isv) {
                f32x4 qa[8], kr[8];   // operand A: lane l holds q[child l%4][d]; operand B: this lane's candidate row
#pragma unroll
                for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qsp + (lane & 3) * 32 + 4 * u);
#pragma unroll
                for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(bp + rd[u]);
                lds_reads_done();   // one wait for the 16 reads (hipcc otherwise threads them through the dependent MFMA chain, a wait each)
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
                    // no index depends on these logits (finest level: no top-k): four interleaved partial d-chains instead of one
                    // sequential chain -- a dependent v_mfma_f32_4x4x1 waits ~28 cycles for its accumulator
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
                if constexpr (p == NPASS - 1) {
                    // ---- logits -> LDS (this stage's buffer: its rows are consumed), one series (child f) per 16-lane row
                    lds_reads_done();
                    float* Sld = const_cast<float*>(reinterpret_cast<const float*>(bp));   // [4][KS]
#pragma unroll
                    for (int pp = 0; pp < NPASS; ++pp)
#pragma unroll
                        for (int f = 0; f < 4; ++f) Sld[f * KS + 64 * pp + lane] = lg[pp][f];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int f = lane >> 4, j = lane & 15;
                    float lv[E];
                    unsigned key[E];
                    const f32x4* sp = reinterpret_cast<const f32x4*>(Sld + f * KS + j * E);
#pragma unroll
                    for (int e4 = 0; e4 < E / 4; ++e4) {
                        const f32x4 v = sp[e4];
                        lv[4 * e4 + 0] = v.x; lv[4 * e4 + 1] = v.y; lv[4 * e4 + 2] = v.z; lv[4 * e4 + 3] = v.w;
                    }
                    unsigned lm = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        key[e] = (j * E + e < K) ? f2ord(lv[e]) : 0u;
                        lm = max(lm, key[e]);
                    }
                    const float m = ord2f(row16_max_u32(lm));
                    float ps[E];
                    float sum = 0.f;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        ps[e] = (j * E + e < K) ? __expf(lv[e] - m) : 0.f;
                        sum += ps[e];
                    }
                    sum = 1.0f / row16_sum_f32(sum);
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        ps[e] = ps[e] * sum;
                        Ald[(j * E + e) * 4 + f] = ps[e];
                    }
                    const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
                    for (int tk = 0; tk < a.topk; ++tk) {   // selection on the logits, (logit desc, position asc)
                        unsigned cur = 0;
#pragma unroll
                        for (int e = 0; e < E; ++e) cur = max(cur, key[e]);
                        const unsigned rm = row16_max_u32(cur);
                        const unsigned long long bal = __ballot(cur == rm);
                        const unsigned bits = (unsigned)(bal >> (f * 16)) & 0xFFFFu;
                        const int wj = __ffs(bits) - 1;  // first lane of the row holding the maximum -> smallest position
                        if (j == wj) {
                            bool done = false;
                            int kpos = 0;
                            float sc2 = 0.f;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const bool hit = !done && key[e] == rm;
                                if (hit) { kpos = j * E + e; sc2 = ps[e]; key[e] = 0u; done = true; }
                            }
                            const size_t o = (((size_t)b * L + lf) * a.topk + tk) * H + h;
                            a.topk_idx[o] = ptp[kpos >> 2] + ((kpos >> 1) & 1) * a.w1 + (kpos & 1);
                            a.topk_score[o] = sc2;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
#pragma unroll
                for (int pp = 0; pp <= p; ++pp) asm volatile("" : "+v"(lg[pp]));   // keep the stage's arithmetic inside the stage
            } else {
                // ---- message += A . V over this pass's rows, two rows per instruction
                if constexpr (p == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                const float* vrow = reinterpret_cast<const float*>(bp) + lane;              // + 64 m : rows 2m | 2m+1
                const float* prow = Ald + (64 * p + (lane >> 5)) * 4 + (lane & 3);          // + 8 m  : P[row][child lane%4]
#pragma unroll
                for (int m = 0; m < 32; ++m)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(prow[8 * m], vrow[64 * m], acc[m & 3], 0, 0, 0);
                if constexpr (p == NPASS - 1) {
                    f32x4 tot;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = (acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c]);
                        const unsigned xi = __float_as_uint(x);
                        const auto sw = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // lanes l and l ^ 32
                        tot[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                    }
                    pend = tot; pend_acc = acc_cur; pend_b = b; pend_l00 = l00; have_pend = true;
                } else {
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                }
            }
        });
        lds_reads_done();
        if (!more) break;
        acc_cur = acc_nx;
        it_cur = it_nx; it_nx = it_pf; more = has_pf;
    }
    flush();
}

template <int NPASS, bool EXACT>
static int launch_fine(const FineArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * 2 * (64 * NPASS * 4 + 2 * 128 + 2 * 32 + 2 * 2048);
    // persistent grid: exactly the workgroups that are resident at once
    static int resident_tab[CASMTR_MAX_DEVICES] = {0};
    int resident = 0;
    if (const int r = resident_workgroups(resident_tab, fine_level_dma_kernel<NPASS, EXACT>, 128, lds, &resident)) return r;
    const long long work = (long long)a.B * a.nquads * a.H;
    long long blocks = resident;
    if (blocks > (work + 1) / 2) blocks = ((work + 1) / 2 + 7) / 8 * 8;
    ProfScope ps(NPASS == 1 ? CASMTR_PROF_QTA_FINE : CASMTR_PROF_QTA_FINE2, s, "fine_level_dma_kernel");
    hipLaunchKernelGGL((fine_level_dma_kernel<NPASS, EXACT>), dim3((unsigned)blocks), dim3(128), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// -> CASMTR_ERR_UNSUPPORTED when the shape is outside this kernel (the caller then uses quad_attn_kernel<H,KMAX,0>)
int casmtr_qta_fine_level_dma(const float* q, const float* key, const float* value, const int64_t* prev_idx, float temp, int topk,
                              float w_level, const float* acc_in, float* message, float* acc_out, float* topk_score,
                              int64_t* topk_idx, int B, int h0, int w0, int h1, int w1, int H, int Kp, hipStream_t s) {
    const int K = 4 * Kp;
    if (K > 128 || Kp > 32 || (H != 8 && H != 4 && H != 2 && H != 1)) return CASMTR_ERR_UNSUPPORTED;
    FineArgs a{};
    a.q = q; a.key = key; a.value = value; a.pidx = prev_idx; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out;
    a.topk_score = topk_score; a.topk_idx = topk_idx; a.temp = temp; a.w_level = w_level; a.topk = topk; a.B = B;
    a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.H = H; a.Kp = Kp; a.nquads = (h0 / 2) * (w0 / 2); a.dbg = g_debug_flags;
    if (topk > 0) return K <= 64 ? launch_fine<1, true>(a, s) : launch_fine<2, true>(a, s);
    return K <= 64 ? launch_fine<1, false>(a, s) : launch_fine<2, false>(a, s);
}
