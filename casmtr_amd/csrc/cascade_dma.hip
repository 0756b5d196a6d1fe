// CascadeQTAttB.forward (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:400-452) for gfx950:
// persistent, wave-per-quad, LDS-DMA + matrix-core streaming kernel.
//
// A WAVE owns a quad of query tokens (the 4 children of a coarse cell share one window list, :450) from the candidate list to
// the stored message, then moves on to its next quad; a workgroup is 2 independent waves -- no block barrier anywhere.  Per
// head the wave walks 4 stages: K(h, pass 0), K(h, pass 1), V(h, pass 0), V(h, pass 1).  A stage's 64 candidate rows x 128 B
// (the head's D = 32 floats of a key / value row = one cache line) arrive by LDS-DMA -- 8 wave-instructions, 8 lanes per
// line, no VGPRs in flight -- in one of two wave-private 8 KB buffers, one stage ahead of the arithmetic:
//   K stage: lane <-> candidate.  The row comes back with 8 conflict-free ds_read_b128 (source-side XOR swizzle) and goes
//            through 32 v_mfma_f32_4x4x1_16B_f32: block b of the instruction multiplies the 4 children's q[d] (operand A, the
//            same 4 values in every block) with candidates 4b..4b+3 (operand B = exactly the lane-per-candidate layout), one
//            instruction per d; a d-sequence of them is the exact d-ascending fmaf chain (tools/probes/mfma4x4_layout.hip:
//            bit-identical to fmaf), and lane k ends up holding the 4 children's logits for candidate k.  The VALU version of
//            this stage needed 128 v_fmac + 32 broadcast LDS reads of q per stage and made the kernel LDS- and VALU-bound.
//   after pass 1: softmax over the K = 4*KW logits per child (wave reductions), probabilities to LDS as [k][4 children];
//   V stage: the same instruction with the roles turned: one v_mfma_f32_4x4x1 per PAIR of candidate rows -- blocks 0-7 take row
//            2m, blocks 8-15 row 2m+1; operand B = the two staged rows as they lie in LDS (lane l reads float l of the 256
//            bytes: one conflict-free ds_read_b32), operand A = the 4 children's probabilities of that row -- so lane l
//            accumulates message[child r][d = l % 32] in register r; the two halves are added with one row-swap and 32 lanes
//            store 128 B per child.  (On the VALU this stage was 832 fmas + a 16-value cross-lane fold per head.)
// The next quad's window positions and queries are fetched into registers while the current quad's stages run.
// Work order: every XCD walks one contiguous range of quads per image pair (neighbouring windows overlap: L2 locality).
// The round-1 kernel (quad_attn_kernel<H,KMAX,1>, qta_fused.hip: workgroup per quad, lane-per-row key reads straight into
// registers) spent 553 of its 886 us per launch in the logits phase at the L1/TA rate of one cache line per lane and clock.
// The int64 `upsampled_idx` (second return value of the reference) is written only on request.
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

template <int H, int NP1, bool HAS_REL>
__global__ __launch_bounds__(128, 2) void cascade_attn_dma_kernel(
    const float* __restrict__ q, const float* __restrict__ key, const float* __restrict__ value,
    const int64_t* __restrict__ topk_pos, const float* __restrict__ rel_pos, float* __restrict__ message,
    int64_t* __restrict__ up_idx, float temp, int dil, int B, int h0, int w0, int h1, int w1, int KW, int nquads, int dbg) {
    constexpr int HD = H * 32, NPASS = NP1 > 0 ? 2 : 1, NS = H * 2 * NPASS;
    constexpr int QV = (HD + 255) / 256;                 // float4s of a child's query row per lane
    constexpr int WAVE_FLOATS = 128 * 4 + 4 * HD + 2 * 2048;   // probabilities [128][4] (also: window positions) | queries [4][HD] | 2 x [64 rows][32]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Ald = smem + wave * WAVE_FLOATS;
    int* ptab = reinterpret_cast<int*>(Ald);              // [KW][2] (row, col): live only before the quad's first softmax
    float* qs = Ald + 512;
    float* buf = qs + 4 * HD;
    const int L = h0 * w0, S = h1 * w1, K = 4 * KW, wq = w0 >> 1;
    // ---- work list: XCD x (= blockIdx % 8, observed placement; speed only) takes quads [x*chunk, x*chunk + cnt) of every pair
    const int xcd = blockIdx.x & 7, chunk = (nquads + 7) >> 3;
    const int cnt = min(chunk, nquads - xcd * chunk);
    const int total = cnt > 0 ? B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    int t = (blockIdx.x >> 3) * 2 + wave;
    if (t >= total) return;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf));
    const int sl = lane >> 3, un = lane & 7;              // DMA: row within the instruction's 8, 16-byte unit of the 128-byte row
    const unsigned swz[2] = {(unsigned)((un ^ (lane >> 4)) * 16), (unsigned)((un ^ (4 + (lane >> 4))) * 16)};   // K stages, DMA instr j even / odd
    unsigned rd[8];                                       // K stages: byte offset of logical unit u in this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    const bool no_dma = dbg & CASMTR_DBG_NO_DMA, no_math = dbg & CASMTR_DBG_NO_MATH;
    const bool qlane = lane * 4 < HD;                     // H = 2: 64 floats per row -> lanes 0..15

    // prefetch registers: the quad's window positions (lane e < KW) and queries
    long long pf_y = 0, pf_x = 0;
    f32x4 pf_q[4][QV];
    auto prefetch = [&](int tt) {
        const int bb = tt / cnt, qd = xcd * chunk + tt % cnt;
        if (lane < KW) {
            const int64_t* pp = topk_pos + (((size_t)bb * nquads + qd) * KW + lane) * 2;
            pf_y = pp[0]; pf_x = pp[1];
        }
        const int ll = (2 * (qd / wq)) * w0 + 2 * (qd % wq);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < QV; ++i)
                if (qlane || i + 1 < QV)
                    pf_q[f][i] = *reinterpret_cast<const f32x4*>(q + ((size_t)bb * L + ll + (f >> 1) * w0 + (f & 1)) * HD + (i * 64 + lane) * 4);
    };
    prefetch(t);
    for (; t < total; t += stride) {
        const int b = t / cnt, quad = xcd * chunk + t % cnt;
        const int l00 = (2 * (quad / wq)) * w0 + 2 * (quad % wq);   // child f -> l00 + (f>>1)*w0 + (f&1)
        // ---- this quad's positions and queries: registers -> LDS
        if (lane < KW) { ptab[2 * lane] = (int)pf_y; ptab[2 * lane + 1] = (int)pf_x; }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < QV; ++i)
                if (qlane || i + 1 < QV) *reinterpret_cast<f32x4*>(qs + f * HD + (i * 64 + lane) * 4) = pf_q[f][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // candidate k: parent e = k / 4 (window cell), child c = k % 4 -> (row + c/2 * dil, col + c%2 * dil), clamped (:419-429)
        auto candidate = [&](int k) {
            const int kk = k < K ? k : K - 1;
            const int e = kk >> 2, c = kk & 3;
            // window positions are grid coordinates of the (h1/2) x (w1/2) grid: 32-bit arithmetic cannot overflow for S < 2^30
            const int id = (ptab[2 * e] * 2 + (c >> 1) * dil) * w1 + ptab[2 * e + 1] * 2 + (c & 1) * dil;
            return id < 0 ? 0 : (id > S - 1 ? S - 1 : id);
        };
        // DMA instruction j of pass p moves local rows 8j .. 8j+7: lane -> (row 8j + lane/8, 16-byte unit lane%8; K stages: swizzled)
        unsigned rowk[NPASS][8], rowv[NPASS][8];
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int j = 0; j < (p == 0 ? 8 : NP1); ++j) {
                const unsigned rb = (unsigned)candidate(64 * p + 8 * j + sl) * (HD * 4);
                rowk[p][j] = rb + swz[j & 1];
                rowv[p][j] = rb + (unsigned)(un * 16);
            }
        if (up_idx) {              // :450 -- every child gets the same list
            const int c0 = candidate(lane), c1 = candidate(64 + lane);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                int64_t* o = up_idx + ((size_t)b * L + l00 + (f >> 1) * w0 + (f & 1)) * K;
                if (lane < K) o[lane] = c0;
                if (64 + lane < K) o[64 + lane] = c1;
            }
        }
        float rel[HAS_REL ? H : 1][NPASS][4];
        if (HAS_REL) {
#pragma unroll
            for (int h = 0; h < H; ++h)
#pragma unroll
                for (int p = 0; p < NPASS; ++p)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int k = 64 * p + lane, lf = l00 + (f >> 1) * w0 + (f & 1);
                        rel[HAS_REL ? h : 0][p][f] = k < K ? rel_pos[(((size_t)b * H + h) * L + lf) * K + k] : 0.f;
                    }
        }
        lds_reads_done();          // ptab has been consumed (the first softmax overwrites it)
        const float* kb = key + (size_t)b * S * HD;
        const float* vb = value + (size_t)b * S * HD;
        auto issue = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int h = s / (2 * NPASS), isv = (s / NPASS) & 1, p = s % NPASS;
            const float* base = (isv ? vb : kb) + h * 32;   // wave-uniform: scalar add
#pragma unroll
            for (int j = 0; j < (p == 0 ? 8 : NP1); ++j)
                glds16(base, isv ? rowv[p][j] : rowk[p][j], buf_lds + (unsigned)((s & 1) * 8192 + j * 1024));
        };
        if (!no_dma) issue(std::integral_constant<int, 0>{});
        if (t + stride < total) prefetch(t + stride);   // lands while the stages below run; consumed at the top of the next iteration

        f32x4 lg[NPASS];           // logits of candidate 64p + lane for the 4 children
        f32x4 acc[4];
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int h = s / (2 * NPASS), isv = (s / NPASS) & 1, p = s % NPASS;
            if (!no_dma) {
                if constexpr (s + 1 < NS) {
                    lds_reads_done();
                    issue(std::integral_constant<int, s + 1>{});
                    glds_wait<((s + 1) % NPASS == 0) ? 8 : NP1>();
                } else {
                    glds_wait<0>();
                }
            }
            const char* bp = reinterpret_cast<const char*>(buf) + (s & 1) * 8192;
            if (no_math) return;
            if constexpr (!isv) {
                f32x4 qa[8], kr[8];   // operand A: lane l holds q[child l%4][h*32 + d]; operand B: this lane's candidate row
#pragma unroll
                for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qs + (lane & 3) * HD + h * 32 + 4 * u);
#pragma unroll
                for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(bp + rd[u]);
                // The logits feed a softmax only (no index depends on them), so the d-sum need not be the sequential chain: four
                // interleaved partial chains (d = 4u + c -> chain c) keep the matrix pipe busy -- a dependent v_mfma_f32_4x4x1
                // waits ~28 cycles for its accumulator (PMC: SQ_WAIT_INST_ANY = 30 % of the wave cycles with one chain)
                f32x4 a4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a4[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, a4[0], 0, 0, 0);
                    a4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, a4[1], 0, 0, 0);
                    a4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, a4[2], 0, 0, 0);
                    a4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, a4[3], 0, 0, 0);
                }
                f32x4 a;
#pragma unroll
                for (int f = 0; f < 4; ++f) a[f] = (a4[0][f] + a4[1][f]) + (a4[2][f] + a4[3][f]);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float x = temp * a[f];
                    if (HAS_REL) x = x + rel[HAS_REL ? h : 0][p][f];
                    lg[p][f] = x;
                }
                if constexpr (p == NPASS - 1) {
                    // ---- softmax over the K candidates of each child (:443), probabilities -> Ald[k][child]
                    f32x4 pr[NPASS];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        float m = (lane < K) ? lg[0][f] : -INFINITY;
                        if (NPASS == 2 && 64 + lane < K) m = fmaxf(m, lg[NPASS - 1][f]);
                        m = wave_max_f32(m);
                        float e[NPASS], sum = 0.f;
#pragma unroll
                        for (int pp = 0; pp < NPASS; ++pp) {
                            e[pp] = (64 * pp + lane < K) ? __expf(lg[pp][f] - m) : 0.f;
                            sum += e[pp];
                        }
                        const float inv = 1.0f / wave_sum_f32(sum);
#pragma unroll
                        for (int pp = 0; pp < NPASS; ++pp) pr[pp][f] = e[pp] * inv;
                    }
#pragma unroll
                    for (int pp = 0; pp < NPASS; ++pp) *reinterpret_cast<f32x4*>(Ald + (64 * pp + lane) * 4) = pr[pp];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                asm volatile("" : "+v"(lg[p]));   // keep the stage's arithmetic inside the stage
            } else {
                // ---- message += A . V over this pass's rows, two rows per instruction
                if constexpr (p == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                const float* vrow = reinterpret_cast<const float*>(bp) + lane;                       // + 64 m : rows 2m | 2m+1
                const float* prow = Ald + (64 * p + (lane >> 5)) * 4 + (lane & 3);                  // + 8 m  : P[row][child lane%4]
#pragma unroll
                for (int m = 0; m < (p == 0 ? 32 : NP1 * 4); ++m)
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
                    if (lane < 32) {
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            message[((size_t)b * L + l00 + (f >> 1) * w0 + (f & 1)) * HD + h * 32 + lane] = tot[f];
                    }
                } else {
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                }
            }
        });
        lds_reads_done();          // the last V stage has read Ald / the buffers before the next quad rewrites them
    }
}

template <int H, int NP1>
static int launch_cas(const float* q, const float* key, const float* value, const int64_t* tp, const float* rel, float temp,
                      int dil, float* message, int64_t* up_idx, int B, int h0, int w0, int h1, int w1, int KW, hipStream_t s) {
    const int nquads = (h0 / 2) * (w0 / 2);
    const size_t lds = sizeof(float) * 2 * (128 * 4 + 4 * H * 32 + 2 * 2048);   // 40 KB at H = 4: 4 workgroups (8 waves) per CU
    // persistent grid: exactly the workgroups that are resident at once (a workgroup that had to wait for a slot would start
    // its statically assigned share of the quads late)
    static int resident_tab[2][CASMTR_MAX_DEVICES] = {{0}, {0}};
    int resident = 0;
    if (const int r = rel ? resident_workgroups(resident_tab[1], cascade_attn_dma_kernel<H, NP1, true>, 128, lds, &resident)
                          : resident_workgroups(resident_tab[0], cascade_attn_dma_kernel<H, NP1, false>, 128, lds, &resident))
        return r;
    const long long work = (long long)B * nquads;
    long long blocks = resident;
    if (blocks > (work + 1) / 2) blocks = ((work + 1) / 2 + 7) / 8 * 8;
    ProfScope ps(CASMTR_PROF_CASCADE_ATTN, s, "cascade_attn_dma_kernel");
    if (rel)
        hipLaunchKernelGGL((cascade_attn_dma_kernel<H, NP1, true>), dim3((unsigned)blocks), dim3(128), lds, s, q, key, value, tp, rel,
                           message, up_idx, temp, dil, B, h0, w0, h1, w1, KW, nquads, g_debug_flags);
    else
        hipLaunchKernelGGL((cascade_attn_dma_kernel<H, NP1, false>), dim3((unsigned)blocks), dim3(128), lds, s, q, key, value, tp, rel,
                           message, up_idx, temp, dil, B, h0, w0, h1, w1, KW, nquads, g_debug_flags);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// -> CASMTR_ERR_UNSUPPORTED when the shape is outside this kernel (the caller then uses quad_attn_kernel<H,KMAX,1>)
int casmtr_cascade_attn_dma(const float* q, const float* key, const float* value, const int64_t* tp, const float* rel, float temp,
                            int dil, float* message, int64_t* up_idx, int B, int h0, int w0, int h1, int w1, int H, int KW,
                            hipStream_t s) {
    const int K = 4 * KW;
    if (K > 128 || KW > 32 || (H != 4 && H != 2)) return CASMTR_ERR_UNSUPPORTED;
#define CAS_CASE(HH)                                                                                                           \
    if (H == HH) {                                                                                                             \
        if (K <= 64) return launch_cas<HH, 0>(q, key, value, tp, rel, temp, dil, message, up_idx, B, h0, w0, h1, w1, KW, s);  \
        if (K <= 104) return launch_cas<HH, 5>(q, key, value, tp, rel, temp, dil, message, up_idx, B, h0, w0, h1, w1, KW, s); \
        return launch_cas<HH, 8>(q, key, value, tp, rel, temp, dil, message, up_idx, B, h0, w0, h1, w1, KW, s);               \
    }
    CAS_CASE(4)
    CAS_CASE(2)
#undef CAS_CASE
    return CASMTR_ERR_UNSUPPORTED;
}
