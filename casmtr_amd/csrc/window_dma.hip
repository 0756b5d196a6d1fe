// CascadeMatching.forward on IMPLICIT windows (src/model/functions/cascade_matching.py:63-161), gfx950.
//
// The reference hands CascadeMatching the int64 [B,N,4*KW] `upsampled_idx` that CascadeQTAttB built from
// topk_pos [B,N/4,KW,2] (cuda_imp/.../modules/quadtree_attention.py:419-450): 16x the bytes of the positions it was
// derived from, identical for the 4 children of a quad.  Here the kernel takes topk_pos itself and expands the candidate
// list in registers -- the index tensor is never read (and never has to be written by the attention layers).
//
// Key rows are staged by LDS-DMA (global_load_lds_dwordx4): a wave-instruction moves 8 candidate rows x 128 B (one cache
// line each, 8 lanes per line) into a wave-private LDS buffer without touching VGPRs; the rows are then consumed
// lane-per-candidate with conflict-free ds_read_b128 (source-side XOR swizzle of the 16-byte units).  The round-1 kernel
// read its candidate rows straight into registers, one row per lane: 64 different cache lines per load instruction, which
// ran at the L1/TA rate of 1 line per clock (0.56 ms per launch, 17 % of the HBM roofline).
//
// Arithmetic is the oracle's: operands pre-scaled by 1/sqrt(C) (division or reciprocal multiply), fp32 fmaf chain over c
// ascending, result scaled by 1/T, masked entries -1e9, argmax = first maximum of the logits.
#include <stdlib.h>
#include <string.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

// matching.hip
int casmtr_window_match_quad_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                                 int recip, float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1,
                                 int KW, int C, int dil, hipStream_t s);

// window_pair.hip: two quads per work item on a shared window box; CASMTR_ERR_UNSUPPORTED for shapes it does not cover
int casmtr_window_match_pair(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T, int recip,
                             float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW, int C, int dil,
                             hipStream_t s);

// candidate k of a quad: parent e = k / 4 (window cell), child t = k % 4 -> (row + t/2 * dil, col + t%2 * dil) on the fine grid,
// clamped like torch.clamp at modules/quadtree_attention.py:429
__device__ __forceinline__ int window_candidate(const int64_t* __restrict__ pos, int k, int K, int w1, int S, int dil) {
    const int kk = k < K ? k : K - 1;
    const int e = kk >> 2, t = kk & 3;
    const long long r = pos[2 * e] * 2 + (t >> 1) * dil, c = pos[2 * e + 1] * 2 + (t & 1) * dil;
    long long id = r * w1 + c;
    id = id < 0 ? 0 : (id > (long long)S - 1 ? (long long)S - 1 : id);
    return (int)id;
}

// One WAVE per quad of query tokens, persistent (a wave walks its share of the quads; every XCD one contiguous range of quads
// per pair), 2 waves per workgroup, no block-level synchronisation.  lane <-> candidates k = lane (pass 0) and 64 + lane (pass
// 1).  Stage (ch, p) = 32 channels of the 64 candidate rows of pass p: 8 (pass 1: NP1) DMA instructions into buffer (stage & 1),
// issued one stage ahead of the arithmetic, across quads.  The arithmetic of a stage: 8 conflict-free ds_read_b128 bring the
// lane's row chunk back, it is pre-scaled (1/sqrt(C)) and goes through 32 v_mfma_f32_4x4x1_16B_f32 -- operand A: the 4 children's
// pre-scaled query channel (the same 4 values in every block), operand B: the lane-per-candidate rows; a c-sequence of them is the
// exact c-ascending fmaf chain (tools/probes/mfma4x4_layout.hip), carried across the chunks in the accumulator.
template <int C, bool RECIP, int NP1>
__global__ __launch_bounds__(128, 2) void window_match_pos_kernel(
    const float* __restrict__ fq, const float* __restrict__ fk, const int64_t* __restrict__ topk_pos,
    const uint8_t* __restrict__ mq, const uint8_t* __restrict__ mk, float sqrtC, float inv_sqrtC, float T, float invT,
    float* __restrict__ conf, float* __restrict__ next_conf, int64_t* __restrict__ next_idx, int B, int h0, int w0, int h1, int w1,
    int KW, int dil, int nquads, int dbg) {
    constexpr int NCH = C / 32, NPASS = NP1 > 0 ? 2 : 1, NS = NCH * NPASS;
    // sqrt(C) a power of two (C = 16, 64, 256): x / sqrt(C) is exact, so (q / sqrt(C)) * (k / sqrt(C)) == (q / C) * k as real numbers and the
    // fmaf chain over them is bit-identical; the key rows are then used as they arrive (32 multiplications per stage and lane saved)
    constexpr bool P2 = C == 16 || C == 64 || C == 256;
    constexpr int QV = (C + 255) / 256;                        // float4s of a child's query row per lane
    constexpr int QS = C + 4;                                 // query row stride: the 4 children's rows (broadcast reads, lane % 4) in different banks
    constexpr int WAVE_FLOATS = 4 * QS + 2 * 64 + 2 * 2048;  // queries [4][QS] | window positions [parity][32][2] | 2 x [64 rows][32]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* qn = smem + wave * WAVE_FLOATS;
    int* ptab = reinterpret_cast<int*>(qn + 4 * QS);
    float* buf = reinterpret_cast<float*>(ptab + 2 * 64);
    const int N = h0 * w0, S = h1 * w1, K = 4 * KW, wq = w0 >> 1;
    const int xcd = blockIdx.x & 7, chunk = (nquads + 7) >> 3;
    const int cnt = min(chunk, nquads - xcd * chunk);
    const int total = cnt > 0 ? B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    const int t0 = (blockIdx.x >> 3) * 2 + wave;
    if (t0 >= total) return;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf));
    const int sl = lane >> 3, un = lane & 7;
    // DMA instruction j of a stage: 8 candidate rows (row 8 j + lane / 8, 16-byte unit `un` of the 128-byte channel chunk <- logical unit
    // un ^ ((row >> 1) & 7)); four instructions share one M0 write (quad_common.hpp: the immediate offset advances the LDS destination AND
    // the source, so instruction j % 4 is pre-biased by 3072 - 1024 (j % 4) against a base pointer that is 3072 bytes low)
    unsigned swzb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) swzb[j] = (unsigned)((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072u - 1024u * j;
    unsigned rd[8];   // read side: byte offset of logical unit u in this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    const bool no_dma = dbg & CASMTR_DBG_NO_DMA, no_math = dbg & CASMTR_DBG_NO_MATH;
    const bool qlane = lane * 4 < C;

    struct Item { int b, quad, l00; };   // pair, quad, first child's token (child f -> l00 + (f>>1)*w0 + (f&1))
    int cb = t0 / cnt, cq = t0 % cnt, cy = (xcd * chunk + cq) / wq, cx = (xcd * chunk + cq) % wq;
    const int sy = stride / wq, sx = stride % wq;
    auto take = [&](Item& it) {
        if (cb >= B) return false;
        it.b = cb; it.quad = xcd * chunk + cq; it.l00 = 2 * cy * w0 + 2 * cx;
        cq += stride;
        if (cq >= cnt) {
            while (cq >= cnt) { cq -= cnt; ++cb; }
            cy = (xcd * chunk + cq) / wq; cx = (xcd * chunk + cq) % wq;
        } else {
            cy += sy; cx += sx;
            if (cx >= wq) { cx -= wq; ++cy; }
        }
        return true;
    };
    // front end, one quad ahead: global -> registers (prefetch), registers -> LDS + DMA row offsets + masks (stage_in)
    long long pf_y = 0, pf_x = 0;
    f32x4 pf_q[4][QV];
    int pf_mq = 1;
    auto prefetch = [&](const Item& it) {
        if (lane < KW) {
            const int64_t* pp = topk_pos + (((size_t)it.b * nquads + it.quad) * KW + lane) * 2;
            pf_y = pp[0]; pf_x = pp[1];
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < QV; ++i)
                if (qlane || i + 1 < QV)
                    pf_q[f][i] = *reinterpret_cast<const f32x4*>(fq + ((size_t)it.b * N + it.l00 + (f >> 1) * w0 + (f & 1)) * C + (i * 64 + lane) * 4);
        if (mq && lane < 4) pf_mq = mq[(size_t)it.b * N + it.l00 + (lane >> 1) * w0 + (lane & 1)];
    };
    unsigned rowb[NPASS][8];
    int cnd_nx[2] = {0, 0}, mk_nx[2] = {1, 1}, mq_nx = 1;   // staged for the next quad: own candidates, their masks, the children's masks
    f32x4 q_nx[4][QV];
    auto put_queries = [&]() {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < QV; ++i)
                if (qlane || i + 1 < QV) *reinterpret_cast<f32x4*>(qn + f * QS + (i * 64 + lane) * 4) = q_nx[f][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto stage_in = [&](const Item& it, int par) {
        int* pt = ptab + par * 64;
        if (lane < KW) pt[lane] = ((int)pf_y * 2) * w1 + (int)pf_x * 2;   // first child of window cell `lane` on the fine grid (fits 32 bits)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < QV; ++i) {   // pre-scaled queries wait in registers until the current quad's last stage has read qn
                f32x4 v = pf_q[f][i];
                if constexpr (P2) {
                    const float inv_C = inv_sqrtC * inv_sqrtC;   // exact: a power of two
                    v.x *= inv_C; v.y *= inv_C; v.z *= inv_C; v.w *= inv_C;
                } else {
                    v.x = div_scalar<RECIP>(v.x, sqrtC, inv_sqrtC); v.y = div_scalar<RECIP>(v.y, sqrtC, inv_sqrtC);
                    v.z = div_scalar<RECIP>(v.z, sqrtC, inv_sqrtC); v.w = div_scalar<RECIP>(v.w, sqrtC, inv_sqrtC);
                }
                q_nx[f][i] = v;
            }
        mq_nx = pf_mq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // candidate k: parent e = k / 4 (window cell), child c = k % 4 -> (row + c/2 * dil, col + c%2 * dil), clamped (:419-429); lists are
        // padded with their last candidate.  This lane's own candidates are k = lane and 64 + lane ...
        const int coff = (((lane & 3) >> 1) * w1 + (lane & 1)) * dil, coff3 = (w1 + 1) * dil;
        auto candidate = [&](int k) {
            const int id = pt[min(k >> 2, KW - 1)] + (k < K ? coff : coff3);
            return id < 0 ? 0 : (id > S - 1 ? S - 1 : id);
        };
        cnd_nx[0] = candidate(lane);
        cnd_nx[1] = candidate(64 + lane);
        // ... and row 8 j + lane / 8 of a DMA instruction is the candidate of lane 8 j + lane / 8: one cross-lane read per instruction
        const int rb[2] = {cnd_nx[0] * (C * 4), cnd_nx[1] * (C * 4)};
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int j = 0; j < (p == 0 ? 8 : NP1); ++j)
                rowb[p][j] = (unsigned)__builtin_amdgcn_ds_bpermute((8 * j + sl) * 4, rb[p]) + swzb[j & 3];
        if (mq) {
            mk_nx[0] = mk[(size_t)it.b * S + cnd_nx[0]];
            mk_nx[1] = mk[(size_t)it.b * S + cnd_nx[1]];
        }
    };
    auto issue = [&](auto sc, int b) {
        constexpr int s = decltype(sc)::value;
        constexpr int ch = s / NPASS, p = s % NPASS;
        const float* base = fk + (size_t)b * S * C + ch * 32 - 768;   // wave-uniform: scalar arithmetic; 3072 bytes low (swzb)
        constexpr int NJ = p == 0 ? 8 : NP1;
        const unsigned dst = buf_lds + (unsigned)((s & 1) * 8192);
        if constexpr (NJ >= 4) glds_chunk(base, rowb[p][0], rowb[p][1], rowb[p][2], rowb[p][3], dst);
        if constexpr (NJ == 8) glds_chunk(base, rowb[p][4], rowb[p][5], rowb[p][6], rowb[p][7], dst + 4096);
        if constexpr (NJ % 4 == 1) glds_chunk1(base, rowb[p][NJ - 1], dst + (unsigned)((NJ - 1) * 1024));
        if constexpr (NJ % 4 == 2) glds_chunk2(base, rowb[p][NJ - 2], rowb[p][NJ - 1], dst + (unsigned)((NJ - 2) * 1024));
        if constexpr (NJ % 4 == 3) glds_chunk3(base, rowb[p][NJ - 3], rowb[p][NJ - 2], rowb[p][NJ - 1], dst + (unsigned)((NJ - 3) * 1024));
    };
    // results of the previous quad, stored one stage late (right behind a DMA wait: vmcnt counts stores too, in order)
    float pe[4][2] = {}, pnc[4] = {};
    int pam[4] = {}, pcnd[2] = {0, 0}, pend_b = 0, pend_l00 = 0;
    bool have_pend = false;
    auto flush = [&]() {
        if (have_pend) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const size_t n = (size_t)pend_b * N + pend_l00 + (f >> 1) * w0 + (f & 1);
                if (conf) {
                    if (lane < K) conf[n * K + lane] = pe[f][0];
                    if (64 + lane < K) conf[n * K + 64 + lane] = pe[f][1];
                }
                if (lane == (pam[f] & 63)) {
                    next_conf[n] = pnc[f];
                    next_idx[n] = pam[f] < 64 ? pcnd[0] : pcnd[1];
                }
            }
        }
        have_pend = false;
    };

    Item it_cur{}, it_nx{}, it_pf{};
    take(it_cur);
    prefetch(it_cur);
    stage_in(it_cur, 0);
    put_queries();
    int cnd[2] = {cnd_nx[0], cnd_nx[1]}, mkv[2] = {mk_nx[0], mk_nx[1]}, mqv = mq_nx;
    int par = 0;
    if (!no_dma) issue(std::integral_constant<int, 0>{}, it_cur.b);
    bool more = take(it_nx), has_pf = false;
    if (more) prefetch(it_nx);
    for (;; par ^= 1) {
        const float* qp = qn;
        f32x4 acc[NPASS], qa[8];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        static_for<0, NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int ch = s / NPASS, p = s % NPASS;
            if constexpr (s == NS - 1) {
                if (more) stage_in(it_nx, par ^ 1);   // the next quad's front end, then its stage 0, under this quad's last stage
            }
            if (!no_dma) {
                lds_reads_done();
                if constexpr (s + 1 < NS) {
                    issue(std::integral_constant<int, s + 1>{}, it_cur.b);
                    glds_wait<((s + 1) % NPASS == 0) ? 8 : NP1>();
                } else {
                    if (more) { issue(std::integral_constant<int, 0>{}, it_nx.b); glds_wait<8>(); }
                    else glds_wait<0>();
                }
            }
            if constexpr (s == 0) flush();
            if constexpr (s == NS - 1) {
                has_pf = more && take(it_pf);
                if (has_pf) prefetch(it_pf);   // behind the wait: a whole quad's time before stage_in consumes it
            }
            if (no_math) return;
            const char* bp = reinterpret_cast<const char*>(buf) + (s & 1) * 8192;
            f32x4 kr[8];          // operand B: this lane's candidate row chunk; operand A (qa): lane l holds qn[child l%4][c]
            if constexpr (p == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qp + (lane & 3) * QS + ch * 32 + 4 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(bp + rd[u]);
            lds_reads_done();
            if constexpr (!P2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    kr[u].x = div_scalar<RECIP>(kr[u].x, sqrtC, inv_sqrtC); kr[u].y = div_scalar<RECIP>(kr[u].y, sqrtC, inv_sqrtC);
                    kr[u].z = div_scalar<RECIP>(kr[u].z, sqrtC, inv_sqrtC); kr[u].w = div_scalar<RECIP>(kr[u].w, sqrtC, inv_sqrtC);
                }
            }
            f32x4 a = acc[p];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, a, 0, 0, 0);
            }
            acc[p] = a;
            asm volatile("" : "+v"(acc[p]));   // keep the stage's arithmetic inside the stage
        });
        // softmax over the K candidates, first argmax of the logits (cascade_matching.py:119-149); stores deferred to flush()
        if (!no_math) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int mqf = __builtin_amdgcn_readlane(mqv, f);
                float x[2] = {0.f, 0.f};
                unsigned key[2] = {0u, 0u};
#pragma unroll
                for (int p = 0; p < NPASS; ++p) {
                    const int k = p * 64 + lane;
                    if (k < K) {
                        float v = div_scalar<RECIP>(acc[p][f], T, invT);
                        if (mq && !(mqf && mkv[p])) v = NEG_FILL;
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
                pe[f][0] = e0; pe[f][1] = e1; pam[f] = am; pnc[f] = am < 64 ? e0 : e1;
            }
            pcnd[0] = cnd[0]; pcnd[1] = cnd[1]; pend_b = it_cur.b; pend_l00 = it_cur.l00; have_pend = true;
        }
        lds_reads_done();
        if (!more) break;
        put_queries();
        cnd[0] = cnd_nx[0]; cnd[1] = cnd_nx[1]; mkv[0] = mk_nx[0]; mkv[1] = mk_nx[1]; mqv = mq_nx;
        it_cur = it_nx; it_nx = it_pf; more = has_pf;
    }
    flush();
}

template <int C, bool RECIP, int NP1>
static int launch_wm_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                         float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW, int dil,
                         hipStream_t s) {
    const float sqrtC = (float)sqrt((double)C);
    const int nquads = (h0 / 2) * (w0 / 2);
    const size_t lds = sizeof(float) * 2 * (4 * (C + 4) + 2 * 64 + 2 * 2048);
    static int resident_tab[CASMTR_MAX_DEVICES] = {0};   // persistent grid: exactly the workgroups that are resident at once
    int resident = 0;
    if (const int r = resident_workgroups(resident_tab, window_match_pos_kernel<C, RECIP, NP1>, 128, lds, &resident)) return r;
    const long long work = (long long)B * nquads;
    long long blocks = resident;
    if (blocks > (work + 1) / 2) blocks = ((work + 1) / 2 + 7) / 8 * 8;
    prof_symbol_args(CASMTR_PROF_WINDOW_MATCH, "<%d,%s,%d>", C, RECIP ? "true" : "false", NP1);
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_WINDOW_MATCH, (window_match_pos_kernel<C, RECIP, NP1>), dim3((unsigned)blocks), dim3(128), lds, s, fq, fk,
                        tp, mq, mk, sqrtC, 1.0f / sqrtC, T, 1.0f / T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, nquads, g_debug_flags);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

template <int C, bool RECIP>
static int dispatch_wm_pos_k(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                             float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW,
                             int dil, hipStream_t s) {
    const int K = 4 * KW;
    if (K <= 64) return launch_wm_pos<C, RECIP, 0>(fq, fk, tp, mq, mk, T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, s);
    if (K <= 104) return launch_wm_pos<C, RECIP, 5>(fq, fk, tp, mq, mk, T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, s);
    return launch_wm_pos<C, RECIP, 8>(fq, fk, tp, mq, mk, T, conf, next_conf, next_idx, B, h0, w0, h1, w1, KW, dil, s);
}

extern "C" int casmtr_window_match_pos_fwd(const float* feat_q, const float* feat_k, const int64_t* topk_pos,
                                           const uint8_t* mask_q, const uint8_t* mask_k, float temperature, int recip,
                                           int dilated, float* conf, float* next_conf, int64_t* next_idx, int B, int h0,
                                           int w0, int h1, int w1, int KW, int C, casmtr_stream_t stream) {
    if (KW <= 0 || 4 * KW > 128 || (h0 & 1) || (w0 & 1) || (mask_q == nullptr) != (mask_k == nullptr)) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || h0 <= 0 || w0 <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // Per launch at 208x208, C = 128, B = 8: window_match_pair_kernel (window_pair.hip; the default for 5 x 5 windows, dilation 1,
    // C = 64 / 128, reciprocal scaling) 0.30 ms; window_match_pos_kernel of this file (every other shape, CASMTR_WINDOW_KERNEL=dma) 0.37 ms;
    // the round-1 wave-per-quad kernel (matching.hip, CASMTR_WINDOW_KERNEL=quad) 0.55 ms.  All three give the same bits.  The
    // variable is read per call: tests switch it.
    const char* ev = getenv("CASMTR_WINDOW_KERNEL");
    if (ev && !strcmp(ev, "quad"))
        return casmtr_window_match_quad_pos(feat_q, feat_k, topk_pos, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx,
                                            B, h0, w0, h1, w1, KW, C, dilated, s);
    if (!(ev && !strcmp(ev, "dma"))) {   // default: quad pairs on shared window boxes (window_pair.hip) where the shape allows it
        const int r = casmtr_window_match_pair(feat_q, feat_k, topk_pos, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx, B, h0,
                                               w0, h1, w1, KW, C, dilated, s);
        if (r != CASMTR_ERR_UNSUPPORTED) return r;
    }
#define WM_CASE(CC)                                                                                                              \
    if (C == CC)                                                                                                                 \
        return recip ? dispatch_wm_pos_k<CC, true>(feat_q, feat_k, topk_pos, mask_q, mask_k, temperature, conf, next_conf,      \
                                                   next_idx, B, h0, w0, h1, w1, KW, dilated, s)                                  \
                     : dispatch_wm_pos_k<CC, false>(feat_q, feat_k, topk_pos, mask_q, mask_k, temperature, conf, next_conf,     \
                                                    next_idx, B, h0, w0, h1, w1, KW, dilated, s);
    WM_CASE(128)
    WM_CASE(64)
    WM_CASE(256)
    WM_CASE(32)
#undef WM_CASE
    return CASMTR_ERR_UNSUPPORTED;
}

// ---- the explicit form, on demand: upsampled_idx [B,h0*w0,4*KW] exactly as CascadeQTAttB returns it (:419-450)
__global__ __launch_bounds__(256) void window_expand_idx_kernel(const int64_t* __restrict__ topk_pos, int64_t* __restrict__ up_idx,
                                                                int h0, int w0, int w1, int S, int KW, int dil, long long total) {
    const int K = 4 * KW, wq = w0 >> 1, Lq = (h0 >> 1) * wq, L = h0 * w0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(t % K);
        const long long tl = t / K;
        const int l = (int)(tl % L), b = (int)(tl / L);
        const int quad = (l / w0 >> 1) * wq + (l % w0 >> 1);
        up_idx[t] = window_candidate(topk_pos + ((size_t)b * Lq + quad) * KW * 2, k, K, w1, S, dil);
    }
}

extern "C" int casmtr_window_expand_idx(const int64_t* topk_pos, int64_t* up_idx, int B, int h0, int w0, int h1, int w1, int KW,
                                        int dilated, casmtr_stream_t stream) {
    if ((h0 & 1) || (w0 & 1) || KW <= 0) return CASMTR_ERR_UNSUPPORTED;
    const long long total = (long long)B * h0 * w0 * 4 * KW;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    ProfScope ps(CASMTR_PROF_WINDOW_WARP, (hipStream_t)stream, "window_expand_idx_kernel");
    hipLaunchKernelGGL(window_expand_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, topk_pos, up_idx, h0,
                       w0, w1, h1 * w1, KW, dilated, total);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
