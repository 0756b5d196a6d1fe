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
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

// matching.hip
int casmtr_window_match_quad_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                                 int recip, float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1,
                                 int KW, int C, int dil, hipStream_t s);

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

// One WAVE per quad of query tokens, 2 quads per workgroup, no block-level synchronisation.
// lane <-> candidates k = lane (pass 0) and 64 + lane (pass 1).  Stage (ch, p) = 32 channels of the 64 candidate rows of
// pass p: 8 (pass 1: NP1) DMA instructions into buffer (stage & 1), issued one stage ahead of the arithmetic.
template <int C, bool RECIP, int NP1>
__global__ __launch_bounds__(128) void window_match_pos_kernel(
    const float* __restrict__ fq, const float* __restrict__ fk, const int64_t* __restrict__ topk_pos,
    const uint8_t* __restrict__ mq, const uint8_t* __restrict__ mk, float sqrtC, float inv_sqrtC, float T, float invT,
    float* __restrict__ conf, float* __restrict__ next_conf, int64_t* __restrict__ next_idx, int h0, int w0, int h1, int w1,
    int KW, int dil, int nquads, int dbg) {
    constexpr int NCH = C / 32, NPASS = NP1 > 0 ? 2 : 1, NS = NCH * NPASS;
    constexpr int WAVE_FLOATS = 4 * C + 2 * 2048;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* qn = smem + wave * WAVE_FLOATS;       // [4 children][C] normalised queries
    float* buf = qn + 4 * C;                      // 2 x [64 rows][32 floats]
    const int b = blockIdx.y;
    const int quad = xcd_chunk_remap(blockIdx.x, gridDim.x) * 2 + wave;   // neighbouring quads (overlapping windows) share an L2
    if (quad >= nquads) return;
    const int N = h0 * w0, S = h1 * w1, K = 4 * KW;
    const int wq = w0 >> 1, qy = quad / wq, qx = quad % wq;
    int tok[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) tok[f] = (2 * qy + (f >> 1)) * w0 + 2 * qx + (f & 1);
    const int64_t* pos = topk_pos + ((size_t)b * nquads + quad) * KW * 2;
    const int c0 = window_candidate(pos, lane, K, w1, S, dil);
    const int c1 = window_candidate(pos, 64 + lane, K, w1, S, dil);
    // DMA source offsets (bytes from the pair's key base): instruction j moves local rows 8j .. 8j+7, lane -> (row 8j + lane/8,
    // physical 16-byte unit lane%8).  Physical unit p of local row r holds logical unit p ^ ((r >> 1) & 7): with 128-byte rows
    // that makes the ds_read_b128 of "lane reads unit u of row lane" hit 16 different units in every 16-lane group.
    unsigned roff[NPASS][8];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int j = 0; j < (p == 0 ? 8 : NP1); ++j) {
            const int r = 8 * j + (lane >> 3);
            const int row = window_candidate(pos, 64 * p + r, K, w1, S, dil);
            roff[p][j] = (unsigned)row * (C * 4) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        }
    unsigned rd[8];   // read side: byte offset of logical unit u in this lane's row
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
#pragma unroll
    for (int f = 0; f < 4; ++f)
        for (int c = lane; c < C; c += 64) qn[f * C + c] = div_scalar<RECIP>(fq[((size_t)b * N + tok[f]) * C + c], sqrtC, inv_sqrtC);
    int mqv[4] = {1, 1, 1, 1};
    int mk0 = 1, mk1 = 1;
    if (mq) {   // fetched before any DMA is outstanding (compiler-generated loads must not share the vmcnt window)
#pragma unroll
        for (int f = 0; f < 4; ++f) mqv[f] = mq[(size_t)b * N + tok[f]];
        mk0 = mk[(size_t)b * S + c0];
        mk1 = mk[(size_t)b * S + c1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every ordinary load above has landed
    const float* kb = fk + (size_t)b * S * C;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(buf));
    float acc[NPASS][4];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[p][f] = 0.f;

    auto issue = [&](int s) {   // s is a compile-time constant at every call site (the loop below is fully unrolled)
        const int ch = s / NPASS, p = s % NPASS;
#pragma unroll
        for (int j = 0; j < (p == 0 ? 8 : NP1); ++j)
            glds16(kb, roff[p][j] + (unsigned)(ch * 128), buf_lds + (unsigned)((s & 1) * 8192 + j * 1024));
    };
    const bool no_dma = dbg & CASMTR_DBG_NO_DMA, no_math = dbg & CASMTR_DBG_NO_MATH;
    if (!no_dma) issue(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int ch = s / NPASS, p = s % NPASS;
        if (no_dma) {
        } else if (s + 1 < NS) {
            lds_reads_done();          // the reads of stage s-1 (same buffer as stage s+1) have returned
            issue(s + 1);
            if ((s + 1) % NPASS == 0) glds_wait<8>(); else glds_wait<NP1>();   // everything but stage s+1 has landed
        } else {
            glds_wait<0>();
        }
        if (no_math) continue;
        const char* bp = reinterpret_cast<const char*>(buf) + (s & 1) * 8192;
        f32x4 kr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(bp + rd[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            kr[u].x = div_scalar<RECIP>(kr[u].x, sqrtC, inv_sqrtC); kr[u].y = div_scalar<RECIP>(kr[u].y, sqrtC, inv_sqrtC);
            kr[u].z = div_scalar<RECIP>(kr[u].z, sqrtC, inv_sqrtC); kr[u].w = div_scalar<RECIP>(kr[u].w, sqrtC, inv_sqrtC);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4* qp = reinterpret_cast<const f32x4*>(qn + f * C + ch * 32);   // broadcast reads
            float a = acc[p][f];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 qv = qp[u];
                a = __builtin_fmaf(qv.x, kr[u].x, a);
                a = __builtin_fmaf(qv.y, kr[u].y, a);
                a = __builtin_fmaf(qv.z, kr[u].z, a);
                a = __builtin_fmaf(qv.w, kr[u].w, a);
            }
            acc[p][f] = a;
        }
        // pin this stage's arithmetic in front of the next stage's asm statements: hipcc otherwise keeps only the LDS reads in
        // place (the asm "memory" clobbers order those), parks their results in scratch and runs every fmaf at the very end
#pragma unroll
        for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(acc[p][f]));
    }
    // softmax over the K candidates, first argmax of the logits (cascade_matching.py:119-149)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int n = tok[f];
        float x[2] = {0.f, 0.f};
        unsigned key[2] = {0u, 0u};
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int k = p * 64 + lane;
            if (k < K) {
                float v = div_scalar<RECIP>(acc[p][f], T, invT);
                if (mq && !(mqv[f] && (p ? mk1 : mk0))) v = NEG_FILL;
                x[p] = v; key[p] = f2ord(v);
            }
        }
        const unsigned wm = wave_max_u32(max(key[0], key[1]));
        const float m = ord2f(wm);
        float e0 = (lane < K) ? expf(x[0] - m) : 0.f;
        float e1 = (64 + lane < K) ? expf(x[1] - m) : 0.f;
        const float sm = wave_sum_f32(e0 + e1);
        e0 = e0 / sm; e1 = e1 / sm;
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

template <int C, bool RECIP, int NP1>
static int launch_wm_pos(const float* fq, const float* fk, const int64_t* tp, const uint8_t* mq, const uint8_t* mk, float T,
                         float* conf, float* next_conf, int64_t* next_idx, int B, int h0, int w0, int h1, int w1, int KW, int dil,
                         hipStream_t s) {
    const float sqrtC = (float)sqrt((double)C);
    const int nquads = (h0 / 2) * (w0 / 2);
    const size_t lds = sizeof(float) * 2 * (4 * C + 2 * 2048);
    ProfScope ps(CASMTR_PROF_WINDOW_MATCH, s);
    hipLaunchKernelGGL((window_match_pos_kernel<C, RECIP, NP1>), dim3((nquads + 1) / 2, B), dim3(128), lds, s, fq, fk, tp, mq, mk,
                       sqrtC, 1.0f / sqrtC, T, 1.0f / T, conf, next_conf, next_idx, h0, w0, h1, w1, KW, dil, nquads, g_debug_flags);
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
    // default: the round-1 wave-per-quad kernel (matching.hip) with the candidate list expanded from topk_pos in registers.
    // CASMTR_WINDOW_KERNEL=dma selects the LDS-DMA staged kernel of this file: measured 0.64 ms per launch against 0.59
    // (8 waves per CU cannot cover its per-quad prologue / epilogue latencies: DESIGN.md section 8, round 2)
    const char* ev = getenv("CASMTR_WINDOW_KERNEL");   // read per call: tests switch it
    const bool dma = ev && !strcmp(ev, "dma");
    if (!dma)
        return casmtr_window_match_quad_pos(feat_q, feat_k, topk_pos, mask_q, mask_k, temperature, recip, conf, next_conf, next_idx,
                                            B, h0, w0, h1, w1, KW, C, dilated, s);
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
    ProfScope ps(CASMTR_PROF_WINDOW_WARP, (hipStream_t)stream);
    hipLaunchKernelGGL(window_expand_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, topk_pos, up_idx, h0,
                       w0, w1, h1 * w1, KW, dilated, total);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
