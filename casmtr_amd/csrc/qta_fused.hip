// Fused QuadTreeAttention kernels for gfx950 (wave64).
//   quad_attn_kernel<H,KMAX,0>  QTAttB.process_fine_level + merge share   modules/quadtree_attention.py:180-229,262-284
//   quad_attn_kernel<H,KMAX,1>  CascadeQTAttB.forward                     modules/quadtree_attention.py:400-452
//   coarse_*                    QTAttB.process_coarse_level               modules/quadtree_attention.py:161-178
// One workgroup per quad (parent token): the 4 children share the candidate list, so every gathered key / value
// row is fetched once per quad.  Nothing between the gather and the message ever goes to HBM: the reference's
// [B,L/4,4,K,H] score / softmax / int64 index intermediates (:206-223) live in LDS and registers.
// Arithmetic for anything that feeds an index: fp32 fmaf chain over d ascending, logits = fl(temp*dot), selection
// on logits ordered (logit desc, position asc) -- identical to oracle/casmtr_oracle.c.
#include <stdlib.h>
#include <string.h>
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

// cascade_dma.hip
int casmtr_cascade_attn_dma(const float* q, const float* key, const float* value, const int64_t* tp, const float* rel, float temp,
                            int dil, float* message, int64_t* up_idx, int B, int h0, int w0, int h1, int w1, int H, int KW,
                            hipStream_t s);

// fine_dma.hip
int casmtr_qta_fine_level_dma(const float* q, const float* key, const float* value, const int64_t* prev_idx, float temp, int topk,
                              float w_level, const float* acc_in, float* message, float* acc_out, float* topk_score,
                              int64_t* topk_idx, int B, int h0, int w0, int h1, int w1, int H, int Kp, hipStream_t s);

// =================================================================================================== layout
// [B,C,HW] -> [B,HW,C] for up to 9 tensors in one launch (a QTAttB call converts 3 pyramids x q,k,v).
// 64x64 tile through LDS; 16-byte global accesses on both sides (pixels contiguous on the way in, channels on the way out).
#define CASMTR_MAX_LAYOUT 9
struct LayoutBatch {
    const float* src[CASMTR_MAX_LAYOUT];
    float* dst[CASMTR_MAX_LAYOUT];
    int C[CASMTR_MAX_LAYOUT], HW[CASMTR_MAX_LAYOUT];
    int tile_begin[CASMTR_MAX_LAYOUT + 1];   // prefix sum of tiles per tensor (per batch element)
    int n;
};

__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const LayoutBatch lb) {
    __shared__ float t[64][65];
    int ti = 0;
#pragma unroll
    for (int i = 1; i < CASMTR_MAX_LAYOUT; ++i)
        if (i < lb.n && (int)blockIdx.x >= lb.tile_begin[i]) ti = i;
    const int C = lb.C[ti], HW = lb.HW[ti];
    const float* __restrict__ x = lb.src[ti];
    float* __restrict__ out = lb.dst[ti];
    const int local = blockIdx.x - lb.tile_begin[ti];
    const int ptiles = (HW + 63) / 64;
    const int p0 = (local % ptiles) * 64, c0 = (local / ptiles) * 64;
    const int b = blockIdx.y, tid = threadIdx.x;
    const bool vin = (HW & 3) == 0, vout = (C & 3) == 0;
    {   // read: thread -> (channel c = tid/16 + 16*i, 4 pixels)
        const int p = p0 + (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = (tid >> 4) + 16 * i, c = c0 + cl;
            if (c >= C) continue;
            const float* src = x + ((size_t)b * C + c) * HW + p;
            if (vin && p + 3 < HW) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                t[cl][(tid & 15) * 4 + 0] = v.x; t[cl][(tid & 15) * 4 + 1] = v.y;
                t[cl][(tid & 15) * 4 + 2] = v.z; t[cl][(tid & 15) * 4 + 3] = v.w;
            } else {
                for (int u = 0; u < 4; ++u) if (p + u < HW) t[cl][(tid & 15) * 4 + u] = src[u];
            }
        }
    }
    __syncthreads();
    {   // write: thread -> (pixel p = tid/16 + 16*i, 4 channels)
        const int cl = (tid & 15) * 4, c = c0 + cl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = (tid >> 4) + 16 * i, p = p0 + pl;
            if (p >= HW || c >= C) continue;
            float* dst = out + ((size_t)b * HW + p) * C + c;
            if (vout && c + 3 < C) {
                *reinterpret_cast<f32x4*>(dst) = (f32x4){t[cl][pl], t[cl + 1][pl], t[cl + 2][pl], t[cl + 3][pl]};
            } else {
                for (int u = 0; u < 4; ++u) if (c + u < C) dst[u] = t[cl + u][pl];
            }
        }
    }
}

extern "C" int casmtr_nchw_to_tokens_multi(const float* const* src, float* const* dst, const int* C, const int* HW, int n,
                                           int B, casmtr_stream_t stream) {
    if (n <= 0 || B <= 0) return 0;
    if (n > CASMTR_MAX_LAYOUT) return CASMTR_ERR_UNSUPPORTED;
    LayoutBatch lb{};
    lb.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        lb.src[i] = src[i]; lb.dst[i] = dst[i]; lb.C[i] = C[i]; lb.HW[i] = HW[i];
        lb.tile_begin[i] = tiles;
        tiles += ((HW[i] + 63) / 64) * ((C[i] + 63) / 64);
    }
    lb.tile_begin[n] = tiles;
    if (tiles == 0) return 0;
    ProfScope ps(CASMTR_PROF_LAYOUT, (hipStream_t)stream, "nchw_to_tokens_kernel");
    hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, lb);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" int casmtr_nchw_to_tokens(const float* x, float* out, int B, int C, int HW, casmtr_stream_t stream) {
    return casmtr_nchw_to_tokens_multi(&x, &out, &C, &HW, 1, B, stream);
}

// =================================================================================================== quad kernel
struct QuadArgs {
    const float* q;        // [B,L,H*32]
    const float* key;      // [B,S,H*32]
    const float* value;    // [B,S,H*32]
    const int64_t* pidx;   // MODE 0: prev_idx [B,Lq,Kp,H] ; MODE 1: topk_pos [B,Lq,KW,2]
    const float* rel_pos;  // MODE 1 only, nullable [B,H,L,K]
    const float* acc_in;   // MODE 0, nullable [B,Lq,H*32]
    float* message;        // nullable [B,L,H*32]
    float* acc_out;        // nullable [B,L,H*32]
    float* topk_score;     // [B,L,topk,H]
    int64_t* topk_idx;     // [B,L,topk,H]
    int64_t* up_idx;       // MODE 1, nullable [B,L,K]
    float temp, w_level;
    int topk, dilated;
    int h0, w0, h1, w1, Kp;  // Kp = parent entries per quad (K = 4*Kp)
};

template <int H, int KMAX, int MODE>
__global__ __launch_bounds__(256, (MODE == 0 ? 6 : 4)) void quad_attn_kernel(const QuadArgs a) {
    constexpr int HD = H * 32;
    constexpr int PPH = KMAX / 64;    // 64-candidate passes per head
    constexpr int E = KMAX / 16;      // elements per lane in the 16-lane-row softmax / top-k
    constexpr int NSL = 256 / (H * 8);  // candidate slices in the aggregation phase
    // LDS strides padded off the 256-byte bank period: the per-head bases of the logits / probabilities / candidate lists
    // otherwise all fall on the same bank (PMC: bank-conflict cycles 3.7x the useful LDS cycles in the cascade kernel)
    constexpr int KS = KMAX + 4;          // logits row stride (floats)
    constexpr int AS = 4 * KMAX + 4;      // probabilities per-head stride (floats)
    constexpr int CS = KMAX + 1;          // candidate list per-head stride (ints)
    constexpr int CANDN = (MODE == 0 ? H * CS : CS) + 3 & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* cand = reinterpret_cast<int*>(smem);          // [H][KMAX] (MODE 0) or [KMAX]
    float* Sld = smem + CANDN;                         // [4][H][KS] logits
    float* Ald = Sld + 4 * H * KS;                     // [H][AS] = [H][KMAX][4] probabilities
    float* red = Sld;                                  // [NSL][4][H*8] float4 partial sums (4096 floats): reuses Sld/Ald
                                                       // after a barrier (both are dead once the A.V loop has finished)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = a.w0 >> 1, Lq = (a.h0 >> 1) * wq, L = a.h0 * a.w0, S = a.h1 * a.w1;
    const int qidx = xcd_chunk_remap(blockIdx.x, gridDim.x);  // each XCD walks one contiguous range of quads (L2 locality)
    const int b = qidx / Lq, n = qidx % Lq;
    const int K = 4 * a.Kp;
    const int qy = n / wq, qx = n % wq;
    const int l00 = (2 * qy) * a.w0 + 2 * qx;  // child f -> l00 + (f>>1)*w0 + (f&1)

    // ---- phase 0: candidate list
    if (MODE == 0) {
        const int w1p = a.w1 >> 1;
        for (int e = tid; e < a.Kp * H; e += 256) {
            const int kp = e / H, h = e % H;
            const int64_t p = a.pidx[(((size_t)b * Lq + n) * a.Kp + kp) * H + h];
            const int r = (int)(p / w1p) * 2, c = (int)(p % w1p) * 2;
            int* cp = cand + h * CS + kp * 4;
            cp[0] = r * a.w1 + c;
            cp[1] = r * a.w1 + c + 1;
            cp[2] = (r + 1) * a.w1 + c;
            cp[3] = (r + 1) * a.w1 + c + 1;
        }
    } else {
        for (int e = tid; e < a.Kp; e += 256) {
            const int64_t r = a.pidx[(((size_t)b * Lq + n) * a.Kp + e) * 2 + 0] * 2;
            const int64_t c = a.pidx[(((size_t)b * Lq + n) * a.Kp + e) * 2 + 1] * 2;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int64_t id = (r + (t >> 1) * a.dilated) * a.w1 + c + (t & 1) * a.dilated;
                id = id < 0 ? 0 : (id > (int64_t)S - 1 ? (int64_t)S - 1 : id);  // torch.clamp, :429
                cand[e * 4 + t] = (int)id;
            }
        }
    }
    __syncthreads();
    if (MODE == 1 && a.up_idx) {
        for (int e = tid; e < 4 * K; e += 256) {
            const int f = e / K, k = e % K;
            a.up_idx[((size_t)b * L + l00 + (f >> 1) * a.w0 + (f & 1)) * K + k] = cand[k];
        }
    }

    // ---- phase 1: logits.  wave <-> (head, 64-candidate pass).
    // Coalesced key reads without an LDS transpose: the 8 lanes of a group share a 128-byte row (one dwordx4 piece each),
    // so every load instruction covers 8 whole cache lines (a lane-per-row walk touches 64 lines per instruction and is
    // bound by the L1/TA rate).  The sequential fmaf chain over d then runs as an 8-lane systolic array: lane p owns
    // d = 4p..4p+3 of its group's 8 rows; a row's accumulator enters at lane 0 and is handed from lane p to lane p+1 with
    // a DPP shift, so the arithmetic is exactly the reference's d-ascending chain.  The two groups of a 16-lane DPP row are
    // interleaved (even lanes / odd lanes): the hand-over is row_shr:2 with zero fill, which gives both piece-0 lanes the
    // chain's initial 0 for free (adjacent groups + row_shr:1 needed a v_mov 0 and a v_cndmask per child and step).
    // A 3-stage register barrel rotation (by the lane's piece index) skews the rows so that every lane uses the same
    // register slot in the same step.
    if (MODE == 0) for (int p = wave; p < H * PPH; p += 4) {
        const int h = p / PPH;
        const int kb0 = (p % PPH) * 64;
        const int g = ((lane >> 4) << 1) | (lane & 1), pc = (lane >> 1) & 7;
        const int* cb = cand + (MODE == 0 ? h * CS : 0);
        const float* kbase = a.key + (size_t)b * S * HD + h * 32 + pc * 4;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(kbase + (size_t)cb[min(kb0 + 8 * g + j, K - 1)] * HD);
        f32x4 qv[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
            qv[f] = *reinterpret_cast<const f32x4*>(a.q + ((size_t)b * L + l00 + (f >> 1) * a.w0 + (f & 1)) * HD + h * 32 + pc * 4);
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {   // slot j <- row (j - pc) mod 8
            f32x4 t8[8];
            const bool on = (pc & m) != 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 x0 = v[j], x1 = v[(j - m) & 7];
                t8[j].x = on ? x1.x : x0.x; t8[j].y = on ? x1.y : x0.y; t8[j].z = on ? x1.z : x0.z; t8[j].w = on ? x1.w : x0.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = t8[j];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 15; ++t) {
            const f32x4 kv = v[t & 7];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float in = dpp_zfill_f32<0x112>(acc[f]);    // row_shr:2 -> the accumulator piece p-1 produced in step t-1;
                                                             // piece 0 (lanes 0,1 of the row) reads 0: a chain starts there
                in = __builtin_fmaf(qv[f].x, kv.x, in);
                in = __builtin_fmaf(qv[f].y, kv.y, in);
                in = __builtin_fmaf(qv[f].z, kv.z, in);
                in = __builtin_fmaf(qv[f].w, kv.w, in);
                acc[f] = in;
            }
            if (t >= 7 && pc == 7) {                         // row t-7 of the group has seen all 8 pieces
                const int k = kb0 + 8 * g + (t - 7);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float lg = a.temp * acc[f];
                    if (MODE == 1 && a.rel_pos && k < K) {
                        const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
                        lg = lg + a.rel_pos[(((size_t)b * H + h) * L + lf) * K + k];
                    }
                    Sld[(f * H + h) * KS + k] = lg;
                }
            }
        }
    }    // CascadeQTAttB (MODE 1) keeps the lane-per-row walk: with K = 100 the second systolic pass would be half empty and
    // the extra registers cost it two waves per SIMD (measured 3.5 -> 4.4 ms).
    if (MODE == 1) for (int p = wave; p < H * PPH; p += 4) {
        const int h = p / PPH;
        const int k = (p % PPH) * 64 + lane;
        const bool valid = k < K;
        const int row = cand[(MODE == 0 ? h * CS : 0) + (valid ? k : 0)];
        const f32x4* kp = reinterpret_cast<const f32x4*>(a.key + ((size_t)b * S + row) * HD + h * 32);
        f32x4 kr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kr[i] = kp[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
            const cfloat_p qp = as_const(a.q + ((size_t)b * L + lf) * HD + h * 32);  // wave-uniform -> s_load
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_fmaf(qp[4 * i + 0], kr[i].x, acc);
                acc = __builtin_fmaf(qp[4 * i + 1], kr[i].y, acc);
                acc = __builtin_fmaf(qp[4 * i + 2], kr[i].z, acc);
                acc = __builtin_fmaf(qp[4 * i + 3], kr[i].w, acc);
            }
            float lg = a.temp * acc;
            if (MODE == 1 && a.rel_pos && valid) lg = lg + a.rel_pos[(((size_t)b * H + h) * L + lf) * K + k];
            Sld[(f * H + h) * KS + k] = lg;
        }
    }
    __syncthreads();

    // ---- phase 2: softmax (+ top-k) per series (child f, head h); one series per 16-lane DPP row, 4 per wave pass.
    {
        const int rowi = lane >> 4, j = lane & 15;
        for (int g = wave; g < (4 * H + 3) / 4; g += 4) {
            const int sid = g * 4 + rowi;
            const bool svalid = sid < 4 * H;
            const int f = svalid ? sid / H : 0, h = svalid ? sid % H : 0;
            float lv[E];
            unsigned key[E];
            const f32x4* sp = reinterpret_cast<const f32x4*>(Sld + (f * H + h) * KS + j * E);
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
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] = (j * E + e < K) ? __expf(lv[e] - m) : 0.f;
                s += ps[e];
            }
            s = 1.0f / row16_sum_f32(s);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] = ps[e] * s;
                Ald[h * AS + (j * E + e) * 4 + f] = ps[e];
            }
            if (MODE == 0) {
                const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
                for (int t = 0; t < a.topk; ++t) {
                    unsigned cur = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) cur = max(cur, key[e]);
                    const unsigned rm = row16_max_u32(cur);
                    const unsigned long long bal = __ballot(cur == rm);
                    const unsigned bits = (unsigned)(bal >> (rowi * 16)) & 0xFFFFu;
                    const int wj = __ffs(bits) - 1;  // first lane of the row holding the maximum -> smallest position
                    if (j == wj) {
                        bool done = false;
                        int kpos = 0;
                        float sc = 0.f;
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const bool hit = !done && key[e] == rm;
                            if (hit) { kpos = j * E + e; sc = ps[e]; key[e] = 0u; done = true; }
                        }
                        if (svalid) {
                            const size_t o = (((size_t)b * L + lf) * a.topk + t) * H + h;
                            a.topk_idx[o] = cand[h * CS + kpos];
                            a.topk_score[o] = sc;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 3: message = A . V ; thread <-> (candidate slice, head, float4 of D), 4 children accumulated together.
    {
        const int s = tid / (H * 8), item = tid % (H * 8), h = item >> 3, dq = item & 7;
        f32x4 acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int* cp = cand + (MODE == 0 ? h * CS : 0);
        const float* vb = a.value + (size_t)b * S * HD + h * 32 + dq * 4;
#pragma unroll 8
        for (int k = s; k < K; k += NSL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vb + (size_t)cp[k] * HD);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(Ald + h * AS + k * 4);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                acc[f].x = __builtin_fmaf(a4[f], v.x, acc[f].x);
                acc[f].y = __builtin_fmaf(a4[f], v.y, acc[f].y);
                acc[f].z = __builtin_fmaf(a4[f], v.z, acc[f].z);
                acc[f].w = __builtin_fmaf(a4[f], v.w, acc[f].w);
            }
        }
        f32x4* r4 = reinterpret_cast<f32x4*>(red);
        __syncthreads();  // every thread is done reading Ald before `red` overwrites it
#pragma unroll
        for (int f = 0; f < 4; ++f) r4[(s * 4 + f) * (H * 8) + item] = acc[f];
        __syncthreads();
        if (tid < 4 * H * 8) {
            const int f = tid / (H * 8), it = tid % (H * 8);
            f32x4 tot = r4[f * (H * 8) + it];
#pragma unroll
            for (int ss = 1; ss < NSL; ++ss) {
                const f32x4 p = r4[(ss * 4 + f) * (H * 8) + it];
                tot.x += p.x; tot.y += p.y; tot.z += p.z; tot.w += p.w;
            }
            const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
            const size_t o = ((size_t)b * L + lf) * HD + it * 4;
            if (a.message) *reinterpret_cast<f32x4*>(a.message + o) = tot;
            if (a.acc_out) {
                f32x4 base = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.acc_in) base = *reinterpret_cast<const f32x4*>(a.acc_in + ((size_t)b * Lq + n) * HD + it * 4);
                f32x4 r;  // final = final[parent] + m * weight   (:277-281): separate multiply and add
                r.x = base.x + tot.x * a.w_level; r.y = base.y + tot.y * a.w_level;
                r.z = base.z + tot.z * a.w_level; r.w = base.w + tot.w * a.w_level;
                *reinterpret_cast<f32x4*>(a.acc_out + o) = r;
            }
        }
    }
}

template <int H, int KMAX, int MODE>
static int launch_quad(const QuadArgs& a, int B, hipStream_t s) {
    constexpr int CANDN = (MODE == 0 ? H * (KMAX + 1) : KMAX + 1) + 3 & ~3;
    constexpr int BODY = 4 * H * (KMAX + 4) + H * (4 * KMAX + 4);
    const size_t lds = sizeof(float) * (CANDN + (BODY > 4096 ? BODY : 4096));
    const int Lq = (a.h0 / 2) * (a.w0 / 2);
    // the attribute is per device: set on every call (a host-side table write) so that a process driving several GPUs, or
    // calling from several threads, never launches the 74 KB instantiations against the 64 KB default
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(quad_attn_kernel<H, KMAX, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ProfScope ps(MODE == 0 ? (KMAX <= 64 ? CASMTR_PROF_QTA_FINE : CASMTR_PROF_QTA_FINE2) : CASMTR_PROF_CASCADE_ATTN, s, "quad_attn_kernel (token-major)");
    hipLaunchKernelGGL((quad_attn_kernel<H, KMAX, MODE>), dim3(Lq * B), dim3(256), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

template <int MODE>
static int dispatch_quad(const QuadArgs& a, int B, int H, hipStream_t s) {
    const int K = 4 * a.Kp;
#define QUAD_CASE(HH)                                                    \
    if (H == HH) {                                                       \
        if (K <= 64) return launch_quad<HH, 64, MODE>(a, B, s);          \
        if (K <= 128) return launch_quad<HH, 128, MODE>(a, B, s);        \
        if (K <= 256) return launch_quad<HH, 256, MODE>(a, B, s);        \
        return CASMTR_ERR_UNSUPPORTED;                                   \
    }
    QUAD_CASE(8)
    QUAD_CASE(4)
    QUAD_CASE(2)
#undef QUAD_CASE
    return CASMTR_ERR_UNSUPPORTED;
}

extern "C" int casmtr_qta_fine_level_fwd(const float* q, const float* key, const float* value, const int64_t* prev_idx,
                                         float temp, int topk, float w_level, const float* acc_in, float* message,
                                         float* acc_out, float* topk_score, int64_t* topk_idx, int B, int h0, int w0,
                                         int h1, int w1, int H, int D, int Kp, casmtr_stream_t stream) {
    if (D != 32 || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1) || topk > 4 * Kp) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || h0 <= 0 || w0 <= 0) return 0;
    // Two kernels, identical results.  Default for 4*Kp <= 64 (the finest level of every shipped config): the persistent
    // wave-per-(quad, head) LDS-DMA + MFMA kernel (fine_dma.hip; 0.29 ms per launch at 104x104, K = 64, B = 8 against 0.335 for
    // the round-1 kernel); the round-1 workgroup-per-quad kernel below for longer lists (K = 128 with top-16: 0.21 ms against
    // 0.25 -- the iterated top-k dominates and four waves share it).  (A third one, values in registers instead of LDS at 12 instead
    // of 8 waves per CU, measured 0.285 against 0.291 ms and nothing in the whole step; pruned in round 5)
    // CASMTR_FINE_KERNEL=dma | quad forces one of them where the shape allows (read per call: tests switch it).
    {
        const char* ev = getenv("CASMTR_FINE_KERNEL");
        const bool force_dma = ev && !strcmp(ev, "dma"), force_quad = ev && !strcmp(ev, "quad");
        if (force_dma || (!force_quad && 4 * Kp <= 64)) {
            const int r = casmtr_qta_fine_level_dma(q, key, value, prev_idx, temp, topk, w_level, acc_in, message, acc_out, topk_score,
                                                    topk_idx, B, h0, w0, h1, w1, H, Kp, (hipStream_t)stream);
            if (r != CASMTR_ERR_UNSUPPORTED) return r;
        }
    }
    QuadArgs a{};
    a.q = q; a.key = key; a.value = value; a.pidx = prev_idx; a.rel_pos = nullptr; a.acc_in = acc_in;
    a.message = message; a.acc_out = acc_out; a.topk_score = topk_score; a.topk_idx = topk_idx; a.up_idx = nullptr;
    a.temp = temp; a.w_level = w_level; a.topk = topk; a.dilated = 1;
    a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.Kp = Kp;
    return dispatch_quad<0>(a, B, H, (hipStream_t)stream);
}

extern "C" int casmtr_cascade_attn_fwd(const float* q, const float* key, const float* value, const int64_t* topk_pos,
                                       const float* rel_pos, float temp, int dilated, float* message, int64_t* up_idx,
                                       int B, int h0, int w0, int h1, int w1, int nhead, int D, int KW,
                                       casmtr_stream_t stream) {
    if (D != 32 || (h0 & 1) || (w0 & 1)) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || h0 <= 0 || w0 <= 0) return 0;
    // default: the wave-per-quad LDS-DMA kernel (cascade_dma.hip); CASMTR_CASCADE_KERNEL=quad selects the round-1
    // workgroup-per-quad kernel below (also the path for shapes the DMA kernel does not cover: nhead 8, 4*KW > 128)
    const char* ev = getenv("CASMTR_CASCADE_KERNEL");   // read per call: tests switch it
    const bool dma = !(ev && !strcmp(ev, "quad"));
    if (dma) {
        const int r = casmtr_cascade_attn_dma(q, key, value, topk_pos, rel_pos, temp, dilated, message, up_idx, B, h0, w0, h1, w1,
                                              nhead, KW, (hipStream_t)stream);
        if (r != CASMTR_ERR_UNSUPPORTED) return r;
    }
    QuadArgs a{};
    a.q = q; a.key = key; a.value = value; a.pidx = topk_pos; a.rel_pos = rel_pos; a.acc_in = nullptr;
    a.message = message; a.acc_out = nullptr; a.topk_score = nullptr; a.topk_idx = nullptr; a.up_idx = up_idx;
    a.temp = temp; a.w_level = 1.f; a.topk = 0; a.dilated = dilated;
    a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.Kp = KW;
    return dispatch_quad<1>(a, B, nhead, (hipStream_t)stream);
}

// =================================================================================================== coarsest level
// (1) logits[bh][l][s] = temp * <q[l,h,:], k[s,h,:]> on the fp32 matrix cores.  v_mfma_f32_32x32x2_f32 is an exact
//     k-ordered fmaf chain (MI355X guide), so the result is bit-identical to the oracle's sequential chain.
__global__ __launch_bounds__(256) void coarse_logits_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            float* __restrict__ Sg, float temp, int L, int S, int Spad,
                                                            int H) {
    __shared__ float Qs[64][33];
    __shared__ float Ks[64][33];
    const int bh = blockIdx.z, b = bh / H, h = bh % H;
    const int l0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int row = tid >> 2, c0 = (tid & 3) * 8;
        const int l = l0 + row, s = s0 + row;
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f}, q0 = z, q1 = z, k0 = z, k1 = z;
        if (l < L) {
            const f32x4* p = reinterpret_cast<const f32x4*>(q + (((size_t)b * L + l) * H + h) * 32 + c0);
            q0 = p[0]; q1 = p[1];
        }
        if (s < S) {
            const f32x4* p = reinterpret_cast<const f32x4*>(k + (((size_t)b * S + s) * H + h) * 32 + c0);
            k0 = p[0]; k1 = p[1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Qs[row][c0 + i] = q0[i]; Qs[row][c0 + 4 + i] = q1[i];
            Ks[row][c0 + i] = k0[i]; Ks[row][c0 + 4 + i] = k1[i];
        }
    }
    __syncthreads();
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float av = Qs[wr * 32 + (lane & 31)][2 * kk + (lane >> 5)];
        const float bv = Ks[wc * 32 + (lane & 31)][2 * kk + (lane >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    const int s = s0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = l0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (l < L) Sg[((size_t)bh * L + l) * Spad + s] = temp * acc[r];
    }
}

// (2) one wave per (b,h,l) row: softmax statistics (max, sum) and top-k.  The probabilities themselves are NOT written back: the
//     A.V kernel recomputes exp(x - max) / sum from the logits (fast exp, reciprocal multiply: the message carries a 1e-4
//     tolerance and no index depends on it), which makes this pass a pure read of the [B,H,L,S_pad] workspace (122 MB less HBM / Infinity Cache traffic per call at 26x26, B = 8).
template <int EMAX>
__global__ __launch_bounds__(256) void coarse_row_kernel(const float* __restrict__ Sg, float* __restrict__ rowstat, float* __restrict__ topk_score,
                                                         int64_t* __restrict__ topk_idx, int32_t* __restrict__ topk_tab, int topk, int B, int L, int S,
                                                         int Spad, int H) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowid = blockIdx.x * 4 + wave;
    if (rowid >= B * H * L) return;
    const int bh = rowid / L, l = rowid % L, b = bh / H, h = bh % H;
    const float* row = Sg + (size_t)rowid * Spad;
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
    for (int e = 0; e < EMAX; ++e) ps[e] = ps[e] / s;
    if (lane == 0) { rowstat[2 * (size_t)rowid] = m; rowstat[2 * (size_t)rowid + 1] = s; }
    // ---- top-k, fast path.  (logit desc, position asc) is a total order, so the answer is the sorted prefix of ANY superset of
    // the k best.  theta := the k-th largest of the 64 per-lane maxima (a 64-lane bitonic sort of keys): at least k elements
    // are >= theta, and for rows that are not pathologically concentrated in a few lanes at most 64 are.  Those are compacted
    // in position order to one per lane through wave-private LDS, sorted across the wave (bitonic, 21 compare-exchange
    // steps on (key, position)) and the first k lanes store the list.  Probabilities are recomputed from the key with the
    // formula that produced ps[] above, so they are bit-identical.  Anything else (more than 64 survivors, k > 64) takes
    // the iterative wave-argmax loop below, which costs ~50 instructions per extracted element.
    if (topk <= 64) {
        __shared__ unsigned cbuf[4][2][64];
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
            unsigned* ck = cbuf[wave][0];
            unsigned* cp = cbuf[wave][1];
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
                if (topk_tab) topk_tab[(size_t)rowid * topk + lane] = (int32_t)sp;   // [B,H,L,topk]: the finer level's parents (fine_quad.hip)
            }
            return;
        }
    }
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
                        if (topk_tab) topk_tab[(size_t)rowid * topk + t] = e * 64 + src;
                        key[e] = 0u;
                    }
                }
            }
        }
    }
}

// (3) message[b,l,h,:] = sum_s P[bh][l][s] * v[b,s,h,:] on the fp32 matrix cores (s ascending chain).
__global__ __launch_bounds__(256) void coarse_av_kernel(const float* __restrict__ Pg, const float* __restrict__ rowstat,
                                                        const float* __restrict__ v, float* __restrict__ message,
                                                        float* __restrict__ acc_out, float w_level, int L, int S, int Spad, int H) {
    __shared__ float Ps[128][33];
    __shared__ float Vs[32][33];
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int l0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // chunk s0 of P (128 x 32; thread -> row tid>>1, 16 columns) and V (32 x 32; thread -> row tid>>3, 4 columns) is fetched
    // into registers while the MFMAs of chunk s0 - 32 run (the loop used to be load -> barrier -> MFMA -> barrier, and with
    // 1.5 workgroups per CU nothing else covered the load latency)
    const int prow = tid >> 1, pc0 = (tid & 1) * 16, pl = l0 + prow;
    const int vr = tid >> 3, vc = (tid & 7) * 4;
    const size_t prow_id = (size_t)bh * L + (pl < L ? pl : L - 1);
    const float* pbase = Pg + prow_id * Spad + pc0;   // logits; probability = exp(x - max) / sum
    const float rmax = rowstat[2 * prow_id], rinv = 1.0f / rowstat[2 * prow_id + 1];
    f32x4 pp[4], vv;
    auto fetch = [&](int s0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = *reinterpret_cast<const f32x4*>(pbase + s0 + 4 * i);
        vv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s0 + vr < S) vv = *reinterpret_cast<const f32x4*>(v + (((size_t)b * S + s0 + vr) * H + h) * 32 + vc);
    };
    fetch(0);
    for (int s0 = 0; s0 < Spad; s0 += 32) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // zero beyond the last key (the logits workspace holds temp * 0 there) and beyond the last row
                const int sc = s0 + pc0 + 4 * i + c;
                Ps[prow][pc0 + 4 * i + c] = (pl < L && sc < S) ? __expf(pp[i][c] - rmax) * rinv : 0.f;   // message: 1e-4 tolerance, no index depends on it
            }
        }
        Vs[vr][vc + 0] = vv.x; Vs[vr][vc + 1] = vv.y; Vs[vr][vc + 2] = vv.z; Vs[vr][vc + 3] = vv.w;
        if (s0 + 32 < Spad) fetch(s0 + 32);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float av = Ps[wave * 32 + (lane & 31)][2 * kk + (lane >> 5)];
            const float bv = Vs[2 * kk + (lane >> 5)][lane & 31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    const int d = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int l = l0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (l < L) {
            const size_t o = (((size_t)b * L + l) * H + h) * 32 + d;
            if (message) message[o] = acc[r];
            if (acc_out) acc_out[o] = acc[r] * w_level;  // final_message = m * weight[0]  (:274)
        }
    }
}

// coarse_tile.hip
int casmtr_qta_coarse_level_tile(const float* q, const float* k, const float* v, float temp, int topk, float w_level, float* message,
                                 float* acc_out, float* topk_score, int64_t* topk_idx, int32_t* topk_tab, int B, int L, int S, int H,
                                 hipStream_t s);

// Two implementations, identical indices.  Default (round 4): the register-tile kernel (coarse_tile.hip: logits born in the
// selection's layout, no workspace; S <= 1024, topk <= 60).  CASMTR_COARSE_KERNEL=three selects the round-1 three-kernel path below
// (also the fallback for shapes outside the tile kernel).  Read per call.  (The round-2 LDS-tile kernel, 197-206 us per call against
// 143-153 for the three kernels and 104 for the tile kernel at 26x26, H = 8, B = 8, was pruned in round 5.)
enum { COARSE_TILE = 0, COARSE_THREE = 1 };
static int coarse_mode(int S, int topk) {
    const char* ev = getenv("CASMTR_COARSE_KERNEL");
    if (ev && !strcmp(ev, "three")) return COARSE_THREE;
    return (S <= 1024 && topk >= 1 && topk <= 60 && topk <= S) ? COARSE_TILE : COARSE_THREE;
}

extern "C" size_t casmtr_qta_coarse_level_ws_floats_k(int B, int L, int S, int H, int topk) {
    if (coarse_mode(S, topk) != COARSE_THREE) return 1;
    const size_t Spad = ((size_t)S + 63) / 64 * 64;
    return (size_t)B * H * L * Spad + 2 * (size_t)B * H * L;   // logits + per-row (max, sum)
}
extern "C" size_t casmtr_qta_coarse_level_ws_floats(int B, int L, int S, int H) {   // topk unknown: the three-kernel path's need
    return casmtr_qta_coarse_level_ws_floats_k(B, L, S, H, 0);
}

extern "C" int casmtr_topk_idx_to_tab(const int64_t* idx, int32_t* tab, int B, int L, int K, int H, casmtr_stream_t stream);

extern "C" int casmtr_qta_coarse_level_fwd(const float* q, const float* k, const float* v, float temp, int topk,
                                           float w_level, float* logits_ws, float* message, float* acc_out,
                                           float* topk_score, int64_t* topk_idx, int B, int L, int S, int H, int D,
                                           casmtr_stream_t stream) {
    return casmtr_qta_coarse_level_tab_fwd(q, k, v, temp, topk, w_level, logits_ws, message, acc_out, topk_score, topk_idx, nullptr, B,
                                           L, S, H, D, stream);
}

extern "C" int casmtr_qta_coarse_level_tab_fwd(const float* q, const float* k, const float* v, float temp, int topk,
                                               float w_level, float* logits_ws, float* message, float* acc_out,
                                               float* topk_score, int64_t* topk_idx, int32_t* topk_tab, int B, int L, int S, int H,
                                               int D, casmtr_stream_t stream) {
    if (D != 32 || topk > S) return CASMTR_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0 || S <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int mode = coarse_mode(S, topk);
    if (mode == COARSE_TILE) {
        const int r = casmtr_qta_coarse_level_tile(q, k, v, temp, topk, w_level, message, acc_out, topk_score, topk_idx, topk_tab, B, L, S, H, s);
        if (r != CASMTR_ERR_UNSUPPORTED) return r;
    }
    // casmtr_qta_coarse_level_ws_floats_k() promised a 1-float workspace for the tile mode: if that kernel still declines
    // the shape (e.g. S * H * 128 >= 2^31 in the tile kernel) the three-kernel path below must NOT run on that dummy workspace
    if (mode != COARSE_THREE) return CASMTR_ERR_UNSUPPORTED;
    if (!logits_ws || !topk_score || !topk_idx) return CASMTR_ERR_UNSUPPORTED;   // the three-kernel path needs its workspace and writes both lists
    const int Spad = (S + 63) / 64 * 64;
    {
        ProfScope ps(CASMTR_PROF_COARSE_LOGITS, s, "coarse_logits_kernel");
        hipLaunchKernelGGL(coarse_logits_kernel, dim3(Spad / 64, (L + 63) / 64, B * H), dim3(256), 0, s, q, k, logits_ws,
                           temp, L, S, Spad, H);
    }
    CASMTR_CHECK_LAUNCH();
    const int rows = B * H * L;
    float* rowstat = logits_ws + (size_t)rows * Spad;
    const dim3 rg((rows + 3) / 4);
    const int E = Spad / 64;
    prof_begin(CASMTR_PROF_COARSE_ROW, s);
    prof_symbol(CASMTR_PROF_COARSE_ROW, "coarse_row_kernel<E>");
    if (E <= 4)
        hipLaunchKernelGGL(coarse_row_kernel<4>, rg, dim3(256), 0, s, logits_ws, rowstat, topk_score, topk_idx, topk_tab, topk, B, L, S, Spad, H);
    else if (E <= 8)
        hipLaunchKernelGGL(coarse_row_kernel<8>, rg, dim3(256), 0, s, logits_ws, rowstat, topk_score, topk_idx, topk_tab, topk, B, L, S, Spad, H);
    else if (E <= 12)
        hipLaunchKernelGGL(coarse_row_kernel<12>, rg, dim3(256), 0, s, logits_ws, rowstat, topk_score, topk_idx, topk_tab, topk, B, L, S, Spad, H);
    else if (E <= 16)
        hipLaunchKernelGGL(coarse_row_kernel<16>, rg, dim3(256), 0, s, logits_ws, rowstat, topk_score, topk_idx, topk_tab, topk, B, L, S, Spad, H);
    else if (E <= 32)
        hipLaunchKernelGGL(coarse_row_kernel<32>, rg, dim3(256), 0, s, logits_ws, rowstat, topk_score, topk_idx, topk_tab, topk, B, L, S, Spad, H);
    else
        return CASMTR_ERR_UNSUPPORTED;
    prof_end(CASMTR_PROF_COARSE_ROW, s);
    CASMTR_CHECK_LAUNCH();
    {
        ProfScope ps(CASMTR_PROF_COARSE_AV, s, "coarse_av_kernel");
        hipLaunchKernelGGL(coarse_av_kernel, dim3((L + 127) / 128, B * H), dim3(256), 0, s, logits_ws, rowstat, v, message, acc_out,
                           w_level, L, S, Spad, H);
    }
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== window generator
__global__ __launch_bounds__(256) void window_warp_idx_kernel(const int64_t* __restrict__ idx, int64_t* __restrict__ out,
                                                              long long total, int H, int W, int ws) {
    const int r = ws / 2, ww = ws * ws;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total * ww;
         t += (long long)gridDim.x * blockDim.x) {
        const long long tok = t / ww;
        const int e = (int)(t % ww);
        const long long y = idx[tok] / W, x = idx[tok] % W;
        const long long uy = y - r < 0 ? y - r : 0, ux = x - r < 0 ? x - r : 0;
        const long long oy = y + r >= H ? y + r - (H - 1) : 0, ox = x + r >= W ? x + r - (W - 1) : 0;
        out[t * 2 + 0] = y + (e / ws - r) - uy - oy;
        out[t * 2 + 1] = x + (e % ws - r) - ux - ox;
    }
}

extern "C" int casmtr_window_warp_idx(const int64_t* idx, int64_t* out, int B, int N, int H, int W, int ws,
                                      casmtr_stream_t stream) {
    const long long total = (long long)B * N;
    if (total <= 0) return 0;
    long long blocks = (total * ws * ws + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(CASMTR_PROF_WINDOW_WARP, (hipStream_t)stream, "window_warp_idx_kernel");
    hipLaunchKernelGGL(window_warp_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx, out, total,
                       H, W, ws);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" int casmtr_abi_version(void) { return CASMTR_ABI_VERSION; }
