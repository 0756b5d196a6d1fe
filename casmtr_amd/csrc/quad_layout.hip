// Layout passes into the quad-major per-head form the round-3 fine-level kernel reads (fine_quad.hip):
//   x_qm[b][hd][Q][c][d],  Q = (r/2)*(w/2) + (col/2),  c = (r&1)*2 + (col&1),  d < 32
// i.e. the reference's "b c (h t1) (w t2) -> b (h w) (t1 t2) c" of cuda_imp/.../modules/quadtree_attention.py:188-189 applied to
// q, key and value alike, heads outermost.  Pure data movement: values are copied bit for bit.
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

#define CASMTR_MAX_QLAYOUT 18   // q, k, v x 3 pyramid levels x the two directions of a layer
struct QuadLayoutBatch {
    const float* src[CASMTR_MAX_QLAYOUT];
    float* dst[CASMTR_MAX_QLAYOUT];
    int C[CASMTR_MAX_QLAYOUT], h[CASMTR_MAX_QLAYOUT], w[CASMTR_MAX_QLAYOUT];
    int tokens[CASMTR_MAX_QLAYOUT];           // 1: plain token-major [B, h*w, C] (the coarsest level's operands) instead of quad-major
    int tile_begin[CASMTR_MAX_QLAYOUT + 1];   // prefix sum of tiles per tensor (per batch element)
    int n;
};

// NCHW -> quad-major.  Tile = one head (32 channels) x one quad row (2 image rows) x 32 quads (64 pixels): 16 KB in, 16 KB out.
// In : per (channel, image row) 256 contiguous bytes, 8-byte accesses (w even is all the layout needs).
// Out: 32 quads x 512 B = one contiguous 16 KB run, 16-byte accesses (lane -> 4 consecutive d).
// LDS : t[r][ch][x] at r * (32*65 + 2) + ch * 65 + x: the transposing read (lanes = 8 d-quartets x 4 children) touches 32 different banks.
__global__ __launch_bounds__(256) void nchw_to_quads_kernel(const QuadLayoutBatch lb) {
    constexpr int RS = 32 * 65 + 2;
    __shared__ float t[2 * RS];
    // XCD-contiguous tile order: neighbouring tiles (next 64 pixels of the same image rows, next quad row) share the cache lines their
    // 256-byte segments straddle (rows are 416 / 832 B, not line multiples); round-robin placement made every XCD fetch those lines
    // for itself: FETCH_SIZE 1.5x the input bytes (profiles/r03e_pmc_fetch_size.csv)
    const int bid = xcd_chunk_remap(blockIdx.x, gridDim.x);
    int ti = 0;
#pragma unroll
    for (int i = 1; i < CASMTR_MAX_QLAYOUT; ++i)
        if (i < lb.n && bid >= lb.tile_begin[i]) ti = i;
    const int C = lb.C[ti], h = lb.h[ti], w = lb.w[ti], H = C >> 5, hq = h >> 1, wq = w >> 1, xt_n = (wq + 31) >> 5;
    const float* __restrict__ x = lb.src[ti];
    float* __restrict__ out = lb.dst[ti];
    int local = bid - lb.tile_begin[ti];
    if (lb.tokens[ti]) {
        // [B,C,HW] -> [B,HW,C]: 64 channels x 64 pixels through the same LDS region (the tile of casmtr_nchw_to_tokens, qta_fused.hip)
        float (*tt)[65] = reinterpret_cast<float (*)[65]>(t);
        const int HW = h * w, ptiles = (HW + 63) / 64;
        const int p0 = (local % ptiles) * 64, c0 = (local / ptiles) * 64;
        const int b = blockIdx.y, tid = threadIdx.x;
        const bool vin = (HW & 3) == 0;
        {
            const int p = p0 + (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = (tid >> 4) + 16 * i, c = c0 + cl;
                if (c >= C) continue;
                const float* sp = x + ((size_t)b * C + c) * HW + p;
                if (vin && p + 3 < HW) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sp);
                    tt[cl][(tid & 15) * 4 + 0] = v.x; tt[cl][(tid & 15) * 4 + 1] = v.y;
                    tt[cl][(tid & 15) * 4 + 2] = v.z; tt[cl][(tid & 15) * 4 + 3] = v.w;
                } else {
                    for (int u = 0; u < 4; ++u) if (p + u < HW) tt[cl][(tid & 15) * 4 + u] = sp[u];
                }
            }
        }
        __syncthreads();
        {
            const int cl = (tid & 15) * 4, c = c0 + cl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pl = (tid >> 4) + 16 * i, p = p0 + pl;
                if (p >= HW || c >= C) continue;   // C % 32 == 0: a 4-channel group is in range or out of range as a whole
                *reinterpret_cast<f32x4*>(out + ((size_t)b * HW + p) * C + c) = (f32x4){tt[cl][pl], tt[cl + 1][pl], tt[cl + 2][pl], tt[cl + 3][pl]};
            }
        }
        return;
    }
    const int xt = local % xt_n; local /= xt_n;
    const int qy = local % hq, hd = local / hq;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int q0 = xt * 32;   // first quad of the tile in its row
    if ((w & 3) == 0) {   // read: thread -> (channel tid/32 + 8 i, image row (tid/16)%2, 4 pixels tid%16): 16-byte accesses
        const int x4 = tid & 15, r = (tid >> 4) & 1;
        const int px = 2 * q0 + 4 * x4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = (tid >> 5) + 8 * i;
            if (px < w) {   // w % 4 == 0: the 4 pixels are in range or out of range together
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * C + hd * 32 + ch) * h + 2 * qy + r) * w + px);
                float* tp = t + r * RS + ch * 65 + 4 * x4;
                tp[0] = v.x; tp[1] = v.y; tp[2] = v.z; tp[3] = v.w;
            }
        }
    } else {              // thread -> (channel tid/64 + 4 i, image row (tid/32)%2, pixel pair tid%32): 8-byte accesses (w even)
        const int x2 = tid & 31, r = (tid >> 5) & 1;
        const int px = 2 * (q0 + x2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = (tid >> 6) + 4 * i;
            if (px < w) {
                const float2 v = *reinterpret_cast<const float2*>(x + (((size_t)b * C + hd * 32 + ch) * h + 2 * qy + r) * w + px);
                t[r * RS + ch * 65 + 2 * x2] = v.x;
                t[r * RS + ch * 65 + 2 * x2 + 1] = v.y;
            }
        }
    }
    __syncthreads();
    {   // write: thread -> (quad tid/32 + 8 i, child (tid/8)%4, d-quartet tid%8)
        const int d4 = tid & 7, c = (tid >> 3) & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = (tid >> 5) + 8 * i;
            if (q0 + q < wq) {
                const float* tp = t + (c >> 1) * RS + (4 * d4) * 65 + 2 * q + (c & 1);
                const f32x4 v = (f32x4){tp[0], tp[65], tp[130], tp[195]};
                *reinterpret_cast<f32x4*>(out + ((((size_t)b * H + hd) * (hq * wq) + qy * wq + q0 + q) * 4 + c) * 32 + 4 * d4) = v;
            }
        }
    }
}

extern "C" int casmtr_nchw_to_quads_multi(const float* const* src, float* const* dst, const int* C, const int* h, const int* w,
                                          const int* tokens, int n, int B, casmtr_stream_t stream) {
    if (n <= 0 || B <= 0) return 0;
    if (n > CASMTR_MAX_QLAYOUT) return CASMTR_ERR_UNSUPPORTED;
    QuadLayoutBatch lb{};
    lb.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const int tok = tokens ? tokens[i] != 0 : 0;
        if ((C[i] & 31) || C[i] <= 0 || h[i] <= 0 || w[i] <= 0 || (!tok && ((h[i] & 1) || (w[i] & 1)))) return CASMTR_ERR_UNSUPPORTED;
        lb.src[i] = src[i]; lb.dst[i] = dst[i]; lb.C[i] = C[i]; lb.h[i] = h[i]; lb.w[i] = w[i]; lb.tokens[i] = tok;
        lb.tile_begin[i] = tiles;
        tiles += tok ? ((h[i] * w[i] + 63) / 64) * ((C[i] + 63) / 64) : (C[i] / 32) * (h[i] / 2) * ((w[i] / 2 + 31) / 32);
    }
    lb.tile_begin[n] = tiles;
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LAYOUT, nchw_to_quads_kernel, dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, lb);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// token-major [B, h*w, C] -> quad-major: a thread moves 16 bytes (4 consecutive d of one token); reads and writes are both whole
// 128-byte head rows.  Used where the operands already are token-major (QuadtreeAttention's projections, tests).
__global__ __launch_bounds__(256) void tokens_to_quads_kernel(const float* __restrict__ x, float* __restrict__ out, long long total4,
                                                              int C, int h, int w) {
    const int H = C >> 5, wq = w >> 1, Lq = (h >> 1) * wq, c4 = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int u = (int)(e % c4);          // 16-byte unit within the token row
        const long long tok = e / c4;
        const int l = (int)(tok % (h * w));
        const long long b = tok / (h * w);
        const int r = l / w, col = l % w, hd = u >> 3, d4 = u & 7;
        const int Q = (r >> 1) * wq + (col >> 1), c = (r & 1) * 2 + (col & 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + e * 4);
        *reinterpret_cast<f32x4*>(out + ((((size_t)b * H + hd) * Lq + Q) * 4 + c) * 32 + 4 * d4) = v;
    }
}

extern "C" int casmtr_tokens_to_quads(const float* x, float* out, int B, int C, int h, int w, casmtr_stream_t stream) {
    if ((C & 31) || (h & 1) || (w & 1)) return CASMTR_ERR_UNSUPPORTED;
    const long long total4 = (long long)B * h * w * (C / 4);
    if (total4 <= 0) return 0;
    long long blocks = (total4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(CASMTR_PROF_LAYOUT, (hipStream_t)stream, "tokens_to_quads_kernel");
    hipLaunchKernelGGL(tokens_to_quads_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, total4, C, h, w);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// [B,L,K,H] int64 -> [B,H,L,K] int32
__global__ __launch_bounds__(256) void topk_idx_to_tab_kernel(const int64_t* __restrict__ idx, int32_t* __restrict__ tab, long long total,
                                                              int L, int K, int H) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int k = (int)(e % K);
        long long r = e / K;
        const int l = (int)(r % L); r /= L;
        const int hd = (int)(r % H);
        const long long b = r / H;
        tab[e] = (int32_t)idx[(((size_t)b * L + l) * K + k) * H + hd];
    }
}

extern "C" int casmtr_topk_idx_to_tab(const int64_t* idx, int32_t* tab, int B, int L, int K, int H, casmtr_stream_t stream) {
    const long long total = (long long)B * L * K * H;
    if (total <= 0) return 0;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(topk_idx_to_tab_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx, tab, total, L, K, H);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
