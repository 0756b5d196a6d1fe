// Token-major element kernels of the blocks that CALL the attention modules: the Mlp's depth-wise 3x3
// convolution and the LayerNorms around QuadtreeAttention / CascadeQuadtreeAttention.
//   casmtr_dwconv3x3_tokens_fwd   DWConv inside Mlp (ReLU -> depth-wise 3x3 -> GELU)   src/model/modules/transformer.py:52-94
//                                 and PosCNN (x + depth-wise 3x3)                     src/model/backbone/gvt.py:397-411
//   casmtr_layer_norm_fwd         norm1 / norm2 of QuadtreeBlock, CascadeQuadtreeBlock  transformer.py:141-196, 305-345
// The reference transposes the [B,N,C] tokens to NCHW for the convolution and back (two full copies); MIOpen has no tuned
// depth-wise fp32 kernel for these shapes on gfx950 and runs a naive one (2 ms per call at 832x832).  On token-major data the
// convolution is a pure HBM stream: channels are the contiguous axis, every thread owns four of them.
// Arithmetic (the oracle's): acc = bias[c]; for ky, kx row-major: acc = fmaf(in(y+ky-1, x+kx-1, c), w[c][ky][kx], acc) with
// taps outside the grid skipped; in() applies max(., 0) first when CASMTR_DW_PRE_RELU; then GELU (erf form) when
// CASMTR_DW_POST_GELU; then + x[y, x, c] (the raw input) when CASMTR_DW_ADD_INPUT.
// LayerNorm: mean = (sum x) / C, var = (sum (x - mean)^2) / C (two passes over registers), y = (x - mean) * rsqrt(var + eps) * g + b.
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

#define DW_TX 8

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int FLAGS>
__global__ __launch_bounds__(256) void dwconv3x3_tokens_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                               int C4, int strips_x, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int c4 = (int)(g % C4);
    long long s = g / C4;
    const int xs = (int)(s % strips_x) * DW_TX; s /= strips_x;
    const int yy = (int)(s % H);
    const int b = (int)(s / H);
    // weights of the four channels: w[c][9]
    float wt[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[i][t] = w[(size_t)(4 * c4 + i) * 9 + t];
    f32x4 acc[DW_TX];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = reinterpret_cast<const f32x4*>(bias)[c4];
#pragma unroll
    for (int j = 0; j < DW_TX; ++j) acc[j] = bv;
    const f32x4* xin = reinterpret_cast<const f32x4*>(x) + (size_t)b * H * W * C4 + c4;
    f32x4 centre[DW_TX];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int r = yy + ky - 1;
        if (r < 0 || r >= H) continue;
        f32x4 in[DW_TX + 2];
#pragma unroll
        for (int j = 0; j < DW_TX + 2; ++j) {
            const int cx = xs + j - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cx >= 0 && cx < W) v = xin[((size_t)r * W + cx) * C4];
            if (ky == 1 && j >= 1 && j <= DW_TX) centre[j - 1] = v;
            if (FLAGS & CASMTR_DW_PRE_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            in[j] = v;
        }
#pragma unroll
        for (int j = 0; j < DW_TX; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int cx = xs + j + kx - 1;
                if (cx < 0 || cx >= W) continue;   // skipped taps: identical to adding 0 * w
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = fmaf(in[j + kx][i], wt[i][3 * ky + kx], acc[j][i]);
            }
    }
    f32x4* yo = reinterpret_cast<f32x4*>(y) + (size_t)b * H * W * C4 + c4;
#pragma unroll
    for (int j = 0; j < DW_TX; ++j) {
        if (xs + j >= W) break;
        f32x4 o = acc[j];
        if (FLAGS & CASMTR_DW_POST_GELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = gelu_erf(o[i]);
        }
        if (FLAGS & CASMTR_DW_ADD_INPUT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += centre[j][i];
        }
        yo[((size_t)yy * W + xs + j) * C4] = o;
    }
}

extern "C" int casmtr_dwconv3x3_tokens_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                           int flags, casmtr_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (C % 4 != 0 || (flags & ~7) || x == y) return CASMTR_ERR_UNSUPPORTED;
    const int strips_x = (W + DW_TX - 1) / DW_TX;
    const long long total = (long long)B * H * strips_x * (C / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    ProfScope ps(CASMTR_PROF_GLUE, (hipStream_t)stream);
#define DW_LAUNCH(F)                                                                                                        \
    case F:                                                                                                                 \
        hipLaunchKernelGGL(dwconv3x3_tokens_kernel<F>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, H, W, C / 4, \
                           strips_x, total);                                                                                \
        break;
    switch (flags) {
        DW_LAUNCH(0) DW_LAUNCH(1) DW_LAUNCH(2) DW_LAUNCH(3) DW_LAUNCH(4) DW_LAUNCH(5) DW_LAUNCH(6) DW_LAUNCH(7)
    }
#undef DW_LAUNCH
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// one wave per row; VPL float4 values per lane (C = 256 * VPL).  C = 128: lanes 32..63 idle (VPL = 1, guarded).
template <int VPL>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gam,
                                                         const float* __restrict__ bet, const float* __restrict__ res,
                                                         float* __restrict__ y, long long rows, int C4, float eps, float inv_c) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + (size_t)row * C4;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < C4) v[i] = xr[c];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    s = wave_sum_f32(s);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
        }
    }
    q = wave_sum_f32(q);
    const float rstd = 1.0f / sqrtf(q * inv_c + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y) + (size_t)row * C4;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c >= C4) continue;
        const f32x4 g4 = reinterpret_cast<const f32x4*>(gam)[c], b4 = reinterpret_cast<const f32x4*>(bet)[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf((v[i][e] - mean) * rstd, g4[e], b4[e]);
        if (res) {
            const f32x4 r4 = reinterpret_cast<const f32x4*>(res)[(size_t)row * C4 + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += r4[e];
        }
        yr[c] = o;
    }
}

extern "C" int casmtr_layer_norm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                     long long rows, int C, float eps, casmtr_stream_t stream) {
    if (rows <= 0) return 0;
    if (C % 4 != 0 || C > 1024 || !gamma || !beta) return CASMTR_ERR_UNSUPPORTED;
    const int C4 = C / 4, vpl = (C4 + 63) / 64;
    const dim3 grid((unsigned)((rows + 3) / 4));
    ProfScope ps(CASMTR_PROF_GLUE, (hipStream_t)stream);
#define LN_LAUNCH(V)                                                                                                          \
    case V:                                                                                                                   \
        hipLaunchKernelGGL(layer_norm_kernel<V>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, residual, y, rows, \
                           C4, eps, 1.0f / (float)C);                                                                         \
        break;
    switch (vpl) { LN_LAUNCH(1) LN_LAUNCH(2) LN_LAUNCH(3) LN_LAUNCH(4) }
#undef LN_LAUNCH
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Window self-attention of the local blocks (GroupAttention.forward_mask: src/model/modules/cascade_attention.py:124-157,
// src/model/backbone/gvt.py:102-133) on the un-padded token-major output of the fused qkv projection.
// The reference zero-pads the grid to a multiple of ws, projects the padding too, builds the [windows, ws^2, ws^2] -1000 mask,
// and materialises Q.K^T, the scaled copy, the masked copy, the softmax and the permuted output.  For a real query the mask
// removes exactly the padded keys (exp(-1000 - max) == 0 in fp32), so the result is the softmax over the real tokens of the
// window -- which is what this kernel computes, without ever touching padding:
//   s_j = (chain_d fmaf(q[d], k_j[d])) * scale;  p_j = exp(s_j - max_j s);  o = (sum_j p_j v_j) * (1 / sum_j p_j),  j in window raster order.
// One wave per (window, head): lane <-> query token of the window (ws^2 <= 64), K / V rows of the head staged in LDS by
// coalesced 8-lanes-per-row loads and read back as broadcasts; head_dim = 32.
template <int WS>
__global__ __launch_bounds__(256) void window_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int H, int W,
                                                          int NH, int GW, int GH, float scale) {
    constexpr int T = WS * WS;
    __shared__ float kv[4][2][T][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int win = blockIdx.x;
    const int wx = win % GW, wy = (win / GW) % GH, b = win / (GW * GH);
    const int C = NH * 32;
    const size_t row_stride = (size_t)3 * C;                       // floats between consecutive tokens of qkv
    const float* base = qkv + (size_t)b * H * W * row_stride;
    float* obase = out + (size_t)b * H * W * C;
    const int rows = min(WS, H - wy * WS), cols = min(WS, W - wx * WS);
    const int n = rows * cols;                                     // real tokens of this window
    for (int h = wave; h < NH; h += 4) {
        // stage K and V rows: 8 lanes per 128-byte row
        for (int j0 = 0; j0 < n; j0 += 8) {
            const int j = j0 + (lane >> 3);
            if (j < n) {
                const int ty = wy * WS + j / cols, tx = wx * WS + j % cols;
                const float* src = base + ((size_t)ty * W + tx) * row_stride + h * 32 + (lane & 7) * 4;
                const f32x4 kk = *reinterpret_cast<const f32x4*>(src + C);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(src + 2 * C);
                *reinterpret_cast<f32x4*>(&kv[wave][0][j][(lane & 7) * 4]) = kk;
                *reinterpret_cast<f32x4*>(&kv[wave][1][j][(lane & 7) * 4]) = vv;
            }
        }
        const bool act = lane < n;
        const int qi = act ? lane : 0;
        const int qy = wy * WS + qi / cols, qx = wx * WS + qi % cols;
        const size_t tok = (size_t)qy * W + qx;
        float q[32];
        {
            const f32x4* qp = reinterpret_cast<const f32x4*>(base + tok * row_stride + h * 32);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x4 t = qp[p];
                q[4 * p] = t[0]; q[4 * p + 1] = t[1]; q[4 * p + 2] = t[2]; q[4 * p + 3] = t[3];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        float s[T];
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            float a = 0.f;
            if (j < n) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(&kv[wave][0][j][4 * p]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fmaf(q[4 * p + e], kk[e], a);
                }
                a *= scale;
                m = fmaxf(m, a);
            }
            s[j] = a;
        }
        float o[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            if (j < n) {
                const float p = __expf(s[j] - m);
                sum += p;
#pragma unroll
                for (int pz = 0; pz < 8; ++pz) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(&kv[wave][1][j][4 * pz]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * pz + e] = fmaf(p, vv[e], o[4 * pz + e]);
                }
            }
        }
        const float inv = 1.0f / sum;
        if (act) {
            f32x4* op = reinterpret_cast<f32x4*>(obase + tok * C + h * 32);
#pragma unroll
            for (int p = 0; p < 8; ++p) op[p] = f32x4{o[4 * p] * inv, o[4 * p + 1] * inv, o[4 * p + 2] * inv, o[4 * p + 3] * inv};
        }
        __builtin_amdgcn_wave_barrier();                           // all lanes done with kv[wave] before the next head overwrites it
    }
}

extern "C" int casmtr_window_attn_fwd(const float* qkv, float* out, int B, int H, int W, int nhead, int head_dim, int ws, float scale,
                                      casmtr_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (head_dim != 32 || ws != 7 || nhead <= 0) return CASMTR_ERR_UNSUPPORTED;
    const int GW = (W + ws - 1) / ws, GH = (H + ws - 1) / ws;
    ProfScope ps(CASMTR_PROF_GLUE, (hipStream_t)stream);
    hipLaunchKernelGGL(window_attn_kernel<7>, dim3((unsigned)(B * GW * GH)), dim3(256), 0, (hipStream_t)stream, qkv, out, H, W, nhead,
                       GW, GH, scale);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// POLA: patch-based overlapping self-attention of the indoor model's local blocks (NeighborWindowAttention + the window /
// neighbourhood plumbing of POLATransBlock, src/model/modules/POLAttention.py:70-172, 280-320).  Queries of a ws x ws window attend
// to the 3 x 3 windows around it (441 keys for ws = 7), with a learned bias indexed by the relative offset.
// The reference pads the normalised map with zeros (to a multiple of ws, plus one window on every side), unfolds the 9x larger
// neighbourhoods, projects them, and materialises [windows, heads, 49, 441] logits.  Here q, k0, v0 are the projections of the
// UN-padded tokens (k0, v0 without bias: the key bias shifts all logits of a row by the same amount and the value bias is added by
// the caller after the attention); a key position outside the map is the reference's zero padding: logit = bias only, value 0.
//   s_j = (chain_d fmaf(q[d], k0_j[d])) * scale + bias[rel(q, j)];  o = (sum_j exp(s_j - max) v0_j) / (sum_j exp(s_j - max)),
// running (max, sum) over the 9 neighbour windows, one 49-key window in LDS at a time.  One wave per (window, head).
template <int WS>
__global__ __launch_bounds__(256) void pola_attn_kernel(const float* __restrict__ q, const float* __restrict__ k0,
                                                        const float* __restrict__ v0, const float* __restrict__ table,
                                                        float* __restrict__ out, int H, int W, int NH, int GW, int GH, float scale) {
    constexpr int T = WS * WS, SPAN = 4 * WS - 1;   // (n_win + 1) * ws - 1 with n_win = 3
    __shared__ float kv[4][2][T][32];
    __shared__ float tab[4][SPAN * SPAN];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int win = blockIdx.x;
    const int wx = win % GW, wy = (win / GW) % GH, b = win / (GW * GH);
    const int C = NH * 32;
    const float* qb = q + (size_t)b * H * W * C;
    const float* kb = k0 + (size_t)b * H * W * C;
    const float* vb = v0 + (size_t)b * H * W * C;
    float* ob = out + (size_t)b * H * W * C;
    const int qy = lane / WS, qx = lane % WS;                      // lanes >= T idle along
    const int ty = wy * WS + qy, tx = wx * WS + qx;
    const bool act = lane < T && ty < H && tx < W;
    const size_t tok = act ? (size_t)ty * W + tx : 0;
    for (int h = wave; h < NH; h += 4) {
        for (int i = lane; i < SPAN * SPAN; i += 64) tab[wave][i] = table[(size_t)i * NH + h];
        float qr[32];
        {
            const f32x4* qp = reinterpret_cast<const f32x4*>(qb + tok * C + h * 32);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const f32x4 t = qp[p];
                qr[4 * p] = t[0]; qr[4 * p + 1] = t[1]; qr[4 * p + 2] = t[2]; qr[4 * p + 3] = t[3];
            }
        }
        float o[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        float m = -3.0e38f, sum = 0.f;
        for (int nb = 0; nb < 9; ++nb) {
            const int ny = nb / 3, nx = nb % 3;                    // neighbour window (ny - 1, nx - 1)
            const int by = (wy + ny - 1) * WS, bx = (wx + nx - 1) * WS;
            __builtin_amdgcn_wave_barrier();                       // previous chunk fully consumed
            for (int j0 = 0; j0 < T; j0 += 8) {                    // stage the window's K and V rows (zeros outside the map)
                const int j = j0 + (lane >> 3);
                if (j < T) {
                    const int yy = by + j / WS, xx = bx + j % WS;
                    f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const size_t off = ((size_t)yy * W + xx) * C + h * 32 + (lane & 7) * 4;
                        kk = *reinterpret_cast<const f32x4*>(kb + off);
                        vv = *reinterpret_cast<const f32x4*>(vb + off);
                    }
                    *reinterpret_cast<f32x4*>(&kv[wave][0][j][(lane & 7) * 4]) = kk;
                    *reinterpret_cast<f32x4*>(&kv[wave][1][j][(lane & 7) * 4]) = vv;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // relative index of key (ky, kx) of this neighbour for this lane's query: (qy - (ky + ny ws) + 3 ws - 1) * SPAN + (qx - (kx + nx ws) + 3 ws - 1)
            const int ry0 = qy - ny * WS + 3 * WS - 1, rx0 = qx - nx * WS + 3 * WS - 1;
            float s[T];
            float cm = -3.0e38f;
#pragma unroll
            for (int j = 0; j < T; ++j) {
                float a = 0.f;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(&kv[wave][0][j][4 * p]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fmaf(qr[4 * p + e], kk[e], a);
                }
                a = a * scale + tab[wave][(ry0 - j / WS) * SPAN + (rx0 - j % WS)];
                cm = fmaxf(cm, a);
                s[j] = a;
            }
            const float mn = fmaxf(m, cm);
            const float corr = __expf(m - mn);
            sum *= corr;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] *= corr;
            m = mn;
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const float p = __expf(s[j] - m);
                sum += p;
#pragma unroll
                for (int pz = 0; pz < 8; ++pz) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(&kv[wave][1][j][4 * pz]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * pz + e] = fmaf(p, vv[e], o[4 * pz + e]);
                }
            }
        }
        const float inv = 1.0f / sum;
        if (act) {
            f32x4* op = reinterpret_cast<f32x4*>(ob + tok * C + h * 32);
#pragma unroll
            for (int p = 0; p < 8; ++p) op[p] = f32x4{o[4 * p] * inv, o[4 * p + 1] * inv, o[4 * p + 2] * inv, o[4 * p + 3] * inv};
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int casmtr_pola_attn_fwd(const float* q, const float* k0, const float* v0, const float* bias_table, float* out, int B,
                                    int H, int W, int nhead, int head_dim, int ws, float scale, casmtr_stream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (head_dim != 32 || ws != 7 || nhead <= 0) return CASMTR_ERR_UNSUPPORTED;
    const int GW = (W + ws - 1) / ws, GH = (H + ws - 1) / ws;
    ProfScope ps(CASMTR_PROF_GLUE, (hipStream_t)stream);
    hipLaunchKernelGGL(pola_attn_kernel<7>, dim3((unsigned)(B * GW * GH)), dim3(256), 0, (hipStream_t)stream, q, k0, v0, bias_table,
                       out, H, W, nhead, GW, GH, scale);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
