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
