// The callers either side of the attention kernels, token-major throughout, for gfx950.
//   casmtr_linear_fwd       q_proj / k_proj / v_proj (nn.Conv2d(dim,dim,1)) and proj (nn.Linear) of QuadtreeAttention /
//                           CascadeQuadtreeAttention        src/model/modules/quadtree_attention.py:31-33,44,79-81,98,158-160,168
//   casmtr_token_pool_fwd   the F.avg_pool2d(kernel 2, stride 2) pyramid loop    src/model/modules/quadtree_attention.py:82-90
// The reference moves [B,N,C] tokens to NCHW (permute + contiguous, :73-74), convolves, pools, and QTAttB moves every
// level back to token-major (cuda_imp/.../modules/quadtree_attention.py:165-167,185-186).  Here the projections are a
// token-major NT GEMM whose output the level kernels read directly: both layout changes and the NCHW pyramid copies
// disappear.  Arithmetic (the oracle's): y[m,n] = fl32(chain_k fmaf(x[m,k], w[n,k], acc), acc0 = 0, k ascending) + bias[n];
// pooled = (((a + b) + c) + d) * 0.25 in (row, col) order of the 2x2 window (torch's avg_pool2d accumulation order).
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

#define LIN_BM 128
#define LIN_BN 128
#define LIN_BK 32
#define LIN_MAXP 4

struct LinearBatch {
    const float* x[LIN_MAXP];
    const float* w[LIN_MAXP];
    const float* bias[LIN_MAXP];  // nullable
    float* y[LIN_MAXP];
};

// 128x128 block tile, 4 waves x (64x64), v_mfma_f32_32x32x2_f32 (an exact k-ordered fmaf chain), BK = 32 with the next
// k-tile prefetched into registers while the MFMAs of the current one run.  The weight panel (N*K*4 <= 256 KB) lives in
// L2; consecutive tiles of one activation row-panel are placed on the same XCD.
__global__ __launch_bounds__(256, 3) void linear_nt_kernel(const LinearBatch lb, int M, int N, int K, int NJB) {
    __shared__ __attribute__((aligned(16))) float As[LIN_BM][33];
    __shared__ __attribute__((aligned(16))) float Bs[LIN_BN][33];
    const int t = xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int tI = t / NJB, tJ = t - tI * NJB;
    const int p = blockIdx.y, i0 = tI * LIN_BM, j0 = tJ * LIN_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const float* __restrict__ X = lb.x[p];
    const float* __restrict__ W = lb.w[p];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid >> 1, lc0 = (tid & 1) * 16;
    const float* ap = X + (size_t)(i0 + lrow < M ? i0 + lrow : M - 1) * K + lc0;
    const float* bp = W + (size_t)(j0 + lrow < N ? j0 + lrow : N - 1) * K + lc0;
    f32x4 av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
        bv[i] = *reinterpret_cast<const f32x4*>(bp + 4 * i);
    }
    for (int k0 = 0; k0 < K; k0 += LIN_BK) {
        __syncthreads();  // previous k-tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                As[lrow][lc0 + 4 * i + c] = av[i][c];
                Bs[lrow][lc0 + 4 * i + c] = bv[i][c];
            }
        if (k0 + LIN_BK < K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = *reinterpret_cast<const f32x4*>(ap + k0 + LIN_BK + 4 * i);
                bv[i] = *reinterpret_cast<const f32x4*>(bp + k0 + LIN_BK + 4 * i);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < LIN_BK / 2; ++kk) {
            const int kc = 2 * kk + (lane >> 5), rr = lane & 31;
            const float a0 = As[wr * 64 + rr][kc], a1 = As[wr * 64 + 32 + rr][kc];
            const float b0 = Bs[wc * 64 + rr][kc], b1 = Bs[wc * 64 + 32 + rr][kc];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // epilogue straight from the accumulators: lane (hi, ln) holds rows {(r&3) + 8(r>>2) + 4hi}, column ln of each 32x32
    // block, so every store instruction writes two 128-byte row segments
    const int hi = lane >> 5, ln = lane & 31;
    const float* __restrict__ bias = lb.bias[p];
    float* __restrict__ Y = lb.y[p];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int gj = j0 + wc * 64 + tj * 32 + ln;
        if (gj >= N) continue;
        const float bj = bias ? bias[gj] : 0.f;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = i0 + wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (gi < M) Y[(size_t)gi * N + gj] = bias ? acc[ti][tj][r] + bj : acc[ti][tj][r];
            }
    }
}

extern "C" int casmtr_linear_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                                 int nprob, int M, int N, int K, casmtr_stream_t stream) {
    if (nprob <= 0 || M <= 0 || N <= 0) return 0;
    if (nprob > LIN_MAXP || K <= 0 || K % LIN_BK != 0) return CASMTR_ERR_UNSUPPORTED;
    LinearBatch lb{};
    for (int i = 0; i < nprob; ++i) {
        lb.x[i] = x[i]; lb.w[i] = w[i]; lb.bias[i] = bias ? bias[i] : nullptr; lb.y[i] = y[i];
    }
    const int NIB = (M + LIN_BM - 1) / LIN_BM, NJB = (N + LIN_BN - 1) / LIN_BN;
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LINEAR, linear_nt_kernel, dim3(NIB * NJB, nprob), dim3(256), 0, (hipStream_t)stream, lb, M, N, K, NJB);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// =================================================================================================== token pyramid
#define POOL_MAXT 4

struct PoolBatch {
    const float* src[POOL_MAXT];
    float* dst[POOL_MAXT];
};

// one thread per output float4: [B,H,W,C] -> [B,H/2,W/2,C]; a row of C/4 threads reads four contiguous token rows
__global__ __launch_bounds__(256) void token_pool_kernel(const PoolBatch pb, int H, int W, int C4, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int Ho = H >> 1, Wo = W >> 1;
    const int c4 = (int)(g % C4);
    long long tkn = g / C4;
    const int xo = (int)(tkn % Wo); tkn /= Wo;
    const int yo = (int)(tkn % Ho);
    const int b = (int)(tkn / Ho);
    const f32x4* s = reinterpret_cast<const f32x4*>(pb.src[blockIdx.y]);
    const size_t r0 = (((size_t)b * H + 2 * yo) * W + 2 * xo) * C4 + c4;
    const f32x4 a = s[r0], bb = s[r0 + C4], c = s[r0 + (size_t)W * C4], d = s[r0 + (size_t)W * C4 + C4];
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (((a[i] + bb[i]) + c[i]) + d[i]) * 0.25f;
    reinterpret_cast<f32x4*>(pb.dst[blockIdx.y])[g] = o;
}

extern "C" int casmtr_token_pool_fwd(const float* const* src, float* const* dst, int n, int B, int H, int W, int C,
                                     casmtr_stream_t stream) {
    if (n <= 0 || B <= 0) return 0;
    if (n > POOL_MAXT || C % 4 != 0 || H < 2 || W < 2) return CASMTR_ERR_UNSUPPORTED;
    PoolBatch pb{};
    for (int i = 0; i < n; ++i) { pb.src[i] = src[i]; pb.dst[i] = dst[i]; }
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    ProfScope ps(CASMTR_PROF_TOKEN_POOL, (hipStream_t)stream, "token_pool_kernel");
    hipLaunchKernelGGL(token_pool_kernel, dim3((unsigned)((total + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, pb,
                       H, W, C / 4, total);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
