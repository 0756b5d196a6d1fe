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
#include "linear_pc.hpp"

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
    // quad-major output (casmtr_linear_quads_fwd): rows are the tokens of qh x qw grids, y = [M/(qh*qw)][N/32][(qh/2)*(qw/2)][4][32]
    // (the layout of quad_layout.hip); qw == 0: plain [M][N].  Division of the in-grid position by qw via m = ceil(2^32 / qw),
    // n / qw = umulhi(n, m), exact for n * (m * qw - 2^32) < 2^32 (checked on the host)
    int qh, qw;
    unsigned magic_w;
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
    if (lb.qw) {
        // quad-major: a 32-column block is one head's 32 dims = the 128 contiguous bytes of a (quad, child) row; only the row offset
        // differs from the token-major store
        const int hw = lb.qh * lb.qw, wq = lb.qw >> 1, Lq = (lb.qh >> 1) * wq, Hh = N >> 5;
        const unsigned b0 = (unsigned)i0 / (unsigned)hw, rem0 = (unsigned)i0 - b0 * (unsigned)hw;   // wave-uniform
        float bj[2];
        float* yh[2];
        bool okj[2];
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int gj = j0 + wc * 64 + tj * 32 + ln;
            okj[tj] = gj < N;
            bj[tj] = bias && okj[tj] ? bias[gj] : 0.f;
            yh[tj] = Y + (size_t)(gj >> 5) * Lq * 128 + ln;
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned gi = (unsigned)(i0 + wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                if ((int)gi >= M) continue;
                unsigned b = b0, pp = rem0 + (gi - (unsigned)i0);   // the tile's 128 rows start in pair b0 and rarely leave it
                while (pp >= (unsigned)hw) { pp -= (unsigned)hw; ++b; }
                const unsigned y = __umulhi(pp, lb.magic_w), x = pp - y * (unsigned)lb.qw;
                const size_t roff = ((size_t)b * Hh * Lq + (size_t)((y >> 1) * wq + (x >> 1))) * 128 + ((y & 1) * 2 + (x & 1)) * 32;
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    if (okj[tj]) yh[tj][roff] = bias ? acc[ti][tj][r] + bj[tj] : acc[ti][tj][r];
            }
        return;
    }
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

static bool lin_magic(unsigned d, unsigned nmax, unsigned* m) {   // n / d == umulhi(n, m) for every n <= nmax ?
    if (d == 0) return false;
    if (d < 2) return false;
    const unsigned long long mm = (0x100000000ull + d - 1) / d;
    const unsigned long long e = mm * d - 0x100000000ull;
    if ((unsigned long long)nmax * e >= 0x100000000ull) return false;
    *m = (unsigned)mm;
    return true;
}

static int linear_launch(const float* const* x, const float* const* w, const float* const* bias, float* const* y, int nprob, int M, int N,
                         int K, int qh, int qw, casmtr_stream_t stream) {
    if (nprob <= 0 || M <= 0 || N <= 0) return 0;
    if (nprob > LIN_MAXP || K <= 0 || K % LIN_BK != 0) return CASMTR_ERR_UNSUPPORTED;
    LinearBatch lb{};
    if (qw) {
        if (qh <= 0 || qw <= 1 || (qh & 1) || (qw & 1) || (N & 31) || M % (qh * qw) != 0) return CASMTR_ERR_UNSUPPORTED;
        if (!lin_magic((unsigned)qw, (unsigned)(qh * qw), &lb.magic_w)) return CASMTR_ERR_UNSUPPORTED;
        lb.qh = qh; lb.qw = qw;
    }
    for (int i = 0; i < nprob; ++i) {
        lb.x[i] = x[i]; lb.w[i] = w[i]; lb.bias[i] = bias ? bias[i] : nullptr; lb.y[i] = y[i];
    }
    const int NIB = (M + LIN_BM - 1) / LIN_BM, NJB = (N + LIN_BN - 1) / LIN_BN;
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LINEAR, linear_nt_kernel, dim3(NIB * NJB, nprob), dim3(256), 0, (hipStream_t)stream, lb, M, N, K, NJB);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" int casmtr_linear_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                                 int nprob, int M, int N, int K, casmtr_stream_t stream) {
    return linear_launch(x, w, bias, y, nprob, M, N, K, 0, 0, stream);
}

// The same projections with the result written quad-major per head, [B][N/32][(h/2)*(w/2)][4][32] (M = B*h*w token rows of h x w
// grids): what the fine-level / cascade kernels read (fine_quad.hip, cascade_quad.hip), so that the token -> quad layout pass
// (casmtr_tokens_to_quads) between the projection and the attention disappears.  Same arithmetic, same values.
extern "C" int casmtr_linear_quads_fwd(const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                                       int nprob, int M, int N, int K, int h, int w_, casmtr_stream_t stream) {
    if (h <= 0 || w_ <= 0) return CASMTR_ERR_UNSUPPORTED;
    return linear_launch(x, w, bias, y, nprob, M, N, K, h, w_, stream);
}

// =================================================================================================== projections on the f16 matrix pipe
// The same projections, fp32-accurate, on v_mfma_f32_32x32x16_f16: the two-term f16 split of ds_split.hip applied to
// y = x . w^T.  fp32 MFMA runs at the fp32 vector rate (1/16 of the f16 MFMA rate) and linear_nt_kernel sits at 63 % of it; the
// reference's own conv / linear is a BLAS call with unspecified accumulation order (TF32 under torch 1.10), so the contract of this path
// is a tolerance, not the chain: every operand row is scaled by a power of two so that its largest element lies in [512, 1024) and
// split a = hi + lo + r, hi = f16(a), lo = f16(a - hi), |r| <= 2^-22 |a|; f16 x f16 is exact in fp32, so
//      x.w  ~  lo_x.hi_w + hi_x.lo_w + hi_x.hi_w        (dropped: lo.lo and the r terms, <= 3 x 2^-22 sum|x w|),
// accumulated in fp32 over K / 16 MFMA steps: |y_split - y_chain| <= 2^-15 |x_m| |w_n| (the budget of ds_split.hip, K <= 256; measured
// ~1e-7 |x||w|, tests/test_gpu_callers.py).  Default stays the exact chain (casmtr_linear_fwd); callers opt in (ops.linear_multi(gemm=)).
//   * weights: split once into the GEMM's tile image (casmtr_linear_split_prep; cached by the caller per weight tensor):
//       img[jb = n / 128][ks = k / 32][kg = (k / 8) % 4][part hi | lo][row n % 128][8 f16]  (16 KB per (jb, ks)), fac[n] = 2^e_n
//   * activations: the GEMM reads the fp32 rows itself, finds every row's exponent in a first pass over its 128 x K tile (the second pass,
//     the k-loop, re-reads the tile from the L2 it has just been pulled into) and splits the rows on the way into LDS: no activation
//     image and no exponent array in HBM -- the kernel is bound by reading x and writing y (64 flop per byte at K = N = 256).  (A
//     separate exponent pass, one wave per row, cost 1.2 ms per step next to 3.9 ms of GEMMs: one more read of every activation.)
//   * linear16_kernel: linear_nt_kernel's structure (128 x 128 tile, 4 waves x 64 x 64, next k-chunk prefetched into registers under the
//     MFMAs of the current one), 12 MFMAs per wave and 16 channels, small terms first; epilogue acc * 2^e_m * 2^e_n + bias, token-major or
//     quad-major rows exactly as linear_nt_kernel.
typedef _Float16 l16_h8 __attribute__((ext_vector_type(8)));

struct Linear16Batch {
    const float* x[LIN_MAXP];
    const char* wimg[LIN_MAXP];     // weight tile image
    const float* wfac[LIN_MAXP];    // [N] 2^e_n
    const float* bias[LIN_MAXP];    // nullable
    float* y[LIN_MAXP];
    int qh, qw;
    unsigned magic_w;
};

// workgroup per weight row: exponent, factor, split into the tile image (N x K <= 64 K elements: latency-bound, run once per weight)
__global__ __launch_bounds__(64) void lin_wprep_kernel(const float* __restrict__ w, int N, int K, char* __restrict__ img, float* __restrict__ fac) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* p = w + (size_t)n * K;
    float mx = 0.f;
    for (int c = lane; c < K; c += 64) mx = fmaxf(mx, fabsf(p[c]));
    mx = wave_max_f32(mx);
    const int e = (mx > 0.f && mx < INFINITY) ? ilogbf(mx) - 9 : 0;
    if (lane == 0) fac[n] = ldexpf(1.0f, e);
    for (int g8 = lane; g8 < K / 8; g8 += 64) {
        l16_h8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xn = ldexpf(p[8 * g8 + c], -e);
            const _Float16 h = (_Float16)xn;
            hi[c] = h;
            lo[c] = (_Float16)(xn - (float)h);
        }
        char* o = img + ((size_t)(n >> 7) * (K / 32) + (g8 >> 2)) * 16384 + (g8 & 3) * 4096 + (n & 127) * 16;
        *reinterpret_cast<l16_h8*>(o) = hi;
        *reinterpret_cast<l16_h8*>(o + 2048) = lo;
    }
}

__global__ __launch_bounds__(256, 3) void linear16_kernel(const Linear16Batch lb, int M, int N, int K, int NJB) {
    __shared__ __attribute__((aligned(16))) char As[16384];   // one k-chunk of 32 channels: [kg 4][hi | lo][128 rows][8 f16]
    __shared__ __attribute__((aligned(16))) char Bs[16384];
    __shared__ float facA[LIN_BM], facB[LIN_BN];
    const int t = xcd_chunk_remap(blockIdx.x, gridDim.x);
    const int tI = t / NJB, tJ = t - tI * NJB;
    const int p = blockIdx.y, i0 = tI * LIN_BM, j0 = tJ * LIN_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int KS = K >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // activations: thread <-> (row tid / 2, 16 channels of the chunk): two kg planes of the image
    const int lrow = tid >> 1, half = tid & 1;
    const int gi = i0 + lrow < M ? i0 + lrow : M - 1;
    const float* ap = lb.x[p] + (size_t)gi * K + half * 16;
    // first pass: the row's largest |element| -> the exponent that puts it into [512, 1024) (ds_rownorm_kernel's rule); the two threads
    // of a row hold one half of every 32-channel chunk each
    float mx = 0.f;
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ap + ks * 32 + 4 * i);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    mx = fmaxf(mx, dpp_f32<0xB1>(mx));   // lane ^ 1
    const int e = (mx > 0.f && mx < INFINITY) ? ilogbf(mx) - 9 : 0;
    const float sc = ldexpf(1.0f, -e);
    if (half == 0) facA[lrow] = ldexpf(1.0f, e);
    if (tid < LIN_BN) facB[tid] = lb.wfac[p][j0 + tid];
    const char* bp = lb.wimg[p] + (size_t)tJ * KS * 16384 + tid * 16;
    char* const a_dst = As + (half * 2) * 4096 + lrow * 16;
    f32x4 av[4];
    u32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
        bv[i] = *reinterpret_cast<const u32x4*>(bp + 4096 * i);
    }
    const int hi = lane >> 5, ln = lane & 31;
    const char* fa_base = As + hi * 4096 + (wr * 64 + ln) * 16;   // k16 sub-stage s: + s * 8192; tile ti: + ti * 512; lo part: + 2048
    const char* fb_base = Bs + hi * 4096 + (wc * 64 + ln) * 16;
    for (int ks = 0; ks < KS; ++ks) {
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int kgl = 0; kgl < 2; ++kgl) {
            l16_h8 vh, vl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float xn = av[2 * kgl + (c >> 2)][c & 3] * sc;   // exact: a power of two
                const _Float16 h = (_Float16)xn;
                vh[c] = h;
                vl[c] = (_Float16)(xn - (float)h);
            }
            *reinterpret_cast<l16_h8*>(a_dst + kgl * 4096) = vh;
            *reinterpret_cast<l16_h8*>(a_dst + kgl * 4096 + 2048) = vl;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(Bs + 4096 * i + tid * 16) = bv[i];
        if (ks + 1 < KS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = *reinterpret_cast<const f32x4*>(ap + (ks + 1) * 32 + 4 * i);
                bv[i] = *reinterpret_cast<const u32x4*>(bp + (size_t)(ks + 1) * 16384 + 4096 * i);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            l16_h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const char* pa = fa_base + s * 8192 + ti * 512;
                const char* pb = fb_base + s * 8192 + ti * 512;
                ah[ti] = *reinterpret_cast<const l16_h8*>(pa);
                al[ti] = *reinterpret_cast<const l16_h8*>(pa + 2048);
                bh[ti] = *reinterpret_cast<const l16_h8*>(pb);
                bl[ti] = *reinterpret_cast<const l16_h8*>(pb + 2048);
            }
            // small terms first; every accumulator is touched again only after three other MFMAs (ds_gemm16_kernel's order)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ti], bh[tj], acc[ti][tj], 0, 0, 0);
        }
    }
    // epilogue: y = acc * 2^e_m * 2^e_n (+ bias); lane (hi, ln) holds rows {(r&3) + 8(r>>2) + 4hi}, column ln of each 32x32 block
    const float* __restrict__ bias = lb.bias[p];
    float* __restrict__ Y = lb.y[p];
    float bj[2], fbv[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int cj = wc * 64 + tj * 32 + ln;
        fbv[tj] = facB[cj];
        bj[tj] = bias ? bias[j0 + cj] : 0.f;
    }
    if (lb.qw) {   // quad-major rows (see linear_nt_kernel)
        const int hw = lb.qh * lb.qw, wq = lb.qw >> 1, Lq = (lb.qh >> 1) * wq, Hh = N >> 5;
        const unsigned b0 = (unsigned)i0 / (unsigned)hw, rem0 = (unsigned)i0 - b0 * (unsigned)hw;   // wave-uniform
        float* yh[2];
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) yh[tj] = Y + (size_t)((j0 + wc * 64 + tj * 32) >> 5) * Lq * 128 + ln;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const unsigned gr = (unsigned)(i0 + lr);
                if ((int)gr >= M) continue;
                unsigned b = b0, pp = rem0 + (unsigned)lr;
                while (pp >= (unsigned)hw) { pp -= (unsigned)hw; ++b; }
                const unsigned y = __umulhi(pp, lb.magic_w), x = pp - y * (unsigned)lb.qw;
                const size_t roff = ((size_t)b * Hh * Lq + (size_t)((y >> 1) * wq + (x >> 1))) * 128 + ((y & 1) * 2 + (x & 1)) * 32;
                const float fa = facA[lr];
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const float v = (acc[ti][tj][r] * fa) * fbv[tj];
                    yh[tj][roff] = bias ? v + bj[tj] : v;
                }
            }
        return;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int gr = i0 + lr;
            if (gr >= M) continue;
            const float fa = facA[lr];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const float v = (acc[ti][tj][r] * fa) * fbv[tj];
                Y[(size_t)gr * N + j0 + wc * 64 + tj * 32 + ln] = bias ? v + bj[tj] : v;
            }
        }
}

// The same GEMM with the ACTIVATION TILE STATIONARY (round 6, second version).  linear16_kernel reads and splits a 128 x K activation
// tile once per 128 output columns and per problem: at K = N = 256 the q / k / v projections of a self-attention layer pull the same
// rows through the L2 twelve times and convert them six times, and every tile pays its own first pass + eight-chunk latency chain
// (69 us per 86 528 x 256 x 256 problem against 39 us for its bytes).  Here a workgroup owns 64 rows of ONE activation tensor: it reads
// them once (registers), finds the exponents, splits them into LDS for the whole K (64 KB at K = 256) and then walks every job that
// uses these rows -- every problem of the launch with this x pointer, every 128-column tile of its weights -- with the prepared weight
// fragments read from the L2 straight into registers.  x is read from HBM exactly once per launch; nothing else is.
struct Lin16sBatch {
    const float* x[LIN_MAXP];           // distinct activation tensors (groups)
    int first[LIN_MAXP], count[LIN_MAXP];   // group g: problems prob[first[g]] .. + count[g]
    const char* wimg[LIN_MAXP];         // per problem, in group order
    const float* wfac[LIN_MAXP];
    const float* bias[LIN_MAXP];
    float* y[LIN_MAXP];
    int qh, qw;
    unsigned magic_w;
    int xflags;                         // experiment switches (CASMTR_LIN_FLAGS, tools/lin_time.py): 1 no stores, 2 no MFMAs, 4 rows not loaded
};

template <int K>
__global__ __launch_bounds__(256, 2) void linear16s_kernel(const Lin16sBatch lb, int M, int N) {
    constexpr int KS = K / 32, RB = 64;
    constexpr int NV = K / 16;   // float4 loads per thread (4 threads per row)
    extern __shared__ __attribute__((aligned(16))) char lds16s[];
    char* As = lds16s;                       // [ks][kg 4][hi | lo][64 rows][8 f16]: 8 KB per 32-channel chunk
    float* facA = reinterpret_cast<float*>(As + KS * 8192);   // [64]
    const int g = blockIdx.y, i0 = blockIdx.x * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int hi = lane >> 5, ln = lane & 31;
    // ---- phase 1: the 64 rows, once.  Thread <-> (row tid / 4, quarter tid % 4 of the channels)
    {
        const int row = tid >> 2, qt = tid & 3;
        const int gi = i0 + row < M ? i0 + row : M - 1;
        const float* ap = lb.x[g] + (size_t)gi * K + qt * (K / 4);
        f32x4 v[NV];
        if (lb.xflags & 4) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = f32x4{(float)tid, 1.f, 2.f, (float)i};
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
        }
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        mx = fmaxf(mx, dpp_f32<0xB1>(mx));   // lane ^ 1
        mx = fmaxf(mx, dpp_f32<0x4E>(mx));   // lane ^ 2
        const int e = (mx > 0.f && mx < INFINITY) ? ilogbf(mx) - 9 : 0;   // ds_rownorm_kernel's rule: largest |element| -> [512, 1024)
        const float sc = ldexpf(1.0f, -e);
        if (qt == 0) facA[row] = ldexpf(1.0f, e);
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) {   // groups of 8 channels: g8 = qt * NV / 2 + j -> chunk g8 / 4, plane kg = g8 % 4
            const int g8 = qt * (NV / 2) + j;
            l16_h8 vh, vl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float xn = v[2 * j + (c >> 2)][c & 3] * sc;   // exact: a power of two
                const _Float16 h = (_Float16)xn;
                vh[c] = h;
                vl[c] = (_Float16)(xn - (float)h);
            }
            char* d = As + (g8 >> 2) * 8192 + (g8 & 3) * 2048 + row * 16;
            *reinterpret_cast<l16_h8*>(d) = vh;
            *reinterpret_cast<l16_h8*>(d + 1024) = vl;
        }
    }
    // ---- phase 2: every job on these rows.  4 waves = 2 (32 rows) x 2 (64 columns) of the 64 x 128 output tile
    const int NJB = N / LIN_BN;
    const char* fa_base = As + hi * 2048 + (wr * 32 + ln) * 16;     // chunk ks: + ks * 8192; k16 stage s: + s * 4096; lo part: + 1024
    // The weight fragments go from the prepared image (L2-resident: N K 4 bytes per problem) STRAIGHT into registers -- a lane's 16 bytes
    // of plane (kg = lane / 32, part) of column wc * 64 + tj * 32 + lane % 32 are what the MFMA wants, and a half-wave reads 512 contiguous
    // bytes: no LDS buffer for B (64 KB + 16 KB + the factors would be 768 bytes over two workgroups per CU) and no barrier in the k-loop.
    const unsigned b_lane = (unsigned)(hi * 4096 + (wc * 64 + ln) * 16);
    __syncthreads();   // phase 1 complete
    for (int pi = 0; pi < lb.count[g]; ++pi) {
        const int p = lb.first[g] + pi;
        const float* __restrict__ bias = lb.bias[p];
        float* __restrict__ Y = lb.y[p];
        for (int tJ = 0; tJ < NJB; ++tJ) {
            const int j0 = tJ * LIN_BN;
            const char* bq = lb.wimg[p] + (size_t)tJ * KS * 16384 + b_lane;   // stage (ks, st): + ks * 16384 + st * 8192; tile tj: + tj * 512; lo: + 2048
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            // k16 stages s = 2 ks + st: stage s + 1's fragments are loaded while stage s multiplies (two register sets; the loop is
            // unrolled by two so that the set index is a constant -- fully unrolled, the compiler hoists all 16 stages' loads: 255 VGPRs)
            l16_h8 bh[2][2], bl[2][2];
            auto loadb = [&](int st, l16_h8 (&h)[2], l16_h8 (&l)[2]) {
                const char* pb = bq + st * 8192;
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    h[tj] = *reinterpret_cast<const l16_h8*>(pb + tj * 512);
                    l[tj] = *reinterpret_cast<const l16_h8*>(pb + tj * 512 + 2048);
                }
            };
            auto stage = [&](int st, const l16_h8 (&h)[2], const l16_h8 (&l)[2]) {
                const char* pa = fa_base + st * 4096;
                const l16_h8 ah = *reinterpret_cast<const l16_h8*>(pa), al = *reinterpret_cast<const l16_h8*>(pa + 1024);
                // small terms first (linear16_kernel's / ds_gemm16_kernel's order)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, h[tj], acc[tj], 0, 0, 0);
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, l[tj], acc[tj], 0, 0, 0);
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, h[tj], acc[tj], 0, 0, 0);
            };
            loadb(0, bh[0], bl[0]);
#pragma unroll 1
            for (int st = 0; st < ((lb.xflags & 2) ? 2 : 2 * KS); st += 2) {
                loadb(st + 1, bh[1], bl[1]);
                stage(st, bh[0], bl[0]);
                if (st + 2 < 2 * KS) loadb(st + 2, bh[0], bl[0]);
                stage(st + 1, bh[1], bl[1]);
            }
            // epilogue of the job: y = acc * 2^e_m * 2^e_n (+ bias)
            float bj[2], fbv[2];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int cj = wc * 64 + tj * 32 + ln;
                fbv[tj] = lb.wfac[p][j0 + cj];
                bj[tj] = bias ? bias[j0 + cj] : 0.f;
            }
            if ((lb.xflags & 1) && acc[0][0] != 12345.678f) continue;
            if (lb.qw) {   // quad-major rows (see linear_nt_kernel)
                const int hw = lb.qh * lb.qw, wq = lb.qw >> 1, Lq = (lb.qh >> 1) * wq, Hh = N >> 5;
                const unsigned b0 = (unsigned)i0 / (unsigned)hw, rem0 = (unsigned)i0 - b0 * (unsigned)hw;   // wave-uniform
                float* yh[2];
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) yh[tj] = Y + (size_t)((j0 + wc * 64 + tj * 32) >> 5) * Lq * 128 + ln;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (i0 + lr >= M) continue;
                    unsigned b = b0, pp = rem0 + (unsigned)lr;
                    while (pp >= (unsigned)hw) { pp -= (unsigned)hw; ++b; }
                    const unsigned yy = __umulhi(pp, lb.magic_w), xx = pp - yy * (unsigned)lb.qw;
                    const size_t roff = ((size_t)b * Hh * Lq + (size_t)((yy >> 1) * wq + (xx >> 1))) * 128 + ((yy & 1) * 2 + (xx & 1)) * 32;
                    const float fa = facA[lr];
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) {
                        const float v = (acc[tj][r] * fa) * fbv[tj];
                        yh[tj][roff] = bias ? v + bj[tj] : v;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int gr = i0 + lr;
                    if (gr >= M) continue;
                    const float fa = facA[lr];
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) {
                        const float v = (acc[tj][r] * fa) * fbv[tj];
                        Y[(size_t)gr * N + j0 + wc * 64 + tj * 32 + ln] = bias ? v + bj[tj] : v;
                    }
                }
            }
        }
    }
}

template <int K>
static int launch_linear16s(const Lin16sBatch& lb, int ngroups, int M, int N, hipStream_t s) {
    constexpr size_t lds = (K / 32) * 8192 + 64 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear16s_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    prof_symbol_args(CASMTR_PROF_LINEAR, "<%d>", K);
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LINEAR, linear16s_kernel<K>, dim3((unsigned)((M + 63) / 64), (unsigned)ngroups), dim3(256), lds, s, lb, M, N);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t casmtr_linear_split_prep_bytes(int N, int K) {
    return (N > 0 && K > 0 && N % LIN_BN == 0 && K % LIN_BK == 0) ? (size_t)N * K * 4 + (size_t)N * 4 : 0;   // image (2 f16 per element), then fac[N]
}

extern "C" int casmtr_linear_split_prep(const float* w, void* prep, int N, int K, casmtr_stream_t stream) {
    if (N <= 0 || K <= 0 || N % LIN_BN != 0 || K % LIN_BK != 0 || K > 256) return CASMTR_ERR_UNSUPPORTED;
    char* img = reinterpret_cast<char*>(prep);
    hipLaunchKernelGGL(lin_wprep_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, w, N, K, img, reinterpret_cast<float*>(img + (size_t)N * K * 4));
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// the persistent producer / consumer kernel (linear_pc.hip): problems grouped by activation tensor; y1 / y2: the pooled levels (quad mode)
static int linear_pc_dispatch(const float* const* x, const void* const* wprep, const float* const* bias, float* const* y,
                              float* const* y1, float* const* y2, int y1_tokens, int nprob, int M, int N, int K, int B, int h, int w_,
                              hipStream_t s) {
    Lin16pArgs a{};
    a.M = M; a.N = N; a.h = h; a.w = w_; a.y1_tokens = y1_tokens;
    if (w_) {
        a.nbx = (w_ + 7) / 8; a.nby = (h + 7) / 8;
        a.nsub = B * a.nbx * a.nby;
    } else a.nsub = (M + 63) / 64;
    int ng = 0, np = 0;
    bool used[L16P_MAXP] = {};
    for (int i = 0; i < nprob; ++i) {
        if (used[i]) continue;
        a.x[ng] = x[i]; a.first[ng] = np;
        for (int j = i; j < nprob; ++j) {
            if (used[j] || x[j] != x[i]) continue;
            used[j] = true;
            a.wimg[np] = reinterpret_cast<const char*>(wprep[j]);
            a.wfac[np] = reinterpret_cast<const float*>(a.wimg[np] + (size_t)N * K * 4);
            a.bias[np] = bias ? bias[j] : nullptr;
            a.y0[np] = y[j];
            a.y1[np] = y1 ? y1[j] : nullptr;
            a.y2[np] = y2 ? y2[j] : nullptr;
            ++np;
        }
        a.count[ng] = np - a.first[ng];
        ++ng;
    }
    a.ngroups = ng; a.nprob = np;
    const char* xf = getenv("CASMTR_LIN_FLAGS");
    a.xflags = xf ? atoi(xf) : 0;
    return linear16p_launch(a, K, s);
}

extern "C" int casmtr_linear_split_fwd(const float* const* x, const void* const* wprep, const float* const* bias, float* const* y,
                                       int nprob, int M, int N, int K, int h, int w_, casmtr_stream_t stream) {
    if (nprob <= 0 || M <= 0 || N <= 0) return 0;
    if (nprob > L16P_MAXP || K <= 0 || K % LIN_BK != 0 || K > 256 || N % LIN_BN != 0) return CASMTR_ERR_UNSUPPORTED;
    Linear16Batch lb{};
    if (w_) {
        if (h <= 0 || w_ <= 1 || (h & 1) || (w_ & 1) || M % (h * w_) != 0) return CASMTR_ERR_UNSUPPORTED;
        if (!lin_magic((unsigned)w_, (unsigned)(h * w_), &lb.magic_w)) return CASMTR_ERR_UNSUPPORTED;
        lb.qh = h; lb.qw = w_;
    }
    hipStream_t s = (hipStream_t)stream;
    const char* ev = getenv("CASMTR_LINEAR16");   // "tile": one workgroup per 128 x 128 output tile; "stationary": linear16s_kernel; for A/B
    if ((K == 128 || (K == 256 && N % 256 == 0)) && nprob * N <= L16P_MAXFAC && !ev)   // (else: the stationary kernel below)
        return linear_pc_dispatch(x, wprep, bias, y, nullptr, nullptr, 0, nprob, M, N, K, w_ ? M / (h * w_) : 0, h, w_, s);
    if (nprob > LIN_MAXP) return CASMTR_ERR_UNSUPPORTED;   // the earlier kernels take four problems
    if ((K == 256 || K == 128) && !(ev && ev[0] == 't')) {   // activation-stationary kernel: problems grouped by activation tensor
        Lin16sBatch sb{};
        sb.qh = lb.qh; sb.qw = lb.qw; sb.magic_w = lb.magic_w;
        const char* xf = getenv("CASMTR_LIN_FLAGS");
        sb.xflags = xf ? atoi(xf) : 0;
        int ng = 0, np = 0;
        bool used[LIN_MAXP] = {false, false, false, false};
        for (int i = 0; i < nprob; ++i) {
            if (used[i]) continue;
            sb.x[ng] = x[i]; sb.first[ng] = np;
            for (int j = i; j < nprob; ++j) {
                if (used[j] || x[j] != x[i]) continue;
                used[j] = true;
                sb.wimg[np] = reinterpret_cast<const char*>(wprep[j]);
                sb.wfac[np] = reinterpret_cast<const float*>(sb.wimg[np] + (size_t)N * K * 4);
                sb.bias[np] = bias ? bias[j] : nullptr;
                sb.y[np] = y[j];
                ++np;
            }
            sb.count[ng] = np - sb.first[ng];
            ++ng;
        }
        return K == 256 ? launch_linear16s<256>(sb, ng, M, N, s) : launch_linear16s<128>(sb, ng, M, N, s);
    }
    for (int i = 0; i < nprob; ++i) {
        lb.x[i] = x[i];
        lb.wimg[i] = reinterpret_cast<const char*>(wprep[i]);
        lb.wfac[i] = reinterpret_cast<const float*>(lb.wimg[i] + (size_t)N * K * 4);
        lb.bias[i] = bias ? bias[i] : nullptr;
        lb.y[i] = y[i];
    }
    const int NIB = (M + LIN_BM - 1) / LIN_BM, NJB = N / LIN_BN;
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LINEAR, linear16_kernel, dim3(NIB * NJB, nprob), dim3(256), 0, s, lb, M, N, K, NJB);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// q / k / v projections WITH their pyramid (QuadtreeAttention.forward, src/model/modules/quadtree_attention.py:78-88): y0 quad-major as
// casmtr_linear_split_fwd(h, w) writes it, y1 = avg_pool2d(y0, 2, 2) quad-major (or token-major with y1_tokens: the coarsest level of a
// two-level pyramid), y2 = avg_pool2d(y1, 2, 2) token-major (what the coarsest level's kernel reads) -- the three tensors casmtr_quad_pool_fwd
// produces from y0 in two more launches, bit for bit, without reading y0 or y1 back.  y1 / y2 nullable (arrays or entries).
extern "C" int casmtr_linear_split_pyramid_fwd(const float* const* x, const void* const* wprep, const float* const* bias, float* const* y0,
                                               float* const* y1, float* const* y2, int y1_tokens, int nprob, int B, int h, int w_, int N,
                                               int K, casmtr_stream_t stream) {
    if (nprob <= 0 || B <= 0 || N <= 0) return 0;
    if (nprob > L16P_MAXP || (K != 128 && K != 256) || N % LIN_BN != 0 || h < 2 || w_ < 2 || (h & 1) || (w_ & 1)) return CASMTR_ERR_UNSUPPORTED;
    if ((K == 256 && N % 256 != 0) || nprob * N > L16P_MAXFAC) return CASMTR_ERR_UNSUPPORTED;
    if (y2 && (!y1 || y1_tokens || (h & 3) || (w_ & 3))) return CASMTR_ERR_UNSUPPORTED;
    if (y1 && !y1_tokens && ((h & 3) || (w_ & 3))) return CASMTR_ERR_UNSUPPORTED;
    if ((long long)B * h * w_ > 0x7fffffffLL) return CASMTR_ERR_UNSUPPORTED;
    return linear_pc_dispatch(x, wprep, bias, y0, y1, y2, y1_tokens, nprob, B * h * w_, N, K, B, h, w_, (hipStream_t)stream);
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


// =================================================================================================== pyramid on quad-major tensors
// F.avg_pool2d(kernel 2, stride 2) of a quad-major level: the 2x2 window of pooled token (Y, X) IS quad (Y, X)'s four children, in the
// (row, col) order of torch's accumulation: pooled = (((c0 + c1) + c2) + c3) * 0.25, one 128-byte row per head.  The pooled level is
// written quad-major again ([B][H][(h/4)*(w/4)][4][32], for the next finer level's kernels) or token-major ([B][(h/2)*(w/2)][C], what
// the coarsest level's kernel reads).  Lane <-> (quad l / 8, 16-byte unit l % 8): four 128-byte pieces per load instruction and quad,
// 128-byte rows out.
__global__ __launch_bounds__(256) void quad_pool_kernel(const PoolBatch pb, int Hh, int hq, int wq, int to_tokens, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;   // (b, head, quad, unit)
    if (g >= total) return;
    const int u = (int)(g & 7);
    long long t = g >> 3;
    const int Lq = hq * wq;
    const int quad = (int)(t % Lq); t /= Lq;
    const int hd = (int)(t % Hh);
    const int b = (int)(t / Hh);
    const f32x4* s = reinterpret_cast<const f32x4*>(pb.src[blockIdx.y]) + ((size_t)(b * Hh + hd) * Lq + quad) * 32 + u;
    const f32x4 a = s[0], bb = s[8], c = s[16], d = s[24];
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (((a[i] + bb[i]) + c[i]) + d[i]) * 0.25f;
    const int Y = quad / wq, X = quad - Y * wq;
    size_t off;   // in 16-byte units
    if (to_tokens) off = ((size_t)b * Lq + quad) * (Hh * 8) + hd * 8 + u;
    else off = (((size_t)(b * Hh + hd) * ((hq >> 1) * (wq >> 1)) + (Y >> 1) * (wq >> 1) + (X >> 1)) * 4 + (Y & 1) * 2 + (X & 1)) * 8 + u;
    reinterpret_cast<f32x4*>(pb.dst[blockIdx.y])[off] = o;
}

// src: n quad-major tensors of B x (h x w tokens) x C; dst: the pooled (h/2 x w/2) levels, quad-major or (to_tokens) token-major
extern "C" int casmtr_quad_pool_fwd(const float* const* src, float* const* dst, int n, int B, int h, int w, int C, int to_tokens,
                                    casmtr_stream_t stream) {
    if (n <= 0 || B <= 0) return 0;
    if (n > POOL_MAXT || (C & 31) || h < 2 || w < 2 || (h & 1) || (w & 1) || (!to_tokens && ((h & 3) || (w & 3)))) return CASMTR_ERR_UNSUPPORTED;
    PoolBatch pb{};
    for (int i = 0; i < n; ++i) { pb.src[i] = src[i]; pb.dst[i] = dst[i]; }
    const long long total = (long long)B * (C / 32) * (h / 2) * (w / 2) * 8;
    ProfScope ps(CASMTR_PROF_TOKEN_POOL, (hipStream_t)stream, "quad_pool_kernel");
    hipLaunchKernelGGL(quad_pool_kernel, dim3((unsigned)((total + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, pb, C / 32, h / 2, w / 2,
                       to_tokens, total);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
