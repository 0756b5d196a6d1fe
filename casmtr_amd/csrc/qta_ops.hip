// Drop-in primitives: the three ops the reference ships as CUDA extensions, as standalone gfx950 kernels.
//   a1  casmtr_qta_score_fwd/bwd      <- cuda_imp/QuadTreeAttention/QuadtreeAttention/src/score_computation_kernal.cu:21-184
//   a2  casmtr_qta_value_agg_fwd/bwd  <- .../src/value_aggregation_kernel.cu:21-86
//   a8  casmtr_window_score_fwd/bwd   <- cuda_imp/score_cuda/src/score_computation_kernel.cu:22-123
// These keep the reference's op granularity (the model calls them through torch.autograd.Function); the fused
// per-level kernels in qta_fused.hip / matching.hip are what the hot path actually runs.
// Arithmetic: fp32 fmaf chain over the contracted axis, ascending, acc0 = 0 (bit-identical to the oracle).
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

// ---------------------------------------------------------------------------------------------------- a1 forward
// One workgroup per quad (b, n1).  The 4 children's queries (4*H*D floats) sit in LDS; every lane owns one
// (candidate k, head h) key row, loads it once (float4 x D/4) and produces the 4 children's scores from it, so a key
// row is fetched once per quad instead of once per child (reference: one 32-thread block per (b, n1, f)).
// Stores are coalesced: out[b,n,f,k,h] with (k,h) == the lane's item index.
template <int D>
__global__ __launch_bounds__(256) void qta_score_fwd_kernel(const float* __restrict__ q, const float* __restrict__ key,
                                                            const int64_t* __restrict__ idx, float* __restrict__ out,
                                                            int N1, int N2, int K, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][H][D+4]
    const int b = blockIdx.y, n = blockIdx.x;
    const int HD = H * D, DP = D + 4;
    const float* qb = q + ((size_t)b * N1 + n) * 4 * HD;
    for (int e = threadIdx.x; e < 4 * HD; e += blockDim.x) smem[(e / D) * DP + (e % D)] = qb[e];
    __syncthreads();
    const int R = K * H;
    const int64_t* ib = idx + ((size_t)b * N1 + n) * R;
    float* ob = out + ((size_t)b * N1 + n) * 4 * R;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int h = r % H;
        const int j = (int)ib[r];
        const f32x4* kp = reinterpret_cast<const f32x4*>(key + (((size_t)b * N2 + j) * H + h) * D);
        f32x4 kr[D / 4];
#pragma unroll
        for (int i = 0; i < D / 4; ++i) kr[i] = kp[i];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4* qp = reinterpret_cast<const f32x4*>(smem + (f * H + h) * DP);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                f32x4 qv = qp[i];
                acc = __builtin_fmaf(qv.x, kr[i].x, acc);
                acc = __builtin_fmaf(qv.y, kr[i].y, acc);
                acc = __builtin_fmaf(qv.z, kr[i].z, acc);
                acc = __builtin_fmaf(qv.w, kr[i].w, acc);
            }
            ob[(size_t)f * R + r] = acc;
        }
    }
}

// generic-D fallback (scalar loads), same arithmetic
__global__ __launch_bounds__(256) void qta_score_fwd_generic(const float* __restrict__ q, const float* __restrict__ key,
                                                             const int64_t* __restrict__ idx, float* __restrict__ out,
                                                             int N1, int N2, int K, int H, int D) {
    const int b = blockIdx.y, n = blockIdx.x;
    const int R = K * H;
    const int64_t* ib = idx + ((size_t)b * N1 + n) * R;
    for (int e = threadIdx.x; e < 4 * R; e += blockDim.x) {
        const int f = e / R, r = e % R, h = r % H;
        const float* qp = q + ((((size_t)b * N1 + n) * 4 + f) * H + h) * D;
        const float* kp = key + (((size_t)b * N2 + (int)ib[r]) * H + h) * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc = __builtin_fmaf(qp[d], kp[d], acc);
        out[((size_t)b * N1 + n) * 4 * R + e] = acc;
    }
}

extern "C" int casmtr_qta_score_fwd(const float* q, const float* key, const int64_t* idx, float* out, int B, int N1,
                                    int N2, int K, int H, int D, casmtr_stream_t stream) {
    if (B <= 0 || N1 <= 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(N1, B);
    if (D == 32) {
        hipLaunchKernelGGL(qta_score_fwd_kernel<32>, grid, dim3(256), 4 * H * (D + 4) * sizeof(float), s, q, key, idx,
                           out, N1, N2, K, H);
    } else if (D == 64) {
        hipLaunchKernelGGL(qta_score_fwd_kernel<64>, grid, dim3(256), 4 * H * (D + 4) * sizeof(float), s, q, key, idx,
                           out, N1, N2, K, H);
    } else {
        hipLaunchKernelGGL(qta_score_fwd_generic, grid, dim3(256), 0, s, q, key, idx, out, N1, N2, K, H, D);
    }
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- a1 backward
// dq[b,n,f,h,:]  = sum_k grad[b,n,f,k,h] * key[b,idx,h,:]      (no atomics: one thread owns (f,h,d))
// dkey[b,idx,h,:] += grad[b,n,f,k,h] * q[b,n,f,h,:]            (atomics: rows are shared between quads)
__global__ __launch_bounds__(256) void qta_score_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ q,
                                                            const float* __restrict__ key, const int64_t* __restrict__ idx,
                                                            float* __restrict__ dq, float* __restrict__ dkey, int N1,
                                                            int N2, int K, int H, int D) {
    const int b = blockIdx.y, n = blockIdx.x;
    const int R = K * H, HD = H * D;
    const int64_t* ib = idx + ((size_t)b * N1 + n) * R;
    const float* gb = grad + ((size_t)b * N1 + n) * 4 * R;
    const float* qb = q + ((size_t)b * N1 + n) * 4 * HD;
    for (int e = threadIdx.x; e < 4 * HD; e += blockDim.x) {
        const int f = e / HD, h = (e % HD) / D, d = e % D;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const int j = (int)ib[k * H + h];
            acc = __builtin_fmaf(gb[(size_t)f * R + k * H + h], key[(((size_t)b * N2 + j) * H + h) * D + d], acc);
        }
        dq[((size_t)b * N1 + n) * 4 * HD + e] = acc;
    }
    for (int e = threadIdx.x; e < R * D; e += blockDim.x) {
        const int r = e / D, d = e % D, h = r % H;
        const int j = (int)ib[r];
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) acc = __builtin_fmaf(gb[(size_t)f * R + r], qb[(f * H + h) * D + d], acc);
        atomicAdd(dkey + (((size_t)b * N2 + j) * H + h) * D + d, acc);
    }
}

extern "C" int casmtr_qta_score_bwd(const float* grad, const float* q, const float* key, const int64_t* idx, float* dq,
                                    float* dkey, int B, int N1, int N2, int K, int H, int D, casmtr_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dkey, 0, sizeof(float) * (size_t)B * N2 * H * D, s);
    if (e != hipSuccess) return (int)e;
    if (B <= 0 || N1 <= 0) return 0;
    hipLaunchKernelGGL(qta_score_bwd_kernel, dim3(N1, B), dim3(256), 0, s, grad, q, key, idx, dq, dkey, N1, N2, K, H, D);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- a2 forward
// One workgroup per output token (b, n); its K*H scores and indices are staged in LDS once (the reference re-reads
// them from global for each of the D lanes); thread (h,d) walks k sequentially -> 128-byte coalesced value reads.
__global__ __launch_bounds__(256) void qta_value_agg_fwd_kernel(const float* __restrict__ score,
                                                                const float* __restrict__ value,
                                                                const int64_t* __restrict__ idx, float* __restrict__ out,
                                                                int N, int K, int H, int M, int D) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [K*H] scores, [K*H] int idx
    const int b = blockIdx.y, n = blockIdx.x;
    const int R = K * H, HD = H * D;
    float* ss = smem;
    int* si = reinterpret_cast<int*>(smem + R);
    const size_t base = ((size_t)b * N + n) * R;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        ss[r] = score[base + r];
        si[r] = (int)idx[base + r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HD; e += blockDim.x) {
        const int h = e / D;
        const float* vb = value + (size_t)b * M * HD + e;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(ss[k * H + h], vb[(size_t)si[k * H + h] * HD], acc);
        out[((size_t)b * N + n) * HD + e] = acc;
    }
}

extern "C" int casmtr_qta_value_agg_fwd(const float* score, const float* value, const int64_t* idx, float* out, int B,
                                        int N, int K, int H, int M, int D, casmtr_stream_t stream) {
    if (B <= 0 || N <= 0) return 0;
    const size_t lds = (size_t)K * H * 8;
    if (lds > 150 * 1024) return CASMTR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(qta_value_agg_fwd_kernel, dim3(N, B), dim3(256), lds, (hipStream_t)stream, score, value, idx,
                       out, N, K, H, M, D);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- a2 backward
__global__ __launch_bounds__(256) void qta_value_agg_bwd_kernel(const float* __restrict__ grad_out,
                                                                const float* __restrict__ score,
                                                                const float* __restrict__ value,
                                                                const int64_t* __restrict__ idx,
                                                                float* __restrict__ grad_score,
                                                                float* __restrict__ grad_value, int N, int K, int H, int M,
                                                                int D) {
    const int b = blockIdx.y, n = blockIdx.x;
    const int R = K * H, HD = H * D;
    const size_t base = ((size_t)b * N + n) * R;
    const float* go = grad_out + ((size_t)b * N + n) * HD;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int h = r % H;
        const int j = (int)idx[base + r];
        const float* vp = value + (((size_t)b * M + j) * H + h) * D;
        float* gv = grad_value + (((size_t)b * M + j) * H + h) * D;
        const float sc = score[base + r];
        float gs = 0.f;
        for (int d = 0; d < D; ++d) {
            const float g = go[h * D + d];
            gs = __builtin_fmaf(g, vp[d], gs);
            atomicAdd(gv + d, g * sc);
        }
        grad_score[base + r] = gs;
    }
}

// D == 32: half-wave <-> one (k, h) row, lane <-> d.  The value row is one coalesced 128-byte read, the grad_value contribution one
// coalesced 128-byte atomic (the generic kernel above walks d inside a thread: 32 scattered 4-byte atomics and reads per thread, 100x the
// forward kernel's time at the CasMTR shapes); grad_score = the half-wave's sum of g[d] * v[d].
__global__ __launch_bounds__(256) void qta_value_agg_bwd_d32_kernel(const float* __restrict__ grad_out, const float* __restrict__ score,
                                                                    const float* __restrict__ value, const int64_t* __restrict__ idx,
                                                                    float* __restrict__ grad_score, float* __restrict__ grad_value,
                                                                    int N, int K, int H, int M) {
    const int b = blockIdx.y, n = blockIdx.x;
    const int R = K * H, d = threadIdx.x & 31;
    const size_t base = ((size_t)b * N + n) * R;
    const float* go = grad_out + ((size_t)b * N + n) * H * 32;
    for (int r = threadIdx.x >> 5; r < R; r += 8) {
        const int h = r % H;
        const int j = (int)idx[base + r];
        const size_t row = (((size_t)b * M + j) * H + h) * 32 + d;
        const float g = go[h * 32 + d];
        atomicAdd(grad_value + row, g * score[base + r]);
        float gs = g * value[row];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) gs += __shfl_xor(gs, o, 32);
        if (d == 0) grad_score[base + r] = gs;
    }
}

extern "C" int casmtr_qta_value_agg_bwd(const float* grad_out, const float* score, const float* value,
                                        const int64_t* idx, float* grad_score, float* grad_value, int B, int N, int K,
                                        int H, int M, int D, casmtr_stream_t stream) {
    // grad_value is accumulated into: the caller zero-initialises it (value_aggregation.cpp:33-60 contract)
    if (B <= 0 || N <= 0) return 0;
    if (D == 32) {
        hipLaunchKernelGGL(qta_value_agg_bwd_d32_kernel, dim3(N, B), dim3(256), 0, (hipStream_t)stream, grad_out, score, value,
                           idx, grad_score, grad_value, N, K, H, M);
        CASMTR_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(qta_value_agg_bwd_kernel, dim3(N, B), dim3(256), 0, (hipStream_t)stream, grad_out, score, value,
                       idx, grad_score, grad_value, N, K, H, M, D);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- a8 forward
// One wave per query token: the query row (C floats) is read through wave-uniform scalar loads, each lane owns
// candidates k = lane, lane+64, ... and walks its key row with float4 loads (fmaf chain over c, ascending).
// The reference does C global read-modify-writes of `output` per (n,k); here the accumulator lives in a register.
template <int C>
__global__ __launch_bounds__(256) void window_score_fwd_kernel(const float* __restrict__ q, const float* __restrict__ key,
                                                               const int64_t* __restrict__ idx, float* __restrict__ out,
                                                               int N1, int N2, int K) {
    const int b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N1) return;
    const float* qp = q + ((size_t)b * N1 + n) * C;  // wave-uniform -> s_load
    for (int k = lane; k < K; k += 64) {
        const int j = (int)idx[((size_t)b * N1 + n) * K + k];
        const f32x4* kp = reinterpret_cast<const f32x4*>(key + ((size_t)b * N2 + j) * C);
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < C / 4; ++i) {
            const f32x4 kv = kp[i];
            acc = __builtin_fmaf(qp[4 * i + 0], kv.x, acc);
            acc = __builtin_fmaf(qp[4 * i + 1], kv.y, acc);
            acc = __builtin_fmaf(qp[4 * i + 2], kv.z, acc);
            acc = __builtin_fmaf(qp[4 * i + 3], kv.w, acc);
        }
        out[((size_t)b * N1 + n) * K + k] = acc;
    }
}

__global__ __launch_bounds__(256) void window_score_fwd_generic(const float* __restrict__ q, const float* __restrict__ key,
                                                                const int64_t* __restrict__ idx, float* __restrict__ out,
                                                                int N1, int N2, int K, int C) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N1) return;
    const float* qp = q + ((size_t)b * N1 + n) * C;
    for (int k = lane; k < K; k += 64) {
        const float* kp = key + ((size_t)b * N2 + (int)idx[((size_t)b * N1 + n) * K + k]) * C;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = __builtin_fmaf(qp[c], kp[c], acc);
        out[((size_t)b * N1 + n) * K + k] = acc;
    }
}

extern "C" int casmtr_window_score_fwd(const float* q, const float* key, const int64_t* idx, float* out, int B, int N1,
                                       int N2, int K, int C, casmtr_stream_t stream) {
    if (B <= 0 || N1 <= 0 || K <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((N1 + 3) / 4, B);
    if (C == 128)
        hipLaunchKernelGGL(window_score_fwd_kernel<128>, grid, dim3(256), 0, s, q, key, idx, out, N1, N2, K);
    else if (C == 64)
        hipLaunchKernelGGL(window_score_fwd_kernel<64>, grid, dim3(256), 0, s, q, key, idx, out, N1, N2, K);
    else
        hipLaunchKernelGGL(window_score_fwd_generic, grid, dim3(256), 0, s, q, key, idx, out, N1, N2, K, C);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------- a8 backward
__global__ __launch_bounds__(256) void window_score_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ q,
                                                               const float* __restrict__ key, const int64_t* __restrict__ idx,
                                                               float* __restrict__ dq, float* __restrict__ dkey, int N1,
                                                               int N2, int K, int C) {
    const int b = blockIdx.y, n = blockIdx.x;
    const float* g = grad + ((size_t)b * N1 + n) * K;
    const int64_t* ib = idx + ((size_t)b * N1 + n) * K;
    const float* qp = q + ((size_t)b * N1 + n) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        const float qc = qp[c];
        for (int k = 0; k < K; ++k) {
            const int j = (int)ib[k];
            acc = __builtin_fmaf(g[k], key[((size_t)b * N2 + j) * C + c], acc);
            atomicAdd(dkey + ((size_t)b * N2 + j) * C + c, g[k] * qc);
        }
        dq[((size_t)b * N1 + n) * C + c] = acc;
    }
}

extern "C" int casmtr_window_score_bwd(const float* grad, const float* q, const float* key, const int64_t* idx, float* dq,
                                       float* dkey, int B, int N1, int N2, int K, int C, casmtr_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dkey, 0, sizeof(float) * (size_t)B * N2 * C, s);
    if (e != hipSuccess) return (int)e;
    if (B <= 0 || N1 <= 0) return 0;
    hipLaunchKernelGGL(window_score_bwd_kernel, dim3(N1, B), dim3(128), 0, s, grad, q, key, idx, dq, dkey, N1, N2, K, C);
    CASMTR_CHECK_LAUNCH();
    return 0;
}
