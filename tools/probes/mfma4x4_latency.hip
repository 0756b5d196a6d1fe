// Probe: issue interval / dependent latency of v_mfma_f32_4x4x1_16B_f32 on gfx950: NCH independent accumulator chains, one wave per SIMD
// and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 mfma4x4_latency.hip -o mfma4x4_latency && ./mfma4x4_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ void k(float* out, int iters, float a, float b) {
    f32x4 acc[NCH];
    for (int i = 0; i < NCH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a + u, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH>
void run(float* d, int threads) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NCH><<<256, threads>>>(d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    k<NCH><<<256, threads>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * NCH;   // MFMAs per wave
    printf("chains %d, %d waves per SIMD: %.2f ns per MFMA per wave (%.1f cycles at 2.1 GHz)\n", NCH, threads / 256, ms * 1e6 / n, ms * 1e6 / n * 2.1);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    for (int threads : {256, 512}) {
        run<1>(d, threads); run<2>(d, threads); run<4>(d, threads);
    }
    return 0;
}
