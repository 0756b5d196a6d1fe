// Probe: issue rate of v_mfma_f32_32x32x16_f16 on gfx950 and the clock the chip holds while every SIMD runs them back to back.
//   hipcc --offload-arch=gfx950 -O2 mfma16_rate.hip -o mfma16_rate && ./mfma16_rate
// One wave per SIMD (W = 4 waves per workgroup, one workgroup per CU) or two (W = 8); NACC independent accumulators in rotation;
// random (non-zero) operands.  Prints shader cycles per MFMA (s_memtime) and MFMAs per second per SIMD (wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void rate(const h8* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int l = threadIdx.x;
    h8 a = in[l & 63], b = in[64 + (l & 63)];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int waves, int ncu, const h8* din, float* dout, long long* dcyc) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate<NACC><<<ncu, 64 * waves>>>(din, dout, dcyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<NACC><<<ncu, 64 * waves>>>(din, dout, dcyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NACC;   // MFMAs per wave
    printf("%d accumulators, %d waves per CU: %.1f counter ticks per MFMA and wave, %.2f us; per SIMD %.1f ns per MFMA (32 cycles at 2.4 GHz = 13.3 ns)\n",
           NACC, waves, (double)c / n, ms * 1e3, ms * 1e6 / (n * waves / 4.0));
}

int main() {
    h8* din; float* dout; long long* dcyc;
    hipMalloc(&din, 128 * sizeof(h8)); hipMalloc(&dout, 256 * 1024 * 4); hipMalloc(&dcyc, 256 * 8);
    _Float16 h[128 * 8];
    srand(3);
    for (int i = 0; i < 128 * 8; ++i) h[i] = (_Float16)((float)rand() / RAND_MAX - 0.5f);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%d CUs, clock rate reported %d kHz\n", ncu, p.clockRate);
    run<4>(4, ncu, din, dout, dcyc);
    run<2>(4, ncu, din, dout, dcyc);
    run<1>(4, ncu, din, dout, dcyc);
    run<4>(8, ncu, din, dout, dcyc);
    run<2>(8, ncu, din, dout, dcyc);
    run<4>(4, 1, din, dout, dcyc);
    return 0;
}
