// Probe: operand / result lane layout of v_mfma_f32_4x4x1_16B_f32 on gfx950, and whether a k-sequence of them is the
// exact fmaf chain.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma4x4_layout.hip -o mfma4x4 && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D, int K) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + l], B[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}

int main() {
    const int K = 32;
    float *hA = (float*)malloc(K * 64 * 4), *hB = (float*)malloc(K * 64 * 4), hD[256];
    srand(1);
    for (int i = 0; i < K * 64; ++i) { hA[i] = (float)rand() / RAND_MAX - 0.5f; hB[i] = (float)rand() / RAND_MAX - 0.5f; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, K * 64 * 4); hipMalloc(&dB, K * 64 * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, K * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 64 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, K);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4*blk + j ; acc[r] = sum_k A[k][4*blk + r] * B[k][4*blk + j]  (fmaf chain, k ascending)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, j = l & 3;
            float c = 0.f;
            for (int k = 0; k < K; ++k) c = fmaf(hA[k * 64 + 4 * blk + r], hB[k * 64 + 4 * blk + j], c);
            if (c != hD[l * 4 + r]) { if (bad < 5) printf("lane %d r %d: got %.9g want %.9g\n", l, r, hD[l * 4 + r], c); ++bad; }
        }
    printf("hypothesis D[lane=4b+j][r] = chain_k A[4b+r]*B[4b+j]: %d mismatches of 256\n", bad);
    return 0;
}
