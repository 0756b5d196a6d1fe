// Probe: operand / result lane layout of v_mfma_f32_16x16x4_f32 on gfx950, and whether a k-sequence of them is the exact
// k-ascending fmaf chain (within an instruction k = 0..3 in order, instructions in order).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -w mfma16x16x4_layout.hip -o mfma16 && ./mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D, int KS) {   // A [16][4*KS], B [4*KS][16], D [16][16]
    const int l = threadIdx.x, mn = l & 15, k = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < KS; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mn * 4 * KS + 4 * j + k], B[(4 * j + k) * 16 + mn], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + mn] = acc[r];   // hypothesis: lane (n = l%16, rows 4*(l/16)+r)
}

int main() {
    const int KS = 8, K = 4 * KS;
    float *hA = (float*)malloc(16 * K * 4), *hB = (float*)malloc(K * 16 * 4), hD[256];
    srand(1);
    for (int i = 0; i < 16 * K; ++i) { hA[i] = (float)rand() / RAND_MAX - 0.5f; hB[i] = (float)rand() / RAND_MAX - 0.5f; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, K * 16 * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 16 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, KS);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float c = 0.f;
            for (int k = 0; k < K; ++k) c = fmaf(hA[m * K + k], hB[k * 16 + n], c);
            if (c != hD[m * 16 + n]) { if (bad < 5) printf("D[%d][%d]: got %.9g want %.9g\n", m, n, hD[m * 16 + n], c); ++bad; }
        }
    printf("v_mfma_f32_16x16x4_f32 == k-ascending fmaf chain, D[row 4*(l/16)+r][col l%%16]: %d mismatches of 256\n", bad);
    return 0;
}
