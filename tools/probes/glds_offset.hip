// Probe: what does the instruction's immediate `offset:` do on global_load_lds_dwordx4 (gfx950)?
//   (a) is it added to the GLOBAL address, (b) is it added to the LDS destination (M0 base + offset + lane * 16)?
// One wave; LDS pre-filled with -1; source buffer holds its own float index.  Prints where the data landed and what it was.
//   hipcc --offload-arch=gfx950 -O2 glds_offset.hip -o glds_offset && ./glds_offset
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void probe(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = -1.f;
    __syncthreads();
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds);
    const unsigned voff = lane * 16;
    // one statement: M0 <- LDS base, then two DMA instructions, the second with offset:1024
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "s_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(src), "s"(dst) : "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
    float *src, *out;
    hipMalloc(&src, 65536 * 4);
    hipMalloc(&out, 4096 * 4);
    float* h = new float[65536];
    for (int i = 0; i < 65536; ++i) h[i] = (float)i;
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    float r[4096];
    hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    // report runs of written floats
    int i = 0;
    while (i < 4096) {
        if (r[i] < 0) { ++i; continue; }
        int j = i;
        while (j + 1 < 4096 && r[j + 1] == r[j] + 1) ++j;
        printf("LDS floats [%d, %d] (bytes %d..%d) <- source floats [%g, %g] (bytes %g..)\n", i, j, i * 4, j * 4 + 3, r[i], r[j], r[i] * 4);
        i = j + 1;
    }
    printf("expected if offset applies to BOTH: LDS bytes 0..1023 <- src bytes 0.. and LDS bytes 2048..3071 <- src bytes 2048..\n");
    return 0;
}
