// What does one LDS-DMA wave-instruction cost the CU's texture-address path as a function of its width?  fine_quad's front end issues
// four global_load_lds_dword (256 B each) per item next to sixteen global_load_lds_dwordx4 (1 KB each): if the cost is per address
// (64 lanes) rather than per byte, the four narrow ones are a fifth of the item's address-path time and worth merging into one wide one.
//   hipcc --offload-arch=gfx950 -O2 dma_width.hip -o dma_width && ./dma_width
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W> __device__ __forceinline__ void dma(const float* base, unsigned off, unsigned lds) {
    if (W == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(lds) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(off), "s"(base), "s"(lds) : "memory");
}
template <int W, int SAME>   // SAME: all 64 lanes in one contiguous run (a staging read) instead of 8 random 128-B rows
__global__ __launch_bounds__(64) void k(const float* kv, int nq, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x;
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
    const float* base = kv + (size_t)(blockIdx.x & 7) * nq * 128;
    unsigned s = (blockIdx.x * 64 + (SAME ? 0 : (lane >> 3))) * 2654435761u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned row = (s >> 8) % (unsigned)(nq * 4 - 8);
            const unsigned off = SAME ? row * 128u + lane * (W * 4) : row * 128u + (lane & 7) * (W * 4);
            dma<W>(base, off, lds + j * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    if (smem[lane] == 123.f) sink[0] = 1.f;
}
template <int W, int SAME> void run(const float* kv, unsigned long long* cyc, float* sink, int per_cu, const char* what) {
    const int blocks = 256 * per_cu, iters = 400;
    static unsigned long long h[8192];
    k<W, SAME><<<blocks, 64, 8192>>>(kv, 2704, iters, cyc, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<W, SAME><<<blocks, 64, 8192>>>(kv, 2704, iters, cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int i = 0; i < blocks; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-40s %2d waves/CU: %.1f cycles per wave-instruction and CU, %.1f B/clk/CU (%.3f ms)\n", what, per_cu,
           (double)mx / ((double)iters * 8 * per_cu), (double)iters * 8 * per_cu * 64 * W * 4 / (double)mx, ms);
}
int main() {
    float *kv, *sink; unsigned long long* cyc;
    hipMalloc(&kv, (size_t)8 * 2704 * 512 + (1 << 20)); hipMemset(kv, 0, (size_t)8 * 2704 * 512 + (1 << 20));
    hipMalloc(&cyc, 8192 * 8); hipMalloc(&sink, 64);
    for (int pc : {8, 16}) {
        run<4, 0>(kv, cyc, sink, pc, "dwordx4, 8 random 128-B rows");
        run<1, 0>(kv, cyc, sink, pc, "dword,   8 random rows (32 B of each)");
        run<4, 1>(kv, cyc, sink, pc, "dwordx4, one contiguous 1 KB run");
        run<1, 1>(kv, cyc, sink, pc, "dword,   one contiguous 256 B run");
    }
    return 0;
}
