// Probe: the same randomly gathered 128-byte rows as gather_bw.hip, but moved by LDS-DMA (global_load_lds_dwordx4, 8 rows per
// wave-instruction, 8 instructions per 8 KB wave-private buffer, two buffers, one batch in flight while the previous one is
// read back with ds_read_b128) instead of loads into registers.  Question: is the DMA path's rate per CU below the
// register path's for L2-resident tables (tools/probes/gather_bw.hip: 29 TB/s at <= 3 MB per XCD)?
//   hipcc --offload-arch=gfx950 -O2 gather_dma_bw.hip -o gather_dma_bw && ./gather_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* base, unsigned byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory");
}

template <int WAVES, int READBACK>
__global__ __launch_bounds__(WAVES * 64) void gather_dma(const float* __restrict__ tab, float* __restrict__ out, int rows_per_region,
                                                         int nregion, int iters, int pitch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, pc = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* buf = smem + wave * 4096;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)buf);
    const int xcd = blockIdx.x & 7;
    const int region = xcd % nregion;
    const float* base = tab + (size_t)region * rows_per_region * pitch;
    unsigned s = (blockIdx.x * 256 + threadIdx.x / 8) * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int bsel) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = (s >> 8) % (unsigned)rows_per_region;
            glds16(base, (r * (unsigned)pitch + pc * 4) * 4u, buf_lds + bsel * 8192 + j * 1024);
        }
    };
    issue(0);
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < iters) { issue((it + 1) & 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (READBACK) {
            const float* bp = buf + (it & 1) * 2048;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bp + lane * 32 + ((u ^ ((lane >> 1) & 7)) * 4));
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            asm volatile("" : "+v"(acc));
        }
    }
    if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int WAVES, int READBACK>
static void run(const float* tab, float* out, int wg_per_cu, size_t maxrows) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wg_per_cu * 8, iters = 16;
    const size_t lds = (size_t)WAVES * 16384;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather_dma<WAVES, READBACK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int pitch : {32, 128})
    for (int kb : {512, 2816, 11264, 22528, 65536}) {
        const int rows = kb * 1024 / 128;
        if ((size_t)rows * pitch * 4 * 8 > maxrows * 128) continue;
        gather_dma<WAVES, READBACK><<<blocks, WAVES * 64, lds>>>(tab, out, rows, 8, iters, pitch);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) gather_dma<WAVES, READBACK><<<blocks, WAVES * 64, lds>>>(tab, out, rows, 8, iters, pitch);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)blocks * WAVES * iters * 8 * 1024;
        printf("DMA waves/WG %d x %d WG/CU readback %d | row pitch %4d B, region %7d KB per XCD: %.3f ms, %.2f TB/s\n", WAVES, wg_per_cu,
               READBACK, pitch * 4, kb, ms, bytes / ms / 1e9);
    }
}

int main() {
    const size_t maxrows = (size_t)8 * 1024 * 1024;
    float *tab, *out;
    hipMalloc(&tab, maxrows * 128);
    hipMalloc(&out, 4096);
    hipMemset(tab, 0, maxrows * 128);
    run<2, 1>(tab, out, 4, maxrows);    //  8 waves / CU (the window / cascade kernels)
    run<2, 0>(tab, out, 4, maxrows);
    run<4, 1>(tab, out, 2, maxrows);    //  8 waves / CU as 2 x 4
    run<4, 1>(tab, out, 4, maxrows);    // 16 waves / CU needs 64 KB x 4 -> does not fit: hardware caps at 2
    run<1, 1>(tab, out, 8, maxrows);    //  8 single-wave workgroups per CU
    return 0;
}
