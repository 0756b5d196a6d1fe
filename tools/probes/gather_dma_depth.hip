// Probe: how the LDS-DMA gather rate of random 128-byte rows depends on the pipeline depth per wave at a FIXED LDS budget.
// The shipped DMA kernels (fine / cascade / window) keep two 8 KB buffers per wave: one 8-instruction batch in flight while the other
// is consumed.  Here a wave owns a ring of NCH chunks of CH KB (CH DMA instructions each); NCH - 1 chunks are in flight while the
// oldest is read back.  Waves are single-wave workgroups, as many per CU as the LDS allows.  Persistent-style: fixed iterations.
//   hipcc --offload-arch=gfx950 -O2 gather_dma_depth.hip -o gather_dma_depth && ./gather_dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* base, unsigned byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int CH, int NCH>
__global__ __launch_bounds__(64) void gather_ring(const float* __restrict__ tab, float* __restrict__ out, int rows_per_region, int iters,
                                                  int pitch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, pc = lane & 7;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
    const int xcd = blockIdx.x & 7;
    const float* base = tab + (size_t)xcd * rows_per_region * pitch;
    unsigned s = (blockIdx.x * 64 + threadIdx.x / 8) * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int slot) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = (s >> 8) % (unsigned)rows_per_region;
            glds16(base, (r * (unsigned)pitch + pc * 4) * 4u, buf_lds + (unsigned)((slot * CH + j) * 1024));
        }
    };
#pragma unroll
    for (int c = 0; c < NCH - 1; ++c) issue(c);
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        // free slot = the one consumed in the previous iteration (slot - 1): refill it, then wait for the oldest chunk
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue((slot + NCH - 1) % NCH);
        vmwait<CH * (NCH - 1)>();
        const float* bp = smem + slot * CH * 256;
#pragma unroll
        for (int j = 0; j < CH; ++j) {   // read the chunk back: 1 KB per DMA instruction = 64 lanes x 16 B
            const f32x4 v = *reinterpret_cast<const f32x4*>(bp + j * 256 + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        asm volatile("" : "+v"(acc));
        slot = (slot + 1) % NCH;
    }
    vmwait<0>();
    if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int CH, int NCH>
static void run(const float* tab, float* out, int region_kb) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)CH * NCH * 1024;
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 32) per_cu = 32;
    const int blocks = 256 * per_cu, iters = 8 * 64 / CH;   // 512 KB per wave
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather_ring<CH, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int rows = region_kb * 1024 / 128, pitch = 256;   // rows 1 KB apart: one head's slice of token-major rows
    gather_ring<CH, NCH><<<blocks, 64, lds>>>(tab, out, rows, iters, pitch);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) gather_ring<CH, NCH><<<blocks, 64, lds>>>(tab, out, rows, iters, pitch);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)blocks * iters * CH * 1024;
    printf("chunk %d KB x ring %d (%2zu KB LDS/wave, %2d KB in flight/wave, %2d waves/CU, %3d KB in flight/CU) region %5d KB/XCD: %.3f ms %.2f TB/s\n",
           CH, NCH, lds / 1024, CH * (NCH - 1), per_cu, CH * (NCH - 1) * per_cu, region_kb, ms, bytes / ms / 1e9);
}

int main() {
    float *tab, *out;
    const size_t bytes = (size_t)8 * 23000 * 1024 * 8;   // 8 regions of up to 22.5 MB of 128-B rows at 1 KB pitch
    hipMalloc(&tab, bytes);
    hipMalloc(&out, 4096);
    hipMemset(tab, 0, bytes);
    for (int kb : {2816, 22528}) {
        run<8, 2>(tab, out, kb);    // the shipped kernels' pipeline
        run<4, 4>(tab, out, kb);    // same LDS, deeper
        run<2, 8>(tab, out, kb);
        run<1, 16>(tab, out, kb);
        run<8, 3>(tab, out, kb);    // more LDS per wave, fewer waves
        run<8, 4>(tab, out, kb);
        run<4, 2>(tab, out, kb);    // less LDS per wave, more waves
        run<2, 2>(tab, out, kb);
        run<2, 4>(tab, out, kb);
        run<1, 8>(tab, out, kb);
    }
    return 0;
}
