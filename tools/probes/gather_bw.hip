// Probe: what rate can a CU array pull randomly gathered 128-byte rows at (8 lanes x dwordx4 per row, 8 rows per wave-load,
// 8 loads in flight per wave, 24 waves per CU), as a function of the table size an XCD gathers from?  This is the access
// pattern of quad_attn_kernel's key / value reads with everything else removed.
//   hipcc --offload-arch=gfx950 -O2 gather_bw.hip -o gather_bw && ./gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 6) void gather(const float* __restrict__ tab, float* __restrict__ out, int rows_per_region,
                                                 int nregion, int iters, int pitch) {
    const int lane = threadIdx.x & 63, g = lane >> 3, pc = lane & 7;
    // block -> region like xcd_chunk_remap: block i lands on XCD i % 8; each XCD owns nregion/8 regions
    const int xcd = blockIdx.x & 7;
    const int region = xcd % nregion;
    const float* base = tab + (size_t)region * rows_per_region * pitch;
    unsigned s = (blockIdx.x * 256 + threadIdx.x / 8) * 2654435761u + 12345u;   // same seed for the 8 lanes of a row group
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = (s >> 8) % (unsigned)rows_per_region;
            v[j] = *reinterpret_cast<const f32x4*>(base + (size_t)r * pitch + pc * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w + g;
}

int main() {
    const size_t maxrows = (size_t)8 * 4 * 1024 * 1024;   // 8 regions x 4M rows x 128 B = 4 GB max
    float *tab, *out;
    hipMalloc(&tab, maxrows * 128);
    hipMalloc(&out, 4096);
    hipMemset(tab, 0, maxrows * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 6 * 8, iters = 16;
    for (int pitch : {32, 256})
    for (int kb : {512, 2048, 2816, 8192, 11264, 22528, 65536}) {   // bytes actually gathered from, per region, in KB
        const int rows = kb * 1024 / 128;
        if ((size_t)rows * pitch * 4 * 8 > maxrows * 128) continue;
        gather<<<blocks, 256>>>(tab, out, rows, 8, iters, pitch);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) gather<<<blocks, 256>>>(tab, out, rows, 8, iters, pitch);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)blocks * 4 * iters * 8 * 1024;   // waves x iters x 8 loads x 1 KB
        printf("row pitch %4d B, region %7d KB per XCD: %.3f ms, %.2f TB/s of gathered rows\n", pitch * 4, kb, ms, bytes / ms / 1e9);
    }
    return 0;
}
