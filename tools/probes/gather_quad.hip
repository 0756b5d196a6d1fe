// Probe for the round-3 fine-level rewrite: LDS-DMA gather rate of the QTAttB fine-level access pattern as a function of
//   * the K/V layout: token-major (128-B head rows at a 1 KB pitch, 4 children of a parent = 2 + 2 rows in two image rows) against
//     quad-major per head (the 4 children of a parent = one contiguous 512-B run, a head's slice contiguous),
//   * the side streams that pass through the same L2 (queries, outputs, previous-level indices: int64 [.,K,H] lines shared by the
//     8 heads = 8x over-fetch per XCD, against a compact int32 per-head table),
//   * ring shape / waves per CU at a fixed LDS budget.
// An "item" gathers PAR parents x 512 B of K and of V from a region of NQ quads (one (pair, head) slice: 2704 quads = 1.38 MB + 1.38 MB).
//   hipcc --offload-arch=gfx950 -O2 gather_quad.hip -o gather_quad && ./gather_quad
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 4 DMA instructions (4 KB) with ONE M0 write: LDS destination advanced with s_add between the loads
__device__ __forceinline__ void glds4(const float* base, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %4\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %4"
                 :: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(lds_dst) : "memory");
}

struct P {
    const float* kv;      // 8 regions (one per XCD), each 2 * region floats (K then V)
    const float* qstream; // streaming reads
    float* ostream;       // streaming writes
    const long long* idx; // index stream
    float* sink;
    int nq;               // quads per region
    int layout;           // 0 = token-major rows (1 KB pitch, 2x2 children), 1 = quad-major (512 B contiguous)
    int wq;               // quads per image row (layout 0)
    int stream;           // 0 none, 1 = q + out + strided int64 index lines (1 KB per item), 2 = q + out + compact 64-B index
    int items;            // items per wave
};

// CHK = 4 KB chunks per stage (1 or 2), two stages in the ring: one in flight while the other is read back
template <int CHK>
__global__ __launch_bounds__(64) void gather_items(const P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
    const int xcd = blockIdx.x & 7, wave_in_xcd = blockIdx.x >> 3, waves_per_xcd = gridDim.x >> 3;
    const size_t region_floats = (size_t)p.nq * 128 * (p.layout == 0 ? 8 : 1);   // token-major: all 8 heads' rows interleaved
    const float* kbase = p.kv + (size_t)xcd * 2 * region_floats;
    const float* vbase = kbase + region_floats;
    unsigned s = (blockIdx.x * 64 + (lane >> 5)) * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // lane -> (parent slot within the instruction = lane/32, child = (lane/8)%4, 16-B unit = lane%8)
    const unsigned child = (lane >> 3) & 3, un = lane & 7;
    auto off_of = [&](unsigned parent) -> unsigned {
        if (p.layout == 1) return parent * 512u + child * 128u + un * 16u;
        const unsigned r = parent / p.wq * 2 + (child >> 1), c = parent % p.wq * 2 + (child & 1);
        return (r * (unsigned)(2 * p.wq) + c) * 1024u + un * 16u;   // head 0's 128-B slice of a 1 KB token row
    };
    auto issue = [&](const float* base, int stage, int half) {   // 4 KB = 8 parents
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u;
            o[j] = off_of((s >> 8) % (unsigned)p.nq);
        }
        glds4(base, o[0], o[1], o[2], o[3], buf_lds + (unsigned)((stage * CHK + half) * 4096));
    };
    auto consume = [&](int stage) {
#pragma unroll
        for (int j = 0; j < 4 * CHK; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (stage * CHK * 4 + j) * 256 + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        asm volatile("" : "+v"(acc));
    };
    // stage sequence per item: K (2 / CHK stages), V (2 / CHK stages); 16 parents = 8 KB each
    constexpr int SPI = 4 / CHK;   // stages per item
    int st = 0;
    auto issue_stage = [&](int n) {   // n-th stage of the wave's stream
        const int w = n % SPI;
        const float* base = (w < SPI / 2) ? kbase : vbase;
#pragma unroll
        for (int hh = 0; hh < CHK; ++hh) issue(base, n & 1, hh);
    };
    issue_stage(0);
    const int total = p.items * SPI;
    for (int n = 0; n < total; ++n) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (n + 1 < total) { issue_stage(n + 1); vmwait<4 * CHK>(); } else vmwait<0>();
        if (p.stream && (n % SPI) == 0) {
            // side streams, issued behind the wait (vmcnt is in order)
            const size_t item = (size_t)(n / SPI) * waves_per_xcd * 8 + blockIdx.x;
            if (lane < 32) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(p.qstream + item * 128 + lane * 4);
                acc.x += q.x;
                *reinterpret_cast<f32x4*>(p.ostream + item * 128 + lane * 4) = acc;
            }
            if (p.stream == 1) { if (lane < 16) acc.y += (float)p.idx[item * 128 + lane * 8]; }        // 16 x 8 B at a 64-B stride
            else { if (lane < 16) acc.y += (float)reinterpret_cast<const int*>(p.idx)[item * 16 + lane]; }  // 64 B contiguous
        }
        consume(n & 1);
        (void)st;
    }
    if (acc.x == 123.456f) p.sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    (void)wave_in_xcd;
}

template <int CHK>
static void run(P p, int per_cu, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)2 * CHK * 4096;
    const int blocks = 256 * per_cu;
    p.items = 2704 * 8 * 8 / blocks;   // one level-0 launch: 8 pairs x 2704 quads x 8 heads = 173 056 items
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather_items<CHK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    gather_items<CHK><<<blocks, 64, lds>>>(p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) gather_items<CHK><<<blocks, 64, lds>>>(p);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double bytes = (double)blocks * p.items * 16384;
    printf("%-46s stage %d KB, %2d waves/CU: %.3f ms  %.2f TB/s gathered  (%.0f ns per item per wave)\n", what, 4 * CHK, per_cu, ms,
           bytes / ms / 1e9, ms * 1e6 / p.items);
    fflush(stdout);
}

int main() {
    P p{};
    float* kv; float *qs, *os, *sink; long long* idx;
    const int nq = 2704, wq = 52;
    const size_t kv_floats = (size_t)8 * 2 * nq * 128 * 8;
    hipMalloc(&kv, kv_floats * 4);
    hipMemset(kv, 0, kv_floats * 4);
    const size_t n_items = (size_t)2704 * 8 * 8 + 4096;
    hipMalloc(&qs, n_items * 512); hipMalloc(&os, n_items * 512); hipMalloc(&idx, n_items * 1024); hipMalloc(&sink, 4096);
    hipMemset(qs, 0, n_items * 512); hipMemset(idx, 0, n_items * 1024);
    p.kv = kv; p.qstream = qs; p.ostream = os; p.idx = idx; p.sink = sink; p.nq = nq; p.wq = wq;
    for (int stream = 0; stream <= 2; ++stream)
        for (int layout = 0; layout <= 1; ++layout) {
            if (stream == 1 && layout == 1) continue;
            if (stream == 2 && layout == 0) continue;
            p.layout = layout; p.stream = stream;
            char what[128];
            snprintf(what, sizeof what, "%s, %s", layout ? "quad-major 512-B runs" : "token-major 128-B rows",
                     stream == 0 ? "no side streams" : stream == 1 ? "q/out + int64 idx lines" : "q/out + compact idx");
            run<2>(p, 8, what);
            run<2>(p, 10, what);
            run<1>(p, 16, what);
            run<1>(p, 20, what);
            run<1>(p, 24, what);
        }
    return 0;
}
