// Round-5 attribution probe for the quad-major gather kernels: what separates the 22 B/clk/CU of fine_quad_kernel from the 36 B/clk/CU
// of the pure gather (tools/probes/gather_dma_depth.hip)?  An "item" is fine_quad's level-0 item: 16 parents x 512 B of K and of V
// (16 KB) through a two-slot ring of 4 KB chunks, 4 DMA instructions per chunk behind one M0 write.  Switches add, one at a time, what
// the real kernel does besides gathering:
//   reads   0 none | 1 every gathered byte once by ds_read_b128 | 2 K by ds_read_b128, V by ds_read_b32 (the kernel's read set)
//   bcast   n extra broadcast ds_read_b128 per item (the kernel: 8 for the queries + 8 for the probabilities)
//   rt      softmax-like LDS round trip per item (4 ds_write_b32 + 1 ds_read_b128 + 2 ds_write_b64)
//   mfma    n v_mfma_f32_4x4x1 per item (the kernel: 64)
//   lds     bytes of LDS per wave (occupancy knob: the kernel has 11 392)
//   pairs   regions walked one after the other per XCD (1: everything L2-resident after the first touch; 8: the kernel's pair walk
//           with its compulsory misses and slice transitions)
//   side    side streams (q 512 B in, 512 B out, 64-B index line per item)
// Reported: us for one level-0 launch worth of items (173 056), TB/s of gathered rows, B/clk/CU from s_memtime of the slowest sampled wave.
//   hipcc --offload-arch=gfx950 -O2 gather_attrib.hip -o gather_attrib && ./gather_attrib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds_chunk(const float* base_m3072, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, %4 offset:3072"
                 :: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base_m3072), "s"(lds_dst) : "memory");
}

struct P {
    const float* kv;       // [8 XCDs][pairs][2][nq*128]
    const float* qstream;
    float* ostream;
    const int* idx;
    float* sink;
    unsigned long long* cyc;   // per block: s_memtime span
    int nq, pairs, items_per_pair;   // items per wave and pair
    int reads, bcast, rt, mfma, side;
    int serial;   // 1: at most ONE 4 KB chunk in flight per wave (chunk n + 1 is issued when chunk n has landed, then n is consumed)
};

__global__ __launch_bounds__(64) void probe(const P p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const unsigned ring = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)smem);
    float* extra = smem + 2048;   // 1 KB+ scratch behind the ring (broadcast reads, round trip)
    const int xcd = blockIdx.x & 7;
    const size_t slice = (size_t)p.nq * 128;
    unsigned s = (blockIdx.x * 64 + (lane >> 5)) * 2654435761u + 12345u;
    const unsigned child = (lane >> 3) & 3, un = lane & 7;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto issue = [&](const float* base, int slot) {
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u;
            o[j] = ((s >> 8) % (unsigned)p.nq) * 512u + child * 128u + un * 16u + 3072u - j * 1024u;
        }
        glds_chunk(base - 768, o[0], o[1], o[2], o[3], ring + (unsigned)slot * 4096u);
    };
    auto read128 = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + slot * 1024 + j * 256 + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    };
    auto read32 = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] += smem[slot * 1024 + j * 64 + lane];
    };
    auto mfmas = [&](int n) {
        for (int i = 0; i < n; i += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(acc[k], acc[(k + 1) & 3], c4[k], 0, 0, 0);
        }
    };
    unsigned long long t0 = 0;
    if (p.serial) {
        // chunk-granular pipeline: K0 K1 V0 V1 of item after item; in flight while chunk n is consumed: chunk n + 1 only
        const long long nchunks = (long long)p.pairs * p.items_per_pair * 4;
        auto base_of = [&](long long n) {
            const int pr = (int)(n / (4ll * p.items_per_pair));
            return p.kv + ((size_t)xcd * p.pairs + pr) * 2 * slice + (((n >> 1) & 1) ? slice : 0);
        };
        issue(base_of(0), 0);
        for (long long n = 0; n < nchunks; ++n) {
            if (n == 4) t0 = __builtin_readcyclecounter();
            vmwait<0>();
            if (n + 1 < nchunks) issue(base_of(n + 1), (int)((n + 1) & 1));
            const int c = (int)(n & 1);
            if (p.side && (n & 3) == 0) {
                const size_t item = ((size_t)(n >> 2) * gridDim.x + blockIdx.x);
                if (lane < 32) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(p.qstream + item * 128 + lane * 4);
                    *reinterpret_cast<f32x4*>(p.ostream + item * 128 + lane * 4) = acc;
                    acc.x += q.x;
                } else if (lane < 48) acc.y += (float)p.idx[item * 16 + (lane - 32)];
            }
            for (int b = 0; b < p.bcast / 4; ++b) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(extra + ((lane & 3) * 2 + (lane >> 5)) * 36 + (b & 7) * 4);
                acc.y += v.x; acc.w += v.z;
            }
            if (p.reads == 1 || (p.reads == 2 && !(n & 2))) read128(c); else if (p.reads == 2) read32(c);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mfmas(p.mfma / 4);
            if (p.rt && (n & 3) == 1) {
#pragma unroll
                for (int f = 0; f < 4; ++f) extra[f * 68 + lane] = acc[f];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4 v = *reinterpret_cast<const f32x4*>(extra + (lane >> 4) * 68 + (lane & 15) * 4);
                acc.x += v.y;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                *reinterpret_cast<f32x2*>(extra + ((lane >> 4) * 2) * 36 + 2 * (lane & 15)) = (f32x2){v.x, v.z};
                *reinterpret_cast<f32x2*>(extra + ((lane >> 4) * 2 + 1) * 36 + 2 * (lane & 15)) = (f32x2){v.y, v.w};
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    } else
    for (int pr = 0; pr < p.pairs; ++pr) {
        const float* kb = p.kv + ((size_t)xcd * p.pairs + pr) * 2 * slice;
        const float* vb = kb + slice;
        if (pr == 0) { issue(kb, 0); issue(kb, 1); }
        for (int it = 0; it < p.items_per_pair; ++it) {
            const bool last = (pr + 1 == p.pairs) && (it + 1 == p.items_per_pair);
            const bool lastp = it + 1 == p.items_per_pair;
            if (pr == 0 && it == 1) t0 = __builtin_readcyclecounter();
            // ---- K pass: both slots
            vmwait<0>();
            if (p.side) {
                const size_t item = ((size_t)(pr * p.items_per_pair + it) * gridDim.x + blockIdx.x);
                if (lane < 32) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(p.qstream + item * 128 + lane * 4);
                    *reinterpret_cast<f32x4*>(p.ostream + item * 128 + lane * 4) = acc;
                    acc.x += q.x;
                } else if (lane < 48) acc.y += (float)p.idx[item * 16 + (lane - 32)];
            }
            if (p.bcast) {   // 8 independent broadcast reads in flight together, as the kernel's query operand
                f32x4 v[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) v[b] = *reinterpret_cast<const f32x4*>(extra + (lane & 3) * 36 + b * 4);
#pragma unroll
                for (int b = 0; b < 8; ++b) { acc.x += v[b].x; acc.z += v[b].w; }
            }
            if (p.reads) { read128(0); read128(1); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(vb, 0); issue(vb, 1);
            mfmas(p.mfma / 2);
            if (p.rt) {
#pragma unroll
                for (int f = 0; f < 4; ++f) extra[f * 68 + lane] = acc[f];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4 v = *reinterpret_cast<const f32x4*>(extra + (lane >> 4) * 68 + (lane & 15) * 4);
                acc.x += v.y;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                *reinterpret_cast<f32x2*>(extra + ((lane >> 4) * 2) * 36 + 2 * (lane & 15)) = (f32x2){v.x, v.z};
                *reinterpret_cast<f32x2*>(extra + ((lane >> 4) * 2 + 1) * 36 + 2 * (lane & 15)) = (f32x2){v.y, v.w};
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            // ---- V chunks
            const float* nk = lastp ? kb + 2 * slice : kb;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                vmwait<4>();
                if (p.bcast) {
                    f32x4 v[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[b] = *reinterpret_cast<const f32x4*>(extra + ((lane & 3) * 2 + (lane >> 5)) * 36 + (4 * c + b) * 4);
#pragma unroll
                    for (int b = 0; b < 4; ++b) { acc.y += v[b].x; acc.w += v[b].z; }
                }
                if (p.reads == 1) read128(c); else if (p.reads == 2) read32(c);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!last) issue(nk, c);
                mfmas(p.mfma / 4);
            }
        }
    }
    vmwait<0>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) p.cyc[blockIdx.x] = t1 - t0;
    const float r = acc.x + acc.y + acc.z + acc.w + c4[0].x + c4[1].y + c4[2].z + c4[3].w;
    if (r == 123.456f) p.sink[threadIdx.x] = r;
}

int main(int argc, char** argv) {
    const int nq = 2704, PAIRS = 8;
    float *kv, *qs, *os, *sink; int* idx; unsigned long long* cyc;
    const size_t kv_floats = (size_t)8 * PAIRS * 2 * nq * 128;
    hipMalloc(&kv, kv_floats * 4 + (4 << 20)); hipMemset(kv, 0, kv_floats * 4 + (4 << 20));
    const size_t n_items = (size_t)2704 * 8 * 8 + 65536;
    hipMalloc(&qs, n_items * 512); hipMalloc(&os, n_items * 512); hipMalloc(&idx, n_items * 64); hipMalloc(&sink, 4096);
    hipMalloc(&cyc, 8192 * 8);
    hipMemset(qs, 0, n_items * 512); hipMemset(idx, 0, n_items * 64);
    struct Cfg { const char* what; int per_cu, lds, pairs, reads, bcast, rt, mfma, side, serial; };
    const Cfg cfgs[] = {
        {"dma only", 10, 8192, 1, 0, 0, 0, 0, 0, 0},
        {"dma only", 16, 8192, 1, 0, 0, 0, 0, 0, 0},
        {"+ K b128 / V b32 reads", 10, 8192, 1, 2, 0, 0, 0, 0, 0},
        {"+ 16 broadcast b128 (independent)", 10, 9600, 1, 2, 16, 0, 0, 0, 0},
        {"+ 16 broadcast b128 (independent)", 16, 9600, 1, 2, 16, 0, 0, 0, 0},
        {"+ softmax round trip", 10, 9600, 1, 2, 16, 1, 0, 0, 0},
        {"+ softmax round trip", 16, 9600, 1, 2, 16, 1, 0, 0, 0},
        {"+ 64 mfma", 10, 9600, 1, 2, 16, 1, 64, 0, 0},
        {"+ 64 mfma", 16, 9600, 1, 2, 16, 1, 64, 0, 0},
        {"+ side streams", 10, 9600, 1, 2, 16, 1, 64, 1, 0},
        {"+ side streams", 16, 9600, 1, 2, 16, 1, 64, 1, 0},
        {"kernel-like, 8 pairs walked", 10, 11392, 8, 2, 16, 1, 64, 1, 0},
        {"kernel-like, 8 pairs walked", 12, 11392, 8, 2, 16, 1, 64, 1, 0},
        {"kernel-like, 8 pairs walked", 14, 11392, 8, 2, 16, 1, 64, 1, 0},
        {"kernel-like LDS 9.6 KB, 8 pairs", 16, 9600, 8, 2, 16, 1, 64, 1, 0},
        {"dma only, 8 pairs walked", 10, 8192, 8, 0, 0, 0, 0, 0, 0},
        {"dma only, 8 pairs walked", 16, 8192, 8, 0, 0, 0, 0, 0, 0},
        {"dma only, 8 pairs, ONE chunk in flight", 10, 8192, 8, 0, 0, 0, 0, 0, 1},
        {"dma only, 8 pairs, ONE chunk in flight", 16, 8192, 8, 0, 0, 0, 0, 0, 1},
        {"dma only, 8 pairs, ONE chunk in flight", 20, 8192, 8, 0, 0, 0, 0, 0, 1},
        {"dma only, 1 region, ONE chunk in flight", 16, 8192, 1, 0, 0, 0, 0, 0, 1},
        {"dma only, 1 region, ONE chunk in flight", 20, 8192, 1, 0, 0, 0, 0, 0, 1},
        {"kernel-like, 8 pairs, ONE chunk in flight", 12, 9600, 8, 2, 16, 1, 64, 1, 1},
        {"kernel-like, 8 pairs, ONE chunk in flight", 16, 9600, 8, 2, 16, 1, 64, 1, 1},
        {"kernel-like, 1 region, ONE chunk in flight", 16, 9600, 1, 2, 16, 1, 64, 1, 1},
    };
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (const Cfg& c : cfgs) {
        P p{};
        p.kv = kv; p.qstream = qs; p.ostream = os; p.idx = idx; p.sink = sink; p.cyc = cyc; p.nq = nq;
        p.pairs = c.pairs; p.reads = c.reads; p.bcast = c.bcast; p.rt = c.rt; p.mfma = c.mfma; p.side = c.side; p.serial = c.serial;
        const int blocks = 256 * c.per_cu;
        p.items_per_pair = 2704 * 8 * 8 / blocks / c.pairs;
        const double items = (double)blocks * p.items_per_pair * c.pairs;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<blocks, 64, c.lds>>>(p);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) probe<<<blocks, 64, c.lds>>>(p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        static unsigned long long h[8192];
        hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < blocks; ++i) mx = h[i] > mx ? h[i] : mx;
        const double us_launch = ms * 1e3 * 173056.0 / items;
        const double bpc = (double)(p.items_per_pair * c.pairs - 1) * 16384.0 * c.per_cu / (double)mx;
        printf("%-44s %2d waves/CU lds %5d: %7.1f us per 173056 items  %5.2f TB/s  %5.1f B/clk/CU (memtime; %.2f GHz-equiv)\n", c.what, c.per_cu,
               c.lds, us_launch, items * 16384 / ms / 1e9, bpc, (double)mx / (ms * 1e6) * 1.0);
        fflush(stdout);
        if (hipGetLastError() != hipSuccess) { printf("error\n"); return 1; }
    }
    return 0;
}
