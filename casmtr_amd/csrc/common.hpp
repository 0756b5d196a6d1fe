// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>

#define CASMTR_WAVE 64
#define NEG_FILL (-1e9f)  // INF = 1e9 in coarse_matching.py:6 / cascade_matching.py:8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace casmtr {

// Order-preserving map float -> uint32 (total order of the finite floats; -0 < +0, irrelevant for us).
__device__ __forceinline__ unsigned f2ord(float f) {   // branch-free: 3 VALU (the ?: form compiles to cmp + not + or + cndmask)
    const unsigned u = __float_as_uint(f);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);   // negative: ~u, otherwise u | 0x80000000
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned t = (unsigned)((int)o >> 31) & 0x7fffffffu;   // top bit set (was >= +0): flip it only; clear: ~o
    return __uint_as_float(~(o ^ t));
}

// DPP controls (gfx9): quad_perm = 0x00..0xFF, row_shr:n = 0x110+n, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}

// Same, but lanes whose source falls outside the row read 0 (bound_ctrl): no `old` operand to materialise.
template <int CTRL>
__device__ __forceinline__ float dpp_zfill_f32(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}

// All-lanes max / sum over each 16-lane DPP row (4 independent rows per wave).
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v));  // row_half_mirror
    v = max(v, dpp_u32<0x140>(v));  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum_f32(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}

// Whole-wave (64 lane) reductions, wave-uniform result (SGPR): DPP inside each 16-lane row, then one v_readlane per
// row and scalar combines -- no ds_bpermute / LDS round trip on the dependent chain.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = row16_max_u32(v);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = row16_sum_f32(v);
    const float a = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    const float b = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    const float c = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    const float d = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48));
    return (a + b) + (c + d);
}

// Softmax over a quad's window candidates (CascadeMatching.forward, cascade_matching.py:141), two candidates per lane: probabilities
// from the hardware exponential (v_exp_f32 of (x - m) log2 e; x - m <= 0) and ONE exact division for 1 / sum -- within 2 ulp of
// expf(x - m) / sum.  The probabilities carry the 1e-4 softmax tolerance and no index depends on them (the argmax is taken from the
// logits); the three window-matching kernels share this function so that their outputs stay bit-equal to each other.
__device__ __forceinline__ void window_softmax2(float x0, float x1, float m, bool v0, bool v1, float& e0, float& e1) {
    e0 = v0 ? __expf(x0 - m) : 0.f;
    e1 = v1 ? __expf(x1 - m) : 0.f;
    const float inv = 1.0f / wave_sum_f32(e0 + e1);
    e0 *= inv;
    e1 *= inv;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    return ord2f(wave_max_u32(f2ord(v)));
}

// torch CPU: true fp32 division; torch GPU kernels: x * fl32(1/s)  (see oracle/casmtr_oracle.c header).
// Compile-time switch: with a runtime flag hipcc emits the ~12-instruction IEEE division sequence at every call site
// (behind a branch), which blew the matching kernels up to 40-65 KB of code -- more than the instruction cache.
template <bool RECIP>
__device__ __forceinline__ float div_scalar(float x, float s, float inv_s) {
    return RECIP ? __fmul_rn(x, inv_s) : __fdiv_rn(x, s);
}

// Wave-private transposition through LDS: 64 rows x 32 floats (128 B = one cache line each).  Global side: 8 lanes
// share a row (one dwordx4 each), so every load instruction covers 8 whole lines (coalesced; a lane-per-row walk
// touches 64 lines per instruction and runs at a quarter of the L1/TA rate).  LDS side: rows padded to 36 floats ->
// the ds_write_b128 (8-lane groups = one row) and the ds_read_b128 (lane l reads row l) are both conflict-free.
// row_ptr(r) returns the global address of row r (0..63) and must be valid for every r.
#define CASMTR_SLAB_FLOATS (64 * 36)
template <typename RowPtr>
__device__ __forceinline__ void wave_rows32_to_lanes(float* slab, int lane, RowPtr row_ptr, f32x4 (&out)[8]) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(row_ptr(8 * j + (lane >> 3)) + (lane & 7) * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(slab + (8 * j + (lane >> 3)) * 36 + (lane & 7) * 4) = v[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = *reinterpret_cast<const f32x4*>(slab + lane * 36 + i * 4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Bijective "XCD-contiguous" remap of a 1-D block id: the dispatcher is observed to place block b on XCD b % 8; this
// gives each XCD one contiguous chunk of the logical index space so that neighbouring tiles share an L2.  Speed only.
__device__ __forceinline__ int xcd_chunk_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Read-only, wave-uniform operands (a query row shared by the whole wave): viewing the pointer in the constant
// address space tells the compiler the data is invariant for the kernel, so it emits s_load_dwordxN (scalar cache,
// SGPR operands for v_fmac) even inside loops that also store to other buffers.  Only for data written by EARLIER kernels.
typedef const float __attribute__((address_space(4))) * cfloat_p;
__device__ __forceinline__ cfloat_p as_const(const float* p) {
    return (cfloat_p)(unsigned long long)p;
}

// ---- LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B = 1 KiB per wave-instruction straight from global memory into LDS.
// Destination: wave-uniform LDS byte address + lane * 16 (lane-linear); the SOURCE address is per lane, so any LDS-side
// swizzle is applied to the source pointer.  No VGPRs hold the data in flight.  Issued through inline asm: hipcc neither
// counts these loads in its s_waitcnt bookkeeping nor preserves M0 around the statement (guide section 5.5), so the
// caller waits with glds_wait<N>() -- N = the number of LATER loads that may still be in flight -- and keeps
// compiler-generated vector-memory instructions out of the span in which a DMA is outstanding.
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
    return (unsigned)(unsigned long long)p;   // low 32 bits of a flat pointer into the LDS aperture = the LDS offset
}
__device__ __forceinline__ void glds16(const float* base, unsigned byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void glds_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// every LDS read issued so far has returned (before a DMA may overwrite the buffer they read)
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// v of lane (l ^ J) for a compile-time J in {1, 2, 4, 8, 16, 32}, on the VALU (DPP quad permutes / row shifts / row rotation,
// gfx950 row and half swaps) instead of __shfl_xor's ds_bpermute: a 64-lane bitonic sort is 21 DEPENDENT exchange steps, and each
// ds_bpermute round trip costs ~70 cycles of LDS latency on that chain.
template <int J>
__device__ __forceinline__ unsigned wave_xor_u32(unsigned v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "power of two below the wave size");
    if constexpr (J == 1) return dpp_u32<0xB1>(v);         // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_u32<0x4E>(v);    // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
        // banks = the four quads of a 16-lane row: quads 0, 2 read 4 lanes up (row_shl:4), quads 1, 3 read 4 lanes down (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
    } else if constexpr (J == 8) return dpp_u32<0x128>(v);  // row_ror:8
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0] = halves [lo,lo], r[1] = [hi,hi]
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E).  `#pragma unroll` leaves long stage loops rolled
// (then every per-stage constant -- buffer parity, vmcnt immediates, register-array indices -- becomes a runtime value)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// sum over the lanes {l, l^8, l^16, l^32} (the 8 "slices" of a wave whose lane index is slice*8 + unit), result in every lane:
// xor 8 by DPP row rotation, xor 16 / 32 by the gfx950 row / half swaps -- VALU only, no ds_bpermute round trips
__device__ __forceinline__ float sum_over_slices8(float x) {
    x += dpp_f32<0x128>(x);   // row_ror:8
    const unsigned a = __float_as_uint(x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const unsigned b = __float_as_uint(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}

// prof.hip.  Work counters of the persistent gather kernels (dynamic item claiming): -> one slot of WORK_SLOT_INTS zeroed ints on the
// current device: XCD x's claim counter at [x * WORK_XCD_INTS], its exit counter WORK_EXIT_OFF ints behind it.  EVERY COUNTER IN ITS OWN
// 128-BYTE LINE: a line is owned by the L2 of the XCD whose waves use it and the atomics run at L2 speed; with the eight XCDs' counters
// in one line the line ping-pongs between the eight L2s and each claim costs ~90 ns (cascade_quad_kernel: 2.0 ms instead of 0.6,
// profiles/r06_cq_fields.txt).  The kernels leave the counters zero again (the last wave of an XCD to exit resets its pair), so a slot
// needs no clearing launch; slots rotate (64 per device) so that launches in flight on different streams do not share one.  nullptr
// when unavailable (allocation failure, a stream capture in progress at first use): callers then use their static schedule.
#define WORK_XCD_INTS 64
#define WORK_EXIT_OFF 32
#define WORK_SLOT_INTS (8 * WORK_XCD_INTS)
int* work_counters(hipStream_t stream);
// Claim (whole wave active, wave-uniform `doit`): lane 0 adds 1 to the counter, the pre-increment value arrives in `ret` ASYNCHRONOUSLY,
// like a load: it is valid once a LATER s_waitcnt vmcnt(N) has passed with N <= the number of vector-memory operations issued behind
// the claim (vmcnt retires in order); read it with work_claimed(ret) behind such a wait.  doit == false: the instruction runs with
// EXEC = 0 and does nothing, so the statement is unconditional for the compiler.  Inline asm on purpose:
//   (1) the compiler's atomic optimiser turns a uniform atomicAdd into "atomic, s_waitcnt vmcnt(0), v_readfirstlane" on the spot, which
//       exposes the round trip and drains every LDS-DMA in flight;
//   (2) to the compiler an asm output is valid at once.  `ret` must therefore be defined by this statement and consumed by
//       work_claimed() in the SAME loop iteration with nothing else reading it: a loop-carried or conditionally defined variable gets
//       copied (phi moves) a few instructions behind the atomic, before the data has arrived (seen in the first version; AGPRs as a
//       mailbox make the allocator split the whole kernel's registers into VGPR + AGPR halves with ~900 copies).
//       tools/check_quad_isa.py checks that no instruction touches the register between the two statements.
__device__ __forceinline__ void work_claim_issue(int* ctr, bool doit, int& ret) {
    const int m = __builtin_amdgcn_readfirstlane(doit ? 1 : 0);   // (the compiler cannot always prove `doit` uniform: pin it to an SGPR)
    unsigned long long save;
    const int zero = 0, one = 1;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %3, %4, %5 sc0\n\ts_mov_b64 exec, %1"
                 : "=v"(ret), "=&s"(save) : "s"(m), "v"(zero), "v"(one), "s"(ctr) : "memory");
}
__device__ __forceinline__ int work_claimed(const int& ret) {   // lane 0's value, wave-uniform
    int s;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(ret));
    return s;
}
__device__ __forceinline__ void work_leave(int* ctr, int nwaves) {   // one lane per wave; the last of the XCD's nwaves waves re-zeroes the pair
    if (atomicAdd(ctr + WORK_EXIT_OFF, 1) == nwaves - 1) { atomicExch(ctr, 0); atomicExch(ctr + WORK_EXIT_OFF, 0); }
}
extern int g_debug_flags;   // casmtr_debug_set(): phase-elimination switches for timing experiments (results become garbage)
#define CASMTR_DBG_NO_DMA 1      // DMA kernels: do not issue / wait for the key and value row transfers
#define CASMTR_DBG_NO_MATH 2     // DMA kernels: skip the per-stage LDS reads and arithmetic
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
// Single-kernel scope with the two events ATTACHED TO THE DISPATCH (hipExtLaunchKernelGGL): they carry the kernel's own start / end
// timestamps and put no barrier packet on the stream.  (Events recorded around a launch cost 5.8 us of idle stream each: rocprofv3
// kernel trace of the bench step, 24 records per step.)  prof_pair() -> false when kernel `id` is not being timed.
bool prof_pair(int id, hipEvent_t* a, hipEvent_t* b);
// Which kernel ran under scope `id` last (casmtr_prof_symbol): the instantiated symbol of a single-kernel scope, a literal list for
// the multi-kernel scopes.  `sym` must have static storage duration.
void prof_symbol(int id, const char* sym);
// template arguments of the instantiation about to be launched under scope `id` (printf-style, e.g. "<%d,%d>"): replaces the
// "<...>" of the stringified kernel expression CASMTR_LAUNCH_TIMED records.  No-op unless `id` is being timed.
void prof_symbol_args(int id, const char* fmt, ...);
#define CASMTR_LAUNCH_TIMED(id, kernel, grid, block, lds, stream, ...)                                          \
    do {                                                                                                       \
        hipEvent_t ea__, eb__;                                                                                 \
        if (casmtr::prof_pair(id, &ea__, &eb__)) {                                                             \
            casmtr::prof_symbol(id, #kernel);                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ea__, eb__, 0, __VA_ARGS__);               \
        } else                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(int id_, hipStream_t s_, const char* sym = nullptr) : id(id_), s(s_) {
        prof_begin(id, s);
        if (sym) prof_symbol(id, sym);
    }
    ~ProfScope() { prof_end(id, s); }
};

}  // namespace casmtr

#define CASMTR_CHECK_LAUNCH()                         \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

#define CASMTR_ERR_UNSUPPORTED 1001  // shape outside what the kernel was built for (caller must not fall back to CPU)

// Size of a persistent grid: the workgroups of `kernel` that are resident at once on the CURRENT device (CU count x occupancy, a
// multiple of 8 so that every XCD gets the same number).  Cached per device id -- a process may drive several parts, or partition
// modes with different CU counts -- with relaxed atomics (concurrent first calls compute the same value).
#define CASMTR_MAX_DEVICES 64
template <typename KernelT>
static inline int resident_workgroups(int* cache /* [CASMTR_MAX_DEVICES], zero-initialised */, KernelT kernel, int threads, size_t lds,
                                      int* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= CASMTR_MAX_DEVICES) return CASMTR_ERR_UNSUPPORTED;
    int r = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
    if (!r) {
        int ncu = 0, per_cu = 0;
        e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds);
        if (e != hipSuccess) return (int)e;
        if (ncu <= 0 || per_cu <= 0) return CASMTR_ERR_UNSUPPORTED;
        r = ncu * per_cu / 8 * 8;
        if (r <= 0) return CASMTR_ERR_UNSUPPORTED;
        __atomic_store_n(&cache[dev], r, __ATOMIC_RELAXED);
    }
    *out = r;
    return 0;
}
