// Finest QTAttB level with LOADER-WAVE SPECIALISATION (round 5): the same items, layout and arithmetic as fine_quad_kernel<1, false, true>
// (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:180-229 with lists of exactly 64 candidates and no top-k:
// the finest level of every shipped config) -- results bit-equal to that kernel's -- but the waves of a workgroup take two roles.
//
// Why.  In fine_quad.hip every wave issues its own LDS-DMA gathers and then works through an item's serial instruction stream (LDS reads,
// 64 MFMAs, softmax through LDS); while it does, it has nothing in flight.  Counters and probes: on average 8 KB of
// reads are in flight per CU out of the 80 KB ten waves could have; the gather alone runs at 105-127 us per launch (the texture-address
// path's ~19 cycles per 1 KB instruction), the kernel at 190-204; a consumer that never waits needs ~2800 cycles per item.  More waves
// are not available (the eight (pair, head) slices an XCD walks stop fitting its L2 when too many gathers queue up), so: few waves
// that only GATHER, with their queue always full, and waves that only COMPUTE.
//
// Workgroup = 1 loader wave + 2 consumer waves; 4 workgroups per CU (12 waves, 8 of them consumers).  Per consumer in LDS: a K ring
// (64 rows x 128 B = the candidate rows of one item), a V ring of the same size, the probability buffer, a double-buffered 704-byte
// staging area (queries 512 B, parent list 64 B, final[parent] row 128 B) and six 4-byte counters.
//
// Hand-shakes without a single blocking wait in the loader: the loader never waits for its own DMAs.  Behind the eight gather
// instructions of a K (or V) set, and behind a staging instruction, it issues ONE more LDS-DMA instruction with a single active lane
// that copies a word known to be >= 0 (the first index of the item's own parent list: a line the staging instruction has just
// fetched) over the consumer's `klanded` / `vlanded` / `slanded` stamp word, which holds -1.  A wave's vector-memory operations return in order (that
// is what vmcnt counts), so when the stamp is >= 0 the rows are in LDS: the hardware publishes the landing.  The reader of a stamp
// resets it to -1 before it frees the ring.  In the other direction the consumer bumps `kfreed` / `vfreed` / `sfreed` with plain LDS
// stores once its reads have completed (lgkmcnt(0)).  The loader walks its two consumers round robin and issues whatever is allowed:
//   staging of item s      : s <= (item being gathered) + 1, and sfreed >= s - 1 (the buffer's previous user is done with it)
//   K rows of item i       : its staging stamp is set (the parent list is there) and kfreed >= i (the K ring has been read)
//   V rows of item i       : after its K rows, and vfreed >= i
// so the K rows of item i + 1 are on their way while item i is still in its softmax, its V rows while item i + 1 runs its K pass.
// Every spin is bounded: a protocol error would end the kernel with wrong results and a raised flag, not hang the device.
// The counters are read and written with explicit DS instructions: through a generic pointer the compiler emits FLAT loads, and the
// s_waitcnt vmcnt(0) it puts behind each one made every poll wait for all of the wave's gathers (first version, commit fe1e2be:
// 1 loader + 3 consumers, landings published after an in-order `s_waitcnt vmcnt` retire through a FIFO of group codes, 1.2 ms per
// launch; this version with FLAT polls 0.5 ms).
//
// Result: bit-equal to fine_quad_kernel<1, false, true>, 270-280 us per launch against its 220-240 on the same
// boxes.  A K or V ring is reserved from the moment its gathers are issued: ~5000 cycles of flight for ~600 cycles of use, the same
// LDS bytes x time per item as fine_quad's, and the consumers' own ~2800 issue cycles per item are what a SIMD can do anyway (four
// SIMDs x 1 item / 2800 cycles = 197 us: the shipped kernel sits on that bound, not on the gather).  Kept as a measurement
// variant: CASMTR_FQ_VARIANT=lw, tools/fq_lw.py.
#include <stdio.h>
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct FineLwArgs {
    const float* q;          // [B,H,Lq0,4,32]
    const float* key;        // [B,H,Lq1,4,32]
    const float* value;      // [B,H,Lq1,4,32]
    const int32_t* parents;  // [B,H,Lq0,16]
    const float* acc_in;     // nullable [B,Lq0,H*32]
    float* message;          // nullable [B,L,H*32]
    float* acc_out;          // nullable [B,L,H*32]
    int* err;                // nullable: set to 1 when a bounded spin ran out
    unsigned long long* dbg; // nullable (CASMTR_LW_DEBUG): [0] consumer cycles waiting for K, [1] for V, [3] total, [4] items; [6] loader idle, [7] loader total
    float temp, w_level;
    int B, h0, w0, H, nquads, lq1;
};

#define LW_SPIN_MAX (1 << 22)

__device__ __forceinline__ const char* lw_uniform_ptr(const char* p) {   // pins a wave-uniform address to an SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// LDS words by byte address with explicit DS instructions (through a generic pointer the compiler emits FLAT loads, whose s_waitcnt
// vmcnt(0) would make every poll wait for all of the wave's gathers)
__device__ __forceinline__ int lw_ld(unsigned addr) {            // wave-uniform result
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lw_st(unsigned addr, int val) {   // one lane stores
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(keep) : "v"(addr), "v"(val) : "memory");
}

constexpr int LW_PST = 36, LW_P_FLOATS = 8 * LW_PST, LW_KS = 68;
constexpr int LW_STG = 176;                         // floats per staging buffer: q 128 | parents 16 | final[parent] 32
constexpr int LW_CW = 2 * 2048 + LW_P_FLOATS + 2 * LW_STG + 8;   // floats per consumer: K ring | V ring | P | staging x 2 | counters

template <int NC, int WGS>   // consumers per workgroup; workgroups per CU
__global__ __launch_bounds__(64 * (1 + NC), WGS) void fine_lw_kernel(const FineLwArgs a) {
    constexpr int PST = LW_PST, P_FLOATS = LW_P_FLOATS, KS = LW_KS, STG = LW_STG, CW = LW_CW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, HD = H * 32, Kp = 16;
    const int L = a.h0 * a.w0, wq = a.w0 >> 1, Lq = a.nquads;
    // work list (as fine_quad): XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = (g < G && cnt > 0) ? a.B * cnt : 0;
    const int stride = (gridDim.x >> 3) * NC;
    // words of consumer c (ints at the end of its LDS block): 0 klanded 1 vlanded 2, 3 slanded per staging buffer (stamps: -1 = not landed, written by DMA) | 4 kfreed 5 vfreed 6 sfreed (counters)
    if (threadIdx.x < NC * 8) reinterpret_cast<volatile int*>(smem + (threadIdx.x >> 3) * CW + CW - 8)[threadIdx.x & 7] = (threadIdx.x & 7) < 4 ? -1 : 0;
    __syncthreads();
    auto items_of = [&](int t) { return t < total ? (total - t + stride - 1) / stride : 0; };

    if (wave == 0) {
        // ============================================================================================================ loader
        const size_t pair_pitch = (size_t)H * a.lq1 * 128;
        const float* const k0 = a.key + (size_t)h * a.lq1 * 128 - 768;     // this head's slice of pair 0, 3072 bytes low
        const float* const v0 = a.value + (size_t)h * a.lq1 * 128 - 768;
        const int un = lane & 7;
        unsigned cK[4];   // DMA source offset inside a parent's 512-byte run, instruction j of a 32-row half (rows 8 j + lane / 8; fine_quad.hip)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            cK[j] = (unsigned)(((lane >> 3) & 3) * 128 + ((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072 - j * 1024);
        // staging source per lane (one 16-byte unit each, lanes 0..43): q 512 B | parents 64 B | final[parent] 128 B
        unsigned long long sbase;
        unsigned mulq, mula;
        {
            const int u = lane < 44 ? lane : 43;
            if (u < 32) {
                const int r = u >> 3, pu = u & 7;
                sbase = (unsigned long long)a.q + (unsigned)(r * 128 + ((pu ^ (r >> 1)) * 16));
                mulq = 512u; mula = 0u;
            } else if (u < 36 || !a.acc_in) {
                sbase = (unsigned long long)a.parents + (unsigned)(((u - 32) & 3) * 16);
                mulq = (unsigned)(Kp * 4); mula = 0u;
            } else {
                sbase = (unsigned long long)a.acc_in + (unsigned)(h * 128 + (u - 36) * 16);
                mulq = 0u; mula = (unsigned)(HD * 4);
            }
        }
        const int t0 = (blockIdx.x >> 3) * NC;
        int T[NC], it[NC], s[NC], ph[NC];           // items; item being gathered; stagings issued; 0: offsets missing, 1: K next, 2: V next
        int scb[NC], scq[NC], ccb[NC], ccq[NC];     // cursors (pair, index in the XCD's chunk) of the next staging / of the item being gathered
        unsigned voff[NC][8];
        unsigned kring[NC], stgl[NC], ctrl[NC];     // LDS byte addresses
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            T[c] = items_of(t0 + c); it[c] = 0; s[c] = 0; ph[c] = 0;
            scb[c] = ccb[c] = (t0 + c) / max(cnt, 1); scq[c] = ccq[c] = (t0 + c) % max(cnt, 1);
            kring[c] = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem + c * CW));
            stgl[c] = kring[c] + (unsigned)((4096 + P_FLOATS) * 4);
            ctrl[c] = kring[c] + (unsigned)((CW - 8) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) voff[c][j] = 0;
        }
        const bool hh = lane >> 5;
        // one LDS-DMA instruction with ONE active lane: the first word of the item's parent list (an index >= 0) -> the stamp word at LDS byte
        // address `dst`, which held -1; it lands behind everything this wave issued before
        const char* const parb = lw_uniform_ptr(reinterpret_cast<const char*>(a.parents));
        auto stamp = [&](unsigned qd, unsigned dst) {
            const unsigned off = qd * (unsigned)(Kp * 4);
            const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
            unsigned long long keep;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b64 exec, %0"
                         : "=&s"(keep) : "v"(off), "s"(parb), "s"(d) : "memory");
        };
        unsigned cqd[NC];   // (pair, head, quad) index of the item being gathered
        int spins = 0, outst = 0;
        unsigned long long l_idle = 0, l_blk = 0, l_iss = 0;
        const unsigned long long l_begin = __builtin_readcyclecounter();
        for (;;) {
            bool all_done = true, issued = false;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const unsigned cc = ctrl[c];
                // ---- the front end of item s: at most one item ahead of the gathers, its buffer (item s - 2's) released
                if (s[c] < T[c] && s[c] <= it[c] + 1 && (s[c] < 2 || lw_ld(cc + 24u) >= s[c] - 1)) {
                    const unsigned quad = (unsigned)(g * chunk + scq[c]);
                    const unsigned qd = (unsigned)((scb[c] * H + h) * Lq) + quad, bq = (unsigned)(scb[c] * Lq) + quad;
                    const unsigned long long addr = sbase + (unsigned long long)qd * mulq + (unsigned long long)bq * mula;
                    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(stgl[c] + (unsigned)((s[c] & 1) * STG * 4)));
                    {
                        unsigned long long keep;   // lanes 0..43
                        asm volatile("s_mov_b64 %0, exec\n\ts_bfm_b64 exec, 44, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
                                     : "=&s"(keep) : "v"(addr), "s"(dst) : "memory");
                    }
                    stamp(qd, ctrl[c] + 8u + (unsigned)((s[c] & 1) * 4));
                    ++s[c];
                    outst += 2; issued = true;
                    scq[c] += stride;
                    while (scq[c] >= cnt) { scq[c] -= cnt; ++scb[c]; }
                }
                if (it[c] < T[c]) {
                    all_done = false;
                    // ---- the parent list of item it -> the eight DMA offsets (K and V rows share them)
                    if (ph[c] == 0 && lw_ld(cc + 8u + (unsigned)((it[c] & 1) * 4)) >= 0) {
                        asm volatile("" ::: "memory");
                        const int* t2 = reinterpret_cast<const int*>(smem + c * CW + 4096 + P_FLOATS + (it[c] & 1) * STG + 128);
#pragma unroll
                        for (int hc = 0; hc < 2; ++hc) {
                            const int4 pa4 = *reinterpret_cast<const int4*>(t2 + 8 * hc), pb4 = *reinterpret_cast<const int4*>(t2 + 8 * hc + 4);
                            voff[c][4 * hc + 0] = ((unsigned)(hh ? pa4.y : pa4.x) << 9) + cK[0];
                            voff[c][4 * hc + 1] = ((unsigned)(hh ? pa4.w : pa4.z) << 9) + cK[1];
                            voff[c][4 * hc + 2] = ((unsigned)(hh ? pb4.y : pb4.x) << 9) + cK[2];
                            voff[c][4 * hc + 3] = ((unsigned)(hh ? pb4.w : pb4.z) << 9) + cK[3];
                        }
                        cqd[c] = (unsigned)((ccb[c] * H + h) * Lq + g * chunk + ccq[c]);
                        lw_st(cc + 8u + (unsigned)((it[c] & 1) * 4), -1);   // consumed: the next stamp of this word (item it + 2's) is issued later
                        ph[c] = 1;
                    }
                    // ---- K rows (ring read by the consumer up to item it - 1), then V rows
                    if ((ph[c] == 1 && lw_ld(cc + 16u) >= it[c]) || (ph[c] == 2 && lw_ld(cc + 20u) >= it[c])) {
                        const bool isv = ph[c] == 2;
                        const float* base = reinterpret_cast<const float*>(lw_uniform_ptr(reinterpret_cast<const char*>((isv ? v0 : k0) + (size_t)ccb[c] * pair_pitch)));
                        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(kring[c] + (isv ? 8192u : 0u)));
                        const unsigned long long t0_ = __builtin_readcyclecounter();
                        glds_chunk(base, voff[c][0], voff[c][1], voff[c][2], voff[c][3], dst);
                        glds_chunk(base, voff[c][4], voff[c][5], voff[c][6], voff[c][7], dst + 4096u);
                        stamp(cqd[c], ctrl[c] + (isv ? 4u : 0u));   // klanded / vlanded
                        l_iss += __builtin_readcyclecounter() - t0_;
                        outst += 9; issued = true;
                        if (isv) {
                            ph[c] = 0; ++it[c];
                            ccq[c] += stride;
                            while (ccq[c] >= cnt) { ccq[c] -= cnt; ++ccb[c]; }
                        } else ph[c] = 2;
                    }
                }
            }
            if (all_done) break;
            if (outst > 36) {   // vmcnt is a 6-bit counter
                const unsigned long long t0_ = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                l_blk += __builtin_readcyclecounter() - t0_;
                outst = 24;
            }
            if (issued) spins = 0;
            else {
                const unsigned long long t0_ = __builtin_readcyclecounter();
                __builtin_amdgcn_s_sleep(1);
                l_idle += __builtin_readcyclecounter() - t0_;
                if (++spins > LW_SPIN_MAX) { if (a.err && lane == 0) *a.err = 1; break; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.dbg && lane == 0) { atomicAdd(a.dbg + 5, l_blk); atomicAdd(a.dbg + 2, l_iss); atomicAdd(a.dbg + 6, l_idle); atomicAdd(a.dbg + 7, (unsigned long long)(__builtin_readcyclecounter() - l_begin)); }
        return;
    }

    // ================================================================================================================ consumer
    const int c = wave - 1;
    const int t = (blockIdx.x >> 3) * NC + c;
    const int T = items_of(t);
    if (T == 0) return;
    float* ring = smem + c * CW;        // K rows: 64 x 128 B (XOR-swizzled 16-byte units); V rows behind them
    float* Pld = ring + 4096;
    float* stg = Pld + P_FLOATS;
    const unsigned my = lds_byte_addr(smem + c * CW + CW - 8);
    unsigned va[8];    // V chunk: byte offset of V[row 2 mm + lane/32][d = lane%32] for mm % 8 == x, minus mm * 256
#pragma unroll
    for (int x = 0; x < 8; ++x) va[x] = (unsigned)((lane >> 5) * 128 + ((((lane & 31) >> 2) ^ x) * 16) + (lane & 3) * 4);
    const float* pa = Pld + ((lane & 3) * 2 + (lane >> 5)) * PST;   // operand A of the V chunks: P[child lane%4][parity lane/32][.]
    const char* kb = reinterpret_cast<const char*>(ring) + lane * 128;
    int cb = t / cnt, cq = t % cnt, cy = (g * chunk + cq) / wq, cx = (g * chunk + cq) % wq;
    const int sy = stride / wq, sx = stride % wq;
    bool failed = false;
    unsigned long long w_acc[2] = {0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
    auto wait_for = [&](int word, int which) {
        const unsigned long long t0_ = __builtin_readcyclecounter();
        int spins = 0;
        while (lw_ld(my + (unsigned)word * 4u) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > LW_SPIN_MAX) { failed = true; break; }
        }
        asm volatile("" ::: "memory");
        w_acc[which] += __builtin_readcyclecounter() - t0_;
    };
    for (int i = 0; i < T && !failed; ++i) {
        const int b = cb, l00 = 2 * cy * a.w0 + 2 * cx;
        const float* qs = stg + (i & 1) * STG;
        // ---- K pass (the K stamp lies behind the item's staging instruction: queries and final[parent] are there too)
        wait_for(0, 0);
        const float acc_cur = a.acc_in ? qs[144 + (lane & 31)] : 0.f;   // final[parent] of the item for d = lane % 32 (:277)
        f32x4 qa[8], kr[8];   // operand A: lane l holds q[child l%4][d]; operand B: this lane's candidate row
#pragma unroll
        for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qs + (lane & 3) * 32 + ((u ^ ((lane & 3) >> 1)) * 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kb + ((u ^ ((lane >> 1) & 7)) * 16));
        lds_reads_done();
        { lw_st(my, -1); lw_st(my + 16u, i + 1); }   // the K ring is free (LDS operations of a wave execute in order)
        f32x4 c4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c4[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // no index depends on these logits: four interleaved partial d-chains
            c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c4[0], 0, 0, 0);
            c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c4[1], 0, 0, 0);
            c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c4[2], 0, 0, 0);
            c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c4[3], 0, 0, 0);
        }
        // ---- softmax, one series (child) per 16-lane row (fine_quad.hip: softmax_select without the selection)
        {
            const int f = lane >> 4, j = lane & 15;
#pragma unroll
            for (int ff = 0; ff < 4; ++ff) Pld[ff * KS + lane] = a.temp * ((c4[0][ff] + c4[1][ff]) + (c4[2][ff] + c4[3][ff]));
            wave_lds_fence();
            const f32x4 v = *reinterpret_cast<const f32x4*>(Pld + f * KS + j * 4);
            float fm = -3.0e38f;
            fm = fmaxf(fm, v.x); fm = fmaxf(fm, v.y); fm = fmaxf(fm, v.z); fm = fmaxf(fm, v.w);
            fm = row16_max_f32(fm);
            float ps[4] = {__expf(v.x - fm), __expf(v.y - fm), __expf(v.z - fm), __expf(v.w - fm)};
            float sum = 0.f;   // (the same order of additions as fine_quad.hip: the two kernels' results are bit-equal)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += ps[e];
            const float rinv = __builtin_amdgcn_rcpf(row16_sum_f32(sum));
            wave_lds_fence();   // every lane has its logits: the buffer becomes P (candidate 4 j + e -> P[f][e & 1][2 j + e / 2])
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2*>(Pld + (f * 2 + 0) * PST + 2 * j) = (f32x2){ps[0] * rinv, ps[2] * rinv};
            *reinterpret_cast<f32x2*>(Pld + (f * 2 + 1) * PST + 2 * j) = (f32x2){ps[1] * rinv, ps[3] * rinv};
            wave_lds_fence();
        }
        lw_st(my + 24u, i + 1);   // this item's staging area is no longer needed
        // ---- V rows
        wait_for(1, 1);
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const char* sb = reinterpret_cast<const char*>(ring) + 8192 + cc * 4096;
            f32x4 pv[4];   // operand A of MFMA mm: P[child lane%4][parity lane/32][16 cc + mm]
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const f32x4*>(pa + 16 * cc + 4 * k);
            float vb[16];
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) vb[mm] = *reinterpret_cast<const float*>(sb + va[mm & 7] + mm * 256);
            lds_reads_done();
            if (cc == 1) { lw_st(my + 4u, -1); lw_st(my + 20u, i + 1); }   // the V ring is free
#pragma unroll
            for (int mm = 0; mm < 16; ++mm)
                acc[mm & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(pv[mm >> 2][mm & 3], vb[mm], acc[mm & 3], 0, 0, 0);
        }
        {
            f32x4 tot;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = (acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k]);
                const unsigned xi = __float_as_uint(x);
                const auto sw = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // lanes l and l ^ 32
                tot[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            const int hi = lane >> 5;
            const float vA = hi ? tot[2] : tot[0], vB = hi ? tot[3] : tot[1];
            const size_t o = ((size_t)b * L + l00 + hi * a.w0) * HD + h * 32 + (lane & 31);
            if (a.message) { a.message[o] = vA; a.message[o + HD] = vB; }
            if (a.acc_out) {   // separate multiply and add (:277-281)
                a.acc_out[o] = acc_cur + vA * a.w_level;
                a.acc_out[o + HD] = acc_cur + vB * a.w_level;
            }
        }
        // next item of this consumer
        cq += stride;
        if (cq >= cnt) {
            while (cq >= cnt) { cq -= cnt; ++cb; }
            cy = (g * chunk + cq) / wq; cx = (g * chunk + cq) % wq;
        } else {
            cy += sy; cx += sx;
            if (cx >= wq) { cx -= wq; ++cy; }
        }
    }
    if (failed && a.err && lane == 0) *a.err = 1;
    if (a.dbg && lane == 0) {
        atomicAdd(a.dbg + 0, w_acc[0]); atomicAdd(a.dbg + 1, w_acc[1]);
        atomicAdd(a.dbg + 3, (unsigned long long)(__builtin_readcyclecounter() - t_begin)); atomicAdd(a.dbg + 4, (unsigned long long)T);
    }
}

int casmtr_qta_fine_level_lw(const float* q, const float* key, const float* value, const int32_t* parents, float temp, float w_level,
                             const float* acc_in, float* message, float* acc_out, int B, int h0, int w0, int h1, int w1, int H, int Kp,
                             hipStream_t s) {
    if (Kp != 16 || (H != 8 && H != 4 && H != 2 && H != 1) || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1)) return CASMTR_ERR_UNSUPPORTED;
    const long long lq0 = (long long)(h0 / 2) * (w0 / 2), lq1 = (long long)(h1 / 2) * (w1 / 2);
    if (lq1 >= (1 << 22) || (long long)B * H * lq0 * 512 >= (1ll << 32)) return CASMTR_ERR_UNSUPPORTED;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CASMTR_MAX_DEVICES) return CASMTR_ERR_UNSUPPORTED;
    FineLwArgs a{};
    a.q = q; a.key = key; a.value = value; a.parents = parents; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out;
    a.err = nullptr; a.dbg = nullptr;
    a.temp = temp; a.w_level = w_level; a.B = B; a.h0 = h0; a.w0 = w0; a.H = H; a.nquads = (int)lq0; a.lq1 = (int)lq1;
    const char* en = getenv("CASMTR_LW_NC");   // measurement knob: consumers per loader
    const int LW_NC = en && en[0] == '1' ? 1 : en && en[0] == '3' ? 3 : 2;
    const size_t lds = sizeof(float) * LW_NC * LW_CW;
    static int resident[3][CASMTR_MAX_DEVICES] = {{0}};
    int res = 0;
    auto kern = LW_NC == 1 ? fine_lw_kernel<1, 8> : LW_NC == 3 ? fine_lw_kernel<3, 2> : fine_lw_kernel<2, 4>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (const int r = resident_workgroups(resident[LW_NC - 1], kern, 64 * (1 + LW_NC), lds, &res)) return r;
    long long blocks = res;
    const char* ev = getenv("CASMTR_LW_BLOCKS");   // measurement knob: workgroups in the persistent grid (multiple of 8)
    if (ev && atoi(ev) > 0 && atoi(ev) < blocks) blocks = atoi(ev) / 8 * 8;
    const int G = 8 / H;
    const long long per_xcd = (long long)B * ((lq0 + G - 1) / G);
    if (blocks / 8 * LW_NC > per_xcd) blocks = (per_xcd + LW_NC - 1) / LW_NC * 8;
    if (blocks < 8) blocks = 8;
    static unsigned long long* dbg[CASMTR_MAX_DEVICES] = {nullptr};
    if (getenv("CASMTR_LW_DEBUG")) {
        if (!dbg[dev]) (void)hipMalloc(&dbg[dev], 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg[dev], 0, 8 * sizeof(unsigned long long), s);
        a.dbg = dbg[dev];
    }
    prof_symbol_args(CASMTR_PROF_QTA_FINE, "%s", "");
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_QTA_FINE, kern, dim3((unsigned)blocks), dim3(64 * (1 + LW_NC)), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    if (a.dbg) {
        unsigned long long hdbg[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hdbg, a.dbg, sizeof hdbg, hipMemcpyDeviceToHost);
        const double it = (double)(hdbg[4] ? hdbg[4] : 1);
        fprintf(stderr, "fine_lw: %lld workgroups (%d resident); per item: wait K %.0f, V %.0f of %.0f cycles; loader idle %.0f %%, in vmcnt %.0f %%, issuing 9-instruction groups %.0f %% of %.0f cycles\n", blocks, res,
                hdbg[0] / it, hdbg[1] / it, hdbg[3] / it, 100.0 * hdbg[6] / (double)(hdbg[7] ? hdbg[7] : 1), 100.0 * hdbg[5] / (double)(hdbg[7] ? hdbg[7] : 1),
                100.0 * hdbg[2] / (double)(hdbg[7] ? hdbg[7] : 1), hdbg[7] / (double)blocks);
    }
    return 0;
}
