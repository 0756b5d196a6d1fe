// Finest QTAttB level with LOADER-WAVE SPECIALISATION (round 5): the same items, layout and arithmetic as fine_quad_kernel<1, false, true>
// (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:180-229 with lists of exactly 64 candidates and no top-k:
// the finest level of every shipped config), but the waves of a workgroup take two roles.
//
// Why.  In fine_quad.hip every wave issues its own LDS-DMA gathers and then works through an item's serial instruction stream (LDS reads,
// 64 dependent-ish MFMAs, softmax through LDS); while it does, it has nothing in flight.  Counters and probes: on
// average 8 KB of reads are in flight per CU out of the 80 KB ten waves could have; the gather alone runs at 105-127 us per launch
// (the texture-address path's ~19 cycles per 1 KB instruction), the kernel at 190-204; and more waves are not available because the
// eight (pair, head) slices an XCD walks stop fitting its L2 when more than ~3 MB of gathers are in flight.  So: keep the number of
// waves that GATHER small and their queue always full, and let the others only compute.
//
// Workgroup = 1 loader wave + 3 consumer waves, 3 workgroups per CU.  Per consumer: a ring of three 4 KB chunk slots, the probability
// buffer, a double-buffered 1 KB staging area (queries, parent list, final[parent] row) and three counters in LDS:
//   landed  (loader -> consumer)  chunks of this consumer whose DMA has landed
//   freed   (consumer -> loader)  chunks it has finished reading
//   sfreed  (consumer -> loader)  items whose staging area it no longer needs
// An item is four chunks, K0 K1 V0 V1 (8 parents x 512 B each), chunk n in slot n % 3.  The loader walks its three consumers round
// robin: stages the next item's front end (one 16-byte-wide DMA instruction), turns the staged parent list into DMA offsets, issues a
// chunk whenever its slot is free, and retires its DMA groups IN ORDER (vmcnt counts them in order): a FIFO of (owner, kind) codes in a
// 64-bit register tells whose counter the oldest group bumps.  The consumer polls `landed`, reads, bumps `freed`, computes; its global
// stores are ordinary compiler-visible stores (its own vmcnt has nothing hand-counted in it).
// Every spin is bounded: a protocol error would end the kernel with wrong results and a raised flag, not hang the device.
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
    unsigned long long* dbg; // nullable (CASMTR_LW_DEBUG): [0..3] consumer cycles waiting for K / V0 / V1 / total, [4] items; [5..7] loader: blocked in vmcnt, idle, total
    float temp, w_level;
    int B, h0, w0, H, nquads, lq1;
};

#define LW_SPIN_MAX (1 << 22)

__device__ __forceinline__ const char* uniform_ptr(const char* p) {   // pins a wave-uniform address to an SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

// wait until at most n (wave-uniform, 0..63) vector-memory operations are outstanding: s_waitcnt takes an immediate
__device__ __forceinline__ void vmwait_dyn(int n) {
#define LW_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        LW_W(0) LW_W(1) LW_W(2) LW_W(3) LW_W(4) LW_W(5) LW_W(6) LW_W(7) LW_W(8) LW_W(9) LW_W(10) LW_W(11) LW_W(12) LW_W(13) LW_W(14) LW_W(15)
        LW_W(16) LW_W(17) LW_W(18) LW_W(19) LW_W(20) LW_W(21) LW_W(22) LW_W(23) LW_W(24) LW_W(25) LW_W(26) LW_W(27) LW_W(28) LW_W(29) LW_W(30)
        LW_W(31) LW_W(32) LW_W(33) LW_W(34) LW_W(35) LW_W(36) LW_W(37) LW_W(38) LW_W(39) LW_W(40) LW_W(41) LW_W(42) LW_W(43) LW_W(44) LW_W(45)
        LW_W(46) LW_W(47) LW_W(48) LW_W(49) LW_W(50) LW_W(51) LW_W(52) LW_W(53) LW_W(54) LW_W(55) LW_W(56) LW_W(57) LW_W(58) LW_W(59) LW_W(60)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef LW_W
}

__global__ __launch_bounds__(256, 3) void fine_lw_kernel(const FineLwArgs a) {
    constexpr int NC = 3;                  // consumers per workgroup
    constexpr int SLOTS = 3;               // chunk slots per consumer
    constexpr int DEPTH = 3;               // DMA groups the loader leaves in flight while it still has something to issue
    constexpr int PST = 36, P_FLOATS = 8 * PST, KS = 68, STG = 256;
    constexpr int CW = SLOTS * 1024 + P_FLOATS + 2 * STG;   // floats per consumer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    volatile int* ctrl = reinterpret_cast<volatile int*>(smem + NC * CW);   // [NC][4]: landed, freed, sfreed, -
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, HD = H * 32, Kp = 16;
    const int L = a.h0 * a.w0, wq = a.w0 >> 1, Lq = a.nquads;
    // work list (as fine_quad): XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = (g < G && cnt > 0) ? a.B * cnt : 0;
    const int stride = (gridDim.x >> 3) * NC;
    if (threadIdx.x < NC * 4) ctrl[threadIdx.x] = 0;
    __syncthreads();
    auto items_of = [&](int t) { return t < total ? (total - t + stride - 1) / stride : 0; };
    const size_t pair_pitch = (size_t)H * a.lq1 * 128;
    const int un = lane & 7;
    unsigned cK[4];   // DMA source offset inside a parent's 512-byte run, instruction j of a chunk (rows 8 j + lane / 8; see fine_quad.hip)
#pragma unroll
    for (int j = 0; j < 4; ++j)
        cK[j] = (unsigned)(((lane >> 3) & 3) * 128 + ((un ^ (((j & 1) * 4 + (lane >> 4)) & 7)) * 16) + 3072 - j * 1024);

    if (wave == 0) {
        // ============================================================================================================ loader
        const float* const k0 = a.key + (size_t)h * a.lq1 * 128 - 768;     // this head's slice of pair 0, 3072 bytes low
        const float* const v0 = a.value + (size_t)h * a.lq1 * 128 - 768;
        // staging source per lane (one 16-byte unit each; fine_quad.hip): q 512 B | parents 64 B (4 units) | final[parent] 128 B
        unsigned long long sbase;
        unsigned mulq, mula;
        {
            const int u = lane < 48 ? lane : 47;
            if (u < 32) {
                const int r = u >> 3, pu = u & 7;
                sbase = (unsigned long long)a.q + (unsigned)(r * 128 + ((pu ^ (r >> 1)) * 16));
                mulq = 512u; mula = 0u;
            } else if (u < 40 || !a.acc_in) {
                sbase = (unsigned long long)a.parents + (unsigned)(min((u - 32) & 7, Kp / 4 - 1) * 16);
                mulq = (unsigned)(Kp * 4); mula = 0u;
            } else {
                sbase = (unsigned long long)a.acc_in + (unsigned)(h * 128 + (u - 40) * 16);
                mulq = 0u; mula = (unsigned)(HD * 4);
            }
        }
        const int t0 = (blockIdx.x >> 3) * NC;
        int T[NC], n[NC], s[NC], sl[NC];            // items, chunks issued, stagings issued, stagings landed
        int scb[NC], scq[NC], ccb[NC], ccq[NC];     // cursors (pair, index in the XCD's chunk) of the next staging / of the current chunk item
        unsigned ring_lds[NC], stg_lds[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            T[c] = items_of(t0 + c); n[c] = 0; s[c] = 0; sl[c] = 0;
            scb[c] = ccb[c] = (t0 + c) / max(cnt, 1); scq[c] = ccq[c] = (t0 + c) % max(cnt, 1);
            ring_lds[c] = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem + c * CW));
            stg_lds[c] = ring_lds[c] + (unsigned)((SLOTS * 1024 + P_FLOATS) * 4);
        }
        unsigned long long fifo = 0;     // 3 bits per outstanding DMA group, oldest in the low bits: owner (2 bits) | kind (bit 2: 1 = chunk)
        int flen = 0, outst = 0;         // groups / instructions in flight
        int spins = 0;
        unsigned long long l_blk = 0, l_idle = 0;
        const unsigned long long l_begin = __builtin_readcyclecounter();
        const bool hh = lane >> 5;
        for (;;) {
            bool all_done = true;
            bool issued = false;
            // one snapshot of the consumers' counters per sweep (three LDS reads in flight together, one wait)
            int freed[NC], sfreed[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { freed[c] = ctrl[c * 4 + 1]; sfreed[c] = ctrl[c * 4 + 2]; }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                // ---- (a) the front end of item s[c]: at most one item ahead of the chunks, and its buffer (used by item s - 2) released
                if (s[c] < T[c] && s[c] <= (n[c] >> 2) + 1 && (s[c] < 2 || sfreed[c] >= s[c] - 1)) {
                    const unsigned quad = (unsigned)(g * chunk + scq[c]);
                    const unsigned qd = (unsigned)((scb[c] * H + h) * Lq) + quad, bq = (unsigned)(scb[c] * Lq) + quad;
                    const unsigned long long addr = sbase + (unsigned long long)qd * mulq + (unsigned long long)bq * mula;
                    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(stg_lds[c] + (unsigned)((s[c] & 1) * STG * 4)));
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(addr), "s"(dst) : "memory");
                    fifo |= (unsigned long long)c << (3 * flen);
                    ++flen; ++outst; ++s[c]; issued = true;
                    scq[c] += stride;
                    while (scq[c] >= cnt) { scq[c] -= cnt; ++scb[c]; }
                }
                // ---- (b) chunks n[c] ..: the item's parent list has landed and the slot (chunk n - 3's) has been read
                for (int rep = 0; rep < SLOTS; ++rep) {
                    const int i = n[c] >> 2;
                    if (!(i < T[c] && sl[c] > i && n[c] - freed[c] < SLOTS)) break;
                    const int kind = n[c] & 3, hc = kind & 1;       // K0 K1 V0 V1; half of the parent list
                    const int* t2 = reinterpret_cast<const int*>(smem + c * CW + SLOTS * 1024 + P_FLOATS + (i & 1) * STG + 128);
                    const int4 pa4 = *reinterpret_cast<const int4*>(t2 + 8 * hc), pb4 = *reinterpret_cast<const int4*>(t2 + 8 * hc + 4);
                    const unsigned o0 = ((unsigned)(hh ? pa4.y : pa4.x) << 9) + cK[0], o1 = ((unsigned)(hh ? pa4.w : pa4.z) << 9) + cK[1];
                    const unsigned o2 = ((unsigned)(hh ? pb4.y : pb4.x) << 9) + cK[2], o3 = ((unsigned)(hh ? pb4.w : pb4.z) << 9) + cK[3];
                    const float* base = reinterpret_cast<const float*>(uniform_ptr(reinterpret_cast<const char*>(((kind & 2) ? v0 : k0) + (size_t)ccb[c] * pair_pitch)));
                    glds_chunk(base, o0, o1, o2, o3, (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds[c] + (unsigned)((n[c] % SLOTS) * 4096))));
                    fifo |= (unsigned long long)(c | 4) << (3 * flen);
                    ++flen; outst += 4; ++n[c]; issued = true;
                    if ((n[c] & 3) == 0) {
                        ccq[c] += stride;
                        while (ccq[c] >= cnt) { ccq[c] -= cnt; ++ccb[c]; }
                    }
                }
                all_done = all_done && s[c] >= T[c] && n[c] >= 4 * T[c];
            }
            if (all_done && flen == 0) break;   // everything issued AND retired (= published)
            // ---- (c) retire in order: keep at most DEPTH groups in flight behind the ones just issued; when nothing could be issued,
            //      retire the oldest one at once (its owner is probably waiting for it)
            bool retired = false;
            while (flen > (issued ? DEPTH : 0)) {
                const int code = (int)(fifo & 7ull), owner = code & 3, size = (code & 4) ? 4 : 1;
                { const unsigned long long t0_ = __builtin_readcyclecounter(); vmwait_dyn(outst - size); l_blk += __builtin_readcyclecounter() - t0_; }
                fifo >>= 3; --flen; outst -= size;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (owner == c) {
                        if (code & 4) { if (lane == 0) ctrl[c * 4 + 0] = ctrl[c * 4 + 0] + 1; }
                        else ++sl[c];
                    }
                retired = true;
                if (!issued) break;   // one at a time while idle: a slot may have been freed meanwhile
            }
            if (issued || retired) spins = 0;
            else {
                const unsigned long long t0_ = __builtin_readcyclecounter();
                __builtin_amdgcn_s_sleep(1);
                l_idle += __builtin_readcyclecounter() - t0_;
                if (++spins > LW_SPIN_MAX) { if (a.err && lane == 0) *a.err = 1; break; }
            }
        }
        if (a.dbg && lane == 0) {
            atomicAdd(a.dbg + 5, l_blk); atomicAdd(a.dbg + 6, l_idle); atomicAdd(a.dbg + 7, (unsigned long long)(__builtin_readcyclecounter() - l_begin));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================================================================================================ consumer
    const int c = wave - 1;
    const int t = (blockIdx.x >> 3) * NC + c;
    const int T = items_of(t);
    if (T == 0) return;
    float* ring = smem + c * CW;
    float* Pld = ring + SLOTS * 1024;
    float* stg = Pld + P_FLOATS;
    volatile int* my = ctrl + c * 4;
    unsigned va[8];    // V chunk: byte offset of V[row 2 mm + lane/32][d = lane%32] for mm % 8 == x, minus mm * 256
#pragma unroll
    for (int x = 0; x < 8; ++x) va[x] = (unsigned)((lane >> 5) * 128 + ((((lane & 31) >> 2) ^ x) * 16) + (lane & 3) * 4);
    const float* pa = Pld + ((lane & 3) * 2 + (lane >> 5)) * PST;   // operand A of the V chunks: P[child lane%4][parity lane/32][.]
    int cb = t / cnt, cq = t % cnt, cy = (g * chunk + cq) / wq, cx = (g * chunk + cq) % wq;
    const int sy = stride / wq, sx = stride % wq;
    bool failed = false;
    unsigned long long w_acc[3] = {0, 0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
    auto wait_landed = [&](int need) {
        int spins = 0;
        while (my[0] < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > LW_SPIN_MAX) { failed = true; break; }
        }
        asm volatile("" ::: "memory");
    };
    for (int i = 0; i < T && !failed; ++i) {
        const int b = cb, l00 = 2 * cy * a.w0 + 2 * cx;
        const float* qs = stg + (i & 1) * STG;
        // ---- K pass: candidates 0..31 in slot (4 i) % 3, 32..63 in slot (4 i + 1) % 3
        { const unsigned long long t0_ = __builtin_readcyclecounter(); wait_landed(4 * i + 2); w_acc[0] += __builtin_readcyclecounter() - t0_; }
        const int s0 = (4 * i) % SLOTS, s1 = (4 * i + 1) % SLOTS;
        const char* kb = reinterpret_cast<const char*>(ring) + (lane < 32 ? s0 : s1) * 4096 + (lane & 31) * 128;
        const float acc_cur = a.acc_in ? qs[160 + (lane & 31)] : 0.f;   // final[parent] of the item for d = lane % 32 (:277)
        f32x4 qa[8], kr[8];   // operand A: lane l holds q[child l%4][d]; operand B: this lane's candidate row
#pragma unroll
        for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qs + (lane & 3) * 32 + ((u ^ ((lane & 3) >> 1)) * 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kb + ((u ^ ((lane >> 1) & 7)) * 16));
        lds_reads_done();
        if (lane == 0) my[1] = 4 * i + 2;
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
        if (lane == 0) my[2] = i + 1;   // this item's staging area is no longer needed
        // ---- V chunks
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            { const unsigned long long t0_ = __builtin_readcyclecounter(); wait_landed(4 * i + 3 + cc); w_acc[1 + cc] += __builtin_readcyclecounter() - t0_; }
            const char* sb = reinterpret_cast<const char*>(ring) + ((4 * i + 2 + cc) % SLOTS) * 4096;
            f32x4 pv[4];   // operand A of MFMA mm: P[child lane%4][parity lane/32][16 cc + mm]
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const f32x4*>(pa + 16 * cc + 4 * k);
            float vb[16];
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) vb[mm] = *reinterpret_cast<const float*>(sb + va[mm & 7] + mm * 256);
            lds_reads_done();
            if (lane == 0) my[1] = 4 * i + 3 + cc;
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
        atomicAdd(a.dbg + 0, w_acc[0]); atomicAdd(a.dbg + 1, w_acc[1]); atomicAdd(a.dbg + 2, w_acc[2]);
        atomicAdd(a.dbg + 3, (unsigned long long)(__builtin_readcyclecounter() - t_begin)); atomicAdd(a.dbg + 4, (unsigned long long)T);
    }
}

int casmtr_qta_fine_level_lw(const float* q, const float* key, const float* value, const int32_t* parents, float temp, float w_level,
                             const float* acc_in, float* message, float* acc_out, int B, int h0, int w0, int h1, int w1, int H, int Kp,
                             hipStream_t s) {
    if (Kp != 16 || (H != 8 && H != 4 && H != 2 && H != 1) || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1)) return CASMTR_ERR_UNSUPPORTED;
    const long long lq0 = (long long)(h0 / 2) * (w0 / 2), lq1 = (long long)(h1 / 2) * (w1 / 2);
    if (lq1 >= (1 << 22) || (long long)B * H * lq0 * 512 >= (1ll << 32)) return CASMTR_ERR_UNSUPPORTED;
    FineLwArgs a{};
    a.q = q; a.key = key; a.value = value; a.parents = parents; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out; a.err = nullptr;
    a.temp = temp; a.w_level = w_level; a.B = B; a.h0 = h0; a.w0 = w0; a.H = H; a.nquads = (int)lq0; a.lq1 = (int)lq1;
    constexpr size_t lds = sizeof(float) * 3 * (3 * 1024 + 8 * 36 + 2 * 256) + 64;
    static int resident[CASMTR_MAX_DEVICES] = {0};
    int res = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fine_lw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (const int r = resident_workgroups(resident, fine_lw_kernel, 256, lds, &res)) return r;
    long long blocks = res;
    const char* ev = getenv("CASMTR_LW_BLOCKS");   // measurement knob: workgroups in the persistent grid (multiple of 8)
    if (ev && atoi(ev) > 0 && atoi(ev) < blocks) blocks = atoi(ev) / 8 * 8;
    const int G = 8 / H;
    const long long per_xcd = (long long)B * ((lq0 + G - 1) / G);
    if (blocks / 8 * 3 > per_xcd) blocks = (per_xcd + 2) / 3 * 8;
    if (blocks < 8) blocks = 8;
    static unsigned long long* dbg = nullptr;
    if (getenv("CASMTR_LW_DEBUG")) {
        if (!dbg) (void)hipMalloc(&dbg, 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 8 * sizeof(unsigned long long), s);
        a.dbg = dbg;
    }
    prof_symbol_args(CASMTR_PROF_QTA_FINE, "%s", "");
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_QTA_FINE, fine_lw_kernel, dim3((unsigned)blocks), dim3(256), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    if (a.dbg) {
        unsigned long long hdbg[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hdbg, dbg, sizeof hdbg, hipMemcpyDeviceToHost);
        const double it = (double)(hdbg[4] ? hdbg[4] : 1), nl = (double)blocks;
        fprintf(stderr, "fine_lw: %lld workgroups; per item: wait K %.0f, V0 %.0f, V1 %.0f of %.0f cycles; loader: blocked in vmcnt %.0f %%, idle %.0f %% of %.0f cycles\n",
                blocks, hdbg[0] / it, hdbg[1] / it, hdbg[2] / it, hdbg[3] / it, 100.0 * hdbg[5] / (double)hdbg[7], 100.0 * hdbg[6] / (double)hdbg[7], hdbg[7] / nl);
    }
    return 0;
}
