// Finest QTAttB level with the gathers IN FLIGHT IN VGPRS (round 5): the same items, LDS layout and arithmetic as
// fine_quad_kernel<1, false, true> (cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py:180-229 with lists of
// exactly 64 candidates and no top-k: the finest level of every shipped config) -- results bit-equal to that kernel's.
//
// Why.  fine_quad.hip gathers an item's 64 K rows and 64 V rows (16 KB) straight into LDS with LDS-DMA and computes when they have
// landed.  The rows come from a 2.8 MB (pair, head) slice that an XCD's 4 MB L2 holds only in part (14 % of the gathered bytes miss),
// a 1 KB gather instruction touches eight lines, so seven out of ten instructions wait for HBM and a group of eight lands after
// 4000-5000 cycles; the wave then computes for ~2800.  While the rows are in flight their 16 KB of LDS are reserved and empty:
// LDS bytes x time is what bounds that kernel (ten waves per CU, one item per wave per ~6800 cycles = 197 us per launch), and the
// loader-wave variant (fine_lw.hip) showed the same bound from the other side.  The register file is three times the LDS: here the
// NEXT item's rows fly into 64 VGPRs per lane with ordinary coalesced loads (the same lane -> 16-byte-unit map as the DMA
// instructions) while the wave computes the current item from an 8 KB LDS buffer; when they have landed eight ds_write_b128
// reproduce the DMA's LDS image, first of the K rows and, once the K pass has read them, of the V rows in the same buffer.
// Per wave: 8 KB rows + probabilities + three 704-byte staging buffers = 11.2 KB of LDS, twelve waves per CU.
#include <stdio.h>
#include <stdlib.h>
#include "quad_common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct FineVsArgs {
    const float* q;          // [B,H,Lq0,4,32]
    const float* key;        // [B,H,Lq1,4,32]
    const float* value;      // [B,H,Lq1,4,32]
    const int32_t* parents;  // [B,H,Lq0,16]
    const float* acc_in;     // nullable [B,Lq0,H*32]
    float* message;          // nullable [B,L,H*32]
    float* acc_out;          // nullable [B,L,H*32]
    unsigned long long* dbg; // nullable (CASMTR_VS_DEBUG): [0] cycles waiting for the rows, [1] total, [2] items
    float temp, w_level;
    int B, h0, w0, H, nquads, lq1, xflags;
};

constexpr int VS_PST = 36, VS_P_FLOATS = 8 * VS_PST, VS_KS = 68;
constexpr int VS_STG = 176;                                         // floats per staging buffer: q 128 | parents 16 | final[parent] 32
constexpr int VS_WW = 2048 + VS_P_FLOATS + 3 * VS_STG;              // floats per wave: rows | P | staging x 3

#define VS_LOAD(dst, off, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")

template <int NST, int ABL>   // output stores per item (2 per written tensor); bit 0: cycle counters; other bits: ablations (timing only)
__global__ __launch_bounds__(256, 3) void fine_vs_kernel(const FineVsArgs a) {
    constexpr bool DBG = ABL & 1;
    constexpr int PST = VS_PST, P_FLOATS = VS_P_FLOATS, KS = VS_KS, STG = VS_STG, WW = VS_WW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, HD = H * 32, Kp = 16;
    const int L = a.h0 * a.w0, wq = a.w0 >> 1, Lq = a.nquads;
    // work list (as fine_quad): XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = (g < G && cnt > 0) ? a.B * cnt : 0;
    const int stride = (gridDim.x >> 3) * 4;
    const int t = (blockIdx.x >> 3) * 4 + wave;
    const int T = t < total ? (total - t + stride - 1) / stride : 0;
    if (T == 0) return;

    float* ring = smem + wave * WW;     // 64 rows x 128 B, XOR-swizzled 16-byte units: K rows, then the V rows
    float* Pld = ring + 2048;
    float* stg = Pld + P_FLOATS;
    const unsigned stg_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(stg));
    const size_t pair_pitch = (size_t)H * a.lq1 * 128;
    const float* const k0 = a.key + (size_t)h * a.lq1 * 128;     // this head's slice of pair 0
    const float* const v0 = a.value + (size_t)h * a.lq1 * 128;
    const int un = lane & 7;
    unsigned cK[2];   // source offset inside a parent's 512-byte run for load j (rows 8 j + lane / 8; the DMA instructions' map, fine_quad.hip)
#pragma unroll
    for (int j = 0; j < 2; ++j) cK[j] = (unsigned)(((lane >> 3) & 3) * 128 + ((un ^ ((j * 4 + (lane >> 4)) & 7)) * 16));
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
    unsigned va[8];    // V rows: byte offset of V[row 2 mm + lane/32][d = lane%32] for mm % 8 == x, minus mm * 256
#pragma unroll
    for (int x = 0; x < 8; ++x) va[x] = (unsigned)((lane >> 5) * 128 + ((((lane & 31) >> 2) ^ x) * 16) + (lane & 3) * 4);
    const float* pa = Pld + ((lane & 3) * 2 + (lane >> 5)) * PST;   // operand A of the V pass: P[child lane%4][parity lane/32][.]
    const char* kb = reinterpret_cast<const char*>(ring) + lane * 128;
    char* wr = reinterpret_cast<char*>(ring) + lane * 16;           // this lane's 16-byte unit of load j goes to wr + 1024 j
    const bool hh = lane >> 5;

    // item cursors: `cur` the item being computed, `nx` the one whose rows are in flight, `st` the next one to stage
    struct Cur { int b, q; };
    auto advance = [&](Cur& c) { c.q += stride; while (c.q >= cnt) { c.q -= cnt; ++c.b; } };
    Cur cur{t / cnt, t % cnt}, nx = cur, st = cur;
    int cy = (g * chunk + cur.q) / wq, cx = (g * chunk + cur.q) % wq;
    const int sy = stride / wq, sx = stride % wq;

    auto stage = [&](const Cur& c, int buf) {   // one 44-lane LDS-DMA instruction
        const unsigned quad = (unsigned)(g * chunk + c.q);
        const unsigned qd = (unsigned)((c.b * H + h) * Lq) + quad, bq = (unsigned)(c.b * Lq) + quad;
        const unsigned long long addr = sbase + (unsigned long long)qd * mulq + (unsigned long long)bq * mula;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(stg_lds + (unsigned)(buf * STG * 4)));
        unsigned long long keep;
        asm volatile("s_mov_b64 %0, exec\n\ts_bfm_b64 exec, 44, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(addr), "s"(dst) : "memory");
    };
    f32x4 kreg[8], vreg[8];
    unsigned voff[8];
    auto offsets = [&](int buf) {   // the staged parent list -> the eight source offsets (K and V rows share them)
        const int* t2 = reinterpret_cast<const int*>(stg + buf * STG + 128);
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
            const int4 pa4 = *reinterpret_cast<const int4*>(t2 + 8 * hc), pb4 = *reinterpret_cast<const int4*>(t2 + 8 * hc + 4);
            voff[4 * hc + 0] = ((unsigned)(hh ? pa4.y : pa4.x) << 9) + cK[0];
            voff[4 * hc + 1] = ((unsigned)(hh ? pa4.w : pa4.z) << 9) + cK[1];
            voff[4 * hc + 2] = ((unsigned)(hh ? pb4.y : pb4.x) << 9) + cK[0];
            voff[4 * hc + 3] = ((unsigned)(hh ? pb4.w : pb4.z) << 9) + cK[1];
        }
    };
    auto uniform_base = [&](const float* p0, int b) {
        const unsigned long long v = (unsigned long long)(p0 + (size_t)b * pair_pitch);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (const float*)(((unsigned long long)hi << 32) | lo);
    };
    auto load_k = [&](int b) {
        const float* base = uniform_base(k0, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { if constexpr (ABL & 16) kreg[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; else VS_LOAD(kreg[j], voff[j], base); }
    };
    auto load_v = [&](int b) {
        const float* base = uniform_base((ABL & 128) ? k0 : v0, b);   // (128: timing experiment, the V rows come from the K slice: half the L2 working set)
#pragma unroll
        for (int j = 0; j < 8; ++j) { if constexpr (ABL & 16) vreg[j] = (f32x4){1.f, 0.f, 0.f, 0.f}; else VS_LOAD(vreg[j], voff[j], base); }
    };
#define VS_LANDED(N)                                                                                                                       \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                                 \
                 : "+v"(kreg[0]), "+v"(kreg[1]), "+v"(kreg[2]), "+v"(kreg[3]), "+v"(kreg[4]), "+v"(kreg[5]), "+v"(kreg[6]), "+v"(kreg[7]), \
                   "+v"(vreg[0]), "+v"(vreg[1]), "+v"(vreg[2]), "+v"(vreg[3]), "+v"(vreg[4]), "+v"(vreg[5]), "+v"(vreg[6]), "+v"(vreg[7])  \
                 :: "memory")
    // all the loads this wave has in flight; the `younger` (0, 2 or 4) output stores of the previous item may stay in flight (vector
    // memory operations complete in order); the operands keep every use of the registers behind the wait
    auto landed = [&]() {
        if constexpr (NST == 4) VS_LANDED(4);
        else if constexpr (NST == 2) VS_LANDED(2);
        else VS_LANDED(0);
    };
    unsigned long long w_acc = 0;
    const unsigned long long t_begin = DBG ? __builtin_readcyclecounter() : 0ull;

    // prologue: stage items 0 and 1, gather item 0
    int b0 = 0, b1 = 1, b2 = 2;   // staging buffers of items i, i + 1, i + 2
    stage(st, b0); advance(st);
    if (T > 1) { stage(st, b1); advance(st); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    offsets(b0);
    load_k(nx.b); load_v(nx.b); advance(nx);
    VS_LANDED(0);   // (no stores are in flight yet)

    for (int i = 0; i < T; ++i) {
        const int b = cur.b, l00 = 2 * cy * a.w0 + 2 * cx;
        const float* qs = stg + b0 * STG;
        {
            const unsigned long long t0_ = DBG ? __builtin_readcyclecounter() : 0ull;
            landed();   // rows of item i in registers; staging of item i + 1 in LDS
            if constexpr (DBG) w_acc += __builtin_readcyclecounter() - t0_;
        }
        // ---- K rows -> LDS; the next item's K rows take off
#pragma unroll
        for (int j = 0; j < ((ABL & 32) ? 1 : 8); ++j) *reinterpret_cast<f32x4*>(wr + 1024 * j) = kreg[j];
        const bool more = i + 1 < T;
        if (more) { offsets(b1); load_k(nx.b); }
        if (i + 2 < T) { stage(st, b2); advance(st); }
        wave_lds_fence();
        const float acc_cur = a.acc_in ? qs[144 + (lane & 31)] : 0.f;   // final[parent] of the item for d = lane % 32 (:277)
        f32x4 c4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c4[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // operand A: lane l holds q[child l%4][d]; operand B: this lane's candidate row.  Two halves of the feature dimension: the rows
        // in flight take 64 registers of the 168
#pragma unroll
        for (int half = (ABL & 8) ? 1 : 0; half < 2; ++half) {
            f32x4 qa[4], kr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qs + (lane & 3) * 32 + (((4 * half + u) ^ ((lane & 3) >> 1)) * 4));
#pragma unroll
            for (int u = 0; u < 4; ++u) kr[u] = *reinterpret_cast<const f32x4*>(kb + (((4 * half + u) ^ ((lane >> 1) & 7)) * 16));
            if (half == 1) {
                // ---- V rows -> the same buffer; the next item's V rows take off
                lds_reads_done();
                wave_lds_fence();
#pragma unroll
                for (int j = 0; j < ((ABL & 32) ? 1 : 8); ++j) *reinterpret_cast<f32x4*>(wr + 1024 * j) = vreg[j];
                if (more) { load_v(nx.b); advance(nx); }
            }
#pragma unroll
            for (int u = 0; u < ((ABL & 8) ? 1 : 4); ++u) {   // no index depends on these logits: four interleaved partial d-chains
                c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c4[0], 0, 0, 0);
                c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c4[1], 0, 0, 0);
                c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c4[2], 0, 0, 0);
                c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c4[3], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        // ---- softmax, one series (child) per 16-lane row (fine_quad.hip: softmax_select without the selection)
        if constexpr (ABL & 2) {
            Pld[lane] = (c4[0][0] + c4[1][1]) + (c4[2][2] + c4[3][3]);
            wave_lds_fence();
        } else {
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
        // ---- V pass
        f32x4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = (ABL & 4) ? 1 : 0; cc < 2; ++cc) {
            const char* sb = reinterpret_cast<const char*>(ring) + cc * 4096;
            f32x4 pv[4];   // operand A of MFMA mm: P[child lane%4][parity lane/32][16 cc + mm]
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const f32x4*>(pa + 16 * cc + 4 * k);
            float vb[16];
#pragma unroll
            for (int mm = 0; mm < ((ABL & 4) ? 1 : 16); ++mm) vb[mm] = *reinterpret_cast<const float*>(sb + va[mm & 7] + mm * 256);
#pragma unroll
            for (int mm = 0; mm < ((ABL & 4) ? 1 : 16); ++mm)
                acc[mm & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(pv[mm >> 2][mm & 3], vb[mm], acc[mm & 3], 0, 0, 0);
        }
        lds_reads_done();
        wave_lds_fence();   // the rows have been read: the next iteration overwrites them
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
            if ((ABL & 64) && vA != 12345.f) {
            } else if (a.message) { a.message[o] = vA; a.message[o + HD] = vB; }
            if ((ABL & 64) && vB != 12345.f) {
            } else if (a.acc_out) {   // separate multiply and add (:277-281)
                a.acc_out[o] = acc_cur + vA * a.w_level;
                a.acc_out[o + HD] = acc_cur + vB * a.w_level;
            }
        }
        // next item
        { const int r = b0; b0 = b1; b1 = b2; b2 = r; }
        cur.q += stride;
        if (cur.q >= cnt) {
            while (cur.q >= cnt) { cur.q -= cnt; ++cur.b; }
            cy = (g * chunk + cur.q) / wq; cx = (g * chunk + cur.q) % wq;
        } else {
            cy += sy; cx += sx;
            if (cx >= wq) { cx -= wq; ++cy; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DBG && a.dbg && lane == 0) {
        atomicAdd(a.dbg + 0, w_acc); atomicAdd(a.dbg + 1, (unsigned long long)(__builtin_readcyclecounter() - t_begin)); atomicAdd(a.dbg + 2, (unsigned long long)T);
    }
}

int casmtr_qta_fine_level_vs(const float* q, const float* key, const float* value, const int32_t* parents, float temp, float w_level,
                             const float* acc_in, float* message, float* acc_out, int B, int h0, int w0, int h1, int w1, int H, int Kp,
                             hipStream_t s) {
    if (Kp != 16 || (H != 8 && H != 4 && H != 2 && H != 1) || (h0 & 1) || (w0 & 1) || (h1 & 1) || (w1 & 1)) return CASMTR_ERR_UNSUPPORTED;
    const long long lq0 = (long long)(h0 / 2) * (w0 / 2), lq1 = (long long)(h1 / 2) * (w1 / 2);
    if (lq1 >= (1 << 22) || (long long)B * H * lq0 * 512 >= (1ll << 32)) return CASMTR_ERR_UNSUPPORTED;
    FineVsArgs a{};
    a.q = q; a.key = key; a.value = value; a.parents = parents; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out;
    a.temp = temp; a.w_level = w_level; a.B = B; a.h0 = h0; a.w0 = w0; a.H = H; a.nquads = (int)lq0; a.lq1 = (int)lq1;
    constexpr size_t lds = sizeof(float) * 4 * VS_WW;
    static int resident[CASMTR_MAX_DEVICES] = {0};
    int res = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fine_vs_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (const int r = resident_workgroups(resident, fine_vs_kernel<0, 0>, 256, lds, &res)) return r;
    long long blocks = res;
    const char* ev = getenv("CASMTR_VS_BLOCKS");   // measurement knob: workgroups in the persistent grid (multiple of 8)
    if (ev && atoi(ev) > 0 && atoi(ev) < blocks) blocks = atoi(ev) / 8 * 8;
    const int G = 8 / H;
    const long long per_xcd = (long long)B * ((lq0 + G - 1) / G);
    if (blocks / 8 * 4 > per_xcd) blocks = (per_xcd + 3) / 4 * 8;
    if (blocks < 8) blocks = 8;
    { const char* ef = getenv("CASMTR_VS_FLAGS"); a.xflags = ef ? atoi(ef) : 0; }   // measurement knob: 1 = wait for the previous item's stores too
    static unsigned long long* dbg = nullptr;
    if (getenv("CASMTR_VS_DEBUG")) {
        if (!dbg) (void)hipMalloc(&dbg, 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 8 * sizeof(unsigned long long), s);
        a.dbg = dbg;
    }
    prof_symbol_args(CASMTR_PROF_QTA_FINE, "%s", "");
    const int nst = (a.xflags & 1) ? 0 : (message ? 2 : 0) + (acc_out ? 2 : 0);
#define VS_GO(N, D)                                                                                                                       \
    do {                                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fine_vs_kernel<N, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        CASMTR_LAUNCH_TIMED(CASMTR_PROF_QTA_FINE, (fine_vs_kernel<N, D>), dim3((unsigned)blocks), dim3(256), lds, s, a);                   \
    } while (0)
    const char* ea = getenv("CASMTR_VS_ABLATE");   // measurement knob (wrong results): 2 no softmax, 4 no V pass, 8 no K pass, 16 no row loads, 32 no row writes to LDS, 64 no stores
    const int abl = ea && nst == 4 ? atoi(ea) : 0;
    if (a.dbg) VS_GO(0, 1);
    else if (abl == 2) VS_GO(4, 2);
    else if (abl == 4) VS_GO(4, 4);
    else if (abl == 8) VS_GO(4, 8);
    else if (abl == 16) VS_GO(4, 16);
    else if (abl == 32) VS_GO(4, 32);
    else if (abl == 64) VS_GO(0, 64);
    else if (abl == 126) VS_GO(0, 126);
    else if (abl == 46) VS_GO(4, 46);
    else if (abl == 128) VS_GO(4, 128);
    else if (abl == 80) VS_GO(0, 80);
    else if (nst == 4) VS_GO(4, 0);
    else if (nst == 2) VS_GO(2, 0);
    else VS_GO(0, 0);
#undef VS_GO
    CASMTR_CHECK_LAUNCH();
    if (a.dbg) {
        unsigned long long hdbg[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hdbg, a.dbg, sizeof hdbg, hipMemcpyDeviceToHost);
        const double it = (double)(hdbg[2] ? hdbg[2] : 1);
        fprintf(stderr, "fine_vs: %lld workgroups (%d resident); per item: waiting for the rows %.0f of %.0f cycles\n", blocks, res, hdbg[0] / it, hdbg[1] / it);
    }
    return 0;
}
