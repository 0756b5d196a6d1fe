// QTAttB.process_fine_level for candidate lists of at most 64 (the finest level of every shipped config): the fine_dma.hip
// pipeline with the VALUE rows kept out of LDS.
//
// fine_dma.hip stages both the key rows and the value rows of an item (one head of one quad) in LDS, 16 KB per item, which
// limits the CU to 8 waves -- and the bytes a CU can have in flight to what fits in LDS.  Here only the keys go through LDS
// (they have to: the logits need the lane <-> candidate transposition, done by the XOR-swizzled DMA + ds_read_b128), into ONE
// 8 KB buffer per wave that is refilled for the next item as soon as its rows are in registers.  The value rows are loaded
// straight into 32 VGPRs in the layout the A.V matrix-core instruction consumes (lane l <-> rows 2m + l/32, column l % 32: two
// full 128-byte lines per wave instruction), issued at the top of the item and landing under its logits / softmax / top-k.
// 11.4 KB of LDS per wave -> 12 waves per CU, each with 8 KB (next keys) + 8 KB (values) in flight.
// Arithmetic, selection and output are those of fine_dma.hip (and of quad_attn_kernel<H,KMAX,0>): bit-identical results.
//
// The value loads are inline asm (global_load_dword with a scalar base): as compiler-visible loads the scheduler sinks them
// next to their uses.  Their registers are made "ready" by the s_waitcnt asm that takes them as in-out operands; the kernel
// must not spill (checked at build time: ScratchSize 0).
#include "common.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

struct FineVArgs {
    const float* q;        // [B,L,H*32]
    const float* key;      // [B,S,H*32]
    const float* value;    // [B,S,H*32]
    const int64_t* pidx;   // [B,Lq,Kp,H]
    const float* acc_in;   // nullable [B,Lq,H*32]
    float* message;        // nullable [B,L,H*32]
    float* acc_out;        // nullable [B,L,H*32]
    float* topk_score;     // [B,L,topk,H]
    int64_t* topk_idx;     // [B,L,topk,H]
    float temp, w_level;
    int topk, B, h0, w0, h1, w1, H, Kp, nquads;
};

__device__ __forceinline__ float gload_f32(const float* base, unsigned byte_off) {
    float v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
    return v;
}

template <bool EXACT>   // EXACT: the logits feed a top-k (bit-exact sequential d-chain); otherwise only a softmax
__global__ __launch_bounds__(128, 3) void fine_level_vreg_kernel(const FineVArgs a) {
    constexpr int KMAX = 64, E = 4, KS = KMAX + 4;
    constexpr int WAVE_FLOATS = 2912;   // keys [64][32] | probabilities [64][4] | q [2][4][32] | parents [2][32] | logits [4][KS], padded to 128 B
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(2048 + KMAX * 4 + 2 * 128 + 2 * 32 + 4 * KS <= WAVE_FLOATS, "LDS layout");
    float* kb = smem + wave * WAVE_FLOATS;
    float* Ald = kb + 2048;
    float* qs = Ald + KMAX * 4;
    int* ptab = reinterpret_cast<int*>(qs + 2 * 128);
    float* Sld = reinterpret_cast<float*>(ptab + 2 * 32);
    const int H = a.H, HD = H * 32, Kp = a.Kp, K = 4 * Kp;
    const int L = a.h0 * a.w0, S = a.h1 * a.w1, wq = a.w0 >> 1, Lq = a.nquads, w1p = a.w1 >> 1;
    // ---- work list: XCD x -> head x % H; the 8 / H XCDs sharing a head split every pair's quads into contiguous chunks
    const int xcd = blockIdx.x & 7, h = xcd % H, G = 8 / H, g = xcd / H;
    const int chunk = (Lq + G - 1) / G, cnt = min(chunk, Lq - g * chunk);
    const int total = cnt > 0 ? a.B * cnt : 0, stride = (gridDim.x >> 3) * 2;
    const int t = (blockIdx.x >> 3) * 2 + wave;
    if (g >= G || t >= total) return;
    const unsigned kb_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(kb));
    const int sl = lane >> 3, un = lane & 7;
    const unsigned swz[2] = {(unsigned)((un ^ (lane >> 4)) * 16), (unsigned)((un ^ (4 + (lane >> 4))) * 16)};
    unsigned rd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rd[u] = (unsigned)(lane * 128 + ((u ^ ((lane >> 1) & 7)) * 16));
    const float* kh = a.key + h * 32;
    const float* vh = a.value + h * 32;
    const unsigned voff_lane = (unsigned)((lane >> 5) * HD * 4 + (lane & 31) * 4);   // value row 2m + lane/32, column lane % 32
    const unsigned row_bytes = (unsigned)(HD * 4);

    long long pf_p = 0;
    f32x4 pf_q = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pf_acc = 0.f, acc_cur = 0.f, acc_nx = 0.f;
    struct Item { int b, quad, l00; };
    int cb = t / cnt, cq = t % cnt, cy = (g * chunk + cq) / wq, cx = (g * chunk + cq) % wq;
    const int sy = stride / wq, sx = stride % wq;
    auto take = [&](Item& it) {
        if (cb >= a.B) return false;
        it.b = cb; it.quad = g * chunk + cq; it.l00 = 2 * cy * a.w0 + 2 * cx;
        cq += stride;
        if (cq >= cnt) {
            while (cq >= cnt) { cq -= cnt; ++cb; }
            cy = (g * chunk + cq) / wq; cx = (g * chunk + cq) % wq;
        } else {
            cy += sy; cx += sx;
            if (cx >= wq) { cx -= wq; ++cy; }
        }
        return true;
    };
    auto prefetch = [&](const Item& it) {
        // lanes Kp..15 repeat the last parent: the table then holds 16 valid rows (their candidates carry probability 0)
        if (lane < 16) pf_p = a.pidx[(((size_t)it.b * Lq + it.quad) * Kp + min(lane, Kp - 1)) * H + h];
        if (lane < 32) {
            const int f = lane >> 3, lf = it.l00 + (f >> 1) * a.w0 + (f & 1);
            pf_q = *reinterpret_cast<const f32x4*>(a.q + ((size_t)it.b * L + lf) * HD + h * 32 + un * 4);
            if (a.acc_in) pf_acc = a.acc_in[((size_t)it.b * Lq + it.quad) * HD + h * 32 + lane];
        }
    };
    unsigned rowb[8];   // key DMA instruction j moves candidate rows 8j .. 8j+7: lane -> row 8j + lane/8
    auto stage_in = [&](int par) {
        if (lane < 16) {
            const int p = (int)pf_p;
            ptab[par * 32 + lane] = (p / w1p) * 2 * a.w1 + (p % w1p) * 2;   // :193-199, children (+0,+0),(+0,+1),(+1,+0),(+1,+1)
        }
        if (lane < 32) *reinterpret_cast<f32x4*>(qs + par * 128 + lane * 4) = pf_q;
        acc_nx = pf_acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * j + sl;   // rows >= K read the repeated last parent's children: valid memory, probability 0
            rowb[j] = (unsigned)(ptab[par * 32 + (k >> 2)] + ((k >> 1) & 1) * a.w1 + (k & 1)) * row_bytes;
        }
    };
    auto issue_keys = [&](int b) {
        const float* base = kh + (size_t)b * S * HD;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(base, rowb[j] + swz[j & 1], kb_lds + (unsigned)(j * 1024));
    };

    Item it_cur{}, it_nx{}, it_pf{};
    take(it_cur);
    prefetch(it_cur);
    stage_in(0);
    acc_cur = acc_nx;
    int par = 0;
    f32x4 pend = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pend_acc = 0.f;
    int pend_b = 0, pend_l00 = 0;
    bool have_pend = false;
    auto flush = [&]() {
        if (have_pend && lane < 32) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const size_t o = ((size_t)pend_b * L + pend_l00 + (f >> 1) * a.w0 + (f & 1)) * HD + h * 32 + lane;
                if (a.message) a.message[o] = pend[f];
                if (a.acc_out) a.acc_out[o] = pend_acc + pend[f] * a.w_level;   // separate multiply and add (:277-281)
            }
        }
        have_pend = false;
    };
    issue_keys(it_cur.b);
    bool more = take(it_nx), has_pf = false;
    if (more) prefetch(it_nx);
    for (;; par ^= 1) {
        const int b = it_cur.b, l00 = it_cur.l00;
        const float* qsp = qs + par * 128;
        const int* ptp = ptab + par * 32;
        // ---- (a) this item's value rows -> registers: 32 loads of two full lines each, in flight until (g)
        float vv[32];
        {
            const float* vbase = vh + (size_t)b * S * HD;
            int pt[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 t4 = *reinterpret_cast<const int4*>(ptp + 4 * i);
                pt[4 * i] = t4.x; pt[4 * i + 1] = t4.y; pt[4 * i + 2] = t4.z; pt[4 * i + 3] = t4.w;
            }
#pragma unroll
            for (int m = 0; m < 32; ++m)   // rows 2m, 2m+1 = children (m & 1, 0), (m & 1, 1) of parent m / 2
                vv[m] = gload_f32(vbase, (unsigned)(pt[m >> 1] + (m & 1) * a.w1) * row_bytes + voff_lane);
        }
        // ---- (b) this item's keys have landed (everything older than the 32 value loads)
        glds_wait<32>();
        flush();   // the previous item's results: stores issued here retire under this item's arithmetic
        // ---- (c) keys and queries -> registers
        f32x4 qa[8], kr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) qa[u] = *reinterpret_cast<const f32x4*>(qsp + (lane & 3) * 32 + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) kr[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(kb) + rd[u]);
        lds_reads_done();
        // ---- (d) the key buffer is free: next item's front end and its key DMA; (e) the item after next: index / query loads
        if (more) {
            stage_in(par ^ 1);
            issue_keys(it_nx.b);
        }
        has_pf = more && take(it_pf);
        if (has_pf) prefetch(it_pf);
        // ---- (f) logits on the matrix cores, softmax, top-k
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (EXACT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c, 0, 0, 0);
            }
        } else {
            f32x4 c4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].x, kr[u].x, c4[0], 0, 0, 0);
                c4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].y, kr[u].y, c4[1], 0, 0, 0);
                c4[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].z, kr[u].z, c4[2], 0, 0, 0);
                c4[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[u].w, kr[u].w, c4[3], 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) c[f] = (c4[0][f] + c4[1][f]) + (c4[2][f] + c4[3][f]);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) Sld[f * KS + lane] = a.temp * c[f];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int f = lane >> 4, j = lane & 15;
            float lv[E];
            unsigned key[E];
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(Sld + f * KS + j * E);
            lv[0] = v4.x; lv[1] = v4.y; lv[2] = v4.z; lv[3] = v4.w;
            unsigned lm = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                key[e] = (j * E + e < K) ? f2ord(lv[e]) : 0u;
                lm = max(lm, key[e]);
            }
            const float mx = ord2f(row16_max_u32(lm));
            float ps[E];
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] = (j * E + e < K) ? __expf(lv[e] - mx) : 0.f;
                sum += ps[e];
            }
            sum = 1.0f / row16_sum_f32(sum);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ps[e] = ps[e] * sum;
                Ald[(j * E + e) * 4 + f] = ps[e];
            }
            const int lf = l00 + (f >> 1) * a.w0 + (f & 1);
            for (int tk = 0; tk < a.topk; ++tk) {   // selection on the logits, (logit desc, position asc)
                unsigned cur = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) cur = max(cur, key[e]);
                const unsigned rm = row16_max_u32(cur);
                const unsigned long long bal = __ballot(cur == rm);
                const unsigned bits = (unsigned)(bal >> (f * 16)) & 0xFFFFu;
                const int wj = __ffs(bits) - 1;
                if (j == wj) {
                    bool done = false;
                    int kpos = 0;
                    float sc2 = 0.f;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const bool hit = !done && key[e] == rm;
                        if (hit) { kpos = j * E + e; sc2 = ps[e]; key[e] = 0u; done = true; }
                    }
                    const size_t o = (((size_t)b * L + lf) * a.topk + tk) * H + h;
                    a.topk_idx[o] = ptp[kpos >> 2] + ((kpos >> 1) & 1) * a.w1 + (kpos & 1);
                    a.topk_score[o] = sc2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- (g) the value rows have landed: everything older than the next item's 8 key DMAs (the index / query loads and
        //          top-k stores issued after them only make this wait more conservative)
        // One wait statement for both cases (a second one in an else-branch made the register allocator copy the value registers
        // in front of it, i.e. before the data had landed): without a next item nothing younger is in flight, so drain first.
        if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(8)"
                     : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]),
                       "+v"(vv[8]), "+v"(vv[9]), "+v"(vv[10]), "+v"(vv[11]), "+v"(vv[12]), "+v"(vv[13]), "+v"(vv[14]), "+v"(vv[15])
                     :: "memory");
        asm volatile(""
                     : "+v"(vv[16]), "+v"(vv[17]), "+v"(vv[18]), "+v"(vv[19]), "+v"(vv[20]), "+v"(vv[21]), "+v"(vv[22]), "+v"(vv[23]),
                       "+v"(vv[24]), "+v"(vv[25]), "+v"(vv[26]), "+v"(vv[27]), "+v"(vv[28]), "+v"(vv[29]), "+v"(vv[30]), "+v"(vv[31])
                     :: "memory");
        // ---- (h) message = A . V, two rows per instruction, values from registers
        {
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* prow = Ald + (lane >> 5) * 4 + (lane & 3);   // + 8 m : P[row 2m + lane/32][child lane % 4]
#pragma unroll
            for (int m = 0; m < 32; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(prow[8 * m], vv[m], acc[m & 3], 0, 0, 0);
            f32x4 tot;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float x = (acc[0][cc] + acc[1][cc]) + (acc[2][cc] + acc[3][cc]);
                const unsigned xi = __float_as_uint(x);
                const auto sw = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);   // lanes l and l ^ 32
                tot[cc] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            pend = tot; pend_acc = acc_cur; pend_b = b; pend_l00 = l00; have_pend = true;
        }
        lds_reads_done();
        if (!more) break;
        acc_cur = acc_nx;
        it_cur = it_nx; it_nx = it_pf; more = has_pf;
    }
    flush();
}

template <bool EXACT>
static int launch_fine_vreg(const FineVArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * 2 * 2912;
    static int resident_tab[CASMTR_MAX_DEVICES] = {0};
    int resident = 0;
    if (const int r = resident_workgroups(resident_tab, fine_level_vreg_kernel<EXACT>, 128, lds, &resident)) return r;
    const long long work = (long long)a.B * a.nquads * a.H;
    long long blocks = resident;
    if (blocks > (work + 1) / 2) blocks = ((work + 1) / 2 + 7) / 8 * 8;
    ProfScope ps(CASMTR_PROF_QTA_FINE, s, "fine_level_vreg_kernel");
    hipLaunchKernelGGL((fine_level_vreg_kernel<EXACT>), dim3((unsigned)blocks), dim3(128), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

// -> CASMTR_ERR_UNSUPPORTED when the shape is outside this kernel (the caller then uses fine_dma.hip / quad_attn_kernel)
int casmtr_qta_fine_level_vreg(const float* q, const float* key, const float* value, const int64_t* prev_idx, float temp, int topk,
                               float w_level, const float* acc_in, float* message, float* acc_out, float* topk_score,
                               int64_t* topk_idx, int B, int h0, int w0, int h1, int w1, int H, int Kp, hipStream_t s) {
    if (4 * Kp > 64 || Kp < 1 || (H != 8 && H != 4 && H != 2 && H != 1)) return CASMTR_ERR_UNSUPPORTED;
    if ((long long)h1 * w1 * H * 32 * 4 >= (1ll << 32)) return CASMTR_ERR_UNSUPPORTED;   // 32-bit row offsets inside one pair
    FineVArgs a{};
    a.q = q; a.key = key; a.value = value; a.pidx = prev_idx; a.acc_in = acc_in; a.message = message; a.acc_out = acc_out;
    a.topk_score = topk_score; a.topk_idx = topk_idx; a.temp = temp; a.w_level = w_level; a.topk = topk; a.B = B;
    a.h0 = h0; a.w0 = w0; a.h1 = h1; a.w1 = w1; a.H = H; a.Kp = Kp; a.nquads = (h0 / 2) * (w0 / 2);
    return topk > 0 ? launch_fine_vreg<true>(a, s) : launch_fine_vreg<false>(a, s);
}
