// Shared pieces of the dual-softmax (CoarseMatching) kernels: workspace carve, tile epilogue.
//   matching.hip  exact fp32 GEMM (v_mfma_f32_32x32x2_f32 = the oracle's fmaf chain), statistics passes, selection
//   ds_split.hip  fp32-accurate GEMM on the f16 matrix pipe (3 products of a two-term f16 split) + exact re-decision of near-ties
#pragma once
#include "common.hpp"

#define DS_BM 128
#define DS_BN 128
#define DS_BK 32
#define DS_CAND_CAP 8        // near-tie candidates kept per row / column; more -> the whole call falls back to the exact GEMM
#define DS_X_CAP 8192        // borderline entries (confidence near thr / near a row or column runner-up) re-decided exactly per call
#define DS_XL_PCAP 1024      // rows (and columns) per image pair whose softmax statistics are recomputed with the exact chain
#define DS_XL_G 8            // listed lines of one pair and side that share one pass over the other side's features

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace casmtr {

struct DsWs {  // carve of stats_ws
    float *rp_m, *rp_s; int* rp_a;   // row partials  [B][NJB][L]
    float *cp_m, *cp_s; int* cp_a;   // col partials  [B][NIB][S]
    float *rmax, *rsum, *cmax, *csum;  // [B*L], [B*S]
    unsigned long long *rbest, *cbest;  // packed (conf bits << 32 | ~idx)
    // ---- split path only (everything from rcnt to ovf is zeroed together with rbest / cbest)
    int *rcnt, *ccnt;                // candidates found per row / column
    unsigned *namax, *nbmax;         // [B] max row norm (float bits)
    int* ovf;                        // != 0: some row / column had more than DS_CAND_CAP candidates (or a borderline list overflowed)
    // exact re-decision of borderline match-list entries (ds_split.hip, "match list exact by construction")
    int* xcnt;                       // [4]: entries (the rest spare)
    int* xln;                        // [2][B]: listed rows / listed columns per pair
    unsigned char *rneed, *cneed;    // [B*L], [B*S] line already on rlist / clist
    unsigned char* rdec;             // [B*L] bit 0: the row's match decision was re-made exactly, bit 1: it is a match
    char* zero_end;
    unsigned char* flags;            // [B*L]
    int64_t* jsel;                   // [B*L]
    float* csel;                     // [B*L]
    int* blk;                        // compaction block counts
    float *na, *nb;                  // [B][Lp], [B][Sp] |row| / sqrt(C), rounded up
    float *fa, *fb;                  // [B][Lp], [B][Sp] epilogue factors (2^e / (C T), 2^e)
    int *exA, *exB;                  // [B][Lp], [B][Sp] normalisation exponents e
    float *rthr, *cthr;              // [B*L], [B*S] candidate thresholds
    int *fl_rn, *fl_cn, *fl_tn;      // flagged-segment lists (ds_flagged_launch): lines per (pair, column block) / (pair, row block); work items
    int *fl_r, *fl_c, *fl_t;         // [B][NJB][L] flagged rows, [B][NIB][S] flagged columns, [B (NJB NIB' ...)] work items
    int *rcand, *ccand;              // [B*L][CAP], [B*S][CAP]
    int* xent;                       // [DS_X_CAP][2] (global row b*L+i, column j)
    float* xcf;                      // [DS_X_CAP] exact confidence of the entry
    int *rlist, *clist;              // [B][DS_XL_PCAP] row / column indices within the pair
    int* rdec_j;                     // [B*L] exactly re-decided rows: the row's best column ...
    float* rdec_cf;                  // [B*L] ... and its exact confidence
    _Float16 *imgA, *imgB;           // tile images [B][NIB][C/32][4 kg][2 parts][128 rows][8]
};

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// C == 0: the exact path's workspace; C > 0: plus the split path's buffers
static inline size_t ds_carve(DsWs* w, char* base, int B, int L, int S, int C) {
    const size_t NJB = (S + DS_BN - 1) / DS_BN, NIB = (L + DS_BM - 1) / DS_BM;
    const size_t Lp = NIB * DS_BM, Sp = NJB * DS_BN;
    size_t off = 0;
#define CARVE(field, type, count)                                   \
    do {                                                            \
        if (w) w->field = reinterpret_cast<type*>(base + off);      \
        off += align256(sizeof(type) * (size_t)(count));            \
    } while (0)
    CARVE(rp_m, float, (size_t)B * NJB * L); CARVE(rp_s, float, (size_t)B * NJB * L); CARVE(rp_a, int, (size_t)B * NJB * L);
    CARVE(cp_m, float, (size_t)B * NIB * S); CARVE(cp_s, float, (size_t)B * NIB * S); CARVE(cp_a, int, (size_t)B * NIB * S);
    CARVE(rmax, float, (size_t)B * L); CARVE(rsum, float, (size_t)B * L);
    CARVE(cmax, float, (size_t)B * S); CARVE(csum, float, (size_t)B * S);
    CARVE(rbest, unsigned long long, (size_t)B * L); CARVE(cbest, unsigned long long, (size_t)B * S);
    CARVE(fl_rn, int, (size_t)B * NJB); CARVE(fl_cn, int, (size_t)B * NIB); CARVE(fl_tn, int, 4);
    if (C > 0) {
        CARVE(rcnt, int, (size_t)B * L); CARVE(ccnt, int, (size_t)B * S);
        CARVE(namax, unsigned, B); CARVE(nbmax, unsigned, B); CARVE(ovf, int, 1);
        CARVE(xcnt, int, 4); CARVE(xln, int, 2 * (size_t)B); CARVE(rneed, unsigned char, (size_t)B * L); CARVE(cneed, unsigned char, (size_t)B * S);
        CARVE(rdec, unsigned char, (size_t)B * L);
    }
    if (w) w->zero_end = base + off;
    CARVE(flags, unsigned char, (size_t)B * L); CARVE(jsel, int64_t, (size_t)B * L); CARVE(csel, float, (size_t)B * L);
    CARVE(blk, int, ((size_t)B * L + 1023) / 1024 + 8);
    CARVE(fl_r, int, (size_t)B * NJB * L); CARVE(fl_c, int, (size_t)B * NIB * S); CARVE(fl_t, int, 2 * (size_t)B * NJB * NIB + 8);
    if (C > 0) {
        CARVE(na, float, (size_t)B * Lp); CARVE(nb, float, (size_t)B * Sp);
        CARVE(fa, float, (size_t)B * Lp); CARVE(fb, float, (size_t)B * Sp);
        CARVE(exA, int, (size_t)B * Lp); CARVE(exB, int, (size_t)B * Sp);
        CARVE(rthr, float, (size_t)B * L); CARVE(cthr, float, (size_t)B * S);
        CARVE(rcand, int, (size_t)B * L * DS_CAND_CAP); CARVE(ccand, int, (size_t)B * S * DS_CAND_CAP);
        CARVE(xent, int, 2 * DS_X_CAP); CARVE(xcf, float, DS_X_CAP); CARVE(rlist, int, (size_t)B * DS_XL_PCAP); CARVE(clist, int, (size_t)B * DS_XL_PCAP);
        CARVE(rdec_j, int, (size_t)B * L); CARVE(rdec_cf, float, (size_t)B * L);
        CARVE(imgA, _Float16, (size_t)B * Lp * C * 2); CARVE(imgB, _Float16, (size_t)B * Sp * C * 2);
    }
#undef CARVE
    return off;
}

// Tile epilogue of both GEMM kernels.  acc[ti][tj] = the wave's 64 x 64 sub-tile in the 32x32 MFMA C/D layout
// (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  Works from the accumulator registers (a version that
// parked the tile in LDS and looped over it cost as much as the whole fp32 MFMA loop at one workgroup per CU):
//   1. scale, apply the padding mask, store the tile straight from registers (128-B coalesced half-waves);
//   2. column (max, [first argmax,] sum exp) over the wave's 64 rows: in-lane scan + one xor-32 exchange;
//   3. row statistics over the wave's 64 columns: 32-row slabs through a wave-private LDS region, lane <-> row;
//   4. the two waves sharing rows / columns combine through a small LDS exchange -> one partial per 128-wide block.
// SPLIT: x = acc * facA[row] * facB[col] (factors staged in LDS), no row argmax (the indices come from the exact re-decision of
// the near-tie candidates); otherwise
// x = acc / T and the first argmax is tracked in both directions.
// `scratch` >= 4*32*65 + 2*2*128*3 floats, free of live data (caller has synchronised the workgroup).
template <bool RECIP, bool SPLIT, bool STORE = true>   // STORE == false (split path only): the matrix is not written
__device__ __forceinline__ void ds_tile_epilogue(f32x16 (&acc)[2][2], float* scratch, const float* facA, const float* facB,
                                                 const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1,
                                                 float* __restrict__ sim, const DsWs& w, int b, int tI, int tJ, int L, int S,
                                                 float T, float invT, int NJB, int NIB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int i0 = tI * DS_BM, j0 = tJ * DS_BN;
    float* wl = scratch + wave * (32 * 65);            // wave-private [32][65]
    float* rowx = scratch + 4 * 32 * 65;               // [2 wc][128 rows][3]
    float* colx = rowx + 2 * 128 * 3;                  // [2 wr][128 cols][3]
    const int hi = lane >> 5, ln = lane & 31;
    bool colok[2];
    unsigned char m1v[2] = {1, 1};
    float fbv[2] = {1.f, 1.f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int gj = j0 + wc * 64 + tj * 32 + ln;
        colok[tj] = gj < S;
        if (mask0 && colok[tj]) m1v[tj] = mask1[(size_t)b * S + gj];
        if (SPLIT) fbv[tj] = facB[wc * 64 + tj * 32 + ln];
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int gi = i0 + lr;
            const bool rowok = gi < L;
            const bool m0v = (mask0 && rowok) ? mask0[(size_t)b * L + gi] != 0 : true;
            const float fav = SPLIT ? facA[lr] : 0.f;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                float x = SPLIT ? __fmul_rn(__fmul_rn(acc[ti][tj][r], fav), fbv[tj]) : div_scalar<RECIP>(acc[ti][tj][r], T, invT);
                if (mask0 && !(m0v && m1v[tj])) x = NEG_FILL;
                const bool ok = rowok && colok[tj];
                if (STORE && ok) sim[((size_t)b * L + gi) * S + j0 + wc * 64 + tj * 32 + ln] = x;
                acc[ti][tj][r] = ok ? x : -INFINITY;  // out-of-range entries never win a max and add exp(-inf) = 0
            }
        }
    // ---- 2. columns
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        float m = -INFINITY; int am = 0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            if (SPLIT) {
                float g16 = acc[ti][tj][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) g16 = fmaxf(g16, acc[ti][tj][r]);
                m = fmaxf(m, g16);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // (ti, r>>2, r&3) ascending == row ascending for this half-wave
                    const float x = acc[ti][tj][r];
                    if (x > m) { m = x; am = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }
                }
            }
        }
        const float pm = __shfl_xor(m, 32);
        if (SPLIT) m = fmaxf(m, pm);
        else {
            const int pa = __shfl_xor(am, 32);
            if (pm > m || (pm == m && pa < am)) { m = pm; am = pa; }
        }
        float sm = 0.f;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += __expf(acc[ti][tj][r] - m);
        sm += __shfl_xor(sm, 32);
        if (hi == 0) {
            float* o = colx + (wr * 128 + wc * 64 + tj * 32 + ln) * 3;
            o[0] = m; o[1] = sm; o[2] = __int_as_float(wr * 64 + am);
        }
    }
    // ---- 3. rows
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) wl[((r & 3) + 8 * (r >> 2) + 4 * hi) * 65 + tj * 32 + ln] = acc[ti][tj][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* rp = wl + ln * 65 + hi * 32;   // lane <-> (row ln, column half hi)
        float v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = rp[c];
        float m = -INFINITY; int am = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            if (SPLIT) m = fmaxf(m, v[c]);
            else if (v[c] > m) { m = v[c]; am = hi * 32 + c; }
        }
        const float pm = __shfl_xor(m, 32);
        if (SPLIT) m = fmaxf(m, pm);
        else {
            const int pa = __shfl_xor(am, 32);
            if (pm > m || (pm == m && pa < am)) { m = pm; am = pa; }
        }
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) sm += __expf(v[c] - m);
        sm += __shfl_xor(sm, 32);
        if (hi == 0) {
            float* o = rowx + (wc * 128 + wr * 64 + ti * 32 + ln) * 3;
            o[0] = m; o[1] = sm; o[2] = __int_as_float(wc * 64 + am);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    // ---- 4. combine the two waves that share a row (wc = 0,1) / a column (wr = 0,1); the lower block wins ties
    {
        const float* x0 = (tid < 128 ? rowx : colx) + (tid & 127) * 3;
        const float* x1 = x0 + 128 * 3;
        const float ma = x0[0], mb = x1[0];
        const float mm = mb > ma ? mb : ma;
        const int aa = __float_as_int(mb > ma ? x1[2] : x0[2]);
        float tot = 0.f;
        if (ma > -INFINITY) tot += x0[1] * __expf(ma - mm);
        if (mb > -INFINITY) tot += x1[1] * __expf(mb - mm);
        if (tid < 128) {
            if (i0 + tid < L) {
                const size_t o = ((size_t)b * NJB + tJ) * L + i0 + tid;
                w.rp_m[o] = mm; w.rp_s[o] = tot;
                if (!SPLIT) w.rp_a[o] = j0 + aa;
            }
        } else if (j0 + tid - 128 < S) {
            const size_t o = ((size_t)b * NIB + tI) * S + j0 + tid - 128;
            w.cp_m[o] = mm; w.cp_s[o] = tot;
            if (!SPLIT) w.cp_a[o] = i0 + aa;
        }
    }
}

// ds_split.hip
int ds_split_launch(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, const DsWs& w, int B, int L, int S,
                    int C, float temperature, int recip, hipStream_t s);
int ds_gemm16_launch(const uint8_t* mask0, const uint8_t* mask1, float* sim, const DsWs& w, int B, int L, int S, int C, int store, hipStream_t s);
int ds_flagged_launch(const float* feat0, const float* feat1, const DsWs& w, int B, int L, int S, int C, float thr, float kthr, hipStream_t s);
int ds_flag_lists_launch(const DsWs& w, int B, int L, int S, float thr, int exact_rows_only, hipStream_t s);
int ds_fix_launch(const float* feat0, const float* feat1, const DsWs& w, int B, int L, int S, int C, float temperature, int recip,
                  int64_t* next_idx01, int64_t* next_idx10, hipStream_t s);
int ds_xdecide_launch(const float* feat0, const float* feat1, const uint8_t* mask0, const uint8_t* mask1, const DsWs& w, int B, int L,
                      int S, int C, float temperature, int recip, float thr, float* next_conf01, float* next_conf10, hipStream_t s);

// Relative band inside which an approximate (split-path) confidence cannot be ordered against another one, or against thr, with
// certainty: conf = p01 * p10, each p = exp(x - max) / sum with every logit of the row / column off by at most
// e = 2^-15 |a||b| / (C T) (the pair's largest norms: namax * nbmax) -> 4 e per factor, 8 e for the product, plus the two paths'
// different summation orders (~1e-6).  `kthr` = 2^-14 / T (= 2 e per unit norm product).  Entries further apart than 2 * band
// compare the same way in the exact path.
#define DS_BAND_MAX 0.04f   // 2 band must stay below the 0.1 margin of cmin = 0.9 thr (ds_xnear_kernel raises the exact fallback beyond it)
__device__ __forceinline__ float ds_conf_band(float kthr, unsigned namax_bits, unsigned nbmax_bits) {
    return 4.25f * kthr * __uint_as_float(namax_bits) * __uint_as_float(nbmax_bits) + 4e-6f;
}
// appends (global row, column) to the borderline list; overflow raises the fallback flag (the exact passes then decide everything)
__device__ __forceinline__ void ds_x_append(const DsWs& w, int ro, int j) {
    const int slot = atomicAdd(w.xcnt, 1);
    if (slot < DS_X_CAP) { w.xent[2 * slot] = ro; w.xent[2 * slot + 1] = j; } else *w.ovf = 1;
}

}  // namespace casmtr
