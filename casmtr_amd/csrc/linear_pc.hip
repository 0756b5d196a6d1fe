// The split-f16 projections as a PERSISTENT kernel whose waves have two roles (round 6, third split kernel), with the token pyramid of
// QuadtreeAttention.forward (src/model/modules/quadtree_attention.py:78-88: q/k/v = conv1x1(x), then F.avg_pool2d(.., 2, 2) twice)
// produced on the way out.
//
// Where the earlier split kernels (callers.hip) spend their time, measured with tools/lin_time.py on isolated launches (16 images of
// 104 x 104 tokens, C = 256, q/k/v from one x): 340-370 us against 130 us for a device copy of the same bytes, and the parts -- rows in,
// MFMAs + weight fragments, stores -- ADD UP: a workgroup loads, multiplies and stores in turn, the workgroups of a CU in step.  A wave
// that stores and then loads weight fragments also waits for its stores' acknowledgements: gfx9 counts both with one in-order vmcnt.
// Here one workgroup of eight waves per CU walks the row blocks, and the waves that multiply never touch memory except for weights:
//   waves 4-7 (helpers), rows in: read the NEXT block's 64 (K = 256) / 128 (K = 128) activation rows from HBM -- two adjacent 16-byte
//       loads per lane and 128-byte line, four lanes per row -- find each row's exponent, split into f16 hi / lo in the registers the
//       floats arrived in, and copy them into the operand buffer in the window between two barriers.
//   waves 0-3 (multiply): wave c owns a quarter of the output columns of every job (problem x column group) on these rows: per k16
//       stage 4 ds_read_b128 (all rows, next stage's read under this stage's MFMAs) + the wave's own weight fragments straight from the
//       prepared image in the L2 (no other wave reads them: half the L2 -> register traffic of linear16s_kernel's 2 x 2 wave grid),
//       prefetched three stages ahead through four register sets -- across jobs and blocks --, 12 MFMAs: the same products in the same
//       order as linear16s_kernel / linear16_kernel, results bit-identical.  A finished tile is scaled (+ bias; the column factors of
//       all problems sit in LDS) and written to an LDS out tile.
//   waves 4-7 (helpers), tiles out: read the out tile row-wise (16 bytes per lane) and store it: token-major rows 1 KB per instruction, or the
//       quad-major layout of the attention kernels in 128-byte lines -- and, because a lane then holds a quad's four children, its
//       level-1 quad's four quads and so the 4 x 4 tokens of a level-2 token, the two pooled pyramid levels come out of three adds and a
//       multiply per value in registers, in torch's order (((c0 + c1) + c2) + c3) * 0.25 = quad_pool_kernel's: bit-identical to
//       projecting and pooling in three launches, without reading a projected level back (0.83 ms per step of the callers leg).
// Hand-offs: four monotonic counters in LDS (operand buffer free / filled per block, out tile filled / free per job), not s_barrier -- see
// pc_signal / pc_wait below; the multiply waves run up to one job ahead of the stores.  Rows of a block in quad mode are the 8 x 8 tokens
// of one image tile in Z-order: local row = 32 ty + 16 tx + 8 qy + 4 qx + 2 cy + cx for token (4 ty + 2 qy + cy, 4 tx + 2 qx + cx).
//
// Measured (profiles/r06_lin_time.txt; one box, us per launch: this kernel / linear16s_kernel / a copy of the bytes): q,k,v 323 / 350 /
// 127; with the pyramid 302 against 350 + 117 + 26 for projection + two pooling launches; merge projection 115 / 139 / 67; cascade
// q | k,v 225 / 274 / 160; cascade merge 86 / 98 / 67.  (With s_barrier hand-offs: 335 / 132 / 255 / 104.)  Without stores 239, without
// row loads 254, with neither 218: the multiply waves' own pace -- 62 % of the matrix pipe at the clock it really runs at -- is what is
// left.  Two findings on the way: (1) tools/probes/mfma16_rate.hip -- with every SIMD issuing v_mfma_f32_32x32x16_f16 on random operands
// the chip holds 1.3-1.55 GHz (32 cycles per MFMA at every accumulator count; one CU alone: 1.95 GHz), so the three products of the q,k,v
// launch are 134 us of matrix pipe, as long as its bytes take; (2) a run-time `if` around a load inside the k-loop, or a conditionally
// loaded register (bias), makes the compiler's wait-count insertion fall back to vmcnt(0) / vmcnt(1) right behind the load: experiment
// switches live outside the hot loop only, and the column factors / biases come from LDS.
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"
#include "linear_pc.hpp"
#include "../../include/casmtr_hip.h"

using namespace casmtr;

namespace {

typedef _Float16 l16_h8 __attribute__((ext_vector_type(8)));

// Hand-offs between the two kinds of waves go through four monotonic counters in LDS instead of s_barrier: a barrier makes every wave wait
// for the slowest one at every job, so a helper whose stores are queued behind a busy memory system held the matrix pipe up twice per job.
// With counters a wave waits only for the event it needs (a tile read, a tile written, the operand buffer free / filled), and the multiply
// waves run up to one job ahead of the stores.  signal: this wave's LDS traffic is complete (lgkmcnt(0): LDS instructions of a wave complete
// in order), then one lane adds 1; wait: spin on the counter with s_sleep.
__device__ __forceinline__ void pc_signal(int* c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pc_wait(const int* c, int target) {
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(c)) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// RT: 64-row sub-blocks per block; NT: 32-column MFMA tiles per multiply wave and job (a job = 128 NT columns of one problem)
template <int K, int RT, int NT, bool QUADS>
__global__ __launch_bounds__(512, 1) void linear16p_kernel(const Lin16pArgs a) {
    constexpr int KS = K / 32, S2 = 2 * KS, RB = 64 * RT, NTI = 2 * RT;
    constexpr int PLANE = 2 * RB * 16 + 32;   // [hi | lo][RB rows][16 B]; + 32 B: the four planes a helper wave writes with one instruction start 8 banks apart
    constexpr int CHUNK = 4 * PLANE;          // 32 channels: k-groups (8 channels) 0..3
    constexpr int CW = 32 * NT;               // output columns per multiply wave and job
    constexpr int TW = 4 * CW;                // columns of a job = of the out tile
    constexpr int HT = TW / 32;               // heads per out tile
    extern __shared__ __attribute__((aligned(16))) char lds_pc[];
    char* As = lds_pc;
    float* facA = reinterpret_cast<float*>(As + KS * CHUNK);   // [2 parities][RB]
    float* Ot = facA + 2 * RB;                                  // out tile [RB][TW]
    float* facW = Ot + RB * TW;                                 // [nprob][N] 2^e_n, then [nprob][N] bias (0 where a problem has none)
    int* cnt = reinterpret_cast<int*>(facW + 2 * L16P_MAXFAC);  // cA, cB, cW, cR (16 bytes apart)
    int* cA = cnt;        // + 1 per multiply wave that has read the last stage of a block   (operand buffer free)
    int* cB = cnt + 4;    // + 1 per helper wave that has written its share of a block        (operand buffer filled)
    int* cW = cnt + 8;    // + 1 per multiply wave that has written its strip of an out tile (tile filled)
    int* cR = cnt + 12;   // + 1 per helper wave that has read its share of an out tile       (tile free)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NU = (a.nsub + RT - 1) / RT, total = a.ngroups * NU;
    const int nit = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nb = a.nbx * a.nby;     // tiles per image (quad mode)
    const int ncg = a.N / TW;         // column groups per problem
    auto unit_of = [&](int it) { return (int)blockIdx.x + it * (int)gridDim.x; };
    auto njobs = [&](int u) { return a.count[u / NU] * ncg; };
    // The column factors and biases of every problem, once, into LDS: a multiply wave that fetched its job's factors from memory would
    // have to wait for them behind its own weight prefetch (one in-order vmcnt) -- the compiler settles for vmcnt(2) at the end of every
    // job, which drains the three stages just prefetched for the next one.
    for (int i = tid; i < a.nprob * a.N; i += 512) {
        const int pq = i / a.N, c = i - pq * a.N;
        facW[i] = a.wfac[pq][c];
        facW[a.nprob * a.N + i] = a.bias[pq] ? a.bias[pq][c] : 0.f;
    }
    if (tid < 16) cnt[tid] = 0;
    __syncthreads();

    if (wave >= 4) {
        // ================================================================================ helper waves: rows in, tiles out
        const int ht = tid - 256;                 // 0..255
        const int prow = ht >> 2, p = ht & 3;     // fetch: four lanes per row, rows prow (+ 64)
        // a line's 8 floats as they arrive; after convert() [0] = their 8 f16 hi parts, [1] = the 8 lo parts: the split takes the registers
        // the floats came in (as separate arrays the two generations were allocated side by side: spills under the 168-register cap of
        // twelve waves per CU).  One integer type throughout, so that the array stays in registers.
        u32x4 raw[RT][KS][2];
        float fexp[RT];
        auto issue = [&](int u) {
            const int g = u / NU, ub = u - g * NU;
#pragma unroll
            for (int n = 0; n < RT; ++n) {
                int sb = ub * RT + n;
                if (sb >= a.nsub) sb = a.nsub - 1;       // a tail unit's missing half: any readable rows, never stored
                size_t gi;
                if constexpr (QUADS) {
                    const int img = sb / nb, rem = sb - img * nb, by = rem / a.nbx, bx = rem - by * a.nbx;
                    int y = by * 8 + ((prow >> 5) & 1) * 4 + ((prow >> 3) & 1) * 2 + ((prow >> 1) & 1);
                    int x = bx * 8 + ((prow >> 4) & 1) * 4 + ((prow >> 2) & 1) * 2 + (prow & 1);
                    y = y < a.h ? y : a.h - 1;
                    x = x < a.w ? x : a.w - 1;
                    gi = ((size_t)img * a.h + y) * a.w + x;
                } else {
                    const int r = sb * 64 + prow;
                    gi = (size_t)(r < a.M ? r : a.M - 1);
                }
                const unsigned* ap = reinterpret_cast<const unsigned*>(a.x[g] + gi * K + p * 8);
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    raw[n][i][0] = *reinterpret_cast<const u32x4*>(ap + 32 * i);
                    raw[n][i][1] = *reinterpret_cast<const u32x4*>(ap + 32 * i + 4);
                }
            }
        };
        auto convert = [&]() {
#pragma unroll
            for (int n = 0; n < RT; ++n) {
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < KS; ++i)
#pragma unroll
                    for (int c = 0; c < 8; ++c) mx = fmaxf(mx, fabsf(__uint_as_float(raw[n][i][c >> 2][c & 3])));
                mx = fmaxf(mx, dpp_f32<0xB1>(mx));   // lane ^ 1
                mx = fmaxf(mx, dpp_f32<0x4E>(mx));   // lane ^ 2
                const int e = (mx > 0.f && mx < INFINITY) ? ilogbf(mx) - 9 : 0;   // ds_rownorm_kernel's rule: largest |element| -> [512, 1024)
                const float sc = ldexpf(1.0f, -e);
                fexp[n] = ldexpf(1.0f, e);
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    unsigned hb[8], lb[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float xn = __uint_as_float(raw[n][i][c >> 2][c & 3]) * sc;   // exact: a power of two
                        const _Float16 hh = (_Float16)xn;
                        hb[c] = __builtin_bit_cast(unsigned short, hh);
                        lb[c] = __builtin_bit_cast(unsigned short, (_Float16)(xn - (float)hh));
                    }
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        raw[n][i][0][d] = hb[2 * d] | (hb[2 * d + 1] << 16);
                        raw[n][i][1][d] = lb[2 * d] | (lb[2 * d + 1] << 16);
                    }
                }
            }
        };
        // job j's out tile -> registers (tile_read), registers -> memory (tile_store; the tile may be overwritten in between)
        f32x4 tv[16];
        auto tile_read = [&](int u) {
            const int ub = u - (u / NU) * NU;
            if constexpr (!QUADS) {
                constexpr int C4 = TW / 4;
#pragma unroll
                for (int k = 0; k < RB * C4 / 256; ++k) {
                    const int c = ht + 256 * k;
                    tv[k] = *reinterpret_cast<const f32x4*>(Ot + (c / C4) * TW + (c % C4) * 4);
                }
            } else {
                // lane <-> (level-2 token of the block, head, 16-byte piece of the head's 32 channels): RT * TW combinations
                const int c = ht, piece = c & 7, hl = (c >> 3) % HT, t2u = c / (8 * HT);
                const int ty = (t2u >> 1) & 1, tx = t2u & 1;
                if (c < RT * TW && ub * RT + (t2u >> 2) < a.nsub) {
                    const float* src = Ot + ((t2u >> 2) * 64 + ty * 32 + tx * 16) * TW + hl * 32 + piece * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = *reinterpret_cast<const f32x4*>(src + r * TW);   // r = 4 t1 + c0
                }
            }
        };
        auto tile_store = [&](int u, int j) {
            if (a.xflags & 1) return;
            const int g = u / NU, ub = u - g * NU;
            const int pi = j / ncg, cg = j - pi * ncg, pp = a.first[g] + pi;
            if constexpr (!QUADS) {
                constexpr int C4 = TW / 4;
                float* __restrict__ Y = a.y0[pp] + (size_t)cg * TW;
#pragma unroll
                for (int k = 0; k < RB * C4 / 256; ++k) {
                    const int c = ht + 256 * k, col4 = c % C4, row = c / C4;
                    const int gr = (ub * RT + row / 64) * 64 + (row & 63);
                    if (gr < a.M) *reinterpret_cast<f32x4*>(Y + (size_t)gr * a.N + col4 * 4) = tv[k];
                }
            } else {
                const int wq = a.w >> 1, Lq = (a.h >> 1) * wq, Hh = a.N >> 5;
                const int c = ht, piece = c & 7, hl = (c >> 3) % HT, t2u = c / (8 * HT);
                const int sb = ub * RT + (t2u >> 2), ty = (t2u >> 1) & 1, tx = t2u & 1;
                if (c >= RT * TW || sb >= a.nsub) return;
                const int img = sb / nb, rem = sb - img * nb, by = rem / a.nbx, bx = rem - by * a.nbx;
                const int head = cg * HT + hl;
                const int T2Y = by * 2 + ty, T2X = bx * 2 + tx;
                float* y0h = a.y0[pp] + ((size_t)img * Hh + head) * Lq * 128 + piece * 4;
                f32x4 p1[4];
                bool ok0 = false;
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1) {   // the four quads of the level-1 quad, (qy, qx) = (t1 >> 1, t1 & 1)
                    const int QY = 2 * T2Y + (t1 >> 1), QX = 2 * T2X + (t1 & 1);
                    const bool ok = 2 * QY < a.h && 2 * QX < a.w;
                    if (t1 == 0) ok0 = ok;
                    const f32x4 (&v)[4] = reinterpret_cast<const f32x4 (&)[4]>(tv[t1 * 4]);
                    if (ok) {
                        float* d = y0h + (size_t)(QY * wq + QX) * 128;
#pragma unroll
                        for (int c0 = 0; c0 < 4; ++c0) *reinterpret_cast<f32x4*>(d + c0 * 32) = v[c0];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) p1[t1][i] = (((v[0][i] + v[1][i]) + v[2][i]) + v[3][i]) * 0.25f;
                    if (a.y1[pp] && ok) {
                        if (a.y1_tokens)
                            *reinterpret_cast<f32x4*>(a.y1[pp] + ((size_t)img * Lq + QY * wq + QX) * a.N + head * 32 + piece * 4) = p1[t1];
                        else
                            *reinterpret_cast<f32x4*>(a.y1[pp] + (((size_t)img * Hh + head) * (Lq >> 2) + T2Y * (wq >> 1) + T2X) * 128 + t1 * 32 +
                                                      piece * 4) = p1[t1];
                    }
                }
                if (a.y2[pp] && ok0) {   // h, w multiples of 4: a level-2 token's 4 x 4 tokens are all inside the image or all outside
                    f32x4 p2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) p2[i] = (((p1[0][i] + p1[1][i]) + p1[2][i]) + p1[3][i]) * 0.25f;
                    *reinterpret_cast<f32x4*>(a.y2[pp] + ((size_t)img * (Lq >> 2) + T2Y * (a.w >> 2) + T2X) * a.N + head * 32 + piece * 4) = p2;
                }
            }
        };
        int tiles = 0;   // out tiles taken so far
        auto drain = [&](int u, int j) {
            pc_wait(cW, 4 * (tiles + 1));   // the four strips of the tile are written
            tile_read(u);
            pc_signal(cR);                  // ... and read: the multiply waves may write the next one
            ++tiles;
            tile_store(u, j);
        };
        issue(unit_of(0));
        convert();
        for (int it = 0; it < nit; ++it) {
            const int u = unit_of(it), nj = njobs(u);
            pc_wait(cA, 4 * it);   // the multiply waves have read the last stage of the previous block
#pragma unroll
            for (int n = 0; n < RT; ++n) {
#pragma unroll
                for (int i = 0; i < KS; ++i) {   // line i = chunk i, this lane's 8 channels = k-group p
                    char* d = As + i * CHUNK + p * PLANE + (n * 64 + prow) * 16;
                    *reinterpret_cast<u32x4*>(d) = raw[n][i][0];
                    *reinterpret_cast<u32x4*>(d + RB * 16) = raw[n][i][1];
                }
                if (p == 0) facA[(it & 1) * RB + n * 64 + prow] = fexp[n];
            }
            pc_signal(cB);         // block `it` is in the buffer
            // the previous block's last tile: taken only now, so that the buffer was refilled while the multiply waves wrote that tile
            if (it > 0) { const int up = unit_of(it - 1); drain(up, njobs(up) - 1); }
            const bool more = it + 1 < nit;
            if (more && !(a.xflags & 4)) issue(unit_of(it + 1));
            for (int j = 0; j + 1 < nj; ++j) drain(u, j);
            if (more) convert();   // (the loads have had the block's first jobs to arrive)
        }
        { const int ul = unit_of(nit - 1); drain(ul, njobs(ul) - 1); }
        return;
    }

    // ==================================================================================== multiply waves
    const int hi = lane >> 5, ln = lane & 31;
    const char* xa = As + hi * PLANE + ln * 16;   // stage s: + (s >> 1) * CHUNK + 2 (s & 1) * PLANE; 32-row tile ti: + ti * 512; lo part: + RB * 16
    auto wptr = [&](int u, int j) -> const char* {      // job j of unit u: this lane's 16 bytes of stage 0, tile 0, hi part
        const int g = u / NU, pi = j / ncg, cg = j - pi * ncg;
        const int col0 = cg * TW + wave * CW;
        return a.wimg[a.first[g] + pi] + (size_t)(col0 >> 7) * KS * 16384 + hi * 4096 + ((col0 & 127) + ln) * 16;
    };
    l16_h8 wh[4][NT], wl[4][NT];
    auto loadw = [&](const char* q, l16_h8 (&h)[NT], l16_h8 (&l)[NT]) {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            h[tj] = *reinterpret_cast<const l16_h8*>(q + tj * 512);
            l[tj] = *reinterpret_cast<const l16_h8*>(q + tj * 512 + 2048);
        }
    };
    const char* q = wptr(unit_of(0), 0);
    loadw(q, wh[0], wl[0]);
    loadw(q + 8192, wh[1], wl[1]);
    loadw(q + 2 * 8192, wh[2], wl[2]);
    int tiles = 0;   // out tiles written so far
    for (int it = 0; it < nit; ++it) {
        const int u = unit_of(it), g = u / NU, nj = njobs(u);
        pc_wait(cB, 4 * (it + 1));   // block `it` is in the operand buffer
        for (int j = 0; j < nj; ++j) {
            const int pi = j / ncg, cg = j - pi * ncg, pp = a.first[g] + pi, col0 = cg * TW + wave * CW;
            // the job after this one (its first three stages are prefetched under this job's last three)
            const char* qn = q;
            if (j + 1 < nj) qn = wptr(u, j + 1);
            else if (it + 1 < nit) qn = wptr(unit_of(it + 1), 0);
            f32x16 acc[NTI][NT];
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
            l16_h8 ah[2][NTI], al[2][NTI];
            auto loada = [&](int s, l16_h8 (&h)[NTI], l16_h8 (&l)[NTI]) {
                const char* pa = xa + (s >> 1) * CHUNK + 2 * (s & 1) * PLANE;
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti) {
                    h[ti] = *reinterpret_cast<const l16_h8*>(pa + ti * 512);
                    l[ti] = *reinterpret_cast<const l16_h8*>(pa + ti * 512 + RB * 16);
                }
            };
            auto stage = [&](const l16_h8 (&xh)[NTI], const l16_h8 (&xl)[NTI], const l16_h8 (&h)[NT], const l16_h8 (&l)[NT]) {
                // small terms first (linear16_kernel's / ds_gemm16_kernel's order)
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[ti], h[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ti], l[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ti], h[tj], acc[ti][tj], 0, 0, 0);
            };
            loada(0, ah[0], al[0]);
            // No run-time switches in this loop (a conditional load makes the compiler wait for every weight fragment right behind its load),
            // and no peeled first group either (starting from a literal-zero accumulator would save 64 v_mov per job, but in straight-line
            // code the scheduler sinks the prefetch loads down to their uses: vmcnt(0) again).
#pragma unroll 1
            for (int s0 = 0; s0 < S2; s0 += 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int s = s0 + jj, pf = s + 3;
                    const char* wp = pf < S2 ? q + (size_t)pf * 8192 : qn + (size_t)(pf - S2) * 8192;
                    loadw(wp, wh[(jj + 3) & 3], wl[(jj + 3) & 3]);
                    if (jj < 3 || s0 + 4 < S2) loada(s + 1, ah[(jj + 1) & 1], al[(jj + 1) & 1]);
                    stage(ah[jj & 1], al[jj & 1], wh[jj], wl[jj]);
                }
            }
            q = qn;
            if (j + 1 == nj) pc_signal(cA);   // this wave has read the block's last stage: the helpers may refill the buffer
            // y = acc * 2^e_m * 2^e_n (+ bias) -> out tile, once the helpers have taken the previous one
            const float* fa_par = facA + (it & 1) * RB;
            float fbv[NT], bj[NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                fbv[tj] = facW[pp * a.N + col0 + tj * 32 + ln];
                bj[tj] = facW[(a.nprob + pp) * a.N + col0 + tj * 32 + ln];
            }
            pc_wait(cR, 4 * tiles);
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 fa4 = *reinterpret_cast<const f32x4*>(fa_par + ti * 32 + 8 * rq + 4 * hi);
#pragma unroll
                    for (int c0 = 0; c0 < 4; ++c0) {
                        const int row = ti * 32 + 8 * rq + 4 * hi + c0;
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) {
                            const float v = (acc[ti][tj][4 * rq + c0] * fa4[c0]) * fbv[tj];
                            Ot[row * TW + wave * CW + tj * 32 + ln] = v + bj[tj];   // bj = 0 without a bias: v is never -0 (exact zeros are +0)
                        }
                    }
                }
            pc_signal(cW);
            ++tiles;
        }
    }
}

template <int K, int RT, int NT, bool QUADS>
int launch_pc(const Lin16pArgs& a, hipStream_t s) {
    constexpr int KS = K / 32, RB = 64 * RT, PLANE = 2 * RB * 16 + 32;
    constexpr size_t lds = (size_t)KS * 4 * PLANE + 2 * RB * 4 + (size_t)RB * 128 * NT * 4 + 2 * L16P_MAXFAC * 4 + 64;
    static int cache[CASMTR_MAX_DEVICES];
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear16p_kernel<K, RT, NT, QUADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = 0;
    const int rc = resident_workgroups(cache, linear16p_kernel<K, RT, NT, QUADS>, 512, lds, &grid);
    if (rc) return rc;
    const int NU = (a.nsub + RT - 1) / RT, total = a.ngroups * NU;
    if (grid > total) grid = total;
    prof_symbol_args(CASMTR_PROF_LINEAR, "<%d,%d,%d,%s>", K, RT, NT, QUADS ? "true" : "false");
    CASMTR_LAUNCH_TIMED(CASMTR_PROF_LINEAR, (linear16p_kernel<K, RT, NT, QUADS>), dim3((unsigned)grid), dim3(512), lds, s, a);
    CASMTR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

namespace casmtr {
// K in {128, 256}; N % 256 == 0, or K == 128 and N % 128 == 0; nprob * N <= L16P_MAXFAC (else CASMTR_ERR_UNSUPPORTED: the caller runs
// linear16s_kernel)
int linear16p_launch(const Lin16pArgs& a, int K, hipStream_t s) {
    const bool quads = a.w > 0, wide = a.N % 256 == 0;
    if (a.nprob * a.N > L16P_MAXFAC) return CASMTR_ERR_UNSUPPORTED;
    if (K == 256 && wide) return quads ? launch_pc<256, 1, 2, true>(a, s) : launch_pc<256, 1, 2, false>(a, s);
    if (K == 128 && wide) return quads ? launch_pc<128, 1, 2, true>(a, s) : launch_pc<128, 1, 2, false>(a, s);
    if (K == 128 && a.N % 128 == 0) return quads ? launch_pc<128, 2, 1, true>(a, s) : launch_pc<128, 2, 1, false>(a, s);
    return CASMTR_ERR_UNSUPPORTED;
}
}  // namespace casmtr
