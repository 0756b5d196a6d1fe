// Argument block of the persistent producer / consumer projection kernel (linear_pc.hip), filled by the C ABI entries in callers.hip.
#pragma once
#include <hip/hip_runtime.h>

#define L16P_MAXP 8          // problems per launch (the q / k / v projections of both directions of a layer: 6)
#define L16P_MAXFAC 2048   // column factors + biases of all problems live in LDS: nprob * N <= this

struct Lin16pArgs {
    const float* x[L16P_MAXP];                  // distinct activation tensors ("groups"); group g serves problems first[g] .. + count[g]
    int first[L16P_MAXP], count[L16P_MAXP];
    const char* wimg[L16P_MAXP];                // per problem, in group order: prepared weight image (casmtr_linear_split_prep)
    const float* wfac[L16P_MAXP];               // [N] 2^e_n
    const float* bias[L16P_MAXP];               // nullable
    float* y0[L16P_MAXP];                       // token-major [M][N], or quad-major [B][N/32][(h/2)(w/2)][4][32] (w > 0)
    float* y1[L16P_MAXP];                       // nullable (quad mode): avg_pool2d(y0, 2, 2), quad-major or (y1_tokens) token-major
    float* y2[L16P_MAXP];                       // nullable (quad mode, y1 quad-major): avg_pool2d(y1, 2, 2), token-major
    int nprob, ngroups, nsub;                          // 64-row sub-blocks per activation tensor (quad mode: 8 x 8 token tiles)
    int M, N;
    int h, w, nbx, nby;                         // quad mode: token grid per image, tiles per image row / column; w == 0: token mode
    int y1_tokens;
    int xflags;                                 // experiment switches (CASMTR_LIN_FLAGS, tools/lin_time.py): 1 tiles not stored, 4 rows read once (switches inside the k-loop are not offered: a conditional load there makes the compiler wait for every weight fragment right behind its load)
};

namespace casmtr {
int linear16p_launch(const Lin16pArgs& a, int K, hipStream_t s);
}
