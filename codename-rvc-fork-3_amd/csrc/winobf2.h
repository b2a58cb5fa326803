// winobf2.hip (K3y, one transform point per wave): parameters, the input transforms' coefficient tables, the block geometry in LDS
// (round 4's balanced form K3z, winobf3.hip, shared this header: parity-green, 19 % slower, removed in round 5 --
// profiles/r04_winobf3_balanced_ab.txt)
#pragma once
#include "conv.h"

namespace rvc {

struct Wbf2Params {
    const float *x = nullptr;        // [batch][c_in][L]
    const void *u = nullptr;         // winobf_pack_host's slab in its point-major form (128 rows per block)
    const float *bias = nullptr;     // [c_out]
    const float *res = nullptr;      // [batch][c_out][L] or null
    const float *accin = nullptr;    // [batch][c_out][L] or null
    float *y = nullptr;              // [batch][c_out][L]
    int c_in = 0, c_out = 0;
    int64_t L = 0;
    int dil = 1;
    int sb_per_block = 0;            // super-blocks (of d tiles) per block
    int n_tile_blocks = 0;           // blocks along time
    float slope = 1.f, out_scale = 1.f;
    int batch = 1;
};

constexpr int W2_MAX_DIL = 5;
constexpr int W2_NTH = 512, W2_BNT = 64, W2_CIC = 16, W2_CP = 8, W2_R = 4, W2_LOADER = 7;

typedef float w2_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w2_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w2_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ w2_f32x2 w2_lrelu2(w2_f32x2 v, float slope) {
    const w2_f32x2 sv = v * slope;
    return w2_f32x2{__builtin_fmaxf(v.x, sv.x), __builtin_fmaxf(v.y, sv.y)};
}
constexpr int W2_RSRC_FLAGS = 0x00020000;

// rows of the F(4,4) input transform B^T for the points 0, 1, -1, 1/2, -1/2, 2, inf (the expressions of wino.hip / winobf.hip
// multiplied out; every entry is exact in fp32): X_p = sum_n W2_BT[p][n] * x[n]
static __constant__ float W2_BT[7][8] = {
    {-0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f, 0.f},
    {0.f, 0.5f, 0.25f, -2.25f, -1.f, 1.f, 0.f, 0.f},
    {0.f, -0.5f, 0.75f, 1.75f, -3.f, 1.f, 0.f, 0.f},
    {0.f, 1.f, 1.5f, -2.f, -1.5f, 1.f, 0.f, 0.f},
    {0.f, -1.f, 2.5f, 0.f, -2.5f, 1.f, 0.f, 0.f},
    {0.f, 0.25f, 0.f, -1.25f, 0.f, 1.f, 0.f, 0.f},
    {0.f, -0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},
};
// ... and of F(4,3)'s (three taps: one group, six points 0, 1, -1, 2, -2, inf on a six-sample window; wino.hip's expressions)
static __constant__ float W2_BT3[6][8] = {
    {4.f, 0.f, -5.f, 0.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, -4.f, -4.f, 1.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, 4.f, -4.f, -1.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, -2.f, -1.f, 2.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, 2.f, -1.f, -2.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, 4.f, 0.f, -5.f, 0.f, 1.f, 0.f, 0.f},
};

template <int KW, int BM>
struct W2Geom {
    static constexpr int R = KW == 3 ? 3 : W2_R;                         // taps per group: F(4,3) for the 3-tap layers, F(4,4) groups otherwise
    static constexpr int NP = R + 3;                                     // transform points = samples of a window = compute waves
    static constexpr int G = (KW + R - 1) / R;
    static constexpr int C0 = (KW - 1) / 2;
    // one window's samples sit at offsets (n - C0) d, n = 0..NP-1: super-block steps MLO .. MHI
    static constexpr int MLO = -((C0 + 3) / 4);
    static constexpr int MHI = (NP - 1 - C0) / 4;                        // 1 for three taps (sample 4 d is the next super-block's first), else 0
    static_assert(NP - 1 - C0 >= 0 && MHI <= 1, "");
    static constexpr int WM = BM / 32, WN = W2_BNT / 32;
    static constexpr int XT = W2_BNT + (G - 1 - MLO + MHI) * W2_MAX_DIL;   // raw tiles per row: 64 + (G - 1) d windows + the -MLO d tiles in front (+ MHI d behind)
    static constexpr int XTS = ((XT - 12 + 31) / 32) * 32 + 12;          // row stride in float2, == 12 mod 32 (ds_write_b64 of 4 rows: 4 bank groups)
    static_assert(XTS >= XT + 2, "");
    static constexpr int RAW_BYTES = W2_CP * 4 * XTS * 8;               // one raw chunk; two buffers
    static constexpr int XB = W2_BNT + (G - 1) * W2_MAX_DIL;             // window fragments per (split, k half): the products read up to here
    static constexpr int B_PLANE = XB * 16;
    static constexpr int B_WAVE = 3 * 2 * B_PLANE;                      // [split][k half][window][8 bf16], one chunk of one point
    static constexpr int B_BYTES = NP * 2 * B_WAVE;                  // every compute wave: two chunks
    static constexpr int LOOP_BYTES = 2 * RAW_BYTES + B_BYTES;
    static constexpr int NJ = (4 * XT + 63) / 64;                       // staged samples per loader lane per channel row
    // epilogue, per pass of one row-block pair (64 channels x 64 columns): the 7 points' accumulators, then the output tile
    static constexpr int RED_BYTES = NP * 4 * 4096;
    static constexpr int YS = 4 * W2_BNT + 4;
    static constexpr int OUT_BYTES = 64 * YS * 4;
    static constexpr int EPI_BYTES = RED_BYTES > OUT_BYTES ? RED_BYTES : OUT_BYTES;
    static constexpr int LDS_BYTES = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};


}  // namespace rvc
