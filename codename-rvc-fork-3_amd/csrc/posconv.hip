// K14 -- HuBERT's convolutional position embedding (transformers' HubertPositionalConvEmbedding behind rvc/infer/pipeline.py:450:
// weight-normed Conv1d(D, D, 128, padding 64, groups 16) -> drop the last frame (HubertSamePadLayer) -> GELU) on the bf16 matrix
// cores with exact bf16x3 operands, over TIME-MAJOR frames x [T][D]:
//
//   y[t][g CG + o] = gelu( bias + sum_{k < 128} sum_{c < CG} W[g CG + o][c][k] * x[t + k - 64][g CG + c] ),   CG = D / groups = 48
//
// Per group this is a 48-row GEMM with K = 128 x 48 = 6144: the library path is an im2col of the whole tensor (236 MB written and read
// back) plus a GEMM whose tiles multiply mostly padding, 0.45 ms per 30 s clip; K12's 128-row blocks would multiply 2.7 x zeros.
// Here a block owns (group, 128 frames):
//   * the frames it needs -- 128 + 127 rows x 48 channels -- are read ONCE, split into three bf16 planes and stay in LDS for the whole
//     K loop (row stride 112 bytes, an odd multiple of 16: conflict-free ds_read_b128): the matrix instruction's window fragment for
//     tap k is the same rows shifted by k;
//   * the weights stream through a two-chunk LDS ring by LDS-DMA, one chunk = one tap = 3 steps of 16 channels x 2 row tiles x 3
//     splits x 1 KiB; the lanes of the second row tile that hold rows >= 48 point outside the buffer (they read zeros and fetch nothing);
//   * 8 waves = 2 row tiles x 4 column tiles of 32 x 32, 18 matrix instructions per wave and tap, one barrier per tap;
//   * epilogue: bias, erf GELU, 16-byte stores of 4 channels of a frame.
// 16 groups x 12 frame tiles = 192 blocks for a 30 s clip: one round of the CUs.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

struct PosConvParams {
    const float *x = nullptr;      // [T][D]
    const void *a = nullptr;       // slab [group][tap][chunk][row tile 2][split 3][lane 64][8 bf16]
    const float *bias = nullptr;   // [D] or null
    float *y = nullptr;            // [T][D]
    int64_t T = 0;
    int D = 0, groups = 0, taps = 0, pad = 0;
};

typedef float pc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pc_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pc_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pc_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) *pc_lptr_t;

constexpr int PC_BN = 128, PC_NTH = 512;

template <int CG>   // channels per group: 48 (HuBERT base) or 64
__global__ void __launch_bounds__(PC_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
posconv_kernel(const PosConvParams p) {
    constexpr int NCC = CG / 16;                 // 16-channel steps per tap
    constexpr int WS = CG * 2 + 16;              // window row stride in bytes (112 / 144: odd multiples of 16)
    constexpr int CHUNK = NCC * 6 * 1024;        // one tap of weight fragments
    constexpr int NP = NCC * 6, UPW = (NP + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 8);
    const int rt = wave & 1, ct = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int g = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * PC_BN;
    const int KT = p.taps, R = PC_BN + KT - 1;   // window rows
    unsigned char *const ring = pc_smem;                          // [2][CHUNK]
    unsigned char *const win = pc_smem + 2 * CHUNK;               // [split 3][R][WS]
    const int plane = R * WS;

    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, (int)((int64_t)p.groups * KT * CHUNK), 0x00020000);
    const int a_group = g * KT * CHUNK;
    // piece q of a chunk = (step cc, row tile, split); the second row tile's lanes with rows >= CG read outside the buffer: zeros, no fetch
    auto dma = [&](int k) __attribute__((always_inline)) {
        unsigned char *dst = ring + (k & 1) * CHUNK;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            int q = wave + 8 * i;
            if (q >= NP) q -= 8;                                  // same bytes to the same place
            const int prt = (q / 3) & 1;
            const bool dead = prt == 1 && l31 >= CG - 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (pc_lptr_t)(dst + q * 1024), 16, dead ? (int)0x80000000 : 16 * lane,
                                                     a_group + k * CHUNK + q * 1024, 0, 0);
        }
    };
    dma(0);

    // ---- the window: x[t0 - pad .. t0 - pad + R) of this group's CG channels, split exactly into three bf16 planes ----------------
    {
        const float *const xg = p.x + (int64_t)g * CG;
        constexpr int Q = CG / 4;
        for (int idx = tid; idx < R * Q; idx += PC_NTH) {
            const int r = idx / Q, c4 = idx - r * Q;
            const int64_t t = t0 - p.pad + r;
            const bool in = t >= 0 && t < p.T;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xg + (in ? t : 0) * p.D + 4 * c4);
            pc_f32x2 v0 = {in ? v.x : 0.f, in ? v.y : 0.f}, v1 = {in ? v.z : 0.f, in ? v.w : 0.f};
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, pc_bf16x2));
                const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, pc_bf16x2));
                *reinterpret_cast<pc_u32x2 *>(win + sp * plane + r * WS + c4 * 8) = pc_u32x2{w0, w1};
                v0 = v0 - pc_f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
                v1 = v1 - pc_f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
            }
        }
    }
    f32x16 acc2[2];                            // two accumulator chains (a matrix instruction then never waits for its predecessor), summed at the end
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][r] = acc2[1][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    const unsigned char *const a_lane = ring + rt * 3 * 1024 + lane * 16;
    const unsigned char *const b_lane = win + (ct * 32 + l31) * WS + half * 16;
    auto rd = [&](const unsigned char *ptr) __attribute__((always_inline)) {
        return __builtin_bit_cast(pc_bf16x8, *reinterpret_cast<const pc_u32x4 *>(ptr));
    };
    for (int k = 0; k < KT; ++k) {
        if (k + 1 < KT) dma(k + 1);
        const unsigned char *ab = a_lane + (k & 1) * CHUNK, *bb = b_lane + k * WS;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            pc_bf16x8 fa[3], fb[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                fa[sp] = rd(ab + (cc * 6 + sp) * 1024);
                fb[sp] = rd(bb + sp * plane + cc * 32);
            }
            constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int i = 0; i < 6; ++i) acc2[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ia[i]], fb[ib[i]], acc2[i & 1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tap k + 1 has landed
        lds_barrier();                                            // ... for every wave; tap k's buffer is free
    }

    // ---- epilogue: lane (frame t, channels 4 half + 8 rg + 0..3 of the row tile) -------------------------------------------------
    const f32x16 acc = acc2[0] + acc2[1];
    const int64_t t = t0 + ct * 32 + l31;
    if (t >= p.T) return;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int o = rt * 32 + 4 * half + 8 * rg;
        if (o >= CG) continue;
        const int m = g * CG + o;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[4 * rg + r];
        if (p.bias) v += *reinterpret_cast<const f32x4 *>(p.bias + m);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));
        *reinterpret_cast<f32x4 *>(p.y + t * p.D + m) = v;
    }
}

// conv.weight [D][CG][taps] (weight-norm folded) -> [group][tap][chunk][row tile 2][split 3][lane 64][8] bf16; rows >= CG of the second tile = 0
static void posconv_pack_host(const float *w, int D, int groups, int taps, std::vector<uint16_t> *out) {
    const int CG = D / groups, NCC = CG / 16;
    out->assign((size_t)groups * taps * NCC * 6 * 512, 0);
    for (int g = 0; g < groups; ++g)
        for (int k = 0; k < taps; ++k)
            for (int cc = 0; cc < NCC; ++cc)
                for (int rt = 0; rt < 2; ++rt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int o = rt * 32 + (lane & 31);
                        if (o >= CG) continue;
                        for (int e = 0; e < 8; ++e) {
                            const int c = cc * 16 + 8 * (lane >> 5) + e;
                            float r = w[((size_t)(g * CG + o) * CG + c) * taps + k];
                            for (int sp = 0; sp < 3; ++sp) {
                                const uint16_t h = bf16_rne(r);
                                uint32_t bits = (uint32_t)h << 16;
                                float f;
                                memcpy(&f, &bits, 4);
                                r -= f;   // exact in fp32
                                const size_t piece = ((((size_t)g * taps + k) * NCC + cc) * 2 + rt) * 3 + sp;
                                (*out)[piece * 512 + lane * 8 + e] = h;
                            }
                        }
                    }
}

static bool posconv_shape_ok(int d, int groups, int taps) {
    if (d <= 0 || groups <= 0 || d % groups) return false;
    const int cg = d / groups;
    return (cg == 48 || cg == 64) && taps >= 1 && taps <= 128;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_posconv_bf16x3_weight_bytes(int d, int groups, int taps, size_t *bytes) {
    if (!bytes || !posconv_shape_ok(d, groups, taps))
        return fail("rvc_posconv_bf16x3_weight_bytes: built for 48 or 64 channels per group and <= 128 taps (got d %d, groups %d, taps %d)", d, groups, taps);
    *bytes = (size_t)groups * taps * (d / groups / 16) * 6 * 1024;
    return 0;
}

extern "C" int rvc_posconv_bf16x3_pack_weight(const float *w_host, int d, int groups, int taps, void *a_dev, void *stream) {
    if (!w_host || !a_dev) return fail("rvc_posconv_bf16x3_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_posconv_bf16x3_weight_bytes(d, groups, taps, &bytes)) return 1;
    std::vector<uint16_t> packed;
    posconv_pack_host(w_host, d, groups, taps, &packed);
    if (packed.size() * sizeof(uint16_t) != bytes) return fail("rvc_posconv_bf16x3_pack_weight: internal size mismatch");
    hipError_t e = hipMemcpyAsync(a_dev, packed.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);   // the staging vector dies with this call
    if (e != hipSuccess) return fail("rvc_posconv_bf16x3_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

template <int CG>
static int posconv_launch(const PosConvParams &p, hipStream_t stream) {
    if (reserve_whole_cu((const void *)posconv_kernel<CG>, "posconv")) return 1;
    dim3 grid((unsigned)ceil_div(p.T, PC_BN), (unsigned)p.groups, 1);
    hipLaunchKernelGGL(posconv_kernel<CG>, grid, dim3(PC_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_posconv_gelu_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, float *y_dev, int64_t n_frames, int d,
                                       int groups, int taps, int padding, void *stream) {
    if (!x_dev || !a_dev || !y_dev) return fail("rvc_posconv_gelu_bf16x3: null pointer");
    if (!posconv_shape_ok(d, groups, taps)) return fail("rvc_posconv_gelu_bf16x3: built for 48 or 64 channels per group and <= 128 taps (got d %d, groups %d, taps %d)", d, groups, taps);
    if (padding < 0 || padding >= taps) return fail("rvc_posconv_gelu_bf16x3: padding must lie in [0, taps)");
    if (n_frames <= 0) return 0;
    if (n_frames * d * 4 >= ((int64_t)1 << 40)) return fail("rvc_posconv_gelu_bf16x3: tensor too large");
    PosConvParams p;
    p.x = x_dev; p.a = a_dev; p.bias = bias_dev; p.y = y_dev; p.T = n_frames; p.D = d; p.groups = groups; p.taps = taps; p.pad = padding;
    const int cg = d / groups;
    const size_t lds = 2 * (size_t)(cg / 16) * 6 * 1024 + 3 * (size_t)(PC_BN + taps - 1) * (cg * 2 + 16);
    if (lds > (size_t)LDS_WHOLE_CU) return fail("rvc_posconv_gelu_bf16x3: %zu bytes of LDS needed", lds);
    return cg == 48 ? posconv_launch<48>(p, (hipStream_t)stream) : posconv_launch<64>(p, (hipStream_t)stream);
}
