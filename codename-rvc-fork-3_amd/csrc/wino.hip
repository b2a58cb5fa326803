// K3w -- the ResBlock convolutions as a fast (Winograd / Toom-Cook) convolution on the fp32 matrix cores.
// (The text below derives the F(4,3) form, used for 3 taps; 7 and 11 taps run the same scheme with F(4,4) groups of four
// taps on seven points -- template parameter R of the kernel.)
//
// 98 % of the vocoder's 3.5 TFLOP per utterance are 3-, 7- and 11-tap dilated conv1d layers (residuals.py:75-86).  The
// direct implicit GEMM (conv.hip) is within ~20 % of the fp32 MFMA peak on them, so the remaining lever is arithmetic:
// F(4,3) computes 4 outputs of a 3-tap correlation with 6 multiplications instead of 12, and a K-tap kernel is the sum of
// G = ceil(K / 3) three-tap kernels applied to shifted inputs:
//
//   y[t0 + i d] = sum_g sum_kk w[3g + kk] x[t0 + (i + 3g + kk - c) d]            c = (K - 1) / 2, i = 0..3, dilation d
//               = sum_p AT[i][p] * sum_g sum_ci U[p][g][ci][co] * X[p][g][ci][tile]
//   U[p][g]  = G3 w[3g .. 3g + 2]                (6 x 3 transform of the taps, per lane at fragment-load time)
//   X[p][g]  = BT (x[t0 + (n + 3g - c) d])_{n = 0..5}   (6 x 6 transform of the input window, at fragment-load time)
//
// The sum over groups commutes with the output transform, so each of the 6 transform points is ONE GEMM with
// K = G * C_in: 6 G multiply-adds per 4 outputs per (c_in, c_out) instead of 4 K -- 2.0x / 1.56x / 1.83x fewer for
// K = 3 / 7 / 11, 1.75x over a ResBlock.  Everything stays fp32; the transforms have small integer / dyadic entries
// (|BT| <= 5, |AT| <= 8) and the measured error of one layer is 4-7e-7 relative RMS against float64 (the direct fp32 form:
// 1.4-2.4e-7), three orders below the 1e-3 waveform gate.
//
// A dilated conv touches every d-th sample, so a tile is 4 outputs d apart: time t = sb * 4d + i * d + phi (super-block sb,
// i = 0..3, phase phi < d) belongs to tile tau = sb * d + phi, output i.  In LDS the input chunk is stored de-interleaved,
// X[ci / 2][i][tau][ci % 2], so that the 32 lanes of an MFMA "B" operand (consecutive tiles) read consecutive addresses
// whatever d is.
//
// Block = 4 waves (2 x 2: 64 channels x 64 tiles), two blocks per CU; wave (wm, wn) owns 32 output channels x 32 tiles
// (128 outputs) and keeps one accumulator per transform point (6 x 16 registers).  K is walked in chunks of CIC input
// channels, double-buffered in LDS, one barrier per chunk.  The epilogue applies AT, bias, residual, running sum and scale
// like conv.hip's.  What shaped the loop is written at the kernel.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

struct WinoParams {
    const float *x = nullptr;        // [batch][c_in][L]
    const void *u = nullptr;         // [3 G][c_in / 2][c_out][2]: the taps (zero-padded to a multiple of three), channel pairs interleaved;
                                     // fp32, or (u_bf16) one 32-bit word per pair: even channel in the low half
    bool u_bf16 = false;
    const float *bias = nullptr;     // [c_out]
    const float *res = nullptr;      // [batch][c_out][L] or null
    const float *accin = nullptr;    // [batch][c_out][L] or null
    float *y = nullptr;              // [batch][c_out][L]
    int c_in = 0, c_out = 0;
    int64_t L = 0;
    int dil = 1;
    int sb_per_block = 0;            // super-blocks (of d tiles) per block
    int64_t n_sb = 0;                // super-blocks in the sequence: ceil(L / 4d)
    int n_tile_blocks = 0;           // blocks along time: ceil(n_sb / sb_per_block)
    float slope = 1.f, out_scale = 1.f;
    int batch = 1;
};

constexpr int WINO_MAX_DIL = 5;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1))) *wino_gptr_t;
typedef void __attribute__((address_space(3))) *wino_lptr_t;

__device__ __forceinline__ f32x2 fma2(float a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(f32x2{a, a}, b, c); }
// leaky ReLU for 0 <= slope <= 1 as max(v, slope * v): one packed multiply + two v_max (fmaxf would add a canonicalising
// v_max per operand; NaN in gives NaN out either way)
__device__ __forceinline__ f32x2 lrelu2(f32x2 v, float slope) {
    const f32x2 sv = v * slope;
    float a, b;
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(v.y), "v"(sv.y));
    return f32x2{a, b};
}
// the same without inline assembly (ablation build: bisection of the cohabitation finding)
__device__ __forceinline__ f32x2 lrelu2_plain(f32x2 v, float slope) {
    const f32x2 sv = v * slope;
    return f32x2{__builtin_fmaxf(v.x, sv.x), __builtin_fmaxf(v.y, sv.y)};
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
constexpr int WINO_RSRC_FLAGS = 0x00020000;   // raw buffer, 32-bit data format

// KW taps; WM x WN waves (each 32 channels x 32 tiles); CIC input channels per chunk.
//
// The fp32 matrix instruction runs on the SIMD's own fp32 lanes: measured, matrix-busy + VALU-busy add up to ~1 on this
// kernel and on conv.hip's, and every ablation of non-matrix work (transforms, staging, index arithmetic) came off the
// run time one for one (tools/ablate_wino.sh).  So the loop is written for the FEWEST vector instructions, not for overlap:
//   * two k-steps are transformed at once in packed fp32 (v_pk_fma / v_pk_add): LDS holds channel PAIRS interleaved, one
//     ds_read_b64 brings the operands of k-steps 2P and 2P + 1 and the 18 packed operations of the two transforms serve 12
//     matrix instructions (the scalar version: 48);
//   * the transforms carry no scale factors -- the rows of G are scaled by (1/4, -1/6, -1/6, 1/24, 1/24, 1) in the
//     epilogue instead, once per accumulator;
//   * the weight taps go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane): no registers, no vector instructions;
//   * the input rows are staged with two fixed per-thread offsets (one division set per block, not per element) and only
//     the first / last block of a row masks for the conv's zero padding.
// WB16: the taps are stored as bf16 (BASELINE cfg 4), one 32-bit word per input-channel pair; they stay bf16 in LDS and are
// widened (two shifts / masks per tap pair) when the fragment is read -- half the tap stream and LDS footprint.
// R: taps per group.  3: F(4,3), 6 transform points (above).  4: F(4,4) on the points {0, 1, -1, 1/2, -1/2, 2, inf}: 7 products
// per 4 outputs per FOUR taps -- 14 instead of 18 for the 7-tap layers (2 groups instead of 3), 21 instead of 24 for 11 taps
// -- for larger transforms (43 packed operations per two k-steps instead of 18) and a seventh accumulator.
template <int KW, int WM, int WN, int CIC, int DBG = 0, bool WB16 = false, int R = 3>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
wino_conv_kernel(const WinoParams p) {
    constexpr int NP = R + 3;                            // transform points = window samples
    constexpr int G = (KW + R - 1) / R;
    constexpr int G3 = (KW + 2) / 3;                     // the tap slab in HBM always holds 3 ceil(K / 3) taps (>= R G)
    constexpr int C0 = (KW - 1) / 2;
    constexpr int BM = 32 * WM;
    constexpr int BNT = 32 * WN;                         // tile columns per block
    constexpr int NW = WM * WN;
    constexpr int NTH = 64 * NW;
    constexpr int CP = CIC / 2;                          // channel pairs per chunk
    static_assert(CIC % 4 == 0, "k-steps are processed in pairs of channel pairs");
    // window offsets (n + 3g - c) / 4 range over [MLO, MHI] super-block steps
    constexpr int SMIN = -C0, SMAX = NP - 1 + R * (G - 1) - C0;
    constexpr int MLO = SMIN >= 0 ? SMIN / 4 : -((-SMIN + 3) / 4);
    constexpr int MHI = SMAX / 4;
    constexpr int XT = BNT + (MHI - MLO) * WINO_MAX_DIL;  // staged tiles per row (enough for d <= 5)
    constexpr int XTS = ((XT + 31) / 32) * 32 + 8;        // row stride in float2: == 8 mod 32, the 4 de-interleaved rows of a ds_write_b64 land in 4 bank groups
    constexpr int XTOT = CP * 4 * XTS;                    // float2 per input buffer
    constexpr int UROWS = R * G * CP;                     // weight rows of BM channel pairs
    constexpr int UPAIR = WB16 ? 4 : 8;                   // bytes per stored pair
    constexpr int UINSTR = (UROWS * BM * UPAIR + 1023) / 1024;   // LDS-DMA wave-instructions (1 KiB each) per chunk
    constexpr int UTOT = UINSTR * 1024 / 8;               // float2-sized slots reserved per weight buffer
    constexpr int UPW = (UINSTR + NW - 1) / NW;           // per wave
    constexpr int NJ = (4 * XT + NTH - 1) / NTH;          // staged samples per thread per channel

    extern __shared__ __attribute__((aligned(16))) float wino_smem[];
    if constexpr (DBG & 1024) __builtin_amdgcn_s_setprio(3);   // ablation: this kernel's waves win every issue arbitration
    f32x2 *us = reinterpret_cast<f32x2 *>(wino_smem);     // [2][UTOT]
    f32x2 *xs = us + 2 * UTOT;                            // [2][XTOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    // Blocks go round-robin over the 8 XCDs (each with its own L2).  The n_m channel blocks of one time tile read the same
    // input rows, so they are given ids 8 apart: same XCD, dispatched together, the rows come from HBM once (measured
    // before this mapping: 2.27x the input's bytes at the L2's memory side with two channel blocks, profiles/r02_pmc_wino.txt).
    const int n_m = p.c_out / BM;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile_blk = (seq / n_m) * 8 + xcd;
    if (tile_blk >= p.n_tile_blocks) return;
    const int m0 = (seq % n_m) * BM;
    const int d = p.dil;
    const int64_t sb0 = (int64_t)tile_blk * p.sb_per_block;       // first super-block of this block
    const int n_tiles_blk = p.sb_per_block * d;                     // valid tile columns (<= BNT)
    const int64_t L = p.L;
    const float *const px = p.x + (int64_t)b * p.c_in * L;
    const int c_in = p.c_in, c_out = p.c_out;
    const float slope = p.slope;
    const int n_chunks = c_in / CIC;
    // staged time range: super-blocks [sb0 + MLO, sb0 + sb_per_block + MHI), i.e. xt_used tiles
    const int xt_used = (p.sb_per_block + MHI - MLO) * d;
    const int64_t t_start = (sb0 + MLO) * 4 * d;
    const int span = 4 * xt_used;                                   // samples per input row in the staged range
    const bool edge = t_start < 0 || t_start + span > L;            // block-uniform: some staged samples are conv padding

    // ---- staging plan, once per block ----------------------------------------------------------------------------
    // Both streams use buffer instructions: a scalar row / chunk offset plus a per-lane 32-bit offset fixed for the whole
    // block, so staging costs no address arithmetic on the vector pipe.
    __builtin_assume(wave >= 0 && wave < NW);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)px, 0, (int)((int64_t)c_in * L * 4), WINO_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t urs =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)((int64_t)3 * G3 * (c_in / 2) * c_out * UPAIR), WINO_RSRC_FLAGS);
    const int L4 = (int)(L * 4);
    unsigned goff[NJ];    // byte offset of the sample inside a channel row (clamped into [0, L))
    int loff[NJ];         // float2 offset inside a channel pair's 4 rows: ii * XTS + tile (lanes with nothing to stage: a pad column)
    unsigned inb = 0;     // bit j: the sample lies inside [0, L)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int tl = tid + j * NTH;
        const bool have = tl < span;
        const int tlc = have ? tl : 0;
        const int64_t t = t_start + tlc;
        const int64_t tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
        goff[j] = (unsigned)tc * 4u;
        // t_start is a multiple of 4d: tl = sbl * 4d + ii * d + phi  ->  row ii, tile sbl * d + phi
        const int sbl = tlc / (4 * d);
        const int r = tlc - sbl * 4 * d;
        const int ii = r / d;
        const int phi = r - ii * d;
        loff[j] = have ? ii * XTS + sbl * d + phi : XT + (lane & 7);   // columns >= XT of a row are never read
        if (t >= 0 && t < L) inb |= 1u << j;
    }
    unsigned woff[UPW];   // byte offset of this lane's 16 bytes of DMA piece (wave + NW * i) from the chunk's weight base
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int o = (wave + NW * i) * 1024 + 16 * lane;
        int row = o / (BM * UPAIR);
        const int within = o - row * (BM * UPAIR);
        if (row >= UROWS) row = UROWS - 1;             // overhang of the last piece (bf16, odd row counts): a harmless repeat
        const int tap = row / CP, cp = row - tap * CP;
        woff[i] = (unsigned)(((tap * (c_in / 2) + cp) * c_out) * UPAIR + within);
    }

    f32x2 xr[CP * NJ];
    auto load_x = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int s0 = (c * CIC + 2 * cp) * L4;       // wave-uniform byte offset of the pair's first row
#pragma unroll
            for (int j = 0; j < NJ; ++j) xr[cp * NJ + j] = f32x2{buf_load(xrs, goff[j], s0), buf_load(xrs, goff[j], s0 + L4)};
        }
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
        f32x2 *dst = xs + buf * XTOT;
        if (edge) {   // first / last blocks of a row: staged samples outside [0, L) are the conv's zero padding
#pragma unroll
            for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x2 v = (DBG & 2048) ? lrelu2_plain(xr[cp * NJ + j], slope) : lrelu2(xr[cp * NJ + j], slope);
                    dst[cp * 4 * XTS + loff[j]] = ((inb >> j) & 1) ? v : f32x2{0.f, 0.f};
                }
        } else {
#pragma unroll
            for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                for (int j = 0; j < NJ; ++j) dst[cp * 4 * XTS + loff[j]] = (DBG & 2048) ? lrelu2_plain(xr[cp * NJ + j], slope) : lrelu2(xr[cp * NJ + j], slope);
        }
    };
    // DBG & 64 (ablation build; correct results): the taps by plain loads into registers and ds_write instead of LDS-DMA
    f32x4 ureg[(DBG & 64) ? UPW : 1];
    auto dma_u = [&](int buf, int c) __attribute__((always_inline)) {
        const int s0 = (c * CP * c_out + m0) * UPAIR;     // byte offset of the chunk's first weight row, this block's channels
        char *dst = reinterpret_cast<char *>(us + buf * UTOT);
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int n = wave + NW * i;
            if constexpr (DBG & 64) {
                if (n < UINSTR) ureg[i] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.u) + s0 + woff[i]);
            } else {
                if (n < UINSTR) __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (wino_lptr_t)(dst + n * 1024), 16, (int)woff[i], s0, 0, 0);
            }
        }
    };
    auto store_u = [&](int buf) __attribute__((always_inline)) {
        if constexpr (DBG & 64) {
            char *dst = reinterpret_cast<char *>(us + buf * UTOT);
#pragma unroll
            for (int i = 0; i < UPW; ++i) {
                const int n = wave + NW * i;
                if (n < UINSTR) *reinterpret_cast<f32x4 *>(dst + n * 1024 + 16 * lane) = ureg[i];
            }
        }
    };
    auto block_barrier = [&]() __attribute__((always_inline)) {
        if constexpr (DBG & 128) __syncthreads(); else lds_barrier();
    };

    f32x16 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    load_x(0);
    dma_u(0, 0);
    store_x(0);
    store_u(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (n_chunks > 1) load_x(1);
    block_barrier();
    const int col = wn * 32 + l31;                       // this lane's tile column inside the block
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        // buffer buf ^ 1 was last read in iteration c - 1 and every wave has passed that iteration's barrier
        if (!(DBG & 4) && c + 1 < n_chunks) dma_u(buf ^ 1, c + 1);
        if constexpr (DBG & 4096) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ablation: no memory return lands while the matrix instructions run
        const f32x2 *ua = us + buf * UTOT + wm * 32 + l31;
        const unsigned *ua16 = reinterpret_cast<const unsigned *>(us + buf * UTOT) + wm * 32 + l31;
        const f32x2 *xb = xs + buf * XTOT + col - MLO * d;
        // software pipeline over the S = G * CIC / 4 double k-steps: the 6 input pairs and 3 tap pairs of step s + 1 are
        // read from LDS before the transforms + 12 matrix instructions of step s are issued
        constexpr int S = G * (CIC / 4);
        f32x2 dv[2][NP], wv[2][R];
        auto fetch = [&](int st, f32x2 (&dd)[NP], f32x2 (&ww)[R]) __attribute__((always_inline)) {
            const int g = st / (CIC / 4), P = st - g * (CIC / 4);
            const int cpi = 2 * P + half;               // this lane's channel pair: k-step 2P takes .x, k-step 2P + 1 takes .y
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int sh = n + R * g - C0;           // window offset in units of d
                dd[n] = xb[(cpi * 4 + (sh & 3)) * XTS + (sh >> 2) * d];
            }
#pragma unroll
            for (int kt = 0; kt < R; ++kt) {
                if constexpr (WB16) {
                    const unsigned bits = ua16[((R * g + kt) * CP + cpi) * BM];
                    ww[kt] = f32x2{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)};
                } else {
                    ww[kt] = ua[((R * g + kt) * CP + cpi) * BM];
                }
            }
        };
        fetch(0, dv[0], wv[0]);
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (st + 1 < S) fetch(st + 1, dv[(st + 1) & 1], wv[(st + 1) & 1]);   // reads only: nothing here depends on them yet
            __builtin_amdgcn_sched_barrier(0);
            const f32x2(&dq)[NP] = dv[st & 1];
            f32x2 aq[NP], xq[NP];
            if constexpr (R == 3) {
                const f32x2 w0 = wv[st & 1][0], w1 = wv[st & 1][1], w2 = wv[st & 1][2];
                if (DBG & 1) {
                    aq[0] = w0; aq[1] = w1; aq[2] = w2; aq[3] = w0; aq[4] = w1; aq[5] = w2;
                } else {   // rows of G3 without their scale factors (applied in the epilogue)
                    const f32x2 ts = w0 + w2, tv = fma2(4.f, w2, w0);
                    aq[0] = w0;
                    aq[1] = ts + w1;
                    aq[2] = ts - w1;
                    aq[3] = fma2(2.f, w1, tv);
                    aq[4] = fma2(-2.f, w1, tv);
                    aq[5] = w2;
                }
                if (DBG & 2) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) xq[q] = dq[q];
                } else {   // BT d
                    xq[0] = fma2(4.f, dq[0], fma2(-5.f, dq[2], dq[4]));
                    const f32x2 t1 = fma2(-4.f, dq[2], dq[4]);
                    const f32x2 t2 = fma2(-4.f, dq[1], dq[3]);
                    xq[1] = t1 + t2;
                    xq[2] = t1 - t2;
                    const f32x2 t3 = dq[4] - dq[2];
                    const f32x2 t4 = dq[3] - dq[1];
                    xq[3] = fma2(2.f, t4, t3);
                    xq[4] = fma2(-2.f, t4, t3);
                    xq[5] = fma2(4.f, dq[1], fma2(-5.f, dq[3], dq[5]));
                }
            } else {
                // F(4,4), points 0, 1, -1, 1/2, -1/2, 2, inf.  Taps: Vandermonde rows without their 1 / N_j (epilogue).
                const f32x2 w0 = wv[st & 1][0], w1 = wv[st & 1][1], w2 = wv[st & 1][2], w3 = wv[st & 1][3];
                const f32x2 e1 = w0 + w2, o1 = w1 + w3;
                const f32x2 e2 = fma2(0.25f, w2, w0), o2 = fma2(0.125f, w3, w1 * 0.5f);
                aq[0] = w0;
                aq[1] = e1 + o1;
                aq[2] = e1 - o1;
                aq[3] = e2 + o2;
                aq[4] = e2 - o2;
                aq[5] = fma2(8.f, w3, fma2(4.f, w2, fma2(2.f, w1, w0)));
                aq[6] = w3;
                // input: row j holds the coefficients of prod_{l != j} (x - a_l); the +- pairs share their even / odd halves
                xq[0] = fma2(-0.5f, dq[0], fma2(0.25f, dq[1], fma2(2.5f, dq[2], fma2(-1.25f, dq[3], fma2(-2.f, dq[4], dq[5])))));
                const f32x2 A = fma2(0.5f, dq[2], fma2(-0.25f, dq[3], fma2(-2.f, dq[4], dq[5])));
                const f32x2 B = fma2(0.5f, dq[1], fma2(-0.25f, dq[2], fma2(-2.f, dq[3], dq[4])));
                xq[1] = A + B;
                xq[2] = A - B;
                const f32x2 A2 = fma2(2.f, dq[2], fma2(-2.f, dq[4], dq[5] - dq[3]));
                const f32x2 B2 = fma2(-0.5f, dq[2], fma2(0.5f, dq[4], dq[1] - dq[3]));
                xq[3] = A2 + B2;
                xq[4] = A2 - B2;
                xq[5] = fma2(0.25f, dq[1], fma2(-1.25f, dq[3], dq[5]));
                xq[6] = fma2(-0.5f, dq[1], fma2(0.25f, dq[2], fma2(2.5f, dq[3], fma2(-1.25f, dq[4], fma2(-2.f, dq[5], dq[6])))));
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[q] = mfma32(aq[q].x, xq[q].x, acc[q]);
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[q] = mfma32(aq[q].y, xq[q].y, acc[q]);
            if constexpr (DBG & 256) {   // ablation: nothing may overwrite a matrix instruction's source registers for 80 cycles
#pragma unroll
                for (int z = 0; z < 5; ++z) asm volatile("s_nop 15");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < n_chunks) {
            if (!(DBG & 4)) {
                store_x(buf ^ 1);                                   // chunk c + 1's rows, loaded a whole chunk ago
                store_u(buf ^ 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA pieces of chunk c + 1 have landed
                if (c + 2 < n_chunks) load_x(c + 2);
            }
            if (!(DBG & 8)) block_barrier();   // chunk c + 2's rows stay in flight across it
        }
    }

    if constexpr (DBG & 512) __builtin_amdgcn_s_sleep(20);   // ablation: ~1300 cycles between the last matrix instruction and the first read of an accumulator
    // ---- epilogue: y_i = AT diag(s) D, + bias, + residual, + running sum, * scale ----------------------------------
    const float *bias = p.bias;
    const float *res = p.res ? p.res + (int64_t)b * c_out * L : nullptr;
    const float *accin = p.accin ? p.accin + (int64_t)b * c_out * L : nullptr;
    float *y = p.y + (int64_t)b * c_out * L;
    const float out_scale = p.out_scale;
    const int row_l = wm * 32 + 4 * half;                // + (r & 3) + 8 (r >> 2): this lane's rows inside the block
    f32x4 o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = bias ? bias[m0 + row_l + (r & 3) + 8 * (r >> 2)] : 0.f;
        if constexpr (R == 3) {
            const float d0 = acc[0][r], d1 = acc[1][r], d2 = acc[2][r], d3 = acc[3][r], d4 = acc[4][r], d5 = acc[5][r];
            const float s12 = (d1 + d2) * (-1.f / 6.f), m12 = (d1 - d2) * (-1.f / 6.f), s34 = (d3 + d4) * (1.f / 24.f), m34 = (d3 - d4) * (1.f / 24.f);
            o[r].x = fmaf(0.25f, d0, s12 + s34) + bv;
            o[r].y = fmaf(2.f, m34, m12) + bv;
            o[r].z = fmaf(4.f, s34, s12) + bv;
            o[r].w = fmaf(8.f, m34, m12) + d5 + bv;
        } else {   // AT diag(1 / N_j): N = (-1/2, -3/2, -9/2, 9/16, 15/16, 45/2, 1)
            const float t0 = acc[0][r] * -2.f, t1 = acc[1][r] * (-2.f / 3.f), t2 = acc[2][r] * (-2.f / 9.f), t3 = acc[3][r] * (16.f / 9.f),
                        t4 = acc[4][r] * (16.f / 15.f), t5 = acc[5][r] * (2.f / 45.f), t6 = acc[6][r];
            const float s12 = t1 + t2, m12 = t1 - t2, s34 = t3 + t4, m34 = t3 - t4;
            o[r].x = (t0 + s12) + (s34 + t5) + bv;
            o[r].y = fmaf(0.5f, m34, m12) + fmaf(2.f, t5, bv);
            o[r].z = fmaf(0.25f, s34, s12) + fmaf(4.f, t5, bv);
            o[r].w = fmaf(0.125f, m34, m12) + fmaf(8.f, t5, t6) + bv;
        }
    }
    if (DBG & 32) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += o[r].x + o[r].y + o[r].z + o[r].w;
        if (sum != 12345.678f) return;
    }
    const bool l4 = (L & 3) == 0;
    if (d == 1 && l4) {
        // a lane's 4 outputs are 16 contiguous bytes: straight from the registers.  All residual / running-sum loads are
        // issued before the first is used (one memory latency per block, not one per row).
        const int64_t t0 = (sb0 + col) * 4;
        if (col < n_tiles_blk && t0 < L) {
            const int64_t base = (int64_t)(m0 + row_l) * L + t0;
            if (res) {
                f32x4 rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = *reinterpret_cast<const f32x4 *>(res + base + (int64_t)((r & 3) + 8 * (r >> 2)) * L);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] += rv[r];
            }
            if (accin) {
                f32x4 av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = *reinterpret_cast<const f32x4 *>(accin + base + (int64_t)((r & 3) + 8 * (r >> 2)) * L);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] += av[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<f32x4 *>(y + base + (int64_t)((r & 3) + 8 * (r >> 2)) * L) = o[r] * out_scale;
        }
        return;
    }
    // Dilated (a lane's outputs are d apart) or ragged rows: through LDS, so that HBM sees whole contiguous rows.
    // yt[row][t - t_blk0], row stride YS floats; the chunk buffers are dead once every wave has left the main loop.
    constexpr int YS = 4 * BNT + 4;
    // (the launch reserves max(chunk buffers, BM * YS floats): wino_lds_bytes)
    float *yt = wino_smem;
    lds_barrier();
    {
        const int sbl = col / d;
        const int tl0 = sbl * 4 * d + (col - sbl * d);
        if (col < n_tiles_blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *dst = yt + (row_l + (r & 3) + 8 * (r >> 2)) * YS + tl0;
                dst[0] = o[r].x; dst[d] = o[r].y; dst[2 * d] = o[r].z; dst[3 * d] = o[r].w;
            }
        }
    }
    lds_barrier();
    const int64_t t_blk0 = sb0 * 4 * d;                             // a multiple of 4
    const int64_t left = L - t_blk0;
    const int n_t = (int)(left < 4 * n_tiles_blk ? left : 4 * n_tiles_blk);   // valid outputs per row in this block
    if (l4) {   // 16-byte pieces: BNT threads per row, NTH / BNT rows per pass
        constexpr int RPP = NTH / BNT, PASSES = BM / RPP;
        const int tq = (tid % BNT) * 4, rq = tid / BNT;
        if (tq < n_t) {
            const int64_t base = (int64_t)(m0 + rq) * L + t_blk0 + tq;
            f32x4 v[PASSES];
#pragma unroll
            for (int k = 0; k < PASSES; ++k) v[k] = *reinterpret_cast<const f32x4 *>(yt + (rq + k * RPP) * YS + tq);
            if (res) {
                f32x4 rv[PASSES];
#pragma unroll
                for (int k = 0; k < PASSES; ++k) rv[k] = *reinterpret_cast<const f32x4 *>(res + base + (int64_t)k * RPP * L);
#pragma unroll
                for (int k = 0; k < PASSES; ++k) v[k] += rv[k];
            }
            if (accin) {
                f32x4 av[PASSES];
#pragma unroll
                for (int k = 0; k < PASSES; ++k) av[k] = *reinterpret_cast<const f32x4 *>(accin + base + (int64_t)k * RPP * L);
#pragma unroll
                for (int k = 0; k < PASSES; ++k) v[k] += av[k];
            }
#pragma unroll
            for (int k = 0; k < PASSES; ++k) *reinterpret_cast<f32x4 *>(y + base + (int64_t)k * RPP * L) = v[k] * out_scale;
        }
    } else {    // row length not a multiple of 4: single samples
        for (int e = tid; e < BM * 4 * BNT; e += NTH) {
            const int rq = e / (4 * BNT), tq = e - rq * (4 * BNT);
            if (tq >= n_t) continue;
            const int64_t at = (int64_t)(m0 + rq) * L + t_blk0 + tq;
            float v = yt[rq * YS + tq];
            if (res) v += res[at];
            if (accin) v += accin[at];
            y[at] = v * out_scale;
        }
    }
}

template <int KW, int WM, int WN, int CIC, bool WB16 = false, int R = 3>
static size_t wino_lds_bytes() {
    constexpr int G = (KW + R - 1) / R, C0 = (KW - 1) / 2;
    constexpr int SMIN = -C0, SMAX = R + 2 + R * (G - 1) - C0;
    constexpr int MLO = SMIN >= 0 ? SMIN / 4 : -((-SMIN + 3) / 4), MHI = SMAX / 4;
    constexpr int XT = 32 * WN + (MHI - MLO) * WINO_MAX_DIL;
    constexpr int XTS = ((XT + 31) / 32) * 32 + 8;
    const size_t upieces = ((size_t)R * G * (CIC / 2) * 32 * WM * (WB16 ? 4 : 8) + 1023) / 1024;
    const size_t chunks = 2 * (upieces * 1024 + (size_t)(CIC / 2) * 4 * XTS * 2 * sizeof(float));
    const size_t out_tile = (size_t)32 * WM * (4 * 32 * WN + 4) * sizeof(float);   // the dilated epilogue's transposed tile
    return chunks > out_tile ? chunks : out_tile;
}

template <int KW, int WM, int WN, int CIC, int DBG = 0, bool WB16 = false, int R = 3>
static int wino_launch_cfg(WinoParams p, hipStream_t stream) {
    constexpr int BNT = 32 * WN;
    p.sb_per_block = BNT / p.dil;
    p.n_sb = ceil_div(p.L, (int64_t)4 * p.dil);
    const size_t lds = wino_lds_bytes<KW, WM, WN, CIC, WB16, R>();
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [lds] {
        err = hipFuncSetAttribute((const void *)wino_conv_kernel<KW, WM, WN, CIC, DBG, WB16, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (err != hipSuccess) return fail("wino conv: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(err));
    p.n_tile_blocks = (int)ceil_div(p.n_sb, p.sb_per_block);
    const int n_m = p.c_out / (32 * WM);
    dim3 grid((unsigned)(ceil_div(p.n_tile_blocks, 8) * 8 * n_m), 1, (unsigned)p.batch);
    hipLaunchKernelGGL((wino_conv_kernel<KW, WM, WN, CIC, DBG, WB16, R>), grid, dim3(64 * WM * WN), lds, stream, p);
    RVC_LAUNCH_CHECK();
    return 0;
}

template <int KW>
static int wino_launch_kw(const WinoParams &p, hipStream_t stream) {
    constexpr int CIC = 8;
    if (p.c_in % CIC) return fail("wino conv: c_in %d is not a multiple of %d", p.c_in, CIC);
#ifdef RVC_ABLATE
    if (KW == 11 && p.c_out % 64 == 0) {   // ablations (wrong results): where does the time go (tools/ablate_wino.sh)
        static const int dbg = knob("RVC_WINO_DBG", 0);
        switch (dbg) {
            case 1: return wino_launch_cfg<11, 2, 2, 8, 1>(p, stream);
            case 2: return wino_launch_cfg<11, 2, 2, 8, 2>(p, stream);
            case 3: return wino_launch_cfg<11, 2, 2, 8, 3>(p, stream);
            case 7: return wino_launch_cfg<11, 2, 2, 8, 7>(p, stream);
            case 15: return wino_launch_cfg<11, 2, 2, 8, 15>(p, stream);
            case 32: return wino_launch_cfg<11, 2, 2, 8, 32>(p, stream);
            case 47: return wino_launch_cfg<11, 2, 2, 8, 47>(p, stream);
            default: break;
        }
    }
#endif
    // taps per group: F(4,4) for the kernel sizes in RVC_WINO_R4's mask (1: 7 taps, 2: 11 taps; default both).  Measured
    // against F(4,3) (tools/bench_conv.py): 7 taps 0.86-0.93 of the time on every shape (530 -> 463 us at C = 128),
    // 11 taps 0.94-0.98 (678 -> 641 us).
    static const int r4_mask = knob("RVC_WINO_R4", 3);
    if constexpr (KW == 7 || KW == 11) {
        if (r4_mask & (KW == 7 ? 1 : 2)) {
            if (p.u_bf16) {
                if (p.c_out % 64 == 0) return wino_launch_cfg<KW, 2, 2, CIC, 0, true, 4>(p, stream);
                if (p.c_out % 32 == 0) return wino_launch_cfg<KW, 1, 4, CIC, 0, true, 4>(p, stream);
            }
            if (p.c_out % 64 == 0) return wino_launch_cfg<KW, 2, 2, CIC, 0, false, 4>(p, stream);
            if (p.c_out % 32 == 0) return wino_launch_cfg<KW, 1, 4, CIC, 0, false, 4>(p, stream);
        }
    }
#ifdef RVC_ABLATE
    if constexpr (KW == 3) {   // bisection of profiles/r03_mfma_cohabitation.txt (correct results): 64 no LDS-DMA, 128 __syncthreads, 256 s_nop behind the matrix instructions, 512 s_sleep before the epilogue, 1024 s_setprio 3
        static const int fix = knob("RVC_WINO_FIX", 0);
        if (p.c_out % 64 == 0 && !p.u_bf16) switch (fix) {
            case 64: return wino_launch_cfg<3, 2, 2, CIC, 64>(p, stream);
            case 128: return wino_launch_cfg<3, 2, 2, CIC, 128>(p, stream);
            case 192: return wino_launch_cfg<3, 2, 2, CIC, 192>(p, stream);
            case 256: return wino_launch_cfg<3, 2, 2, CIC, 256>(p, stream);
            case 512: return wino_launch_cfg<3, 2, 2, CIC, 512>(p, stream);
            case 1024: return wino_launch_cfg<3, 2, 2, CIC, 1024>(p, stream);
            case 768: return wino_launch_cfg<3, 2, 2, CIC, 768>(p, stream);
            case 2048: return wino_launch_cfg<3, 2, 2, CIC, 2048>(p, stream);
            case 4096: return wino_launch_cfg<3, 2, 2, CIC, 4096>(p, stream);
            case 4160: return wino_launch_cfg<3, 2, 2, CIC, 4160>(p, stream);
            default: break;
        }
    }
#else
    if constexpr (KW != 3) return fail("wino conv: c_out %d is not a multiple of 32", p.c_out);   // F(4,3) on 7 / 11 taps: ablation build only
    else
#endif
    {
        if (p.u_bf16) {
            if (p.c_out % 64 == 0) return wino_launch_cfg<KW, 2, 2, CIC, 0, true>(p, stream);
            if (p.c_out % 32 == 0) return wino_launch_cfg<KW, 1, 4, CIC, 0, true>(p, stream);
        }
        if (p.c_out % 64 == 0) return wino_launch_cfg<KW, 2, 2, CIC>(p, stream);
        if (p.c_out % 32 == 0) return wino_launch_cfg<KW, 1, 4, CIC>(p, stream);
        return fail("wino conv: c_out %d is not a multiple of 32", p.c_out);
    }
}

// the staging streams address one batch item's input and the weight slab with 32-bit byte offsets
bool wino_fits(int c_in, int c_out, int64_t L) { return (int64_t)c_in * L < ((int64_t)1 << 29) && (int64_t)12 * c_in * c_out < ((int64_t)1 << 29); }

bool wino_supported(int k, int dil) { return (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= WINO_MAX_DIL; }

int launch_wino_conv(const float *x, const void *u, bool u_bf16, const float *bias, const float *res, const float *accin, float *y,
                     int batch, int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream) {
    if (!wino_supported(k, dil)) return fail("wino conv: unsupported kernel size %d / dilation %d", k, dil);
    if (!(slope >= 0.f && slope <= 1.f)) return fail("wino conv: leaky slope %g outside [0, 1]", (double)slope);
    if (c_in % 8 || c_out % 32) return fail("wino conv: %d -> %d channels unsupported (multiples of 8 / 32)", c_in, c_out);
    if (!wino_fits(c_in, c_out, L)) return fail("wino conv: %d x %lld samples exceed the 2 GiB buffer addressing of the fast form", c_in, (long long)L);
    if (L <= 0 || batch <= 0) return 0;
    WinoParams p;
    p.x = x; p.u = u; p.u_bf16 = u_bf16; p.bias = bias; p.res = res; p.accin = accin; p.y = y;
    p.c_in = c_in; p.c_out = c_out; p.L = L; p.dil = dil; p.slope = slope; p.out_scale = out_scale; p.batch = batch;
    switch (k) {
        case 3: return wino_launch_kw<3>(p, stream);
        case 7: return wino_launch_kw<7>(p, stream);
        default: return wino_launch_kw<11>(p, stream);
    }
}

// w_host [c_out][c_in][k] -> [3 G][c_in / 2][c_out][2] (taps zero-padded to a multiple of three, channel pairs interleaved)
void wino_pack_host(const float *w_host, int c_out, int c_in, int k, std::vector<float> *out) {
    const int G = (k + 2) / 3;
    out->assign((size_t)3 * G * c_in * c_out, 0.f);
    for (int tap = 0; tap < k; ++tap)
        for (int ci = 0; ci < c_in; ++ci)
            for (int co = 0; co < c_out; ++co)
                (*out)[(((size_t)tap * (c_in / 2) + ci / 2) * c_out + co) * 2 + (ci & 1)] = w_host[((size_t)co * c_in + ci) * k + tap];
}

// the same slab with each channel pair as one word of two bf16 (round to nearest even; exact for bf16-valued weights)
void wino_pack_host_bf16(const float *w_host, int c_out, int c_in, int k, std::vector<uint32_t> *out) {
    std::vector<float> f;
    wino_pack_host(w_host, c_out, c_in, k, &f);
    out->resize(f.size() / 2);
    for (size_t i = 0; i < out->size(); ++i) (*out)[i] = (uint32_t)bf16_rne(f[2 * i]) | ((uint32_t)bf16_rne(f[2 * i + 1]) << 16);
}

int wino_pack_weight(const float *w_host, int c_out, int c_in, int k, float **out_dev) {
    if (c_in % 2) return fail("wino_pack_weight: odd c_in %d", c_in);
    std::vector<float> u;
    wino_pack_host(w_host, c_out, c_in, k, &u);
    hipError_t e = hipMalloc((void **)out_dev, u.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(*out_dev, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail("wino_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv1d_wino_pack_weight(const float *w_host, int c_out, int c_in, int k, float *u_dev, void *stream) {
    if (!w_host || !u_dev || c_out <= 0 || c_in <= 0 || !(k == 3 || k == 7 || k == 11)) return fail("rvc_conv1d_wino_pack_weight: bad argument");
    float *tmp = nullptr;
    if (wino_pack_weight(w_host, c_out, c_in, k, &tmp)) return 1;
    const size_t bytes = (size_t)3 * ((k + 2) / 3) * c_in * c_out * sizeof(float);
    hipError_t e = hipMemcpyAsync(u_dev, tmp, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail("rvc_conv1d_wino_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv1d_wino_forward(const float *x_dev, const float *u_dev, const float *bias_dev, const float *res_dev,
                                       const float *acc_dev, float *y_dev, int batch, int c_in, int c_out, int64_t length, int k,
                                       int dilation, float slope_in, float out_scale, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_conv1d_wino_forward: null pointer");
    return launch_wino_conv(x_dev, u_dev, false, bias_dev, res_dev, acc_dev, y_dev, batch, c_in, c_out, length, k, dilation, slope_in,
                            out_scale, (hipStream_t)stream);
}
