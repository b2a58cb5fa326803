// K3x -- the 7- and 11-tap ResBlock convolutions in Winograd F(4,4) form on the bf16 MATRIX cores, fp32-exact operands.
//
// wino.hip runs the same transform on v_mfma_f32_32x32x2_f32, which executes at the fp32 VECTOR rate (157 TF) and shares
// the SIMD's lanes with the transforms: it sits at ~0.6 of that pipe.  The bf16 matrix pipe is 16x faster and separate
// from the vector ALU.  This kernel feeds it fp32 values without giving up fp32 accuracy:
//
//   every fp32 number splits EXACTLY into three bf16 numbers   v = v0 + v1 + v2      (8 + 8 + 8 significand bits,
//   round-to-nearest at each level: |v1| <= 2^-8 |v|, |v2| <= 2^-16 |v|, nothing left over)
//   a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)  + [a1 b2 + a2 b1 + a2 b2   <= 2^-23 |a b|, dropped]
//
// i.e. SIX v_mfma_f32_32x32x16_bf16 per 16-deep k-step with fp32 accumulation: each product of two bf16 is exact in fp32,
// the dropped terms are below one fp32 ulp of the product, and the accumulator sees 6 roundings per 16 input channels where
// the fp32 instruction's fma chain sees 16.  Six of these cost 6 x 32 = 192 matrix cycles against 8 x 64 = 512 for the same
// multiply-adds on the fp32 instruction.
//
// Winograd domain (the derivation is in wino.hip; F(4,4) on the points 0, 1, -1, 1/2, -1/2, 2, inf; G = ceil(K / 4) tap
// groups):    Y_p[co][tile] = sum_g sum_ci U_p[g][co][ci] * X_p[ci][tile + g d]          p = 0..6
//   U_p: the tap transform, computed in float64 on the host ONCE at load, rounded to fp32, split into three bf16 and laid out
//        as ready-made matrix-instruction A fragments (1 KiB each: 32 channels x 16 input channels);
//   X_p: the input transform of the 7-sample window that starts at `tile`, in fp32 (packed v_pk_fma, the same expressions as
//        wino.hip), split into three bf16 with v_cvt_pk_bf16_f32 and written to LDS as B fragments.  Group g of tile t uses
//        the window of tile t + g d, so each window is transformed ONCE per block (wino.hip transforms it per group and per
//        channel-block wave: 6x) and read back 2 G times.
//
// Block = BM output channels x BNT window columns, 8 waves (BM / 32 x BNT / 32, each 32 channels x 32 tiles x 7 points = 112
// accumulator registers), one block per CU, in two shapes picked per layer by winobf_block_rows():
//   64 x 128   the transformed taps are 7 G / K x 1.5 = 2.9x the raw fp32 taps in bytes and a block can only amortise them
//              over the columns its accumulators cover: at this shape the tap stream is 16 B/clk/CU at full matrix rate, from
//              L2.  Of the 128 columns, 128 - (G - 1) d are valid tiles and the rest are the windows the later groups reach;
//              the 8 channel pairs x 128 windows of a chunk are two rounds of the 512 threads;
//   128 x 64   the same matrix work per step with HALF the window transforms, splits and raw-row staging (one round of the
//              512 threads) and twice the tap bytes (36 KiB per 11-tap step: LDS is full to 128 bytes); it gives up
//              (G - 1) d of 64 columns instead of 128.  The kernels run power-limited, so the vector work
//              removed is time removed: -12 .. -17 % on the 7-tap layers, -7 % on the 256-channel 11-tap ones; level on the
//              128-channel 11-tap ones, which keep 64 x 128.
//
// The K loop runs over STEPS = (chunk of 16 input channels) x (point p): a step needs 18 KiB of tap fragments (64 x 128, 11 taps;
// LDS-DMA, ring of three slots, issued two steps ahead from inside the step) and the 14 KiB X_p of the chunk (ring of two), does 6 G matrix
// instructions per wave into ONE accumulator, and ends in one barrier.  The raw input rows of the next chunk are fetched in
// two halves into the second of two raw buffers.
//
// What shaped the step body (each item measured on MI355X, tools/ablate_winobf.sh, tools/pmc_winobf.sh):
//   * a dependent v_mfma_f32_32x32x16_bf16 issues ~64 cycles after its predecessor (the pipe takes one every 32), so the 18
//     instructions of a step are a 1152-cycle chain per wave; the two waves of a SIMD fill each other's slots.  Everything
//     else a step does -- the input transform + split of the NEXT step (LDS reads, ~40 vector instructions, LDS writes), the
//     fragment reads of the next tap group, the three LDS-DMA issues, the raw-row stores -- is cut into ~15 stages that are
//     placed, one per gap, BETWEEN the matrix instructions of the wave's own chain (sched_barrier pins the order).  Left to
//     the compiler the matrix instructions are issued back to back and the transform after them, and the two add up: 563 us
//     instead of 458 at C = 128, K = 11 (first version: separate transform and multiply phases in opposite order on the two
//     waves of a SIMD -- no overlap at all, the per-wave serial path is what counts);
//   * the window fragments are single-buffered: the products of a group are ordered so that b2 and b1 die early and are
//     reloaded for the next group while the current one finishes;
//   * measured limits: matrix pipe busy 35 % (the chain latency + ~900 cycles per step of barrier, LDS latency after the
//     barrier and DMA issue), LDS 35 % busy, 13 % of its cycles bank conflicts (the 4-byte B-fragment writes).  A second
//     accumulator chain per step (products alternating between the point's accumulator and a step-local one) would let one
//     wave saturate the pipe, but 7 x 16 + 16 accumulator registers next to the operand tuples do not fit 256 registers
//     without spilling whole accumulators (tried with VGPR and with AGPR-pinned accumulators: 676 us); 16x16x32 instructions
//     (four independent 16x16 tiles per wave tile, two splits concatenated along K: (a0|a1).(b1|b0) = a0 b1 + a1 b0 ...) need
//     1.7x the LDS fragment reads and measured 524 us.  Four waves of 512 registers, each 32 channels x 64 tiles (two
//     accumulators per point = two chains per wave), measured 467 us: without a partner wave the ~550 cycles of barrier /
//     LDS latency / DMA issue per step are exposed.  What remains is the per-step synchronisation, not the chain.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "conv.h"

namespace rvc {

struct WinoBfParams {
    const float *x = nullptr;        // [batch][c_in][L]
    const void *u = nullptr;         // winobf_pack_host's slab
    const float *bias = nullptr;     // [c_out]
    const float *res = nullptr;      // [batch][c_out][L] or null
    const float *accin = nullptr;    // [batch][c_out][L] or null
    float *y = nullptr;              // [batch][c_out][L]
    int c_in = 0, c_out = 0;
    int64_t L = 0;
    int dil = 1;
    int sb_per_block = 0;            // super-blocks (of d tiles) per block
    int64_t n_sb = 0;                // super-blocks in the sequence: ceil(L / 4d)
    int n_tile_blocks = 0;           // blocks along time
    float slope = 1.f, out_scale = 1.f;
    int batch = 1;
};

constexpr int WBF_MAX_DIL = 5;
constexpr int WBF_NW = 8, WBF_NTH = 512, WBF_CIC = 16, WBF_CP = 8, WBF_NP = 7, WBF_R = 4;
// block shapes: 64 channels x 128 window columns (waves 2 x 4), or 128 channels x 64 columns (4 x 2) where c_out allows -- all of a
// 128-channel layer's outputs of a time tile in ONE block, so the raw rows are staged and transformed once instead of twice

typedef float wbf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wbf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wbf_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wbf_u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) *wbf_lptr_t;

__device__ __forceinline__ wbf_f32x2 wbf_fma2(float a, wbf_f32x2 b, wbf_f32x2 c) { return __builtin_elementwise_fma(wbf_f32x2{a, a}, b, c); }
__device__ __forceinline__ wbf_f32x2 wbf_lrelu2(wbf_f32x2 v, float slope) {
    const wbf_f32x2 sv = v * slope;
    float a, b;
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(v.x), "v"(sv.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(v.y), "v"(sv.y));
    return wbf_f32x2{a, b};
}
__device__ __forceinline__ float wbf_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
constexpr int WBF_RSRC_FLAGS = 0x00020000;

// geometry shared by the kernel and the launcher
template <int KW, int BM, int BNT>
struct WbfGeom {
    static constexpr int G = (KW + WBF_R - 1) / WBF_R;
    static constexpr int C0 = (KW - 1) / 2;
    static constexpr int SMIN = -C0, SMAX = WBF_NP - 1 + WBF_R * (G - 1) - C0;
    static constexpr int MLO = SMIN >= 0 ? SMIN / 4 : -((-SMIN + 3) / 4);
    static constexpr int MHI = SMAX / 4;
    // a block's valid tiles + the (G - 1) d windows behind them are 128 transformed windows, so the raw rows span at most
    // 128 + (MHI - MLO - G + 1) d tiles
    static constexpr int WM = BM / 32, WN = BNT / 32;
    static_assert(WM * WN == WBF_NW, "eight waves of 32 x 32");
    static constexpr int XT = BNT + (MHI - MLO - G + 1) * WBF_MAX_DIL;
    static constexpr int XTS = ((XT - 12 + 31) / 32) * 32 + 12;         // raw row stride in float2, == 12 mod 32: the 4 de-interleaved
                                                                        // rows of a ds_write_b64 land in 4 bank groups
    static_assert(XTS >= XT + 2, "");
    static constexpr int RAW_BYTES = WBF_CP * 4 * XTS * 8;              // one raw chunk; two buffers
    static constexpr int XB = BNT + (G - 1) * WBF_MAX_DIL;               // transformed windows per point (the matrix side reads up to here)
    static constexpr int XBP = BNT == 128 ? ((XB + 15) / 16) * 16 : XB;  // plane stride in tiles (the 128 x 64 shape has 128 bytes of LDS to spare)
    static constexpr int B_SLOT = 3 * 2 * XBP * 16;                     // [split][k half][tile][8 bf16]
    static constexpr int A_PIECES = G * WM * 3;                         // 1 KiB fragments per step: [g][32-channel block][split]
    static constexpr int A_SLOT = A_PIECES * 1024;
    static constexpr int UPW = (A_PIECES + WBF_NW - 1) / WBF_NW;        // DMA pieces per wave per step
    static_assert(UPW <= (G == 3 ? 5 : 3), "the step body places UPW pieces by hand");
    static constexpr int NJ = (4 * XT + WBF_NTH - 1) / WBF_NTH;         // staged samples per thread per channel row
    static constexpr int LOOP_BYTES = 2 * RAW_BYTES + 2 * B_SLOT + 3 * A_SLOT;
    static constexpr int YS = 4 * BNT + 4;
    static constexpr int OUT_BYTES = BM * YS * 4;
    static constexpr int TQ = BNT / 64;                                 // transform rounds per step: 8 channel pairs x BNT windows / 512 threads
    static constexpr int LDS_BYTES = LOOP_BYTES > OUT_BYTES ? LOOP_BYTES : OUT_BYTES;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

// DBG (ablations, wrong results; tools/ablate_winobf.sh): 1 no input transform, 2 no matrix instructions, 4 no tap DMA,
// 8 no per-step barrier, 16 no raw-row staging after the first chunk
template <int KW, int DBG = 0, int BM = 64, int BNT = 128>
__global__ void __launch_bounds__(WBF_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
winobf_conv_kernel(const WinoBfParams p) {
    using GM = WbfGeom<KW, BM, BNT>;
    constexpr int NP = WBF_NP, R = WBF_R, G = GM::G, C0 = GM::C0, MLO = GM::MLO, MHI = GM::MHI;
    constexpr int NW = WBF_NW, NTH = WBF_NTH, CIC = WBF_CIC, CP = WBF_CP, WM = GM::WM, WN = GM::WN, TQ = GM::TQ;
    constexpr int XT = GM::XT, XTS = GM::XTS, XBP = GM::XBP, NJ = GM::NJ, UPW = GM::UPW;

    extern __shared__ __attribute__((aligned(16))) float wbf_smem[];
    unsigned char *const smem = reinterpret_cast<unsigned char *>(wbf_smem);
    wbf_f32x2 *const xs = reinterpret_cast<wbf_f32x2 *>(smem);                       // raw chunks [2][CP][4][XTS]
    constexpr int XRAW = CP * 4 * XTS;                                                // float2 per raw buffer
    unsigned char *const bs = smem + 2 * GM::RAW_BYTES;                               // [2][B_SLOT]
    unsigned char *const as = bs + 2 * GM::B_SLOT;                                    // [3][A_SLOT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int n_m = p.c_out / BM;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile_blk = (seq / n_m) * 8 + xcd;               // the n_m channel blocks of one time tile: ids 8 apart, same XCD
    if (tile_blk >= p.n_tile_blocks) return;
    const int mblk = seq % n_m;
    const int m0 = mblk * BM;
    const int d = p.dil;
    const int64_t sb0 = (int64_t)tile_blk * p.sb_per_block;
    const int n_tiles_blk = p.sb_per_block * d;
    const int64_t L = p.L;
    const float *const px = p.x + (int64_t)b * p.c_in * L;
    const int c_in = p.c_in, c_out = p.c_out;
    const float slope = p.slope;
    const int n_chunks = c_in / CIC;
    const int n_steps = n_chunks * NP;
    const int xt_used = (p.sb_per_block + MHI - MLO) * d;
    const int64_t t_start = (sb0 + MLO) * 4 * d;
    const int span = 4 * xt_used;
    const bool edge = t_start < 0 || t_start + span > L;

    // ---- staging plan of the raw input rows (as wino.hip: one division set per block) -----------------------------
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)px, 0, (int)((int64_t)c_in * L * 4), WBF_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t urs =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)((int64_t)c_in * c_out * NP * G * 6), WBF_RSRC_FLAGS);
    const int L4 = (int)(L * 4);
    unsigned goff[NJ];
    int loff[NJ];
    unsigned inb = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int tl = tid + j * NTH;
        const bool have = tl < span;
        const int tlc = have ? tl : 0;
        const int64_t t = t_start + tlc;
        const int64_t tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
        goff[j] = (unsigned)tc * 4u;
        const int sbl = tlc / (4 * d);
        const int r = tlc - sbl * 4 * d;
        const int ii = r / d;
        const int phi = r - ii * d;
        loff[j] = have ? ii * XTS + sbl * d + phi : XT + (lane & 1);   // columns >= XT of a row are never read
        if (t >= 0 && t < L) inb |= 1u << j;
    }
    // a chunk's rows are fetched in two halves of four channel pairs (16 registers in flight instead of 32) and land in the
    // raw buffer the transforms are NOT reading (chunk c lives in buffer c & 1)
    constexpr int HP = CP / 2;
    wbf_f32x2 xr[HP * NJ];
    auto load_x = [&](int c, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int cp = 0; cp < HP; ++cp) {
            const int s0 = (c * CIC + 2 * (h * HP + cp)) * L4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) xr[cp * NJ + j] = wbf_f32x2{wbf_buf_load(xrs, goff[j], s0), wbf_buf_load(xrs, goff[j], s0 + L4)};
        }
    };
    static_assert(NJ <= 2, "the zero-padding mask below holds one keep word for j = 0 and one for j = 1");
    const unsigned keep0 = (!edge || (inb & 1)) ? 0xffffffffu : 0u, keep1 = (!edge || (inb & 2)) ? 0xffffffffu : 0u;   // conv zero padding
    auto store_x1 = [&](int c, int h, int cp) __attribute__((always_inline)) {   // one channel pair of the half
        wbf_f32x2 *const dst = xs + (c & 1) * XRAW + (h * HP + cp) * 4 * XTS;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const wbf_f32x2 v = wbf_lrelu2(xr[cp * NJ + j], slope);
            const unsigned k = j == 0 ? keep0 : keep1;
            dst[loff[j]] = wbf_f32x2{__uint_as_float(__float_as_uint(v.x) & k), __uint_as_float(__float_as_uint(v.y) & k)};
        }
    };
    auto store_x = [&](int c, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int cp = 0; cp < HP; ++cp) store_x1(c, h, cp);
    };
    // tap fragments of step s = c * 7 + pt: A_PIECES contiguous KiB in HBM/L2 -> A ring slot, UPW pieces per wave (the
    // overhang of the last round repeats a piece this wave has already issued: same bytes to the same place)
    const int step_bytes = GM::A_SLOT;
    const int blk_base = mblk * n_steps * step_bytes;
    auto dma_a1 = [&](int s, int i) __attribute__((always_inline)) {   // piece i of this wave
        unsigned char *dst = as + (s % 3) * GM::A_SLOT;
        const int s0 = blk_base + s * step_bytes;
        int n = wave + NW * i;
        if (n >= GM::A_PIECES) n -= NW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (wbf_lptr_t)(dst + n * 1024), 16, 16 * lane, s0 + n * 1024, 0, 0);
    };
    auto dma_a = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < UPW; ++i) dma_a1(s, i);
    };

    // ---- input transform of one point for every window of the chunk in LDS ------------------------------------------------
    // X_p for window tau' (first output of tile tau'): samples n = 0..6 at offset (n - C0) d, i.e. raw row (n - C0) & 3 of
    // raw tile tau' + ((n - C0) >> 2) d (raw tiles start MLO super-blocks before the block).  A block transforms 128 windows
    // per point (its valid tiles + the (G - 1) d that the later tap groups reach), so the 8 channel pairs x 128 windows are
    // exactly two rounds of the 512 threads: round q, wave w -> channel pair (w >> 1) + 4 q, window 64 (w & 1) + lane.
    // The work of a round is cut into stages (LDS reads / transform / two split levels / LDS writes) that the step body
    // places between the matrix instructions.
    static_assert(GM::XB <= GM::XBP, "");
    int t_src[TQ], t_dst[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        const int cp = wave / TQ + (NW / TQ) * q, tau = 64 * (wave % TQ) + lane;
        t_src[q] = cp * 4 * XTS - MLO * d + tau;
        t_dst[q] = (cp >> 2) * (XBP * 16) + tau * 16 + (cp & 3) * 4;
    }
    wbf_f32x2 tq[TQ][NP];          // the window samples a point needs (round q)
    wbf_f32x2 tv[TQ];              // transformed pair, then the split residual
    unsigned tw[TQ][3];            // the three bf16 pairs
    auto t_read = [&](auto PT, int q, const wbf_f32x2 *raw) __attribute__((always_inline)) {
        constexpr int pt = decltype(PT)::value;
        const wbf_f32x2 *const src = raw + t_src[q];
        constexpr int lo = pt == 0 ? 0 : 1, hi = pt == 0 ? 5 : (pt == 6 ? 6 : 5);
#pragma unroll
        for (int n = lo; n <= hi; ++n) {
            if (pt == 5 && (n == 2 || n == 4)) continue;
            const int sh = n - C0;
            tq[q][n] = src[(sh & 3) * XTS + (sh >> 2) * d];
        }
    };
    auto t_xform = [&](auto PT, int q) __attribute__((always_inline)) {
        constexpr int pt = decltype(PT)::value;
        const wbf_f32x2(&dq)[NP] = tq[q];
        if constexpr (pt == 0) {
            tv[q] = wbf_fma2(-0.5f, dq[0], wbf_fma2(0.25f, dq[1], wbf_fma2(2.5f, dq[2], wbf_fma2(-1.25f, dq[3], wbf_fma2(-2.f, dq[4], dq[5])))));
        } else if constexpr (pt == 1 || pt == 2) {
            const wbf_f32x2 A = wbf_fma2(0.5f, dq[2], wbf_fma2(-0.25f, dq[3], wbf_fma2(-2.f, dq[4], dq[5])));
            const wbf_f32x2 B = wbf_fma2(0.5f, dq[1], wbf_fma2(-0.25f, dq[2], wbf_fma2(-2.f, dq[3], dq[4])));
            tv[q] = pt == 1 ? A + B : A - B;
        } else if constexpr (pt == 3 || pt == 4) {
            const wbf_f32x2 A2 = wbf_fma2(2.f, dq[2], wbf_fma2(-2.f, dq[4], dq[5] - dq[3]));
            const wbf_f32x2 B2 = wbf_fma2(-0.5f, dq[2], wbf_fma2(0.5f, dq[4], dq[1] - dq[3]));
            tv[q] = pt == 3 ? A2 + B2 : A2 - B2;
        } else if constexpr (pt == 5) {
            tv[q] = wbf_fma2(0.25f, dq[1], wbf_fma2(-1.25f, dq[3], dq[5]));
        } else {
            tv[q] = wbf_fma2(-0.5f, dq[1], wbf_fma2(0.25f, dq[2], wbf_fma2(2.5f, dq[3], wbf_fma2(-1.25f, dq[4], wbf_fma2(-2.f, dq[5], dq[6])))));
        }
    };
    // exact three-way split: v = v0 + v1 + v2, each a pair of bf16 (even channel in the low half)
    auto t_split = [&](int q, int level) __attribute__((always_inline)) {
        const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(tv[q], wbf_bf16x2));
        tw[q][level] = w;
        if (level < 2) tv[q] = tv[q] - wbf_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    };
    auto t_write = [&](int q, unsigned char *const dstb) __attribute__((always_inline)) {
        unsigned char *o = dstb + t_dst[q];
        *reinterpret_cast<unsigned *>(o) = tw[q][0];
        *reinterpret_cast<unsigned *>(o + 2 * XBP * 16) = tw[q][1];
        *reinterpret_cast<unsigned *>(o + 4 * XBP * 16) = tw[q][2];
    };
    auto transform_all = [&](auto PT, int slot) __attribute__((always_inline)) {   // prologue only
        unsigned char *const dstb = bs + slot * GM::B_SLOT;
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            t_read(PT, q, xs); t_xform(PT, q); t_split(q, 0); t_split(q, 1); t_split(q, 2); t_write(q, dstb);
        }
    };

    f32x16 acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // matrix side of a step: per tap group three tap-split fragments x three window-split fragments, six products
    const int b_lane = half * (XBP * 16) + (wn * 32 + l31) * 16;
    // Fragment registers: the tap fragments of two groups (double-buffered), the window fragments of ONE -- the products of a
    // group are ordered so that b2 and b1 die after the first and third instruction and are reloaded for the next group while
    // the current one finishes; b0 follows after the sixth.
    wbf_bf16x8 fa[2][3], fb[3];
    auto f_read_a = [&](int g, const unsigned char *ab) __attribute__((always_inline)) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fa[g & 1][sp] = __builtin_bit_cast(wbf_bf16x8, *reinterpret_cast<const wbf_u32x4 *>(ab + (g * WM * 3 + sp) * 1024));
    };
    auto f_read_b = [&](int g, int sp, const unsigned char *bb) __attribute__((always_inline)) {
        fb[sp] = __builtin_bit_cast(wbf_bf16x8, *reinterpret_cast<const wbf_u32x4 *>(bb + sp * 2 * (XBP * 16) + g * d * 16));
    };

    // ---- prologue --------------------------------------------------------------------------------------------------
    load_x(0, 0);
    dma_a(0);
    if (n_steps > 1) dma_a(1);
    store_x(0, 0);
    load_x(0, 1);
    store_x(0, 1);
    lds_barrier();
    transform_all(std::integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tap fragments of steps 0 and 1 have landed
    lds_barrier();

    // ---- main loop: step s = (chunk c, point pt) -----------------------------------------------------------------------
    // The matrix instructions of a step form ONE dependent chain per wave (one accumulator per point): each waits ~64 cycles
    // for its predecessor while the pipe takes one every 32, so the two waves of a SIMD fill each other's slots -- and the
    // vector / LDS work of the NEXT step's input transform is placed, stage by stage, in this wave's own gaps (the compiler
    // left to itself issues the 18 matrix instructions back to back and the transform after them: measured, the two then add
    // up).  sched_barrier pins the order.
    for (int c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks;
        auto step = [&](auto PT) __attribute__((always_inline)) {
            constexpr int pt = decltype(PT)::value;
            using PN = std::integral_constant<int, (pt + 1) % NP>;
            const int s = c * NP + pt;
            const bool tail = s + 2 >= n_steps;
            const bool dma = !tail && !(DBG & 4);     // the tap fragments of step s + 2 (issued from the gaps below)
            // chunk c + 1's rows: half h is fetched in step 2 h and written to the idle raw buffer in step 2 h + 1; the barriers of
            // steps 3..5 put them in front of the transform of (c + 1, 0), which runs in step 6.  (Both halves requested in step 0 and
            // written in steps 2 and 3 -- one batch of HBM latency per chunk instead of two -- measured 3 % SLOWER at C = 64, same box.)
            constexpr bool LOADS = (pt == 0 || pt == 2) && !(DBG & 16), STORES = (pt == 1 || pt == 3) && !(DBG & 16);
            if (LOADS && more) load_x(c + 1, pt / 2);
            unsigned char *const dstb = bs + ((s + 1) & 1) * GM::B_SLOT;
            const unsigned char *ab = as + (s % 3) * GM::A_SLOT + wm * 3 * 1024 + lane * 16;   // [g][32-channel block wm][split]
            const unsigned char *bb = bs + (s & 1) * GM::B_SLOT + b_lane;
            const wbf_f32x2 *const raw = xs + ((pt == NP - 1 ? c + 1 : c) & 1) * XRAW;    // the chunk of step s + 1
            constexpr bool T = !(DBG & 1), M = !(DBG & 2);
            // filler stage k goes after matrix instruction k of the step
            auto filler = [&](int k) __attribute__((always_inline)) {
                constexpr bool T1 = T && TQ > 1;     // a second transform round (128-column blocks)
                if constexpr (G == 3) {
                    switch (k) {
                        case 0: if (M) f_read_a(1, ab); break;
                        case 1: if (T) t_xform(PN{}, 0); break;
                        case 2: if (M) { f_read_b(1, 2, bb); f_read_b(1, 1, bb); } break;
                        case 3: if (T) t_split(0, 0); if (dma) dma_a1(s + 2, 0); break;
                        case 4: if (T) { t_split(0, 1); t_split(0, 2); } break;
                        case 5: if (M) { f_read_b(1, 0, bb); f_read_a(2, ab); } break;
                        case 6: if (T) t_write(0, dstb); if (T1) t_read(PN{}, TQ - 1, raw); break;
                        case 7: if (STORES && more) store_x1(c + 1, pt / 2, 0); if (dma) dma_a1(s + 2, 1); break;
                        case 8: if (M) { f_read_b(2, 2, bb); f_read_b(2, 1, bb); } break;
                        case 9: if (T1) t_xform(PN{}, TQ - 1); if (dma && UPW > 3) dma_a1(s + 2, 3); break;
                        case 10: if (T1) t_split(TQ - 1, 0); if (dma && UPW > 4) dma_a1(s + 2, 4); break;
                        case 11: if (M) f_read_b(2, 0, bb); break;
                        case 12: if (T1) { t_split(TQ - 1, 1); t_split(TQ - 1, 2); } break;
                        case 13: if (T1) t_write(TQ - 1, dstb); if (dma) dma_a1(s + 2, 2); break;
                        case 14: if (STORES && more) store_x1(c + 1, pt / 2, 1); break;
                        case 15: if (STORES && more) store_x1(c + 1, pt / 2, 2); break;
                        case 16: if (STORES && more) store_x1(c + 1, pt / 2, 3); break;
                        default: break;
                    }
                    static_assert(UPW <= 5, "");
                } else {
                    switch (k) {
                        case 0: if (M) f_read_a(1, ab); break;
                        case 1: if (T) t_xform(PN{}, 0); break;
                        case 2: if (M) { f_read_b(1, 2, bb); f_read_b(1, 1, bb); } break;
                        case 3: if (T) t_split(0, 0); if (dma) dma_a1(s + 2, 0); break;
                        case 4: if (T) { t_split(0, 1); t_split(0, 2); } break;
                        case 5: if (M) f_read_b(1, 0, bb); if (dma) dma_a1(s + 2, 1); break;
                        case 6: if (T) t_write(0, dstb); if (T1) t_read(PN{}, TQ - 1, raw); break;
                        case 7: if (STORES && more) { store_x1(c + 1, pt / 2, 0); store_x1(c + 1, pt / 2, 1); } break;
                        case 8: if (T1) t_xform(PN{}, TQ - 1); if (dma && UPW > 2) dma_a1(s + 2, 2); break;
                        case 9: if (T1) t_split(TQ - 1, 0); break;
                        case 10: if (T1) { t_split(TQ - 1, 1); t_split(TQ - 1, 2); } break;
                        case 11: if (T1) t_write(TQ - 1, dstb); if (STORES && more) { store_x1(c + 1, pt / 2, 2); store_x1(c + 1, pt / 2, 3); } break;
                        default: break;
                    }
                    static_assert(UPW <= 3, "");
                }
            };
            if (M) { f_read_a(0, ab); f_read_b(0, 2, bb); f_read_b(0, 1, bb); f_read_b(0, 0, bb); }
            if (T) t_read(PN{}, 0, raw);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                constexpr int ia[6] = {0, 1, 0, 2, 1, 0}, ib[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    if (M) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][ia[i]], fb[ib[i]], acc[pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    filler(g * 6 + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!M) {   // ablation: the fillers that did not get a slot
#pragma unroll
                for (int k = 6 * G; k < 18; ++k) filler(k);
            }
            static_assert(6 * G >= 12, "twelve filler stages need twelve matrix instructions");
            // everything older than this step's own memory operations has completed: the fragments of step s + 1 are in LDS
            if (tail || (DBG & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (LOADS && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW + 2 * HP * NJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");
            if (!(DBG & 8)) lds_barrier();
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
    }

    // ---- epilogue: y_i = AT diag(1 / N_j) D, + bias, + residual, + running sum, * scale (as wino.hip, R = 4) ----------
    const float *bias = p.bias;
    const float *res = p.res ? p.res + (int64_t)b * c_out * L : nullptr;
    const float *accin = p.accin ? p.accin + (int64_t)b * c_out * L : nullptr;
    float *y = p.y + (int64_t)b * c_out * L;
    const float out_scale = p.out_scale;
    const int col = wn * 32 + l31;
    const int row_l = wm * 32 + 4 * half;
    f32x4 o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = bias ? bias[m0 + row_l + (r & 3) + 8 * (r >> 2)] : 0.f;
        const float t0 = acc[0][r] * -2.f, t1 = acc[1][r] * (-2.f / 3.f), t2 = acc[2][r] * (-2.f / 9.f), t3 = acc[3][r] * (16.f / 9.f),
                    t4 = acc[4][r] * (16.f / 15.f), t5 = acc[5][r] * (2.f / 45.f), t6 = acc[6][r];
        const float s12 = t1 + t2, m12 = t1 - t2, s34 = t3 + t4, m34 = t3 - t4;
        o[r].x = (t0 + s12) + (s34 + t5) + bv;
        o[r].y = fmaf(0.5f, m34, m12) + fmaf(2.f, t5, bv);
        o[r].z = fmaf(0.25f, s34, s12) + fmaf(4.f, t5, bv);
        o[r].w = fmaf(0.125f, m34, m12) + fmaf(8.f, t5, t6) + bv;
    }
    if (DBG & 32) {   // ablation: no epilogue traffic (and the register count of the loop alone)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += o[r].x + o[r].y + o[r].z + o[r].w;
        if (sum == 12345.678f) y[tid] = sum;
        return;
    }
    const bool l4 = (L & 3) == 0;
    if (d == 1 && l4) {
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
    // dilated (a lane's outputs are d apart) or ragged rows: through LDS, so that HBM sees whole contiguous rows
    constexpr int YS = GM::YS;
    float *yt = wbf_smem;
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
    const int64_t t_blk0 = sb0 * 4 * d;
    const int64_t left = L - t_blk0;
    const int n_t = (int)(left < 4 * n_tiles_blk ? left : 4 * n_tiles_blk);
    if (l4) {
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
    } else {
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

template <int KW, int DBG = 0, int BM = 64, int BNT = 128>
static int winobf_launch(WinoBfParams p, hipStream_t stream) {
    using GM = WbfGeom<KW, BM, BNT>;
    p.sb_per_block = (BNT - (GM::G - 1) * p.dil) / p.dil;   // valid tiles + the (G - 1) d windows behind them = BNT transformed windows
    p.n_sb = ceil_div(p.L, (int64_t)4 * p.dil);
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute((const void *)winobf_conv_kernel<KW, DBG, BM, BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    });
    if (err != hipSuccess) return fail("winobf conv: cannot reserve %d bytes of LDS: %s", GM::LDS_BYTES, hipGetErrorString(err));
    p.n_tile_blocks = (int)ceil_div(p.n_sb, p.sb_per_block);
    const int n_m = p.c_out / BM;
    dim3 grid((unsigned)(ceil_div(p.n_tile_blocks, 8) * 8 * n_m), 1, (unsigned)p.batch);
    hipLaunchKernelGGL((winobf_conv_kernel<KW, DBG, BM, BNT>), grid, dim3(WBF_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

// the block shape a layer runs with (the tap fragments are packed for it).  128 channels x 64 columns halves the window
// transforms and raw-row staging per product but loses (G - 1) d of 64 columns instead of 128 to the tap-group overlap:
// measured (profiles/r03_convbf_shapes.txt) it wins 12-17 % on the 7-tap layers and 7 % on 256-channel 11-tap ones, and is
// level (+5 .. -3 %) on 128-channel 11-tap ones, which keep 64 x 128.  RVC_WBF_BM128=0 forces 64 x 128 everywhere.
bool winobf2_enabled() {
    static const int on = knob("RVC_WBF_V2", 1);
    return on != 0;
}

// does this layer take winobf2.hip's form (its fragments are then packed point-major at 128 rows per block)?
static bool winobf_takes_v2(int c_in, int c_out, int k) { return winobf2_enabled() && winobf2_supported(c_in, c_out, k, 1); }

int winobf_block_rows(int c_out, int k) {
    if (winobf2_enabled() && c_out % 128 == 0) return 128;   // the fragments are packed for the kernel that will read them
    static const int wide = knob("RVC_WBF_BM128", 1);
    if (!wide || c_out % 128) return 64;
    return (k == 7 || c_out >= 256 || wide == 2) ? 128 : 64;
}

// (3 taps: winobf2.hip's F(4,3) form only, i.e. c_out % 128 == 0)
bool winobf_supported(int c_in, int c_out, int k, int dil) {
    if (k == 3) return dil >= 1 && dil <= WBF_MAX_DIL && winobf_takes_v2(c_in, c_out, k);
    return (k == 7 || k == 11) && dil >= 1 && dil <= WBF_MAX_DIL && c_in % WBF_CIC == 0 && c_out % 64 == 0;
}

bool winobf_fits(int c_in, int c_out, int64_t L) {
    return (int64_t)c_in * L < ((int64_t)1 << 29) && (int64_t)c_in * c_out * WBF_NP * 3 * 6 < ((int64_t)1 << 31);
}

size_t winobf_weight_bytes(int c_out, int c_in, int k) { return (size_t)c_out * c_in * (k == 3 ? 6 : WBF_NP * ((k + 3) / 4)) * 3 * 2; }

int launch_winobf_conv(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch,
                       int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream) {
    if (!winobf_supported(c_in, c_out, k, dil)) return fail("winobf conv: unsupported shape (%d -> %d channels, %d taps, dilation %d)", c_in, c_out, k, dil);
    if (!(slope >= 0.f && slope <= 1.f)) return fail("winobf conv: leaky slope %g outside [0, 1]", (double)slope);
    if (!winobf_fits(c_in, c_out, L)) return fail("winobf conv: %d x %lld samples exceed the 2 GiB buffer addressing", c_in, (long long)L);
    if (L <= 0 || batch <= 0) return 0;
    if (winobf_takes_v2(c_in, c_out, k)) return launch_winobf2_conv(x, u, bias, res, accin, y, batch, c_in, c_out, L, k, dil, slope, out_scale, stream);
    WinoBfParams p;
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.accin = accin; p.y = y;
    p.c_in = c_in; p.c_out = c_out; p.L = L; p.dil = dil; p.slope = slope; p.out_scale = out_scale; p.batch = batch;
#ifdef RVC_ABLATE
    if (k == 11 && winobf_block_rows(c_out, k) == 64) {   // ablations (wrong results): where does the time go
        static const int dbg = knob("RVC_WBF_DBG", 0);
        switch (dbg) {
            case 1: return winobf_launch<11, 1>(p, stream);
            case 2: return winobf_launch<11, 2>(p, stream);
            case 3: return winobf_launch<11, 3>(p, stream);
            case 4: return winobf_launch<11, 4>(p, stream);
            case 8: return winobf_launch<11, 8>(p, stream);
            case 9: return winobf_launch<11, 9>(p, stream);
            case 13: return winobf_launch<11, 13>(p, stream);
            case 29: return winobf_launch<11, 29>(p, stream);
            case 11: return winobf_launch<11, 11>(p, stream);
            case 15: return winobf_launch<11, 15>(p, stream);
            case 16: return winobf_launch<11, 16>(p, stream);
            case 31: return winobf_launch<11, 31>(p, stream);
            case 32: return winobf_launch<11, 32>(p, stream);
            default: break;
        }
    }
#endif
    if (winobf_block_rows(c_out, k) == 128) return k == 7 ? winobf_launch<7, 0, 128, 64>(p, stream) : winobf_launch<11, 0, 128, 64>(p, stream);
    return k == 7 ? winobf_launch<7>(p, stream) : winobf_launch<11>(p, stream);
}

// w_host [c_out][c_in][k] -> [c_out / BM][c_in / 16][point 7][group G][32-channel block BM / 32][split 3][lane 64][8] bf16
// (BM = winobf_block_rows(c_out, k)):
// the tap transform (Vandermonde rows of the points 0, 1, -1, 1/2, -1/2, 2, inf without their 1 / N_j, which the epilogue
// applies) in float64, rounded to fp32, split exactly into three bf16, in the lane order of a 32x32x16 A fragment
// (lane l: channel l & 31, input channels 8 (l >> 5) .. + 7).
void winobf_pack_host(const float *w_host, int c_out, int c_in, int k, std::vector<uint16_t> *out) {
    const int BM = winobf_block_rows(c_out, k), WM = BM / 32;
    const int R = k == 3 ? 3 : 4, NPT = R + 3;          // 3 taps: F(4,3), one group on the points 0, 1, -1, 2, -2, inf (winobf2.hip only)
    const int G = (k + R - 1) / R, n_chunks = c_in / WBF_CIC, n_m = c_out / BM;
    const bool point_major = winobf_takes_v2(c_in, c_out, k);
    out->assign((size_t)c_out * c_in * NPT * G * 3, 0);
    auto split3 = [](float v, uint16_t s[3]) {
        float r = v;
        for (int i = 0; i < 3; ++i) {
            s[i] = bf16_rne(r);
            uint32_t bits = (uint32_t)s[i] << 16;
            float f;
            memcpy(&f, &bits, 4);
            r -= f;                      // exact in fp32
        }
    };
    for (int mb = 0; mb < n_m; ++mb)
        for (int c = 0; c < n_chunks; ++c)
            for (int pt = 0; pt < NPT; ++pt)
                for (int g = 0; g < G; ++g)
                    for (int mi = 0; mi < WM; ++mi)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int co = mb * BM + mi * 32 + (lane & 31);
                                const int ci = c * WBF_CIC + 8 * (lane >> 5) + e;
                                double w[4];
                                for (int kk = 0; kk < 4; ++kk) {
                                    const int tap = R * g + kk;
                                    w[kk] = tap < k ? (double)w_host[((size_t)co * c_in + ci) * k + tap] : 0.0;
                                }
                                double u;
                                if (R == 3) switch (pt) {   // rows of G3 without their scale factors (the epilogue's)
                                    case 0: u = w[0]; break;
                                    case 1: u = w[0] + w[1] + w[2]; break;
                                    case 2: u = w[0] - w[1] + w[2]; break;
                                    case 3: u = w[0] + 2.0 * w[1] + 4.0 * w[2]; break;
                                    case 4: u = w[0] - 2.0 * w[1] + 4.0 * w[2]; break;
                                    default: u = w[2]; break;
                                }
                                else switch (pt) {
                                    case 0: u = w[0]; break;
                                    case 1: u = w[0] + w[1] + w[2] + w[3]; break;
                                    case 2: u = w[0] - w[1] + w[2] - w[3]; break;
                                    case 3: u = w[0] + 0.5 * w[1] + 0.25 * w[2] + 0.125 * w[3]; break;
                                    case 4: u = w[0] - 0.5 * w[1] + 0.25 * w[2] - 0.125 * w[3]; break;
                                    case 5: u = w[0] + 2.0 * w[1] + 4.0 * w[2] + 8.0 * w[3]; break;
                                    default: u = w[3]; break;
                                }
                                uint16_t s[3];
                                split3((float)u, s);
                                // this kernel walks (chunk, point); winobf2.hip's waves each own a point and walk its chunks
                                const size_t step = point_major ? ((size_t)mb * NPT + pt) * n_chunks + c : ((size_t)mb * n_chunks + c) * NPT + pt;
                                for (int sp = 0; sp < 3; ++sp) {
                                    const size_t piece = ((step * G + g) * WM + mi) * 3 + sp;
                                    (*out)[piece * 512 + lane * 8 + e] = s[sp];
                                }
                            }
}

int winobf_pack_weight(const float *w_host, int c_out, int c_in, int k, void **out_dev) {
    if (!winobf_supported(c_in, c_out, k, 1)) return fail("winobf_pack_weight: unsupported shape");
    std::vector<uint16_t> u;
    winobf_pack_host(w_host, c_out, c_in, k, &u);
    hipError_t e = hipMalloc(out_dev, u.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(*out_dev, u.data(), u.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail("winobf_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv1d_winobf_weight_bytes(int c_out, int c_in, int k, size_t *bytes) {
    if (!bytes || c_in <= 0 || c_out <= 0 || !winobf_supported(c_in, c_out, k, 1))
        return fail("rvc_conv1d_winobf_weight_bytes: c_in must be a multiple of 16, c_out of 64, k 7 or 11 (k 3: c_out a multiple of 128)");
    *bytes = winobf_weight_bytes(c_out, c_in, k);
    return 0;
}

extern "C" int rvc_conv1d_winobf_pack_weight(const float *w_host, int c_out, int c_in, int k, void *u_dev, void *stream) {
    if (!w_host || !u_dev) return fail("rvc_conv1d_winobf_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_conv1d_winobf_weight_bytes(c_out, c_in, k, &bytes)) return 1;
    std::vector<uint16_t> u;
    winobf_pack_host(w_host, c_out, c_in, k, &u);
    hipError_t e = hipMemcpyAsync(u_dev, u.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_conv1d_winobf_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv1d_winobf_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev,
                                         const float *acc_dev, float *y_dev, int batch, int c_in, int c_out, int64_t length, int k,
                                         int dilation, float slope_in, float out_scale, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_conv1d_winobf_forward: null pointer");
    return launch_winobf_conv(x_dev, u_dev, bias_dev, res_dev, acc_dev, y_dev, batch, c_in, c_out, length, k, dilation, slope_in,
                              out_scale, (hipStream_t)stream);
}
