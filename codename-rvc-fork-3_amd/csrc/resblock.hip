// Fused ResBlock layer for the narrowest vocoder stage (C = 32; C = 64 is built but off by default):
//     y = conv2( leaky( conv1_d( leaky(x) ) ) ) + x  [+ running sum] [* 1/3]          (residuals.py:75-86, one dilation)
// in ONE launch.  At C = 32 a conv is only 64-230 flop per byte and its HBM phases are not hidden behind the MFMAs, so
// the two convs of a layer as separate launches (5 tensor passes) lose to one launch that reads the activation once
// and writes it once (2 passes): 238 / 453 / 676 us per layer at k = 3 / 7 / 11 against 2 x 187 / 272 / 356 us.
// At C = 64 the wider unfused tiles win (464 vs 2 x 211 us at k = 3), see resblock_layer_supported().
//
// Block = 4 waves.  LDS holds the RAW input tile xs[C][N1 + (K-1)*dil] (leaky is applied when fragments are read,
// 2 VALU ops per MFMA operand, because the residual needs the raw values), the intermediate tile ts[C][N1 + K - 1]
// and a double-buffered slab of the current conv's weights.  N1 = columns of conv1 computed by the block (128 at
// C = 32, 64 at C = 64: sized so that 2-3 blocks share a CU's 160 KB); the block emits BN = N1 - (K - 1) output
// columns, i.e. both GEMMs run N1 wide and the K - 1 overlapping columns are recomputed by the neighbour.
#include <stdlib.h>

#include <mutex>

#include "conv.h"

namespace rvc {

// N1 = conv1 columns per block; the 4 waves are arranged WM (output-channel tiles) x WN (column tiles)
template <int KW, int C, int N1, int WM>
struct FusedCfg {
    static constexpr int WN = 4 / WM;
    static constexpr int MT = C / 32 / WM;            // row tiles per wave
    static constexpr int NT = N1 / 32 / WN;           // column tiles per wave
    static constexpr int CIC = KW <= 3 ? 16 : 4;      // input channels per weight slab
    static constexpr int XW = N1 + (KW - 1) * 5;      // staged input row (dilation <= 5)
    static constexpr int TW = N1 + (KW - 1);          // intermediate row
    static constexpr int WSLAB = KW * CIC * C;
    static_assert(MT >= 1 && NT >= 1, "tile split");
};

template <int KW, int C, int N1, int WM>
__global__ void __launch_bounds__(256)
resblock_layer_kernel(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1,
                      const float *__restrict__ w2, const float *__restrict__ b2, const float *accin,
                      float *y, int64_t L, int dil, float slope, float out_scale) {
    using F = FusedCfg<KW, C, N1, WM>;
    constexpr int MT = F::MT, NT = F::NT, WN = F::WN, CIC = F::CIC, XW = F::XW, TW = F::TW, WSLAB = F::WSLAB;
    constexpr int BN = N1 - (KW - 1);
    constexpr int H2 = (KW - 1) / 2;
    constexpr int W4 = WSLAB / 4;
    constexpr int WN4 = (W4 + 255) / 256;
    constexpr int NCH = C / CIC;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ws = smem;                       // [2][WSLAB]
    float *xs = ws + 2 * WSLAB;             // [C][XW]   raw input
    float *ts = xs + C * XW;                // [C][TW]   leaky(conv1 + b1), zero outside [0, L)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = wm * MT * 32, colw = wn * NT * 32;   // this wave's output-channel and column offsets
    const int64_t b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * BN;          // first output column of the block
    const int h1 = H2 * dil;
    const float *xb = x + b * C * L;
    const int64_t x_time0 = t0 - H2 - h1;                 // time of xs[.][0]

    // ---- stage the input tile: wave w takes rows w, w+4, ...; lanes run along time (256 B per load instruction) ----
    // Loads are unconditional (clamped address, masked at the LDS store) and issued RB rows at a time, so RB * CJ of
    // them are in flight per lane; a load inside a divergent branch would be waited for before the branch closes.
    {
        constexpr int RPW = C / 4;                 // rows per wave
        constexpr int CJ = (XW + 63) / 64;         // 64-column pieces per row
        constexpr int RB = 8;
        static_assert(RPW % RB == 0, "row batches");
        const int xw_need = N1 + (KW - 1) * dil;   // columns the two convs and the residual actually read
#pragma unroll 1
        for (int rb = 0; rb < RPW; rb += RB) {
            float v[RB][CJ];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float *rowp = xb + (int64_t)(wave + 4 * (rb + r)) * L;
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    int64_t t = x_time0 + lane + 64 * j;
                    t = t < 0 ? 0 : (t >= L ? L - 1 : t);
                    v[r][j] = rowp[t];
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int col = lane + 64 * j;
                    const int64_t t = x_time0 + col;
                    if (col < xw_need) xs[(wave + 4 * (rb + r)) * XW + col] = (t >= 0 && t < L) ? v[r][j] : 0.f;
                }
        }
    }
    for (int idx = tid; idx < C * (KW - 1); idx += 256) {  // tail columns of ts read by the wasted conv2 columns
        const int ci = idx / (KW - 1), u = N1 + idx - ci * (KW - 1);
        ts[ci * TW + u] = 0.f;
    }

    f32x4 wr[WN4];  // native vector type: copies of HIP's float4 struct become memcpys that pin the array in scratch
    auto load_w = [&](const float *w, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            int idx4 = tid + i * 256;
            if ((i + 1) * 256 > W4) idx4 = idx4 < W4 ? idx4 : W4 - 1;
            const int row = idx4 / (C / 4), c4 = idx4 - row * (C / 4);   // row = tap * CIC + ci
            const int tap = row / CIC, ci = row - tap * CIC;
            wr[i] = *reinterpret_cast<const f32x4 *>(w + ((int64_t)tap * C + c * CIC + ci) * C + c4 * 4);
        }
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            const int idx4 = tid + i * 256;
            if (idx4 < W4) *reinterpret_cast<f32x4 *>(&ws[buf * WSLAB + idx4 * 4]) = wr[i];
        }
    };

    f32x16 acc[MT][NT];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    };
    // one conv as NCH weight slabs; B fragments come from `src` (row stride `sw`), tap offset `td`, leaky on read
    auto run_conv = [&](const float *w, const float *src, int sw, int td, bool act) __attribute__((always_inline)) {
        zero_acc();
        load_w(w, 0);
        store_w(0);
        if (NCH > 1) load_w(w, 1);
        __syncthreads();
        for (int c = 0; c < NCH; ++c) {
            const int buf = c & 1;
            const float *wa = &ws[buf * WSLAB + half * C + row0 + l31];
            const float *sb = &src[(c * CIC + half) * sw + colw + l31];
            // LDS fragments one k-step ahead of the MFMAs (see conv.hip)
            constexpr int STEPS = KW * (CIC / 2);
            float a[2][MT], bb[2][NT];
            auto frag = [&](int st, float (&av)[MT], float (&bv)[NT]) __attribute__((always_inline)) {
                const int tap = st / (CIC / 2), kk = st - tap * (CIC / 2);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = wa[(tap * CIC + 2 * kk) * C + m * 32];
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[n] = sb[(2 * kk) * sw + n * 32 + tap * td];
            };
            frag(0, a[0], bb[0]);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + 1 < STEPS) frag(st + 1, a[(st + 1) & 1], bb[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[st & 1][n] = act ? lrelu(bb[st & 1][n], slope) : bb[st & 1][n];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma32(a[st & 1][m], bb[st & 1][n], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 1 < NCH) {
                store_w(buf ^ 1);
                if (c + 2 < NCH) load_w(w, c + 2);
                lds_barrier();   // slab c + 2's loads stay in flight (conv.hip)
            }
        }
    };

    // ---- conv1 (dilated) -> ts ----
    run_conv(w1, xs, XW, dil, true);   // the barrier inside also publishes xs
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + m * 32 + mfma32_row(r, lane);
            const float bv = b1[row];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int u = colw + n * 32 + l31;
                const int64_t t = t0 - H2 + u;
                // conv2 zero-pads ITS input: positions outside the signal are 0, not conv1 of the padding
                ts[row * TW + u] = (t >= 0 && t < L) ? lrelu(acc[m][n][r] + bv, slope) : 0.f;
            }
        }
    __syncthreads();   // ts complete; every wave is done with the conv1 weight slabs

    // ---- conv2 (dilation 1) + bias + residual (+ running sum, scale) -> y ----
    run_conv(w2, ts, TW, 1, false);
    float *yb = y + b * C * L;
    const float *ab = accin ? accin + b * C * L : nullptr;
    // (conv2 + bias) + x, the residual read back from the staged raw tile
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + m * 32 + mfma32_row(r, lane);
            const float bv = b2[row];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n][r] = acc[m][n][r] + bv + xs[row * XW + colw + n * 32 + l31 + H2 + h1];
        }
    // uniform base + 32-bit byte offsets (launch checks C * L < 2^30), see conv.hip's epilogue
    const uint32_t lrow_b = (uint32_t)L * 4u;
    bool ok[NT];
    uint32_t cb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int j = colw + n * 32 + l31;
        ok[n] = j < BN && t0 + j < L;
        // clamped column: loads of masked-off lanes stay inside the row, so they can be unconditional
        int64_t tc = t0 + (j < BN ? j : BN - 1);
        tc = tc < L ? tc : L - 1;
        cb[n] = (uint32_t)tc * 4u;
    }
    auto boff = [&](int m, int r, int n) __attribute__((always_inline)) {
        return (uint32_t)(row0 + 4 * half + m * 32 + (r & 3) + 8 * (r >> 2)) * lrow_b + cb[n];
    };
    if (ab) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n][r] += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ab) + boff(m, r, n));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (ok[n]) *reinterpret_cast<float *>(reinterpret_cast<char *>(yb) + boff(m, r, n)) = acc[m][n][r] * out_scale;
}

template <int KW, int C, int N1, int WM>
static int launch_fused(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *accin,
                        float *y, int batch, int64_t L, int dil, float slope, float out_scale, hipStream_t stream) {
    using F = FusedCfg<KW, C, N1, WM>;
    constexpr int BN = N1 - (KW - 1);
    const size_t lds = (size_t)(2 * F::WSLAB + C * F::XW + C * F::TW) * sizeof(float);
    static std::once_flag attr_once;      // per template instance; forwards run on several host threads
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [lds] {
        attr_err = hipFuncSetAttribute((const void *)resblock_layer_kernel<KW, C, N1, WM>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) return fail("resblock_layer: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(attr_err));
    hipLaunchKernelGGL((resblock_layer_kernel<KW, C, N1, WM>), dim3((unsigned)ceil_div(L, BN), batch), dim3(256), lds, stream, x, w1,
                       b1, w2, b2, accin, y, L, dil, slope, out_scale);
    RVC_LAUNCH_CHECK();
    return 0;
}

bool resblock_layer_supported(int c, int k) {
    static const int off = knob("RVC_NO_FUSED_RESBLOCK", 0);
    // measured on MI355X (profiles/r01_decoder_kernels.txt): fused beats two launches for every k at C = 32 (238 vs 2 x 187 us at
    // k = 3, 453 vs 2 x 272 at k = 7, 676 vs 2 x 356 at k = 11); at C = 64 the wider unfused tiles win (k = 3: 464 vs 2 x 211 us)
    static const int c64 = knob("RVC_FUSED_C64", 0);
    // Against the fast (Winograd) form of the unfused convs (wino.hip) the fusion only still pays at k = 3: 238 us vs 128 + 139;
    // k = 7: 453 vs 181 + 186, k = 11: 676 vs 214 + 231.
    const bool fast = wino_enabled();
    return !off && ((c == 32 && (k == 3 || (!fast && (k == 7 || k == 11)))) || (c64 && c == 64 && k == 3));
}

// x, y: [batch][c][L] (must NOT alias: blocks read their neighbours' columns); w1/w2 packed [k][c][c]
int launch_resblock_layer(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *accin,
                          float *y, int batch, int c, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream) {
    if (x == y) return fail("resblock_layer: in-place operation is not supported");
    if (dil < 1 || dil > 5) return fail("resblock_layer: dilation %d out of range", dil);
    if ((int64_t)c * L >= ((int64_t)1 << 30)) return fail("resblock_layer: a %d x %lld slab exceeds the 4 GB the kernel addresses", c, (long long)L);
    static const int wide64 = knob("RVC_FUSED_C64_WIDE", 0);
#define RVC_FUSED_CASE(KW, CC, NN, WMM) if (k == KW && c == CC) return launch_fused<KW, CC, NN, WMM>(x, w1, b1, w2, b2, accin, y, batch, L, dil, slope, out_scale, stream)
    RVC_FUSED_CASE(3, 32, 128, 1); RVC_FUSED_CASE(7, 32, 128, 1); RVC_FUSED_CASE(11, 32, 128, 1);
    if (wide64) { RVC_FUSED_CASE(3, 64, 128, 1); RVC_FUSED_CASE(7, 64, 128, 1); RVC_FUSED_CASE(11, 64, 128, 1); }
    RVC_FUSED_CASE(3, 64, 64, 2); RVC_FUSED_CASE(7, 64, 64, 2); RVC_FUSED_CASE(11, 64, 64, 2);
#undef RVC_FUSED_CASE
    return fail("resblock_layer: unsupported shape c=%d k=%d", c, k);
}

}  // namespace rvc
