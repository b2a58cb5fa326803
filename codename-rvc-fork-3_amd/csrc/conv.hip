// K3 -- fp32 conv1d as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, same rate as the vector
// ALUs' peak but reached from one wave per SIMD).  This one kernel family carries 97 % of the vocoder's
// 3.5 TFLOP per 30 s utterance: the dilated ResBlock convs (residuals.py:75-86), conv_pre, the polyphase
// form of the ConvTranspose1d upsamplers with the harmonic-source noise conv folded in as extra input
// rows (hifigan_nsf.py:184-202), and RefineGAN's concat + input_conv.
//
// GEMM view:  M = output channels (C_out), N = time, K = taps x input channels.
//   A[m][(tap, ci)] = W[tap][ci][m]                 (weights repacked once to [tap][ci][m]: m contiguous)
//   B[(tap, ci)][t] = act(x[ci][t + tap*dil - padl]) (ONE staged x tile serves every tap: a tap is an offset)
// Per block: BM x BN outputs, 4 waves, each wave MT x NT tiles of 32x32.  K is walked in chunks of 8 input
// channels (all taps of those channels): the chunk's x rows [8][BN + halo] (activation applied once, at
// staging) and weight slab [KW][8][BM] go HBM/L2 -> registers -> LDS while the previous chunk is multiplied.
// Fragment reads are ds_read_b32 with 32 consecutive lanes on consecutive addresses (conflict-free).
// Epilogue fuses bias, the residual add, the running sum of the three parallel ResBlocks and its 1/3.
#include <stdlib.h>

#include "conv.h"

namespace rvc {

constexpr int CONV_CH_ALIGN = 8;  // input channel counts must be multiples of this
constexpr int CONV_MAX_DIL = 5;

// CONV_CIC = input channels per staged chunk
template <int KW, int MT, int NT, int WM, int WN, int CONV_CIC>
__global__ void __launch_bounds__(WM *WN * 64)
conv_mfma_kernel(const ConvParams p) {
    constexpr int BM = 32 * MT * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int NTH = 64 * WM * WN;
    constexpr int XW = BN + (KW - 1) * CONV_MAX_DIL;          // staged row width (enough for dil <= 5)
    constexpr int XTOT = CONV_CIC * XW;
    constexpr int XN = (XTOT + NTH - 1) / NTH;                // staged x floats per thread
    constexpr int W4TOT = KW * CONV_CIC * BM / 4;
    constexpr int WN4 = (W4TOT + NTH - 1) / NTH;              // staged weight float4s per thread

    // two LDS buffers: chunk c+1 is written while other waves still multiply chunk c -> one barrier per chunk
    constexpr int WTOT = KW * CONV_CIC * BM;
    __shared__ __attribute__((aligned(16))) float ws[2 * WTOT];
    __shared__ float xs[2 * XTOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int b = blockIdx.z;
    const int m0 = blockIdx.y * BM;
    const int64_t col0 = (int64_t)blockIdx.x * BN;
    // scalar copies: lambdas that referenced the by-value struct forced it into scratch memory
    const float *const px1 = p.x1 + (int64_t)b * p.x1_bstride;
    const float *const px2 = p.x2 ? p.x2 + (int64_t)b * p.x2_bstride : nullptr;
    const float *const pw = p.w;
    const int c1 = p.c1;
    const int ctot = p.c1 + p.c2;
    const int m_total = p.m_total;
    const int64_t l_in1 = p.l_in;
    const int64_t l_in2 = p.l_in2 ? p.l_in2 : p.l_in;
    const int padl = p.padl;
    const float slope1 = p.slope1, slope2 = p.slope2;
    const int n_chunks = ctot / CONV_CIC;
    const int dil = p.dil;

    float xr[XN];
    float4 wr[WN4];

    auto load_chunk = [&](int c) {
        const int ci0 = c * CONV_CIC;
        const float *src;
        float slope;
        int64_t l_in;
        if (ci0 < c1) {
            l_in = l_in1;
            src = px1 + (int64_t)ci0 * l_in;
            slope = slope1;
        } else {
            l_in = l_in2;
            src = px2 + (int64_t)(ci0 - c1) * l_in;
            slope = slope2;
        }
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = tid + i * NTH;
            float v = 0.f;
            if (idx < XTOT) {
                const int ci = idx / XW;
                const int cc = idx - ci * XW;
                const int64_t t = col0 + cc - padl;
                if (t >= 0 && t < l_in && !(p.debug & 1)) v = lrelu(src[(int64_t)ci * l_in + t], slope);
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            const int idx4 = tid + i * NTH;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx4 < W4TOT) {
                const int row = idx4 / (BM / 4);          // (tap, ci) row of BM floats
                const int c4 = idx4 - row * (BM / 4);
                const int tap = row / CONV_CIC;
                const int ci = row - tap * CONV_CIC;
                v = *reinterpret_cast<const float4 *>(pw + ((int64_t)tap * ctot + ci0 + ci) * m_total + m0 + c4 * 4);
            }
            wr[i] = v;  // unconditional: a partially-defined register array is demoted to scratch
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = tid + i * NTH;
            if (idx < XTOT) xs[buf * XTOT + idx] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            const int idx4 = tid + i * NTH;
            if (idx4 < W4TOT) *reinterpret_cast<float4 *>(&ws[buf * WTOT + idx4 * 4]) = wr[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    if (n_chunks > 1) load_chunk(1);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        const float *wa = &ws[buf * WTOT + half * BM + wm * MT * 32 + l31];
        const float *xb = &xs[buf * XTOT + half * XW + wn * NT * 32 + l31];
#pragma unroll
        for (int tap = 0; tap < KW; ++tap) {
#pragma unroll
            for (int kk = 0; kk < CONV_CIC / 2; ++kk) {
                float a[MT], bb[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = wa[(tap * CONV_CIC + 2 * kk) * BM + m * 32];
#pragma unroll
                for (int n = 0; n < NT; ++n) bb[n] = xb[(2 * kk) * XW + n * 32 + tap * dil];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mfma32(a[m], bb[n], acc[m][n]);
            }
        }
        if (c + 1 < n_chunks) {
            // buffer buf^1 was last read during chunk c-1; every wave passed the barrier that ended it
            store_chunk(buf ^ 1);
            if (c + 2 < n_chunks) load_chunk(c + 2);
            __syncthreads();
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const float *bias = p.bias ? p.bias + (int64_t)b * p.bias_bstride : nullptr;
    float *y = p.y + (int64_t)b * p.y_bstride;
    if (p.up_stride == 0) {
        const float *res = p.res ? p.res + (int64_t)b * p.y_bstride : nullptr;
        const float *accin = p.accin ? p.accin + (int64_t)b * p.y_bstride : nullptr;
        // Two passes.  res/accin may alias y (the ResBlock updates its state in place), so a fused
        // load-add-store loop would have to wait for every load before the next store (measured: ~110 us per block,
        // 380 us per launch); gathering all addends into the accumulators first keeps 64 loads per lane in flight.
        if ((res || accin) && !(p.debug & 4)) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * MT * 32 + m * 32 + mfma32_row(r, lane);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int64_t col = col0 + wn * NT * 32 + n * 32 + l31;
                        if (col < p.n_cols) {
                            const int64_t o = (int64_t)row * p.l_out + col;
                            float add = 0.f;
                            if (res) add = res[o];
                            if (accin) add += accin[o];
                            acc[m][n][r] += add;
                        }
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * MT * 32 + m * 32 + mfma32_row(r, lane);
                const float bv = bias ? bias[row] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int64_t col = col0 + wn * NT * 32 + n * 32 + l31;
                    if (col < p.n_cols && (!(p.debug & 2) || acc[m][n][r] == 12345.678f)) y[(int64_t)row * p.l_out + col] = (acc[m][n][r] + bv) * p.out_scale;
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * MT * 32 + m * 32 + mfma32_row(r, lane);
                const int phase = row / p.c_out;
                const int co = row - phase * p.c_out;
                const float bv = bias ? bias[co] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int64_t col = col0 + wn * NT * 32 + n * 32 + l31;
                    const int64_t t = col * p.up_stride + phase - p.up_pad;
                    if (col < p.n_cols && t >= 0 && t < p.l_out)
                        y[(int64_t)co * p.l_out + t] = (acc[m][n][r] + bv) * p.out_scale;
                }
            }
        }
    }
}

template <int KW, int MT, int NT, int WM, int WN, int CIC>
static int launch_cfg(const ConvParams &p, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM;
    constexpr int BN = 32 * NT * WN;
    dim3 grid((unsigned)ceil_div(p.n_cols, BN), (unsigned)(p.m_total / BM), (unsigned)p.batch);
    hipLaunchKernelGGL((conv_mfma_kernel<KW, MT, NT, WM, WN, CIC>), grid, dim3(64 * WM * WN), 0, stream, p);
    RVC_LAUNCH_CHECK();
    return 0;
}

// tuning knobs for experiments (tools/bench_conv.py): RVC_CONV_CIC=4|8 overrides the chunk depth,
// RVC_CONV_TILE=1 selects the 128x256 block tile where it applies
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int KW, int MT, int NT, int WM, int WN>
static int launch_cic(const ConvParams &p, hipStream_t stream, int cic) {
    switch (cic) {
        case 4: return launch_cfg<KW, MT, NT, WM, WN, 4>(p, stream);
        case 16:
            if constexpr (KW <= 3) return launch_cfg<KW, MT, NT, WM, WN, 16>(p, stream);
        default: return launch_cfg<KW, MT, NT, WM, WN, 8>(p, stream);
    }
}

template <int KW>
static int launch_kw(const ConvParams &p, hipStream_t stream) {
    static const int cic_env = env_int("RVC_CONV_CIC", 0);
    // chunk depth: deep kernels (7, 11 taps) stage 4 channels per chunk (staging registers, 3 blocks/CU);
    // shallow ones amortise the barrier over more channels
    int cic = KW >= 7 ? 4 : 8;
    if (cic_env) cic = cic_env;
    if ((p.c1 % cic) || (p.c2 % cic)) cic = 8;
    if (p.m_total % 128 == 0) return launch_cic<KW, 2, 2, 2, 2>(p, stream, cic);   // 128 x 128
    if (p.m_total % 64 == 0) return launch_cic<KW, 2, 2, 1, 4>(p, stream, cic);    //  64 x 256
    if (p.m_total % 32 == 0) return launch_cic<KW, 1, 4, 1, 4>(p, stream, cic);    //  32 x 512
    return fail("conv: GEMM rows (%d) must be a multiple of 32", p.m_total);
}

int launch_conv(const ConvParams &p_in, hipStream_t stream) {
    static const int dbg = env_int("RVC_CONV_DEBUG", 0);
    ConvParams p = p_in;
    p.debug = dbg;
    if ((p.c1 % CONV_CH_ALIGN) || (p.c2 % CONV_CH_ALIGN) || p.c1 + p.c2 <= 0)
        return fail("conv: input channels (%d + %d) must be multiples of %d", p.c1, p.c2, CONV_CH_ALIGN);
    if (p.dil < 1 || p.dil > CONV_MAX_DIL) return fail("conv: dilation %d out of range 1..%d", p.dil, CONV_MAX_DIL);
    if (p.n_cols <= 0 || p.batch <= 0) return 0;
    switch (p.kw) {
        case 1: return launch_kw<1>(p, stream);
        case 2: return launch_kw<2>(p, stream);
        case 3: return launch_kw<3>(p, stream);
        case 7: return launch_kw<7>(p, stream);
        case 11: return launch_kw<11>(p, stream);
        default: return fail("conv: unsupported kernel size %d (1, 2, 3, 7, 11)", p.kw);
    }
}

int pack_conv_weight(const float *w_host, int c_out, int c_in, int k, float **out_dev) {
    const size_t n = (size_t)c_out * c_in * k;
    std::string err;
    float *tmp = (float *)malloc(n * sizeof(float));
    if (!tmp) return fail("out of host memory");
    for (int co = 0; co < c_out; ++co)
        for (int ci = 0; ci < c_in; ++ci)
            for (int t = 0; t < k; ++t) tmp[((size_t)t * c_in + ci) * c_out + co] = w_host[((size_t)co * c_in + ci) * k + t];
    hipError_t e = hipMalloc((void **)out_dev, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(*out_dev, tmp, n * sizeof(float), hipMemcpyHostToDevice);
    free(tmp);
    if (e != hipSuccess) return fail("pack_conv_weight: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv1d_pack_weight(const float *w_host, int c_out, int c_in, int k, float *w_packed_dev,
                                      void *stream) {
    if (!w_host || !w_packed_dev || c_out <= 0 || c_in <= 0 || k <= 0) return fail("rvc_conv1d_pack_weight: bad argument");
    float *tmp = nullptr;
    if (pack_conv_weight(w_host, c_out, c_in, k, &tmp)) return 1;
    hipError_t e = hipMemcpyAsync(w_packed_dev, tmp, (size_t)c_out * c_in * k * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail("rvc_conv1d_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv1d_forward(const float *x_dev, const float *w_packed_dev, const float *bias_dev,
                                  const float *res_dev, const float *acc_dev, float *y_dev, int batch, int c_in,
                                  int c_out, int64_t length, int k, int dilation, float slope_in, float out_scale,
                                  void *stream) {
    if (!x_dev || !w_packed_dev || !y_dev) return fail("rvc_conv1d_forward: null pointer");
    if (!(k & 1)) return fail("rvc_conv1d_forward: kernel size must be odd, got %d", k);
    ConvParams p;
    p.x1 = x_dev; p.c1 = c_in; p.slope1 = slope_in; p.x1_bstride = (int64_t)c_in * length;
    p.l_in = length;
    p.w = w_packed_dev; p.bias = bias_dev; p.res = res_dev; p.accin = acc_dev;
    p.y = y_dev; p.y_bstride = (int64_t)c_out * length;
    p.m_total = c_out; p.c_out = c_out; p.n_cols = length; p.l_out = length;
    p.kw = k; p.dil = dilation; p.padl = (k - 1) / 2 * dilation;
    p.out_scale = out_scale; p.batch = batch;
    return launch_conv(p, (hipStream_t)stream);
}
