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

#include <atomic>

#include "conv.h"

namespace rvc {

constexpr int CONV_CH_ALIGN = 8;  // input channel counts must be multiples of this
constexpr int CONV_MAX_DIL = 5;

// CONV_CIC = input channels per staged chunk
// UP: scatter-mode epilogue (polyphase ConvTranspose1d), see conv.h
// WB16: the weight slab is read from HBM as bf16 (p.w16) and widened to fp32 on its way into LDS -- the arithmetic is the
// fp32 one on bf16-valued weights, bit for bit what the fp32 path computes on the same (bf16-rounded) values
template <int KW, int MT, int NT, int WM, int WN, int CONV_CIC, bool UP, bool WB16 = false>
__global__ void __launch_bounds__(WM *WN * 64) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 2 : 3, MT == 1 ? 2 : 3)))
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
    const uint16_t *const pw16 = p.w16;
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
    f32x4 wr[WB16 ? 1 : WN4];  // native vector, not HIP's float4 struct: struct copies become memcpys that keep the array in scratch
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 wr16[WB16 ? WN4 : 1];   // four bf16 weights per 8-byte load

    // Every global load below is unconditional (clamped address, value masked later): a load inside a divergent
    // branch makes the compiler wait for it (s_waitcnt vmcnt(0)) before the branch closes, which serialised the
    // five loads of a chunk -- ~100 us per launch on the 196 MB ResBlock tensors.  The activation is applied when
    // the chunk is stored to LDS, one chunk period later, so nothing here depends on the loaded values.
    auto chunk_src = [&](int c, const float *&src, int64_t &l_in, float &slope) __attribute__((always_inline)) {
        const int ci0 = c * CONV_CIC;
        if (ci0 < c1) {
            l_in = l_in1;
            src = px1 + (int64_t)ci0 * l_in;
            slope = slope1;
        } else {
            l_in = l_in2;
            src = px2 + (int64_t)(ci0 - c1) * l_in;
            slope = slope2;
        }
    };
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int ci0 = c * CONV_CIC;
        const float *src;
        float slope;
        int64_t l_in;
        chunk_src(c, src, l_in, slope);
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            int idx = tid + i * NTH;
            if ((i + 1) * NTH > XTOT) idx = idx < XTOT ? idx : XTOT - 1;
            const int ci = idx / XW;
            const int cc = idx - ci * XW;
            int64_t t = col0 + cc - padl;
            t = t < 0 ? 0 : (t >= l_in ? l_in - 1 : t);
            xr[i] = src[(int64_t)ci * l_in + t];
        }
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            int idx4 = tid + i * NTH;
            if ((i + 1) * NTH > W4TOT) idx4 = idx4 < W4TOT ? idx4 : W4TOT - 1;
            const int row = idx4 / (BM / 4);          // (tap, ci) row of BM floats
            const int c4 = idx4 - row * (BM / 4);
            const int tap = row / CONV_CIC;
            const int ci = row - tap * CONV_CIC;
            const int64_t woff = ((int64_t)tap * ctot + ci0 + ci) * m_total + m0 + c4 * 4;
            if constexpr (WB16) wr16[i] = *reinterpret_cast<const u32x2 *>(pw16 + woff);
            else wr[i] = *reinterpret_cast<const f32x4 *>(pw + woff);
        }
    };
    auto store_chunk = [&](int buf, int c) __attribute__((always_inline)) {
        const float *src;
        float slope;
        int64_t l_in;
        chunk_src(c, src, l_in, slope);
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = tid + i * NTH;
            if (idx < XTOT) {
                const int ci = idx / XW;
                const int64_t t = col0 + (idx - ci * XW) - padl;
                xs[buf * XTOT + idx] = (t >= 0 && t < l_in) ? lrelu(xr[i], slope) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < WN4; ++i) {
            const int idx4 = tid + i * NTH;
            if (idx4 < W4TOT) {
                if constexpr (WB16) {
                    const f32x4 wv = {__uint_as_float(wr16[i].x << 16), __uint_as_float(wr16[i].x & 0xffff0000u),
                                      __uint_as_float(wr16[i].y << 16), __uint_as_float(wr16[i].y & 0xffff0000u)};
                    *reinterpret_cast<f32x4 *>(&ws[buf * WTOT + idx4 * 4]) = wv;
                } else {
                    *reinterpret_cast<f32x4 *>(&ws[buf * WTOT + idx4 * 4]) = wr[i];
                }
            }
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
    store_chunk(0, 0);
    if (n_chunks > 1) load_chunk(1);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        const float *wa = &ws[buf * WTOT + half * BM + wm * MT * 32 + l31];
        const float *xb = &xs[buf * XTOT + half * XW + wn * NT * 32 + l31];
        // fragments of k-step s+1 are read from LDS before the MFMAs of step s are issued (register double buffer):
        // left to itself the compiler emits read, s_waitcnt lgkmcnt(0), 4 x MFMA per step and every step eats the
        // LDS latency
        constexpr int STEPS = KW * (CONV_CIC / 2);
        float a[2][MT], bb[2][NT];
        auto frag = [&](int s, float (&av)[MT], float (&bv)[NT]) __attribute__((always_inline)) {
            const int tap = s / (CONV_CIC / 2), kk = s - tap * (CONV_CIC / 2);
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = wa[(tap * CONV_CIC + 2 * kk) * BM + m * 32];
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = xb[(2 * kk) * XW + n * 32 + tap * dil];
        };
        frag(0, a[0], bb[0]);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS) frag(st + 1, a[(st + 1) & 1], bb[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads of step st+1 ahead of the MFMAs of step st
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mfma32(a[st & 1][m], bb[st & 1][n], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < n_chunks) {
            // buffer buf^1 was last read during chunk c-1; every wave passed the barrier that ended it
            store_chunk(buf ^ 1, c + 1);
            if (c + 2 < n_chunks) load_chunk(c + 2);
            lds_barrier();   // not __syncthreads(): chunk c + 2's loads stay in flight across the barrier
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const float *bias = p.bias ? p.bias + (int64_t)b * p.bias_bstride : nullptr;
    float *y = p.y + (int64_t)b * p.y_bstride;
    if constexpr (!UP) {
        const float *res = p.res ? p.res + (int64_t)b * p.y_bstride : nullptr;
        const float *accin = p.accin ? p.accin + (int64_t)b * p.y_bstride : nullptr;
        const int64_t n_cols = p.n_cols, l_out = p.l_out;
        const float out_scale = p.out_scale;
        const int row0 = m0 + wm * MT * 32;
        // Element addresses as uniform base + 32-bit byte offset (launch_conv checks that one [C_out, L] slab stays
        // below 4 GB): global_load/store take the base from SGPRs and one VGPR offset, so 64 loads in flight cost 64
        // destination registers, not 64 + 128 for 64-bit addresses (that version ran at one wave per SIMD).
        const int64_t cbase = col0 + wn * NT * 32 + l31;
        const int64_t base = (int64_t)(row0 + 4 * half) * l_out + cbase;
        const uint32_t base_b = (uint32_t)base * 4u;
        const uint32_t lrow_b = (uint32_t)l_out * 4u;
        auto boff = [&](int m, int r, int n) __attribute__((always_inline)) {
            return base_b + (uint32_t)(m * 32 + (r & 3) + 8 * (r >> 2)) * lrow_b + (uint32_t)(n * 128);
        };
        auto ld = [](const float *q, uint32_t byte_off) __attribute__((always_inline)) {
            return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(q) + byte_off);
        };
        // reference order of the adds: (conv + bias) + residual, then the running sum of the parallel ResBlocks
        if (bias) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias[row0 + m * 32 + mfma32_row(r, lane)];
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n][r] += bv;
                }
        }
        if (col0 + BN <= n_cols) {
            // interior tile: no bounds checks, so the 64 loads per addend are issued back to back and stay in
            // flight together (a load inside a divergent branch is waited for before the branch closes).
            // res/accin may alias y (the ResBlock updates its state in place): all loads precede all stores.
            if (!(p.debug & 4)) {
                if (res) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc[m][n][r] += ld(res, boff(m, r, n));
                }
                if (accin) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc[m][n][r] += ld(accin, boff(m, r, n));
                }
            }
            if (!(p.debug & 2)) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            *reinterpret_cast<float *>(reinterpret_cast<char *>(y) + boff(m, r, n)) = acc[m][n][r] * out_scale;
            } else {  // experiment knob: keep every accumulator live without storing the tile
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
                if (sum == 12345.678f) y[0] = sum;
            }
        } else {
            // the last tile of a row: same arithmetic, per-element bounds checks (one block column per launch)
            bool ok[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) ok[n] = cbase + n * 32 < n_cols;
            if (res) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (ok[n]) acc[m][n][r] += ld(res, boff(m, r, n));
            }
            if (accin) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (ok[n]) acc[m][n][r] += ld(accin, boff(m, r, n));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        if (ok[n]) *reinterpret_cast<float *>(reinterpret_cast<char *>(y) + boff(m, r, n)) = acc[m][n][r] * out_scale;
        }
    } else {
        // polyphase ConvTranspose1d: GEMM row = (phase, channel); output time = col * stride + phase - pad
        const int c_out = p.c_out, up_stride = p.up_stride, up_pad = p.up_pad;
        const int64_t n_cols = p.n_cols, l_out = p.l_out;
        const float out_scale = p.out_scale;
        const int64_t cbase = col0 + wn * NT * 32 + l31;
        const float *bsrc = bias ? bias : p.w;  // always a valid address: the loads below stay unconditional
        if (p.up_interleave) {
            // rows = (channel, phase), phase innermost, even rate: registers r, r + 1 of a lane (r even) are two neighbouring phases
            // of one channel at one input column, i.e. outputs t, t + 1 -- one 8-byte store; with rate % 4 == 0 the four
            // registers of a group are four neighbouring phases -- one 16-byte store.  (The (phase, channel) order below writes
            // single floats `rate` floats apart and comes back for the gaps rate - 1 times.)
            typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const bool quad = (up_stride & 3) == 0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = m0 + wm * MT * 32 + m * 32 + mfma32_row(r, lane);
                    const int co = row / up_stride, ph = row - co * up_stride;
                    float bv = bsrc[co];
                    bv = bias ? bv : 0.f;
                    float *q = y + (int64_t)co * l_out + ph - up_pad;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int64_t col = cbase + n * 32;
                        const int64_t t = col * up_stride + ph - up_pad;
                        const float v0 = (acc[m][n][r] + bv) * out_scale, v1 = (acc[m][n][r + 1] + bv) * out_scale;
                        if (col >= n_cols) continue;
                        if (quad) {
                            if ((r & 2) == 0) {
                                const float v2 = (acc[m][n][r + 2] + bv) * out_scale, v3 = (acc[m][n][r + 3] + bv) * out_scale;
                                if (t >= 0 && t + 3 < l_out) {
                                    *reinterpret_cast<f32x4u *>(q + col * up_stride) = f32x4u{v0, v1, v2, v3};
                                } else {
                                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (t + e >= 0 && t + e < l_out) q[col * up_stride + e] = vv[e];
                                }
                            }
                        } else if (t >= 0 && t + 1 < l_out) {
                            *reinterpret_cast<f32x2u *>(q + col * up_stride) = f32x2u{v0, v1};
                        } else {
                            if (t >= 0 && t < l_out) q[col * up_stride] = v0;
                            if (t + 1 >= 0 && t + 1 < l_out) q[col * up_stride + 1] = v1;
                        }
                    }
                }
            return;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * MT * 32 + m * 32 + mfma32_row(r, lane);
                const int phase = row / c_out;
                const int co = row - phase * c_out;
                float bv = bsrc[co];
                bv = bias ? bv : 0.f;
                float *q = y + (int64_t)co * l_out + phase - up_pad;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int64_t col = cbase + n * 32;
                    const int64_t t = col * up_stride + phase - up_pad;
                    if (col < n_cols && t >= 0 && t < l_out) q[col * up_stride] = (acc[m][n][r] + bv) * out_scale;
                }
            }
    }
}

static std::atomic<int> g_concurrency{1};   // rvc_set_concurrency_hint: process-wide default
static thread_local int t_concurrency = 0;  // > 0 while the calling thread is inside a forward of a handle that has its own hint
static int concurrency_hint() { return t_concurrency > 0 ? t_concurrency : g_concurrency.load(std::memory_order_relaxed); }
ConcurrencyScope::ConcurrencyScope(int hint) : saved(t_concurrency) { if (hint > 0) t_concurrency = hint; }
ConcurrencyScope::~ConcurrencyScope() { t_concurrency = saved; }

template <int KW, int MT, int NT, int WM, int WN, int CIC>
static int launch_cfg(const ConvParams &p, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM;
    constexpr int BN = 32 * NT * WN;
    dim3 grid((unsigned)ceil_div(p.n_cols, BN), (unsigned)(p.m_total / BM), (unsigned)p.batch);
    if (p.up_stride > 0) {
        if constexpr (KW == 2) {
            hipLaunchKernelGGL((conv_mfma_kernel<KW, MT, NT, WM, WN, CIC, true>), grid, dim3(64 * WM * WN), 0, stream, p);
        } else {
            return fail("conv: scatter mode is built for 2-tap (polyphase) kernels only, got %d taps", KW);
        }
    } else if (p.w16) {
        // bf16 weight slabs exist for the ResBlock / MRF-layer shapes only (3, 7, 11 taps at their usual chunk depth)
        if constexpr ((KW == 3 && CIC == 8) || ((KW == 7 || KW == 11) && CIC == 4)) {
            hipLaunchKernelGGL((conv_mfma_kernel<KW, MT, NT, WM, WN, CIC, false, true>), grid, dim3(64 * WM * WN), 0, stream, p);
        } else {
            return fail("conv: no bf16-weight kernel for %d taps at chunk depth %d", KW, CIC);
        }
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<KW, MT, NT, WM, WN, CIC, false>), grid, dim3(64 * WM * WN), 0, stream, p);
    }
    RVC_LAUNCH_CHECK();
    return 0;
}

// tuning knobs for experiments (tools/bench_conv.py, ablation build only): RVC_CONV_CIC=4|8 overrides the chunk depth

template <int KW, int MT, int NT, int WM, int WN>
static int launch_cic(const ConvParams &p, hipStream_t stream, int cic) {
    // 16-channel chunks were measured too (RVC_CONV_CIC=16 in an earlier build): slower everywhere (LDS per block, spills)
#ifdef RVC_ABLATE
    if (cic == 4) return launch_cfg<KW, MT, NT, WM, WN, 4>(p, stream);
    return launch_cfg<KW, MT, NT, WM, WN, 8>(p, stream);
#else
    // the product build instantiates only the depth launch_kw picks: 4 for 7 / 11 taps, 8 below (the other pairing misses
    // its register target: "desired occupancy was 3, final occupancy is 1")
    (void)cic;
    return launch_cfg<KW, MT, NT, WM, WN, (KW >= 7 ? 4 : 8)>(p, stream);
#endif
}

template <int KW>
static int launch_kw(const ConvParams &p, hipStream_t stream) {
    static const int cic_env = knob("RVC_CONV_CIC", 0);
    // chunk depth: deep kernels (7, 11 taps) stage 4 channels per chunk (staging registers, 3 blocks/CU);
    // shallow ones amortise the barrier over more channels
    int cic = KW >= 7 ? 4 : 8;
    if (cic_env && !p.w16) cic = cic_env;
    if ((p.c1 % cic) || (p.c2 % cic)) cic = 8;
    if (p.m_total % 128 == 0) {
        // few 128 x 128 tiles (the 38k-column first vocoder stage: 600 tiles over 768 block slots, 3 on some CUs and 2
        // on the others) -> halve the tile width so the CUs finish together
        // ... unless the caller keeps several utterances in flight (rvc_set_concurrency_hint): other streams fill the idle
        // block slots, and the wide tile streams each weight slab through L2 half as often (-0.4 ms per utterance)
        static const int narrow_env = knob("RVC_CONV_NARROW", 1);
        const int64_t tiles = ceil_div(p.n_cols, 128) * (p.m_total / 128) * p.batch;
        if (narrow_env && tiles < 1536 && concurrency_hint() <= 1) return launch_cic<KW, 2, 1, 2, 2>(p, stream, cic);   // 128 x 64
        return launch_cic<KW, 2, 2, 2, 2>(p, stream, cic);                                   // 128 x 128
    }
    if (p.m_total % 64 == 0) return launch_cic<KW, 2, 2, 1, 4>(p, stream, cic);    //  64 x 256
    if (p.m_total % 32 == 0) return launch_cic<KW, 1, 4, 1, 4>(p, stream, cic);    //  32 x 512
    return fail("conv: GEMM rows (%d) must be a multiple of 32", p.m_total);
}

// The ResBlock layers take the fast (Winograd) form wherever wino.hip supports the shape: on MI355X it is 1.11-1.62x the
// direct implicit GEMM on every one of the 24 (C, K, dilation) shapes of the 48 kHz vocoders (tools/bench_conv.py,
// profiles/r02_conv_shapes.txt).  RVC_WINO=0 switches it off (A/B runs, and the direct form's own tests).
bool wino_enabled() {
    static const int mode = knob("RVC_WINO", 1);
    return mode != 0;
}

// The 7- / 11-tap layers at >= 64 channels run the same Winograd form on the bf16 matrix cores with fp32 operands split
// exactly into three bf16 (winobf.hip): 1.06-1.36x the fp32-matrix-instruction form on every such shape of the 48 kHz
// vocoders (tools/bench_convbf.py, profiles/r03_convbf_shapes.txt).  RVC_WINOBF=0 switches it off.
bool winobf_enabled() {
    static const int mode = knob("RVC_WINOBF", 1);
    return mode != 0;
}

int launch_conv(const ConvParams &p_in, hipStream_t stream) {
    // one dense input, same length out, per-channel bias: what the Winograd kernels take (winobf.hip for any channel pair,
    // wino.hip for square layers)
    const bool plain = !p_in.x2 && p_in.up_stride == 0 && p_in.bias_bstride == 0 &&
        p_in.l_in == p_in.l_out && p_in.n_cols == p_in.l_out && p_in.x1_bstride == (int64_t)p_in.c1 * p_in.l_in &&
        p_in.y_bstride == (int64_t)p_in.m_total * p_in.l_out && p_in.slope1 >= 0.f && p_in.slope1 <= 1.f && p_in.padl == (p_in.kw - 1) / 2 * p_in.dil;
    if (p_in.w_direct1 && plain && p_in.c1 == p_in.m_total && p_in.x1 != p_in.y && convbf1_supported(p_in.c1, p_in.kw, p_in.dil) && convbf1_fits(p_in.c1, p_in.l_in))
        return launch_convbf1(p_in.x1, p_in.w_direct1, p_in.bias, p_in.res, p_in.accin, p_in.y, p_in.batch, p_in.c1, p_in.l_out, p_in.kw, p_in.dil,
                              p_in.slope1, p_in.out_scale, stream);
    if (p_in.w_winobf && plain && wino_enabled() && winobf_enabled() && winobf_supported(p_in.c1, p_in.m_total, p_in.kw, p_in.dil) &&
        winobf_fits(p_in.c1, p_in.m_total, p_in.l_in))
        return launch_winobf_conv(p_in.x1, p_in.w_winobf, p_in.bias, p_in.res, p_in.accin, p_in.y, p_in.batch, p_in.c1, p_in.m_total, p_in.l_out,
                                  p_in.kw, p_in.dil, p_in.slope1, p_in.out_scale, stream);
    if ((p_in.w_wino || p_in.w_wino16) && !p_in.x2 && p_in.up_stride == 0 && p_in.c1 == p_in.m_total && p_in.bias_bstride == 0 &&
        p_in.l_in == p_in.l_out && p_in.n_cols == p_in.l_out && p_in.x1_bstride == (int64_t)p_in.c1 * p_in.l_in &&
        p_in.y_bstride == (int64_t)p_in.m_total * p_in.l_out && p_in.c1 % 32 == 0 && p_in.slope1 >= 0.f && p_in.slope1 <= 1.f && p_in.padl == (p_in.kw - 1) / 2 * p_in.dil &&
        wino_enabled() && wino_supported(p_in.kw, p_in.dil) && wino_fits(p_in.c1, p_in.m_total, p_in.l_in))
        return launch_wino_conv(p_in.x1, p_in.w_wino16 ? (const void *)p_in.w_wino16 : (const void *)p_in.w_wino, p_in.w_wino16 != nullptr, p_in.bias, p_in.res, p_in.accin, p_in.y, p_in.batch, p_in.c1, p_in.m_total, p_in.l_out,
                                p_in.kw, p_in.dil, p_in.slope1, p_in.out_scale, stream);
    static const int dbg = knob("RVC_CONV_DEBUG", 0);
    ConvParams p = p_in;
    p.debug = dbg;
    if ((p.c1 % CONV_CH_ALIGN) || (p.c2 % CONV_CH_ALIGN) || p.c1 + p.c2 <= 0)
        return fail("conv: input channels (%d + %d) must be multiples of %d", p.c1, p.c2, CONV_CH_ALIGN);
    if (p.dil < 1 || p.dil > CONV_MAX_DIL) return fail("conv: dilation %d out of range 1..%d", p.dil, CONV_MAX_DIL);
    if (p.n_cols <= 0 || p.batch <= 0) return 0;
    if (p.up_stride == 0 && (int64_t)p.m_total * p.l_out >= ((int64_t)1 << 30))
        return fail("conv: one output slab of %d x %lld floats exceeds the 4 GB the epilogue addresses", p.m_total,
                    (long long)p.l_out);
    switch (p.kw) {
        case 1: return launch_kw<1>(p, stream);
        case 2: return launch_kw<2>(p, stream);
        case 3: return launch_kw<3>(p, stream);
        case 5: return launch_kw<5>(p, stream);
        case 7: return launch_kw<7>(p, stream);
        case 11: return launch_kw<11>(p, stream);
        default: return fail("conv: unsupported kernel size %d (1, 2, 3, 5, 7, 11)", p.kw);
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

extern "C" int rvc_set_concurrency_hint(int utterances_in_flight) {
    if (utterances_in_flight < 1) return fail("rvc_set_concurrency_hint: need at least 1, got %d", utterances_in_flight);
    g_concurrency.store(utterances_in_flight, std::memory_order_relaxed);
    return 0;
}

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
