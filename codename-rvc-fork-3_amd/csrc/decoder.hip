// K2 -- the vocoder ("dec" of Synthesizer, synthesizers.py:254-258) as a chain of launches on one stream:
//   source module (sine + noise excitation)  -> har_source [B, L]
//   conv_pre (+ speaker conditioning folded into a per-utterance bias)
//   per stage: leaky -> ConvTranspose1d (polyphase GEMM) with noise_conv(har_source) folded in as extra
//              input rows (first stage: as its own launch, see Stage::nc_rows) -> three parallel dilated ResBlocks, mean
//   leaky(0.01) -> conv_post -> tanh
// NSF: hifigan_nsf.py:173-207 + hifigan.py:156-228;  MRF: hifigan_mrf.py:339-366, 129-175.
// All dense contractions go through conv.hip / wino.hip (fp32 MFMA; the ResBlock layers in Winograd form); this file holds the small
// element-wise kernels, the weight repacks and the schedule.
#include <stdlib.h>

#include "decoder.h"

namespace rvc {

// ----------------------------------------------------------------------------------------------------------
// small kernels
// ----------------------------------------------------------------------------------------------------------

// hifigan.py:172-177: carry[i] = fmod(cumsum_{i' < i}(fmod(f0[i']/sr*upp + 0.5, 1) - 0.5), 1).
// torch's CPU cumsum accumulates in double and rounds every prefix to float; so does this.  The double prefix needs no particular
// order: every term fmodf(x + 0.5f, 1) - 0.5f with x >= 0 is a multiple of 2^-24 of magnitude <= 0.5 (x + 0.5f >= 0.5 has an ulp
// >= 2^-24; fmodf and the subtraction are exact), so any partial sum of fewer than 2^28 terms is exact in double and a parallel scan
// gives the bits of the sequential sum.  Three steps per chunk of 4096 frames: each thread sums its own 16 consecutive terms,
// thread 0 scans the 256 totals, each thread adds its offset.  (One lane adding all 3198 terms in order took 43 us at the very
// front of the vocoder, with the whole chip waiting; a plain loop over global memory 0.86 ms.)
constexpr int CARRY_CHUNK = 4096;
__global__ void __launch_bounds__(256)
nsf_carry_kernel(const float *__restrict__ f0, int64_t T, float sr, float upp, float *__restrict__ carry) {
    __shared__ double pre[CARRY_CHUNK];
    __shared__ double part[257];
    __shared__ double cum_s;
    constexpr int SEG = CARRY_CHUNK / 256;
    const int64_t b = blockIdx.x;
    const float *f = f0 + b * T;
    float *c = carry + b * T;
    if (threadIdx.x == 0) {
        c[0] = 0.f;
        cum_s = 0.0;
    }
    for (int64_t base = 0; base + 1 < T; base += CARRY_CHUNK) {
        const int n = (int)((T - 1 - base) < CARRY_CHUNK ? (T - 1 - base) : CARRY_CHUNK);
        for (int i = threadIdx.x; i < n; i += 256) {
            const float last = __fmul_rn(__fdiv_rn(f[base + i], sr), upp);
            pre[i] = (double)__fsub_rn(fmodf(__fadd_rn(last, 0.5f), 1.0f), 0.5f);
        }
        __syncthreads();
        const int lo = threadIdx.x * SEG, hi = lo + SEG < n ? lo + SEG : n;
        double run = 0.0;
        for (int i = lo; i < hi; ++i) {
            run += pre[i];
            pre[i] = run;
        }
        part[threadIdx.x + 1] = run;
        __syncthreads();
        if (threadIdx.x == 0) {
            double cum = cum_s;
            part[0] = cum;
            for (int i = 1; i <= 256; ++i) {
                cum += part[i];
                part[i] = cum;
            }
            cum_s = cum;
        }
        __syncthreads();
        const double off = part[threadIdx.x];
        for (int i = lo; i < hi; ++i) pre[i] += off;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) c[base + i + 1] = fmodf((float)pre[i], 1.0f);
        __syncthreads();
    }
}

// hifigan.py:186-226 + hifigan_nsf.py:50-51 for harmonic_num = 0
__global__ void __launch_bounds__(256)
nsf_source_kernel(const float *__restrict__ f0, const float *__restrict__ carry, const float *__restrict__ randn,
                  int64_t T, int upp, float sr, float lin_w, float lin_b, float *__restrict__ har) {
    const int64_t L = T * upp;
    const int64_t b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const int64_t i = t / upp;
    const int j = (int)(t - i * upp);
    const float f = f0[b * T + i];
    const float r = __fdiv_rn(f, sr);
    const float phase = __fadd_rn(__fmul_rn(r, (float)(j + 1)), carry[b * T + i]);
    const float sine = __fmul_rn(sinf(__fmul_rn(6.2831853071795864769f, phase)), 0.1f);
    const float uv = f > 0.f ? 1.f : 0.f;
    const float amp = __fadd_rn(__fmul_rn(uv, 0.003f), __fmul_rn(1.f - uv, (float)(0.1 / 3)));
    const float v = __fadd_rn(__fmul_rn(sine, uv), __fmul_rn(amp, randn[b * L + t]));
    har[b * L + t] = tanhf(__fadd_rn(__fmul_rn(v, lin_w), lin_b));
}

// ---- MRF / RefineGAN sine generator (hifigan_mrf.py:129-175, refinegan.py:220-260) -------------------
// rad = (f0*h/sr) % 1 is constant inside a frame (f0 is nearest-upsampled), so both sample-rate cumsums
// reduce to per-frame prefixes (sequential, double, T elements) plus closed forms inside the frame.
// Only implemented for piecewise-constant f0 (MRF); RefineGAN interpolates f0 linearly (TODO there).

struct MrfSrcParams {
    const float *f0; const float *rand_ini; const float *randn; float *har;
    double *p1; double *p2; int *wraps; float *tot_unused;
    int64_t T; int upp; int dim; float sr;
    float lin_w[MRF_MAX_DIM]; float lin_b;
};

__device__ __forceinline__ float mrf_rad(float f0, int h, float sr) {
    // f0_buf[..., h] = f0 * (h + 1) (fp32), rad = (f0_buf / sr) % 1
    const float fh = (h == 0) ? f0 : __fmul_rn(f0, (float)(h + 1));
    const float q = __fdiv_rn(fh, sr);
    return q - floorf(q);  // torch remainder for a positive divisor
}

// pass 1: P1[b][h][i] = double prefix of rad over all samples before frame i (rand_ini added to sample 0).
// pass 3 (below): P2, the same with the wrap shifts.  Both prefixes are added strictly in frame order, in double, like the
// reference's CPU cumsum -- but only ADDED by the one lane per harmonic: the per-frame terms (a division, a floor, exact products) are
// computed by the whole block into LDS first and the prefixes leave through LDS too, so the serial chain is an LDS read and one
// double add per frame (round 5; with the load, the division and the store inside the chain the two passes took 0.35 + 0.46 ms of
// one wave at the front of the MRF vocoder, the whole chip waiting).
constexpr int MRF_CH = 384;              // frames per LDS chunk
constexpr int MRF_CS = MRF_CH + 1;       // row stride in doubles (the harmonics' rows on different banks)

template <int PASS>
__device__ __forceinline__ double mrf_prefix_term(const MrfSrcParams &p, int b, int h, int64_t i) {
    const float rho = mrf_rad(p.f0[b * p.T + i], h, p.sr);
    if constexpr (PASS == 1) {
        if (i == 0) {
            const float r0 = __fadd_rn(rho, h == 0 ? 0.f : p.rand_ini[b * p.dim + h]);
            return (double)r0 + (double)(p.upp - 1) * (double)rho;
        }
        return (double)p.upp * (double)rho;
    } else {
        const float rho_m1 = __fadd_rn(rho, -1.0f);
        const int w = p.wraps[((int64_t)b * p.dim + h) * p.T + i];
        if (i == 0) {
            const float r0 = __fadd_rn(rho, h == 0 ? 0.f : p.rand_ini[b * p.dim + h]);
            return (double)r0 + (double)(p.upp - 1 - w) * (double)rho + (double)w * (double)rho_m1;
        }
        return (double)(p.upp - w) * (double)rho + (double)w * (double)rho_m1;
    }
}

template <int PASS>
__global__ void __launch_bounds__(256) mrf_prefix_kernel(MrfSrcParams p) {
    __shared__ double term[MRF_MAX_DIM * MRF_CS];
    const int b = blockIdx.x, tid = threadIdx.x;
    double *const P = (PASS == 1 ? p.p1 : p.p2) + (int64_t)b * p.dim * p.T;
    double cum = 0.0;                                     // thread h < dim: the running sum of harmonic h
    for (int64_t base = 0; base < p.T; base += MRF_CH) {
        const int n = (int)(p.T - base < MRF_CH ? p.T - base : MRF_CH);
        for (int idx = tid; idx < p.dim * n; idx += 256) {
            const int h = idx / n, i = idx - h * n;
            term[h * MRF_CS + i] = mrf_prefix_term<PASS>(p, b, h, base + i);
        }
        __syncthreads();
        if (tid < p.dim) {
            double *row = term + tid * MRF_CS;
            int i = 0;
            for (; i + 8 <= n; i += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = row[i + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    row[i + j] = cum;                     // the prefix BEFORE frame i + j
                    cum += v[j];
                }
            }
            for (; i < n; ++i) {
                const double v = row[i];
                row[i] = cum;
                cum += v;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < p.dim * n; idx += 256) {
            const int h = idx / n, i = idx - h * n;
            P[(int64_t)h * p.T + base + i] = term[h * MRF_CS + i];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float mrf_tmp(const MrfSrcParams &p, const double *P1, int b, int h, int64_t i, int j) {
    // tmp_over_one at sample (i, j): float(cumsum) % 1
    const float rho = mrf_rad(p.f0[b * p.T + i], h, p.sr);
    double s = P1[i];
    if (i == 0) {
        const float r0 = __fadd_rn(rho, h == 0 ? 0.f : p.rand_ini[b * p.dim + h]);
        s += (double)r0 + (double)j * (double)rho;
    } else {
        s += (double)(j + 1) * (double)rho;
    }
    const float c = (float)s;
    return c - floorf(c);
}

// pass 2: per frame, count the wrap events (tmp[t] - tmp[t-1] < 0); one wave per (b, h, frame)
__global__ void __launch_bounds__(64) mrf_wraps_kernel(MrfSrcParams p) {
    const int64_t i = blockIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const double *P1 = p.p1 + ((int64_t)b * p.dim + h) * p.T;
    int cnt = 0;
    for (int j = threadIdx.x; j < p.upp; j += 64) {
        if (i == 0 && j == 0) continue;  // cumsum_shift[:, 0] stays 0
        const float cur = mrf_tmp(p, P1, b, h, i, j);
        const float prev = (j == 0) ? mrf_tmp(p, P1, b, h, i - 1, p.upp - 1) : mrf_tmp(p, P1, b, h, i, j - 1);
        if (__fsub_rn(cur, prev) < 0.f) ++cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (threadIdx.x == 0) p.wraps[((int64_t)b * p.dim + h) * p.T + i] = cnt;
}

// pass 3: P2[b][h][i] = double prefix of (rad + shift) over all samples before frame i: mrf_prefix_kernel<3> above

// pass 4: one block per (frame, b): in-frame prefix of the wrap flags, sines, noise, Linear(dim -> 1), tanh
__global__ void __launch_bounds__(256) mrf_source_kernel(MrfSrcParams p) {
    __shared__ int wave_tot[4];
    const int64_t i = blockIdx.x;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t L = p.T * p.upp;
    const float f = p.f0[b * p.T + i];
    const float uv = f > 0.f ? 1.f : 0.f;
    const float amp = __fadd_rn(__fmul_rn(uv, 0.003f), __fdiv_rn(__fmul_rn(1.f - uv, 0.1f), 3.f));
    const int n_it = (p.upp + 255) / 256;
    float out_acc[4] = {0.f, 0.f, 0.f, 0.f};  // upp <= 1024 -> at most 4 slices of 256 samples
    for (int h = 0; h < p.dim; ++h) {
        const double *P1 = p.p1 + ((int64_t)b * p.dim + h) * p.T;
        const double P2 = p.p2[((int64_t)b * p.dim + h) * p.T + i];
        const float rho = mrf_rad(f, h, p.sr);
        const float rho_m1 = __fadd_rn(rho, -1.0f);
        const float r0 = (i == 0) ? __fadd_rn(rho, h == 0 ? 0.f : p.rand_ini[b * p.dim + h]) : rho;
        const float lw = p.lin_w[h];
        int base = 0;  // wraps in the frame before this 256-sample slice
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (it < n_it) {  // block-uniform
                const int j = it * 256 + threadIdx.x;
                int flag = 0;
                if (j < p.upp && !(i == 0 && j == 0)) {
                    const float cur = mrf_tmp(p, P1, b, h, i, j);
                    const float prev = (j == 0) ? mrf_tmp(p, P1, b, h, i - 1, p.upp - 1) : mrf_tmp(p, P1, b, h, i, j - 1);
                    flag = __fsub_rn(cur, prev) < 0.f ? 1 : 0;
                }
                // inclusive prefix count of flags over the block
                int incl = flag;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                __syncthreads();
                if (lane == 63) wave_tot[wave] = incl;
                __syncthreads();
                int before = base;
                for (int w = 0; w < wave; ++w) before += wave_tot[w];
                const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
                const int wcount = before + incl;  // wraps among samples [0..j] of this frame
                if (j < p.upp) {
                    // cumsum(rad + shift) up to and including sample j, in double like torch's CPU cumsum
                    double s = P2;
                    if (i == 0) s += (double)r0 + (double)(j - wcount) * (double)rho + (double)wcount * (double)rho_m1;
                    else s += (double)(j + 1 - wcount) * (double)rho + (double)wcount * (double)rho_m1;
                    const float c = (float)s;
                    const float sine = __fmul_rn(sinf(__fmul_rn(__fmul_rn(c, 2.f), 3.14159265358979323846f)), 0.1f);
                    const int64_t t = i * p.upp + j;
                    const float nz = p.randn[((int64_t)b * L + t) * p.dim + h];
                    const float v = __fadd_rn(__fmul_rn(sine, uv), __fmul_rn(amp, nz));
                    out_acc[it] = fmaf(v, lw, out_acc[it]);
                }
                base += total;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = it * 256 + threadIdx.x;
        if (it < n_it && j < p.upp) p.har[(int64_t)b * L + i * p.upp + j] = tanhf(out_acc[it] + p.lin_b);
    }
}

// V[b][k][q] = har[b][q*S + k - P] (0 outside): the strided views the noise conv reads, as GEMM rows
__global__ void __launch_bounds__(256)
unfold_src_kernel(const float *__restrict__ har, int64_t L, int64_t S, int64_t P, int k_valid, int k_rows, int64_t nq,
                  float *__restrict__ V) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (q >= nq) return;
    float v = 0.f;
    if (k < k_valid) {
        const int64_t idx = q * S + k - P;
        if (idx >= 0 && idx < L) v = har[b * L + idx];
    }
    V[(b * k_rows + k) * nq + q] = v;
}

// bias'[b][co] = conv_pre.bias[co] + cond.bias[co] + cond.weight[co][:] . g[b][:]   (hifigan_nsf.py:179-182)
__global__ void __launch_bounds__(64)
cond_bias_kernel(const float *__restrict__ pre_b, const float *__restrict__ cond_w, const float *__restrict__ cond_b,
                 const float *__restrict__ g, int gin, int c0, float *__restrict__ out) {
    const int co = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < gin; c += 64) s = fmaf(cond_w[(int64_t)co * gin + c], g[(int64_t)b * gin + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[(int64_t)b * c0 + co] = pre_b[co] + (cond_b[co] + s);
}

// out[b][t] = tanh(bias + sum_{ci,k} w[ci][k] * leaky(x[b][ci][t + k - 3], slope))   (hifigan_nsf.py:204-205)
// One lane = 4 consecutive outputs: per input channel three aligned 16-byte loads cover x[t-4 .. t+7] (unconditional: loads inside
// divergent bounds checks are waited for one by one), 28 FMAs; four channels' loads are requested before the first is used.
// (8 outputs per lane -- lanes 32 bytes apart, four half-coalesced loads per channel -- measured 262 us against 125.)  Blocks that
// touch either end of the signal, and lengths that are not a multiple of 4, take the scalar path.  The sums run channel by channel,
// tap by tap, in both paths.
constexpr int POST_OUT = 4;      // outputs per lane
__global__ void __launch_bounds__(256)
conv_post_kernel(const float *__restrict__ x, const float *__restrict__ w, float bias, int c_in, int64_t L, float slope,
                 float *__restrict__ out) {
    const int64_t b = blockIdx.y;
    const int64_t t0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * POST_OUT;
    const int64_t blk0 = (int64_t)blockIdx.x * blockDim.x * POST_OUT;
    const float *xb = x + b * c_in * L;
    const bool fast = (L & 3) == 0 && (c_in & 3) == 0 && blk0 >= 4 && blk0 + (int64_t)blockDim.x * POST_OUT + 4 <= L;   // block-uniform
    if (fast) {
        float acc[POST_OUT];
#pragma unroll
        for (int o = 0; o < POST_OUT; ++o) acc[o] = bias;
        for (int c0 = 0; c0 < c_in; c0 += 4) {
            f32x4 q[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 *xr = reinterpret_cast<const f32x4 *>(xb + (int64_t)(c0 + u) * L + t0 - 4);
#pragma unroll
                for (int j = 0; j < 3; ++j) q[u][j] = xr[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[12];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = q[u][j][e];
#pragma unroll
                for (int i = 1; i < 11; ++i) v[i] = lrelu(v[i], slope);
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float wk = w[(c0 + u) * 7 + k];
#pragma unroll
                    for (int o = 0; o < POST_OUT; ++o) acc[o] = fmaf(wk, v[o + k + 1], acc[o]);
                }
            }
        }
        *reinterpret_cast<f32x4 *>(out + b * L + t0) = f32x4{tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3])};
        return;
    }
    // edge blocks / odd lengths: the same sums from clamped, unconditional scalar loads (a load behind a per-element bounds check is
    // waited for on its own: 896 serial round trips per lane made these two blocks most of the launch's duration)
    if (t0 >= L) return;
    float acc[POST_OUT];
#pragma unroll
    for (int o = 0; o < POST_OUT; ++o) acc[o] = bias;
    for (int ci = 0; ci < c_in; ++ci) {
        const float *xr = xb + (int64_t)ci * L;
        float v[POST_OUT + 6];
#pragma unroll
        for (int i = 0; i < POST_OUT + 6; ++i) {
            const int64_t tt = t0 + i - 3;
            const int64_t tc = tt < 0 ? 0 : (tt >= L ? L - 1 : tt);
            const float raw = xr[tc];
            v[i] = (tt >= 0 && tt < L) ? lrelu(raw, slope) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float wk = w[ci * 7 + k];
#pragma unroll
            for (int o = 0; o < POST_OUT; ++o) acc[o] = fmaf(wk, v[o + k], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < POST_OUT; ++o)
        if (t0 + o < L) out[b * L + t0 + o] = tanhf(acc[o]);
}

int launch_unfold_src(const float *har, int batch, int64_t L, int64_t S, int64_t P, int k_valid, int k_rows, int64_t nq, float *V,
                      hipStream_t stream) {
    hipLaunchKernelGGL(unfold_src_kernel, dim3((unsigned)ceil_div(nq, 256), k_rows, batch), dim3(256), 0, stream, har, L, S, P,
                       k_valid, k_rows, nq, V);
    RVC_LAUNCH_CHECK();
    return 0;
}

int launch_cond_bias(const float *pre_b, const float *cond_w, const float *cond_b, const float *g, int batch, int gin, int c0,
                     float *out, hipStream_t stream) {
    hipLaunchKernelGGL(cond_bias_kernel, dim3(c0, batch), dim3(64), 0, stream, pre_b, cond_w, cond_b, g, gin, c0, out);
    RVC_LAUNCH_CHECK();
    return 0;
}

int launch_conv_post(const float *x, const float *w, float bias, int batch, int c_in, int64_t L, float slope, float *out,
                     hipStream_t stream) {
    hipLaunchKernelGGL(conv_post_kernel, dim3((unsigned)ceil_div(L, 256 * POST_OUT), batch), dim3(256), 0, stream, x, w, bias, c_in, L, slope,
                       out);
    RVC_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------

}  // namespace rvc

using namespace rvc;

namespace rvc {
const HostTensor *find(const rvc_decoder *d, const std::string &name) {
    auto it = d->host.find(name);
    return it == d->host.end() ? nullptr : &it->second;
}

int need(const rvc_decoder *d, const std::string &name, const HostTensor **out, std::vector<int64_t> shape) {
    const HostTensor *t = find(d, name);
    if (!t) return fail("decoder: tensor '%s' was never set", name.c_str());
    if (t->shape != shape) {
        std::string got, want;
        for (auto s : t->shape) got += std::to_string(s) + ",";
        for (auto s : shape) want += std::to_string(s) + ",";
        return fail("decoder: tensor '%s' has shape [%s], expected [%s]", name.c_str(), got.c_str(), want.c_str());
    }
    *out = t;
    return 0;
}

int build_conv(const rvc_decoder *d, const std::string &prefix, int c_out, int c_in, int k, bool bias, ConvW *out, bool as_bf16) {
    const HostTensor *w, *b;
    if (need(d, prefix + ".weight", &w, {c_out, c_in, k})) return 1;
    std::vector<float> packed((size_t)c_out * c_in * k);
    for (int co = 0; co < c_out; ++co)
        for (int ci = 0; ci < c_in; ++ci)
            for (int t = 0; t < k; ++t)
                packed[((size_t)t * c_in + ci) * c_out + co] = w->data[((size_t)co * c_in + ci) * k + t];
    if (as_bf16) {   // HBM copy in bf16 (round to nearest even; exact when the caller hands over bf16-valued weights)
        std::vector<uint16_t> half(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) half[i] = bf16_rne(packed[i]);
        if (out->w16.upload(half)) return 1;
    } else if (out->w.upload(packed)) {
        return 1;
    }
    if (c_in == c_out && c_out % 32 == 0 && wino_supported(k, 1)) {   // ResBlock layers: second copy for the fast form
        if (as_bf16) {
            std::vector<uint32_t> words;
            wino_pack_host_bf16(w->data.data(), c_out, c_in, k, &words);
            std::vector<uint16_t> halves(words.size() * 2);
            memcpy(halves.data(), words.data(), words.size() * 4);
            if (out->wu16.upload(halves)) return 1;
            if (convbf1_supported(c_out, k, 1) && convbf1_preferred(c_out, k)) {   // direct form with one-term taps (K3d): taken before the Winograd form
                std::vector<uint16_t> frags;
                convbf1_pack_host(w->data.data(), c_out, k, &frags);
                if (out->wd.upload(frags)) return 1;
            }
            if (winobf_enabled() && winobf_supported(c_in, c_out, k, 1)) {   // the bf16-matrix-core form on the bf16-VALUED taps: a transformed
                std::vector<float> vals((size_t)c_out * c_in * k);           // tap is a sum of four of them and needs its three-way split like any other
                for (size_t i = 0; i < vals.size(); ++i) {
                    const uint32_t bits = (uint32_t)bf16_rne(w->data[i]) << 16;
                    memcpy(&vals[i], &bits, 4);
                }
                std::vector<uint16_t> frags;
                winobf_pack_host(vals.data(), c_out, c_in, k, &frags);
                if (out->wx.upload(frags)) return 1;
            }
        } else {
            wino_pack_host(w->data.data(), c_out, c_in, k, &packed);
            if (out->wu.upload(packed)) return 1;
            if (winobf_enabled() && winobf_supported(c_in, c_out, k, 1)) {   // third copy: the bf16-matrix-core form's fragments
                std::vector<uint16_t> frags;
                winobf_pack_host(w->data.data(), c_out, c_in, k, &frags);
                if (out->wx.upload(frags)) return 1;
            }
        }
    }
    // 7- / 11-tap layers that change the channel count (conv_pre, RefineGAN's mel_conv / input_conv): the bf16-matrix-core
    // Winograd form takes any (c_in % 16, c_out % 64) pair; the fp32 Winograd kernel is square-only, so these only carry
    // the direct-form slab and the bf16x3 fragments
    if (c_in != c_out && !as_bf16 && k >= 7 && winobf_enabled() && winobf_supported(c_in, c_out, k, 1)) {
        std::vector<uint16_t> frags;
        winobf_pack_host(w->data.data(), c_out, c_in, k, &frags);
        if (out->wx.upload(frags)) return 1;
    }
    if (bias) {
        if (need(d, prefix + ".bias", &b, {c_out})) return 1;
        if (out->b.upload(b->data)) return 1;
    }
    out->c_in = c_in; out->c_out = c_out; out->k = k;
    return 0;
}
}  // namespace rvc

extern "C" int rvc_decoder_create(const rvc_decoder_config *cfg, rvc_decoder **out) {
    if (!cfg || !out) return fail("rvc_decoder_create: null pointer");
    if (cfg->kind != RVC_DEC_NSF && cfg->kind != RVC_DEC_MRF && cfg->kind != RVC_DEC_REFINE)
        return fail("rvc_decoder_create: unknown decoder kind %d", cfg->kind);
    if (cfg->n_ups < 1 || cfg->n_ups > 8) return fail("rvc_decoder_create: n_ups out of range");
    if (cfg->n_res_kernels < 1 || cfg->n_res_kernels > 4 || cfg->n_res_dilations < 1 || cfg->n_res_dilations > 4)
        return fail("rvc_decoder_create: resblock configuration out of range");
    if (cfg->weight_storage != 0 && cfg->weight_storage != 1) return fail("rvc_decoder_create: weight_storage must be 0 (fp32) or 1 (bf16)");
    if (cfg->weight_storage == 1 && cfg->kind == RVC_DEC_REFINE)
        return fail("rvc_decoder_create: bf16 weight storage is built for the NSF / MRF decoders (BASELINE cfg 4), not RefineGAN");
    if (cfg->in_channels % 8 || cfg->upsample_initial_channel % (32 << cfg->n_ups))
        return fail("rvc_decoder_create: channel counts must keep every stage a multiple of 32");
    rvc_decoder *d = new rvc_decoder();
    d->cfg = *cfg;
    d->upp = 1;
    for (int i = 0; i < cfg->n_ups; ++i) d->upp *= cfg->upsample_rates[i];
    if (d->upp > 1024) { delete d; return fail("rvc_decoder_create: prod(upsample_rates) > 1024"); }
    d->dim = cfg->kind == RVC_DEC_MRF ? 9 : 1;
    *out = d;
    return 0;
}

extern "C" int rvc_decoder_set_tensor(rvc_decoder *dec, const char *name, const float *data_host, const int64_t *shape,
                                      int ndim) {
    if (!dec || !name || !data_host || !shape || ndim < 1 || ndim > 4) return fail("rvc_decoder_set_tensor: bad argument");
    if (dec->finalized) return fail("rvc_decoder_set_tensor: decoder already finalized");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    const int64_t n = t.numel();
    if (n <= 0) return fail("rvc_decoder_set_tensor: empty tensor '%s'", name);
    t.data.assign(data_host, data_host + n);
    dec->host[name] = std::move(t);
    return 0;
}

extern "C" int rvc_decoder_finalize(rvc_decoder *d) {
    if (!d) return fail("rvc_decoder_finalize: null decoder");
    if (d->finalized) return 0;
    if (d->cfg.kind == RVC_DEC_REFINE) {
        if (refine_finalize(d)) return 1;
        d->host.clear();
        d->finalized = true;
        return 0;
    }
    const rvc_decoder_config &c = d->cfg;
    const bool mrf = c.kind == RVC_DEC_MRF;
    const HostTensor *t;
    // source merge: Linear(dim -> 1) + tanh
    if (need(d, "m_source.l_linear.weight", &t, {1, d->dim})) return 1;
    for (int h = 0; h < d->dim; ++h) d->lin_w[h] = t->data[h];
    if (need(d, "m_source.l_linear.bias", &t, {1})) return 1;
    d->lin_b = t->data[0];
    // conv_pre + cond
    const int c0 = c.upsample_initial_channel;
    if (build_conv(d, "conv_pre", c0, c.in_channels, 7, true, &d->pre)) return 1;
    if (need(d, "cond.weight", &t, {c0, c.gin_channels, 1})) return 1;
    if (d->cond_w.upload(t->data)) return 1;
    if (need(d, "cond.bias", &t, {c0})) return 1;
    if (d->cond_b.upload(t->data)) return 1;
    // stages
    d->stages.clear();
    d->stages.resize(c.n_ups);
    for (int i = 0; i < c.n_ups; ++i) {
        Stage &s = d->stages[i];
        s.c_in = c0 >> i;
        s.c_out = c0 >> (i + 1);
        s.rate = c.upsample_rates[i];
        s.ksize = c.upsample_kernel_sizes[i];
        // hifigan_nsf.py:113-117,127
        s.pad = (s.rate % 2 == 0) ? (s.ksize - s.rate) / 2 : s.rate / 2 + s.rate % 2;
        s.opad = s.rate % 2;
        s.taps = (s.ksize + s.rate - 1) / s.rate;
        if (s.taps > 2) return fail("decoder: upsample kernel %d with rate %d needs %d polyphase taps (max 2)", s.ksize, s.rate, s.taps);
        if (s.taps < 2) s.taps = 2;  // the GEMM kernel is instantiated for 2 taps; the second is zero
        int stride_f0 = 1;
        for (int j = i + 1; j < c.n_ups; ++j) stride_f0 *= c.upsample_rates[j];
        s.nc_stride = stride_f0;
        // hifigan_nsf.py:142-144
        s.nc_k = stride_f0 == 1 ? 1 : stride_f0 * 2 - stride_f0 % 2;
        s.nc_pad = stride_f0 == 1 ? 0 : (s.nc_k - stride_f0) / 2;
        s.vk = (s.rate - 1) * s.nc_stride + s.nc_k;
        s.vk_rows = (s.vk + 7) / 8 * 8;
        s.S = (int64_t)s.rate * s.nc_stride;
        s.P = (int64_t)s.pad * s.nc_stride + s.nc_pad;
        static const int nc_sep_env = knob("RVC_NC_SEPARATE", 1);
        const bool nc_separate = nc_sep_env && s.vk_rows * 2 >= s.c_in;   // folding would at least 1.5x the upsampler's GEMM
        if (nc_separate) {
            s.nc_rows = (s.nc_k + 7) / 8 * 8;
            s.vk = 0;
            s.vk_rows = 0;
        }

        const std::string up = (mrf ? "upsamples." : "ups.") + std::to_string(i);
        const std::string nc = "noise_convs." + std::to_string(i);
        const HostTensor *uw, *ub, *nw, *nb;
        if (need(d, up + ".weight", &uw, {s.c_in, s.c_out, s.ksize})) return 1;
        if (need(d, up + ".bias", &ub, {s.c_out})) return 1;
        if (need(d, nc + ".weight", &nw, {s.c_out, 1, s.nc_k})) return 1;
        if (need(d, nc + ".bias", &nb, {s.c_out})) return 1;
        // polyphase repack.  Output t = q*rate + phase - pad gets sum_j sum_ci W[ci][co][phase + j*rate] x[ci][q - j];
        // as GEMM taps with dil = 1, padl = taps-1: tap' = taps-1-j.  Noise rows only on the offset-0 tap.
        const int ctot = s.c_in + s.vk_rows;
        const int m_total = s.rate * s.c_out;
        // GEMM row of (phase, channel): phase-major, or channel-major for even rates (ConvParams::up_interleave: 8- / 16-byte output stores)
        s.interleave = s.rate % 2 == 0;
        auto mrow = [&](int ph, int co) { return s.interleave ? co * s.rate + ph : ph * s.c_out + co; };
        std::vector<float> packed((size_t)s.taps * ctot * m_total, 0.f);
        for (int tp = 0; tp < s.taps; ++tp) {
            const int j = s.taps - 1 - tp;
            for (int ci = 0; ci < s.c_in; ++ci)
                for (int ph = 0; ph < s.rate; ++ph) {
                    const int kk = ph + j * s.rate;
                    if (kk >= s.ksize) continue;
                    for (int co = 0; co < s.c_out; ++co)
                        packed[((size_t)tp * ctot + ci) * m_total + mrow(ph, co)] =
                            uw->data[((size_t)ci * s.c_out + co) * s.ksize + kk];
                }
        }
        const int tp0 = s.taps - 1;  // the tap with input offset 0
        for (int kq = 0; kq < s.vk; ++kq)
            for (int ph = 0; ph < s.rate; ++ph) {
                const int k = kq - ph * s.nc_stride;
                if (k < 0 || k >= s.nc_k) continue;
                for (int co = 0; co < s.c_out; ++co)
                    packed[((size_t)tp0 * ctot + s.c_in + kq) * m_total + mrow(ph, co)] = nw->data[(size_t)co * s.nc_k + k];
            }
        if (s.w.upload(packed)) return 1;
        std::vector<float> bias(s.c_out);
        for (int co = 0; co < s.c_out; ++co) bias[co] = ub->data[co] + nb->data[co];
        {   // K3u: the same GEMM on the bf16 matrix cores (exact bf16x3 operands).  The stage's noise conv is folded into the GEMM straight
            // from har_source, with the bias as a row of ones -- except where decoder.hip already runs it as its own launch (nc_rows).
            // Tiles of 64 GEMM rows (the last stage of the 48 k vocoder) stay on the fp32 kernel: 300 against 191 us (stager-bound).
            static const int ups_bf = knob("RVC_UPS_BF", 1);
            const int k_n = s.nc_rows > 0 ? 0 : s.nc_k;
            const int vk = upsbf_fold_noise(k_n) ? (s.rate - 1) * s.nc_stride + k_n : 0;
            if (ups_bf && s.opad == 0 && s.rate * s.c_out >= 128 && upsbf_supported(s.c_in, s.c_out, s.rate, s.ksize, vk)) {
                std::vector<uint16_t> frags;
                upsbf_pack_host(uw->data.data(), k_n ? nw->data.data() : nullptr, bias.data(), s.c_in, s.c_out, s.rate, s.ksize, vk, k_n, s.nc_stride, &frags);
                if (s.wub.upload(frags)) return 1;
                s.ub_vk = vk;
            }
        }
        if (s.nc_rows) {
            std::vector<float> ncw((size_t)s.nc_rows * s.c_out, 0.f);
            for (int k = 0; k < s.nc_k; ++k)
                for (int co = 0; co < s.c_out; ++co) ncw[(size_t)k * s.c_out + co] = nw->data[(size_t)co * s.nc_k + k];
            if (s.nc_w.upload(ncw)) return 1;
        }
        if (s.b.upload(bias)) return 1;

        const int nb_ = c.n_res_kernels * c.n_res_dilations;
        s.c1 = std::vector<ConvW>(nb_);
        s.c2 = std::vector<ConvW>(nb_);
        s.pair = std::vector<DevBuf16>(nb_);
        for (int m = 0; m < c.n_res_kernels; ++m)
            for (int j = 0; j < c.n_res_dilations; ++j) {
                std::string p1, p2;
                if (mrf) {
                    const std::string base = "mrfs." + std::to_string(i) + "." + std::to_string(m) + ".layers." + std::to_string(j);
                    p1 = base + ".conv1";
                    p2 = base + ".conv2";
                } else {
                    const std::string base = "resblocks." + std::to_string(i * c.n_res_kernels + m);
                    p1 = base + ".convs1." + std::to_string(j);
                    p2 = base + ".convs2." + std::to_string(j);
                }
                const int k = c.res_kernel_sizes[m];
                const bool b16 = c.weight_storage == 1 && (k == 3 || k == 7 || k == 11);
                if (build_conv(d, p1, s.c_out, s.c_out, k, true, &s.c1[m * c.n_res_dilations + j], b16)) return 1;
                if (build_conv(d, p2, s.c_out, s.c_out, k, true, &s.c2[m * c.n_res_dilations + j], b16)) return 1;
                const int nts = b16 ? 1 : 3;   // bf16 weight storage: the taps ARE their first split -- one-term fragments, three products per multiply-add
                if (resblock_bf_enabled() && resblock_bf_supported(s.c_out, k, 1) && resblock_bf_preferred(s.c_out, k, nts)) {   // the fused pair on the bf16 matrix cores (K3f)
                    const HostTensor *w1, *w2;
                    if (need(d, p1 + ".weight", &w1, {s.c_out, s.c_out, k}) || need(d, p2 + ".weight", &w2, {s.c_out, s.c_out, k})) return 1;
                    std::vector<uint16_t> frags;
                    resblock_bf_pack_host(w1->data.data(), w2->data.data(), s.c_out, k, &frags, nts);   // (nts = 1 rounds to bf16: what build_conv(b16) stores)
                    s.pair_splits = nts;
                    if (s.pair[m * c.n_res_dilations + j].upload(frags)) return 1;
                }
            }
    }
    // conv_post: [1][c_last][7]
    d->post_cin = c0 >> c.n_ups;
    if (need(d, "conv_post.weight", &t, {1, d->post_cin, 7})) return 1;
    if (d->post_w.upload(t->data)) return 1;
    d->post_b = 0.f;
    if (mrf) {
        if (need(d, "conv_post.bias", &t, {1})) return 1;
        d->post_b = t->data[0];
    }
    d->host.clear();
    d->finalized = true;
    return 0;
}

extern "C" int rvc_decoder_destroy(rvc_decoder *dec) {
    delete dec;
    return 0;
}

extern "C" int rvc_decoder_upp(const rvc_decoder *dec) { return dec ? dec->upp : 0; }

namespace {
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; }
};
struct Layout {
    size_t har, carry, biasp, V, p1, p2, wraps, buf[2 + 2 * (MAX_BRANCH_LANES + 1)], total;
    int n_buf;
    int64_t max_cl;
};
// a stage at most this long (batch x channels x samples) runs its ResBlock branches side by side: 3 rounds of K3y's 128 x 256 blocks
constexpr int64_t BRANCH_PARALLEL_MAX_ELEMS = (int64_t)3 * 256 * 128 * 256;
// branches on side streams: every lane owns its two ping-pong buffers
int branch_lanes_wanted(const rvc_decoder *d) {
    const int nk = d->cfg.n_res_kernels, want = d->branch_parallel.load(std::memory_order_relaxed);
    const int most = nk - 1 < MAX_BRANCH_LANES ? nk - 1 : MAX_BRANCH_LANES;
    return want < 0 || want > most ? most : want;
}
Layout make_layout(const rvc_decoder *d, int batch, int64_t T) {
    Layout l;
    Carve c;
    const int64_t L = T * d->upp;
    l.har = c.take((size_t)batch * L * 4);
    l.carry = c.take((size_t)batch * T * 4);
    l.biasp = c.take((size_t)batch * d->cfg.upsample_initial_channel * 4);
    int64_t len = T, max_cl = (int64_t)d->cfg.upsample_initial_channel * T, max_v = 0;
    for (const Stage &s : d->stages) {
        const int64_t nq = len + 1;
        max_v = std::max<int64_t>(max_v, (int64_t)s.vk_rows * nq);
        len = (len - 1) * s.rate - 2 * s.pad + s.ksize + s.opad;
        max_v = std::max<int64_t>(max_v, (int64_t)s.nc_rows * len);
        max_cl = std::max<int64_t>(max_cl, (int64_t)s.c_out * len);
    }
    l.V = c.take((size_t)batch * max_v * 4);
    const size_t scan = d->dim > 1 ? (size_t)batch * d->dim * T : 0;
    l.p1 = c.take(scan * 8);
    l.p2 = c.take(scan * 8);
    l.wraps = c.take(scan * 4);
    l.n_buf = 2 + 2 * (branch_lanes_wanted(d) + 1);
    for (int i = 0; i < l.n_buf; ++i) l.buf[i] = c.take((size_t)batch * max_cl * 4);
    l.total = c.off;
    l.max_cl = max_cl;
    return l;
}
}  // namespace

extern "C" int rvc_decoder_workspace_bytes(const rvc_decoder *dec, int batch, int64_t n_frames, size_t *bytes) {
    if (!dec || !bytes || batch <= 0 || n_frames <= 0) return fail("rvc_decoder_workspace_bytes: bad argument");
    if (!dec->finalized) return fail("rvc_decoder_workspace_bytes: decoder not finalized");
    *bytes = dec->cfg.kind == RVC_DEC_REFINE ? refine_workspace_bytes(dec, batch, n_frames) : make_layout(dec, batch, n_frames).total;
    return 0;
}

extern "C" int rvc_decoder_set_tap(rvc_decoder *dec, int stage, float *tap_dev) {
    if (!dec) return fail("rvc_decoder_set_tap: null decoder");
    dec->tap_stage = tap_dev ? stage : -2;
    dec->tap_dev = tap_dev;
    return 0;
}

int rvc::BranchLanes::create(int n) {
    n_side = n;
    for (int i = 0; i < n; ++i) RVC_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    RVC_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i <= MAX_BRANCH_LANES; ++i) RVC_HIP(hipEventCreateWithFlags(&last[i], hipEventDisableTiming));   // one per BRANCH
    return 0;
}
rvc::BranchLanes::~BranchLanes() {
    for (int i = 0; i < n_side; ++i) if (side[i]) { (void)hipStreamSynchronize(side[i]); (void)hipStreamDestroy(side[i]); }
    if (fork) (void)hipEventDestroy(fork);
    for (int i = 0; i <= MAX_BRANCH_LANES; ++i) if (last[i]) (void)hipEventDestroy(last[i]);
}

namespace {
// the caller stream's lanes (created on first use, kept for the life of the handle)
int lanes_for(rvc_decoder *d, hipStream_t stream, int n_side, BranchLanes **out) {
    std::lock_guard<std::mutex> g(d->lanes_mu);
    BranchLanes *&l = d->lanes[stream];
    if (l && l->n_side != n_side) { delete l; l = nullptr; }
    if (!l) {
        l = new BranchLanes();
        if (l->create(n_side)) { delete l; l = nullptr; return 1; }
    }
    *out = l;
    return 0;
}
}  // namespace

extern "C" int rvc_decoder_set_branch_parallel(rvc_decoder *dec, int side_streams) {
    if (!dec) return fail("rvc_decoder_set_branch_parallel: null decoder");
    dec->branch_parallel.store(side_streams, std::memory_order_relaxed);
    return 0;
}

extern "C" int rvc_decoder_set_concurrency_hint(rvc_decoder *dec, int utterances_in_flight) {
    if (!dec) return fail("rvc_decoder_set_concurrency_hint: null decoder");
    if (utterances_in_flight < 0) return fail("rvc_decoder_set_concurrency_hint: negative hint %d", utterances_in_flight);
    dec->concurrency.store(utterances_in_flight, std::memory_order_relaxed);
    return 0;
}

extern "C" int rvc_decoder_forward(rvc_decoder *d, const float *z_dev, const float *f0_dev, const float *g_dev,
                                   const rvc_decoder_noise *noise, int batch, int64_t T, float *out_dev,
                                   void *workspace_dev, size_t workspace_bytes, void *stream_) {
    if (!d || !z_dev || !f0_dev || !g_dev || !noise || !out_dev || !workspace_dev) return fail("rvc_decoder_forward: null pointer");
    ConcurrencyScope scope(d->concurrency.load(std::memory_order_relaxed));
    if (!d->finalized) return fail("rvc_decoder_forward: decoder not finalized");
    if (batch <= 0 || T <= 0) return fail("rvc_decoder_forward: empty batch");
    if (!noise->src_randn_dev) return fail("rvc_decoder_forward: src_randn_dev is required");
    if (d->cfg.kind == RVC_DEC_REFINE)
        return refine_forward(d, z_dev, f0_dev, g_dev, noise, batch, T, out_dev, workspace_dev, workspace_bytes, (hipStream_t)stream_);
    const rvc_decoder_config &c = d->cfg;
    const bool mrf = c.kind == RVC_DEC_MRF;
    if (mrf && !noise->src_rand_dev) return fail("rvc_decoder_forward: src_rand_dev is required for the MRF decoder");
    hipStream_t stream = (hipStream_t)stream_;
    const Layout lay = make_layout(d, batch, T);
    if (workspace_bytes < lay.total) return fail("rvc_decoder_forward: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
    char *ws = (char *)workspace_dev;
    float *har = (float *)(ws + lay.har);
    float *carry = (float *)(ws + lay.carry);
    float *biasp = (float *)(ws + lay.biasp);
    float *V = (float *)(ws + lay.V);
    float *buf[2 + 2 * (MAX_BRANCH_LANES + 1)];
    for (int i = 0; i < lay.n_buf; ++i) buf[i] = (float *)(ws + lay.buf[i]);
    const int n_side = (lay.n_buf - 4) / 2;
    BranchLanes *lanes = nullptr;
    if (n_side > 0 && lanes_for(d, stream, n_side, &lanes)) return 1;
    const int64_t L = T * d->upp;
    const float sr = (float)c.sample_rate;

    // ---- source module ----
    if (!mrf) {
        hipLaunchKernelGGL(nsf_carry_kernel, dim3(batch), dim3(256), 0, stream, f0_dev, T, sr, (float)d->upp, carry);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(nsf_source_kernel, dim3((unsigned)ceil_div(L, 256), batch), dim3(256), 0, stream, f0_dev, carry,
                           noise->src_randn_dev, T, d->upp, sr, d->lin_w[0], d->lin_b, har);
        RVC_LAUNCH_CHECK();
    } else {
        MrfSrcParams sp;
        sp.f0 = f0_dev; sp.rand_ini = noise->src_rand_dev; sp.randn = noise->src_randn_dev; sp.har = har;
        sp.p1 = (double *)(ws + lay.p1); sp.p2 = (double *)(ws + lay.p2); sp.wraps = (int *)(ws + lay.wraps);
        sp.tot_unused = nullptr;
        sp.T = T; sp.upp = d->upp; sp.dim = d->dim; sp.sr = sr;
        for (int h = 0; h < MRF_MAX_DIM; ++h) sp.lin_w[h] = d->lin_w[h];
        sp.lin_b = d->lin_b;
        hipLaunchKernelGGL(mrf_prefix_kernel<1>, dim3(batch), dim3(256), 0, stream, sp);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(mrf_wraps_kernel, dim3((unsigned)T, d->dim, batch), dim3(64), 0, stream, sp);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(mrf_prefix_kernel<3>, dim3(batch), dim3(256), 0, stream, sp);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(mrf_source_kernel, dim3((unsigned)T, batch), dim3(256), 0, stream, sp);
        RVC_LAUNCH_CHECK();
    }
    if (d->tap_stage == -1 && d->tap_dev)
        RVC_HIP(hipMemcpyAsync(d->tap_dev, har, (size_t)batch * L * 4, hipMemcpyDeviceToDevice, stream));

    // ---- conv_pre + cond ----
    const int c0 = c.upsample_initial_channel;
    hipLaunchKernelGGL(cond_bias_kernel, dim3(c0, batch), dim3(64), 0, stream, d->pre.b.p, d->cond_w.p, d->cond_b.p, g_dev,
                       c.gin_channels, c0, biasp);
    RVC_LAUNCH_CHECK();
    float *cur = buf[0];   // stage input / running sum
    {
        ConvParams p;
        p.x1 = z_dev; p.c1 = c.in_channels; p.slope1 = 1.f; p.x1_bstride = (int64_t)c.in_channels * T; p.l_in = T;
        p.w = d->pre.w.p; p.bias = biasp; p.bias_bstride = c0;
        p.y = cur; p.y_bstride = (int64_t)c0 * T; p.m_total = c0; p.c_out = c0; p.n_cols = T; p.l_out = T;
        p.kw = 7; p.dil = 1; p.padl = 3; p.batch = batch;
        if (launch_conv(p, stream)) return 1;
    }

    // ---- upsample stages ----
    int64_t len = T;
    const int nd = c.n_res_dilations, nk = c.n_res_kernels;
    for (int i = 0; i < c.n_ups; ++i) {
        Stage &s = d->stages[i];
        const int64_t nq = len + 1;
        const int64_t len_out = (len - 1) * s.rate - 2 * s.pad + s.ksize + s.opad;
        float *X = buf[1];
        if (s.wub.p && (int64_t)s.c_in * len * 4 < ((int64_t)1 << 31) && (int64_t)s.c_out * len_out * 4 < ((int64_t)1 << 31) && L * 4 < ((int64_t)1 << 31)) {
            if (launch_upsbf(cur, har, L, s.wub.p, s.b.p, X, batch, s.c_in, s.c_out, len, len_out, s.rate, s.ksize, s.pad, s.ub_vk, s.S, s.P, 0.1f, stream))
                return 1;
        } else {
        if (s.vk_rows) {
            hipLaunchKernelGGL(unfold_src_kernel, dim3((unsigned)ceil_div(nq, 256), s.vk_rows, batch), dim3(256), 0, stream, har, L,
                               s.S, s.P, s.vk, s.vk_rows, nq, V);
            RVC_LAUNCH_CHECK();
        }
        {
            ConvParams p;
            p.x1 = cur; p.c1 = s.c_in; p.slope1 = 0.1f; p.x1_bstride = (int64_t)s.c_in * len;
            if (s.vk_rows) { p.x2 = V; p.c2 = s.vk_rows; p.slope2 = 1.f; p.x2_bstride = (int64_t)s.vk_rows * nq; }
            // both sources are indexed by the GEMM column q; x rows are valid on [0, len), V rows on [0, nq):
            // V is stored with row length nq, x with row length len -> give x its own staging length
            p.l_in = len;
            p.w = s.w.p; p.bias = s.b.p;
            p.y = X; p.y_bstride = (int64_t)s.c_out * len_out;
            p.m_total = s.rate * s.c_out; p.c_out = s.c_out; p.n_cols = nq; p.l_out = len_out;
            p.kw = s.taps; p.dil = 1; p.padl = s.taps - 1;
            p.up_stride = s.rate; p.up_pad = s.pad; p.up_interleave = s.interleave ? 1 : 0; p.batch = batch;
            p.l_in2 = nq;
            if (launch_conv(p, stream)) return 1;
        }
        }
        if (s.nc_rows) {   // the noise conv on its own: y += W_nc V2 (hifigan_nsf.py:190-191), bias already in the upsampler's
            hipLaunchKernelGGL(unfold_src_kernel, dim3((unsigned)ceil_div(len_out, 256), s.nc_rows, batch), dim3(256), 0, stream, har,
                               L, (int64_t)s.nc_stride, (int64_t)s.nc_pad, s.nc_k, s.nc_rows, len_out, V);
            RVC_LAUNCH_CHECK();
            ConvParams p;
            p.x1 = V; p.c1 = s.nc_rows; p.slope1 = 1.f; p.x1_bstride = (int64_t)s.nc_rows * len_out; p.l_in = len_out;
            p.w = s.nc_w.p; p.accin = X; p.y = X; p.y_bstride = (int64_t)s.c_out * len_out;
            p.m_total = s.c_out; p.c_out = s.c_out; p.n_cols = len_out; p.l_out = len_out;
            p.kw = 1; p.dil = 1; p.padl = 0; p.batch = batch;
            if (launch_conv(p, stream)) return 1;
        }
        len = len_out;
        const int64_t bs = (int64_t)s.c_out * len;
        // The branches all read X; the last launch of a branch adds into the running sum `cur` after its predecessor's has.
        // A SHORT stage (its whole-CU kernels are a few rounds of the 256 CUs: 300 blocks = 1.17 rounds in stage 0 of a 30 s clip at
        // 48 k, where each launch leaves most of the chip idle through its second round) can run its branches next to each other
        // on the handle's side streams (rvc_decoder_set_branch_parallel; off by default: profiles/r05_branch_streams.txt).
        const bool short_stage = (int64_t)batch * bs <= BRANCH_PARALLEL_MAX_ELEMS;
        const bool par = lanes && short_stage;
        if (par) {
            RVC_HIP(hipEventRecord(lanes->fork, stream));
            for (int q = 0; q < n_side; ++q) RVC_HIP(hipStreamWaitEvent(lanes->side[q], lanes->fork, 0));
        }
        const hipStream_t main_stream = stream;
        for (int r = 0; r < nk; ++r) {
            const int m = r;
            const bool first_branch = r == 0, final_branch = r + 1 == nk;
            const int k = c.res_kernel_sizes[m];
            const float *xin = X;
            // with fewer side streams than branches the first (short) branches share the caller's stream
            const int off = nk - 1 - n_side;
            const int lane = par && r > off ? r - off : 0, lane_prev = par && r - 1 > off ? r - 1 - off : 0;
            const hipStream_t stream = lane > 0 ? lanes->side[lane - 1] : main_stream;
            float *const Y = buf[2 + 2 * lane], *const T1 = buf[3 + 2 * lane];
            auto before_last = [&]() -> int {
                if (par && r > 0 && lane_prev != lane) RVC_HIP(hipStreamWaitEvent(stream, lanes->last[r - 1], 0));
                return 0;
            };
            auto after_last = [&]() -> int {
                if (par) RVC_HIP(hipEventRecord(lanes->last[r], stream));
                return 0;
            };
            if (s.pair[m * nd].p && resblock_bf_fits(s.c_out, len)) {
                // 32- / 64-channel stages: one launch per (dilated conv, conv) pair on the bf16 matrix cores (resblock_bf.hip); blocks
                // read their neighbours' columns, so the outputs ping-pong between Y and T1
                for (int j = 0; j < nd; ++j) {
                    const bool last = j + 1 == nd;
                    float *yout = last ? cur : (j % 2 == 0 ? Y : T1);
                    const float *acc_in = (last && !first_branch) ? cur : nullptr;
                    const float scale = (last && final_branch) ? 1.f / (float)nk : 1.f;
                    if (last && before_last()) return 1;
                    if (launch_resblock_bf(xin, s.pair[m * nd + j].p, s.c1[m * nd + j].b.p, s.c2[m * nd + j].b.p, acc_in, yout, batch,
                                           s.c_out, len, k, c.res_dilations[j], 0.1f, scale, stream, s.pair_splits))
                        return 1;
                    xin = yout;
                }
                if (after_last()) return 1;
                continue;
            }
            if (resblock_layer_supported(s.c_out, k) && !s.c1[m * nd].w16.p) {
                // narrow stages: one fused launch per layer; no in-place update (blocks read neighbours' columns),
                // so the layer outputs ping-pong between Y and T1
                for (int j = 0; j < nd; ++j) {
                    const bool last = j + 1 == nd;
                    float *yout = last ? cur : (j % 2 == 0 ? Y : T1);
                    const float *acc_in = (last && !first_branch) ? cur : nullptr;
                    const float scale = (last && final_branch) ? 1.f / (float)nk : 1.f;
                    if (last && before_last()) return 1;
                    if (launch_resblock_layer(xin, s.c1[m * nd + j].w.p, s.c1[m * nd + j].b.p, s.c2[m * nd + j].w.p,
                                              s.c2[m * nd + j].b.p, acc_in, yout, batch, s.c_out, len, k, c.res_dilations[j], 0.1f,
                                              scale, stream))
                        return 1;
                    xin = yout;
                }
                if (after_last()) return 1;
                continue;
            }
            for (int j = 0; j < nd; ++j) {
                const int dil = c.res_dilations[j];
                ConvParams p;
                p.x1 = xin; p.c1 = s.c_out; p.slope1 = 0.1f; p.x1_bstride = bs; p.l_in = len;
                p.w = s.c1[m * nd + j].w.p; p.w16 = s.c1[m * nd + j].w16.p; p.bias = s.c1[m * nd + j].b.p; p.w_wino = s.c1[m * nd + j].wu.p; p.w_winobf = s.c1[m * nd + j].wx.p; p.w_direct1 = s.c1[m * nd + j].wd.p;
                p.w_wino16 = reinterpret_cast<const uint32_t *>(s.c1[m * nd + j].wu16.p);
                p.y = T1; p.y_bstride = bs; p.m_total = s.c_out; p.c_out = s.c_out; p.n_cols = len; p.l_out = len;
                p.kw = k; p.dil = dil; p.padl = (k - 1) / 2 * dil; p.batch = batch;
                if (launch_conv(p, stream)) return 1;
                ConvParams q;
                q.x1 = T1; q.c1 = s.c_out; q.slope1 = 0.1f; q.x1_bstride = bs; q.l_in = len;
                q.w = s.c2[m * nd + j].w.p; q.w16 = s.c2[m * nd + j].w16.p; q.bias = s.c2[m * nd + j].b.p; q.w_wino = s.c2[m * nd + j].wu.p; q.w_winobf = s.c2[m * nd + j].wx.p; q.w_direct1 = s.c2[m * nd + j].wd.p;
                q.w_wino16 = reinterpret_cast<const uint32_t *>(s.c2[m * nd + j].wu16.p);
                q.res = xin;
                q.y_bstride = bs; q.m_total = s.c_out; q.c_out = s.c_out; q.n_cols = len; q.l_out = len;
                q.kw = k; q.dil = 1; q.padl = (k - 1) / 2;  q.batch = batch;
                if (j + 1 < nd) {
                    q.y = Y;
                } else {  // last layer of the branch: add into the running sum, scale the last one by 1/nk
                    q.y = cur;
                    q.accin = first_branch ? nullptr : cur;
                    q.out_scale = final_branch ? 1.f / (float)nk : 1.f;
                    if (before_last()) return 1;
                }
                if (launch_conv(q, stream)) return 1;
                xin = Y;
            }
            if (after_last()) return 1;
        }
        if (par) RVC_HIP(hipStreamWaitEvent(main_stream, lanes->last[nk - 1], 0));   // (transitively every branch: each last launch waited for its predecessor's)
        if (d->tap_stage == i && d->tap_dev)
            RVC_HIP(hipMemcpyAsync(d->tap_dev, cur, (size_t)batch * bs * 4, hipMemcpyDeviceToDevice, stream));
    }
    if (len != L) return fail("decoder: internal length mismatch (%lld vs %lld)", (long long)len, (long long)L);
    return launch_conv_post(cur, d->post_w.p, d->post_b, batch, d->post_cin, L, 0.01f, out_dev, stream);
}
