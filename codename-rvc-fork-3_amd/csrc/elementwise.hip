// K8 -- the per-conv epilogue of the RMVPE U-Net as ONE pass: out = relu(x + bias[c]) [+ res]
// (ConvBlockRes.forward, rvc/lib/predictors/RMVPE.py:25-64 with eval-mode BatchNorm folded into the conv: bias is the
// folded shift).  PyTorch runs this as a broadcast add, a ReLU and a residual add -- three launches and three trips
// through HBM per conv, ~170 small launches per utterance.  HBM-bound: one read of x (and res), one write.
#include <stdint.h>

#include "common.h"

namespace rvc {

__global__ void __launch_bounds__(256)
bias_relu_add_kernel(const float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ res,
                     float *__restrict__ out, int channels, int64_t inner4, int relu) {
    // grid.y = batch * channels; a row of `inner` floats (multiple of 4) per (b, c)
    const int64_t row = blockIdx.y;
    const float bv = bias ? bias[row % channels] : 0.f;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x) + row * inner4;
    const f32x4 *rr = res ? reinterpret_cast<const f32x4 *>(res) + row * inner4 : nullptr;
    f32x4 *orow = reinterpret_cast<f32x4 *>(out) + row * inner4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < inner4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = xr[i] + bv;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (rr) v += rr[i];
        orow[i] = v;
    }
}

// WaveNet gate (commons.py:142-157 fused_add_tanh_sigmoid_multiply, the conditioning already added by the producing
// conv's bias): acts[c][t] = tanh(x[c][t]) * sigmoid(x[H + c][t]),  x [B][2H][T] -> acts [B][H][T]
__global__ void __launch_bounds__(256)
gate_tanh_sigmoid_kernel(const float *__restrict__ x, float *__restrict__ out, int hidden, int64_t T) {
    const int64_t b = blockIdx.z;
    const int c = blockIdx.y;
    const float *xa = x + (b * 2 * hidden + c) * T;
    const float *xb = xa + (int64_t)hidden * T;
    float *o = out + (b * hidden + c) * T;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (int64_t)gridDim.x * 256)
        o[t] = tanhf(xa[t]) * (1.f / (1.f + expf(-xb[t])));
}


// ---- per-channel GroupNorm + GELU (HuBERT's first layer) ---------------------------------------------------------------------
// One block per (batch, channel) row: sum and sum of squares accumulated in float64 (375 elements per thread at 30 s), then the
// row is re-read, normalised, scaled and passed through the exact (erf) GELU.  Three library launches and two extra passes less.
__global__ void __launch_bounds__(256)
rownorm_gelu_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ y,
                    int channels, int64_t L, float eps) {
    __shared__ double red[2][4];
    const int64_t row = blockIdx.x;
    const int c = (int)(row % channels);
    const float *xr = x + row * L;
    float *yr = y + row * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0, q = 0.0;
    const int64_t L4 = (L % 4 == 0 && (reinterpret_cast<uintptr_t>(xr) & 15) == 0) ? L / 4 : 0;
    for (int64_t i = tid; i < L4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(xr)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += (double)v[e]; q += (double)v[e] * (double)v[e]; }
    }
    for (int64_t i = 4 * L4 + tid; i < L; i += 256) { const double v = xr[i]; s += v; q += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mean = s / (double)L;
    double var = q / (double)L - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)mean;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    auto f = [&](float v) __attribute__((always_inline)) {
        const float n = (v - mu) * rstd * g + b;
        return 0.5f * n * (1.f + erff(n * 0.70710678118654752f));
    };
    const bool y4 = L4 && (reinterpret_cast<uintptr_t>(yr) & 15) == 0;
    if (y4) {
        for (int64_t i = tid; i < L4; i += 256) {
            f32x4 v = reinterpret_cast<const f32x4 *>(xr)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f(v[e]);
            reinterpret_cast<f32x4 *>(yr)[i] = v;
        }
        for (int64_t i = 4 * L4 + tid; i < L; i += 256) yr[i] = f(xr[i]);
    } else {
        for (int64_t i = tid; i < L; i += 256) yr[i] = f(xr[i]);
    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_gate_tanh_sigmoid_f32(const float *x_dev, float *out_dev, int batch, int hidden, int64_t length,
                                         void *stream) {
    if (!x_dev || !out_dev) return fail("rvc_gate_tanh_sigmoid_f32: null pointer");
    if (batch <= 0 || hidden <= 0 || length < 0) return fail("rvc_gate_tanh_sigmoid_f32: bad shape");
    if (length == 0) return 0;
    int64_t bx = ceil_div(length, 256);
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(gate_tanh_sigmoid_kernel, dim3((unsigned)bx, (unsigned)hidden, (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, x_dev, out_dev, hidden, length);
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_rownorm_gelu_f32(const float *x_dev, const float *gamma_dev, const float *beta_dev, float *out_dev, int batch,
                                    int channels, int64_t length, float eps, void *stream) {
    if (!x_dev || !out_dev) return fail("rvc_rownorm_gelu_f32: null pointer");
    if (batch <= 0 || channels <= 0 || length <= 0 || !(eps >= 0.f)) return fail("rvc_rownorm_gelu_f32: bad shape");
    hipLaunchKernelGGL(rownorm_gelu_kernel, dim3((unsigned)((int64_t)batch * channels)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       gamma_dev, beta_dev, out_dev, channels, length, eps);
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_bias_relu_add_f32(const float *x_dev, const float *bias_dev, const float *res_dev, float *out_dev,
                                     int batch, int channels, int64_t inner, int relu, void *stream) {
    if (!x_dev || !out_dev) return fail("rvc_bias_relu_add_f32: null pointer");
    if (batch <= 0 || channels <= 0 || inner < 0) return fail("rvc_bias_relu_add_f32: bad shape");
    if (inner % 4) return fail("rvc_bias_relu_add_f32: the inner extent (%lld) must be a multiple of 4", (long long)inner);
    if (inner == 0) return 0;
    const int64_t inner4 = inner / 4;
    int64_t bx = ceil_div(inner4, 256);
    if (bx > 64) bx = 64;
    dim3 grid((unsigned)bx, (unsigned)(batch * channels));
    hipLaunchKernelGGL(bias_relu_add_kernel, grid, dim3(256), 0, (hipStream_t)stream, x_dev, bias_dev, res_dev, out_dev, channels,
                       inner4, relu);
    RVC_LAUNCH_CHECK();
    return 0;
}

// ---- polyphase FIR resampler (load_audio / load_audio_infer: rvc/lib/utils.py:21-50, 53-85 resample to 16 kHz) ----------
// y[j] = sum_m h[m] * xup[j * down + centre - m], xup = x with (up - 1) zeros between samples (scipy.signal.upfirdn /
// resample_poly convention, centre = (n_taps - 1) / 2).  Only every up-th tap meets a non-zero sample: ~n_taps / up
// multiply-adds per output, in float64 (the filter is designed for > 140 dB of stop-band attenuation, fp32 accumulation
// would sit above its noise floor).  One thread per output sample; x and h are re-read from L2.
namespace rvc {
__global__ void __launch_bounds__(256)
resample_poly_kernel(const double *__restrict__ x, int64_t n_in, int up, int down, const double *__restrict__ h, int64_t n_taps,
                     double *__restrict__ y, int64_t n_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    const int64_t centre = (n_taps - 1) / 2;
    const int64_t t = j * down + centre;         // position on the up-sampled grid
    // taps m with (t - m) % up == 0 and 0 <= (t - m) / up < n_in
    int64_t m = t % up;                           // smallest m >= 0 with (t - m) divisible by up
    int64_t i = (t - m) / up;                     // input sample that tap m meets
    if (i >= n_in) {
        const int64_t skip = i - (n_in - 1);
        m += skip * up;
        i = n_in - 1;
    }
    double acc = 0.0;
    for (; m < n_taps && i >= 0; m += up, --i) acc = fma(h[m], x[i], acc);
    y[j] = acc;
}
}  // namespace rvc

extern "C" int rvc_resample_poly_f64(const double *x_dev, int64_t n_in, int up, int down, const double *h_dev, int64_t n_taps,
                                     double *y_dev, int64_t n_out, void *stream) {
    if (!x_dev || !h_dev || !y_dev) return rvc::fail("rvc_resample_poly_f64: null pointer");
    if (n_in <= 0 || up <= 0 || down <= 0 || n_taps <= 0 || n_out < 0) return rvc::fail("rvc_resample_poly_f64: bad argument");
    if (n_out == 0) return 0;
    hipLaunchKernelGGL(rvc::resample_poly_kernel, dim3((unsigned)rvc::ceil_div(n_out, 256)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       n_in, up, down, h_dev, n_taps, y_dev, n_out);
    RVC_LAUNCH_CHECK();
    return 0;
}
