// K13 -- the first layer of HuBERT's feature extractor (transformers' HubertGroupNormConvLayer behind rvc/infer/pipeline.py:450:
// Conv1d(1 -> 512, 10 taps, stride 5, no bias) -> GroupNorm(512 groups of one channel: statistics over time) -> GELU) producing
// what K12 (linbf.hip) reads for the strided convs that follow: TIME-MAJOR frames [frame][channel] as three bf16 planes (the exact
// split of each fp32 value).
//
// The layer is 1 GFLOP on 1.9 MB of audio; what it costs is its 196 MB fp32 output (30 s clip), which the library graph writes,
// re-reads for the statistics, re-reads and re-writes for the normalisation + GELU (round 3: conv 88 us + K8 262 us), and which K12
// would then need transposed.  Here the conv is computed TWICE instead and its fp32 output never exists:
//   pass 1 (hubert_conv0_stats_kernel + _finish_kernel): blocks of (4 channels, an eighth of the clip) accumulate sum and sum of
//     squares of the conv outputs in float64; the eight partial sums meet in a fixed order -> mean and 1 / sqrt(var + eps) per
//     channel -- reads the audio only;
//   pass 2 (hubert_conv0_apply_kernel): one block per 32 frames x 512 channels recomputes the same fmaf chain (bit-identical values),
//     normalises, applies the erf GELU, splits, and writes 16 bytes per (frame, 8 channels, plane): 295 MB of writes, no reads.
#include "conv.h"

namespace rvc {

typedef float hf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hf_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned hf_u32x4 __attribute__((ext_vector_type(4)));

// conv output of one (frame, channel): the SAME chain in both passes
template <int TAPS>
__device__ __forceinline__ float hf_conv(const float (&x)[TAPS], const float *w) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc = fmaf(w[k], x[k], acc);
    return acc;
}

constexpr int HF_CHUNKS = 8;   // time chunks of the statistics pass (x C / 4 channel groups = 1024 blocks at 512 channels)

// block (channel group of 4, time chunk): partial sum and sum of squares in float64 -> part[chunk][channel][2]
template <int TAPS>
__global__ void __launch_bounds__(256)
hubert_conv0_stats_kernel(const float *__restrict__ wav, int64_t n_frames, int stride, const float *__restrict__ w, int C,
                          double *__restrict__ part) {
    __shared__ double red[8][4];
    const int c0 = blockIdx.x * 4, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (n_frames + HF_CHUNKS - 1) / HF_CHUNKS;
    const int64_t t_lo = blockIdx.y * per, t_hi = t_lo + per < n_frames ? t_lo + per : n_frames;
    float wr[4][TAPS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) wr[c][k] = w[(c0 + c) * TAPS + k];
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int64_t t = t_lo + tid; t < t_hi; t += 256) {
        float x[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) x[k] = wav[t * stride + k];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double v = (double)hf_conv<TAPS>(x, wr[c]);
            s[c] += v;
            q[c] += v * v;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s[c] += __shfl_xor(s[c], off); q[c] += __shfl_xor(q[c], off); }
        if (lane == 0) { red[c][wave] = s[c]; red[4 + c][wave] = q[c]; }
    }
    __syncthreads();
    if (tid < 4) {
        double *o = part + ((int64_t)blockIdx.y * C + c0 + tid) * 2;
        o[0] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
        o[1] = (red[4 + tid][0] + red[4 + tid][1]) + (red[4 + tid][2] + red[4 + tid][3]);
    }
}

// one thread per channel: the chunks' sums in a fixed order -> mean, 1 / sqrt(var + eps) (biased variance, like torch's group_norm)
__global__ void __launch_bounds__(256)
hubert_conv0_finish_kernel(const double *__restrict__ part, int C, int64_t n_frames, float eps, float *__restrict__ stats) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double sum = 0.0, sq = 0.0;
    for (int h = 0; h < HF_CHUNKS; ++h) { sum += part[((int64_t)h * C + c) * 2]; sq += part[((int64_t)h * C + c) * 2 + 1]; }
    const double mean = sum / (double)n_frames;
    double var = sq / (double)n_frames - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[2 * c] = (float)mean;
    stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// block = 32 frames x C channels; thread (frame f = tid / 8, channel group cg = tid % 8) -> channels 64 j + 8 cg + 0..7, j = 0 .. C / 64 - 1
template <int TAPS>
__global__ void __launch_bounds__(256)
hubert_conv0_apply_kernel(const float *__restrict__ wav, int64_t n_frames, int stride, const float *__restrict__ w,
                          const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                          unsigned char *__restrict__ ys, int64_t n_pad, int C) {
    extern __shared__ __attribute__((aligned(16))) float hf_smem[];
    float *const wt = hf_smem;                 // [TAPS][C]
    float *const tab = hf_smem + TAPS * C;     // [4][C]: mean, rstd, gamma, beta
    const int tid = threadIdx.x;
    for (int i = tid; i < TAPS * C; i += 256) {
        const int c = i / TAPS, k = i - c * TAPS;
        wt[k * C + c] = w[i];
    }
    for (int c = tid; c < C; c += 256) {
        tab[c] = stats[2 * c];
        tab[C + c] = stats[2 * c + 1];
        tab[2 * C + c] = gamma ? gamma[c] : 1.f;
        tab[3 * C + c] = beta ? beta[c] : 0.f;
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 32 + (tid >> 3);
    if (t >= n_frames) return;
    const int cg = tid & 7;
    float x[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) x[k] = wav[t * stride + k];
    for (int j = 0; j < C / 64; ++j) {
        const int c0 = j * 64 + cg * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wt + k * C + c0), w1 = *reinterpret_cast<const f32x4 *>(wt + k * C + c0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = fmaf(w0[e], x[k], o[e]); o[4 + e] = fmaf(w1[e], x[k], o[4 + e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const float n = (o[e] - tab[c]) * tab[C + c] * tab[2 * C + c] + tab[3 * C + c];
            o[e] = 0.5f * n * (1.f + erff(n * 0.70710678118654752f));
        }
        hf_f32x2 v[4] = {{o[0], o[1]}, {o[2], o[3]}, {o[4], o[5]}, {o[6], o[7]}};
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            hf_u32x4 pk;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const unsigned wd = __builtin_bit_cast(unsigned, __builtin_convertvector(v[h], hf_bf16x2));
                pk[h] = wd;
                v[h] = v[h] - hf_f32x2{__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)};
            }
            *reinterpret_cast<hf_u32x4 *>(ys + (((int64_t)sp * n_pad + t) * C + c0) * 2) = pk;
        }
    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_hubert_conv0_workspace_bytes(int channels, size_t *bytes) {
    if (!bytes || channels <= 0) return fail("rvc_hubert_conv0_workspace_bytes: bad argument");
    *bytes = (size_t)channels * 2 * sizeof(float) + (size_t)HF_CHUNKS * channels * 2 * sizeof(double);
    return 0;
}

extern "C" int rvc_hubert_conv0_frames_bf16x3(const float *wav_dev, int64_t n_samples, const float *w_dev, int channels, int taps,
                                              int stride, const float *gamma_dev, const float *beta_dev, float eps, void *workspace_dev,
                                              size_t workspace_bytes, void *ys_dev, int64_t n_frames_padded, void *stream) {
    if (!wav_dev || !w_dev || !workspace_dev || !ys_dev) return fail("rvc_hubert_conv0_frames_bf16x3: null pointer");
    if (taps != 10) return fail("rvc_hubert_conv0_frames_bf16x3: built for HuBERT's 10-tap first layer, got %d taps", taps);
    if (channels % 64 || channels <= 0 || channels > 1024 || stride < 1 || !(eps >= 0.f)) return fail("rvc_hubert_conv0_frames_bf16x3: bad argument");
    const int64_t n_frames = n_samples >= taps ? (n_samples - taps) / stride + 1 : 0;
    if (n_frames <= 0) return fail("rvc_hubert_conv0_frames_bf16x3: the clip is shorter than one window");
    if (n_frames_padded < n_frames) return fail("rvc_hubert_conv0_frames_bf16x3: n_frames_padded < frames");
    size_t need = 0;
    (void)rvc_hubert_conv0_workspace_bytes(channels, &need);
    if (workspace_bytes < need) return fail("rvc_hubert_conv0_frames_bf16x3: workspace too small (%zu < %zu)", workspace_bytes, need);
    double *part = reinterpret_cast<double *>(workspace_dev);                                  // (first: 8-byte aligned)
    float *stats = reinterpret_cast<float *>(part + (size_t)HF_CHUNKS * channels * 2);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hubert_conv0_stats_kernel<10>, dim3((unsigned)(channels / 4), HF_CHUNKS), dim3(256), 0, st, wav_dev, n_frames, stride,
                       w_dev, channels, part);
    RVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(hubert_conv0_finish_kernel, dim3((unsigned)ceil_div(channels, 256)), dim3(256), 0, st, part, channels, n_frames, eps, stats);
    RVC_LAUNCH_CHECK();
    const size_t lds = (size_t)(taps + 4) * channels * sizeof(float);
    hipLaunchKernelGGL(hubert_conv0_apply_kernel<10>, dim3((unsigned)ceil_div(n_frames, 32)), dim3(256), lds, st, wav_dev, n_frames, stride,
                       w_dev, stats, gamma_dev, beta_dev, reinterpret_cast<unsigned char *>(ys_dev), n_frames_padded, channels);
    RVC_LAUNCH_CHECK();
    return 0;
}
