// K4 -- RMVPE's log-mel front end (RMVPE.py:342-417 at the :438 parameters): reflect-pad 512, periodic
// hann(1024), hop 160, |rFFT|, 128 HTK mel bands 30..8000 Hz (Slaney-normalised), log(clamp(., 1e-5)), plus the
// reflect pad of the frame axis to a multiple of 32 that mel2hidden applies (RMVPE.py:452-455).
//
// 3201 frames x 1024 points is 7.5 GFLOP as a dense DFT -- 50 microseconds of matrix-core time -- so the
// transform is a GEMM on the same fp32 MFMA kernel as the vocoder (windowed frames are the "channels",
// the [cos; -sin] basis the weights) instead of a radix FFT: no bit reversal, no twiddle recurrences, one
// rounding per product.  The mel projection is sparse (1010 non-zeros) and runs as a gather epilogue.
#include <math.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

constexpr int LM_NFFT = 1024;
constexpr int LM_HOP = 160;
constexpr int LM_BINS = 513;
constexpr int LM_ROWS = 1152;      // re rows [0,513) pad to 576, im rows [576, 576+513) pad to 1152 (multiple of 128)
constexpr int LM_IM0 = 576;
constexpr int LM_MELS = 128;

struct LogmelTables {
    float *basis = nullptr;        // [1][1024][1152] packed conv weight
    float *window = nullptr;       // [1024]
    int *mel_start = nullptr;      // [128]
    int *mel_count = nullptr;      // [128]
    float *mel_w = nullptr;        // [128][max_count]
    int max_count = 0;
    bool ready = false;
};
static LogmelTables g_tab;
static std::mutex g_tab_mutex;

static double hz_to_mel_htk(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel_to_hz_htk(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

static int build_tables() {
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    if (g_tab.ready) return 0;
    // DFT basis, double -> float
    std::vector<float> basis((size_t)LM_NFFT * LM_ROWS, 0.f);
    for (int n = 0; n < LM_NFFT; ++n)
        for (int f = 0; f < LM_BINS; ++f) {
            const int ph = (int)(((int64_t)f * n) % LM_NFFT);  // exact argument reduction
            const double a = 2.0 * M_PI * (double)ph / LM_NFFT;
            basis[(size_t)n * LM_ROWS + f] = (float)cos(a);
            basis[(size_t)n * LM_ROWS + LM_IM0 + f] = (float)(-sin(a));
        }
    std::vector<float> window(LM_NFFT);
    for (int n = 0; n < LM_NFFT; ++n) window[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / LM_NFFT));  // periodic hann
    // librosa.filters.mel(sr=16000, n_fft=1024, n_mels=128, fmin=30, fmax=8000, htk=True, norm="slaney")
    const double sr = 16000.0, fmin = 30.0, fmax = 8000.0;
    std::vector<double> mel_f(LM_MELS + 2);
    const double m0 = hz_to_mel_htk(fmin), m1 = hz_to_mel_htk(fmax);
    const double step = (m1 - m0) / (LM_MELS + 1);  // np.linspace: arange * step + start, last point = stop
    for (int i = 0; i < LM_MELS + 2; ++i) mel_f[i] = mel_to_hz_htk(i == LM_MELS + 1 ? m1 : i * step + m0);
    std::vector<std::vector<float>> rows(LM_MELS);
    std::vector<int> start(LM_MELS, 0), count(LM_MELS, 0);
    int max_count = 1;
    for (int i = 0; i < LM_MELS; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, last = -1;
        std::vector<float> w(LM_BINS, 0.f);
        for (int k = 0; k < LM_BINS; ++k) {
            const double fk = k * sr / LM_NFFT;
            const double lower = (fk - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fk) / (mel_f[i + 2] - mel_f[i + 1]);
            const double v = lower < upper ? lower : upper;
            // librosa: float32 weights = max(0, min(lower, upper)); then `weights *= enorm` (double product, cast back)
            float wf = (float)(v > 0.0 ? v : 0.0);
            wf = (float)((double)wf * enorm);
            w[k] = wf;
            if (wf != 0.f) { if (first < 0) first = k; last = k; }
        }
        if (first >= 0) { start[i] = first; count[i] = last - first + 1; rows[i].assign(w.begin() + first, w.begin() + last + 1); }
        if (count[i] > max_count) max_count = count[i];
    }
    std::vector<float> mel_w((size_t)LM_MELS * max_count, 0.f);
    for (int i = 0; i < LM_MELS; ++i)
        for (int j = 0; j < count[i]; ++j) mel_w[(size_t)i * max_count + j] = rows[i][j];

    auto up = [](const void *h, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes);
        if (e == hipSuccess) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(basis.data(), basis.size() * 4, (void **)&g_tab.basis);
    if (e == hipSuccess) e = up(window.data(), window.size() * 4, (void **)&g_tab.window);
    if (e == hipSuccess) e = up(start.data(), start.size() * 4, (void **)&g_tab.mel_start);
    if (e == hipSuccess) e = up(count.data(), count.size() * 4, (void **)&g_tab.mel_count);
    if (e == hipSuccess) e = up(mel_w.data(), mel_w.size() * 4, (void **)&g_tab.mel_w);
    if (e != hipSuccess) return fail("logmel: table upload failed: %s", hipGetErrorString(e));
    g_tab.max_count = max_count;
    g_tab.ready = true;
    return 0;
}

// F[b][n][t] = window[n] * audio[b][reflect(t*160 + n - 512)]
__global__ void __launch_bounds__(256)
logmel_frames_kernel(const float *__restrict__ audio, int64_t n_samples, const float *__restrict__ window,
                     int64_t n_frames, float *__restrict__ F) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (t >= n_frames) return;
    int64_t j = t * LM_HOP + n - LM_NFFT / 2;
    if (j < 0) j = -j;
    if (j >= n_samples) j = 2 * (n_samples - 1) - j;
    F[(b * LM_NFFT + n) * n_frames + t] = window[n] * audio[b * n_samples + j];
}

// mel[b][m][t] = log(max(sum_k w[m][k] * |C[k] + i C[576 + k]|, 1e-5)); frames >= n_frames mirror (reflect pad)
__global__ void __launch_bounds__(256)
logmel_mel_kernel(const float *__restrict__ C, int64_t n_frames, int64_t n_frames_padded, const int *__restrict__ mel_start,
                  const int *__restrict__ mel_count, const float *__restrict__ mel_w, int max_count, float *__restrict__ mel) {
    const int64_t tp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (tp >= n_frames_padded) return;
    const int64_t t = tp < n_frames ? tp : 2 * (n_frames - 1) - tp;
    const float *Cb = C + b * LM_ROWS * n_frames;
    const int k0 = mel_start[m], cnt = mel_count[m];
    float acc = 0.f;
    for (int j = 0; j < cnt; ++j) {
        const float re = Cb[(int64_t)(k0 + j) * n_frames + t];
        const float im = Cb[(int64_t)(LM_IM0 + k0 + j) * n_frames + t];
        acc = fmaf(mel_w[m * max_count + j], sqrtf(re * re + im * im), acc);
    }
    mel[(b * LM_MELS + m) * n_frames_padded + tp] = logf(fmaxf(acc, 1e-5f));
}

// ---- the same transform for any (n_fft, hop, window, pad, mel matrix): the training-side spectrogram / mel of
// rvc/train/mel_processing.py:53-146 (n_fft 2048, hop 480, 128 Slaney mel bands at 48 kHz; reflect pad (n_fft - hop) / 2,
// center = False, |X| = sqrt(re^2 + im^2 + 1e-6), log(clamp(mel, 1e-5))) is the second consumer of this kernel chain ----
struct MelPlan {
    int n_fft = 0, hop = 0, win = 0, pad = 0, bins = 0, im0 = 0, rows = 0, n_mels = 0, max_count = 0;
    float mag_eps = 0.f, log_floor = 1e-5f;
    float *basis = nullptr, *window = nullptr, *mel_w = nullptr;
    int *mel_start = nullptr, *mel_count = nullptr;
    ~MelPlan() {
        for (void *q : {(void *)basis, (void *)window, (void *)mel_w, (void *)mel_start, (void *)mel_count})
            if (q) (void)hipFree(q);
    }
};

// F[b][n][t] = window[n] * audio[b][reflect(t*hop + n - pad)]   (np.pad-style reflection, repeated if the pad exceeds the signal)
__global__ void __launch_bounds__(256)
mel_frames_kernel(const float *__restrict__ audio, int64_t n_samples, const float *__restrict__ window, int n_fft, int hop, int pad,
                  int64_t n_frames, float *__restrict__ F) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (t >= n_frames) return;
    int64_t j = t * hop + n - pad;
    const int64_t period = 2 * (n_samples - 1);
    if (period > 0) {
        j %= period;
        if (j < 0) j += period;
        if (j >= n_samples) j = period - j;
    } else {
        j = 0;
    }
    F[(b * n_fft + n) * n_frames + t] = window[n] * audio[b * n_samples + j];
}

// spec[b][k][t] = sqrt(re^2 + im^2 + eps) (optional output); mel[b][m][t] = log(max(sum_k w[m][k] spec[k], floor))
__global__ void __launch_bounds__(256)
mel_project_kernel(const float *__restrict__ C, int64_t n_frames, int rows, int im0, const int *__restrict__ mel_start,
                   const int *__restrict__ mel_count, const float *__restrict__ mel_w, int max_count, int n_mels, float mag_eps,
                   float log_floor, float *__restrict__ mel) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (t >= n_frames) return;
    const float *Cb = C + b * rows * n_frames;
    const int k0 = mel_start[m], cnt = mel_count[m];
    float acc = 0.f;
    for (int j = 0; j < cnt; ++j) {
        const float re = Cb[(int64_t)(k0 + j) * n_frames + t];
        const float im = Cb[(int64_t)(im0 + k0 + j) * n_frames + t];
        acc = fmaf(mel_w[m * max_count + j], sqrtf(re * re + im * im + mag_eps), acc);
    }
    mel[(b * n_mels + m) * n_frames + t] = logf(fmaxf(acc, log_floor));
}

__global__ void __launch_bounds__(256)
mel_magnitude_kernel(const float *__restrict__ C, int64_t n_frames, int rows, int im0, int bins, float mag_eps, float *__restrict__ spec) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (t >= n_frames) return;
    const float re = C[(b * rows + k) * n_frames + t], im = C[(b * rows + im0 + k) * n_frames + t];
    spec[(b * bins + k) * n_frames + t] = sqrtf(re * re + im * im + mag_eps);
}

}  // namespace rvc

struct rvc_mel : rvc::MelPlan {};

using namespace rvc;

extern "C" int rvc_mel_create(int n_fft, int hop, int win_length, int pad, float mag_eps, float log_floor, const float *mel_host,
                              int n_mels, rvc_mel **out) {
    if (!out || !mel_host) return fail("rvc_mel_create: null pointer");
    if (n_fft < 64 || n_fft > 8192 || n_fft % 8 || hop <= 0 || win_length <= 0 || win_length > n_fft || pad < 0 || n_mels <= 0 || n_mels > 1024)
        return fail("rvc_mel_create: bad transform (n_fft %d, hop %d, win %d, pad %d, n_mels %d)", n_fft, hop, win_length, pad, n_mels);
    rvc_mel *h = new rvc_mel();
    h->n_fft = n_fft; h->hop = hop; h->win = win_length; h->pad = pad; h->n_mels = n_mels;
    h->mag_eps = mag_eps; h->log_floor = log_floor;
    h->bins = n_fft / 2 + 1;
    h->im0 = (int)align_up((size_t)h->bins, 64);
    h->rows = (int)align_up((size_t)(h->im0 + h->bins), 128);
    std::vector<float> basis((size_t)n_fft * h->rows, 0.f);
    for (int n = 0; n < n_fft; ++n)
        for (int f = 0; f < h->bins; ++f) {
            const int ph = (int)(((int64_t)f * n) % n_fft);
            const double a = 2.0 * M_PI * (double)ph / n_fft;
            basis[(size_t)n * h->rows + f] = (float)cos(a);
            basis[(size_t)n * h->rows + h->im0 + f] = (float)(-sin(a));
        }
    // torch.hann_window(win_length) (periodic), centred inside n_fft as torch.stft does
    std::vector<float> window(n_fft, 0.f);
    const int w0 = (n_fft - win_length) / 2;
    for (int n = 0; n < win_length; ++n) window[w0 + n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / win_length));
    std::vector<int> start(n_mels, 0), count(n_mels, 0);
    int max_count = 1;
    for (int i = 0; i < n_mels; ++i) {
        int first = -1, last = -1;
        for (int k = 0; k < h->bins; ++k)
            if (mel_host[(size_t)i * h->bins + k] != 0.f) { if (first < 0) first = k; last = k; }
        if (first >= 0) { start[i] = first; count[i] = last - first + 1; }
        if (count[i] > max_count) max_count = count[i];
    }
    std::vector<float> mel_w((size_t)n_mels * max_count, 0.f);
    for (int i = 0; i < n_mels; ++i)
        for (int j = 0; j < count[i]; ++j) mel_w[(size_t)i * max_count + j] = mel_host[(size_t)i * h->bins + start[i] + j];
    h->max_count = max_count;
    auto up = [](const void *src, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes);
        if (e == hipSuccess) e = hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(basis.data(), basis.size() * 4, (void **)&h->basis);
    if (e == hipSuccess) e = up(window.data(), window.size() * 4, (void **)&h->window);
    if (e == hipSuccess) e = up(start.data(), start.size() * 4, (void **)&h->mel_start);
    if (e == hipSuccess) e = up(count.data(), count.size() * 4, (void **)&h->mel_count);
    if (e == hipSuccess) e = up(mel_w.data(), mel_w.size() * 4, (void **)&h->mel_w);
    if (e != hipSuccess) { delete h; return fail("rvc_mel_create: table upload failed: %s", hipGetErrorString(e)); }
    *out = h;
    return 0;
}

extern "C" int rvc_mel_destroy(rvc_mel *h) {
    delete h;
    return 0;
}

extern "C" int rvc_mel_frames(const rvc_mel *h, int64_t n_samples, int64_t *n_frames) {
    if (!h || !n_frames) return fail("rvc_mel_frames: null pointer");
    const int64_t padded = n_samples + 2 * (int64_t)h->pad;
    *n_frames = padded >= h->n_fft ? (padded - h->n_fft) / h->hop + 1 : 0;
    return 0;
}

extern "C" int rvc_mel_workspace_bytes(const rvc_mel *h, int batch, int64_t n_samples, size_t *bytes) {
    if (!h || !bytes || batch <= 0 || n_samples <= 0) return fail("rvc_mel_workspace_bytes: bad argument");
    int64_t T = 0;
    rvc_mel_frames(h, n_samples, &T);
    *bytes = align_up((size_t)batch * h->n_fft * (size_t)T * 4, 256) + align_up((size_t)batch * h->rows * (size_t)T * 4, 256);
    return 0;
}

extern "C" int rvc_mel_forward(const rvc_mel *h, const float *audio_dev, int batch, int64_t n_samples, float *mel_dev, float *spec_dev,
                               void *workspace_dev, size_t workspace_bytes, void *stream_) {
    if (!h || !audio_dev || !workspace_dev || (!mel_dev && !spec_dev)) return fail("rvc_mel_forward: null pointer");
    int64_t T = 0;
    rvc_mel_frames(h, n_samples, &T);
    if (batch <= 0 || T <= 0) return fail("rvc_mel_forward: %lld samples give no frame (n_fft %d, pad %d)", (long long)n_samples, h->n_fft, h->pad);
    size_t need = 0;
    if (rvc_mel_workspace_bytes(h, batch, n_samples, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_mel_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t stream = (hipStream_t)stream_;
    float *F = (float *)workspace_dev;
    float *C = (float *)((char *)workspace_dev + align_up((size_t)batch * h->n_fft * (size_t)T * 4, 256));
    hipLaunchKernelGGL(mel_frames_kernel, dim3((unsigned)ceil_div(T, 256), h->n_fft, batch), dim3(256), 0, stream, audio_dev, n_samples,
                       h->window, h->n_fft, h->hop, h->pad, T, F);
    RVC_LAUNCH_CHECK();
    ConvParams p;
    p.x1 = F; p.c1 = h->n_fft; p.slope1 = 1.f; p.x1_bstride = (int64_t)h->n_fft * T; p.l_in = T;
    p.w = h->basis;
    p.y = C; p.y_bstride = (int64_t)h->rows * T; p.m_total = h->rows; p.c_out = h->rows; p.n_cols = T; p.l_out = T;
    p.kw = 1; p.dil = 1; p.padl = 0; p.batch = batch;
    if (launch_conv(p, stream)) return 1;
    if (spec_dev) {
        hipLaunchKernelGGL(mel_magnitude_kernel, dim3((unsigned)ceil_div(T, 256), h->bins, batch), dim3(256), 0, stream, C, T, h->rows,
                           h->im0, h->bins, h->mag_eps, spec_dev);
        RVC_LAUNCH_CHECK();
    }
    if (mel_dev) {
        hipLaunchKernelGGL(mel_project_kernel, dim3((unsigned)ceil_div(T, 256), h->n_mels, batch), dim3(256), 0, stream, C, T, h->rows,
                           h->im0, h->mel_start, h->mel_count, h->mel_w, h->max_count, h->n_mels, h->mag_eps, h->log_floor, mel_dev);
        RVC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int rvc_logmel_workspace_bytes(int batch, int64_t n_samples, size_t *bytes) {
    if (!bytes || batch <= 0 || n_samples <= LM_NFFT / 2) return fail("rvc_logmel_workspace_bytes: bad argument");
    const int64_t T = n_samples / LM_HOP + 1;
    *bytes = align_up((size_t)batch * LM_NFFT * T * 4, 256) + align_up((size_t)batch * LM_ROWS * T * 4, 256);
    return 0;
}

extern "C" int rvc_logmel_rmvpe(const float *audio_dev, int batch, int64_t n_samples, float *mel_dev,
                                int64_t n_frames_padded, void *workspace_dev, size_t workspace_bytes, void *stream_) {
    if (!audio_dev || !mel_dev || !workspace_dev) return fail("rvc_logmel_rmvpe: null pointer");
    if (batch <= 0 || n_samples <= LM_NFFT / 2) return fail("rvc_logmel_rmvpe: need more than %d samples", LM_NFFT / 2);
    const int64_t T = n_samples / LM_HOP + 1;
    if (n_frames_padded < T || n_frames_padded - T >= T) return fail("rvc_logmel_rmvpe: bad n_frames_padded %lld for %lld frames", (long long)n_frames_padded, (long long)T);
    size_t need = 0;
    if (rvc_logmel_workspace_bytes(batch, n_samples, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_logmel_rmvpe: workspace too small (%zu < %zu)", workspace_bytes, need);
    if (!g_tab.ready && build_tables()) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    float *F = (float *)workspace_dev;
    float *C = (float *)((char *)workspace_dev + align_up((size_t)batch * LM_NFFT * T * 4, 256));
    hipLaunchKernelGGL(logmel_frames_kernel, dim3((unsigned)ceil_div(T, 256), LM_NFFT, batch), dim3(256), 0, stream, audio_dev,
                       n_samples, g_tab.window, T, F);
    RVC_LAUNCH_CHECK();
    ConvParams p;
    p.x1 = F; p.c1 = LM_NFFT; p.slope1 = 1.f; p.x1_bstride = (int64_t)LM_NFFT * T; p.l_in = T;
    p.w = g_tab.basis;
    p.y = C; p.y_bstride = (int64_t)LM_ROWS * T; p.m_total = LM_ROWS; p.c_out = LM_ROWS; p.n_cols = T; p.l_out = T;
    p.kw = 1; p.dil = 1; p.padl = 0; p.batch = batch;
    if (launch_conv(p, stream)) return 1;
    hipLaunchKernelGGL(logmel_mel_kernel, dim3((unsigned)ceil_div(n_frames_padded, 256), LM_MELS, batch), dim3(256), 0, stream, C,
                       T, n_frames_padded, g_tab.mel_start, g_tab.mel_count, g_tab.mel_w, g_tab.max_count, mel_dev);
    RVC_LAUNCH_CHECK();
    return 0;
}
