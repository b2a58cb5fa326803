// K6 -- zero-phase 5th-order IIR (the 48 Hz high-pass of rvc/infer/pipeline.py:23-28, applied with
// scipy.signal.filtfilt at pipeline.py:562) in float64 on the device, so the utterance never has to visit the host.
//
// filtfilt(b, a, x) = odd-extend by 18 samples, lfilter forward from zi*ext[0], lfilter backward from zi*y[-1],
// crop.  lfilter is a 5-state linear recurrence (transposed direct form II), strictly sequential as written.  It is
// made parallel by chunking with a warm-up: one thread per chunk of M output samples starts W samples earlier
// from a zero state (or from the true initial state zi*x0 when the warm-up reaches the start of the signal), runs
// SciPy's exact float64 operation sequence (no FMA contraction) and only writes its own M samples.  The filter's
// slowest pole has radius 0.9942, so a state error decays by 0.9942^4096 ~ 4e-11 over the warm-up.
//
// Accuracy: this direct-form high-pass is ill-conditioned (five poles within 0.02 of z = 1): two evaluations of the
// reference's OWN recurrence started 8192 samples apart already differ by 2e-8 (measured with scipy.signal.lfilter),
// so 2e-8 is the noise floor of the reference result itself.  The kernel agrees with scipy.signal.filtfilt to ~5e-8
// absolute on +-0.3 signals, below the float32 rounding (6e-8 relative) the audio undergoes right after
// (pipeline.py:445).  An explicit state-transition scan (A^M) was tried and rejected: the companion matrix is so
// non-normal (|A^256| ~ 4e7) that the scan is unstable in float64.
#include "common.h"

// SciPy's C loop is built without FMA; HIP's __dmul_rn/__dadd_rn are plain operators that clang would contract
// (default -ffp-contract=fast), and a 1-ulp change is amplified ~1e5x by this recurrence: contraction off for the file.
#pragma clang fp contract(off)

namespace rvc {

constexpr int FF_ORD = 5;
constexpr int FF_PAD = 18;   // 3 * max(len(a), len(b))

struct FiltCoef {
    double b[FF_ORD + 1];
    double a[FF_ORD + 1];
    double zi[FF_ORD];
};

// forward pass input: odd extension of x; backward pass input: the forward output reversed
__device__ __forceinline__ double ff_input(const double *__restrict__ x, int64_t n, const double *__restrict__ yf,
                                           int64_t i, int backward) {
    const int64_t ne = n + 2 * FF_PAD;
    if (backward) return yf[ne - 1 - i];
    if (i < FF_PAD) return 2.0 * x[0] - x[FF_PAD - i];
    if (i >= FF_PAD + n) return 2.0 * x[n - 1] - x[n - 2 - (i - FF_PAD - n)];
    return x[i - FF_PAD];
}

__device__ __forceinline__ double ff_step(const FiltCoef &c, double xv, double z[FF_ORD]) {
#pragma clang fp contract(off)
    // scipy/signal/_lfilter.c.src: y = z0 + b0*x;  z_i = z_{i+1} + x*b_{i+1} - y*a_{i+1};  z_last = x*b_n - y*a_n
    // plain operators under contract(off): the __dmul_rn/__dadd_rn helpers are ordinary operators that clang fuses
    const double y = z[0] + c.b[0] * xv;
#pragma unroll
    for (int i = 0; i < FF_ORD - 1; ++i) z[i] = (z[i + 1] + xv * c.b[i + 1]) - y * c.a[i + 1];
    z[FF_ORD - 1] = xv * c.b[FF_ORD] - y * c.a[FF_ORD];
    return y;
}

// forward pass -> yf[ne]; backward pass -> out[n] (reversed back and cropped)
__global__ void __launch_bounds__(64)
ff_chunk_kernel(const double *__restrict__ x, int64_t n, const double *yf_in, int backward, FiltCoef c, int chunk,
                int warm, int64_t n_chunks, double *yf_out, double *__restrict__ out) {
    const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_chunks) return;
    const int64_t ne = n + 2 * FF_PAD;
    const int64_t i0 = ch * chunk, i1 = min(ne, i0 + chunk);
    const int64_t w0 = i0 - warm;
    double z[FF_ORD];
    int64_t i;
    if (w0 <= 0) {  // exact start: zi * first input sample (scipy.signal.filtfilt)
        const double x0 = ff_input(x, n, yf_in, 0, backward);
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) z[k] = c.zi[k] * x0;
        i = 0;
    } else {
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) z[k] = 0.0;
        i = w0;
    }
    // the recurrence is serial but its inputs are not: fetch 8 samples ahead so the loads overlap the dependent chain
    constexpr int PF = 8;
    while (i < i1) {
        double xin[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) xin[u] = (i + u < i1) ? ff_input(x, n, yf_in, i + u, backward) : 0.0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (i + u < i1) {
                const double y = ff_step(c, xin[u], z);
                if (i + u >= i0) {
                    if (!backward) {
                        yf_out[i + u] = y;
                    } else {
                        const int64_t j = ne - 1 - (i + u) - FF_PAD;   // position in the un-reversed, cropped output
                        if (j >= 0 && j < n) out[j] = y;
                    }
                }
            }
        }
        i += PF;
    }
}

}  // namespace rvc

using namespace rvc;

constexpr int FF_CHUNK = 512;
constexpr int FF_WARM = 4096;

extern "C" int rvc_filtfilt_workspace_bytes(int64_t n, size_t *bytes) {
    if (!bytes || n <= FF_PAD) return fail("rvc_filtfilt_workspace_bytes: bad argument");
    *bytes = align_up((size_t)(n + 2 * FF_PAD) * 8, 256) * 2;
    return 0;
}

// coef_host: b[6], a[6] (a[0] == 1), zi[5] (scipy.signal.lfilter_zi)
extern "C" int rvc_filtfilt_order5(const double *x_dev, int64_t n, const double *coef_host, double *y_dev,
                                   void *workspace_dev, size_t workspace_bytes, void *stream_) {
    if (!x_dev || !coef_host || !y_dev || !workspace_dev) return fail("rvc_filtfilt_order5: null pointer");
    if (n <= FF_PAD) return fail("rvc_filtfilt_order5: the input must be longer than padlen = %d", FF_PAD);
    size_t need = 0;
    if (rvc_filtfilt_workspace_bytes(n, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_filtfilt_order5: workspace too small (%zu < %zu)", workspace_bytes, need);
    FiltCoef c;
    memcpy(c.b, coef_host, sizeof(c.b));
    memcpy(c.a, coef_host + 6, sizeof(c.a));
    memcpy(c.zi, coef_host + 12, sizeof(c.zi));
    if (c.a[0] != 1.0) return fail("rvc_filtfilt_order5: a[0] must be 1");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t ne = n + 2 * FF_PAD, n_chunks = ceil_div(ne, FF_CHUNK);
    double *yf = (double *)workspace_dev;
    const dim3 grid((unsigned)ceil_div(n_chunks, 64)), block(64);
    for (int backward = 0; backward < 2; ++backward) {
        hipLaunchKernelGGL(ff_chunk_kernel, grid, block, 0, stream, x_dev, n, yf, backward, c, FF_CHUNK, FF_WARM, n_chunks, yf,
                           y_dev);
        RVC_LAUNCH_CHECK();
    }
    return 0;
}
