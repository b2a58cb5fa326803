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
#include <string.h>

#include <mutex>
#include <vector>

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

// forward pass input: odd extension of x; backward pass input: the forward output reversed.  Branch-free: one
// unconditional load from a clamped index, the odd reflection applied afterwards (a load inside a divergent branch is
// waited for before the branch closes).  i outside [0, ne) returns a finite dummy that the caller does not use.
__device__ __forceinline__ double ff_input(const double *__restrict__ x, int64_t n, const double *__restrict__ yf,
                                           int64_t i, int backward) {
    const int64_t ne = n + 2 * FF_PAD;
    i = i < 0 ? 0 : (i >= ne ? ne - 1 : i);
    if (backward) return yf[ne - 1 - i];                     // uniform branch
    const bool left = i < FF_PAD, right = i >= FF_PAD + n;
    const int64_t idx = left ? FF_PAD - i : (right ? n - 2 - (i - FF_PAD - n) : i - FF_PAD);
    const double v = x[idx];
    const double edge = left ? x[0] : x[n - 1];
    return (left || right) ? 2.0 * edge - v : v;
}

__device__ __forceinline__ double ff_step(const FiltCoef &c, double xv, double z[FF_ORD]) {
#pragma clang fp contract(off)
    // scipy/signal/_lfilter.c.src: y = z0 + b0*x;  z_i = z_{i+1} + x*b_{i+1} - y*a_{i+1};  z_last = x*b_n - y*a_n
    // plain operators under contract(off): the __dmul_rn/__dadd_rn helpers are ordinary operators that clang fuses.
    // Same operations and operand order as SciPy's, written so that only three of the 21 are on the step-to-step
    // dependency chain (y, y*a1, z0): a float64 VALU result takes ~25 cycles to come back, and the straightforward
    // order (mul, dependent add, mul, dependent add, ...) made every one of them wait -- 550 cycles per sample.
    double xb[FF_ORD + 1], t[FF_ORD], ya[FF_ORD];
#pragma unroll
    for (int i = 0; i <= FF_ORD; ++i) xb[i] = xv * c.b[i];          // independent of the state
    const double y = z[0] + xb[0];
#pragma unroll
    for (int i = 0; i < FF_ORD - 1; ++i) t[i] = z[i + 1] + xb[i + 1];   // independent of y
#pragma unroll
    for (int i = 0; i < FF_ORD; ++i) ya[i] = y * c.a[i + 1];
#pragma unroll
    for (int i = 0; i < FF_ORD - 1; ++i) z[i] = t[i] - ya[i];
    z[FF_ORD - 1] = xb[FF_ORD] - ya[FF_ORD - 1];
    return y;
}

// One wave = 64 consecutive chunks, one lane per chunk.  A lane walks its chunk sequentially, so a naive load touches
// 64 different cache lines per step; instead the wave moves data in [64 chunks] x [32 steps] tiles through LDS: tile
// rows are fetched as contiguous 256-byte pieces (two chunk rows per load instruction), one step-block ahead of the
// recurrence, and the outputs of a step-block leave the same way.  Per-step cost drops from ~470 cycles (0.9 ms per
// pass) to the float64 VALU work of the recurrence itself.
// forward pass -> yf[ne]; backward pass -> out[n] (reversed back and cropped)
constexpr int FF_SB = 32;              // steps per tile
constexpr int FF_TS = FF_SB + 1;       // tile row stride in doubles (conflict-free ds_read_b64 down a column)

__global__ void __launch_bounds__(64)
ff_chunk_kernel(const double *__restrict__ x, int64_t n, const double *yf_in, int backward, FiltCoef c, int chunk,
                int warm, int64_t n_chunks, double *yf_out, double *__restrict__ out, const double *__restrict__ zstate) {
    __shared__ double tin[64 * FF_TS];
    __shared__ double tout[64 * FF_TS];
    const int lane = threadIdx.x, u = lane & 31, hrow = lane >> 5;
    const int64_t ch0 = (int64_t)blockIdx.x * 64;
    const int64_t ch = ch0 + lane;
    const int64_t ne = n + 2 * FF_PAD;
    const int64_t i0 = ch * chunk;                    // first output index of this lane's chunk
    const int64_t w0 = i0 - warm;                     // first (virtual) index it runs from; steps with i < 0 are skipped
    const int n_steps = warm + chunk;                 // multiple of FF_SB
    double z[FF_ORD];
    if (zstate) {   // (warm == 0) the state in front of the chunk's first sample, from ff_state_kernel
        const int64_t chc = ch < n_chunks ? ch : n_chunks - 1;
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) z[k] = zstate[chc * FF_ORD + k];
    } else if (w0 <= 0) {  // exact start: zi * first input sample (scipy.signal.filtfilt)
        const double x0 = ff_input(x, n, yf_in, 0, backward);
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) z[k] = c.zi[k] * x0;
    } else {
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) z[k] = 0.0;
    }
    // cooperative fetch of step-block sb: instruction k brings rows 2k (lanes 0-31) and 2k+1 (lanes 32-63)
    double r[32];
    // index range the whole wave touches in step-block sb (uniform): interior blocks take plain, branch-free loads
    auto block_lo = [&](int sb) { return ch0 * chunk - warm + (int64_t)sb * FF_SB; };
    // rows past the last chunk carry clamped dummy data and never store: they do not make a block an edge block
    const int64_t last_row = min(ch0 + 63, n_chunks - 1);
    auto block_hi = [&](int sb) { return last_row * chunk - warm + (int64_t)sb * FF_SB + FF_SB - 1; };
    auto fetch = [&](int sb) __attribute__((always_inline)) {
        const int64_t hi = block_hi(sb);
        const int64_t base = (ch0 + hrow) * chunk - warm + (int64_t)sb * FF_SB + u;      // index of (row hrow, column u)
        // forward: plain loads unless some row of this block holds indices of the left odd extension [0, FF_PAD) -- a row
        // base is a multiple of 32, so that is the row whose base is exactly 0 -- or reaches the right extension.
        // Rows still in their idle prefix (negative indices) are clamped to x[0]: their values are never used.
        const int64_t zero_row = (warm - (int64_t)sb * FF_SB) / chunk;      // row whose base index is 0, if the division is exact
        const bool has_left = (warm - (int64_t)sb * FF_SB) >= 0 && (warm - (int64_t)sb * FF_SB) % chunk == 0 &&
                              zero_row >= ch0 && zero_row <= last_row;
        if (!backward && !has_left && hi < FF_PAD + n) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int64_t i = base + (min(ch0 + hrow + 2 * k, last_row) - (ch0 + hrow)) * chunk - FF_PAD;
                r[k] = x[i < 0 ? 0 : i];
            }
        } else if (backward && hi < ne) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int64_t i = base + (min(ch0 + hrow + 2 * k, last_row) - (ch0 + hrow)) * chunk;
                r[k] = yf_in[ne - 1 - (i < 0 ? 0 : i)];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) r[k] = ff_input(x, n, yf_in, base + (int64_t)2 * k * chunk, backward);
        }
    };
    fetch(0);
    const int n_sb = n_steps / FF_SB;
    const int out_sb0 = warm / FF_SB;                 // first step-block that produces outputs
    for (int sb = 0; sb < n_sb; ++sb) {
        // (lds_barrier, not __syncthreads: the latter also waits for every outstanding global access -- i.e. for the acknowledgement of
        // the previous step-block's 32 output stores, ~7 000 cycles per step-block, as long as the recurrence itself)
        lds_barrier();                                // previous tile fully consumed (single wave: a cheap fence)
#pragma unroll
        for (int k = 0; k < 32; ++k) tin[(2 * k + hrow) * FF_TS + u] = r[k];
        if (sb + 1 < n_sb) fetch(sb + 1);
        lds_barrier();
        const int64_t ib = w0 + (int64_t)sb * FF_SB;
        const bool emit = sb >= out_sb0;              // uniform
        if (block_lo(sb) >= 0 && block_hi(sb) < ne) {
            // interior (uniform): no per-step bounds checks; inputs read 8 ahead, outputs written 8 at a time
#pragma unroll
            for (int s8 = 0; s8 < FF_SB; s8 += 8) {
                double xin[8], yo[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xin[q] = tin[lane * FF_TS + s8 + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) yo[q] = ff_step(c, xin[q], z);
                if (emit) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) tout[lane * FF_TS + s8 + q] = yo[q];
                }
            }
        } else {
            // a step-block that crosses either end of the extended signal for some lane (the first wave's warm-up,
            // the last chunk).  A lane's first real step (i = 0) sits on a multiple of 8 (warm and chunk are), so inside
            // a batch of 8 a lane is all-idle or all-active at the front end: one divergent branch per batch, no
            // per-step selects.  Only the tail (i >= ne inside a batch) needs them.
#pragma unroll 1
            for (int s8 = 0; s8 < FF_SB; s8 += 8) {
                const int64_t ia = ib + s8;
                if (ia < 0 || ia >= ne) continue;                 // per lane: nothing to do in this batch
                double xin[8], yo[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xin[q] = tin[lane * FF_TS + s8 + q];
                if (ia + 7 < ne) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) yo[q] = ff_step(c, xin[q], z);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        double zn[FF_ORD];
#pragma unroll
                        for (int k = 0; k < FF_ORD; ++k) zn[k] = z[k];
                        yo[q] = ff_step(c, xin[q], zn);
#pragma unroll
                        for (int k = 0; k < FF_ORD; ++k) z[k] = ia + q < ne ? zn[k] : z[k];
                    }
                }
                if (emit) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) tout[lane * FF_TS + s8 + q] = yo[q];
                }
            }
        }
        if (sb >= out_sb0) {                          // uniform: the last chunk / FF_SB step-blocks carry outputs
            lds_barrier();
            double yv[32];                            // read unconditionally: an LDS read inside the bounds check below is waited for
#pragma unroll                                        // before the branch closes -- 32 serial round trips per step-block, as long as the recurrence
            for (int k = 0; k < 32; ++k) yv[k] = tout[(2 * k + hrow) * FF_TS + u];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int64_t row_ch = ch0 + 2 * k + hrow;
                const int64_t i = row_ch * chunk - warm + (int64_t)sb * FF_SB + u;   // index of tile element (row, u)
                if (row_ch < n_chunks && i >= 0 && i < ne) {
                    const double y = yv[k];
                    if (!backward) {
                        yf_out[i] = y;
                    } else {
                        const int64_t jj = ne - 1 - i - FF_PAD;   // position in the un-reversed, cropped output
                        if (jj >= 0 && jj < n) out[jj] = y;
                    }
                }
            }
        }
    }
}

// double-double helpers (error-free transformations; explicit fma, the file is compiled with contraction off)
struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    const double lo = s.lo + (a.lo + b.lo);
    const double hi = s.hi + lo;
    return dd{hi, lo - (hi - s.hi)};
}
__device__ __forceinline__ dd dd_mul_d(double ghi, double glo, double x) {   // (ghi + glo) * x
    const double p = ghi * x, e = __builtin_fma(ghi, x, -p);
    return dd{p, e + glo * x};
}

// z(i0) for chunk `ch` (i0 = ch * chunk): one wave per chunk, lane j0 takes terms j0, j0 + 64, ...; four waves = four consecutive
// chunks per block, so the 160 KiB table is fetched into a CU's L1 once for four dot products (one wave per block: 328 MB of L2
// reads per pass, 102 us)
// tab: [4][FF_ORD][W]: G hi, G lo, S hi, S lo  (S[i] = A^i zi, i < W)
constexpr int FF_SW = 4;   // waves (chunks) per block
__global__ void __launch_bounds__(64 * FF_SW)
ff_state_kernel(const double *__restrict__ x, int64_t n, const double *__restrict__ yf_in, int backward, FiltCoef c, int chunk, int W,
                int64_t n_chunks, const double *__restrict__ tab, double *__restrict__ zstate) {
    const int64_t ch = (int64_t)blockIdx.x * FF_SW + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    const int64_t i0 = ch * chunk;
    const int lane = threadIdx.x & 63;
    const double x0 = ff_input(x, n, yf_in, 0, backward);
    if (ch == 0) {
        if (lane < FF_ORD) zstate[lane] = c.zi[lane] * x0;      // scipy.signal.filtfilt's start, as the chunk kernel computes it
        return;
    }
    const int nterm = (int)(i0 < W ? i0 : W);
    dd acc[FF_ORD];
#pragma unroll
    for (int k = 0; k < FF_ORD; ++k) acc[k] = dd{0.0, 0.0};
    for (int j0 = lane; j0 < nterm; j0 += 128) {                  // two terms per trip: both terms' loads are requested before either is used
        const int j1 = j0 + 64;
        const bool two = j1 < nterm;
        const int j1c = two ? j1 : j0;
        const double xa = ff_input(x, n, yf_in, i0 - 1 - j0, backward);
        double xb = ff_input(x, n, yf_in, i0 - 1 - j1c, backward);
        double ga[2][FF_ORD], gb[2][FF_ORD];
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) {
            ga[0][k] = tab[(0 * FF_ORD + k) * W + j0]; ga[1][k] = tab[(1 * FF_ORD + k) * W + j0];
            gb[0][k] = tab[(0 * FF_ORD + k) * W + j1c]; gb[1][k] = tab[(1 * FF_ORD + k) * W + j1c];
        }
        xb = two ? xb : 0.0;
#pragma unroll
        for (int k = 0; k < FF_ORD; ++k) {
            acc[k] = dd_add(acc[k], dd_mul_d(ga[0][k], ga[1][k], xa));
            acc[k] = dd_add(acc[k], dd_mul_d(gb[0][k], gb[1][k], xb));
        }
    }
#pragma unroll
    for (int k = 0; k < FF_ORD; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const dd o = dd{__shfl_xor(acc[k].hi, off, 64), __shfl_xor(acc[k].lo, off, 64)};
            acc[k] = dd_add(acc[k], o);
        }
    }
    if (lane < FF_ORD) {
        dd v = acc[0];
#pragma unroll
        for (int k = 1; k < FF_ORD; ++k) if (lane == k) v = acc[k];
        if (i0 < W) v = dd_add(v, dd_mul_d(tab[(2 * FF_ORD + lane) * W + i0], tab[(3 * FF_ORD + lane) * W + i0], x0));   // the exact start's own response
        zstate[ch * FF_ORD + lane] = v.hi + v.lo;
    }
}

}  // namespace rvc

using namespace rvc;

constexpr int FF_CHUNK = 512;
constexpr int FF_WARM = 4096;
static_assert(FF_CHUNK % 32 == 0 && FF_WARM % 32 == 0, "chunk and warm-up are whole step-blocks");

// the state impulse responses of one coefficient set, tabulated on the host in long double, split into double-double, kept in HBM
namespace {
struct FfTable {
    double key[17];
    int device = -1;
    double *dev = nullptr;
};
std::mutex g_ff_mu;
std::vector<FfTable> g_ff_tables;

int ff_table_for(const double *coef_host, const double **out) {
    int device = 0;
    RVC_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> g(g_ff_mu);
    for (const FfTable &t : g_ff_tables)
        if (t.device == device && memcmp(t.key, coef_host, sizeof(t.key)) == 0) { *out = t.dev; return 0; }
    const int W = FF_WARM;
    std::vector<double> tab((size_t)4 * FF_ORD * W);
    const double *b = coef_host, *a = coef_host + 6, *zi = coef_host + 12;
    long double gv[FF_ORD], sv[FF_ORD];
    for (int i = 0; i < FF_ORD; ++i) {
        gv[i] = (long double)b[i + 1] - (long double)a[i + 1] * (long double)b[0];   // z' = A z + B x with y = z0 + b0 x
        sv[i] = (long double)zi[i];
    }
    auto put = [&](int which, int k, int j, long double v) {
        const double hi = (double)v;
        tab[((size_t)(2 * which) * FF_ORD + k) * W + j] = hi;
        tab[((size_t)(2 * which + 1) * FF_ORD + k) * W + j] = (double)(v - (long double)hi);
    };
    auto step = [&](long double *v) {      // v <- A v: v'_i = v_{i+1} - a_{i+1} v_0, v'_4 = -a_5 v_0
        const long double v0 = v[0];
        for (int i = 0; i < FF_ORD - 1; ++i) v[i] = v[i + 1] - (long double)a[i + 1] * v0;
        v[FF_ORD - 1] = -(long double)a[FF_ORD] * v0;
    };
    for (int j = 0; j < W; ++j) {
        for (int k = 0; k < FF_ORD; ++k) { put(0, k, j, gv[k]); put(1, k, j, sv[k]); }
        step(gv);
        step(sv);
    }
    FfTable t;
    memcpy(t.key, coef_host, sizeof(t.key));
    t.device = device;
    RVC_HIP(hipMalloc((void **)&t.dev, tab.size() * sizeof(double)));
    RVC_HIP(hipMemcpy(t.dev, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    g_ff_tables.push_back(t);
    *out = t.dev;
    return 0;
}
}  // namespace

extern "C" int rvc_filtfilt_workspace_bytes(int64_t n, size_t *bytes) {
    if (!bytes || n <= FF_PAD) return fail("rvc_filtfilt_workspace_bytes: bad argument");
    const int64_t ne = n + 2 * FF_PAD;
    *bytes = align_up((size_t)ne * 8, 256) * 2 + align_up((size_t)ceil_div(ne, FF_CHUNK) * FF_ORD * 8, 256);
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
    const double *tab = nullptr;
    if (ff_table_for(coef_host, &tab)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t ne = n + 2 * FF_PAD, n_chunks = ceil_div(ne, FF_CHUNK);
    double *yf = (double *)workspace_dev;
    double *zstate = (double *)((char *)workspace_dev + align_up((size_t)ne * 8, 256) * 2);
    const dim3 grid((unsigned)ceil_div(n_chunks, 64)), block(64);
    for (int backward = 0; backward < 2; ++backward) {
        hipLaunchKernelGGL(ff_state_kernel, dim3((unsigned)ceil_div(n_chunks, FF_SW)), dim3(64 * FF_SW), 0, stream, x_dev, n, yf, backward, c, FF_CHUNK, FF_WARM, n_chunks, tab, zstate);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(ff_chunk_kernel, grid, block, 0, stream, x_dev, n, yf, backward, c, FF_CHUNK, 0, n_chunks, yf, y_dev, zstate);
        RVC_LAUNCH_CHECK();
    }
    return 0;
}
