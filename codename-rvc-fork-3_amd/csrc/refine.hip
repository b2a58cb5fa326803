// RefineGAN vocoder (rvc/lib/algorithm/generators/refinegan.py:266-405), BASELINE cfg 5.
//
//   f0 --linear interp to T*upp--> SineGenerator (two sample-rate cumsums with wrap compensation, :220-243)
//      --Linear(1,1)+tanh--> har_source [B,1,L]
//   x0 = cat( mel_conv(z) + cond(g),  linear-downsample(pre_conv(har_source)) )                 [B,512,T]
//   4 x { leaky(0.2); Upsample(rate, linear); cat with downsample_blocks[i](har_source);
//         ParallelResBlock = input_conv(k7) -> mean of 3 x [AdaIN(noise) -> ResBlock(k, d=1,3,5) -> AdaIN(noise)] }
//   leaky(0.2); conv_post; tanh
//
// The dense convs (mel_conv, input_conv, the 72 ResBlock convs: 3.9 TFLOP per 30 s utterance) run on the same
// fp32 MFMA implicit-GEMM kernel as the NSF vocoder (conv.hip).  Specific to this file:
//   * the two 1.5 M-sample cumsums: three-phase block scans in float64 (torch's CPU cumsum accumulates float32
//     inputs in double and rounds each prefix to float);
//   * pre_conv + linear down-sampling fused: the interpolation only ever reads two positions per frame, so the
//     256 x 7 conv is evaluated at those 2 T positions instead of at all T*upp samples;
//   * AdaIN (x + randn * w -> leaky) as an element-wise kernel whose second instance also carries the
//     running mean of the three branches.
#include "decoder.h"

namespace rvc {

#pragma clang fp contract(off)

constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_PER_THREAD;

// PyTorch's linear interpolation index/weight rule (align_corners = False), float math
__device__ __forceinline__ void lin_idx(float scale, int64_t dst, int64_t in_size, int64_t &i0, int64_t &i1, float &w0, float &w1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int64_t)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    w1 = src - (float)i0;
    if (w1 < 0.f) w1 = 0.f;
    if (w1 > 1.f) w1 = 1.f;
    w0 = 1.f - w1;
}

// refinegan.py:370-372 + :226: f0 -> sample rate (linear), rad = (f0 / sr) % 1
__global__ void __launch_bounds__(256)
rg_f0_rad_kernel(const float *__restrict__ f0, int64_t T, int64_t L, float scale, float sr, float *__restrict__ f0up,
                 float *__restrict__ rad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (t >= L) return;
    int64_t i0, i1;
    float w0, w1;
    lin_idx(scale, t, T, i0, i1, w0, w1);
    const float v = w0 * f0[b * T + i0] + w1 * f0[b * T + i1];
    f0up[b * L + t] = v;
    const float q = v / sr;
    rad[b * L + t] = q - floorf(q);
}

// ---- float64 inclusive scan of a float array, per batch row ------------------------------------------------
__global__ void __launch_bounds__(256)
scan_partials_kernel(const float *__restrict__ in, int64_t L, int64_t n_blk, double *__restrict__ partial) {
    __shared__ double red[256];
    const int64_t b = blockIdx.y, blk = blockIdx.x;
    const int64_t base = blk * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i)
        if (base + i < L) s += (double)in[b * L + base + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[b * n_blk + blk] = red[0];
}

__global__ void scan_offsets_kernel(double *__restrict__ partial, int64_t n_blk) {
    if (threadIdx.x != 0) return;
    double *p = partial + (int64_t)blockIdx.x * n_blk;
    double run = 0.0;
    for (int64_t i = 0; i < n_blk; ++i) {
        const double v = p[i];
        p[i] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(256)
scan_apply_kernel(const float *__restrict__ in, int64_t L, int64_t n_blk, const double *__restrict__ partial,
                  double *__restrict__ out) {
    __shared__ double sc[256];
    const int64_t b = blockIdx.y, blk = blockIdx.x;
    const int64_t base = blk * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
    double v[SCAN_PER_THREAD];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        v[i] = (base + i < L) ? (double)in[b * L + base + i] : 0.0;
        s += v[i];
    }
    sc[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan of the thread sums
        const double add = threadIdx.x >= o ? sc[threadIdx.x - o] : 0.0;
        __syncthreads();
        sc[threadIdx.x] += add;
        __syncthreads();
    }
    double run = partial[b * n_blk + blk] + (threadIdx.x ? sc[threadIdx.x - 1] : 0.0);
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        run += v[i];
        if (base + i < L) out[b * L + base + i] = run;
    }
}

// refinegan.py:236-239: wrap detection on tmp = float(cumsum) % 1 -> term = rad + shift
__global__ void __launch_bounds__(256)
rg_term_kernel(const double *__restrict__ S1, const float *__restrict__ rad, int64_t L, float *__restrict__ term) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (t >= L) return;
    float shift = 0.f;
    if (t > 0) {
        const float c1 = (float)S1[b * L + t], c0 = (float)S1[b * L + t - 1];
        const float t1 = c1 - floorf(c1), t0 = c0 - floorf(c0);
        if (t1 - t0 < 0.f) shift = -1.f;
    }
    term[b * L + t] = rad[b * L + t] + shift;
}

// refinegan.py:241, 253-263: sine, uv mask, additive noise, Linear(1 -> 1, no bias), tanh
__global__ void __launch_bounds__(256)
rg_source_kernel(const double *__restrict__ S2, const float *__restrict__ f0up, const float *__restrict__ randn, int64_t L,
                 float merge_w, float *__restrict__ har) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (t >= L) return;
    const float c = (float)S2[b * L + t];
    const float sine = sinf((c * 2.f) * 3.14159265358979323846f) * 0.1f;
    const float uv = f0up[b * L + t] > 0.f ? 1.f : 0.f;
    const float amp = uv * 0.003f + ((1.f - uv) * 0.1f) / 3.f;
    const float v = sine * uv + amp * randn[b * L + t];
    har[b * L + t] = tanhf(v * merge_w);
}

// refinegan.py:375-376: x = interpolate(pre_conv(har), size=T, linear) evaluated only where it is read
__global__ void __launch_bounds__(256)
rg_preconv_down_kernel(const float *__restrict__ har, int64_t L, int64_t T, float scale, const float *__restrict__ w,
                       const float *__restrict__ bias, int n_ch, float *__restrict__ out, int64_t out_bstride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (i >= T) return;
    int64_t i0, i1;
    float w0, w1;
    lin_idx(scale, i, L, i0, i1, w0, w1);
    const float *h = har + b * L;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int64_t p0 = i0 + k - 3, p1 = i1 + k - 3;
        const float wk = w[c * 7 + k];
        a0 += wk * ((p0 >= 0 && p0 < L) ? h[p0] : 0.f);
        a1 += wk * ((p1 >= 0 && p1 < L) ? h[p1] : 0.f);
    }
    a0 += bias[c];
    a1 += bias[c];
    out[b * out_bstride + (int64_t)c * T + i] = w0 * a0 + w1 * a1;
}

// refinegan.py:390,397: leaky(0.2) then nn.Upsample(scale_factor=rate, mode="linear")
__global__ void __launch_bounds__(256)
rg_upsample_kernel(const float *__restrict__ x, int64_t x_bstride, int64_t l_in, int64_t l_out, float scale, float slope,
                   float *__restrict__ out, int64_t out_bstride) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (j >= l_out) return;
    int64_t i0, i1;
    float w0, w1;
    lin_idx(scale, j, l_in, i0, i1, w0, w1);
    const float *xr = x + b * x_bstride + (int64_t)c * l_in;
    out[b * out_bstride + (int64_t)c * l_out + j] = w0 * lrelu(xr[i0], slope) + w1 * lrelu(xr[i1], slope);
}

// refinegan.py:398: downsample_blocks[i](har_source): Conv1d(1 -> C, k, stride, pad)
__global__ void __launch_bounds__(256)
rg_down_kernel(const float *__restrict__ har, int64_t L, int stride, int pad, int K, const float *__restrict__ w,
               const float *__restrict__ bias, int64_t l_out, float *__restrict__ out, int64_t out_bstride) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (q >= l_out) return;
    const float *h = har + b * L;
    const float *wk = w + (int64_t)co * K;
    float acc = 0.f;
    const int64_t p0 = q * stride - pad;
    if (p0 >= 0 && p0 + K <= L) {
        // interior: unconditional loads, 8 in flight (a load inside a bounds check is waited for before the next one)
        int k = 0;
        for (; k + 8 <= K; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = h[p0 + k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += wk[k + u] * v[u];
        }
        for (; k < K; ++k) acc += wk[k] * h[p0 + k];
    } else {
        for (int k = 0; k < K; ++k) {
            const int64_t p = p0 + k;
            acc += wk[k] * ((p >= 0 && p < L) ? h[p] : 0.f);
        }
    }
    out[b * out_bstride + (int64_t)co * l_out + q] = acc + bias[co];
}

// refinegan.py:110-113 AdaIN; the second instance of a branch also accumulates torch.stack(...).mean(0) (:170)
__global__ void __launch_bounds__(256)
rg_adain_kernel(const float *__restrict__ x, const float *__restrict__ noise, const float *__restrict__ w, int C, int64_t L,
                float slope, const float *accin, float divisor, float *out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (t >= L) return;
    const int64_t o = (b * C + c) * L + t;
    float v = lrelu(x[o] + noise[o] * w[c], slope);
    if (accin) v = accin[o] + v;
    out[o] = v / divisor;
}

static int upload_vec(const rvc_decoder *d, const std::string &name, std::vector<int64_t> shape, DevBuf *out) {
    const HostTensor *t;
    if (need(d, name, &t, shape)) return 1;
    return out->upload(t->data);
}

int refine_finalize(rvc_decoder *d) {
    const rvc_decoder_config &c = d->cfg;
    const int c0 = c.upsample_initial_channel;
    const HostTensor *t;
    if (need(d, "m_source.merge.0.weight", &t, {1, 1})) return 1;
    d->merge_w = t->data[0];
    if (upload_vec(d, "pre_conv.weight", {c0 / 2, 1, 7}, &d->pre_w)) return 1;
    if (upload_vec(d, "pre_conv.bias", {c0 / 2}, &d->pre_b)) return 1;
    if (build_conv(d, "mel_conv", c0 / 2, c.in_channels, 7, true, &d->mel)) return 1;
    if (upload_vec(d, "cond.weight", {c0 / 2, c.gin_channels, 1}, &d->cond_w)) return 1;
    if (upload_vec(d, "cond.bias", {c0 / 2}, &d->cond_b)) return 1;
    d->rstages.clear();
    d->rstages.resize(c.n_ups);
    int ch = c0;
    for (int i = 0; i < c.n_ups; ++i) {
        RefineStage &s = d->rstages[i];
        s.ch_in = ch;
        s.ch_out = ch / 2;
        s.rate = c.upsample_rates[i];
        s.down_c = c0 >> (i + 2);
        int stride = 1;
        for (int j = i + 1; j < c.n_ups; ++j) stride *= c.upsample_rates[j];
        s.down_stride = stride;
        s.down_k = stride == 1 ? 1 : stride * 2 - stride % 2;      // refinegan.py:321-323
        s.down_pad = stride == 1 ? 0 : (s.down_k - stride) / 2;
        if ((s.ch_in + s.down_c) % 8 || s.ch_out % 32) return fail("refinegan: stage %d channel counts unsupported", i);
        const std::string db = "downsample_blocks." + std::to_string(i);
        if (upload_vec(d, db + ".weight", {s.down_c, 1, s.down_k}, &s.down_w)) return 1;
        if (upload_vec(d, db + ".bias", {s.down_c}, &s.down_b)) return 1;
        const std::string ub = "upsample_conv_blocks." + std::to_string(i);
        if (build_conv(d, ub + ".input_conv", s.ch_out, s.ch_in + s.down_c, 7, true, &s.input_conv)) return 1;
        const int nb = c.n_res_kernels * c.n_res_dilations;
        s.c1 = std::vector<ConvW>(nb);
        s.c2 = std::vector<ConvW>(nb);
        s.pair = std::vector<DevBuf16>(nb);
        for (int m = 0; m < c.n_res_kernels; ++m) {
            const std::string bm = ub + ".blocks." + std::to_string(m);
            if (upload_vec(d, bm + ".0.weight", {s.ch_out}, &s.adain1[m])) return 1;
            if (upload_vec(d, bm + ".2.weight", {s.ch_out}, &s.adain2[m])) return 1;
            for (int j = 0; j < c.n_res_dilations; ++j) {
                const int k = c.res_kernel_sizes[m];
                if (build_conv(d, bm + ".1.convs1." + std::to_string(j), s.ch_out, s.ch_out, k, true, &s.c1[m * c.n_res_dilations + j])) return 1;
                if (build_conv(d, bm + ".1.convs2." + std::to_string(j), s.ch_out, s.ch_out, k, true, &s.c2[m * c.n_res_dilations + j])) return 1;
                // the narrow stages' (conv, conv) pairs as ONE launch on the bf16 matrix cores (resblock_bf.hip, K3f): the ResBlock body
                // is the NSF vocoder's (refinegan.py:59-85 = residuals.py:75-86) with slope 0.2.  Since round 6 this takes the layers
                // that ran wino.hip's fp32 Winograd kernel (64 channels x 3 taps, the 32-channel stage) -- the kernel that returns wrong
                // words next to a co-resident bf16-matrix workgroup (profiles/r05_mfma_cohabitation.txt) is off every default path.
                if (resblock_bf_enabled() && resblock_bf_supported(s.ch_out, k, 1) && resblock_bf_preferred(s.ch_out, k, 3)) {
                    const HostTensor *w1, *w2;
                    if (need(d, bm + ".1.convs1." + std::to_string(j) + ".weight", &w1, {s.ch_out, s.ch_out, k}) ||
                        need(d, bm + ".1.convs2." + std::to_string(j) + ".weight", &w2, {s.ch_out, s.ch_out, k}))
                        return 1;
                    std::vector<uint16_t> frags;
                    resblock_bf_pack_host(w1->data.data(), w2->data.data(), s.ch_out, k, &frags, 3);
                    if (s.pair[m * c.n_res_dilations + j].upload(frags)) return 1;
                }
            }
        }
        ch = s.ch_out;
    }
    d->post_cin = ch;
    if (need(d, "conv_post.weight", &t, {1, ch, 7})) return 1;
    if (d->post_w.upload(t->data)) return 1;
    d->post_b = 0.f;
    return 0;
}

namespace {
struct RLayout {
    size_t har, f0up, rad, term, S, partial, biasp, x0, cat, act[5], total;
    int64_t n_blk;
};
RLayout refine_layout(const rvc_decoder *d, int batch, int64_t T) {
    RLayout l;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const int64_t L = T * d->upp;
    const int c0 = d->cfg.upsample_initial_channel;
    l.n_blk = ceil_div(L, SCAN_BLOCK);
    l.har = take((size_t)batch * L * 4);
    l.f0up = take((size_t)batch * L * 4);
    l.rad = take((size_t)batch * L * 4);
    l.term = take((size_t)batch * L * 4);
    l.S = take((size_t)batch * L * 8);
    l.partial = take((size_t)batch * l.n_blk * 8);
    l.biasp = take((size_t)batch * (c0 / 2) * 4);
    l.x0 = take((size_t)batch * c0 * T * 4);
    int64_t len = T, max_cat = 0, max_act = 0;
    for (const RefineStage &s : d->rstages) {
        len *= s.rate;
        max_cat = std::max<int64_t>(max_cat, (int64_t)(s.ch_in + s.down_c) * len);
        max_act = std::max<int64_t>(max_act, (int64_t)s.ch_out * len);
    }
    l.cat = take((size_t)batch * max_cat * 4);
    for (int i = 0; i < 5; ++i) l.act[i] = take((size_t)batch * max_act * 4);
    l.total = off;
    return l;
}

int run_scan(const float *in, int batch, int64_t L, int64_t n_blk, double *partial, double *out, hipStream_t stream) {
    hipLaunchKernelGGL(scan_partials_kernel, dim3((unsigned)n_blk, batch), dim3(256), 0, stream, in, L, n_blk, partial);
    RVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(batch), dim3(64), 0, stream, partial, n_blk);
    RVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_blk, batch), dim3(256), 0, stream, in, L, n_blk, partial, out);
    RVC_LAUNCH_CHECK();
    return 0;
}
}  // namespace

size_t refine_workspace_bytes(const rvc_decoder *d, int batch, int64_t T) { return refine_layout(d, batch, T).total; }

int refine_forward(rvc_decoder *d, const float *z_dev, const float *f0_dev, const float *g_dev, const rvc_decoder_noise *noise,
                   int batch, int64_t T, float *out_dev, void *workspace_dev, size_t workspace_bytes, hipStream_t stream) {
    const rvc_decoder_config &c = d->cfg;
    if (!noise->adain_randn_dev) return fail("rvc_decoder_forward: adain_randn_dev is required for the RefineGAN decoder");
    const RLayout lay = refine_layout(d, batch, T);
    if (workspace_bytes < lay.total) return fail("rvc_decoder_forward: workspace too small (%zu < %zu)", workspace_bytes, lay.total);
    char *ws = (char *)workspace_dev;
    float *har = (float *)(ws + lay.har), *f0up = (float *)(ws + lay.f0up), *rad = (float *)(ws + lay.rad);
    float *term = (float *)(ws + lay.term), *biasp = (float *)(ws + lay.biasp), *x0 = (float *)(ws + lay.x0);
    float *cat = (float *)(ws + lay.cat);
    double *S = (double *)(ws + lay.S), *partial = (double *)(ws + lay.partial);
    float *act[5];
    for (int i = 0; i < 5; ++i) act[i] = (float *)(ws + lay.act[i]);
    const int64_t L = T * d->upp;
    const int c0 = c.upsample_initial_channel, half = c0 / 2;
    const dim3 gl((unsigned)ceil_div(L, 256), batch);

    // ---- source ----
    hipLaunchKernelGGL(rg_f0_rad_kernel, gl, dim3(256), 0, stream, f0_dev, T, L, (float)T / (float)L, (float)c.sample_rate, f0up, rad);
    RVC_LAUNCH_CHECK();
    if (run_scan(rad, batch, L, lay.n_blk, partial, S, stream)) return 1;
    hipLaunchKernelGGL(rg_term_kernel, gl, dim3(256), 0, stream, S, rad, L, term);
    RVC_LAUNCH_CHECK();
    if (run_scan(term, batch, L, lay.n_blk, partial, S, stream)) return 1;
    hipLaunchKernelGGL(rg_source_kernel, gl, dim3(256), 0, stream, S, f0up, noise->src_randn_dev, L, d->merge_w, har);
    RVC_LAUNCH_CHECK();
    if (d->tap_stage == -1 && d->tap_dev)
        RVC_HIP(hipMemcpyAsync(d->tap_dev, har, (size_t)batch * L * 4, hipMemcpyDeviceToDevice, stream));

    // ---- x0 = cat(mel_conv(z) + cond(g), down(pre_conv(har))) ----
    if (launch_cond_bias(d->mel.b.p, d->cond_w.p, d->cond_b.p, g_dev, batch, c.gin_channels, half, biasp, stream)) return 1;
    {
        ConvParams p;
        p.x1 = z_dev; p.c1 = c.in_channels; p.slope1 = 1.f; p.x1_bstride = (int64_t)c.in_channels * T; p.l_in = T;
        p.w = d->mel.w.p; p.bias = biasp; p.bias_bstride = half;
        p.y = x0; p.y_bstride = (int64_t)c0 * T; p.m_total = half; p.c_out = half; p.n_cols = T; p.l_out = T;
        p.kw = 7; p.dil = 1; p.padl = 3; p.batch = batch;
        if (launch_conv(p, stream)) return 1;
    }
    hipLaunchKernelGGL(rg_preconv_down_kernel, dim3((unsigned)ceil_div(T, 256), half, batch), dim3(256), 0, stream, har, L, T,
                       (float)L / (float)T, d->pre_w.p, d->pre_b.p, half, x0 + (int64_t)half * T, (int64_t)c0 * T);
    RVC_LAUNCH_CHECK();

    // ---- stages ----
    const float slope = 0.2f;
    const float *xprev = x0;
    int64_t len = T;
    const float *nz = noise->adain_randn_dev;
    const int nd = c.n_res_dilations, nk = c.n_res_kernels;
    float *XIN = act[0], *A = act[1], *Y = act[2], *T1 = act[3], *SUM = act[4];
    for (int i = 0; i < c.n_ups; ++i) {
        RefineStage &s = d->rstages[i];
        const int64_t lo = len * s.rate;
        const int ctot = s.ch_in + s.down_c;
        const int64_t cat_bs = (int64_t)ctot * lo;
        hipLaunchKernelGGL(rg_upsample_kernel, dim3((unsigned)ceil_div(lo, 256), s.ch_in, batch), dim3(256), 0, stream, xprev,
                           (int64_t)s.ch_in * len, len, lo, (float)(1.0 / (double)s.rate), slope, cat, cat_bs);
        RVC_LAUNCH_CHECK();
        hipLaunchKernelGGL(rg_down_kernel, dim3((unsigned)ceil_div(lo, 256), s.down_c, batch), dim3(256), 0, stream, har, L,
                           s.down_stride, s.down_pad, s.down_k, s.down_w.p, s.down_b.p, lo, cat + (int64_t)s.ch_in * lo, cat_bs);
        RVC_LAUNCH_CHECK();
        const int64_t bs = (int64_t)s.ch_out * lo;
        {
            ConvParams p;
            p.x1 = cat; p.c1 = ctot; p.slope1 = 1.f; p.x1_bstride = cat_bs; p.l_in = lo;
            p.w = s.input_conv.w.p; p.bias = s.input_conv.b.p; p.w_winobf = s.input_conv.wx.p;   // bf16x3 Winograd where c_out % 64 == 0
            p.y = XIN; p.y_bstride = bs; p.m_total = s.ch_out; p.c_out = s.ch_out; p.n_cols = lo; p.l_out = lo;
            p.kw = 7; p.dil = 1; p.padl = 3; p.batch = batch;
            if (launch_conv(p, stream)) return 1;
        }
        const dim3 ge((unsigned)ceil_div(lo, 256), s.ch_out, batch);
        for (int m = 0; m < nk; ++m) {
            const int k = c.res_kernel_sizes[m];
            hipLaunchKernelGGL(rg_adain_kernel, ge, dim3(256), 0, stream, XIN, nz, s.adain1[m].p, s.ch_out, lo, slope,
                               (const float *)nullptr, 1.f, A);
            RVC_LAUNCH_CHECK();
            nz += (size_t)batch * bs;
            const float *xin = A;
            const bool pairs = s.pair[m * nd].p && resblock_bf_fits(s.ch_out, lo) && nd % 2 == 1;
            for (int j = 0; pairs && j < nd; ++j) {   // A -> Y -> T1 -> Y ... (no in-place: blocks read neighbours' columns; an odd count ends in Y)
                float *yout = (j % 2 == 1) ? T1 : Y;
                if (launch_resblock_bf(xin, s.pair[m * nd + j].p, s.c1[m * nd + j].b.p, s.c2[m * nd + j].b.p, nullptr, yout, batch, s.ch_out, lo,
                                       k, c.res_dilations[j], slope, 1.f, stream, 3))
                    return 1;
                xin = yout;
            }
            const bool fused = !pairs && resblock_layer_supported(s.ch_out, k) && nd == 3;
            for (int j = 0; fused && j < nd; ++j) {   // A -> Y -> T1 -> Y (no in-place: blocks read neighbours' columns)
                float *yout = (j == 1) ? T1 : Y;
                if (launch_resblock_layer(xin, s.c1[m * nd + j].w.p, s.c1[m * nd + j].b.p, s.c2[m * nd + j].w.p, s.c2[m * nd + j].b.p,
                                          nullptr, yout, batch, s.ch_out, lo, k, c.res_dilations[j], slope, 1.f, stream))
                    return 1;
                xin = yout;
            }
            for (int j = 0; !fused && !pairs && j < nd; ++j) {
                const int dil = c.res_dilations[j];
                ConvParams p;
                p.x1 = xin; p.c1 = s.ch_out; p.slope1 = slope; p.x1_bstride = bs; p.l_in = lo;
                p.w = s.c1[m * nd + j].w.p; p.bias = s.c1[m * nd + j].b.p; p.w_wino = s.c1[m * nd + j].wu.p; p.w_winobf = s.c1[m * nd + j].wx.p;
                p.y = T1; p.y_bstride = bs; p.m_total = s.ch_out; p.c_out = s.ch_out; p.n_cols = lo; p.l_out = lo;
                p.kw = k; p.dil = dil; p.padl = (k - 1) / 2 * dil; p.batch = batch;
                if (launch_conv(p, stream)) return 1;
                ConvParams q;
                q.x1 = T1; q.c1 = s.ch_out; q.slope1 = slope; q.x1_bstride = bs; q.l_in = lo;
                q.w = s.c2[m * nd + j].w.p; q.bias = s.c2[m * nd + j].b.p; q.w_wino = s.c2[m * nd + j].wu.p; q.w_winobf = s.c2[m * nd + j].wx.p;
                q.res = xin; q.y = Y;
                q.y_bstride = bs; q.m_total = s.ch_out; q.c_out = s.ch_out; q.n_cols = lo; q.l_out = lo;
                q.kw = k; q.dil = 1; q.padl = (k - 1) / 2; q.batch = batch;
                if (launch_conv(q, stream)) return 1;
                xin = Y;
            }
            hipLaunchKernelGGL(rg_adain_kernel, ge, dim3(256), 0, stream, Y, nz, s.adain2[m].p, s.ch_out, lo, slope,
                               m > 0 ? (const float *)SUM : (const float *)nullptr, m + 1 == nk ? (float)nk : 1.f, SUM);
            RVC_LAUNCH_CHECK();
            nz += (size_t)batch * bs;
        }
        if (d->tap_stage == i && d->tap_dev)
            RVC_HIP(hipMemcpyAsync(d->tap_dev, SUM, (size_t)batch * bs * 4, hipMemcpyDeviceToDevice, stream));
        // the next stage reads SUM as xprev while writing cat/XIN/A/Y/T1, then overwrites SUM only in its last kernel:
        // keep SUM distinct by swapping it with XIN (both are free at that point)
        xprev = SUM;
        std::swap(SUM, XIN);
        len = lo;
    }
    if (len != L) return fail("refinegan: internal length mismatch");
    return launch_conv_post(xprev, d->post_w.p, 0.f, batch, d->post_cin, L, slope, out_dev, stream);
}

}  // namespace rvc
