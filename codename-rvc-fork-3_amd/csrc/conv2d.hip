// K10 -- 3x3 / 1x1 conv2d of RMVPE's U-Net (RMVPE.py:13-287: ConvBlockRes and the shortcut convs) as an fp32 implicit GEMM
// on the matrix cores, with the folded BatchNorm bias, the ReLU and the block's residual in the epilogue.
//
// The U-Net runs ~117 3x3 convs of 1.77 GFLOP each over [C][T][128 >> level] maps whose channel count doubles as the map
// quarters (32 x 1616 x 64 ... 512 x 101 x 4).  A library conv spends ~42 us on every one of them whatever the level; the
// GEMM view is  y[co][p] = sum_{tap, ci} W[tap][ci][co] * x[ci][p + off(tap)]  with p a pixel, K = 9 C_in.
//
// Block = 4 waves, each a 32 channel x 32 pixel tile (one accumulator); the block's pixels are TH = BN / W whole rows of
// the map, so a 32-lane operand reads 32 consecutive floats of the zero-padded LDS patch [ci][TH + 2][W + 2] apart from a
// +2 step at each row end.  K is walked in chunks of 8 input channels: the patch goes HBM -> registers -> LDS (zero
// padding applied on the way), the tap slab [9][8][BM] by LDS-DMA; double-buffered, one LDS-only barrier per chunk
// (common.h).  Deep levels have few pixels (404 at 512 channels: 208 wave tiles for 1024 SIMDs), so the launch splits K
// over gridDim.z and a second kernel sums the partials in a fixed order (deterministic -- no atomics) and applies the
// epilogue.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace rvc {

struct Conv2dParams {
    const float *x = nullptr;        // [batch][c_in][H][W]
    const float *w = nullptr;        // [taps][c_in][m_pad]
    const float *bias = nullptr;     // [c_out] or null
    const float *res = nullptr;      // [batch][c_out][H][W] or null: added AFTER the activation (the block's skip path)
    float *y = nullptr;              // [batch][c_out][H][W]
    float *partial = nullptr;        // [split][batch][c_out][H * W] when split > 1
    int c_in = 0, c_out = 0, m_pad = 0, H = 0, W = 0;
    int taps = 9;                    // 9: 3x3, padding 1;  1: 1x1
    int relu = 0;
    int split = 1, chunks_per_split = 0;
    int batch = 1;
    int debug = 0;                   // experiments (RVC_C2_DEBUG): 1 no input re-staging, 2 no tap DMA in the loop, 4 no barrier, 8 no epilogue stores
};

constexpr int C2_CIC = 8;
constexpr int C2_WMAX = 128;
// floats of one channel's zero-padded patch, the largest over row lengths W = 4 .. min(BN, 128): (BN / W + 2 pad)(W + 2 pad)
constexpr int c2_patch_max(int bn, int pad) {
    int best = 0;
    for (int w = 4; w <= C2_WMAX && w <= bn; w *= 2) {
        const int v = (bn / w + 2 * pad) * (w + 2 * pad);
        best = v > best ? v : best;
    }
    return best;
}
constexpr int C2_RSRC = 0x00020000;
typedef void __attribute__((address_space(3))) *c2_lptr_t;

template <int WM, int WN, int NT, int TAPS>
__global__ void __launch_bounds__(256) conv2d_mfma_kernel(const Conv2dParams p) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTH = 256, NW = 4;   // a wave: 32 channels x 32 NT pixels
    static_assert(WM * WN == NW, "four waves");
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    constexpr int WROWS = TAPS * C2_CIC;                  // weight rows of BM floats per chunk
    constexpr int WPIECES = (WROWS * BM * 4 + 1023) / 1024;   // 1 KiB DMA pieces per chunk
    constexpr int WPW = (WPIECES + NW - 1) / NW;
    constexpr int WBUF = WPIECES * 256;                   // floats per weight buffer
    constexpr int PMAX = c2_patch_max(BN, PAD);           // largest patch of one channel over the supported row lengths
    constexpr int XBUF = C2_CIC * PMAX;
    constexpr int NJ = (PMAX + NTH - 1) / NTH;            // staged positions per thread per channel
    constexpr int XN = C2_CIC * NJ;

    extern __shared__ __attribute__((aligned(16))) float c2_smem[];
    float *ws = c2_smem;                  // [3][WBUF]
    float *xs = c2_smem + 3 * WBUF;       // [2][XBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < NW);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int H = p.H, W = p.W, c_in = p.c_in;
    const int th = BN / W;                                // rows of the map per block
    const int PR = th + 2 * PAD, PC = W + 2 * PAD, PATCH = PR * PC;
    const int n_mblk = p.m_pad / BM;
    const int t0 = (blockIdx.x / n_mblk) * th;
    const int m0 = (blockIdx.x % n_mblk) * BM;            // channel blocks of one pixel tile are neighbours: the patch comes from L2
    const int b = blockIdx.y;
    const int sp = blockIdx.z;
    const int HW = H * W;
    const float *px = p.x + (int64_t)b * c_in * HW;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)px, 0, c_in * HW * 4, C2_RSRC);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, TAPS * c_in * p.m_pad * 4, C2_RSRC);

    // ---- staging plan: position pos = tid + j * 256 of the patch, the same for each of the chunk's 8 channels ------------
    unsigned goff[NJ];    // byte offset of the sample inside a channel plane
    unsigned inb = 0;     // bit j: inside the map (outside: the conv's zero padding, or past the patch)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pos = tid + j * NTH;
        const int pc0 = pos < PATCH ? pos : 0;
        const int pr = pc0 / PC, pc = pc0 - pr * PC;
        const int t = t0 - PAD + pr, w = pc - PAD;
        const bool ok = pos < PATCH && t >= 0 && t < H && w >= 0 && w < W;
        goff[j] = ok ? (unsigned)((t * W + w) * 4) : 0u;
        if (ok) inb |= 1u << j;
    }
    unsigned woff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        int n = wave + NW * i;
        if (n >= WPIECES) n = WPIECES - 1;                           // every wave issues WPW pieces (the vmcnt bookkeeping below
                                                                     // counts them); surplus ones repeat the last piece
        const int o = n * 1024 + 16 * lane;                         // byte inside the chunk's [WROWS][BM] slab
        int row = o / (BM * 4);
        const int within = o - row * (BM * 4);
        if (row >= WROWS) row = WROWS - 1;
        const int tap = row / C2_CIC, ci = row - tap * C2_CIC;
        woff[i] = (unsigned)(((tap * c_in + ci) * p.m_pad) * 4 + within);
    }
    float xr[XN];
    auto load_x = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int ci = 0; ci < C2_CIC; ++ci)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                xr[ci * NJ + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[j], (c * C2_CIC + ci) * HW * 4, 0));
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ci = 0; ci < C2_CIC; ++ci)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int pos = tid + j * NTH;
                if (pos < PATCH) xs[buf * XBUF + ci * PATCH + pos] = ((inb >> j) & 1) ? xr[ci * NJ + j] : 0.f;
            }
    };
    auto dma_w = [&](int buf, int c) __attribute__((always_inline)) {
        const int s0 = (c * C2_CIC * p.m_pad + m0) * 4;
        char *dst = reinterpret_cast<char *>(ws + buf * WBUF);
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            int n = wave + NW * i;
            if (n >= WPIECES) n = WPIECES - 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (c2_lptr_t)(dst + n * 1024), 16, (int)woff[i], s0, 0, 0);
        }
    };

    const int c_begin = sp * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > c_in / C2_CIC) c_end = c_in / C2_CIC;
    const int n_chunks = c_end - c_begin;

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // this lane's pixels (block-relative) and their offsets inside a channel's patch
    int pn[NT], laneoff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        pn[n] = (wn * NT + n) * 32 + l31;
        const int prow = pn[n] / W;
        laneoff[n] = prow * PC + (pn[n] - prow * W);
    }

    // Pipeline (a chunk is 36 NT matrix instructions per wave, 1-2 us): taps by DMA two chunks ahead into three buffers, input
    // rows in registers two chunks ahead, in LDS one chunk ahead.  Memory operations retire in order, so "all but this
    // wave's newest WPW DMA pieces" is exactly "everything chunk k + 1 needs".
    if (n_chunks > 0) {
        load_x(c_begin);
        dma_w(0, c_begin);
        if (n_chunks > 1) dma_w(1, c_begin + 1);
        store_x(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (n_chunks > 1) load_x(c_begin + 1);
        lds_barrier();
    }
    for (int k = 0; k < n_chunks; ++k) {
        const int c = c_begin + k;
        const int wb = k % 3, xbuf = k & 1;
        // buffer (k + 2) % 3 was last read in iteration k - 1; every wave has passed that iteration's barrier
        const bool dbg_x = p.debug & 1, dbg_w = p.debug & 2;
        if (k + 2 < n_chunks && !dbg_w) dma_w((k + 2) % 3, c + 2);
        const float *wa = ws + wb * WBUF + half * BM + wm * 32 + l31;
        const float *xb = xs + xbuf * XBUF + half * PATCH;
        constexpr int STEPS = TAPS * (C2_CIC / 2);
        float a[2], bb[2][NT];
        auto frag = [&](int s, float &av, float (&bv)[NT]) __attribute__((always_inline)) {
            const int tap = s / (C2_CIC / 2), kk = s - tap * (C2_CIC / 2);
            av = wa[(tap * C2_CIC + 2 * kk) * BM];
            const int dh = TAPS == 9 ? tap / 3 : 0, dw = TAPS == 9 ? tap - 3 * (tap / 3) : 0;
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = xb[(2 * kk) * PATCH + dh * PC + dw + laneoff[n]];
        };
        frag(0, a[0], bb[0]);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS) frag(st + 1, a[(st + 1) & 1], bb[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32(a[st & 1], bb[st & 1][n], acc[n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k + 1 < n_chunks) {
            if (!dbg_x) store_x(xbuf ^ 1);                          // chunk k + 1's rows (the compiler waits for those loads)
            if (k + 2 < n_chunks && !dbg_w) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");   // ... and its taps: only chunk k + 2's pieces may be pending
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (k + 2 < n_chunks && !dbg_x) load_x(c + 2);
            if (!(p.debug & 4)) lds_barrier();
        }
    }

    // ---- epilogue: pixel pn of the block is element t0 * W + pn of the [H][W] plane ----------------------------------
    const int co0 = m0 + wm * 32 + 4 * half;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t pix = (int64_t)t0 * W + pn[n];
        if (pix >= HW) continue;
        if ((p.debug & 8) && acc[n][0] != 12345.f) continue;
        if (p.split > 1) {
            float *dst = p.partial + ((int64_t)(sp * p.batch + b) * p.c_out) * HW + pix;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2);
                if (co < p.c_out) dst[(int64_t)co * HW] = acc[n][r];
            }
            continue;
        }
        const float *res = p.res ? p.res + (int64_t)b * p.c_out * HW + pix : nullptr;
        float *y = p.y + (int64_t)b * p.c_out * HW + pix;
        float rv[16];
        if (res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2);
                rv[r] = co < p.c_out ? res[(int64_t)co * HW] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2);
            if (co >= p.c_out) continue;
            float v = acc[n][r] + (p.bias ? p.bias[co] : 0.f);
            if (p.relu) v = fmaxf(v, 0.f);
            if (res) v += rv[r];
            y[(int64_t)co * HW] = v;
        }
    }
}

// y = act(sum_s partial[s] + bias) + res, partials summed in the order s = 0, 1, ...
__global__ void conv2d_finish_kernel(const float *partial, int split, int64_t n_plane, int64_t hw, int c_out, const float *bias,
                                     const float *res, int relu, float *y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over batch * c_out * hw
    if (i >= n_plane) return;
    float v = partial[i];
    for (int s = 1; s < split; ++s) v += partial[(int64_t)s * n_plane + i];
    if (bias) v += bias[(i / hw) % c_out];
    if (relu) v = fmaxf(v, 0.f);
    if (res) v += res[i];
    y[i] = v;
}

int launch_conv2d_finish(const float *partial, int split, int batch, int c_out, int H, int W, const float *bias, const float *res, int relu,
                         float *y, hipStream_t stream) {
    const int64_t n_plane = (int64_t)batch * c_out * H * W;
    hipLaunchKernelGGL(conv2d_finish_kernel, dim3((unsigned)ceil_div(n_plane, 256)), dim3(256), 0, stream, partial, split, n_plane,
                       (int64_t)H * W, c_out, bias, res, relu, y);
    RVC_LAUNCH_CHECK();
    return 0;
}

template <int WM, int WN, int NT, int TAPS>
static size_t conv2d_lds_bytes() {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, PAD = TAPS == 9 ? 1 : 0;
    constexpr int WPIECES = (TAPS * C2_CIC * BM * 4 + 1023) / 1024;
    return (size_t)(3 * WPIECES * 256 + 2 * C2_CIC * c2_patch_max(BN, PAD)) * sizeof(float);
}

template <int WM, int WN, int NT, int TAPS>
static int conv2d_launch(const Conv2dParams &p, hipStream_t stream) {
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    const size_t lds = conv2d_lds_bytes<WM, WN, NT, TAPS>();
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [lds] {
        err = hipFuncSetAttribute((const void *)conv2d_mfma_kernel<WM, WN, NT, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (err != hipSuccess) return fail("conv2d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(err));
    const int th = BN / p.W;
    dim3 grid((unsigned)(ceil_div(p.H, th) * (p.m_pad / BM)), (unsigned)p.batch, (unsigned)p.split);
    hipLaunchKernelGGL((conv2d_mfma_kernel<WM, WN, NT, TAPS>), grid, dim3(256), lds, stream, p);
    RVC_LAUNCH_CHECK();
    return 0;
}

size_t conv2d_workspace_bytes(int batch, int c_in, int c_out, int H, int W, int taps);

// how many ways K is split so that the launch has ~2000 wave tiles
static int conv2d_split(int c_in, int m_pad, int H, int W, int bn) {
    const int64_t tiles = ceil_div(H, bn / W) * (bn / 32) * (int64_t)(m_pad / 32);   // wave tiles: 32 channels x 32 pixels
    const int chunks = c_in / C2_CIC;
    int split = (int)(2048 / (tiles > 0 ? tiles : 1));   // ~2 waves per SIMD
    if (split > chunks / 4) split = chunks / 4;     // at least 4 chunks (32 input channels) per split
    if (split > 8) split = 8;
    return split < 1 ? 1 : split;
}

static int conv2d_bn(int m_pad, int W) {
    // 64 channels x 64 pixels when the channel count allows (the tap slab is reused by two pixel waves), else 32 x 128
    if (W > 64 || m_pad % 64) return 128;
    return 64;
}

size_t conv2d_workspace_bytes(int batch, int c_in, int c_out, int H, int W, int taps) {
    (void)taps;
    const int m_pad = (c_out + 31) / 32 * 32;
    int split = conv2d_split(c_in, m_pad, H, W, conv2d_bn(m_pad, W));
    if (knob("RVC_C2_SPLIT", 0)) split = 64;   // experiments: room for any override
    return split > 1 ? (size_t)split * batch * c_out * H * W * sizeof(float) : 0;
}

int launch_conv2d(const float *x, const float *w, const float *bias, const float *res, float *y, int batch, int c_in, int c_out, int H,
                  int W, int taps, int relu, float *ws, size_t ws_bytes, hipStream_t stream) {
    if (taps != 9 && taps != 1) return fail("conv2d: %d taps unsupported (3x3 or 1x1)", taps);
    if (c_in % C2_CIC) return fail("conv2d: c_in %d is not a multiple of %d", c_in, C2_CIC);
    if (W < 4 || W > 128 || (W & (W - 1))) return fail("conv2d: row length %d unsupported (a power of two in 4..128)", W);
    if ((int64_t)c_in * H * W >= ((int64_t)1 << 29) || (int64_t)c_out * H * W >= ((int64_t)1 << 29))
        return fail("conv2d: a %d x %d x %d map exceeds the kernel's 32-bit addressing", c_in > c_out ? c_in : c_out, H, W);
    if (batch <= 0 || H <= 0) return 0;
    Conv2dParams p;
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y;
    p.c_in = c_in; p.c_out = c_out; p.m_pad = (c_out + 31) / 32 * 32; p.H = H; p.W = W; p.taps = taps; p.relu = relu; p.batch = batch;
    static const int dbg = knob("RVC_C2_DEBUG", 0);
    static const int split_env = knob("RVC_C2_SPLIT", 0);
    p.debug = dbg;
    const int bn = conv2d_bn(p.m_pad, W);
    p.split = conv2d_split(c_in, p.m_pad, H, W, bn);
    if (split_env > 0 && split_env <= c_in / C2_CIC) p.split = split_env;
    p.chunks_per_split = (int)ceil_div(c_in / C2_CIC, p.split);
    p.split = (int)ceil_div(c_in / C2_CIC, p.chunks_per_split);
    if (p.split > 1) {
        const size_t need = (size_t)p.split * batch * c_out * H * W * sizeof(float);
        if (!ws || ws_bytes < need) return fail("conv2d: workspace of %zu bytes needed, %zu given", need, ws_bytes);
        p.partial = ws;
    }
    int rc;
    if (bn == 64) rc = taps == 9 ? conv2d_launch<2, 2, 1, 9>(p, stream) : conv2d_launch<2, 2, 1, 1>(p, stream);
    else rc = taps == 9 ? conv2d_launch<1, 4, 1, 9>(p, stream) : conv2d_launch<1, 4, 1, 1>(p, stream);
    if (rc) return rc;
    if (p.split > 1) return launch_conv2d_finish(p.partial, p.split, batch, c_out, H, W, bias, res, relu, y, stream);
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv2d_pack_weight(const float *w_host, int c_out, int c_in, int kh, int kw, float *w_dev, void *stream) {
    if (!w_host || !w_dev || c_out <= 0 || c_in <= 0 || !((kh == 3 && kw == 3) || (kh == 1 && kw == 1)))
        return fail("rvc_conv2d_pack_weight: bad argument (3x3 or 1x1 kernels)");
    const int taps = kh * kw, m_pad = (c_out + 31) / 32 * 32;
    std::vector<float> packed((size_t)taps * c_in * m_pad, 0.f);
    for (int co = 0; co < c_out; ++co)
        for (int ci = 0; ci < c_in; ++ci)
            for (int t = 0; t < taps; ++t) packed[((size_t)t * c_in + ci) * m_pad + co] = w_host[((size_t)co * c_in + ci) * taps + t];
    hipError_t e = hipMemcpyAsync(w_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_conv2d_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv2d_packed_floats(int c_out, int c_in, int kh, int kw, size_t *out) {
    if (!out || c_out <= 0 || c_in <= 0) return fail("rvc_conv2d_packed_floats: bad argument");
    *out = (size_t)kh * kw * c_in * ((c_out + 31) / 32 * 32);
    return 0;
}

extern "C" int rvc_conv2d_workspace_bytes(int batch, int c_in, int c_out, int height, int width, int kh, int kw, size_t *out) {
    if (!out) return fail("rvc_conv2d_workspace_bytes: null pointer");
    if (width < 4 || width > 128 || (width & (width - 1)) || height <= 0 || batch <= 0 || c_in <= 0 || c_out <= 0)
        return fail("rvc_conv2d_workspace_bytes: row length %d unsupported (a power of two in 4..128), or an empty map", width);
    *out = conv2d_workspace_bytes(batch, c_in, c_out, height, width, kh * kw);
    return 0;
}

extern "C" int rvc_conv2d_forward(const float *x_dev, const float *w_packed_dev, const float *bias_dev, const float *res_dev,
                                  float *y_dev, int batch, int c_in, int c_out, int height, int width, int kh, int kw, int relu,
                                  void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!x_dev || !w_packed_dev || !y_dev) return fail("rvc_conv2d_forward: null pointer");
    if (!((kh == 3 && kw == 3) || (kh == 1 && kw == 1))) return fail("rvc_conv2d_forward: %dx%d kernels unsupported", kh, kw);
    return launch_conv2d(x_dev, w_packed_dev, bias_dev, res_dev, y_dev, batch, c_in, c_out, height, width, kh * kw, relu,
                         (float *)workspace_dev, workspace_bytes, (hipStream_t)stream);
}
