// K11 -- fp32 GEMM / strided conv1d on the bf16 matrix cores with fp32-exact operands ("bf16x3").
//
//   y[m][n] = act( sum_k W[m][k] * X[k][n] + bias[m] ) + res[m][n]
//
// for the GEMM-shaped layers that used to go through hipBLASLt / MIOpen in fp32: HuBERT's attention / FFN projections
// (X = activations [n][k] row-major, 1599 x 768..3072) and its stride-2 feature-extractor convs (X[k = tap * C_in + ci][n] =
// x[ci][n * stride + tap - pad], channel-major).  Same arithmetic as winobf.hip: every fp32 operand is split EXACTLY into three
// bf16 numbers and the six products of order <= 2^-16 are formed by v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped
// terms < 2^-23 of a product) -- 6 x 32 matrix cycles for 16 k where the fp32 matrix instruction needs 8 x 64.
//
// Weights: split once at load into ready-made A fragments (1 KiB: 32 rows x 16 k x one split), streamed by LDS-DMA through a
// ring of three 12 KiB slots.  Activations: fetched by the block's 512 threads (thread = one column x 8 consecutive k), split
// with v_cvt_pk_bf16_f32 and written to LDS as B fragments (ring of two), one k16-step ahead of the matrix instructions.
// Block = 128 rows x 256 columns, 8 waves (2 x 4) of 64 x 64: FOUR independent accumulators per wave, so a wave's matrix
// instructions issue every 32 cycles without waiting on each other (the lesson of winobf.hip, where one accumulator per step
// made a 64-cycle chain); two waves per SIMD, and the staging of the next steps placed between the matrix instructions.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

struct GemmBfParams {
    const void *a = nullptr;         // gemmbf_pack_host's slab
    const float *x = nullptr;
    float *y = nullptr;
    const float *bias = nullptr;     // [M] or null
    const float *res = nullptr;      // y's layout, or null
    int M = 0, K = 0;                // rows, contraction length (conv: taps * c_in)
    int c_in = 0;                    // conv: channels per tap (K for a plain GEMM)
    int64_t N = 0;                   // columns
    int x_mode = 0;                  // 0: x[n][k] row-major (ldx floats per row); 1: x[ci][t] channel-major, t = n * stride + tap * dil - pad
    int64_t ldx = 0, l_in = 0;
    int stride = 1, dil = 1, pad = 0;
    int y_mode = 0;                  // 0: y[n][m] row-major (ldy); 1: y[m][n] channel-major (ldy)
    int64_t ldy = 0;
    int act = 0;                     // 0 none, 1 GELU (erf form)
    int one_channel = 0;             // conv mode: ONE input channel, the step's 16 pseudo-channels are 16 consecutive samples (ldx is then 1
                                     // by construction; the flag, not ldx == 1, selects the clamped path: l_in == 1 is a legal multi-channel input)
    int batch = 1;
    int64_t x_bstride = 0, y_bstride = 0;
    int n_col_blocks = 0;
};

// Block = 128 rows x 256 columns, 8 waves (2 x 4) of 64 x 64, ONE block per CU: the launch asks for the CU's whole LDS although
// it uses 84 KiB of it, so that no other kernel's workgroup can sit next to it -- a workgroup issuing bf16 matrix
// instructions that shares a CU with a workgroup of the fp32 Winograd kernel (wino.hip) corrupts that kernel's results
// (profiles/r03_mfma_cohabitation.txt).  The first form of this kernel (128 x 128, 4 waves, two blocks of 60 KiB per CU with
// independent barriers) was ~10 % faster and is what the measurements in that file were taken with.
constexpr int GBF_BM = 128, GBF_WN = 4, GBF_BN = 64 * GBF_WN, GBF_NW = 2 * GBF_WN, GBF_NTH = 64 * GBF_NW;
constexpr int GBF_UPW = (12 + GBF_NW - 1) / GBF_NW;   // tap-fragment DMA instructions per wave per step (12 KiB pieces)
constexpr int GBF_A_SLOT = 4 * 3 * 1024;            // [32-row block 4][split 3][1 KiB]
constexpr int GBF_B_PLANE = GBF_BN * 16;            // [256 columns][8 bf16]
constexpr int GBF_B_SLOT = 3 * 2 * GBF_B_PLANE;     // [split][k half]
constexpr int GBF_LDS_USED = 3 * GBF_A_SLOT + 2 * GBF_B_SLOT;
constexpr int GBF_LDS = LDS_WHOLE_CU;               // requested: the whole CU (common.h)
static_assert(GBF_LDS_USED <= GBF_LDS, "");

typedef float gbf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gbf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gbf_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gbf_u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) *gbf_lptr_t;

// DBG (ablations, wrong results): 1 no activation staging after the prologue, 2 no matrix instructions, 4 no tap DMA, 8 no barrier
template <int XMODE, int DBG = 0>
__global__ void __launch_bounds__(GBF_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemmbf_kernel(const GemmBfParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gbf_smem[];
    unsigned char *const as = gbf_smem;                       // [3][A_SLOT]
    unsigned char *const bs = gbf_smem + 3 * GBF_A_SLOT;      // [2][B_SLOT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < GBF_NW);
    const int wm = wave / GBF_WN, wn = wave % GBF_WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int bz = blockIdx.z;
    // the row blocks of one column tile read the same activations: ids 8 apart land on the same XCD (its L2)
    const int n_m = p.M / GBF_BM;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int col_blk = (seq / n_m) * 8 + xcd;
    if (col_blk >= p.n_col_blocks) return;
    const int mblk = seq % n_m;
    const int m0 = mblk * GBF_BM;
    const int64_t n0 = (int64_t)col_blk * GBF_BN;
    const int n_steps = p.K / 16;
    const float *const x = p.x + (int64_t)bz * p.x_bstride;

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, (int)((int64_t)p.M * p.K * 6), 0x00020000);
    const int blk_base = mblk * n_steps * GBF_A_SLOT;
    auto dma_a = [&](int s) __attribute__((always_inline)) {
        unsigned char *dst = as + (s % 3) * GBF_A_SLOT;
        const int s0 = blk_base + s * GBF_A_SLOT;
#pragma unroll
        for (int i = 0; i < GBF_UPW; ++i) {
            int n = wave + GBF_NW * i;
            if (n >= 12) n -= GBF_NW;          // the overhang repeats a piece this wave has already issued: same bytes, same place
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (gbf_lptr_t)(dst + n * 1024), 16, 16 * lane, s0 + n * 1024, 0, 0);
        }
    };

    // ---- activation staging: thread = column t_l, k half kh (8 consecutive k of the step) -------------------------------
    const int t_l = tid % GBF_BN, kh = tid / GBF_BN;
    const int64_t n_col = n0 + t_l;
    const bool col_ok = n_col < p.N;
    float xv[8];
    bool lx_ok_prev = true;   // conv mode: was the column of the values now in xv inside the input?
    // (pieces, so that the step body can place them between matrix instructions)
    const float *lx_src = nullptr;
    bool lx_ok = true;
    auto load_x_begin = [&](int s) __attribute__((always_inline)) {
        if constexpr (XMODE == 0) {
            lx_src = x + (col_ok ? n_col : 0) * p.ldx + 16 * s + 8 * kh;
        } else {
            const int k0 = 16 * s;
            const int tap = k0 / p.c_in, ci0 = k0 - tap * p.c_in + 8 * kh;
            const int64_t t_in = n_col * p.stride + (int64_t)tap * p.dil - p.pad;
            lx_ok = col_ok && t_in >= 0 && t_in < p.l_in;
            lx_src = x + (int64_t)ci0 * p.ldx + (lx_ok ? t_in : 0);
        }
    };
    auto load_x_piece = [&](int e) __attribute__((always_inline)) {   // XMODE 0: e = 0, 1 (a float4 each); XMODE 1: e = 0..7
        if constexpr (XMODE == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(lx_src + 4 * e);
            xv[4 * e] = v.x; xv[4 * e + 1] = v.y; xv[4 * e + 2] = v.z; xv[4 * e + 3] = v.w;
        } else if (p.one_channel) {
            // one input channel: the step's 16 pseudo-channels are consecutive samples; the taps beyond k carry zero weights, so
            // the last columns' tail is clamped into x instead of being read past its end
            const int64_t at = (lx_src - x) + e;
            xv[e] = x[at < p.l_in ? at : p.l_in - 1];
        } else {
            xv[e] = lx_src[(int64_t)e * p.ldx];
        }
    };
    constexpr int LX_PIECES = XMODE == 0 ? 2 : 8;
    auto load_x = [&](int s) __attribute__((always_inline)) {
        load_x_begin(s);
#pragma unroll
        for (int e = 0; e < LX_PIECES; ++e) load_x_piece(e);
    };
    // exact three-way split of the 8 values -> three 16-byte B-fragment pieces
    gbf_u32x4 sw[3];
    auto split_pair = [&](int e2) __attribute__((always_inline)) {
        gbf_f32x2 v = {xv[2 * e2], xv[2 * e2 + 1]};
        if (XMODE == 0 ? !col_ok : !lx_ok_prev) v = gbf_f32x2{0.f, 0.f};
        const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, gbf_bf16x2));
        const gbf_f32x2 r1 = v - gbf_f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
        const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, gbf_bf16x2));
        const gbf_f32x2 r2 = r1 - gbf_f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
        const unsigned w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, gbf_bf16x2));
        sw[0][e2] = w0; sw[1][e2] = w1; sw[2][e2] = w2;
    };
    auto store_split = [&](int s) __attribute__((always_inline)) {
        unsigned char *dst = bs + (s & 1) * GBF_B_SLOT + kh * GBF_B_PLANE + t_l * 16;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<gbf_u32x4 *>(dst + sp * 2 * GBF_B_PLANE) = sw[sp];
    };
    auto store_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) split_pair(e2);
        store_split(s);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;

    auto rd = [&](const unsigned char *ptr) __attribute__((always_inline)) {
        return __builtin_bit_cast(gbf_bf16x8, *reinterpret_cast<const gbf_u32x4 *>(ptr));
    };

    // ---- prologue: steps 0 and 1 of the tap stream, step 0 of the activations in LDS, step 1 in registers -----------------
    dma_a(0);
    if (n_steps > 1) dma_a(1);
    load_x(0);
    lx_ok_prev = lx_ok;
    store_x(0);
    if (n_steps > 1) { load_x(1); lx_ok_prev = lx_ok; }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XMODE == 0 ? 2 : 8) : "memory");   // the DMA pieces (older than step 1's loads) have landed
    lds_barrier();

    // A wave's 24 matrix instructions per step run on four independent accumulators and issue every 32 cycles; the staging of
    // the next steps (the three DMA issues, the split of step s + 1's values, their LDS writes, the global loads of step s + 2)
    // is placed in the gaps between them -- left in front of them the two phases add up (193 + 187 + 187 us on the first conv
    // layer: two blocks share a CU but start together and stay in lock step).
    for (int s = 0; s < n_steps; ++s) {
        const bool more1 = s + 1 < n_steps, more2 = s + 2 < n_steps;
        const unsigned char *ab = as + (s % 3) * GBF_A_SLOT + wm * 2 * 3 * 1024 + lane * 16;
        const unsigned char *bb = bs + (s & 1) * GBF_B_SLOT + half * GBF_B_PLANE + (wn * 64 + l31) * 16;
        auto filler = [&](int k) __attribute__((always_inline)) {
            if (k < 3) {
                if (k < GBF_UPW && more2 && !(DBG & 4)) {
                    unsigned char *dst = as + ((s + 2) % 3) * GBF_A_SLOT;
                    int n = wave + GBF_NW * k;
                    if (n >= 12) n -= GBF_NW;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (gbf_lptr_t)(dst + n * 1024), 16, 16 * lane, blk_base + (s + 2) * GBF_A_SLOT + n * 1024, 0, 0);
                }
            } else if (k < 7) {
                if (more1 && !(DBG & 1)) split_pair(k - 3);
            } else if (k == 7) {
                if (more1 && !(DBG & 1)) store_split(s + 1);
            } else if (k == 8) {
                if (more2 && !(DBG & 1)) load_x_begin(s + 2);
            } else if (k < 9 + LX_PIECES) {
                if (more2 && !(DBG & 1)) load_x_piece(k - 9);
            }
        };
        gbf_bf16x8 fb[2][3];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) fb[ni][sp] = rd(bb + sp * 2 * GBF_B_PLANE + ni * 32 * 16);
        gbf_bf16x8 fa[2][3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fa[0][sp] = rd(ab + sp * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (!(DBG & 2)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ia[i]], fb[ni][ib[i]], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int k = mi * 12 + i * 2 + ni;
                    if (k == 1) {
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) fa[1][sp] = rd(ab + (3 + sp) * 1024);
                    }
                    filler(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        lx_ok_prev = lx_ok;
        // everything older than this step's own memory operations has completed: the tap fragments of step s + 1 are in LDS
        if (more2 && !(DBG & 5)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GBF_UPW + (XMODE == 0 ? 2 : 8)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(DBG & 8)) lds_barrier();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    float *const y = p.y + (int64_t)bz * p.y_bstride;
    const float *const res = p.res ? p.res + (int64_t)bz * p.y_bstride : nullptr;
    auto finish = [&](float v, int m) __attribute__((always_inline)) {
        if (p.bias) v += p.bias[m];
        if (p.act == 1) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        return v;
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t n = n0 + wn * 64 + ni * 32 + l31;
            if (n >= p.N) continue;
            const int mb = m0 + wm * 64 + mi * 32 + 4 * half;
            if (p.y_mode == 0) {   // row-major: 4 consecutive rows m are 16 contiguous bytes
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int m = mb + 8 * rg;
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = finish(acc[mi][ni][4 * rg + r], m + r);
                    float *dst = y + n * p.ldy + m;
                    if (res) o += *reinterpret_cast<const f32x4 *>(res + n * p.ldy + m);
                    *reinterpret_cast<f32x4 *>(dst) = o;
                }
            } else {               // channel-major: a wave's 32 columns are 128 contiguous bytes per row
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    float v = finish(acc[mi][ni][r], m);
                    if (res) v += res[(int64_t)m * p.ldy + n];
                    y[(int64_t)m * p.ldy + n] = v;
                }
            }
        }
}

static int gemmbf_launch(GemmBfParams p, hipStream_t stream) {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute((const void *)gemmbf_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, GBF_LDS);
        if (err == hipSuccess) err = hipFuncSetAttribute((const void *)gemmbf_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, GBF_LDS);
    });
    if (err != hipSuccess) return fail("gemm bf16x3: cannot reserve %d bytes of LDS: %s", GBF_LDS, hipGetErrorString(err));
    p.n_col_blocks = (int)ceil_div(p.N, GBF_BN);
    const int n_m = p.M / GBF_BM;
    dim3 grid((unsigned)(ceil_div(p.n_col_blocks, 8) * 8 * n_m), 1, (unsigned)p.batch);
#ifdef RVC_ABLATE
    static const int dbg = knob("RVC_GBF_DBG", 0);
    if (dbg && p.x_mode == 1) {
        auto go = [&](auto k) { (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, GBF_LDS); hipLaunchKernelGGL(k, grid, dim3(GBF_NTH), GBF_LDS, stream, p); };
        switch (dbg) {
            case 1: go(gemmbf_kernel<1, 1>); break;
            case 2: go(gemmbf_kernel<1, 2>); break;
            case 3: go(gemmbf_kernel<1, 3>); break;
            case 4: go(gemmbf_kernel<1, 4>); break;
            case 8: go(gemmbf_kernel<1, 8>); break;
            case 7: go(gemmbf_kernel<1, 7>); break;
            default: go(gemmbf_kernel<1, 0>); break;
        }
        RVC_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (p.x_mode == 0) hipLaunchKernelGGL(gemmbf_kernel<0>, grid, dim3(GBF_NTH), GBF_LDS, stream, p);
    else hipLaunchKernelGGL(gemmbf_kernel<1>, grid, dim3(GBF_NTH), GBF_LDS, stream, p);
    RVC_LAUNCH_CHECK();
    return 0;
}

// W [M][K] row-major fp32 (conv: K = taps * c_in with k = tap * c_in + ci) -> [M / 128][K / 16][32-row block 4][split 3][lane 64][8] bf16
static void gemmbf_pack_rows(const float *w, int M, int K, std::vector<uint16_t> *out) {
    const int n_steps = K / 16, n_m = M / GBF_BM;
    out->assign((size_t)M * K * 3, 0);
    for (int mb = 0; mb < n_m; ++mb)
        for (int s = 0; s < n_steps; ++s)
            for (int mi = 0; mi < 4; ++mi)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int m = mb * GBF_BM + mi * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                        float r = w[(size_t)m * K + k];
                        for (int sp = 0; sp < 3; ++sp) {
                            const uint16_t h = bf16_rne(r);
                            uint32_t bits = (uint32_t)h << 16;
                            float f;
                            memcpy(&f, &bits, 4);
                            r -= f;   // exact in fp32
                            const size_t piece = (((size_t)mb * n_steps + s) * 4 + mi) * 3 + sp;
                            (*out)[piece * 512 + lane * 8 + e] = h;
                        }
                    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_gemm_bf16x3_weight_bytes(int m, int k, size_t *bytes) {
    // k < 16 is ONLY the single-input-channel conv of <= 16 taps (HuBERT's first layer), packed as one zero-padded k16 step;
    // rvc_gemm_bf16x3_pack_weight rejects any other use of a short k
    if (k > 0 && k < 16) k = 16;
    if (!bytes || m <= 0 || k <= 0 || m % GBF_BM || k % 16) return fail("rvc_gemm_bf16x3_weight_bytes: m must be a multiple of 128, k of 16");
    *bytes = (size_t)m * k * 6;
    return 0;
}

extern "C" int rvc_gemm_bf16x3_pack_weight(const float *w_host, int m, int k_total, int conv_taps, void *a_dev, void *stream) {
    if (!w_host || !a_dev) return fail("rvc_gemm_bf16x3_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_gemm_bf16x3_weight_bytes(m, k_total, &bytes)) return 1;
    std::vector<float> rows;
    const float *w = w_host;
    if (conv_taps > 1 && k_total == conv_taps && conv_taps <= 16) {   // one input channel: [m][taps] -> [m][16], zero-padded
        rows.assign((size_t)m * 16, 0.f);
        for (int r = 0; r < m; ++r)
            for (int t = 0; t < conv_taps; ++t) rows[(size_t)r * 16 + t] = w_host[(size_t)r * conv_taps + t];
        std::vector<uint16_t> packed1;
        gemmbf_pack_rows(rows.data(), m, 16, &packed1);
        if (packed1.size() * sizeof(uint16_t) != bytes) return fail("rvc_gemm_bf16x3_pack_weight: internal size mismatch");
        hipError_t e1 = hipMemcpyAsync(a_dev, packed1.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize((hipStream_t)stream);
        if (e1 != hipSuccess) return fail("rvc_gemm_bf16x3_pack_weight: %s", hipGetErrorString(e1));
        return 0;
    }
    if (conv_taps > 1) {   // conv weight [m][c_in][taps] -> [m][tap * c_in + ci]
        if (k_total % conv_taps || (k_total / conv_taps) % 16) return fail("rvc_gemm_bf16x3_pack_weight: c_in must be a multiple of 16");
        const int c_in = k_total / conv_taps;
        rows.resize((size_t)m * k_total);
        for (int r = 0; r < m; ++r)
            for (int ci = 0; ci < c_in; ++ci)
                for (int t = 0; t < conv_taps; ++t) rows[(size_t)r * k_total + (size_t)t * c_in + ci] = w_host[((size_t)r * c_in + ci) * conv_taps + t];
        w = rows.data();
    }
    if (k_total % 16) return fail("rvc_gemm_bf16x3_pack_weight: k %d is not a multiple of 16 (a short k is the one-input-channel conv only: conv_taps == k_total)", k_total);
    std::vector<uint16_t> packed;
    gemmbf_pack_rows(w, m, k_total, &packed);
    if (packed.size() * sizeof(uint16_t) != bytes) return fail("rvc_gemm_bf16x3_pack_weight: internal size mismatch");
    hipError_t e = hipMemcpyAsync(a_dev, packed.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_gemm_bf16x3_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_linear_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, const float *res_dev, float *y_dev,
                                 int64_t n_rows, int in_features, int out_features, int act, void *stream) {
    if (!x_dev || !a_dev || !y_dev) return fail("rvc_linear_bf16x3: null pointer");
    if (out_features % GBF_BM || in_features % 16) return fail("rvc_linear_bf16x3: out_features must be a multiple of 128, in_features of 16");
    if ((int64_t)out_features * in_features * 6 >= ((int64_t)1 << 31)) return fail("rvc_linear_bf16x3: weight slab exceeds 2 GiB");
    if (act < 0 || act > 1) return fail("rvc_linear_bf16x3: act must be 0 (none) or 1 (gelu)");
    if (n_rows <= 0) return 0;
    GemmBfParams p;
    p.a = a_dev; p.x = x_dev; p.y = y_dev; p.bias = bias_dev; p.res = res_dev;
    p.M = out_features; p.K = in_features; p.c_in = in_features; p.N = n_rows;
    p.x_mode = 0; p.ldx = in_features; p.y_mode = 0; p.ldy = out_features; p.act = act;
    return gemmbf_launch(p, (hipStream_t)stream);
}

extern "C" int rvc_conv1d_bf16x3(const float *x_dev, const void *a_dev, const float *bias_dev, float *y_dev, int batch, int c_in,
                                 int c_out, int64_t l_in, int k, int stride, int padding, int act, void *stream) {
    if (!x_dev || !a_dev || !y_dev) return fail("rvc_conv1d_bf16x3: null pointer");
    const bool one_channel = c_in == 1;   // X[k][n] = x[n * stride + k], k < 16: the 16 "channels" are 16 consecutive samples (row pitch 1)
    if (one_channel && (k > 16 || padding != 0 || batch != 1))
        return fail("rvc_conv1d_bf16x3: a single input channel takes <= 16 taps, no padding, batch 1");
    if (c_out % GBF_BM || (!one_channel && c_in % 16) || k < 1 || stride < 1 || padding < 0) return fail("rvc_conv1d_bf16x3: c_out must be a multiple of 128, c_in of 16");
    if ((int64_t)c_out * c_in * k * 6 >= ((int64_t)1 << 31)) return fail("rvc_conv1d_bf16x3: weight slab exceeds 2 GiB");
    if (act < 0 || act > 1) return fail("rvc_conv1d_bf16x3: act must be 0 (none) or 1 (gelu)");
    const int64_t l_out = (l_in + 2 * padding - k) / stride + 1;
    if (l_out <= 0 || batch <= 0) return 0;
    GemmBfParams p;
    p.a = a_dev; p.x = x_dev; p.y = y_dev; p.bias = bias_dev;
    p.M = c_out; p.K = k * c_in; p.c_in = c_in; p.N = l_out;
    p.x_mode = 1; p.ldx = l_in; p.l_in = l_in; p.stride = stride; p.dil = 1; p.pad = padding;
    if (one_channel) { p.K = 16; p.c_in = 16; p.ldx = 1; p.one_channel = 1; }   // one tap of 16 pseudo-channels; weights beyond k are zero
    p.y_mode = 1; p.ldy = l_out; p.act = act; p.batch = batch;
    p.x_bstride = (int64_t)c_in * l_in; p.y_bstride = (int64_t)c_out * l_out;
    return gemmbf_launch(p, (hipStream_t)stream);
}
