// K3u -- the vocoder's upsampling step on the bf16 matrix cores, fp32 operands as exact bf16x3 splits:
//     y = ConvTranspose1d(leaky(x, 0.1)) [+ noise_conv(har_source)] + bias            (hifigan_nsf.py:113-127, 142-154, 184-193;
//                                                                                      hifigan_mrf.py: upsamples / noise_convs)
// in POLYPHASE form: with rate r, kernel size k <= 2 r and padding p, output sample t = q r + ph - p of channel co is
//     sum_ci  W[ci][co][ph] a(x[ci][q]) + W[ci][co][ph + r] a(x[ci][q - 1])
// -- a GEMM with rows m = (co, ph), columns q, K = 2 taps x c_in -- plus, where the stage's noise conv is folded in, K' = (r - 1) s + k_n
// extra rows V[kq][q] = har[q S + kq - P] on the offset-0 tap (decoder.hip's fold: noise_convs[i] has stride s and k_n taps over the
// 1-channel source).  conv.hip runs this GEMM on the fp32 matrix instruction at 43-90 TF; this is convbf1.hip's (K3d's) structure on it:
//   * persistent 8-wave workgroups, one per CU (whole LDS), walking (m-block, 64- or 128-column) tiles; an m-block is MB = 256 / 128 / 64
//     consecutive GEMM rows m = co r + ph, so a tile's outputs are runs of N1 r CONTIGUOUS samples per channel (its first and last
//     channel may be shared with the neighbouring m-block: each writes the phases it owns);
//   * the input channels in 64-channel chunks through a two-buffer LDS ring in [time][split][channel] order (row stride an odd multiple
//     of 16 B): waves 4-7 stage (raw rows HBM -> registers a chunk ahead -> leaky ReLU -> exact three-way bf16 split in plain fp32 VALU ->
//     LDS; the folded noise rows straight from har_source, as one more chunk with the offset-0 tap only), waves 0-3 multiply: tap
//     fragments (three terms: these are fp32 weights) L2 -> registers in a ring of four groups that wraps from tile to tile, window
//     fragments LDS -> registers double-buffered, six products per multiply-add, smallest first, fp32 accumulate;
//   * epilogue: accumulators + bias -> an LDS tile [channel][q r + ph] (the interleave happens here), which the stagers drain with coalesced
//     stores while the compute waves are in the next tile's first chunk.
// One barrier per chunk, one more per launch.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "conv.h"

namespace rvc {

typedef __bf16 ub_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ub_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ub_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ub_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ub_u32x4 __attribute__((ext_vector_type(4)));

constexpr int UB_NTH = 512;
constexpr int UB_RSRC_FLAGS = 0x00020000;
constexpr unsigned UB_OOB = 0x80000000u;
constexpr int UB_CK = 64;                   // input channels per chunk

struct UbParams {
    const float *x = nullptr;        // [batch][c_in][L_in]
    const float *har = nullptr;      // [batch][Lh] or null: the folded noise conv's source
    const void *u = nullptr;         // upsbf_pack_host's slab
    const float *bias = nullptr;     // [c_out] (ups bias + noise-conv bias): added by the stagers when vk == 0 (with folded noise rows it is one of them); may be null
    float *y = nullptr;              // [batch][c_out][L_out]
    int64_t L_in = 0, L_out = 0, Lh = 0;
    int c_in = 0, c_out = 0;
    int n_xchunks = 0;               // c_in / 64
    int vk = 0;                      // folded noise rows (0: none); one more chunk then
    int S = 0, P = 0;                // V[kq][q] = har[q S + kq - P]
    int pad = 0;
    float slope = 0.1f;
    int n_mblk = 0, col_tiles = 0, n_tiles = 0, per_xcd = 0;
};

__device__ __forceinline__ float ub_sub_np(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float ub_mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void ub_split3_np(float a, float b, unsigned w[3]) {
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(ub_f32x2{a, b}, ub_bf16x2));
        w[level] = ww;
        if (level < 2) {
            a = ub_sub_np(a, __uint_as_float(ww << 16));
            b = ub_sub_np(b, __uint_as_float(ww & 0xffff0000u));
        }
    }
}

template <int R, int MB>
struct UbGeom {
    static constexpr int RB_T = MB / 32;                      // 32-row blocks per tile
    static constexpr int RBW = RB_T >= 4 ? RB_T / 4 : 1;      // ... per compute wave
    static constexpr int CG = RB_T >= 4 ? 1 : 4 / RB_T;       // column groups of 64
    static constexpr int N1 = 64 * CG;                        // GEMM columns (input positions q) per tile
    static constexpr int W = N1 * R;                          // output samples per channel and tile
    static constexpr int CPB = (MB - 1) / R + 2;              // channel SLOTS of an m-block's output tile: its MB GEMM rows (co, ph) start and end anywhere inside a channel
    static constexpr int KS = UB_CK / 16;
    static constexpr int NGC = 2 * KS;                        // (tap, k step) groups per x chunk; the noise chunk has KS
    static constexpr int PA = 4;                              // tap-fragment ring
    static constexpr int ROWB = 6 * UB_CK + 16;
    static constexpr int XROWS = N1 + 1;                      // q0 - 1 .. q0 + N1 - 1
    static constexpr int X_BYTES = (XROWS + 1) * ROWB;
    static constexpr int RC32 = (XROWS + 31) / 32;
    static constexpr int NIT = (UB_CK / 8) * RC32 / 4;
    static constexpr int IO_FLOATS = CPB * W;
    static constexpr int IO_BYTES = IO_FLOATS * 4 + 16;
    static constexpr int LDS_BYTES = 2 * X_BYTES + IO_BYTES;
    static constexpr int GROUP_BYTES = RB_T * 3 * 1024;       // [row block][split][lane][8 bf16]
    static_assert(MB == 64 || MB == 128 || MB == 256, "GEMM rows per tile");
    static_assert(R >= 1 && R <= MB, "a channel's phases span at most two m-blocks");
    static_assert((UB_CK / 8) * RC32 % 4 == 0, "the items must divide over the four stager waves");
    static_assert(NGC % PA == 0 && KS % PA == 0, "a chunk's groups must be whole turns of the tap ring");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

template <int R, int MB>
__global__ void __launch_bounds__(UB_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
upsbf_kernel(const UbParams p) {
    using GM = UbGeom<R, MB>;
    constexpr int RB_T = GM::RB_T, RBW = GM::RBW, N1 = GM::N1, W = GM::W, KS = GM::KS, NGC = GM::NGC, PA = GM::PA, ROWB = GM::ROWB;
    constexpr int NIT = GM::NIT, CK = UB_CK;

    extern __shared__ __attribute__((aligned(16))) unsigned char ub_smem[];
    unsigned char *const xs = ub_smem;                                        // [2][X_BYTES]
    float *const io_lds = reinterpret_cast<float *>(ub_smem + 2 * GM::X_BYTES);   // [cpb][W]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t L_in = p.L_in, L_out = p.L_out;
    const int nxc = p.n_xchunks, ncht = nxc + (p.vk > 0 ? 1 : 0);           // chunks per tile
    const int ngt = nxc * NGC + (p.vk > 0 ? KS : 0);                         // tap groups per tile
    const int n_mblk = p.n_mblk;
    const int m_total = p.c_out * R;                          // GEMM rows (co, ph), co-major: m-block b owns rows [MB b, MB b + MB)

    // tiles: (batch, column tile, m-block), m-block fastest -- the m-blocks of one column tile re-read the same x rows from one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int tile_end = (xcd + 1) * p.per_xcd < p.n_tiles ? (xcd + 1) * p.per_xcd : p.n_tiles;
    const int tile0 = xcd * p.per_xcd + slot;
    if (tile0 >= tile_end) return;
    const int my_tiles = (tile_end - tile0 + nslot - 1) / nslot;
    const int n_q = my_tiles * ncht;
    auto tile_coords = [&](int tl, int &mblk, int &ct, int &bb) __attribute__((always_inline)) {
        mblk = tl % n_mblk;
        const int rest = tl / n_mblk;
        bb = rest / p.col_tiles;
        ct = rest - bb * p.col_tiles;
    };

    // Both x buffers start as zeros: the noise chunk stages only the channel quads that hold noise rows (and the ones row), the products of
    // its other k steps multiply zero taps with whatever the buffer holds -- which must then be a finite number, not a stale bit pattern.
    for (int o = tid * 16; o < 2 * GM::X_BYTES; o += UB_NTH * 16) *reinterpret_cast<ub_u32x4 *>(xs + o) = ub_u32x4{0u, 0u, 0u, 0u};
    lds_barrier();
    const int nqp_noise = (p.vk + 1 + 7) / 8;                 // channel quad PAIRS of the noise chunk that carry data

    if (wave >= 4) {
        // ============================================ stagers: HBM <-> LDS ============================================================
        __builtin_amdgcn_s_setprio(1);
        const int sw = wave - 4, ht = tid - 256;
        float xr[NIT][4];
        constexpr int RC = GM::RC32;
        const int lq = lane >> 5;
        const unsigned L4 = (unsigned)(L_in * 4);
        const int x_bytes = (int)((int64_t)p.c_in * L_in * 4);
        // chunk `ci` of tile `tl`: ci < nxc -> input channels 64 ci ..; ci == nxc -> the folded noise rows
        auto x_issue = [&](int tl, int ci) __attribute__((always_inline)) {
            int mblk, ct, bb;
            tile_coords(tl, mblk, ct, bb);
            const int q0m1 = ct * N1 - 1;                                             // q of LDS row 0
            int sw_o = sw;
            asm volatile("" : "+s"(sw_o));
            if (ci < nxc) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (int64_t)bb * p.c_in * L_in), 0, x_bytes, UB_RSRC_FLAGS);
                const unsigned Lu = (unsigned)L_in;
                const int ch0 = ci * CK;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                    const int qd = 2 * qp + lq, r = rc * 32 + l31;
                    const unsigned tg = (unsigned)(q0m1 + r);                               // negative or beyond the row: >= L_in as unsigned
                    const bool ok = r < GM::XROWS && tg < Lu;
                    const unsigned base = (unsigned)(ch0 + 4 * qd) * L4 + tg * 4u;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? base + (unsigned)e * L4 : UB_OOB), 0, 0));
                }
            } else {
                const int h_bytes = (int)(p.Lh * 4);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.har + (int64_t)bb * p.Lh), 0, h_bytes, UB_RSRC_FLAGS);
                const int vk = p.vk;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                    if (qp >= nqp_noise) continue;                                          // (wave-uniform) nothing of the noise rows in these channels
                    const int qd = 2 * qp + lq, r = rc * 32 + l31;
                    const int q = q0m1 + r;
                    const bool okq = r < GM::XROWS && q >= 0 && q <= (int)L_in;              // V exists for q in [0, L_in]
                    const int64_t idx0 = (int64_t)q * p.S - p.P + 4 * qd;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t idx = idx0 + e;
                        const bool ok = okq && 4 * qd + e < vk && idx >= 0 && idx < p.Lh;
                        xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? (unsigned)idx * 4u : UB_OOB), 0, 0));
                        if (4 * qd + e == vk) xr[i][e] = 1.f;   // the ONES row: its taps are the bias (upsbf_pack_host)
                    }
                }
            }
        };
        auto x_write = [&](int q, int ci) __attribute__((always_inline)) {
            unsigned char *const xb = xs + (q & 1) * GM::X_BYTES;
            const float slope = ci < nxc ? p.slope : 1.f;                                   // the noise rows enter as they are
            int sw_o = sw;
            asm volatile("" : "+s"(sw_o));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                if (ci >= nxc && qp >= nqp_noise) continue;                                 // the noise chunk: only the quads that were fetched
                const int qd = 2 * qp + lq, r = rc * 32 + l31;
                unsigned w[2][3];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const float va = xr[i][2 * e2], vb = xr[i][2 * e2 + 1];
                    ub_split3_np(__builtin_fmaxf(va, ub_mul_np(va, slope)), __builtin_fmaxf(vb, ub_mul_np(vb, slope)), w[e2]);
                }
                unsigned char *o = xb + (r < GM::XROWS ? r : GM::XROWS) * ROWB + qd * 8;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<ub_u32x2 *>(o + s * 2 * CK) = ub_u32x2{w[0][s], w[1][s]};
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- the finished tile: io tile [channel][q r + ph] -> HBM, lanes = consecutive output samples (coalesced dword stores) ----------
        const int y_bytes = (int)((int64_t)p.c_out * L_out * 4);
        const bool fold = p.vk > 0;                           // the bias came through the GEMM (the ones row of the noise chunk)
        auto out_store = [&](int tl) __attribute__((always_inline)) {
            int mblk, ct, bb;
            tile_coords(tl, mblk, ct, bb);
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (int64_t)bb * p.c_out * L_out), 0, y_bytes, UB_RSRC_FLAGS);
            const int64_t t_base = (int64_t)ct * N1 * R - p.pad;
            const int m0 = mblk * MB, co0 = m0 / R;                    // the tile's first GEMM row / the channel of slot 0
            constexpr int n_el = GM::CPB * W;
            const unsigned Lo4 = (unsigned)(L_out * 4);
            for (int f0 = 0; f0 < n_el; f0 += 4 * 256) {
                float v[4], bv[4];
                unsigned off[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = f0 + k * 256 + ht;
                    const int row = f / W, i = f - row * W;
                    const int64_t t = t_base + i;
                    const int m = (co0 + row) * R + i % R;                 // this sample's GEMM row: inside the m-block? (its first and last
                    const bool ok = f < n_el && m >= m0 && m < m0 + MB && m < m_total && t >= 0 && t < L_out;   // channel may be shared with a neighbour)
                    v[k] = io_lds[f < n_el ? f : 0];
                    bv[k] = !fold && p.bias && ok ? p.bias[co0 + row] : 0.f;
                    off[k] = ok ? (unsigned)(co0 + row) * Lo4 + (unsigned)t * 4u : UB_OOB;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k] + bv[k]), yrs, (int)off[k], 0, 0);
            }
        };
        // chunk stream cursors: (tile index, chunk index) of the chunk to WRITE next (q + 1) and to REQUEST next (q + 2)
        int wt = 0, wc = 0, it = 0, ic = 0;
        auto adv = [&](int &t_, int &c_) __attribute__((always_inline)) {
            if (++c_ == ncht) { c_ = 0; ++t_; }
        };
        x_issue(tile0, 0);
        x_write(0, 0);
        adv(wt, wc); adv(it, ic);                            // both point at chunk 1
        if (1 < n_q) x_issue(tile0 + it * nslot, ic);
        adv(it, ic);                                         // chunk 2
        int ct_i = 0, cc_i = 0;                              // the chunk the compute waves are on (q)
        for (int q = 0; q < n_q; ++q) {
            lds_barrier();                                    // (A) buffer q & 1 complete; the compute waves are done with buffer (q + 1) & 1
            if (cc_i == 0 && q > 0) out_store(tile0 + (ct_i - 1) * nslot);
            if (q + 1 < n_q) x_write(q + 1, wc);
            adv(wt, wc);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (q + 2 < n_q) x_issue(tile0 + it * nslot, ic);
            adv(it, ic);
            adv(ct_i, cc_i);
        }
        lds_barrier();                                        // (E) the last tile's outputs are in the io tile
        out_store(tile0 + (my_tiles - 1) * nslot);
        return;
    }

    // ================================================ compute waves ==========================================================
    // RB_T >= 4: wave w owns row blocks RBW w .. of all 64 columns; RB_T == 2: wave w owns row block w & 1 of column group w >> 1
    const int rb0 = RB_T >= 4 ? RBW * wave : wave % RB_T;
    const int col0 = RB_T >= 4 ? 0 : 64 * (wave / RB_T);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, n_mblk * ngt * GM::GROUP_BYTES, UB_RSRC_FLAGS);
    ub_bf16x8 fa[PA][RBW][3];
    ub_bf16x8 fb[2][2][3];
    f32x16 acc[RBW][2];
    auto load_a1 = [&](int slot_a, int byte_off, int rb, int s) __attribute__((always_inline)) {
        fa[slot_a][rb][s] = __builtin_bit_cast(ub_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + ((rb0 + rb) * 3 + s) * 1024, byte_off, 0));
    };
    auto load_b1 = [&](int buf, const unsigned char *src, int gc, int cb, int s) __attribute__((always_inline)) {
        const int j = gc / KS, ks = gc - j * KS;             // tap j reads LDS row (column + 1 - j)
        fb[buf][cb][s] = __builtin_bit_cast(ub_bf16x8, *reinterpret_cast<const ub_u32x4 *>(src + (1 - j) * ROWB + ks * 32 + cb * 32 * ROWB + s * 2 * CK));
    };
    const int x_lane = (col0 + l31) * ROWB + half * 16;
    constexpr int ia6[6] = {0, 1, 0, 2, 1, 0}, ib6[6] = {2, 1, 1, 0, 0, 0};   // (tap split, window split): smallest products first
    constexpr int NM = 12 * RBW;                              // matrix instructions per group

    int mblk, ct, bb;
    tile_coords(tile0, mblk, ct, bb);
    int a_base = mblk * ngt * GM::GROUP_BYTES;
#pragma unroll
    for (int g = 0; g < PA - 1; ++g)
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int s = 0; s < 3; ++s) load_a1(g, a_base + g * GM::GROUP_BYTES, rb, s);

    int q = 0;
    for (int t = 0; t < my_tiles; ++t) {
        const int tl = tile0 + t * nslot;
        tile_coords(tl, mblk, ct, bb);
        a_base = mblk * ngt * GM::GROUP_BYTES;
        int a_next = a_base;                                  // where the tap stream continues: the next tile's m-block
        if (t + 1 < my_tiles) {
            int m2, c2, b2;
            tile_coords(tl + nslot, m2, c2, b2);
            a_next = m2 * ngt * GM::GROUP_BYTES;
        }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
        int g0 = 0;                                           // tap group of this chunk's first group within the tile's stream
#pragma unroll 1
        for (int c = 0; c < ncht; ++c, ++q) {
            const bool noise = c == nxc;
            lds_barrier();                                    // (A) chunk q's rows are in buffer q & 1
            const unsigned char *const src = xs + (q & 1) * GM::X_BYTES + x_lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) load_b1(0, src, 0, k & 1, 2 - (k >> 1));
#pragma unroll
            for (int gc = 0; gc < NGC; ++gc) {
                if (gc >= KS && noise) break;                 // the noise chunk has the offset-0 tap only (wave-uniform)
                const int nlast = noise ? KS : NGC;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int rb = 0; rb < RBW; ++rb) {
                            const int k = (2 * i + cb) * RBW + rb;
                            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gc % PA][rb][ia6[i]], fb[gc & 1][cb][ib6[i]], acc[rb][cb], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (k < 6 && gc + 1 < nlast) load_b1((gc + 1) & 1, src, gc + 1, k & 1, 2 - (k >> 1));
                            if (k >= NM - 3 * RBW) {          // the tap fragments of the group PA - 1 ahead (the stream wraps into the next tile)
                                const int kk = k - (NM - 3 * RBW);
                                int gn = g0 + gc + PA - 1;
                                const int off = gn >= ngt ? a_next + (gn - ngt) * GM::GROUP_BYTES : a_base + gn * GM::GROUP_BYTES;
                                load_a1((gc + PA - 1) % PA, off, kk / 3, kk % 3);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
            g0 += noise ? KS : NGC;
        }
        // ---- epilogue: interleaved into the io tile [channel][q r + ph] (the bias is the stagers': they know a row's channel) ------------
        // (the stagers took the previous tile's outputs out of it behind this tile's first barrier A)
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * MB + 32 * (rb0 + rb) + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int co = m / R, ph = m - co * R;
                const int co_l = co - (mblk * MB) / R;        // slot in the tile (< CPB by construction)
                if (m < m_total) {
                    float *const o = io_lds + co_l * W + (col0 + l31) * R + ph;
                    o[0] = acc[rb][0][r];
                    o[32 * R] = acc[rb][1][r];
                }
            }
    }
    lds_barrier();                                            // (E)
}

// ---- host side -------------------------------------------------------------------------------------------------------------

static int ub_mb(int rate, int c_out) {
    const int rows = rate * c_out;
    return rows >= 256 ? 256 : rows >= 128 ? 128 : 64;
}

bool upsbf_supported(int c_in, int c_out, int rate, int ksize, int vk) {
    if (!(rate == 2 || rate == 8 || rate == 10 || rate == 12)) return false;
    if (c_in % UB_CK || c_in < UB_CK || c_out < 1 || ksize > 2 * rate || ksize < rate) return false;
    if (vk < 0 || vk >= UB_CK) return false;     // + the ones row
    return rate <= ub_mb(rate, c_out);
}

static void ub_dims(int c_in, int c_out, int rate, int vk, int *mb, int *n_mblk, int *ngt) {
    *mb = ub_mb(rate, c_out);
    *n_mblk = (rate * c_out + *mb - 1) / *mb;                 // m-blocks of MB consecutive GEMM rows m = co * rate + ph
    *ngt = (c_in / UB_CK) * 8 + (vk > 0 ? 4 : 0);
}

size_t upsbf_weight_bytes(int c_in, int c_out, int rate, int vk) {
    int mb, n_mblk, ngt;
    ub_dims(c_in, c_out, rate, vk, &mb, &n_mblk, &ngt);
    return (size_t)n_mblk * ngt * (mb / 32) * 3 * 1024;
}

// uw: ConvTranspose1d weight [c_in][c_out][ksize]; nw: the folded noise conv's weight [c_out][nc_k] (or null when vk == 0), nc_stride its
// stride: V row kq feeds phase ph through tap k = kq - ph * nc_stride (decoder.hip's fold); row vk is a row of ONES whose taps are the
// bias (ups bias + noise-conv bias) -- with folded noise rows the bias costs nothing.  Slab: [m-block][chunk][tap j][k step]
// [row block][split][lane][8 bf16]; the noise chunk holds tap j = 0 only.  GEMM row m_local = co_l * rate + ph of m-block mb.
void upsbf_pack_host(const float *uw, const float *nw, const float *bias, int c_in, int c_out, int rate, int ksize, int vk, int nc_k,
                     int nc_stride, std::vector<uint16_t> *out) {
    int mb, n_mblk, ngt;
    ub_dims(c_in, c_out, rate, vk, &mb, &n_mblk, &ngt);
    const int RB_T = mb / 32, nxc = c_in / UB_CK;
    out->assign(upsbf_weight_bytes(c_in, c_out, rate, vk) / 2, 0);
    auto put = [&](size_t group, int rb, int lane, int e, float v) {
        float r = v;
        for (int s = 0; s < 3; ++s) {
            const uint16_t h = bf16_rne(r);
            const uint32_t bits = (uint32_t)h << 16;
            float f;
            memcpy(&f, &bits, 4);
            r -= f;
            (*out)[((group * RB_T + rb) * 3 + s) * 512 + lane * 8 + e] = h;
        }
    };
    for (int mblk = 0; mblk < n_mblk; ++mblk)
        for (int rb = 0; rb < RB_T; ++rb)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = mblk * mb + 32 * rb + (lane & 31);          // GEMM row (co, ph), co-major
                const int co = m / rate, ph = m % rate;
                const bool row_ok = co < c_out;
                for (int c = 0; c < nxc; ++c)
                    for (int j = 0; j < 2; ++j)
                        for (int ks = 0; ks < 4; ++ks)
                            for (int e = 0; e < 8; ++e) {
                                const int ci = UB_CK * c + 16 * ks + 8 * (lane >> 5) + e;
                                const int kk = ph + j * rate;
                                const float v = row_ok && kk < ksize ? uw[((size_t)ci * c_out + co) * ksize + kk] : 0.f;
                                put((size_t)mblk * ngt + c * 8 + j * 4 + ks, rb, lane, e, v);
                            }
                if (vk > 0)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int e = 0; e < 8; ++e) {
                            const int kq = 16 * ks + 8 * (lane >> 5) + e;
                            const int k = kq - ph * nc_stride;
                            float v = row_ok && kq < vk && k >= 0 && k < nc_k ? nw[(size_t)co * nc_k + k] : 0.f;
                            if (row_ok && kq == vk && bias) v = bias[co];     // the ones row (upsbf_kernel's x_issue): the bias rides the GEMM
                            put((size_t)mblk * ngt + nxc * 8 + ks, rb, lane, e, v);
                        }
            }
}

static int ub_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

template <int R, int MB>
static int ub_launch(UbParams p, int batch, hipStream_t stream) {
    using GM = UbGeom<R, MB>;
    if (reserve_whole_cu((const void *)upsbf_kernel<R, MB>, "upsbf")) return 1;
    p.col_tiles = (int)ceil_div(p.L_in + 1, GM::N1);
    p.n_tiles = p.col_tiles * p.n_mblk * batch;
    p.per_xcd = (int)ceil_div(p.n_tiles, 8);
    const int cus = ub_cu_count() / 8 * 8;
    const int slots = (int)std::min<int64_t>(cus / 8, p.per_xcd);
    hipLaunchKernelGGL((upsbf_kernel<R, MB>), dim3((unsigned)(slots * 8)), dim3(UB_NTH), LDS_WHOLE_CU, stream, p);
    RVC_LAUNCH_CHECK();
    return 0;
}

// x [batch][c_in][L_in] -> y [batch][c_out][L_out], L_out = (L_in - 1) rate - 2 pad + ksize given by the caller.
// vk > 0: the noise conv is folded into the GEMM (the slab holds its rows and, as a row of ones, the bias); vk == 0: `bias` (may be
// null) is added on the way out and there is no noise conv.
int launch_upsbf(const float *x, const float *har, int64_t Lh, const void *u, const float *bias, float *y, int batch, int c_in, int c_out,
                 int64_t L_in, int64_t L_out, int rate, int ksize, int pad, int vk, int64_t S, int64_t P, float slope, hipStream_t stream) {
    if (!upsbf_supported(c_in, c_out, rate, ksize, vk)) return fail("upsbf: unsupported shape (%d -> %d channels, rate %d, kernel %d, %d noise rows)", c_in, c_out, rate, ksize, vk);
    if (vk > 0 && (!har || Lh <= 0 || Lh * 4 >= ((int64_t)1 << 31) || S > 4096 || P > 65536)) return fail("upsbf: bad noise source");
    if ((int64_t)c_in * L_in * 4 >= ((int64_t)1 << 31) || (int64_t)c_out * L_out * 4 >= ((int64_t)1 << 31)) return fail("upsbf: slab exceeds the 2 GiB buffer addressing");
    if (L_in <= 0 || batch <= 0) return 0;
    UbParams p;
    p.x = x; p.har = har; p.u = u; p.bias = bias; p.y = y; p.L_in = L_in; p.L_out = L_out; p.Lh = vk > 0 ? Lh : 0;
    p.c_in = c_in; p.c_out = c_out; p.n_xchunks = c_in / UB_CK; p.vk = vk; p.S = (int)S; p.P = (int)P; p.pad = pad; p.slope = slope;
    int mb, ngt;
    ub_dims(c_in, c_out, rate, vk, &mb, &p.n_mblk, &ngt);
#define RVC_UB_CASE(RR, MM) if (rate == RR && mb == MM) return ub_launch<RR, MM>(p, batch, stream)
    RVC_UB_CASE(12, 256); RVC_UB_CASE(10, 256); RVC_UB_CASE(8, 256);
    RVC_UB_CASE(12, 128); RVC_UB_CASE(10, 128); RVC_UB_CASE(8, 128);
    RVC_UB_CASE(2, 256); RVC_UB_CASE(2, 128); RVC_UB_CASE(2, 64);
#undef RVC_UB_CASE
    return fail("upsbf: no instantiation for rate %d with %d GEMM rows per tile", rate, mb);
}

// the noise conv rides the GEMM as (rate - 1) stride + nc_k extra rows (+ a row of ones for the bias) whenever there is one
bool upsbf_fold_noise(int nc_k) { return nc_k > 0; }

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_upsample_bf16x3_weight_bytes(int c_in, int c_out, int rate, int ksize, int nc_k, int nc_stride, size_t *bytes) {
    if (!bytes) return fail("rvc_upsample_bf16x3_weight_bytes: null pointer");
    const int vk = upsbf_fold_noise(nc_k) ? (rate - 1) * nc_stride + nc_k : 0;
    if (!upsbf_supported(c_in, c_out, rate, ksize, vk)) return fail("rvc_upsample_bf16x3_weight_bytes: unsupported shape");
    *bytes = upsbf_weight_bytes(c_in, c_out, rate, vk);
    return 0;
}

extern "C" int rvc_upsample_bf16x3_pack_weight(const float *up_w_host, const float *noise_w_host, const float *bias_host, int c_in, int c_out,
                                               int rate, int ksize, int nc_k, int nc_stride, void *u_dev, void *stream) {
    if (!up_w_host || !u_dev || (nc_k > 0 && !noise_w_host)) return fail("rvc_upsample_bf16x3_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_upsample_bf16x3_weight_bytes(c_in, c_out, rate, ksize, nc_k, nc_stride, &bytes)) return 1;
    const int vk = upsbf_fold_noise(nc_k) ? (rate - 1) * nc_stride + nc_k : 0;
    std::vector<uint16_t> u;
    upsbf_pack_host(up_w_host, noise_w_host, bias_host, c_in, c_out, rate, ksize, vk, nc_k, nc_stride, &u);
    hipError_t e = hipMemcpyAsync(u_dev, u.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_upsample_bf16x3_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_upsample_bf16x3_forward(const float *x_dev, const float *har_dev, int64_t har_len, const void *u_dev, const float *bias_dev,
                                           float *y_dev, int batch, int c_in, int c_out, int64_t length_in, int rate, int ksize, int pad,
                                           int nc_k, int nc_stride, int nc_pad, float slope_in, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_upsample_bf16x3_forward: null pointer");
    if (pad < 0 || nc_k < 0 || (nc_k > 0 && nc_stride < 1)) return fail("rvc_upsample_bf16x3_forward: bad argument");
    const int vk = upsbf_fold_noise(nc_k) ? (rate - 1) * nc_stride + nc_k : 0;
    const int64_t l_out = (length_in - 1) * rate - 2 * pad + ksize;
    if (l_out <= 0) return fail("rvc_upsample_bf16x3_forward: empty output");
    return launch_upsbf(x_dev, har_dev, har_len, u_dev, bias_dev, y_dev, batch, c_in, c_out, length_in, l_out, rate, ksize, pad, vk,
                        (int64_t)rate * nc_stride, (int64_t)pad * nc_stride + nc_pad, slope_in, (hipStream_t)stream);
}
