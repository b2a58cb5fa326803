// K10b -- the 3x3 convs of RMVPE's U-Net (RMVPE.py:13-287: ConvBlockRes) on the bf16 matrix cores with fp32-exact operands:
//
//     y[co][t][w] = act( sum_{dh, dw, ci} W[co][ci][dh][dw] * x[ci][t + dh - 1][w + dw - 1] + bias[co] ) + res[co][t][w]
//
// conv2d.hip (K10) runs these ~117 convs of 1.77 GFLOP each as an fp32 implicit GEMM on v_mfma_f32_32x32x2_f32: 8 x 64 matrix
// cycles per 16 input channels of a tap and 32 x 32 tile, one accumulator chain per wave, 24-60 us per conv whatever the level.
// Here every fp32 operand is split EXACTLY into three bf16 (x = x0 + x1 + x2) and the six products of order <= 2^-16 go through
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms < 2^-23 of a product): 6 x 32 cycles for the same 16 channels.
//   * the zero-padded input patch of a pixel tile ((BN / W + 2) rows x (W + 2) columns, BN pixels = whole rows of the map) sits
//     in LDS as bf16 triples in [position][split][channel 16] order, 112 bytes per position (an odd multiple of 16): the window
//     fragment of tap (dh, dw) is ONE conflict-free 16-byte read at position + dh (W + 2) + dw -- no im2col, no transform;
//   * the input channels come in chunks of 16 (one k step per tap) through a two-buffer ring that the four STAGER waves fill
//     (HBM -> registers an item ahead -> split -> LDS) while the four COMPUTE waves multiply the previous chunk (resblock_bf.hip's
//     division of labour); one barrier per chunk;
//   * a compute wave owns 32 output channels x 64 pixels (two accumulators: twelve independent-in-pairs matrix instructions per
//     tap); its tap fragments (taps split at load, 3 KiB per tap and chunk) go L2 -> registers through a ring of NINE taps -- a tap's
//     registers take the next chunk's (or tile's) fragments as soon as its products are issued: one whole item of lead -- and never
//     touch LDS (a ring of three, two taps of lead: equal on the C -> C convs, 3-5 % slower on the 2 C -> C ones);
//   * block = 32 MW channels x 64 (4 / MW) pixels, MW = 1 / 2 / 4 for 32 / 64 / >= 128 (padded) output channels; the workgroup
//     owns its CU (common.h) and walks units = (pixel tile, channel block) persistently when the map has more of them than the chip
//     has CUs (levels 0 and 1: 1504 / 376 units), deep levels split K over gridDim.y so that ~190-250 workgroups run, and
//     conv2d.hip's fixed-order finish pass sums the partials (deterministic -- no atomics) and applies the epilogue.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

typedef __bf16 c2b_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c2b_bf16x2 __attribute__((ext_vector_type(2)));
typedef float c2b_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned c2b_u32x4 __attribute__((ext_vector_type(4)));

constexpr int C2B_NTH = 512;
constexpr int C2B_RSRC = 0x00020000;
constexpr unsigned C2B_OOB = 0x80000000u;   // beyond every tensor this kernel takes: loads return 0
constexpr int C2B_CK = 16;                  // input channels per chunk = one 16-deep k step per tap
constexpr int C2B_ROWB = 112;               // [split 3][channel 16] bf16 + 16 bytes
constexpr int C2B_TAPS = 9;
#ifndef C2B_STAGER_PRIO
#define C2B_STAGER_PRIO 0
#define C2B_COMPUTE_PRIO 1
#endif
constexpr bool C2B_RING9 = true;          // tap-fragment ring of nine taps (one item of lead: L2 / HBM latency) instead of three (two taps)
constexpr int C2B_GROUP = 3 * 1024;         // tap fragments of one (32-row block, chunk, tap): [split 3][lane 64][8 bf16]
static_assert((C2B_ROWB / 16) % 2 == 1, "row stride must be an odd multiple of 16 bytes");

struct C2bParams {
    const float *x = nullptr;        // [batch][c_in][H][W]
    const void *u = nullptr;         // conv2dbf_pack_host's slab [m32][chunk][tap][split][lane][8]
    const float *bias = nullptr;     // [c_out] or null
    const float *res = nullptr;      // [batch][c_out][H][W] or null: added AFTER the activation
    float *y = nullptr;              // [batch][c_out][H][W]
    float *partial = nullptr;        // [split][batch][c_out][H * W] when split > 1
    int c_in = 0, c_out = 0, H = 0, W = 0, log2w = 0, batch = 1, relu = 0;
    int split = 1, chunks_per_split = 0, n_chunks_total = 0;
    int n_px_tiles = 0, n_mblk = 0, n_units = 0;
    int per_xcd = 0;                 // > 0: units per XCD (persistent launches, grid a multiple of 8); 0: unit = workgroup id + k * grid
    unsigned u_bytes = 0;
};

constexpr int c2b_patch_max(int bn) {
    int best = 0;
    for (int w = 4; w <= 128 && w <= bn; w *= 2) {
        const int v = (bn / w + 2) * (w + 2);
        best = v > best ? v : best;
    }
    return best;
}

template <int MW>
struct C2bGeom {
    static constexpr int NW = 4 / MW;                          // compute waves along the pixels
    static constexpr int BM = 32 * MW, BN = 64 * NW;
    static constexpr int PMAX = c2b_patch_max(BN);             // positions of the largest patch over the row lengths 4 .. min(BN, 128)
    static constexpr int NJ = (BN + 2 * (BN < 128 ? BN : 128) + 255) / 256;   // interior positions ((BN / W + 2) W, W <= min(BN, 128)) per stager thread
    static constexpr int X_BYTES = (PMAX + 1) * C2B_ROWB;      // + one row that takes the writes of threads beyond the patch
    static constexpr int IO_BYTES = BM * BN * 4;               // the finished tile [channel][pixel]
    static constexpr int LDS_BYTES = 2 * X_BYTES + IO_BYTES;
    static_assert(X_BYTES % 16 == 0, "");
    static_assert(MW == 1 || MW == 2 || MW == 4, "");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
};

__device__ __forceinline__ float c2b_sub_np(float a, float b) {   // plain fp32 VALU next to another wave's matrix instructions
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a, b) -> three words of two bf16 each whose sums are a and b exactly
__device__ __forceinline__ void c2b_split3(float a, float b, unsigned w[3]) {
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(c2b_f32x2{a, b}, c2b_bf16x2));
        w[level] = ww;
        if (level < 2) {
            a = c2b_sub_np(a, __uint_as_float(ww << 16));
            b = c2b_sub_np(b, __uint_as_float(ww & 0xffff0000u));
        }
    }
}

// offset of a buffer operation that may have to be a no-op: a select, never a branch
__device__ __forceinline__ unsigned c2b_sel(bool ok, unsigned off) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(C2B_OOB), "v"(off), "s"((unsigned long long)__builtin_amdgcn_ballot_w64(ok)));
    return r;
}

// the six products of a multiply-add, smallest first: (tap split, window split) = (0,2) (1,1) (2,0) (0,1) (1,0) (0,0)
__host__ __device__ constexpr int c2b_sa(int pr) { return pr < 3 ? pr : (pr == 4 ? 1 : 0); }
__host__ __device__ constexpr int c2b_sb(int pr) { return pr < 3 ? 2 - pr : (pr == 3 ? 1 : 0); }

// DBG (ablation build only, RVC_C2B_DEBUG): 1 no matrix instructions (wrong results), 64 cycle stamps (tools/stamp_conv2dbf.py)
// RES: the conv has ONE chunk (16 input channels) and one 32-row block: its 27 tap fragments stay in registers for the whole launch
template <int MW, int DBG = 0, bool RES = false>
__global__ void __launch_bounds__(C2B_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv2dbf_kernel(const C2bParams p) {
    using GM = C2bGeom<MW>;
    constexpr int BN = GM::BN, NJ = GM::NJ, ROWB = C2B_ROWB, TAPS = C2B_TAPS, PA = (RES || C2B_RING9) ? TAPS : 3;
    static_assert(TAPS % PA == 0, "a tap's ring slot must be a compile-time register index");

    extern __shared__ __attribute__((aligned(16))) unsigned char c2b_smem[];
    unsigned char *const xs = c2b_smem;                       // [2][X_BYTES]
    float *const io = reinterpret_cast<float *>(c2b_smem + 2 * GM::X_BYTES);   // [BM][BN]: a finished tile on its way out

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int W = p.W, H = p.H, HW = H * W, lw = p.log2w;
    const int th = BN >> lw, PC = W + 2;
    const int n_chunks = p.chunks_per_split, c_begin = (int)blockIdx.y * n_chunks;
    const int G = (int)gridDim.x;
    constexpr int dbg = DBG;
    // Unit u0 + k ustep is this workgroup's k-th.  A map with more units than CUs (p.per_xcd > 0: the grid is a multiple of 8) gives
    // every XCD -- workgroup ids go round the eight of them -- a CONTIGUOUS range of units that its workgroups walk side by side:
    // the two halo rows a pixel tile shares with each neighbour are then in that XCD's L2 (ids b, b + 1 sit on different XCDs: with
    // unit = b + k G the halo came from HBM a third of the time, profiles/r06_pmc_conv2dbf.txt).
    int u0 = (int)blockIdx.x, ustep = G, my_units = (p.n_units - (int)blockIdx.x + G - 1) / G;
    if (p.per_xcd > 0) {
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        ustep = G >> 3;
        u0 = xcd * p.per_xcd + slot;
        const int end = (xcd + 1) * p.per_xcd < p.n_units ? (xcd + 1) * p.per_xcd : p.n_units;
        my_units = u0 < end ? (end - u0 + ustep - 1) / ustep : 0;
    }
    if (my_units <= 0) return;
    const int n_items = my_units * n_chunks;                  // item i = (this workgroup's unit i / n_chunks, chunk c_begin + i % n_chunks)
    auto decode = [&](int k, int &b, int &pxt, int &mblk) __attribute__((always_inline)) {
        const int unit = u0 + k * ustep;
        const int rest = unit / p.n_mblk;
        mblk = unit - rest * p.n_mblk;                        // the channel blocks of one pixel tile are neighbours: the patch comes from L2
        b = rest / p.n_px_tiles;
        pxt = rest - b * p.n_px_tiles;
    };

    // both patch buffers start as zeros (the padding columns stay that way)
    for (int o = tid * 16; o < 2 * GM::X_BYTES; o += C2B_NTH * 16) *reinterpret_cast<c2b_u32x4 *>(xs + o) = c2b_u32x4{0u, 0u, 0u, 0u};
    lds_barrier();

    // DBG & 64 (ablation build): wave 0 and wave 4 write cycle-counter stamps to `partial` (unsplit launches): [block][role 2][64]
    unsigned long long *const stamps = (DBG & 64) ? reinterpret_cast<unsigned long long *>(p.partial) + ((size_t)blockIdx.x * 2 + (wave >= 4 ? 1 : 0)) * 64 : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr ((DBG & 64) != 0) {
            if ((wave == 0 || wave == 4) && lane == 0 && n_stamp < 64) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();

    if (wave >= 4) {
        // ============================================ stagers: HBM -> LDS ================================================================
        __builtin_amdgcn_s_setprio(C2B_STAGER_PRIO);
        const int st = tid - 256;
        // The tile is whole rows of the map, so the patch's first and last columns are the conv's zero padding for every tile: they are
        // zeroed once (above) and only the (th + 2) W interior positions are staged -- interior position q = st + 256 j is row q / W,
        // column q % W of the patch's interior, element (q - W) of the plane relative to the tile's first pixel.
        const int n_int = (th + 2) * W;
        int prow[NJ];
        unsigned ldsoff[NJ], inside = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = st + 256 * j;
            const int pr = q >> lw, pc = q & (W - 1);
            prow[j] = pr - 1;
            if (q < n_int) inside |= 1u << j;
            ldsoff[j] = (unsigned)((q < n_int ? pr * PC + pc + 1 : GM::PMAX) * ROWB);
        }
        float xr[NJ][16];
        const int num_bytes = p.c_in * HW * 4;
        // (Measured, profiles/r06_conv2dbf_stamps.txt: issuing the SAME sequence of memory operations in every phase -- no-op offsets where an
        // item has no store or skip path -- makes the compiler's s_waitcnt counts exact instead of draining the queue, and is slower on
        // every multi-chunk shape: the no-op requests still cost their turn in the address pipeline.  The loaded latency of this
        // access pattern is ~7 k cycles at level 0, where the kernel moves ~3.3 TB/s through L2.)
        auto x_issue = [&](int i) __attribute__((always_inline)) {
            const bool valid = i < n_items;
            const int k = i / n_chunks, c = i - k * n_chunks;
            int b, pxt, mblk;
            decode(k, b, pxt, mblk);
            const int t0 = pxt * th;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (valid ? (int64_t)b * p.c_in * HW : 0)), 0, num_bytes, C2B_RSRC);
            const int ch0 = (c_begin + c) * C2B_CK;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool ok = (int)valid & (int)((inside >> j) & 1) & (int)((unsigned)(t0 + prow[j]) < (unsigned)H);   // (no short circuit: no branches)
                const unsigned vo = c2b_sel(ok, (unsigned)((t0 - 1) * W + st + 256 * j) * 4u);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    xr[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, (ch0 + e) * HW * 4, 0));
            }
        };
        auto x_write = [&](int i) __attribute__((always_inline)) {
            unsigned char *const xb = xs + (i & 1) * GM::X_BYTES;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    unsigned w[4][3];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) c2b_split3(xr[j][hf * 8 + 2 * e2], xr[j][hf * 8 + 2 * e2 + 1], w[e2]);
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        *reinterpret_cast<c2b_u32x4 *>(xb + ldsoff[j] + s * 32 + hf * 16) = c2b_u32x4{w[0][s], w[1][s], w[2][s], w[3][s]};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- the finished tile: act(io tile + bias) + skip path -> HBM, 16 bytes per lane, whole rows of the tile (= consecutive pixels of
        //      the plane); a K split stores the bare partial sums
        constexpr int LPR = BN / 4, RPP = 256 / LPR, PASSES = GM::BM / RPP;
        const int orow = st / LPR, ocol = (st % LPR) * 4;
        const bool whole = p.split == 1;
        const bool has_res = p.res != nullptr && whole, has_bias = p.bias != nullptr && whole, relu = p.relu && whole;
        const int out_bytes = p.c_out * HW * 4;
        f32x4 rres[PASSES];                                  // a unit's skip-path values and bias, requested a phase before they are added
        float rbias[PASSES];
        auto out_offset = [&](bool valid, int mblk, int pxt, int ps) __attribute__((always_inline)) -> unsigned {
            const int co = mblk * GM::BM + ps * RPP + orow, pix = pxt * th * W + ocol;
            return c2b_sel((int)valid & (int)(co < p.c_out) & (int)(pix < HW), (unsigned)(co * HW + pix) * 4u);
        };
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)(has_bias ? p.bias : p.x), 0, p.c_out * 4, C2B_RSRC);
        auto epi_issue = [&](int k, bool valid) __attribute__((always_inline)) {
            int b, pxt, mblk;
            decode(k, b, pxt, mblk);
            const bool vr = (int)valid & (int)has_res;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(vr ? p.res + (int64_t)b * p.c_out * HW : p.x), 0, out_bytes, C2B_RSRC);
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps)
                rres[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)out_offset(vr, mblk, pxt, ps), 0, 0));
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int co = mblk * GM::BM + ps * RPP + orow;
                rbias[ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (int)c2b_sel((int)valid & (int)has_bias & (int)(co < p.c_out), (unsigned)co * 4u), 0, 0));
            }
        };
        auto out_store = [&](int k, bool valid) __attribute__((always_inline)) {
            int b, pxt, mblk;
            decode(k, b, pxt, mblk);
            float *const dst = !valid ? p.y : (whole ? p.y + (int64_t)b * p.c_out * HW : p.partial + ((int64_t)((int)blockIdx.y * p.batch + b) * p.c_out) * HW);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, out_bytes, C2B_RSRC);
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(io + (ps * RPP + orow) * BN + ocol);
                v += rbias[ps];
                if (relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                v += rres[ps];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c2b_u32x4, v), rs, (int)out_offset(valid, mblk, pxt, ps), 0, 0);
            }
        };
        // Per phase, behind barrier A(i) (item i's rows are complete in buffer i & 1, the compute waves are done with the other one, and
        // -- first item of a unit -- the previous unit's accumulators are in the io tile):
        //   write item i + 1 (requested a phase ago) into buffer (i + 1) & 1; request item i + 2; store the previous unit (its bias and
        //   skip-path values were requested a phase ago); request this unit's.
        // A wave's memory operations retire in order: whatever a phase consumes was requested a whole phase earlier, and nothing but
        // older requests stands before it in the queue.
        // A unit of ONE chunk writes its io tile in the phase the previous unit's is drained: barrier B orders the two.
        x_issue(0);
        x_write(0);
        if (1 < n_items) x_issue(1);
        for (int i = 0; i < n_items; ++i) {
            const int k = i / n_chunks, c = i - k * n_chunks;
            lds_barrier();                                    // (A)
            stamp();
            if (i + 1 < n_items) x_write(i + 1);
            stamp();
            if (i + 2 < n_items) x_issue(i + 2);
            stamp();
            if (c == 0 && i > 0) out_store(k - 1, true);      // its bias / skip-path values were requested a phase ago
            if (c == n_chunks - 1) epi_issue(k, true);
            stamp();
            if (n_chunks == 1) lds_barrier();                 // (B)
            stamp();
        }
        lds_barrier();                                        // (E) the last unit's accumulators are in the io tile
        out_store(my_units - 1, true);
        return;
    }

    // ================================================ compute waves ==========================================================
    __builtin_amdgcn_s_setprio(C2B_COMPUTE_PRIO);
    const int mw = wave % MW, nw = wave / MW;
    int xl[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int pn = nw * 64 + cb * 32 + l31;               // this lane's pixel of the tile, and its position in the patch (tap (0, 0))
        const int prw = pn >> lw, pcl = pn & (W - 1);
        xl[cb] = (prw * PC + pcl) * ROWB + half * 16;
    }
    const int rowsh1 = PC * ROWB;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)p.u_bytes, C2B_RSRC);
    c2b_bf16x8 fa[PA][3];
    c2b_bf16x8 fb[2][2][3];
    f32x16 acc[2];
    // tap fragments of group (item base, tap): three splits
    auto load_a = [&](int slot, unsigned vo, int base, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
            fa[slot][s] = __builtin_bit_cast(c2b_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, (int)vo + s * 1024, base + tap * C2B_GROUP, 0));
    };
    auto load_b1 = [&](int buf, const unsigned char *src, int tap, int cb, int s) __attribute__((always_inline)) {
        const int dh = tap / 3, dw = tap - 3 * dh;
        fb[buf][cb][s] = __builtin_bit_cast(c2b_bf16x8, *reinterpret_cast<const c2b_u32x4 *>(src + xl[cb] + dh * rowsh1 + dw * ROWB + s * 32));
    };
    auto item_base = [&](int mblk, int c) __attribute__((always_inline)) -> int {
        return (((mblk * MW + mw) * p.n_chunks_total + c_begin + c) * TAPS) * C2B_GROUP;
    };

    int k = 0, c = 0, b, pxt, mblk;
    decode(0, b, pxt, mblk);
    int base_cur = item_base(mblk, 0);
    const unsigned vo_lane = 16u * (unsigned)lane;
    if constexpr (PA == TAPS) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) load_a(tap, vo_lane, base_cur, tap);
    } else {
        load_a(0, vo_lane, base_cur, 0);
        load_a(1, vo_lane, base_cur, 1);
    }

    for (int i = 0; i < n_items; ++i) {
        // the item behind this one (its first two taps are requested under this item's last two)
        int kn = k, cn = c + 1;
        if (cn == n_chunks) { cn = 0; kn = k + 1; }
        int bnx = b, pxtn = pxt, mblkn = mblk;
        if (cn == 0 && i + 1 < n_items) decode(kn, bnx, pxtn, mblkn);
        const int base_next = item_base(mblkn, cn);
        const unsigned vo_next = i + 1 < n_items ? vo_lane : C2B_OOB;
        if (c == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
        }
        lds_barrier();                                        // (A) item i's rows are in buffer i & 1
        stamp();
        const unsigned char *const src = xs + (i & 1) * GM::X_BYTES;
#pragma unroll
        for (int q = 0; q < 6; ++q) load_b1(0, src, 0, q & 1, 2 - (q >> 1));
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            // twelve matrix instructions, smallest products first; behind instruction q < 6, pinned: one of the NEXT tap's six window
            // fragments (split 2 first); behind the last three: the tap fragments of the tap two ahead
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int q = 2 * pr + cb;
                    if (!(dbg & 1)) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tap % PA][c2b_sa(pr)], fb[tap & 1][cb][c2b_sb(pr)], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < 6 && tap + 1 < TAPS) load_b1((tap + 1) & 1, src, tap + 1, q & 1, 2 - (q >> 1));
                    if (q == 11 && !RES) {
                        if constexpr (PA == TAPS) {
                            load_a(tap, vo_next, base_next, tap);    // a ring of nine: this tap's registers take the NEXT item's tap -- a whole item of lead
                        } else {
                            if (tap + 2 < TAPS) load_a((tap + 2) % PA, vo_lane, base_cur, tap + 2);
                            else load_a((tap + 2) % PA, vo_next, base_next, tap + 2 - TAPS);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        stamp();
        if (c == n_chunks - 1) {
            // ---- epilogue: the accumulators into the io tile (bias, ReLU, skip path and the stores are the stagers')
            if (n_chunks == 1) lds_barrier();                 // (B) the stagers have taken the previous unit out of the io tile
            stamp();
            float *const io_mine = io + (mw * 32 + 4 * half) * BN + nw * 64 + l31;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) io_mine[((r & 3) + 8 * (r >> 2)) * BN + cb * 32] = acc[cb][r];
        }
        stamp();
        k = kn; c = cn; b = bnx; pxt = pxtn; mblk = mblkn;
        base_cur = base_next;
    }
    lds_barrier();                                            // (E)
}

// ---- host side -------------------------------------------------------------------------------------------------------------

static int c2b_mw(int c_out) {
    const int m_pad = (c_out + 31) / 32 * 32;
    if (m_pad == 32) return 1;
    if (m_pad == 64) return 2;
    if (m_pad % 128 == 0) return 4;       // (two waves per 32-row block sharing their tap fragments through L1 -- MW = 2 here -- is 5 % slower)
    return 0;
}

bool conv2dbf_supported(int c_in, int c_out, int H, int W, int taps) {
    if (taps != 9 || c_in <= 0 || c_in % C2B_CK || c_out <= 0 || H <= 0) return false;
    if (W < 4 || W > 128 || (W & (W - 1))) return false;
    const int mw = c2b_mw(c_out);
    if (!mw || W > 64 * (4 / mw)) return false;
    const int m_pad = (c_out + 31) / 32 * 32;
    if ((int64_t)c_in * H * W >= ((int64_t)1 << 29) || (int64_t)c_out * H * W >= ((int64_t)1 << 29)) return false;
    return conv2dbf_weight_bytes(c_out, c_in) < ((size_t)1 << 31) && m_pad <= 4096;
}

size_t conv2dbf_weight_bytes(int c_out, int c_in) { return (size_t)((c_out + 31) / 32) * (c_in / C2B_CK) * C2B_TAPS * C2B_GROUP; }

// w: [c_out][c_in][3][3] (torch) -> [32-row block][chunk][tap][split][lane][8 bf16]: lane l of a fragment holds output channel
// 32 rb + (l & 31), input channels 16 chunk + 8 (l >> 5) .. + 7; rows beyond c_out are zero
void conv2dbf_pack_host(const float *w, int c_out, int c_in, std::vector<uint16_t> *out) {
    const int RB = (c_out + 31) / 32, NC = c_in / C2B_CK;
    out->assign(conv2dbf_weight_bytes(c_out, c_in) / 2, 0);
    for (int rb = 0; rb < RB; ++rb)
        for (int ch = 0; ch < NC; ++ch)
            for (int tap = 0; tap < C2B_TAPS; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = 32 * rb + (lane & 31), ci = C2B_CK * ch + 8 * (lane >> 5) + e;
                        if (co >= c_out) continue;
                        float r = w[((size_t)co * c_in + ci) * C2B_TAPS + tap];
                        const size_t group = ((size_t)rb * NC + ch) * C2B_TAPS + tap;
                        for (int s = 0; s < 3; ++s) {
                            const uint16_t h = bf16_rne(r);
                            const uint32_t bits = (uint32_t)h << 16;
                            float f;
                            memcpy(&f, &bits, 4);
                            r -= f;                          // exact in fp32
                            (*out)[(group * 3 + s) * 512 + lane * 8 + e] = h;
                        }
                    }
}

static bool c2b_resident() {
    static const int on = knob("RVC_C2B_RES", 1);
    return on != 0;
}

static bool c2b_xcd_ranges() {
    static const int on = knob("RVC_C2B_XCD", 1);
    return on != 0;
}

static int c2b_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

// units (pixel tiles x channel blocks x batch), and how many ways K is split so that about one workgroup per CU runs
static void c2b_plan(int batch, int c_in, int c_out, int H, int W, int *units, int *split, int *n_px_tiles, int *n_mblk) {
    const int mw = c2b_mw(c_out), bn = 64 * (4 / mw), th = bn / W;
    const int m_pad = (c_out + 31) / 32 * 32;
    *n_px_tiles = (int)ceil_div(H, th);
    *n_mblk = m_pad / (32 * mw);
    *units = batch * *n_px_tiles * *n_mblk;
    const int nct = c_in / C2B_CK, cus = c2b_cu_count();
    int s = 1;
    while (s * 2 <= 8 && *units * s * 2 <= cus && nct % (s * 2) == 0 && nct / (s * 2) >= 2) s *= 2;
    *split = s;
}

size_t conv2dbf_workspace_bytes(int batch, int c_in, int c_out, int H, int W) {
    int units, split, npt, nmb;
    c2b_plan(batch, c_in, c_out, H, W, &units, &split, &npt, &nmb);
    return split > 1 ? (size_t)split * batch * c_out * H * W * sizeof(float) : 0;
}

template <int MW, int DBG = 0>
static int c2b_launch1(const C2bParams &p, int grid_x, hipStream_t stream) {
    if constexpr (MW == 1 && DBG == 0) {
        if (p.n_chunks_total == 1 && p.n_mblk == 1 && c2b_resident()) {
            if (reserve_whole_cu((const void *)conv2dbf_kernel<1, 0, true>, "conv2d bf16x3")) return 1;
            hipLaunchKernelGGL((conv2dbf_kernel<1, 0, true>), dim3((unsigned)grid_x, 1, 1), dim3(C2B_NTH), LDS_WHOLE_CU, stream, p);
            RVC_LAUNCH_CHECK();
            return 0;
        }
    }
    if (reserve_whole_cu((const void *)conv2dbf_kernel<MW, DBG>, "conv2d bf16x3")) return 1;
    hipLaunchKernelGGL((conv2dbf_kernel<MW, DBG>), dim3((unsigned)grid_x, (unsigned)p.split, 1), dim3(C2B_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

template <int MW>
static int c2b_launch(const C2bParams &p, int grid_x, hipStream_t stream) {
#ifdef RVC_ABLATE
    static const int dbg = knob("RVC_C2B_DEBUG", 0);
    switch (dbg) {
    case 1: return c2b_launch1<MW, 1>(p, grid_x, stream);
    case 64: return c2b_launch1<MW, 64>(p, grid_x, stream);
    default: break;
    }
#endif
    return c2b_launch1<MW, 0>(p, grid_x, stream);
}

int launch_conv2dbf(const float *x, const void *u, const float *bias, const float *res, float *y, int batch, int c_in, int c_out, int H, int W,
                    int relu, float *ws, size_t ws_bytes, hipStream_t stream) {
    if (!conv2dbf_supported(c_in, c_out, H, W, 9)) return fail("conv2d bf16x3: shape %d -> %d channels, %d x %d unsupported", c_in, c_out, H, W);
    if (batch <= 0) return 0;
    C2bParams p;
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.y = y;
    p.c_in = c_in; p.c_out = c_out; p.H = H; p.W = W; p.batch = batch; p.relu = relu;
    p.log2w = __builtin_ctz((unsigned)W);

    p.n_chunks_total = c_in / C2B_CK;
    p.u_bytes = (unsigned)conv2dbf_weight_bytes(c_out, c_in);
    c2b_plan(batch, c_in, c_out, H, W, &p.n_units, &p.split, &p.n_px_tiles, &p.n_mblk);
    p.chunks_per_split = p.n_chunks_total / p.split;
    if (p.split > 1) {
        const size_t need = (size_t)p.split * batch * c_out * H * W * sizeof(float);
        if (!ws || ws_bytes < need) return fail("conv2d bf16x3: workspace of %zu bytes needed, %zu given", need, ws_bytes);
        p.partial = ws;
    }
#ifdef RVC_ABLATE
    if (knob("RVC_C2B_DEBUG", 0) == 64 && p.split == 1) p.partial = ws;   // stamps (tools/stamp_conv2dbf.py hands over a workspace)
#endif
    const int cus = c2b_cu_count();
    const int rounds = (int)ceil_div(p.n_units, cus);
    int grid_x = (int)ceil_div(p.n_units, rounds);            // every workgroup walks `rounds` units (the last ones one fewer)
    if (rounds > 1 && p.split == 1 && c2b_xcd_ranges()) {
        p.per_xcd = (int)ceil_div(p.n_units, 8);
        grid_x = (int)ceil_div(ceil_div(p.per_xcd, rounds), 1) * 8;   // ceil(per_xcd / rounds) workgroups on each of the 8 XCDs
    }
    const int mw = c2b_mw(c_out);
    int rc;
    if (mw == 1) rc = c2b_launch<1>(p, grid_x, stream);
    else if (mw == 2) rc = c2b_launch<2>(p, grid_x, stream);
    else rc = c2b_launch<4>(p, grid_x, stream);
    if (rc) return rc;
    if (p.split > 1) return launch_conv2d_finish(p.partial, p.split, batch, c_out, H, W, bias, res, relu, y, stream);
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv2d_bf16x3_weight_bytes(int c_out, int c_in, int kh, int kw, size_t *bytes) {
    if (!bytes) return fail("rvc_conv2d_bf16x3_weight_bytes: null pointer");
    if (kh != 3 || kw != 3 || c_out <= 0 || c_in <= 0 || c_in % C2B_CK || !c2b_mw(c_out))
        return fail("rvc_conv2d_bf16x3_weight_bytes: 3x3 kernels, c_in a multiple of %d, c_out <= 64 or a multiple of 128 (got %d -> %d, %dx%d)",
                    C2B_CK, c_in, c_out, kh, kw);
    *bytes = conv2dbf_weight_bytes(c_out, c_in);
    return 0;
}

extern "C" int rvc_conv2d_bf16x3_pack_weight(const float *w_host, int c_out, int c_in, int kh, int kw, void *u_dev, void *stream) {
    if (!w_host || !u_dev) return fail("rvc_conv2d_bf16x3_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_conv2d_bf16x3_weight_bytes(c_out, c_in, kh, kw, &bytes)) return 1;
    std::vector<uint16_t> u;
    conv2dbf_pack_host(w_host, c_out, c_in, &u);
    hipError_t e = hipMemcpyAsync(u_dev, u.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_conv2d_bf16x3_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv2d_bf16x3_workspace_bytes(int batch, int c_in, int c_out, int height, int width, size_t *out) {
    if (!out) return fail("rvc_conv2d_bf16x3_workspace_bytes: null pointer");
    if (batch <= 0 || !conv2dbf_supported(c_in, c_out, height, width, 9))
        return fail("rvc_conv2d_bf16x3_workspace_bytes: shape %d -> %d channels, %d x %d unsupported", c_in, c_out, height, width);
    *out = conv2dbf_workspace_bytes(batch, c_in, c_out, height, width);
    return 0;
}

extern "C" int rvc_conv2d_bf16x3_supported(int c_in, int c_out, int height, int width) {
    return conv2dbf_supported(c_in, c_out, height, width, 9) ? 1 : 0;
}

extern "C" int rvc_conv2d_bf16x3_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *y_dev,
                                         int batch, int c_in, int c_out, int height, int width, int relu, void *workspace_dev,
                                         size_t workspace_bytes, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_conv2d_bf16x3_forward: null pointer");
    if (x_dev == y_dev) return fail("rvc_conv2d_bf16x3_forward: y must not alias x");
    return launch_conv2dbf(x_dev, u_dev, bias_dev, res_dev, y_dev, batch, c_in, c_out, height, width, relu, (float *)workspace_dev,
                           workspace_bytes, (hipStream_t)stream);
}
