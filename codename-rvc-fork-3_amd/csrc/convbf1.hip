// K3d -- one square ResBlock conv with bf16-VALUED taps (BASELINE cfg 4: "MRF-HiFi-GAN, bf16 weights, alt ResBlock kernel path") in
// DIRECT form on the bf16 matrix cores, for the layers the fused pair (resblock_bf.hip, K3f) cannot hold in LDS -- C = 256 (every
// tap count) and C = 128 with 11 taps:
//     y = out_scale * ( conv_d( leaky(x) ) + bias [+ res] [+ running sum] )                 (one conv of residuals.py:75-86 /
//                                                                                             MRFLayer.forward, hifigan_mrf.py:13-83)
// A bf16-valued tap is the first term of its own exact split (w = w_0), so w x = w_0 x_0 + w_0 x_1 + w_0 x_2 with the ACTIVATIONS
// split exactly into three bf16: three products per multiply-add, 16 / 16 x 3 = 3.0 matrix products per conv multiply-add.  The
// Winograd form these layers run otherwise (winobf2.hip, K3y: 0.477 x 6 = 2.86 products, a transformed tap needs its full three-way
// split whatever the tap's own precision) pays for the same matrix work with an input transform per point, a seven-wave output
// transform through LDS, 1 KiB of tap fragments per 4 matrix instructions and one workgroup per 256 output samples whatever the
// row length (stage 0 of a 30 s clip: 300 workgroups = 1.17 rounds of the 256 CUs).  Here:
//   * no transform: the x chunk sits in LDS as bf16 triples in [time][split][channel] order (resblock_bf.hip's layout), the window
//     fragment of ANY tap and dilation is one conflict-free 16-byte read at (column + tap d);
//   * one 1 KiB tap fragment per 6 (C = 256: per 6, two row blocks per wave share the window fragments) matrix instructions;
//   * persistent: one 8-wave workgroup per CU walks (time tile) x (64 columns, ALL output channels) tiles -- stage 0 is 600 tiles,
//     2.3 rounds, no tail; the input channels come in chunks of 64 through a two-buffer LDS ring that the four STAGER waves fill
//     (HBM -> registers a chunk ahead -> leaky ReLU -> split -> LDS) while the four COMPUTE waves multiply the previous chunk;
//   * the compute waves issue no memory operation but their tap-fragment loads (L2); outputs leave through an LDS tile
//     [channel][column] that the stagers drain with 16-byte row stores, adding residual / running sum / scale on the way
//     (resblock_bf.hip's division of labour: a wave's memory operations retire in order).
// One barrier per 64-channel chunk, two more per tile.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "conv.h"

namespace rvc {

typedef __bf16 cb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cb_u32x4 __attribute__((ext_vector_type(4)));

constexpr int CB1_NTH = 512;
constexpr int CB1_RSRC_FLAGS = 0x00020000;
constexpr unsigned CB1_OOB = 0x80000000u;   // beyond every tensor this kernel takes: loads return 0, stores are dropped
constexpr int CB1_CK = 64;                  // input channels per chunk
constexpr int CB1_N1 = 64;                  // output columns per tile

struct Cb1Params {
    const float *x = nullptr;        // [batch][C][L]
    const void *u = nullptr;         // convbf1_pack_host's slab
    const float *bias = nullptr;     // [C] or null
    const float *res = nullptr;      // [batch][C][L] or null
    const float *accin = nullptr;    // [batch][C][L] or null (may alias y)
    float *y = nullptr;              // [batch][C][L], must not alias x
    int64_t L = 0;
    int dil = 1;
    float slope = 1.f, out_scale = 1.f;
    int tiles_per_row = 0, n_tiles = 0, per_xcd = 0;
};

// plain (unpacked) fp32 VALU next to another wave's matrix instructions (resblock_bf.hip: packed fp32 there costs ~100 cycles each)
__device__ __forceinline__ float cb_sub_np(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float cb_add_np(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float cb_mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a, b) -> three words of two bf16 each whose sums are a and b exactly
__device__ __forceinline__ void cb_split3_np(float a, float b, unsigned w[3]) {
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(cb_f32x2{a, b}, cb_bf16x2));
        w[level] = ww;
        if (level < 2) {
            a = cb_sub_np(a, __uint_as_float(ww << 16));
            b = cb_sub_np(b, __uint_as_float(ww & 0xffff0000u));
        }
    }
}

template <int KW, int C>
struct Cb1Geom {
    static constexpr int NCH = C / CB1_CK;                    // input-channel chunks per tile
    static constexpr int RBW = C / 128;                       // 32-row blocks per compute wave
    static constexpr int KS = CB1_CK / 16;                    // 16-deep k steps per chunk and tap
    static constexpr int NGC = KW * KS;                       // (tap, k step) groups per chunk
    static constexpr int NGT = NCH * NGC;                     // ... per tile
    static constexpr int H = (KW - 1) / 2;
    static constexpr int ROWB = 6 * CB1_CK + 16;              // [split 3][channel 64] bf16 + 16 bytes: an odd multiple of 16
    static constexpr int XROWS = CB1_N1 + (KW - 1) * 5;       // dilation <= 5
    static constexpr int X_BYTES = (XROWS + 1) * ROWB;        // + one row that takes the writes of items outside the tile
    static constexpr int RC32 = (XROWS + 31) / 32;
    static constexpr int NIT = (CB1_CK / 8) * RC32 / 4;       // (32-row chunk, channel quad pair) items per stager wave and chunk
    static constexpr int IO_BYTES = C * CB1_N1 * 4;           // the finished tile [channel][column]
    static constexpr int LDS_BYTES = 2 * X_BYTES + IO_BYTES;
    static constexpr int ROWBLOCKS = C / 32;
    static constexpr int CONV_BYTES = NGT * ROWBLOCKS * 1024; // [group][row block][lane][8 bf16]
    static_assert(C == 128 || C == 256, "square layers of 128 or 256 channels");
    static_assert((CB1_CK / 8) * RC32 % 4 == 0, "the items must divide over the four stager waves");
    static_assert(4 * NIT + 24 <= 56, "memory operations in flight per stager wave");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd multiple of 16 bytes");
};

template <int KW, int C>
__global__ void __launch_bounds__(CB1_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
convbf1_kernel(const Cb1Params p) {
    using GM = Cb1Geom<KW, C>;
    constexpr int NCH = GM::NCH, RBW = GM::RBW, KS = GM::KS, NGC = GM::NGC, NGT = GM::NGT, H = GM::H, ROWB = GM::ROWB, NIT = GM::NIT;
    constexpr int N1 = CB1_N1, CK = CB1_CK, PA = 8;           // tap-fragment ring: eight groups (a group is 6 RBW matrix instructions)

    extern __shared__ __attribute__((aligned(16))) unsigned char cb_smem[];
    unsigned char *const xs = cb_smem;                                        // [2][X_BYTES]
    float *const io_lds = reinterpret_cast<float *>(cb_smem + 2 * GM::X_BYTES);   // [C][N1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int d = p.dil, XR = N1 + (KW - 1) * d;
    const int64_t L = p.L;
    const unsigned L4 = (unsigned)(L * 4);
    const int num_bytes = (int)((int64_t)C * L * 4);

    // this block's tiles: XCD x owns a contiguous range of tiles and its blocks walk it side by side (neighbouring tiles share their
    // halo columns in the same L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int tile_end = (xcd + 1) * p.per_xcd < p.n_tiles ? (xcd + 1) * p.per_xcd : p.n_tiles;
    const int tile0 = xcd * p.per_xcd + slot;
    if (tile0 >= tile_end) return;
    const int my_tiles = (tile_end - tile0 + nslot - 1) / nslot;
    const int n_q = my_tiles * NCH;                            // chunks this block walks: chunk q = (tile tile0 + (q / NCH) nslot, channels 64 (q % NCH) ..)

    if (wave >= 4) {
        // ============================================ stagers: HBM <-> LDS ============================================================
        __builtin_amdgcn_s_setprio(1);                        // few instructions, on the block's critical path (the barriers)
        const int sw = wave - 4;
        float xr[NIT][4];
        constexpr int RC = GM::RC32;
        const int lq = lane >> 5;
        // item i of this wave = (channel quad pair qp of the chunk, 32-row chunk rc) with sw * NIT + i = qp * RC + rc; the lower half-wave
        // takes quad 2 qp, the upper one quad 2 qp + 1, a lane's row is rc * 32 + (lane & 31)
        auto x_issue = [&](int q) __attribute__((always_inline)) {
            const int tl = tile0 + (q / NCH) * nslot, ch0 = (q % NCH) * CK;
            const int bb = tl / p.tiles_per_row;
            const int xt0 = (tl - bb * p.tiles_per_row) * N1 - H * d;                       // time of row 0
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (int64_t)bb * C * L), 0, num_bytes, CB1_RSRC_FLAGS);
            const unsigned Lu = (unsigned)L;
            int sw_o = sw;
            asm volatile("" : "+s"(sw_o));                    // the per-item scalars are recomputed, not hoisted and spilled
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                const int qd = 2 * qp + lq, r = rc * 32 + l31;
                const unsigned tg = (unsigned)(xt0 + r);                                    // negative or beyond the row: >= L as unsigned
                const bool ok = r < XR && tg < Lu;
                const unsigned base = (unsigned)(ch0 + 4 * qd) * L4 + tg * 4u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? base + (unsigned)e * L4 : CB1_OOB), 0, 0));
            }
        };
        const float slope = p.slope;
        auto x_write = [&](int q) __attribute__((always_inline)) {
            unsigned char *const xb = xs + (q & 1) * GM::X_BYTES;
            int sw_o = sw;
            asm volatile("" : "+s"(sw_o));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                const int qd = 2 * qp + lq, r = rc * 32 + l31;
                unsigned w[2][3];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const float va = xr[i][2 * e2], vb = xr[i][2 * e2 + 1];
                    cb_split3_np(__builtin_fmaxf(va, cb_mul_np(va, slope)), __builtin_fmaxf(vb, cb_mul_np(vb, slope)), w[e2]);
                }
                unsigned char *o = xb + (r < XR ? r : GM::XROWS) * ROWB + qd * 8;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<cb_u32x2 *>(o + s * 2 * CK) = cb_u32x2{w[0][s], w[1][s]};
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- the finished tile: io tile (+ residual + running sum) * scale -> HBM, 16 bytes per lane, whole rows ----------------------
        constexpr int CHUNKS = N1 / 4, RPW = 64 / CHUNKS, PASSES = C / (4 * RPW);
        constexpr int HP = 4, NHF = PASSES / HP;              // four passes at a time: registers
        const int chunk = lane % CHUNKS, rsub = lane / CHUNKS;
        const float out_scale = p.out_scale;
        const bool l4 = (L & 3) == 0;
        const bool has_res = p.res != nullptr, has_acc = p.accin != nullptr;
        auto out_store = [&](int tl) __attribute__((always_inline)) {
            const int bb = tl / p.tiles_per_row;
            const int64_t t0 = (int64_t)(tl - bb * p.tiles_per_row) * N1;
            const bool ok = t0 + 4 * chunk < L;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (int64_t)bb * C * L), 0, num_bytes, CB1_RSRC_FLAGS);
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)((has_res ? p.res : p.y) + (int64_t)bb * C * L), 0, num_bytes, CB1_RSRC_FLAGS);
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)((has_acc ? p.accin : p.y) + (int64_t)bb * C * L), 0, num_bytes, CB1_RSRC_FLAGS);
            const unsigned o0 = ok ? (unsigned)(sw * RPW + rsub) * L4 + (unsigned)(t0 + 4 * chunk) * 4u : CB1_OOB;
            auto load4 = [&](const __amdgpu_buffer_rsrc_t &rsrc, unsigned o) __attribute__((always_inline)) -> f32x4 {
                if (l4) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0));
                float ae[4];                                  // rows not 16-byte aligned: element by element (past the row's end: zero)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned oe = (ok && t0 + 4 * chunk + e < L) ? o + 4u * (unsigned)e : CB1_OOB;
                    ae[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oe, 0, 0));
                }
                return f32x4{ae[0], ae[1], ae[2], ae[3]};
            };
#pragma unroll
            for (int hf = 0; hf < NHF; ++hf) {
                f32x4 v[HP], rv[HP], av[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    const unsigned o = o0 + (unsigned)((hf * HP + k) * 4 * RPW) * L4;         // (an out-of-range o0 stays out of range)
                    rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    av[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (has_res) rv[k] = load4(rrs, o);
                    if (has_acc) av[k] = load4(ars, o);
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) v[k] = *reinterpret_cast<const f32x4 *>(io_lds + (((hf * HP + k) * 4 + sw) * RPW + rsub) * N1 + 4 * chunk);
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    const unsigned o = o0 + (unsigned)((hf * HP + k) * 4 * RPW) * L4;
                    float re[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    const float r4[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w}, a4[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (has_res) re[e] = cb_add_np(re[e], r4[e]);
                        if (has_acc) re[e] = cb_add_np(re[e], a4[e]);
                        re[e] = cb_mul_np(re[e], out_scale);
                    }
                    if (l4) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cb_u32x4, f32x4{re[0], re[1], re[2], re[3]}), yrs, (int)o, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned oe = (ok && t0 + 4 * chunk + e < L) ? o + 4u * (unsigned)e : CB1_OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, re[e]), yrs, (int)oe, 0, 0);
                        }
                    }
                }
            }
        };
        // Chunk q + 1 is requested behind barrier A(q) -- after chunk q + 1's predecessor in these registers (chunk q) has been
        // written -- and written to LDS buffer (q + 1) & 1 behind barrier A(q + 1)'s predecessor ... in program order per phase:
        //   A(q): buffer q & 1 is complete and the compute waves are done with buffer (q + 1) & 1
        //   [first chunk of a tile: the previous tile's outputs are in the io tile -> store them]
        //   write chunk q + 1 (requested a phase ago) into buffer (q + 1) & 1, request chunk q + 2
        // never more than one register set + the (at most 3 x 4 per half) output operations in flight per wave.
        x_issue(0);
        x_write(0);
        if (1 < n_q) x_issue(1);
        for (int q = 0; q < n_q; ++q) {
            lds_barrier();                                    // (A)
            if (q % NCH == 0 && q > 0) out_store(tile0 + (q / NCH - 1) * nslot);
            if (q + 1 < n_q) x_write(q + 1);
            // the output operations are a few thousand cycles old by now: all but eight of them have returned before the next set is
            // requested -- never more than 8 + one set of memory operations in flight per wave (resblock_bf.hip's rule)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (q + 2 < n_q) x_issue(q + 2);
        }
        lds_barrier();                                        // (E) the last tile's outputs are in the io tile
        out_store(tile0 + (my_tiles - 1) * nslot);
        return;
    }

    // ================================================ compute waves ==========================================================
    // wave w owns output channels [32 RBW w, 32 RBW (w + 1)) x all 64 columns: RBW x 2 accumulator tiles
    float bias[RBW][16];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * (RBW * wave + rb) + (r & 3) + 8 * (r >> 2) + 4 * half;
            bias[rb][r] = p.bias ? p.bias[ch] : 0.f;
        }
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, GM::CONV_BYTES, CB1_RSRC_FLAGS);
    cb_bf16x8 fa[PA][RBW];
    cb_bf16x8 fb[2][2][3];
    f32x16 acc[RBW][2];
    // group g (of the tile, 0 .. NGT - 1; the stream wraps: every tile uses the same taps): this wave's RBW row blocks
    auto load_a = [&](int slot_a, int g) __attribute__((always_inline)) {
        const int soff = (g * GM::ROWBLOCKS + RBW * wave) * 1024;
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
            fa[slot_a][rb] = __builtin_bit_cast(cb_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + rb * 1024, soff, 0));
    };
    auto load_a1 = [&](int slot_a, int g, int rb) __attribute__((always_inline)) {
        const int soff = (g * GM::ROWBLOCKS + RBW * wave) * 1024;
        fa[slot_a][rb] = __builtin_bit_cast(cb_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + rb * 1024, soff, 0));
    };
    // one window fragment of group gc (of the chunk): (column tile cb, split s)
    auto load_b1 = [&](int buf, const unsigned char *src, int gc, int cb, int s) __attribute__((always_inline)) {
        const int tap = gc / KS, ks = gc - tap * KS;
        fb[buf][cb][s] = __builtin_bit_cast(cb_bf16x8, *reinterpret_cast<const cb_u32x4 *>(src + tap * d * ROWB + ks * 32 + cb * 32 * ROWB + s * 2 * CK));
    };
#pragma unroll
    for (int g = 0; g < PA - 1; ++g) load_a(g, g);
    const int x_lane = l31 * ROWB + half * 16;
    float *const io_mine = io_lds + (32 * RBW * wave + 4 * half) * N1 + l31;
    constexpr int NM = 6 * RBW;                               // matrix instructions per group

    // The K loop is unrolled over TWO chunks (the ring slot of a group must be a compile-time register index and 2 NGC groups are a
    // multiple of the ring's eight); C = 256 walks its two chunk pairs in a run-time loop.
    static_assert((2 * NGC) % PA == 0 && NCH % 2 == 0, "two chunks of groups must be whole turns of the tap ring");
    int q = 0;
    for (int t = 0; t < my_tiles; ++t) {
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
#pragma unroll 1
        for (int cp = 0; cp < NCH / 2; ++cp) {
            const int g0 = cp * 2 * NGC;                      // first group of this chunk pair (wave-uniform)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2, ++q) {
                lds_barrier();                                // (A) chunk q's rows are in buffer q & 1
                const unsigned char *const src = xs + (q & 1) * GM::X_BYTES + x_lane;
#pragma unroll
                for (int k = 0; k < 6; ++k) load_b1(0, src, 0, k & 1, 2 - (k >> 1));
#pragma unroll
                for (int gc = 0; gc < NGC; ++gc) {
                    const int gl = c2 * NGC + gc;             // compile-time: the group's place in the pair -> its ring slot
                    // 6 RBW matrix instructions (smallest products first: w x_2, w x_1, w x_0); behind instruction k, pinned: one of the
                    // NEXT group's six window fragments (split 2 first), then the tap fragments of the group PA - 1 ahead (the stream
                    // wraps into the next tile: every tile uses the same taps)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int rb = 0; rb < RBW; ++rb) {
                                const int k = (2 * i + cb) * RBW + rb;
                                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gl % PA][rb], fb[gc & 1][cb][2 - i], acc[rb][cb], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if (k < 6 && gc + 1 < NGC) load_b1((gc + 1) & 1, src, gc + 1, k & 1, 2 - (k >> 1));
                                if (k >= NM - RBW) {
                                    int gn = g0 + gl + PA - 1;
                                    gn = gn >= NGT ? gn - NGT : gn;
                                    load_a1((gl + PA - 1) % PA, gn, k - (NM - RBW));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                }
            }
        }
        // ---- epilogue: bias, into the io tile (residual, running sum, scale and the stores are the stagers') -----------------------------
        // (the stagers took the previous tile's outputs out of the io tile behind this tile's first barrier A)
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    io_mine[(32 * rb + (r & 3) + 8 * (r >> 2)) * N1 + cb * 32] = acc[rb][cb][r] + bias[rb][r];
    }
    lds_barrier();                                            // (E)
}

// ---- host side -------------------------------------------------------------------------------------------------------------

bool convbf1_supported(int c, int k, int dil) { return (c == 128 || c == 256) && (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= 5; }

// Where the bf16-storage decoder takes it (profiles/r06_convbf1_shapes.txt, the (conv, conv) pair as two launches against K3y's two):
// C = 256: 98-103 / 180 / 265-273 us at 3 / 7 / 11 taps against 175-182 / 291 / 394-404; C = 128 at 11 taps: 669-678 against 758-796.
// C = 128 at 3 / 7 taps stays on the fused pair (K3f with one-term taps: 234-237 / 475-483 against 326-338 / 479-489 here).
bool convbf1_preferred(int c, int k) {
#ifdef RVC_ABLATE
    static const int on = knob("RVC_CB1", 1);
    if (!on) return false;
    static const int all = knob("RVC_CB1_ALL", 0);
    if (all) return true;
#endif
    return (c == 128 && k == 11) || c == 256;
}

bool convbf1_fits(int c, int64_t L) { return (int64_t)c * L * 4 < ((int64_t)1 << 31); }

size_t convbf1_weight_bytes(int c, int k) { return (size_t)(c / 16) * k * (c / 32) * 1024; }

// w: [c][c][k] (PyTorch Conv1d layout), ROUNDED to bf16 (round to nearest even: what weight_storage = 1 stores) ->
// [chunk][tap][k step][row block][lane][8 bf16]: lane l of a fragment holds output channel 32 rb + (l & 31), input channels
// 64 chunk + 16 ks + 8 (l >> 5) .. + 7
void convbf1_pack_host(const float *w, int c, int k, std::vector<uint16_t> *out) {
    const int NCH = c / CB1_CK, KS = CB1_CK / 16, RB = c / 32;
    out->assign(convbf1_weight_bytes(c, k) / 2, 0);
    for (int ch = 0; ch < NCH; ++ch)
        for (int tap = 0; tap < k; ++tap)
            for (int ks = 0; ks < KS; ++ks)
                for (int rb = 0; rb < RB; ++rb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = 32 * rb + (lane & 31), ci = CB1_CK * ch + 16 * ks + 8 * (lane >> 5) + e;
                            const size_t group = ((size_t)ch * k + tap) * KS + ks;
                            (*out)[(group * RB + rb) * 512 + lane * 8 + e] = bf16_rne(w[((size_t)co * c + ci) * k + tap]);
                        }
}

static int cb1_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

template <int KW, int C>
static int cb1_launch(Cb1Params p, int batch, hipStream_t stream) {
    if (reserve_whole_cu((const void *)convbf1_kernel<KW, C>, "convbf1")) return 1;
    p.tiles_per_row = (int)ceil_div(p.L, CB1_N1);
    p.n_tiles = p.tiles_per_row * batch;
    p.per_xcd = (int)ceil_div(p.n_tiles, 8);
    const int cus = cb1_cu_count() / 8 * 8;
    const int slots = (int)std::min<int64_t>(cus / 8, p.per_xcd);           // blocks per XCD
    hipLaunchKernelGGL((convbf1_kernel<KW, C>), dim3((unsigned)(slots * 8)), dim3(CB1_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

// x, y: [batch][c][L] (y must NOT alias x: blocks read their neighbours' columns; res / accin may alias y); u: convbf1_pack_host's slab
int launch_convbf1(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch, int c,
                   int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream) {
    if (!convbf1_supported(c, k, dil)) return fail("convbf1: unsupported shape (%d channels, %d taps, dilation %d)", c, k, dil);
    if (x == y) return fail("convbf1: in-place operation is not supported");
    if (!(slope >= 0.f && slope <= 1.f)) return fail("convbf1: leaky slope %g outside [0, 1]", (double)slope);
    if (!convbf1_fits(c, L)) return fail("convbf1: a %d x %lld slab exceeds the 2 GiB buffer addressing", c, (long long)L);
    if (L <= 0 || batch <= 0) return 0;
    if ((int64_t)ceil_div(L, CB1_N1) * batch >= ((int64_t)1 << 28)) return fail("convbf1: too many tiles");
    Cb1Params p;
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.accin = accin; p.y = y; p.L = L; p.dil = dil; p.slope = slope; p.out_scale = out_scale;
#define RVC_CB1_CASE(KW, CC) if (k == KW && c == CC) return cb1_launch<KW, CC>(p, batch, stream)
    RVC_CB1_CASE(3, 128); RVC_CB1_CASE(7, 128); RVC_CB1_CASE(11, 128);
    RVC_CB1_CASE(3, 256); RVC_CB1_CASE(7, 256); RVC_CB1_CASE(11, 256);
#undef RVC_CB1_CASE
    return fail("convbf1: unsupported shape c=%d k=%d", c, k);
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_conv1d_bf16w_weight_bytes(int c, int k, size_t *bytes) {
    if (!bytes) return fail("rvc_conv1d_bf16w_weight_bytes: null pointer");
    if (!convbf1_supported(c, k, 1)) return fail("rvc_conv1d_bf16w_weight_bytes: c must be 128 or 256, k 3, 7 or 11");
    *bytes = convbf1_weight_bytes(c, k);
    return 0;
}

extern "C" int rvc_conv1d_bf16w_pack_weight(const float *w_host, int c, int k, void *u_dev, void *stream) {
    if (!w_host || !u_dev) return fail("rvc_conv1d_bf16w_pack_weight: null pointer");
    size_t bytes = 0;
    if (rvc_conv1d_bf16w_weight_bytes(c, k, &bytes)) return 1;
    std::vector<uint16_t> u;
    convbf1_pack_host(w_host, c, k, &u);
    hipError_t e = hipMemcpyAsync(u_dev, u.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("rvc_conv1d_bf16w_pack_weight: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_conv1d_bf16w_forward(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev,
                                        const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                                        float slope_in, float out_scale, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_conv1d_bf16w_forward: null pointer");
    return launch_convbf1(x_dev, u_dev, bias_dev, res_dev, acc_dev, y_dev, batch, c, length, k, dilation, slope_in, out_scale,
                          (hipStream_t)stream);
}
