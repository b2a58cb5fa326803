// K3f -- one ResBlock (dilated conv -> conv) pair of the narrow vocoder stages in ONE launch, on the bf16 matrix cores:
//     y = out_scale * ( conv2( leaky( conv1_d( leaky(x) ) + b1 ) ) + b2 + x [+ running sum] )        (residuals.py:75-86, one dilation)
// for C = 32 and C = 64 channels, K = 3 / 7 / 11 taps, every fp32 operand split exactly into three bf16 (8 + 8 + 8 significand
// bits) and the six products of order <= 2^-16 formed by v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the arithmetic of
// gemmbf.hip / winobf.hip; dropped terms < 2^-23 of a product).
//
// Why this form here.  At 32 / 64 channels the contraction per tap is one or two matrix instructions deep, so the Winograd
// kernels' per-point transforms (VALU + LDS) cost as much as the products they save, and the two convs of a pair as separate
// launches move the activation five times (980 MB per pair at C = 32 where a fused pair moves 392 MB).  The DIRECT form needs
// no transform at all once the operands sit in LDS as bf16 triples in [time][channel] order: the B fragment of ANY tap and
// dilation is one 16-byte LDS read at (column + tap * d) * row stride.  The intermediate never leaves the CU.
//
// Block = 8 waves on one CU that it owns (every bf16 matrix kernel of the library requests the whole LDS, common.h), persistent:
// it walks tiles of BN output columns.  Per tile:
//   * waves 4..7 (STAGERS) fetch the NEXT tile's raw rows from HBM into registers while the compute waves run conv1, and -- once
//     conv1 has released the x tile -- apply the leaky ReLU, split every value into three bf16 and write the x tile
//     [time][split][channel] under the compute waves' conv2, next to the raw values of the block's own columns (the residual) in
//     the io buffer [channel][column]; behind the next barrier they take the finished outputs out of that same buffer, add the
//     running sum, scale, and store whole rows 16 bytes per lane.  Every HBM access of the block is theirs; the compute waves'
//     memory queue holds nothing but tap fragments (L2), so no matrix instruction ever waits behind an HBM round trip or a
//     store's acknowledgement (a wave's memory operations retire in order).
//   * waves 0..3 (COMPUTE) each own a 32-channel row block x 64 columns (two accumulator tiles): per (tap, 16-channel k step) the
//     three tap fragments come L2 -> registers (ring of four groups), the six window fragments from LDS (double-buffered), twelve
//     matrix instructions on two independent accumulators.  conv1's result gets bias, leaky ReLU, the conv's zero padding outside
//     [0, L), the three-way split, and lands in the t tile in the same [time][split][channel] layout; conv2 reads it back.
//   * three block barriers per tile (x tile ready / outputs handed over / t tile ready).
// LDS rows are 6 C + 16 bytes: an odd multiple of 16 bytes mod 256, so the 16 lanes of a ds_read_b128 group hit 16 different
// bank quads whatever the tap offset.
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "conv.h"

namespace rvc {

typedef __bf16 rb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rb_u32x4 __attribute__((ext_vector_type(4)));

constexpr int RBF_NTH = 512;
constexpr int RBF_RSRC_FLAGS = 0x00020000;
constexpr unsigned RBF_OOB = 0x80000000u;   // a buffer offset beyond every tensor this kernel takes: loads return 0, stores are dropped

struct RbfParams {
    const float *x = nullptr;        // [batch][C][L]
    const void *u = nullptr;         // resblock_bf_pack_host's slab
    const float *b1 = nullptr, *b2 = nullptr;
    const float *accin = nullptr;    // [batch][C][L] or null
    float *y = nullptr;              // [batch][C][L], must not alias x
    int64_t L = 0;
    int dil = 1;
    float slope = 1.f, out_scale = 1.f;
    int tiles_per_row = 0, n_tiles = 0, per_xcd = 0;
};

// Plain (unpacked) fp32 VALU for code that runs NEXT TO another wave's matrix instructions on the same SIMD: v_pk_add_f32 /
// v_pk_mul_f32 there waited ~100 cycles each (the stagers' 60 packed operations per tile took 7 800 cycles of a 20 000-cycle tile:
// profiles/r05_rbf_stamps.txt); inline asm, so that neither the vector types nor the SLP vectoriser can pack them again.
__device__ __forceinline__ float rb_sub_np(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float rb_add_np(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float rb_mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a, b) -> three words of two bf16 each whose sums are a and b exactly
__device__ __forceinline__ void rb_split3_np(float a, float b, unsigned w[3]) {
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(rb_f32x2{a, b}, rb_bf16x2));
        w[level] = ww;
        if (level < 2) {
            a = rb_sub_np(a, __uint_as_float(ww << 16));
            b = rb_sub_np(b, __uint_as_float(ww & 0xffff0000u));
        }
    }
}

// NTS = terms of the TAP split: 3 for fp32 taps (six products per multiply-add), 1 for bf16-VALUED taps (BASELINE cfg 4's weight
// storage: w = w_0 exactly, so the three products w_0 x_0 + w_0 x_1 + w_0 x_2 are the whole result -- half the matrix instructions
// and a third of the tap-fragment bytes per group; activations and the intermediate keep their three-way split)
template <int KW, int C, int NTS = 3>
struct RbfGeom {
    static constexpr int KS = C / 16, RB = C / 32, CG = 4 / RB;
    static constexpr int N1 = CG * 64;                       // conv1 columns per block
    static constexpr int H2 = (KW - 1) / 2;
    static constexpr int BN = (N1 - (KW - 1)) / 4 * 4;       // output columns per block
    static constexpr int ROWB = 6 * C + 16;                  // [split 3][channel C] bf16 + 16 bytes
    static constexpr int XROWS = N1 + (KW - 1) * 5;          // dilation <= 5
    static constexpr int TROWS = N1 + (KW - 1);
    static constexpr int X_BYTES = (XROWS + 1) * ROWB, T_BYTES = TROWS * ROWB;   // + one row that takes the writes of items outside the tile
    static constexpr int RC32 = (XROWS + 31) / 32;           // staged rows in 32-row chunks: a half-wave = one chunk of one channel quad
    static constexpr int NIT = (C / 8) * RC32 / 4;           // (32-row chunk, channel quad pair) items per stager wave
    static constexpr int NG = KW * KS;                       // (tap, k step) groups per conv
    static constexpr int GROUP_BYTES = NTS * 1024;           // the splits of one 32 x 16 tap fragment
    static_assert(NTS == 1 || NTS == 3, "tap split: one term (bf16-valued taps) or three (fp32 taps)");
    static constexpr int CONV_BYTES = NG * RB * GROUP_BYTES;
    static constexpr int R_BYTES = C * N1 * 4 + 16;          // the io buffer (+ 4 floats that take the writes of rows outside the block's own columns): residual (raw rows of the block's own columns) in, outputs out
    static constexpr int LDS_BYTES = X_BYTES + T_BYTES + R_BYTES;
    static_assert((C / 8) * RC32 % 4 == 0, "the items must divide over the four stager waves");
    static_assert(4 * NIT + 8 <= 56, "memory operations in flight per stager wave (kept below the 6-bit counter's range: see the stager's prologue)");
    static_assert(LDS_BYTES <= 163840, "LDS budget");
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd multiple of 16 bytes");
};

// v -> three bf16 whose sum is v exactly (round to nearest even each time; the remainders are exact in fp32)
__device__ __forceinline__ void rb_split3(rb_f32x2 v, unsigned w[3]) {
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rb_bf16x2));
        w[level] = ww;
        if (level < 2) v = v - rb_f32x2{__uint_as_float(ww << 16), __uint_as_float(ww & 0xffff0000u)};
    }
}

// DBG (ablation build only): 128 = wave 0 and stager wave 4 write s_memtime stamps to the buffer passed as `accin` (which is then NOT
// added): [block][compute | stager][64], eight per tile -- tools/stamp_resblock_bf.py turns them into a per-phase breakdown
template <int KW, int C, int DBG = 0, int NTS = 3>
__global__ void __launch_bounds__(RBF_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
resblock_bf_kernel(const RbfParams p) {
    using GM = RbfGeom<KW, C, NTS>;
    constexpr int KS = GM::KS, RB = GM::RB, N1 = GM::N1, H2 = GM::H2, BN = GM::BN, ROWB = GM::ROWB, NIT = GM::NIT;
    // tap-fragment ring: four groups with three-term taps (a ring of six: no change, measured); a one-term group is half as long
    // (six matrix instructions), so its ring is twice as deep to keep the same distance in cycles between request and use
    constexpr int NG = GM::NG, PA = NTS == 1 ? 8 : 4;

    extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
    unsigned char *const xs = rb_smem;
    unsigned char *const ts = rb_smem + GM::X_BYTES;
    float *const io_lds = reinterpret_cast<float *>(rb_smem + GM::X_BYTES + GM::T_BYTES);   // [C][N1]: residual in, outputs out

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int d = p.dil, h1 = H2 * d, XR = N1 + 2 * h1;
    const int64_t L = p.L;
    const unsigned L4 = (unsigned)(L * 4);
    const int num_bytes = (int)((int64_t)C * L * 4);

    // this block's tiles: XCD x owns the contiguous range [x per_xcd, (x + 1) per_xcd) and its blocks walk it side by side, so the
    // halo columns two neighbouring tiles share are fetched into the same L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int tile_end = (xcd + 1) * p.per_xcd < p.n_tiles ? (xcd + 1) * p.per_xcd : p.n_tiles;
    int tile = xcd * p.per_xcd + slot;
    if (tile >= tile_end) return;
    unsigned long long *const stamps = (DBG & 128) ? reinterpret_cast<unsigned long long *>(const_cast<float *>(p.accin)) + ((size_t)blockIdx.x * 2 + (wave >= 4 ? 1 : 0)) * 64 : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (DBG & 128) {
            if ((wave == 0 || wave == 4) && lane == 0 && n_stamp < 64) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };

    if (wave >= 4) {
        // ============================================ stagers: HBM <-> LDS ============================================================
        // This wave shares its SIMD with a compute wave and, being the younger one, loses every issue arbitration to it: its few
        // hundred instructions per tile sit on the block's critical path (the barriers), so it gets the issue slots first.
        __builtin_amdgcn_s_setprio(1);
        const int ht = tid - 256, sw = wave - 4;
        // C = 32: two tiles in flight (tile i waits in set i & 1).  C >= 64: one -- 40-48 registers per set, and the shortest conv1
        // there (three taps at 64 channels: 6 500 cycles with its epilogue) still covers an HBM round trip
        constexpr int NSET = C == 32 ? 2 : 1;
        float xr[NSET][NIT][4];
        // item i of this wave = (channel quad pair qp, 32-row chunk rc) with sw * NIT + i = qp * RC + rc: qp and rc are wave-uniform
        // (scalar registers); the lower half-wave takes quad 2 qp, the upper one quad 2 qp + 1, a lane's row is rc * 32 + (lane & 31)
        constexpr int RC = GM::RC32;
        const int lq = lane >> 5;
        auto x_issue = [&](auto SET, int tl) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
            const int bb = tl / p.tiles_per_row;
            const int xt0 = (tl - bb * p.tiles_per_row) * BN - H2 - h1;                     // time of row 0 (a row holds < 2^29 samples)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + (int64_t)bb * C * L), 0, num_bytes, RBF_RSRC_FLAGS);
            const unsigned Lu = (unsigned)L;
            int sw_o = sw;                                    // opaque: the per-item scalars are recomputed here (a dozen scalar operations
            asm volatile("" : "+s"(sw_o));                    // per item) instead of being hoisted out of the tile loop and spilled (333 of them)
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                const int q = 2 * qp + lq, r = rc * 32 + l31;
                const unsigned tg = (unsigned)(xt0 + r);                                    // negative or beyond the row: >= L as unsigned
                const bool ok = r < XR && tg < Lu;
                const unsigned base = (unsigned)(4 * q) * L4 + tg * 4u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xr[st][i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? base + (unsigned)e * L4 : RBF_OOB), 0, 0));
            }
        };
        const float slope = p.slope;
        // no branch per item (an item outside the tile writes to the dump row): the items' dependency chains interleave
        auto x_write = [&](auto SET) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value;
            int sw_o = sw;
            asm volatile("" : "+s"(sw_o));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int wi = sw_o * NIT + i, qp = wi / RC, rc = wi - qp * RC;
                const int q = 2 * qp + lq, r = rc * 32 + l31;
                unsigned w[2][3];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const float va = xr[st][i][2 * e2], vb = xr[st][i][2 * e2 + 1];
                    rb_split3_np(__builtin_fmaxf(va, rb_mul_np(va, slope)), __builtin_fmaxf(vb, rb_mul_np(vb, slope)), w[e2]);
                }
                unsigned char *o = xs + (r < XR ? r : GM::XROWS) * ROWB + q * 8;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<rb_u32x2 *>(o + s * 2 * C) = rb_u32x2{w[0][s], w[1][s]};
                // the raw values of the block's own columns: the residual the compute waves add (lanes = consecutive columns of one row)
                const int j = r - h1 - H2;
                const bool own = r < XR && j >= 0 && j < N1;
                float *const ro = io_lds + (own ? 4 * q * N1 + j : C * N1);
#pragma unroll
                for (int e = 0; e < 4; ++e) ro[own ? e * N1 : e] = xr[st][i][e];
                if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two items (four chains) at a time: more of them spill
            }
        };
        // ---- the finished tile: io buffer -> (+ running sum) * scale -> HBM, 16 bytes per lane, whole rows ------------------------------
        constexpr int CHUNKS = N1 / 4, RPW = 64 / CHUNKS, PASSES = C / (4 * RPW);      // float4 per row, rows per wave and pass
        const int chunk = lane % CHUNKS, rsub = lane / CHUNKS;
        const float out_scale = p.out_scale;
        const bool l4 = (L & 3) == 0;
        auto out_store = [&](int tl, bool with_acc) __attribute__((always_inline)) {
            const int bb = tl / p.tiles_per_row;
            const int64_t t0 = (int64_t)(tl - bb * p.tiles_per_row) * BN;
            const bool ok = 4 * chunk < BN && t0 + 4 * chunk < L;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.y + (int64_t)bb * C * L), 0, num_bytes, RBF_RSRC_FLAGS);
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)((with_acc ? p.accin : p.y) + (int64_t)bb * C * L), 0, num_bytes, RBF_RSRC_FLAGS);
            const unsigned o0 = ok ? (unsigned)(sw * RPW + rsub) * L4 + (unsigned)(t0 + 4 * chunk) * 4u : RBF_OOB;
            constexpr int HP = PASSES / 2;                    // two halves: registers (the x sets stay live next to this)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 v[HP], av[HP];
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    const int ps = hf * HP + k;
                    const unsigned o = o0 + (unsigned)(ps * 4 * RPW) * L4;         // (an out-of-range o0 stays out of range)
                    av[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (with_acc) {   // the running sum of the tile's own columns (one launch in 4.5): requested first, used last
                        if (l4) {
                            av[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)o, 0, 0));
                        } else {      // rows not 16-byte aligned: element by element (elements past the row's end: out of range, zero)
                            float ae[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned oe = (ok && t0 + 4 * chunk + e < L) ? o + 4u * (unsigned)e : RBF_OOB;
                                ae[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, (int)oe, 0, 0));
                            }
                            av[k] = f32x4{ae[0], ae[1], ae[2], ae[3]};
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < HP; ++k) v[k] = *reinterpret_cast<const f32x4 *>(io_lds + (((hf * HP + k) * 4 + sw) * RPW + rsub) * N1 + 4 * chunk);
#pragma unroll
                for (int k = 0; k < HP; ++k) {
                    const unsigned o = o0 + (unsigned)((hf * HP + k) * 4 * RPW) * L4;
                    float re[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    const float ae[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (with_acc) re[e] = rb_add_np(re[e], ae[e]);
                        re[e] = rb_mul_np(re[e], out_scale);
                    }
                    if (l4) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rb_u32x4, f32x4{re[0], re[1], re[2], re[3]}), yrs, (int)o, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned oe = (ok && t0 + 4 * chunk + e < L) ? o + 4u * (unsigned)e : RBF_OOB;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, re[e]), yrs, (int)oe, 0, 0);
                        }
                    }
                }
            }
        };
        const bool has_acc = p.accin != nullptr && !(DBG & 128);
        // rows N1 .. TROWS-1 of the t tile are read by masked output columns only and never written: give them a value once
        for (int o = ht * 16; o < (KW - 1) * ROWB; o += 256 * 16) *reinterpret_cast<rb_u32x4 *>(ts + N1 * ROWB + o) = rb_u32x4{0u, 0u, 0u, 0u};
        // Tile i + 2 is requested behind barrier D of tile i and written behind barrier B of tile i + 1: a whole tile between request
        // and use.  Never more than one set + 8 stores in flight per wave, and tile 0 is written before tile 1 is requested: with the
        // prologue order (request 0, request 1, write 0) -- 96 loads in flight at C = 64 -- the SECOND tile of 4-18 blocks per launch
        // came out wrong in its last columns (errors ~1e-2, non-deterministic; profiles/r05_rbf_notes.txt).  The cause was not
        // isolated: a bare wave with 120 loads in flight reads correct data (tools/micro/vmcnt_burst.hip), so it is not the 6-bit
        // counter by itself.  The order below passed every repetition since.
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, NSET - 1> SL{};   // the "other" set (the same one when there is only one)
        x_issue(S0, tile);
        x_write(S0);
        if (NSET == 2 && tile + nslot < tile_end) x_issue(SL, tile + nslot);
        int prev = -1;                                        // the tile whose outputs the compute waves hand over behind the next barrier A
        auto phase = [&](int tl, auto SET_CUR, auto SET_NEXT) __attribute__((always_inline)) {
            stamp();
            lds_barrier();                                    // (A) x tile + residual complete; the compute waves swap residual <-> outputs
            stamp();
            lds_barrier();                                    // (D) the previous tile's outputs are in the io buffer
            if (prev >= 0) out_store(prev, has_acc);
            // everything older than those (at most eight) stores has returned -- it is a whole phase old -- before the next set is
            // requested: never more than 8 + one set of memory operations in flight (see below)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (tl + NSET * nslot < tile_end) x_issue(SET_CUR, tl + NSET * nslot);   // this set left the registers before barrier A
            stamp();
            lds_barrier();                                    // (B) conv1 has read the x tile
            stamp();
            if constexpr (DBG & 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stamps: separate the wait for the loads from the work
            stamp();
            if (tl + nslot < tile_end) x_write(SET_NEXT);
            stamp();
            n_stamp += 2;
            prev = tl;
        };
        for (; tile < tile_end; tile += 2 * nslot) {
            phase(tile, S0, SL);
            if (tile + nslot < tile_end) phase(tile + nslot, SL, S0);
        }
        // the last tile's outputs
        lds_barrier();                                        // (A')
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                        // (D')
        out_store(prev, has_acc);
        return;
    }

    // ================================================ compute waves ==========================================================
    const int rb = wave % RB, cg = wave / RB;
    const int col0 = cg * 64;
    float bias1[16], bias2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * half;
        bias1[r] = p.b1 ? p.b1[ch] : 0.f;
        bias2[r] = p.b2 ? p.b2[ch] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, 2 * GM::CONV_BYTES, RBF_RSRC_FLAGS);
    const float slope = p.slope;

    rb_bf16x8 fa[PA][NTS];
    rb_bf16x8 fb[2][2][3];
    f32x16 acc[2];
    // group g = tap * KS + ks of conv `cv`: the three splits of this wave's row block
    auto load_a = [&](int slot_a, int cv, int g) __attribute__((always_inline)) {
        const int soff = cv * GM::CONV_BYTES + (g * RB + rb) * GM::GROUP_BYTES;
#pragma unroll
        for (int s = 0; s < NTS; ++s)
            fa[slot_a][s] = __builtin_bit_cast(rb_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + s * 1024, soff, 0));
    };
    // one window fragment of group g: (column tile cb, split s)
    auto load_b1 = [&](int buf, const unsigned char *src, int tapstep, int g, int cb, int s) __attribute__((always_inline)) {
        const int tap = g / KS, ks = g - tap * KS;
        fb[buf][cb][s] = __builtin_bit_cast(rb_bf16x8, *reinterpret_cast<const rb_u32x4 *>(src + tap * tapstep + ks * 32 + cb * 32 * ROWB + s * 2 * C));
    };
    auto load_a1 = [&](int slot_a, int cv, int g, int s) __attribute__((always_inline)) {
        const int soff = cv * GM::CONV_BYTES + (g * RB + rb) * GM::GROUP_BYTES;
        fa[slot_a][s] = __builtin_bit_cast(rb_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + s * 1024, soff, 0));
    };
    auto a_prologue = [&](int cv) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < PA - 1; ++g)
            if (g < NG) load_a(g, cv, g);
    };
    // (tap split, window split): smallest products first; one-term taps: w_0 x_2, w_0 x_1, w_0 x_0
    constexpr int NPR = NTS == 1 ? 3 : 6;
    constexpr int ia6[6] = {0, NTS == 1 ? 0 : 1, 0, 2, 1, 0}, ib6[6] = {2, 1, NTS == 1 ? 0 : 1, 0, 0, 0};
    auto conv_loop = [&](int cv, const unsigned char *src, int tapstep) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) load_b1(0, src, tapstep, 0, k & 1, 2 - (k >> 1));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // twelve (one-term taps: six) matrix instructions; behind instruction k, pinned: one of the NEXT group's six window
            // fragments (split 2 first: the order the products consume them), then the tap fragments of the group PA - 1 ahead
#pragma unroll
            for (int i = 0; i < NPR; ++i)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int k = 2 * i + cb;
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g % PA][ia6[i]], fb[g & 1][cb][ib6[i]], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k < 6 && g + 1 < NG) load_b1((g + 1) & 1, src, tapstep, g + 1, k & 1, 2 - (k >> 1));
                    if (NTS == 3 && k >= 6 && k < 9 && g + PA - 1 < NG) load_a1((g + PA - 1) % PA, cv, g + PA - 1, k - 6);
                    if (NTS == 1 && k == 3 && g + PA - 1 < NG) load_a1((g + PA - 1) % PA, cv, g + PA - 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    const unsigned char *const xsrc = xs + (col0 + l31) * ROWB + half * 16;
    const unsigned char *const tsrc = ts + (col0 + l31) * ROWB + half * 16;
    const int xstep = d * ROWB;
    // The compute waves issue no memory operation but their tap-fragment loads (L2): the residual comes from, and the outputs go
    // to, the io buffer [channel][column] in LDS, which the stagers fill and drain with 16-byte accesses of whole rows.  (Stored by
    // these waves -- 32 four-byte stores per lane -- a tile's outputs cost their issue time plus, a wave's memory operations retiring
    // in order, the stores' write acknowledgement in front of the next tap-fragment wait: ~3 500 cycles of a 19 500-cycle tile.)
    // io[cb][r]: from barrier A to conv2's epilogue the residual of this tile; from there to the next barrier A its outputs.
    float io[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) io[cb][r] = 0.f;
    float *const io_mine = io_lds + (32 * rb + 4 * half) * N1 + col0 + l31;
    // every element of the io buffer is read (residual) and then written (outputs of the previous tile) by the SAME lane: no barrier between
    auto swap_io = [&]() __attribute__((always_inline)) {
        float rsd[2][16];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) rsd[cb][r] = io_mine[((r & 3) + 8 * (r >> 2)) * N1 + cb * 32];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) io_mine[((r & 3) + 8 * (r >> 2)) * N1 + cb * 32] = io[cb][r];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) io[cb][r] = rsd[cb][r];
    };

    a_prologue(0);
    for (; tile < tile_end; tile += nslot) {
        const int bb = tile / p.tiles_per_row;
        const int64_t t0 = (int64_t)(tile - bb * p.tiles_per_row) * BN;
        stamp();
        lds_barrier();                                        // (A) x tile + residual complete; every wave is done with the t tile
        stamp();
        swap_io();
        stamp();
        lds_barrier();                                        // (D) the previous tile's outputs are in the io buffer
        stamp();
        conv_loop(0, xsrc, xstep);
        stamp();
        a_prologue(1);
        // ---- conv1's epilogue: bias, leaky ReLU, the zero padding conv2 sees outside [0, L), split, into the t tile ----------
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int j = col0 + cb * 32 + l31;
            const int64_t tg = t0 - H2 + j;
            const unsigned keep = (tg >= 0 && tg < L) ? 0xffffffffu : 0u;
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                unsigned w[2][3];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int r = 4 * jq + 2 * e2;
                    const rb_f32x2 v = rb_f32x2{acc[cb][r] + bias1[r], acc[cb][r + 1] + bias1[r + 1]};
                    const rb_f32x2 sv = v * slope;
                    const rb_f32x2 a = rb_f32x2{__uint_as_float(__float_as_uint(__builtin_fmaxf(v.x, sv.x)) & keep),
                                                __uint_as_float(__float_as_uint(__builtin_fmaxf(v.y, sv.y)) & keep)};
                    rb_split3(a, w[e2]);
                }
                unsigned char *o = ts + j * ROWB + (32 * rb + 8 * jq + 4 * half) * 2;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<rb_u32x2 *>(o + s * 2 * C) = rb_u32x2{w[0][s], w[1][s]};
            }
        }
        stamp();
        lds_barrier();                                        // (B) the t tile is complete; the x tile is free
        stamp();
        conv_loop(1, tsrc, ROWB);
        stamp();
        a_prologue(0);                                        // the next tile's first tap fragments (the last tile re-reads them: no branch)
        // ---- conv2's epilogue: bias + residual (running sum and scale are the stagers') ------------------------------------------------
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) io[cb][r] = acc[cb][r] + bias2[r] + io[cb][r];
    }
    lds_barrier();                                            // (A') nobody reads the io buffer's residual any more
    swap_io();                                                // (the values read back are not used)
    lds_barrier();                                            // (D') the last tile's outputs are in the io buffer
}

// ---- host side -------------------------------------------------------------------------------------------------------------

// Runtime switch of the PRODUCT build (rvc_resblock_bf16x3_set_enabled): handles created while it is off keep their narrow stages on
// the unfused kernels (launch_resblock_layer / winobf / wino).  The ablation build's RVC_RBF=0 sets the initial value.
static std::atomic<int> g_rbf_on{-1};
bool resblock_bf_enabled() {
    int v = g_rbf_on.load(std::memory_order_relaxed);
    if (v < 0) {
        v = knob("RVC_RBF", 1) != 0;
        g_rbf_on.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
void resblock_bf_set_enabled(bool on) { g_rbf_on.store(on ? 1 : 0, std::memory_order_relaxed); }

bool resblock_bf_supported(int c, int k, int dil) {
    return ((c == 32 || c == 64) && (k == 3 || k == 7 || k == 11) || (c == 128 && (k == 3 || k == 7))) && dil >= 1 && dil <= 5;
}

// Where the decoder takes this kernel (profiles/r05_rbf_shapes.txt, us per pair at the cfg-2 lengths, against the two launches it
// replaces): C = 32: 155-164 / 238-243 / 332-338 at 3 / 7 / 11 taps against 250-272 / 322-341 / 406-434; C = 64: 222-224 / 418-430
// against 300-317 / 453-482 at 3 / 7 taps.  C = 64 with 11 taps stays on winobf.hip's Winograd form (637-641 against 554-566: the
// direct form executes 2.1 x its matrix work there and only 116 of a block's 128 columns are outputs).
// One-term taps (tap_splits = 1, bf16-valued weights): three products per multiply-add instead of six -- every supported shape
// (profiles/r06_rbf1_shapes.txt).
bool resblock_bf_preferred(int c, int k, int tap_splits) {
#ifdef RVC_ABLATE
    static const int all = knob("RVC_RBF_ALL", 0);
    if (all) return true;
    static const int one = knob("RVC_RBF_ONE", 1);       // 0: bf16-valued taps as three-term fragments (round 5's cfg 4)
    if (tap_splits == 1 && !one) tap_splits = 3;
#endif
    if (tap_splits == 1) return true;
    return c == 32 || (c == 64 && k != 11) || (c == 128 && k == 3);
}

bool resblock_bf_fits(int c, int64_t L) { return (int64_t)c * L * 4 < ((int64_t)1 << 31); }

size_t resblock_bf_weight_bytes(int c, int k, int tap_splits) { return (size_t)2 * k * (c / 16) * (c / 32) * tap_splits * 1024; }

// w1, w2: [c][c][k] (PyTorch Conv1d layout) -> [conv][tap][k step][row block][split][lane][8 bf16]: lane l of a fragment holds
// output channel 32 rb + (l & 31), input channels 16 ks + 8 (l >> 5) .. + 7.  tap_splits = 1: the taps ROUNDED to bf16 (round to
// nearest even: what `weight_storage = 1` stores), one fragment per group.
void resblock_bf_pack_host(const float *w1, const float *w2, int c, int k, std::vector<uint16_t> *out, int tap_splits) {
    const int KS = c / 16, RB = c / 32;
    out->assign(resblock_bf_weight_bytes(c, k, tap_splits) / 2, 0);
    for (int cv = 0; cv < 2; ++cv) {
        const float *w = cv ? w2 : w1;
        for (int tap = 0; tap < k; ++tap)
            for (int ks = 0; ks < KS; ++ks)
                for (int rb = 0; rb < RB; ++rb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = 32 * rb + (lane & 31), ci = 16 * ks + 8 * (lane >> 5) + e;
                            float r = w[((size_t)co * c + ci) * k + tap];
                            const size_t group = (((size_t)cv * k + tap) * KS + ks) * RB + rb;
                            for (int s = 0; s < tap_splits; ++s) {
                                const uint16_t h = bf16_rne(r);
                                const uint32_t bits = (uint32_t)h << 16;
                                float f;
                                memcpy(&f, &bits, 4);
                                r -= f;                      // exact in fp32
                                (*out)[(group * tap_splits + s) * 512 + lane * 8 + e] = h;
                            }
                        }
    }
}

static int rbf_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

template <int KW, int C, int DBG = 0, int NTS = 3>
static int rbf_launch(RbfParams p, int batch, hipStream_t stream) {
    using GM = RbfGeom<KW, C, NTS>;
    if (reserve_whole_cu((const void *)resblock_bf_kernel<KW, C, DBG, NTS>, "resblock_bf")) return 1;
    p.tiles_per_row = (int)ceil_div(p.L, GM::BN);
    p.n_tiles = p.tiles_per_row * batch;
    p.per_xcd = (int)ceil_div(p.n_tiles, 8);
    const int cus = rbf_cu_count() / 8 * 8;
    const int slots = (int)std::min<int64_t>(cus / 8, p.per_xcd);           // blocks per XCD
    hipLaunchKernelGGL((resblock_bf_kernel<KW, C, DBG, NTS>), dim3((unsigned)(slots * 8)), dim3(RBF_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

// x, y: [batch][c][L] (must NOT alias: blocks read their neighbours' columns); u: resblock_bf_pack_host's slab on the device, packed
// with the same tap_splits
int launch_resblock_bf(const float *x, const void *u, const float *b1, const float *b2, const float *accin, float *y, int batch, int c,
                       int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream, int tap_splits) {
    if (!resblock_bf_supported(c, k, dil)) return fail("resblock_bf: unsupported shape (%d channels, %d taps, dilation %d)", c, k, dil);
    if (tap_splits != 1 && tap_splits != 3) return fail("resblock_bf: tap split of %d terms (1 or 3)", tap_splits);
    if (x == y) return fail("resblock_bf: in-place operation is not supported");
    if (!(slope >= 0.f && slope <= 1.f)) return fail("resblock_bf: leaky slope %g outside [0, 1]", (double)slope);
    if (!resblock_bf_fits(c, L)) return fail("resblock_bf: a %d x %lld slab exceeds the 2 GiB buffer addressing", c, (long long)L);
    if (L <= 0 || batch <= 0) return 0;
    if ((int64_t)ceil_div(L, 64) * batch >= ((int64_t)1 << 30)) return fail("resblock_bf: too many tiles");
    RbfParams p;
    p.x = x; p.u = u; p.b1 = b1; p.b2 = b2; p.accin = accin; p.y = y; p.L = L; p.dil = dil; p.slope = slope; p.out_scale = out_scale;
#ifdef RVC_ABLATE
    static const int dbg = knob("RVC_RBF_DBG", 0);
#define RVC_RBF_DBG_CASE(KW, CC) if (dbg == 128 && k == KW && c == CC && tap_splits == 3) return rbf_launch<KW, CC, 128>(p, batch, stream)
    RVC_RBF_DBG_CASE(3, 32); RVC_RBF_DBG_CASE(7, 32); RVC_RBF_DBG_CASE(11, 32); RVC_RBF_DBG_CASE(7, 64);
#undef RVC_RBF_DBG_CASE
#define RVC_RBF_DBG_CASE(KW, CC) if (dbg == 128 && k == KW && c == CC && tap_splits == 1) return rbf_launch<KW, CC, 128, 1>(p, batch, stream)
    RVC_RBF_DBG_CASE(11, 32); RVC_RBF_DBG_CASE(7, 64); RVC_RBF_DBG_CASE(11, 64); RVC_RBF_DBG_CASE(7, 128);
#undef RVC_RBF_DBG_CASE
#endif
#define RVC_RBF_CASE(KW, CC)                                                                     \
    if (k == KW && c == CC) return tap_splits == 1 ? rbf_launch<KW, CC, 0, 1>(p, batch, stream) : rbf_launch<KW, CC, 0, 3>(p, batch, stream)
    RVC_RBF_CASE(3, 32); RVC_RBF_CASE(7, 32); RVC_RBF_CASE(11, 32);
    RVC_RBF_CASE(3, 64); RVC_RBF_CASE(7, 64); RVC_RBF_CASE(11, 64);
    RVC_RBF_CASE(3, 128); RVC_RBF_CASE(7, 128);
#undef RVC_RBF_CASE
    return fail("resblock_bf: unsupported shape c=%d k=%d", c, k);
}

}  // namespace rvc

using namespace rvc;

static int rbf_weight_bytes_abi(const char *fn, int c, int k, int tap_splits, size_t *bytes) {
    if (!bytes) return fail("%s: null pointer", fn);
    if (!resblock_bf_supported(c, k, 1)) return fail("%s: c must be 32 or 64 with k 3, 7 or 11, or 128 with k 3 or 7", fn);
    *bytes = resblock_bf_weight_bytes(c, k, tap_splits);
    return 0;
}
static int rbf_pack_abi(const char *fn, const float *w1_host, const float *w2_host, int c, int k, int tap_splits, void *u_dev, void *stream) {
    if (!w1_host || !w2_host || !u_dev) return fail("%s: null pointer", fn);
    size_t bytes = 0;
    if (rbf_weight_bytes_abi(fn, c, k, tap_splits, &bytes)) return 1;
    std::vector<uint16_t> u;
    resblock_bf_pack_host(w1_host, w2_host, c, k, &u, tap_splits);
    hipError_t e = hipMemcpyAsync(u_dev, u.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail("%s: %s", fn, hipGetErrorString(e));
    return 0;
}

extern "C" int rvc_resblock_bf16x3_weight_bytes(int c, int k, size_t *bytes) { return rbf_weight_bytes_abi("rvc_resblock_bf16x3_weight_bytes", c, k, 3, bytes); }

extern "C" int rvc_resblock_bf16x3_pack_weight(const float *w1_host, const float *w2_host, int c, int k, void *u_dev, void *stream) {
    return rbf_pack_abi("rvc_resblock_bf16x3_pack_weight", w1_host, w2_host, c, k, 3, u_dev, stream);
}

extern "C" int rvc_resblock_bf16x3_forward(const float *x_dev, const void *u_dev, const float *b1_dev, const float *b2_dev,
                                           const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                                           float slope, float out_scale, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_resblock_bf16x3_forward: null pointer");
    return launch_resblock_bf(x_dev, u_dev, b1_dev, b2_dev, acc_dev, y_dev, batch, c, length, k, dilation, slope, out_scale,
                              (hipStream_t)stream, 3);
}

// ... with bf16-VALUED taps (BASELINE cfg 4's weight storage): one-term tap split, three products per multiply-add
extern "C" int rvc_resblock_bf16w_weight_bytes(int c, int k, size_t *bytes) { return rbf_weight_bytes_abi("rvc_resblock_bf16w_weight_bytes", c, k, 1, bytes); }

extern "C" int rvc_resblock_bf16w_pack_weight(const float *w1_host, const float *w2_host, int c, int k, void *u_dev, void *stream) {
    return rbf_pack_abi("rvc_resblock_bf16w_pack_weight", w1_host, w2_host, c, k, 1, u_dev, stream);
}

extern "C" int rvc_resblock_bf16w_forward(const float *x_dev, const void *u_dev, const float *b1_dev, const float *b2_dev,
                                          const float *acc_dev, float *y_dev, int batch, int c, int64_t length, int k, int dilation,
                                          float slope, float out_scale, void *stream) {
    if (!x_dev || !u_dev || !y_dev) return fail("rvc_resblock_bf16w_forward: null pointer");
    return launch_resblock_bf(x_dev, u_dev, b1_dev, b2_dev, acc_dev, y_dev, batch, c, length, k, dilation, slope, out_scale,
                              (hipStream_t)stream, 1);
}

extern "C" int rvc_resblock_bf16x3_set_enabled(int enabled) {
    resblock_bf_set_enabled(enabled != 0);
    return 0;
}
