// K3z -- winobf2.hip's kernel with the matrix pipe BALANCED over the four SIMDs.
//
// winobf2.hip (K3y) gives wave w < 7 transform point w of a 128-channel x 64-column block and makes wave 7 the loader.  Seven
// compute waves on four SIMDs: three SIMDs carry two of them, the fourth one, and a chunk ends in a barrier -- the K loop runs
// at the pace of the SIMDs with 2 x 8 accumulator tiles while the fourth has 8 (profiles/r04_winobf2_stamps.txt: 10 900 cycles
// per 16-channel chunk against a matrix-pipe floor of 9 200 on the busy SIMDs).
//
// Here the 7 points x 8 tiles = 56 (point, tile) products of a block are dealt 7 to each of EIGHT waves:
//   * wave w < NP (a "point wave") keeps point w but leaves its LAST tile -- row block 3, column tile 1 -- to wave 7;
//   * wave 7 (the "diagonal wave") computes that tile for every point: per tap group and point three tap fragments from L2 and
//     three window fragments from LDS (the point wave's own, which it may read after the chunk's barrier; the halo windows are its
//     own work), six products -- 1 KiB of fragments per matrix instruction where a point wave needs 0.43, but for one wave in eight;
//   * with three taps (six points) waves 6 and 7 take row block 3's two tiles of every point and the point waves keep six.
//   Every SIMD now carries 2 x 7 tiles: 252 instead of 288 matrix instructions per SIMD, chunk and tap group.
//   * Nobody is "the loader": every wave stages ONE channel pair of every chunk (two rows: 12 one-dword loads a phase ahead, six
//     LDS writes), wave 7 also transforms the (G - 1) d halo windows (it has no point of its own to transform) -- all 7 points of a
//     halo window from ONE read of its samples.
// Everything else -- arithmetic, LDS layout, tap-fragment slab (point-major, winobf_pack_host), epilogue -- is winobf2.hip's.
//
// MEASURED (MI355X, same box, tools/ab_w2.py, C = 128, 383 760 columns): parity-green on the 27 shapes of
// test_conv1d_winograd_bf16x3_matches_float64 and SLOWER than winobf2.hip -- 480 against 403 us at 11 taps, 380 / 319 at 7, 234 / 222
// at 3; with the diagonal wave switched off altogether (RVC_W3_DBG=31) still 430.  A wave's loads return in order behind ONE counter:
// a point wave that also fetches raw rows (HBM, ~5 000 cycles under this kernel's own traffic) waits that long for the tap fragments
// (L2, a few hundred cycles) it requested behind them, once per chunk, whatever the place of the requests in the phase and however
// the SIMD's two waves are staggered; and the diagonal wave's 1 KiB of fragments per matrix instruction makes it the slowest of the
// block.  The dedicated loader wave of winobf2.hip is what keeps the compute waves' load queue short-latency.  This file is therefore
// compiled into the ablation build only (RVC_WBF_V3=1 routes the c_out % 128 == 0 layers here); the product runs winobf2.hip.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "winobf2.h"

#ifndef RVC_ABLATE
namespace rvc {
bool winobf3_enabled() { return false; }
bool winobf3_supported(int, int, int, int) { return false; }
int launch_winobf3_conv(const Wbf2Params &, int, hipStream_t) { return fail("winobf3 conv: ablation build only"); }
}  // namespace rvc
#else
namespace rvc {

// the tables of winobf2.h once more as compile-time constants: the diagonal wave's halo transform uses them as immediates
constexpr float W3_BT[7][7] = {
    {-0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},  {0.f, 0.5f, 0.25f, -2.25f, -1.f, 1.f, 0.f}, {0.f, -0.5f, 0.75f, 1.75f, -3.f, 1.f, 0.f},
    {0.f, 1.f, 1.5f, -2.f, -1.5f, 1.f, 0.f},       {0.f, -1.f, 2.5f, 0.f, -2.5f, 1.f, 0.f},    {0.f, 0.25f, 0.f, -1.25f, 0.f, 1.f, 0.f},
    {0.f, -0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f},
};

// DBG (ablation build only; wrong results), the diagonal wave without: 1 its halo transform, 2 its matrix instructions, 4 its tap loads,
// 8 its window-fragment reads in the loop, 16 its share of the staging
template <int KW, int DBG = 0>
__global__ void __launch_bounds__(W2_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
winobf3_conv_kernel(const Wbf2Params p) {
    constexpr int BM = 128;
    using GM = W2Geom<KW, BM>;
    constexpr int NP = GM::NP, G = GM::G, C0 = GM::C0, MLO = GM::MLO, MHI = GM::MHI, CP = W2_CP, CIC = W2_CIC;
    constexpr int XT = GM::XT, XTS = GM::XTS, NJ = GM::NJ, BNT = W2_BNT;
    constexpr int NPAIR = 2;                              // row-block pairs (two 32-channel blocks each)
    constexpr int NSA = 3;                                // bf16 numbers per transformed tap (exact split of an fp32 value)
    constexpr int NPROD = 6;                              // products per (tap group, accumulator tile): all of order <= 2^-16
    constexpr int ND = 8 - NP;                            // diagonal waves: 1 (7 points) or 2 (6 points)
    static_assert(ND == 1 || ND == 2, "");

    extern __shared__ __attribute__((aligned(16))) float w3_smem[];
    unsigned char *const smem = reinterpret_cast<unsigned char *>(w3_smem);
    w2_f32x2 *const xs = reinterpret_cast<w2_f32x2 *>(smem);              // raw chunks [2][CP][4][XTS]
    constexpr int XRAW = CP * 4 * XTS;                                     // float2 per raw buffer
    unsigned char *const bs_all = smem + 2 * GM::RAW_BYTES;                // [point][2][B_WAVE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 8);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int n_m = p.c_out / BM;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile_blk = (seq / n_m) * 8 + xcd;               // the n_m channel blocks of one time tile: ids 8 apart, same XCD
    if (tile_blk >= p.n_tile_blocks) return;
    const int mblk = seq % n_m;
    const int m0 = mblk * BM;
    const int d = p.dil;
    const int64_t sb0 = (int64_t)tile_blk * p.sb_per_block;
    const int n_tiles_blk = p.sb_per_block * d;               // valid output tiles (columns) of this block
    const int64_t L = p.L;
    const int c_in = p.c_in, c_out = p.c_out;
    const int n_chunks = c_in / CIC;
    const int last = n_chunks - 1;
    const bool diag = wave >= NP;
    const int dcb = 1 - (wave - NP);                          // a diagonal wave's tile: row block 3, column tile dcb (wave NP: 1, wave NP + 1: 0)

    // point wave: tile (row block, column tile) of pair pr -> acc[pr * 4 + rb * 2 + cb]; diagonal wave: point pt -> acc[pt]
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- what the epilogue needs (winobf2.hip) --------------------------------------------------------------------------------
    const float *bias = p.bias;
    const float *res = p.res ? p.res + (int64_t)b * c_out * L : nullptr;
    const float *accin = p.accin ? p.accin + (int64_t)b * c_out * L : nullptr;
    float *y = p.y + (int64_t)b * c_out * L;
    const float out_scale = p.out_scale;
    f32x4 *const red = reinterpret_cast<f32x4 *>(smem);
    float *const yt = w3_smem;
    constexpr int YS = GM::YS;
    const bool l4 = (L & 3) == 0;
    const bool direct = d == 1 && l4;
    const int64_t t_blk0 = sb0 * 4 * d;
    const int64_t left = L - t_blk0;
    const int n_t = (int)(left < 4 * n_tiles_blk ? left : 4 * n_tiles_blk);   // valid outputs per row in this block

    // ---- raw-row staging: wave w stages channel pair w of every chunk ----------------------------------------------------------
    // slot k < NJ of a row is the sample t_start + lane + 64 k (one dword; 2 rows x NJ loads per wave and chunk -- few enough that
    // their width does not matter, so one form serves every L and every block, conv padding applied through a per-lane bit mask)
    const float *const px = p.x + (int64_t)b * c_in * L;
    const float slope = p.slope;
    const int E = (G - 1) * d;                                     // halo windows
    const int xt_used = (p.sb_per_block + G - 1 - MLO + MHI) * d;  // valid tiles + the (G - 1) d windows behind them + the -MLO d in front (+ MHI d behind)
    const int64_t t_start = (sb0 + MLO) * 4 * d;
    const int span = 4 * xt_used;                                  // staged samples per row (a multiple of 4, <= 4 XT)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)px, 0, (int)((int64_t)c_in * L * 4), W2_RSRC_FLAGS);
    const int L4 = (int)(L * 4);
    const int t0 = (int)t_start + lane;
    const int Lm1 = (int)L - 1;
    // (where a sample goes in its LDS row is recomputed at every write -- ten scalar-operand instructions per slot -- rather than
    // held in NJ registers through the K loop: the point waves have none to spare)
    const unsigned inv4d = 65536u / (unsigned)(4 * d) + 1u, invd = 65536u / (unsigned)d + 1u;   // x / m as a multiply-shift (winobf2.hip)
    static_assert(4 * XT < 2048, "");
    auto lds_slot = [&](int k) __attribute__((always_inline)) {
        const int tl = lane + 64 * k;
        const bool have = tl < span;
        const int tlc = have ? tl : 0;
        const int sbl = (int)(((unsigned)tlc * inv4d) >> 16);
        const int r = tlc - sbl * 4 * d;
        const int ii = (int)(((unsigned)r * invd) >> 16);
        const int phi = r - ii * d;
        return have ? ii * XTS + sbl * d + phi : XT + (lane & 1);          // columns >= XT of a row are never read
    };
    unsigned keepm = 0;
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const int t = t0 + 64 * k;
        if (lane + 64 * k < span && t >= 0 && t <= Lm1) keepm |= 1u << k;  // else: conv zero padding
    }
    float ra[2][NJ], rb2[2][NJ];                                  // two rows of a chunk in flight (rb2: the prologue's second chunk)
    auto pair_load = [&](int c, float (&r)[2][NJ]) __attribute__((always_inline)) {
        const int cc = c < last ? c : last;
        const int s0 = (cc * CIC + 2 * wave) * L4;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            int t = t0 + 64 * k;
            t = t < 0 ? 0 : (t > Lm1 ? Lm1 : t);
            r[0][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, t * 4, s0, 0));
            r[1][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, t * 4, s0 + L4, 0));
        }
    };
    auto pair_write = [&](int c, float (&r)[2][NJ]) __attribute__((always_inline)) {   // into raw buffer c & 1
        w2_f32x2 *const dst = xs + (c & 1) * XRAW + wave * 4 * XTS;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const w2_f32x2 v = w2_lrelu2(w2_f32x2{r[0][k], r[1][k]}, slope);
            const unsigned m = (keepm >> k) & 1u ? 0xffffffffu : 0u;
            dst[lds_slot(k)] = w2_f32x2{__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m)};
        }
    };

    // tap fragments: [c_out / BM][point][chunk][group][row block][split][lane][8 bf16], 1 KiB each (winobf_pack_host, point-major)
    const __amdgpu_buffer_rsrc_t urs =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)((int64_t)c_in * c_out * NP * G * NSA * 2), W2_RSRC_FLAGS);
    constexpr int A_GROUP = 2 * NSA * 1024;                       // one (chunk, tap group, row-block pair) of one point
    const int n_groups = n_chunks * G * NPAIR;
    constexpr int ia6[6] = {0, 1, 0, 2, 1, 0}, ib6[6] = {2, 1, 1, 0, 0, 0};

    pair_load(0, ra);
    pair_load(1, rb2);

    if (!diag) {
        // ================================ waves 0 .. NP-1: point `wave`, all tiles but row block 3's last ====================
        const int pt = wave;
        unsigned char *const bs = bs_all + pt * 2 * GM::B_WAVE;
        float bt[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) bt[n] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, KW == 3 ? W2_BT3[pt][n] : W2_BT[pt][n])));
        const int a_base = (mblk * NP + pt) * n_groups * A_GROUP;
        const int a_last = a_base + (n_groups - 1) * A_GROUP;
        w2_bf16x8 fa[2][2][NSA];                           // [buffer][row block of the pair][split]
        // group Q of this wave's stream (clamped: the tail re-reads the last one); PR1: a second pair -- with two diagonal waves its
        // row block 1 (the block's row block 3) is theirs altogether
        auto load_a = [&](int buf, int Q, bool pr1) __attribute__((always_inline)) {
            int soff = a_base + Q * A_GROUP;
            soff = soff < a_last ? soff : a_last;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if (ND == 2 && pr1 && rb == 1) continue;
#pragma unroll
                for (int sp = 0; sp < NSA; ++sp)
                    fa[buf][rb][sp] = __builtin_bit_cast(w2_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + (rb * NSA + sp) * 1024, soff, 0));
            }
        };
        w2_bf16x8 fb[2][3];                                // window fragments of one tap group: [column tile][split], refilled split by split
        const int b_lane = half * GM::B_PLANE + l31 * 16;
        auto read_b = [&](int c, int g, int sp) __attribute__((always_inline)) {
            const unsigned char *bb = bs + (c & 1) * GM::B_WAVE + sp * 2 * GM::B_PLANE + b_lane + g * d * 16;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) fb[cb][sp] = __builtin_bit_cast(w2_bf16x8, *reinterpret_cast<const w2_u32x4 *>(bb + cb * 32 * 16));
        };
        // ---- input transform of chunk c for this point: lane = window, unit u = channel pairs 2u, 2u + 1 (winobf2.hip) ---------
        const int t_src = -MLO * d + lane;
        w2_f32x2 tq[NP];                                   // one channel pair's samples at a time (registers)
        w2_f32x2 tv[2];
        unsigned tw[3][2];
        auto t_read = [&](int c, int u, int e) __attribute__((always_inline)) {
            const w2_f32x2 *const raw = xs + (c & 1) * XRAW + t_src;
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int sh = n - C0;
                tq[n] = raw[((2 * u + e) * 4 + (sh & 3)) * XTS + (sh >> 2) * d];
            }
        };
        auto t_xform = [&](int e) __attribute__((always_inline)) {   // two independent half sums
            w2_f32x2 a = tq[0] * bt[0], b2 = tq[1] * bt[1];
            a = __builtin_elementwise_fma(w2_f32x2{bt[2], bt[2]}, tq[2], a);
            b2 = __builtin_elementwise_fma(w2_f32x2{bt[3], bt[3]}, tq[3], b2);
            a = __builtin_elementwise_fma(w2_f32x2{bt[4], bt[4]}, tq[4], a);
            b2 = __builtin_elementwise_fma(w2_f32x2{bt[5], bt[5]}, tq[5], b2);
            if constexpr (NP == 7) a = __builtin_elementwise_fma(w2_f32x2{bt[6], bt[6]}, tq[6], a);
            tv[e] = a + b2;
        };
        auto t_split = [&](int e, int level) __attribute__((always_inline)) {
            const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(tv[e], w2_bf16x2));
            tw[level][e] = w;
            if (level < 2) tv[e] = tv[e] - w2_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        };
        auto t_write = [&](int c, int u) __attribute__((always_inline)) {
            unsigned char *o = bs + (c & 1) * GM::B_WAVE + (u >> 1) * GM::B_PLANE + lane * 16 + (u & 1) * 8;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<w2_u32x2 *>(o + sp * 2 * GM::B_PLANE) = w2_u32x2{tw[sp][0], tw[sp][1]};
        };
        auto transform_unit = [&](int c, int u) __attribute__((always_inline)) {
            t_read(c, u, 0); t_xform(0);
            t_read(c, u, 1); t_xform(1);
#pragma unroll
            for (int e = 0; e < 2; ++e) { t_split(e, 0); t_split(e, 1); t_split(e, 2); }
            t_write(c, u);
        };

        // ---- prologue ------------------------------------------------------------------------------------------------------
        load_a(0, 0, false);
        pair_write(0, ra);
        pair_load(2, ra);
        lds_barrier();                                    // (P1) chunk 0's rows are in LDS
#pragma unroll
        for (int u = 0; u < 4; ++u) transform_unit(0, u);
        pair_write(1, rb2);
        load_a(1, 1, true);
        read_b(0, 0, 2); read_b(0, 0, 1); read_b(0, 0, 0);
        lds_barrier();                                    // (P2) chunk 1's rows; chunk 0 is transformed

        // ---- main loop: one phase per chunk (winobf2.hip's, minus the tiles that went to the diagonal waves, plus this wave's
        //      channel pair of chunk c + 2: written from the registers it was fetched into a phase ago, which then take chunk c + 3) --
        constexpr int NQ = G * NPAIR;
        static_assert(NQ >= 2, "");
        auto phase = [&](int c, auto PAR) __attribute__((always_inline)) {
            constexpr int par = decltype(PAR)::value;
            const int cn = c + 1 < n_chunks ? c + 1 : c;   // the last phase re-transforms its own chunk (same bits, never needed)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int g = q / NPAIR, pr = q % NPAIR;
                const int bufq = (q + par) & 1;
                const bool b_turn = pr == NPAIR - 1;
                const int gb = (g + 1) % G, cb_c = g + 1 < G ? c : c + 1;
#pragma unroll
                for (int i = 0; i < NPROD; ++i) {
                    const int ia = ia6[i], ib = ib6[i];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            const bool mine = !(pr == 1 && rb == 1 && cb >= 2 - ND);
                            if (mine) acc[pr * 4 + rb * 2 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bufq][rb][ia], fb[cb][ib], acc[pr * 4 + rb * 2 + cb], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            const int k = (i * 2 + rb) * 2 + cb;            // 0 .. 23: fillers behind this slot
#pragma unroll
                            for (int u = 0; u < 4; ++u) {                   // transform unit u lives in group u (NQ - 1) / 4, six slots each
                                if (u * (NQ - 1) / 4 != q) continue;
                                int first = u;
                                while (first > 0 && (first - 1) * (NQ - 1) / 4 == q) --first;
                                const int k0 = (u - first) * 6;
                                if (k == k0) t_read(cn, u, 0);
                                if (k == k0 + 1) { t_xform(0); t_read(cn, u, 1); }
                                if (k == k0 + 2) { t_split(0, 0); t_split(0, 1); t_split(0, 2); }
                                if (k == k0 + 3) t_xform(1);
                                if (k == k0 + 4) { t_split(1, 0); t_split(1, 1); t_split(1, 2); }
                                if (k == k0 + 5) t_write(cn, u);
                            }
                            if (b_turn) {
                                if (k == 3) read_b(cb_c, gb, 2);            // split 2: product 0 only
                                if (k == 11) read_b(cb_c, gb, 1);           // split 1: products 1, 2
                                if (k == 23) read_b(cb_c, gb, 0);           // split 0: products 3..5
                            }
                            if (k == 4 * NPROD - 1) load_a(bufq, c * NQ + q + 2, pr == 1);
                            // staging: a SIMD's two waves at different times of the phase (their loads' HBM latency holds up the tap
                            // loads issued behind them -- one counter, in order -- and the partner should have the pipe meanwhile)
                            if (q == (wave < 4 ? NQ - 1 : NQ / 2 - 1)) {
                                if (k == 9) pair_write(c + 2, ra);
                                if (k == 4 * NPROD - 1) pair_load(c + 3, ra);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
            lds_barrier();                                // the raw rows of chunk c + 2 are in LDS; every wave's fragments of chunk c + 1 too
        };
        for (int c = 0; c < n_chunks; c += 2) {
            phase(c, std::integral_constant<int, 0>{});
            if (c + 1 < n_chunks) phase(c + 1, std::integral_constant<int, NQ & 1>{});
        }
    } else {
        // ================================ wave(s) NP .. 7: row block 3, column tile dcb of EVERY point ========================
        // step s = (tap group g, point pt) of a chunk: six products on acc[pt].  Tap fragments (row block 1 of the point's pair 1)
        // come from L2 two steps ahead, window fragments from the point wave's LDS area split by split behind the last product
        // that used the previous step's -- except across a chunk's barrier, behind which the first step's are read.
        constexpr int NS = G * NP;                         // steps per chunk
        const int a_blk = mblk * NP * n_groups * A_GROUP;
        w2_bf16x8 fa[2][NSA];
        w2_bf16x8 fb[NSA];
        auto load_fa = [&](int buf, int c, int s) __attribute__((always_inline)) {   // step s (may run past the chunk: s >= NS is the next chunk's)
            const int cw = s >= NS ? c + 1 : c, sw = s >= NS ? s - NS : s;
            const int cc = cw < last ? cw : last;
            const int g = sw / NP, pt = sw % NP;
            const int soff = a_blk + (pt * n_groups + (cc * G + g) * NPAIR + 1) * A_GROUP + NSA * 1024;
#pragma unroll
            for (int sp = 0; sp < NSA; ++sp)
                fa[buf][sp] = __builtin_bit_cast(w2_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + sp * 1024, soff, 0));
        };
        const int b_lane = half * GM::B_PLANE + l31 * 16 + dcb * 32 * 16;
        auto read_fb = [&](int c, int s, int sp) __attribute__((always_inline)) {
            const int g = s / NP, pt = s % NP;
            fb[sp] = __builtin_bit_cast(w2_bf16x8, *reinterpret_cast<const w2_u32x4 *>(bs_all + pt * 2 * GM::B_WAVE + (c & 1) * GM::B_WAVE + sp * 2 * GM::B_PLANE + b_lane + g * d * 16));
        };
        // ---- halo windows 64 .. 63 + E of every point (wave 7, G > 1): lane = (window e = lane / 4, unit hu = lane % 4 = channel pairs
        //      2 hu, 2 hu + 1); the window's samples are read ONCE and transformed for the 7 points in turn.  Lanes >= 4 E do the same work
        //      on window 64 and write it to a raw row's unread tail.
        const bool h_on = lane < 4 * E;
        const int he = h_on ? lane >> 2 : 0, hu = lane & 3;
        const int h_src = (2 * hu) * 4 * XTS - MLO * d + BNT + he;
        const int h_dst = (hu >> 1) * GM::B_PLANE + (BNT + he) * 16 + (hu & 1) * 8;
        w2_f32x2 hq[2][7];
        w2_f32x2 hv[2];
        unsigned hw[3][2];
        auto h_read = [&](int c) __attribute__((always_inline)) {
            const w2_f32x2 *const raw = xs + (c & 1) * XRAW + h_src;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int n = 0; n < 7; ++n) {
                    const int sh = n - C0;
                    hq[e2][n] = raw[(e2 * 4 + (sh & 3)) * XTS + (sh >> 2) * d];
                }
        };
        auto h_xform_c = [&](auto HP) __attribute__((always_inline)) {   // the point waves' expression, term for term; coefficients as immediates
            constexpr int hp = decltype(HP)::value;
            constexpr float c0 = W3_BT[hp][0], c1 = W3_BT[hp][1], c2 = W3_BT[hp][2], c3 = W3_BT[hp][3], c4 = W3_BT[hp][4], c5 = W3_BT[hp][5], c6 = W3_BT[hp][6];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                w2_f32x2 a = hq[e2][0] * c0, b2 = hq[e2][1] * c1;
                a = __builtin_elementwise_fma(w2_f32x2{c2, c2}, hq[e2][2], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{c3, c3}, hq[e2][3], b2);
                a = __builtin_elementwise_fma(w2_f32x2{c4, c4}, hq[e2][4], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{c5, c5}, hq[e2][5], b2);
                a = __builtin_elementwise_fma(w2_f32x2{c6, c6}, hq[e2][6], a);
                hv[e2] = a + b2;
            }
        };
        auto h_xform = [&](int hp) __attribute__((always_inline)) {      // (hp is a constant wherever this is called: the switch folds)
            switch (hp) {
                case 0: h_xform_c(std::integral_constant<int, 0>{}); break;
                case 1: h_xform_c(std::integral_constant<int, 1>{}); break;
                case 2: h_xform_c(std::integral_constant<int, 2>{}); break;
                case 3: h_xform_c(std::integral_constant<int, 3>{}); break;
                case 4: h_xform_c(std::integral_constant<int, 4>{}); break;
                case 5: h_xform_c(std::integral_constant<int, 5>{}); break;
                default: h_xform_c(std::integral_constant<int, 6>{}); break;
            }
        };
        auto h_split = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int level = 0; level < 3; ++level) {
                    const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(hv[e2], w2_bf16x2));
                    hw[level][e2] = w;
                    if (level < 2) hv[e2] = hv[e2] - w2_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                }
        };
        auto h_write = [&](int c, int hp) __attribute__((always_inline)) {
            unsigned char *o = h_on ? bs_all + hp * 2 * GM::B_WAVE + (c & 1) * GM::B_WAVE + h_dst
                                    : reinterpret_cast<unsigned char *>(xs + XT);           // (never read)
            const int step = h_on ? 2 * GM::B_PLANE : 0;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<w2_u32x2 *>(o + sp * step) = w2_u32x2{hw[sp][0], hw[sp][1]};
        };
        auto halo_all = [&](int c) __attribute__((always_inline)) {
            if constexpr (G > 1) {
                h_read(c);
#pragma unroll
                for (int hp = 0; hp < 7; ++hp) { h_xform(hp); h_split(); h_write(c, hp); }
            }
        };

        // ---- prologue ------------------------------------------------------------------------------------------------------
        load_fa(0, 0, 0);
        load_fa(1, 0, 1);
        pair_write(0, ra);
        pair_load(2, ra);
        lds_barrier();                                    // (P1)
        halo_all(0);
        pair_write(1, rb2);
        lds_barrier();                                    // (P2)

        auto dphase = [&](int c, auto PAR) __attribute__((always_inline)) {
            constexpr int par = decltype(PAR)::value;
            const int cn = c + 1 < n_chunks ? c + 1 : c;
            read_fb(c, 0, 2); read_fb(c, 0, 1); read_fb(c, 0, 0);   // (behind the barrier: other waves' fragments)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int pt = s % NP;
                const int buf = (s + par) & 1;
#pragma unroll
                for (int i = 0; i < NPROD; ++i) {
                    if (!(DBG & 2)) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][ia6[i]], fb[ib6[i]], acc[pt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < NS && !(DBG & 8)) {                // the next step's window fragments, each split behind its last product
                        if (i == 0) read_fb(c, s + 1, 2);
                        if (i == 2) read_fb(c, s + 1, 1);
                        if (i == 5) read_fb(c, s + 1, 0);
                    }
                    if (i == 5 && !(DBG & 4)) load_fa(buf, c, s + 2);
                    if constexpr (G > 1 && !(DBG & 1)) {           // chunk c + 1's halo windows: the samples in step 0, point hp in step 1 + hp
                        if (s == 0 && i == 1) h_read(cn);
                        if (s >= 1 && s <= 7) {
                            if (i == 1) h_xform(s - 1);
                            if (i == 3) h_split();
                            if (i == 4) h_write(cn, s - 1);
                        }
                    }
                    if (s == NS / 2 && !(DBG & 16)) {
                        if (i == 1) pair_write(c + 2, ra);
                        if (i == 5) pair_load(c + 3, ra);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            lds_barrier();
        };
        static_assert(G == 1 || NS >= 9, "the halo's eight pieces need eight steps");
        for (int c = 0; c < n_chunks; c += 2) {
            dphase(c, std::integral_constant<int, 0>{});
            if (c + 1 < n_chunks) dphase(c + 1, std::integral_constant<int, NS & 1>{});
        }
    }

    // ================================ epilogue: output transform across the waves (winobf2.hip) ==============================
    // bias first, then the residual (a wave's loads return in order), both for every pass
    f32x4 bq_all[NPAIR][2];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = tid + gi * W2_NTH;
            const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
            const int row0 = (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5);
            bq_all[pr][gi] = bias ? *reinterpret_cast<const f32x4 *>(bias + m0 + pr * 64 + row0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 pre_r[NPAIR][8];                                // direct: [group 2][channel 4]; tile path: [row pass 8]
    if (res) {
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            if (direct) {
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int grp = tid + gi * W2_NTH;
                    const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
                    const int col = (tile & 1) * 32 + (ln & 31);
                    const int64_t tt = (sb0 + col) * 4;
                    const bool ok = col < n_tiles_blk && tt < L;
                    const int64_t base = (int64_t)(m0 + pr * 64 + (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5)) * L + (ok ? tt : 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pre_r[pr][gi * 4 + e] = *reinterpret_cast<const f32x4 *>(res + base + (int64_t)e * L);
                }
            } else if (l4) {
                constexpr int RPP = W2_NTH / BNT;
                const int tq4 = (tid % BNT) * 4, rq = tid / BNT;
                const int64_t base = (int64_t)(m0 + pr * 64 + rq) * L + (tq4 < n_t ? t_blk0 + tq4 : 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) pre_r[pr][k] = *reinterpret_cast<const f32x4 *>(res + base + (int64_t)k * RPP * L);
            }
        }
    }
    f32x4 outv[NPAIR][8];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        if (pr > 0) lds_barrier();                        // the previous pass's tile has been read back
        // red[point][tile = rb * 2 + cb of the pair][r >> 2][lane][r & 3]
        if (!diag) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (pr == 1 && rb == 1 && cb >= 2 - ND) continue;       // a diagonal wave's
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        red[((wave * 4 + rb * 2 + cb) * 4 + rq) * 64 + lane] =
                            f32x4{acc[pr * 4 + rb * 2 + cb][4 * rq], acc[pr * 4 + rb * 2 + cb][4 * rq + 1], acc[pr * 4 + rb * 2 + cb][4 * rq + 2], acc[pr * 4 + rb * 2 + cb][4 * rq + 3]};
                }
        } else if (pr == 1) {
#pragma unroll
            for (int pt = 0; pt < NP; ++pt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    red[((pt * 4 + 2 + dcb) * 4 + rq) * 64 + lane] = f32x4{acc[pt][4 * rq], acc[pt][4 * rq + 1], acc[pt][4 * rq + 2], acc[pt][4 * rq + 3]};
        }
        lds_barrier();
        f32x4 o[2][4];                                    // [group][channel of the quad] -> 4 outputs
        int g_row[2], g_col[2];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = tid + gi * W2_NTH;            // (tile, rq, lane'): 4 * 4 * 64 = 1024 groups
            const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
            const int rb = tile >> 1, cb = tile & 1;
            f32x4 v[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) v[q] = red[((q * 4 + tile) * 4 + rq) * 64 + ln];
            g_row[gi] = rb * 32 + 8 * rq + 4 * (ln >> 5);                // + comp: channel inside the pair's 64
            g_col[gi] = cb * 32 + (ln & 31);
            const f32x4 bq = bq_all[pr][gi];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = bq[e];
                if constexpr (KW == 3) {   // F(4,3): A^T diag(1/4, -1/6, -1/6, 1/24, 1/24, 1), wino.hip's expression
                    const float d0 = v[0][e], d1 = v[1][e], d2 = v[2][e], d3 = v[3][e], d4 = v[4][e], d5 = v[5][e];
                    const float s12 = (d1 + d2) * (-1.f / 6.f), m12 = (d1 - d2) * (-1.f / 6.f), s34 = (d3 + d4) * (1.f / 24.f), m34 = (d3 - d4) * (1.f / 24.f);
                    o[gi][e].x = fmaf(0.25f, d0, s12 + s34) + bv;
                    o[gi][e].y = fmaf(2.f, m34, m12) + bv;
                    o[gi][e].z = fmaf(4.f, s34, s12) + bv;
                    o[gi][e].w = fmaf(8.f, m34, m12) + d5 + bv;
                } else {
                    const float t0v = v[0][e] * -2.f, t1 = v[1][e] * (-2.f / 3.f), t2 = v[2][e] * (-2.f / 9.f), t3 = v[3][e] * (16.f / 9.f),
                                t4 = v[4][e] * (16.f / 15.f), t5 = v[5][e] * (2.f / 45.f), t6 = v[NP - 1][e];
                    const float s12 = t1 + t2, m12 = t1 - t2, s34 = t3 + t4, m34 = t3 - t4;
                    o[gi][e].x = (t0v + s12) + (s34 + t5) + bv;
                    o[gi][e].y = fmaf(0.5f, m34, m12) + fmaf(2.f, t5, bv);
                    o[gi][e].z = fmaf(0.25f, s34, s12) + fmaf(4.f, t5, bv);
                    o[gi][e].w = fmaf(0.125f, m34, m12) + fmaf(8.f, t5, t6) + bv;
                }
            }
        }
        if (direct) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int e = 0; e < 4; ++e) outv[pr][gi * 4 + e] = o[gi][e];
            continue;
        }
        lds_barrier();                                    // every thread has read its points: the tile may overlay them
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int col = g_col[gi];
            const int sbl = col / d;
            const int tl0 = sbl * 4 * d + (col - sbl * d);
            if (col < n_tiles_blk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float *dst = yt + (g_row[gi] + e) * YS + tl0;
                    dst[0] = o[gi][e].x; dst[d] = o[gi][e].y; dst[2 * d] = o[gi][e].z; dst[3 * d] = o[gi][e].w;
                }
            }
        }
        lds_barrier();
        if (l4) {   // 16-byte pieces: 64 threads per row, 8 rows per pass
            constexpr int RPP = W2_NTH / BNT, PASSES = 64 / RPP;
            static_assert(PASSES == 8, "a thread holds eight rows of a pass");
            const int tq4 = (tid % BNT) * 4, rq = tid / BNT;
#pragma unroll
            for (int k = 0; k < PASSES; ++k) outv[pr][k] = *reinterpret_cast<const f32x4 *>(yt + (rq + k * RPP) * YS + tq4);
        } else {
            for (int e = tid; e < 64 * 4 * BNT; e += W2_NTH) {
                const int rq = e / (4 * BNT), tq1 = e - rq * (4 * BNT);
                if (tq1 >= n_t) continue;
                const int64_t at = (int64_t)(m0 + pr * 64 + rq) * L + t_blk0 + tq1;
                float v = yt[rq * YS + tq1];
                if (res) v += res[at];
                if (accin) v += accin[at];
                y[at] = v * out_scale;
            }
        }
    }
    // ---- residual, running sum, scale, stores -------------------------------------------------------------------------------
    if (direct) {
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int grp = tid + gi * W2_NTH;
                const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
                const int col = (tile & 1) * 32 + (ln & 31);
                const int64_t tt = (sb0 + col) * 4;
                if (col < n_tiles_blk && tt < L) {
                    const int64_t base = (int64_t)(m0 + pr * 64 + (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5)) * L + tt;
                    f32x4 ov[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = outv[pr][gi * 4 + e];
                    if (res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] += pre_r[pr][gi * 4 + e];
                    }
                    if (accin) {
                        f32x4 av[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[e] = *reinterpret_cast<const f32x4 *>(accin + base + (int64_t)e * L);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] += av[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4 *>(y + base + (int64_t)e * L) = ov[e] * out_scale;
                }
            }
    } else if (l4) {
        constexpr int RPP = W2_NTH / BNT;
        const int tq4 = (tid % BNT) * 4, rq = tid / BNT;
        if (tq4 < n_t) {
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const int64_t base = (int64_t)(m0 + pr * 64 + rq) * L + t_blk0 + tq4;
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = outv[pr][k];
                if (res) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += pre_r[pr][k];
                }
                if (accin) {
                    f32x4 av[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) av[k] = *reinterpret_cast<const f32x4 *>(accin + base + (int64_t)k * RPP * L);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += av[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4 *>(y + base + (int64_t)k * RPP * L) = v[k] * out_scale;
            }
        }
    }
}

template <int KW, int DBG = 0>
static int winobf3_launch(Wbf2Params p, hipStream_t stream) {
    using GM = W2Geom<KW, 128>;
    p.sb_per_block = W2_BNT / p.dil;   // every accumulator column a valid tile (the (G - 1) d windows behind the 64 are wave 7's)
    const int64_t n_sb = ceil_div(p.L, (int64_t)4 * p.dil);
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute((const void *)winobf3_conv_kernel<KW, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    });
    if (err != hipSuccess) return fail("winobf3 conv: cannot reserve %d bytes of LDS: %s", GM::LDS_BYTES, hipGetErrorString(err));
    p.n_tile_blocks = (int)ceil_div(n_sb, p.sb_per_block);
    const int n_m = p.c_out / 128;
    dim3 grid((unsigned)(ceil_div(p.n_tile_blocks, 8) * 8 * n_m), 1, (unsigned)p.batch);
    hipLaunchKernelGGL((winobf3_conv_kernel<KW, DBG>), grid, dim3(W2_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

bool winobf3_enabled() {
    static const int on = knob("RVC_WBF_V3", 0);
    return on != 0;
}

bool winobf3_supported(int c_in, int c_out, int k, int dil) {
    return (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= W2_MAX_DIL && c_in % W2_CIC == 0 && c_out % 128 == 0;
}

// (the caller -- launch_winobf2_conv -- has checked shape, slope and size)
int launch_winobf3_conv(const Wbf2Params &p, int k, hipStream_t stream) {
#ifdef RVC_ABLATE
    if (k == 11) {
        static const int dbg = knob("RVC_W3_DBG", 0);
        switch (dbg) {
            case 1: return winobf3_launch<11, 1>(p, stream);
            case 2: return winobf3_launch<11, 2>(p, stream);
            case 4: return winobf3_launch<11, 4>(p, stream);
            case 8: return winobf3_launch<11, 8>(p, stream);
            case 16: return winobf3_launch<11, 16>(p, stream);
            case 31: return winobf3_launch<11, 31>(p, stream);
            default: break;
        }
    }
#endif
    return k == 3 ? winobf3_launch<3>(p, stream) : k == 7 ? winobf3_launch<7>(p, stream) : winobf3_launch<11>(p, stream);
}

}  // namespace rvc
#endif  // RVC_ABLATE
