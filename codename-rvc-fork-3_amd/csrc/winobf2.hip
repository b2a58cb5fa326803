// K3y -- winobf.hip's arithmetic (Winograd F(4,4) on the bf16 matrix cores, fp32 operands split exactly into bf16) with the
// work laid out the other way round: ONE TRANSFORM POINT PER WAVE.
//
// winobf.hip gives each of its 8 waves one 32 x 32 output tile and walks the 7 points inside the K loop: every matrix
// instruction needs a fresh tap fragment AND a fresh window fragment from LDS (1 KiB of LDS reads per instruction, the LDS
// as busy as the matrix pipe), the six products of a (point, group) form one dependent chain, and a step ends in a barrier
// (seven per 16-channel chunk).  Measured there (profiles/r03_winobf_ablation.txt): two thirds of a launch is NOT matrix work.
//
// Here wave w < 7 owns point w for the WHOLE block (BM output channels x 64 window columns = BM / 32 x 2 accumulator tiles):
//   * its tap fragments U_w are nobody else's: they go HBM/L2 -> registers (buffer_load_dwordx4, one 1 KiB fragment per
//     instruction, loaded one 24-instruction group ahead) and never touch LDS;
//   * its window fragments X_w are nobody else's either: the wave transforms the chunk's 64 windows for ITS point (one lane per
//     window, all 16 channels), splits them and writes them to a PRIVATE LDS area it alone reads back (shifted by g d for tap
//     group g) -- no block barrier between transform and product, only the wave's own program order;
//   * a tap fragment serves 2 column tiles and a window fragment BM / 32 row blocks: 0.375 fragments per matrix instruction
//     instead of 1, a quarter of the LDS reads per product;
//   * 8 (BM = 128) or 4 independent accumulator chains per wave: a matrix instruction never waits for its predecessor;
//   * wave 7 is the LOADER: it alone stages the raw input rows (global -> leaky ReLU -> de-interleaved LDS rows), two chunks
//     ahead of the products, so the compute waves issue nothing but tap loads, LDS traffic of their own point and matrix
//     instructions;
//   * ONE block barrier per 16-channel chunk (the raw-row hand-over), not seven.
// The price: the 7 points of an output element sit in 7 waves, so the output transform goes through LDS once per block
// (epilogue: accumulators -> LDS -> A^T -> bias / residual / running sum -> HBM), and 7 compute waves on 4 SIMDs leave one
// SIMD's matrix pipe half used (7 / 8 of the pipe at best).
//
// bf16-valued taps (BASELINE cfg 4) take the same path with fragments built from those values: a TRANSFORMED tap
// u = sum_k a_p^k w_k is a sum of up to four bf16 numbers at different exponents and needs up to 24 significand bits like
// any fp32 tap, so the three-way split stays (a one-term tap split exists only for the direct form, whose 16 / 16 multiply-adds
// x 3 products = 3.0 cost more matrix work than this form's 0.477 x 6 = 2.86).
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "winobf2.h"

namespace rvc {

// DBG (ablation build only; wrong results): 1 no input transform in the loop, 2 no matrix instructions, 4 no tap loads in the loop,
// 8 no raw-row staging after the prologue, 16 no epilogue, 32 no window-fragment reads in the loop
template <int KW, int BM, int DBG = 0>
__global__ void __launch_bounds__(W2_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
winobf2_conv_kernel(const Wbf2Params p) {
    using GM = W2Geom<KW, BM>;
    constexpr int NP = GM::NP, G = GM::G, C0 = GM::C0, MLO = GM::MLO, MHI = GM::MHI, WM = GM::WM, CP = W2_CP, CIC = W2_CIC;
    constexpr int XT = GM::XT, XTS = GM::XTS, NJ = GM::NJ, BNT = W2_BNT;
    constexpr int NPAIR = WM / 2;                         // row-block pairs (two 32-channel blocks each)
    constexpr int NSA = 3;                                // bf16 numbers per transformed tap (exact split of an fp32 value)
    constexpr int NPROD = 6;                              // products per (tap group, accumulator tile): all of order <= 2^-16

    extern __shared__ __attribute__((aligned(16))) float w2_smem[];
    unsigned char *const smem = reinterpret_cast<unsigned char *>(w2_smem);
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
    const bool loader = wave == W2_LOADER;                 // (three taps: six points, wave 6 only lends a hand with chunk 0 and the epilogue)
    // DBG & 128 (ablation build): wave 0 and the loader write s_memtime stamps to the buffer passed as `accin` (which is then NOT
    // added): [block][wave 0 | loader][32] -- tools/stamp_winobf2.py turns them into a per-phase breakdown
    unsigned long long *const stamps = (DBG & 128) ? reinterpret_cast<unsigned long long *>(const_cast<float *>(p.accin)) + ((size_t)blockIdx.x * 2 + (loader ? 1 : 0)) * 32 : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (DBG & 128) {
            if ((wave == 0 || loader) && lane == 0 && n_stamp < 32) stamps[n_stamp] = __builtin_readcyclecounter();
            ++n_stamp;
        }
    };
    stamp();

    f32x16 acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- what the epilogue needs (an earlier request by the loader wave, which idles through the last phases, was tried: 64 more live
    //      registers in that wave's path spill) --------------------------------------------------------------------------------
    const float *bias = p.bias;
    const float *res = p.res ? p.res + (int64_t)b * c_out * L : nullptr;
    const float *accin = (p.accin && !(DBG & 128)) ? p.accin + (int64_t)b * c_out * L : nullptr;
    float *y = p.y + (int64_t)b * c_out * L;
    const float out_scale = p.out_scale;
    f32x4 *const red = reinterpret_cast<f32x4 *>(smem);
    float *const yt = w2_smem;
    constexpr int YS = GM::YS;
    const bool l4 = (L & 3) == 0;
    const bool direct = d == 1 && l4;
    const int64_t t_blk0 = sb0 * 4 * d;
    const int64_t left = L - t_blk0;
    const int n_t = (int)(left < 4 * n_tiles_blk ? left : 4 * n_tiles_blk);   // valid outputs per row in this block
    // The residual values this thread will add are fetched FIRST, for every pass: one block per CU, so nothing else hides their
    // HBM latency -- it runs under the accumulators' trip through LDS, and the second pass's loads do not queue behind the first
    // pass's stores (the running sum of the last conv of a ResBlock, one launch in six, is fetched where it is added).
    // (a thread with nothing to add fetches the start of its row instead: a load behind a per-thread condition makes hipcc
    // branch around it and wait for each load on its own -- sixteen serial round trips, measured 9 500 cycles)
    f32x4 pre_r[NPAIR][8];                                // direct: [group 2][channel 4]; tile path: [row pass 8]
    auto prefetch_residual = [&]() __attribute__((always_inline)) {
    if (res) {
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            if (direct) {
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int grp = tid + gi * W2_NTH;
                    const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
                    const int col = (tile & 1) * 32 + (ln & 31);
                    const int64_t t0 = (sb0 + col) * 4;
                    const bool ok = col < n_tiles_blk && t0 < L;
                    const int64_t base = (int64_t)(m0 + pr * 64 + (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5)) * L + (ok ? t0 : 0);
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
    };
    // ---- raw-row staging (the loader wave's job; every wave lends a hand with chunk 0, see pair0_load) ---------------------
        const float *const px = p.x + (int64_t)b * c_in * L;
    const float slope = p.slope;
    const int E = (G - 1) * d;                                     // halo windows
    const int xt_used = (p.sb_per_block + G - 1 - MLO + MHI) * d;   // valid tiles + the (G - 1) d windows behind them + the -MLO d in front (+ MHI d behind)
    const int64_t t_start = (sb0 + MLO) * 4 * d;
    const int span = 4 * xt_used;                                  // staged samples per row (a multiple of 4, <= 4 XT)
    const bool edge = t_start < 0 || t_start + span > L;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)px, 0, (int)((int64_t)c_in * L * 4), W2_RSRC_FLAGS);
    const int L4 = (int)(L * 4);
    // (the row width is decided ONCE, outside everything: a runtime `wide` test around each load makes hipcc wait for every load on its own)
    auto run = [&](auto WIDE, auto EDGE) __attribute__((always_inline)) {
    constexpr bool wide = decltype(WIDE)::value, edge_c = decltype(EDGE)::value;   // edge_c: some staged samples are conv padding (first / last blocks of a row)                   // rows 16-byte aligned: lane = 4 consecutive samples
    // plan: slot k < NS of a row is one sample; wide: slots 0..3 = samples 4 lane .. 4 lane + 3 (one 16-byte load), then
    // one-dword slots for samples 256 + lane, 320 + lane, ...; narrow: slot k = sample lane + 64 k
    constexpr int NS_W = 4 + (4 * XT - 256 + 63) / 64, NS_N = NJ, NS = NS_W > NS_N ? NS_W : NS_N;
    unsigned goff[NS];    // byte offset of the slot's (first) sample in a row, clamped into the row
    int loff[NS];
    unsigned keep[NS];
    // x / m for x < 2048, m <= 20 as (x * (65536 / m + 1)) >> 16 (exact there): the plan is on the block's critical path in front
    // of its first loads, and twelve 32-bit divisions by a runtime d are ~300 instructions
    const unsigned inv4d = 65536u / (unsigned)(4 * d) + 1u, invd = 65536u / (unsigned)d + 1u;
    static_assert(4 * XT < 2048, "");
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int tl = wide ? (k < 4 ? 4 * lane + k : 256 + 64 * (k - 4) + lane) : lane + 64 * k;
        const bool have = tl < span && (wide ? k < NS_W : k < NS_N);
        const int tlc = have ? tl : 0;
        const int64_t t = t_start + tlc;
        const int64_t tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
        goff[k] = (unsigned)tc * 4u;
        const int sbl = (int)(((unsigned)tlc * inv4d) >> 16);
        const int r = tlc - sbl * 4 * d;
        const int ii = (int)(((unsigned)r * invd) >> 16);
        const int phi = r - ii * d;
        loff[k] = have ? ii * XTS + sbl * d + phi : XT + (lane & 1);   // columns >= XT of a row are never read
        keep[k] = (!edge || (t >= 0 && t < L)) ? 0xffffffffu : 0u;      // conv zero padding
    }
    if constexpr (wide) {   // the 16-byte load of slots 0..3 starts at slot 0's sample; a group of four lies wholly inside or outside the row
        const int64_t t0 = t_start + 4 * lane;
        goff[0] = (unsigned)(t0 < 0 ? 0 : (t0 > L - 4 ? L - 4 : t0)) * 4u;
    }
    // one tap group (3 taps): a phase is ~4 400 cycles of matrix work, about an HBM round trip under this kernel's own traffic --
    // with one register set (chunk c + 3 requested in phase c AFTER chunk c + 2 has been written out of it) this wave reached
    // every barrier ~2 400 cycles after the compute waves.  Two sets there: chunk c + 3 is requested FIRST THING in phase c.
    constexpr int NSET = G == 1 ? 2 : 1;
    float xr[NSET][2 * CP][NS];                                   // [set][row of the chunk][slot]
    auto load = [&](int c, auto SET) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int row = 0; row < 2 * CP; ++row) {
            const int s0 = (c * CIC + row) * L4;
            if constexpr (wide) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[0], s0, 0));
                xr[st][row][0] = v.x; xr[st][row][1] = v.y; xr[st][row][2] = v.z; xr[st][row][3] = v.w;
#pragma unroll
                for (int k = 4; k < NS_W; ++k) xr[st][row][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0, 0));
            } else {
#pragma unroll
                for (int k = 0; k < NS_N; ++k) xr[st][row][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0, 0));
            }
        }
    };
    auto write = [&](int c, auto SET) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        w2_f32x2 *const dst0 = xs + (c & 1) * XRAW;
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            w2_f32x2 *const dst = dst0 + q * 4 * XTS;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (wide ? k >= NS_W : k >= NS_N) continue;
                const w2_f32x2 v = w2_lrelu2(w2_f32x2{xr[st][2 * q][k], xr[st][2 * q + 1][k]}, slope);
                if constexpr (edge_c) dst[loff[k]] = w2_f32x2{__uint_as_float(__float_as_uint(v.x) & keep[k]), __uint_as_float(__float_as_uint(v.y) & keep[k])};
                else dst[loff[k]] = v;
            }
        }
    };
    // Chunk 0 is staged by ALL eight waves, one channel pair each: with the loader alone the block's first barrier came 10 600
    // cycles after its start (48 loads, their HBM round trip at the moment every CU starts a block, 48 LDS writes, all in one wave).
    // (Staging chunk 1 the same way, also before the first barrier, measured SLOWER: the burst at a block's start runs at the ~11 B/clk
    // a CU gets while every CU starts a block, so twice the bytes in front of the first barrier moved it from 5 900 to 9 400 cycles.)
    // (the loader puts its requests for chunks 1 and 2 BETWEEN this pair's loads and their use: loads return in order, and the first
    // barrier waits for chunk 0 only)
    float r0[NS], r1[NS];
    auto pair0_load = [&](int q) __attribute__((always_inline)) {
        const int s0 = 2 * q * L4;
        if constexpr (wide) {
            const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[0], s0, 0));
            const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[0], s0 + L4, 0));
            r0[0] = v0.x; r0[1] = v0.y; r0[2] = v0.z; r0[3] = v0.w;
            r1[0] = v1.x; r1[1] = v1.y; r1[2] = v1.z; r1[3] = v1.w;
#pragma unroll
            for (int k = 4; k < NS_W; ++k) {
                r0[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0, 0));
                r1[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0 + L4, 0));
            }
        } else {
#pragma unroll
            for (int k = 0; k < NS_N; ++k) {
                r0[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0, 0));
                r1[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)goff[k], s0 + L4, 0));
            }
        }
    };
    auto pair0_write = [&](int q) __attribute__((always_inline)) {
        w2_f32x2 *const dst = xs + q * 4 * XTS;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (wide ? k >= NS_W : k >= NS_N) continue;
            const w2_f32x2 v = w2_lrelu2(w2_f32x2{r0[k], r1[k]}, slope);
            if constexpr (edge_c) dst[loff[k]] = w2_f32x2{__uint_as_float(__float_as_uint(v.x) & keep[k]), __uint_as_float(__float_as_uint(v.y) & keep[k])};
            else dst[loff[k]] = v;
        }
    };
    if (!loader) {            // compute wave w: channel pair w of chunk 0, then on to its own prologue
        pair0_load(wave);
        pair0_write(wave);
        return;
    }
    // halo windows: item = lane + 64 r -> (window 64 + e, point hp, unit hu = channel pairs 2 hu, 2 hu + 1)
    constexpr int HR = G > 1 ? ((G - 1) * W2_MAX_DIL * 28 + 63) / 64 : 1;   // (one tap group: no halo, the arrays below stay unused)
    int h_src[HR], h_dst[HR];
    float hbt[HR][7];
    bool h_on[HR];
#pragma unroll
    for (int r = 0; r < (G > 1 ? HR : 0); ++r) {
        const int it = lane + 64 * r;
        h_on[r] = it < E * 28;
        const int itc = h_on[r] ? it : 0;
        const int e = itc / 28, rem = itc - e * 28, hp = rem >> 2, hu = rem & 3;
        h_src[r] = (2 * hu) * 4 * XTS - MLO * d + BNT + e;
        h_dst[r] = hp * 2 * GM::B_WAVE + (hu >> 1) * GM::B_PLANE + (BNT + e) * 16 + (hu & 1) * 8;
#pragma unroll
        for (int n = 0; n < 7; ++n) hbt[r][n] = W2_BT[hp][n];
    }
    auto halo = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < (G > 1 ? HR : 0); ++r) {
            if (!h_on[r]) continue;
            const w2_f32x2 *const raw = xs + (c & 1) * XRAW + h_src[r];
            unsigned w[3][2];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                w2_f32x2 q7[7];
#pragma unroll
                for (int n = 0; n < 7; ++n) {
                    const int sh = n - C0;
                    q7[n] = raw[(e2 * 4 + (sh & 3)) * XTS + (sh >> 2) * d];
                }
                // the compute waves' expression, term for term (t_xform2)
                w2_f32x2 a = q7[0] * hbt[r][0], b2 = q7[1] * hbt[r][1];
                a = __builtin_elementwise_fma(w2_f32x2{hbt[r][2], hbt[r][2]}, q7[2], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{hbt[r][3], hbt[r][3]}, q7[3], b2);
                a = __builtin_elementwise_fma(w2_f32x2{hbt[r][4], hbt[r][4]}, q7[4], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{hbt[r][5], hbt[r][5]}, q7[5], b2);
                a = __builtin_elementwise_fma(w2_f32x2{hbt[r][6], hbt[r][6]}, q7[6], a);
                w2_f32x2 v = a + b2;
#pragma unroll
                for (int level = 0; level < 3; ++level) {
                    const unsigned ww = __builtin_bit_cast(unsigned, __builtin_convertvector(v, w2_bf16x2));
                    w[level][e2] = ww;
                    if (level < 2) v = v - w2_f32x2{__uint_as_float(ww << 16), __uint_as_float(ww & 0xffff0000u)};
                }
            }
            unsigned char *o = bs_all + (c & 1) * GM::B_WAVE + h_dst[r];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<w2_u32x2 *>(o + sp * 2 * GM::B_PLANE) = w2_u32x2{w[sp][0], w[sp][1]};
        }
    };
    // (Pulling the block's residual tile into L2 from here during the last two phases -- LDS-DMA into a sink, no registers -- cut the
    // epilogue's request time from 5 100 to 3 200 cycles and cost as much in those phases, the requests queueing in front of the
    // compute waves' tap loads: 414 -> 420 us on one box, A/B.)
    const int last = n_chunks - 1;
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, NSET - 1> S1{};   // chunk k waits in set k & 1 (one set: always 0)
    pair0_load(CP - 1);
    load(1 < last ? 1 : last, S1);                    // (in flight while this wave stages its pair of chunk 0)
    if constexpr (NSET == 2) load(2 < last ? 2 : last, S0);
    pair0_write(CP - 1);
    lds_barrier();                                    // (P1) chunk 0's rows are in LDS
    stamp();
    halo(0);
    write(1, S1);
    if constexpr (NSET == 1) load(2 < last ? 2 : last, S0);
    lds_barrier();                                    // (P2) chunk 1's rows; chunk 0 is transformed
    stamp();
    // phase c: chunk c + 2 (requested a phase ago) goes to LDS buffer c & 1 -- chunk c was transformed before the last barrier --
    // and chunk c + 3 is requested: into the registers chunk c + 2 left, or (two sets) up front into the other set
    auto phase_l = [&](int c, auto SET_W, auto SET_L) __attribute__((always_inline)) {
        if (!(DBG & 8)) {
            if (NSET == 2 && c + 3 <= last) load(c + 3, SET_L);
            if (c + 2 < n_chunks) write(c + 2, SET_W);
            if (NSET == 1 && c + 3 <= last) load(c + 3, SET_L);
            if (c + 1 < n_chunks) halo(c + 1);        // chunk c + 1's halo windows, next to the compute waves' own 64
        }
        stamp();   // the loader: staged, now waiting
        lds_barrier();
    };
    for (int c = 0; c < n_chunks; c += 2) {
        phase_l(c, S0, S1);
        if (c + 1 < n_chunks) phase_l(c + 1, S1, S0);
    }
    };
    auto run_staging = [&]() __attribute__((always_inline)) {
        if ((L & 3) == 0) {
            if (edge) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
        } else {
            run(std::false_type{}, std::true_type{});
        }
    };
    if (wave >= NP) {
        // ================================ wave 7: raw input rows, and the halo windows ========================================
        // (a) stages the raw rows: chunk c + 3 is FETCHED during phase c (into registers), chunk c + 2 -- fetched a phase earlier --
        //     is written to LDS buffer c & 1 at the start of phase c.  Fetch and write in the same phase made this wave the slowest
        //     of the block (two serial HBM round trips; and a wave holds at most 63 loads in flight, so 160 one-dword loads per
        //     chunk were three round trips whatever the order): rows are fetched 16 bytes per lane where L % 4 == 0.
        // (b) transforms the (G - 1) d windows BEHIND the block's 64: column `col` of tap group g reads window col + g d, so with
        //     windows 64 .. 63 + (G - 1) d present all 64 accumulator columns are valid outputs -- 64 / d super-blocks per block
        //     instead of (64 - (G - 1) d) / d.  At 383 760 columns and d = 1 that is 1500 blocks = 5.9 rounds of the 256 CUs instead
        //     of 1548 = 6.05, i.e. six rounds instead of seven.  This wave has the time: one lane per (window, point, two channel
        //     pairs), <= 280 items = five rounds of what a compute wave does four of.
        // This wave shares its SIMD with compute wave 3 and, being the youngest, loses every issue arbitration to it: measured, its
        // ~650 instructions per phase took longer than the compute waves' whole phase.  Its work is small and on the block's
        // critical path (the barrier), so it gets the SIMD's issue slots first.
        if (loader) __builtin_amdgcn_s_setprio(2);
        run_staging();
        if (!loader) {   // three taps, wave 6: no point of its own -- its channel pair of chunk 0 is staged, now it keeps the barriers company
            lds_barrier();                                // (P1)
            lds_barrier();                                // (P2)
            for (int c = 0; c < n_chunks; ++c) lds_barrier();
        }
    } else {
        // ================================ waves 0..6: point `wave` of every output tile =====================================
        const int pt = wave;
        unsigned char *const bs = bs_all + pt * 2 * GM::B_WAVE;
        // input transform coefficients of this point (wave-uniform: scalar registers)
        float bt[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) bt[n] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, KW == 3 ? W2_BT3[pt][n] : W2_BT[pt][n])));
        // tap fragments: [c_out / BM][point][chunk][group][row block][split][lane][8 bf16], 1 KiB each -- for one (block, point)
        // the groups (chunk, tap group, row-block pair) follow each other in the order the wave consumes them, 6 KiB apiece: one
        // running scalar offset, the six pieces at immediate offsets
        const __amdgpu_buffer_rsrc_t urs =
            __builtin_amdgcn_make_buffer_rsrc((void *)p.u, 0, (int)((int64_t)c_in * c_out * NP * G * NSA * 2), W2_RSRC_FLAGS);
        constexpr int A_GROUP = 2 * NSA * 1024;
        const int n_groups = n_chunks * G * NPAIR;
        const int a_base = (mblk * NP + pt) * n_groups * A_GROUP;
        const int a_last = a_base + (n_groups - 1) * A_GROUP;
        w2_bf16x8 fa[2][2][NSA];                           // [buffer][row block of the pair][split]
        auto load_a = [&](int buf, int Q) __attribute__((always_inline)) {   // group Q of this wave's stream (clamped: the tail re-reads the last one)
            int soff = a_base + Q * A_GROUP;
            soff = soff < a_last ? soff : a_last;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int sp = 0; sp < NSA; ++sp)
                    fa[buf][rb][sp] = __builtin_bit_cast(w2_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, 16 * lane + (rb * NSA + sp) * 1024, soff, 0));
        };
        // window fragments of one tap group: [column tile][split]; single-buffered -- the products of a group are ordered so that
        // split 2 dies first, then split 1, and each is re-read for the next group while the current one finishes
        w2_bf16x8 fb[2][3];
        const int b_lane = half * GM::B_PLANE + l31 * 16;
        auto read_b = [&](int c, int g, int sp) __attribute__((always_inline)) {
            const unsigned char *bb = bs + (c & 1) * GM::B_WAVE + sp * 2 * GM::B_PLANE + b_lane + g * d * 16;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) fb[cb][sp] = __builtin_bit_cast(w2_bf16x8, *reinterpret_cast<const w2_u32x4 *>(bb + cb * 32 * 16));
        };
        // ---- input transform of chunk c for this point: lane = window, unit u = channel pairs 2u, 2u + 1 -------------------
        const int t_src = -MLO * d + lane;                 // raw tile of the window's own super-block
        w2_f32x2 tq[2][NP];
        w2_f32x2 tv[2];
        unsigned tw[3][2];
        auto t_read = [&](int c, int u) __attribute__((always_inline)) {
            const w2_f32x2 *const raw = xs + (c & 1) * XRAW + t_src;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int n = 0; n < NP; ++n) {
                    const int sh = n - C0;
                    tq[e][n] = raw[((2 * u + e) * 4 + (sh & 3)) * XTS + (sh >> 2) * d];
                }
        };
        // both pairs of a unit at once, each as two independent half sums (four dependency chains instead of one of seven)
        auto t_xform2 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                w2_f32x2 a = tq[e][0] * bt[0], b2 = tq[e][1] * bt[1];
                a = __builtin_elementwise_fma(w2_f32x2{bt[2], bt[2]}, tq[e][2], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{bt[3], bt[3]}, tq[e][3], b2);
                a = __builtin_elementwise_fma(w2_f32x2{bt[4], bt[4]}, tq[e][4], a);
                b2 = __builtin_elementwise_fma(w2_f32x2{bt[5], bt[5]}, tq[e][5], b2);
                if constexpr (NP == 7) a = __builtin_elementwise_fma(w2_f32x2{bt[6], bt[6]}, tq[e][6], a);
                tv[e] = a + b2;
            }
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
            t_read(c, u);
            t_xform2();
#pragma unroll
            for (int e = 0; e < 2; ++e) { t_split(e, 0); t_split(e, 1); t_split(e, 2); }
            t_write(c, u);
        };

        // ---- prologue ------------------------------------------------------------------------------------------------------
        load_a(0, 0);
        run_staging();                                    // this wave's channel pair of chunk 0 (the loader takes the eighth)
        lds_barrier();                                    // (P1)
        stamp();
#pragma unroll
        for (int u = 0; u < 4; ++u) transform_unit(0, u);
        load_a(1, 1);
        read_b(0, 0, 2); read_b(0, 0, 1); read_b(0, 0, 0);
        lds_barrier();                                    // (P2)
        stamp();

        // ---- main loop: one phase per chunk ---------------------------------------------------------------------------------
        // group q = (tap group g, row-block pair pr), q = g * NPAIR + pr: 24 matrix instructions on four independent accumulators
        // (product i, row block rb, column tile cb).  Its tap fragments sit in fa[(q + parity) & 1] (parity: NQ may be odd, so the
        // buffer of "group q" alternates from chunk to chunk); those of the next group are in flight into the other buffer, and when
        // the group has issued its last product the loads of group q + 2 are issued into its own buffer.
        // Behind matrix instruction k of a group come, pinned by sched_barrier:
        //   * the NEXT chunk's input transform, in four units of two channel pairs, spread over the first NQ - 1 groups (the last
        //     group of a chunk re-reads window fragments of the next chunk, so every unit must have been written before it);
        //   * the window fragments of the next tap group, split by split, as soon as the last product on a split has been issued;
        //   * the tap loads of group q + 2.
        constexpr int NQ = G * NPAIR;
        static_assert(NQ >= 2, "the next chunk's transform needs a group in front of the one that re-reads its fragments");
        constexpr int ia6[6] = {0, 1, 0, 2, 1, 0}, ib6[6] = {2, 1, 1, 0, 0, 0};
        auto phase = [&](int c, auto PAR) __attribute__((always_inline)) {
            constexpr int par = decltype(PAR)::value;
            // the last phase re-transforms its own chunk (same bits, never needed): no branch in the body.  (A third instantiation of
            // the body without that work was tried: with three copies of the loop the register allocator spills 878 registers.)
            const int cn = c + 1 < n_chunks ? c + 1 : c;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int g = q / NPAIR, pr = q % NPAIR;
                const int bufq = (q + par) & 1;
                // the window fragments the NEXT group needs: a new tap group's (g + 1, or group 0 of the next chunk) when pr is the last pair
                const bool b_turn = pr == NPAIR - 1;
                const int gb = (g + 1) % G, cb_c = g + 1 < G ? c : c + 1;
                // (A priority that FALLS as a wave advances through its phase -- so that of a SIMD's two compute waves the one behind
                // goes first, instead of the older one finishing early and leaving the younger alone with the pipe -- measured 3-5 %
                // SLOWER, 388 -> 403 us at C = 128 K = 11.)
#pragma unroll
                for (int i = 0; i < NPROD; ++i) {
                    const int ia = ia6[i], ib = ib6[i];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            if (!(DBG & 2)) acc[pr * 2 + rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bufq][rb][ia], fb[cb][ib], acc[pr * 2 + rb][cb], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            const int k = (i * 2 + rb) * 2 + cb;            // 0 .. 23: fillers behind this matrix instruction
#pragma unroll
                            for (int u = 0; u < 4; ++u) {                   // transform unit u lives in group u (NQ - 1) / 4, six slots each
                                if ((DBG & 1) || u * (NQ - 1) / 4 != q) continue;
                                int first = u;                              // its rank among the units of this group
                                while (first > 0 && (first - 1) * (NQ - 1) / 4 == q) --first;
                                const int k0 = (u - first) * 6;
                                if (k == k0) t_read(cn, u);
                                if (k == k0 + 1) t_xform2();
                                if (k == k0 + 2) { t_split(0, 0); t_split(1, 0); }
                                if (k == k0 + 3) { t_split(0, 1); t_split(1, 1); }
                                if (k == k0 + 4) { t_split(0, 2); t_split(1, 2); }
                                if (k == k0 + 5) t_write(cn, u);
                            }
                            if (b_turn && !(DBG & 32)) {
                                if (k == 3) read_b(cb_c, gb, 2);            // split 2: product 0 only
                                if (k == 11) read_b(cb_c, gb, 1);           // split 1: products 1, 2
                                if (k == 23) read_b(cb_c, gb, 0);           // split 0: products 3..5
                            }
                            if (k == 4 * NPROD - 1 && !(DBG & 4)) load_a(bufq, c * NQ + q + 2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
            stamp();   // a compute wave: products and transform done, now waiting
            lds_barrier();                                // the raw rows of chunk c + 2 are in LDS; B of chunk c + 1 is this wave's own
        };
        for (int c = 0; c < n_chunks; c += 2) {
            phase(c, std::integral_constant<int, 0>{});
            if (c + 1 < n_chunks) phase(c + 1, std::integral_constant<int, NQ & 1>{});
        }
    }

    // ================================ epilogue: output transform across the 7 waves ========================================
    // Per row-block pair (64 channels x 64 columns): the compute waves park their four accumulator tiles in LDS
    // red[point][tile = rb * 2 + cb][r >> 2][lane][r & 3]; every thread then takes two (tile, r >> 2, lane) groups, reads the 7
    // points' float4 (4 consecutive channels of one column), applies A^T diag(1 / N_j) exactly as winobf.hip does, and either
    // stores the 4 outputs of a column straight to HBM (d = 1, L % 4 == 0: 16 contiguous bytes) or parks them in the
    // transposed tile yt[channel][t - t_blk0] for whole-row stores.
    if (DBG & 16) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 12345.678f) p.y[tid] = s;
        return;
    }
    if constexpr (DBG & 128) { n_stamp = 20; stamp(); }   // epilogue stamps from slot 20 on
    // the bias of this thread's channel quads, both passes: requested BEFORE the residual (a wave's loads return in order; fetched
    // inside the pass it sat behind the sixteen residual loads and the first pass waited for all of them)
    f32x4 bq_all[NPAIR][2];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int grp = tid + gi * W2_NTH;
            const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
            const int row0 = (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5);
            bq_all[pr][gi] = bias ? *reinterpret_cast<const f32x4 *>(bias + m0 + pr * 64 + row0) : f32x4{0.f, 0.f, 0.f, 0.f};   // row0 is a multiple of 4
        }
    prefetch_residual();
    // Each pass leaves its results in registers; residual, running sum and the stores come after the LAST pass, so the residual's
    // HBM latency runs under both passes' LDS traffic (with add + store inside the pass the first pass waited ~8 000 cycles for it).
    f32x4 outv[NPAIR][8];
    stamp();   // residual loads issued
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        if (pr > 0) lds_barrier();                        // the previous pass's tile has been read back
        if (wave < NP) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        red[((wave * 4 + rb * 2 + cb) * 4 + rq) * 64 + lane] =
                            f32x4{acc[pr * 2 + rb][cb][4 * rq], acc[pr * 2 + rb][cb][4 * rq + 1], acc[pr * 2 + rb][cb][4 * rq + 2], acc[pr * 2 + rb][cb][4 * rq + 3]};
        }
        stamp();   // accumulators written
        lds_barrier();
        stamp();   // accumulators parked
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
            const int row0 = rb * 32 + 8 * rq + 4 * (ln >> 5);           // + comp: channel inside the pair's 64
            g_row[gi] = row0;
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
                    const float t0 = v[0][e] * -2.f, t1 = v[1][e] * (-2.f / 3.f), t2 = v[2][e] * (-2.f / 9.f), t3 = v[3][e] * (16.f / 9.f),
                                t4 = v[4][e] * (16.f / 15.f), t5 = v[5][e] * (2.f / 45.f), t6 = v[6][e];
                    const float s12 = t1 + t2, m12 = t1 - t2, s34 = t3 + t4, m34 = t3 - t4;
                    o[gi][e].x = (t0 + s12) + (s34 + t5) + bv;
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
            stamp();   // outputs in registers
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
    stamp();   // every pass reduced
    // ---- residual, running sum, scale, stores -------------------------------------------------------------------------------
    if (direct) {
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int grp = tid + gi * W2_NTH;
                const int ln = grp & 63, rq = (grp >> 6) & 3, tile = grp >> 8;
                const int col = (tile & 1) * 32 + (ln & 31);
                const int64_t t0 = (sb0 + col) * 4;
                if (col < n_tiles_blk && t0 < L) {
                    const int64_t base = (int64_t)(m0 + pr * 64 + (tile >> 1) * 32 + 8 * rq + 4 * (ln >> 5)) * L + t0;
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
    stamp();   // stores issued
}

template <int KW, int BM, int DBG = 0>
static int winobf2_launch(Wbf2Params p, hipStream_t stream) {
    using GM = W2Geom<KW, BM>;
    p.sb_per_block = W2_BNT / p.dil;   // every accumulator column a valid tile: the (G - 1) d windows behind the 64 are the loader wave's
    const int64_t n_sb = ceil_div(p.L, (int64_t)4 * p.dil);
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute((const void *)winobf2_conv_kernel<KW, BM, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    });
    if (err != hipSuccess) return fail("winobf2 conv: cannot reserve %d bytes of LDS: %s", GM::LDS_BYTES, hipGetErrorString(err));
    p.n_tile_blocks = (int)ceil_div(n_sb, p.sb_per_block);
    const int n_m = p.c_out / BM;
    dim3 grid((unsigned)(ceil_div(p.n_tile_blocks, 8) * 8 * n_m), 1, (unsigned)p.batch);
    hipLaunchKernelGGL((winobf2_conv_kernel<KW, BM, DBG>), grid, dim3(W2_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

// This form runs 128-row blocks; layers whose c_out is not a multiple of 128 (the 64-channel stage: four 16-channel chunks are
// too short a K loop to pay for this kernel's prologue and its output transform through LDS -- measured 1.17-1.28x slower than
// winobf.hip's 64 x 128 blocks, profiles/r04_convbf_shapes_v2_first.txt) stay on winobf.hip.
static bool w2_rows64() {   // ablation build: 64-row blocks for c_out % 128 != 0 (A/B against winobf.hip)
    static const int on = knob("RVC_WBF_V2_64", 0);
    return on != 0;
}
bool winobf2_supported(int c_in, int c_out, int k, int dil) {
    return (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= W2_MAX_DIL && c_in % W2_CIC == 0 && (c_out % 128 == 0 || (w2_rows64() && c_out % 64 == 0 && k != 3));
}

bool winobf2_fits(int c_in, int c_out, int64_t L) {
    return (int64_t)c_in * L < ((int64_t)1 << 29) && (int64_t)c_in * c_out * 7 * 3 * 6 < ((int64_t)1 << 31);
}

int launch_winobf2_conv(const float *x, const void *u, const float *bias, const float *res, const float *accin, float *y, int batch,
                        int c_in, int c_out, int64_t L, int k, int dil, float slope, float out_scale, hipStream_t stream) {
    if (!winobf2_supported(c_in, c_out, k, dil)) return fail("winobf2 conv: unsupported shape (%d -> %d channels, %d taps, dilation %d)", c_in, c_out, k, dil);
    if (!(slope >= 0.f && slope <= 1.f)) return fail("winobf2 conv: leaky slope %g outside [0, 1]", (double)slope);
    if (!winobf2_fits(c_in, c_out, L)) return fail("winobf2 conv: %d x %lld samples exceed the 2 GiB buffer addressing", c_in, (long long)L);
    if (L <= 0 || batch <= 0) return 0;
    Wbf2Params p;
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.accin = accin; p.y = y;
    p.c_in = c_in; p.c_out = c_out; p.L = L; p.dil = dil; p.slope = slope; p.out_scale = out_scale; p.batch = batch;
#ifdef RVC_ABLATE
    if (k == 11) {   // where does the time go (tools/ablate_winobf2.sh; wrong results)
        static const int dbg = knob("RVC_W2_DBG", 0);
        switch (dbg) {
            case 1: return winobf2_launch<11, 128, 1>(p, stream);
            case 2: return winobf2_launch<11, 128, 2>(p, stream);
            case 3: return winobf2_launch<11, 128, 3>(p, stream);
            case 4: return winobf2_launch<11, 128, 4>(p, stream);
            case 8: return winobf2_launch<11, 128, 8>(p, stream);
            case 16: return winobf2_launch<11, 128, 16>(p, stream);
            case 32: return winobf2_launch<11, 128, 32>(p, stream);
            case 37: return winobf2_launch<11, 128, 37>(p, stream);
            case 45: return winobf2_launch<11, 128, 45>(p, stream);
            case 61: return winobf2_launch<11, 128, 61>(p, stream);
            case 63: return winobf2_launch<11, 128, 63>(p, stream);
            case 64: return winobf2_launch<11, 128, 64>(p, stream);
            case 128: return winobf2_launch<11, 128, 128>(p, stream);
            default: break;
        }
    }
    if (k == 3 && c_out % 128 == 0 && knob("RVC_W2_DBG", 0) == 128) return winobf2_launch<3, 128, 128>(p, stream);
    if (c_out % 128) return k == 7 ? winobf2_launch<7, 64>(p, stream) : winobf2_launch<11, 64>(p, stream);
#endif
    return k == 3 ? winobf2_launch<3, 128>(p, stream) : k == 7 ? winobf2_launch<7, 128>(p, stream) : winobf2_launch<11, 128>(p, stream);
}

}  // namespace rvc
