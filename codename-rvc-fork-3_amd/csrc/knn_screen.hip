// K1 (batch regime) -- exact top-8 retrieval with an fp16 matrix-core screening pass.
//
// The search of pipeline.py:497-499 at Q = 1599 queries is a 245.6 GFLOP GEMM (N = 100 k) to 4.9 TFLOP (N = 2 M) with a
// selection epilogue.  In exact fp32 it is bound by the fp32 matrix rate (157 TF: 1.6-2.5 ms at 100 k rows, 50 ms at
// 2 M).  The fp16 matrix cores are 16x faster, and a screening pass does not need exact scores -- it needs a PROVABLE
// superset of the true neighbours, which exact arithmetic then ranks:
//
//   index build (once)   x~ = fp16(x) [N][dim] beside the fp32 rows; ||x||^2; max ||x||, max ||x - x~|| over the index
//   per search
//     1 queries          q~ = fp16(q), ||q||^2, ||q - q~||, ||q~||                                    (knn_to_half_kernel)
//     2 sample pass      s~(q, n) = ||x_n||^2 - 2 q~.x~_n on a strided sample of row tiles (fp16 MFMA, fp32 accumulate);
//                        every lane keeps the minimum over its own rows -> one minimum per (query, disjoint row subset)
//     3 bound            S8 = 8th smallest of those minima: 8 distinct rows score <= S8, so the true 8th-best score is
//                        <= B = S8 + eps;  accept threshold thr = B + eps + delta                      (knn_select_kernel)
//     4 main pass        the same GEMM over ALL rows; a row is appended to its query's candidate list iff s~ < thr
//     5 exact ranking    every candidate is re-scored as sum_k (q_k - x_k)^2 in fp32 in a fixed order (what faiss' flat
//                        scanner computes) and the best 8 by (distance, id) are the result          (knn_finalize_kernel)
//
// eps bounds |s~ - s| rigorously: |q.x - q~.x~| <= ||q - q~|| ||x|| + ||q~|| ||x - x~|| (Cauchy-Schwarz, with the residual
// norms MEASURED, not assumed), plus fp32 accumulation slack.  Any row with true score <= B has s~ < thr, and the true
// top 8 all score <= B, so the candidate list contains them whatever the data; delta covers the rounding of the exact
// fp32 re-scoring itself.  The list has a fixed capacity per query; a query that overflows it (thresholds are only as
// tight as the sample is representative) is answered by an exact scan of the whole index inside the same launch --
// slower, never wrong.  Expected survivors per query are ~8 N / (sample rows), a few dozen.
//
// The final (d2, ids) depend only on the exact re-scoring, so every regime of rvc_knn_search (this one, the fp32 GEMM
// of knn.hip, the streaming kernel for <= 64 queries) returns identical results for identical inputs.
//
// GEMM decomposition (both passes): block = 8 waves = 256 index rows x 256 queries, wave = 128 x 64 (4 x 2 MFMA tiles of
// v_mfma_f32_32x32x16_f16), K walked in 64-element chunks staged HBM/L2 -> registers -> LDS (rows padded to 144 B:
// conflict-free ds_read_b128), double-buffered, one barrier per chunk.  A lane takes the 64 contiguous bytes
// [64 h, 64 h + 64) of its row per chunk (h = lane / 32) as four MFMA operands -- a dot product may walk k in any order
// as long as both operands agree.  The 7 query tiles of one row stripe are placed on ONE XCD so the stripe crosses the
// fabric once and is re-read from that XCD's L2.
#include <limits.h>
#include <stdlib.h>

#include <mutex>

#include "knn_common.h"

namespace rvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int KS_BN = 256;        // queries per block (4 wave columns x 64)
// Tile shape, template parameters of the GEMM kernel:
//   WM  wave rows per block: block = 128 WM index rows x 256 queries, 4 WM waves, each wave 128 rows x 64 queries
//   BK  halves of K per staged chunk (64: 128-byte rows, 4 MFMA steps per chunk; 32: 64-byte rows, 2 steps)
// LDS rows are padded by 16 B (conflict-free ds_read_b128 for both row lengths); two chunk buffers.
//   WM 2, BK 64: 147 KB -> one 8-wave block per CU: every wave of the CU meets at the same barrier each chunk
//   WM 1, BK 32:  61 KB -> two 4-wave blocks per CU with independent barriers: one block's staging overlaps the other's MFMAs
template <int WM, int BK>
struct ScreenTile {
    static constexpr int BM = 128 * WM;
    static constexpr int THREADS = 256 * WM;
    static constexpr int ROWB = 2 * BK + 16;
    static constexpr int KSTEPS = BK / 16;
    static constexpr int PIECES = BK / 8;                          // 16-byte pieces per staged row
    static constexpr int A_LOADS = BM * PIECES / THREADS;          // per thread per chunk
    static constexpr int B_LOADS = KS_BN * PIECES / THREADS;
    static constexpr int ROWS_PER_PASS = THREADS / PIECES;         // rows one staging pass of the block covers
    static constexpr int SLOTS_PER_TILE = 2 * WM;                  // (wave row) x (lane half): disjoint row subsets of a tile
    static constexpr size_t LDS_BYTES = 2 * (size_t)(BM + KS_BN) * ROWB + 2 * BM * sizeof(float);
};

// ---- fp32 -> fp16 rows with the measured rounding residual --------------------------------------------------------
// one wave per row.  norms (optional): sum x^2 in the order knn_norms_kernel uses.  rowstat (optional): {sum x^2,
// sum (x - x~)^2, sum x~^2, max |x|}.  stats (optional, device words, float bits of non-negative values so that
// atomicMax on the integer view orders them): [0] max sum x^2, [1] max residual^2, [2] max |x|.
__global__ void __launch_bounds__(256)
knn_to_half_kernel(const float *__restrict__ x, int64_t n_rows, int dim, _Float16 *__restrict__ xh, float *__restrict__ norms,
                   f32x4 *__restrict__ rowstat, unsigned *__restrict__ stats, float *__restrict__ mins = nullptr, int n_slots = 0,
                   int *__restrict__ cnt = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    // a search's per-query scratch is reset here (the query conversion is its first launch): the sample pass's minima to
    // +inf, the candidate counters to 0 -- two fill launches less per search
    if (mins)
        for (int i = lane; i < n_slots; i += 64) mins[row * n_slots + i] = INFINITY;
    if (cnt && lane == 0) {
        cnt[row] = 0;
        if (row == 0) cnt[n_rows] = 0;       // the exact-scan counter behind the per-query ones
    }
    const f32x4 *p = reinterpret_cast<const f32x4 *>(x + row * dim);
    _Float16 *o = xh + row * dim;
    float s = 0.f, r2 = 0.f, h2 = 0.f, mx = 0.f;
    for (int i = lane; i < dim / 4; i += 64) {
        const f32x4 v = p[i];
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = (_Float16)v[e];              // round to nearest even
            const float back = (float)hv[e];
            const float d = v[e] - back;
            s = fmaf(v[e], v[e], s);
            r2 = fmaf(d, d, r2);
            h2 = fmaf(back, back, h2);
            mx = fmaxf(mx, fabsf(v[e]));
        }
        *reinterpret_cast<f16x4 *>(o + 4 * i) = hv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        r2 += __shfl_xor(r2, off);
        h2 += __shfl_xor(h2, off);
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) {
        if (norms) norms[row] = s;
        if (rowstat) rowstat[row] = f32x4{s, r2, h2, mx};
        if (stats) {
            atomicMax(&stats[0], __float_as_uint(s));
            atomicMax(&stats[1], __float_as_uint(r2));
            atomicMax(&stats[2], __float_as_uint(mx));
        }
    }
}

struct ScreenParams {
    const _Float16 *xh;      // [n_rows][dim]
    const float *xn;         // [n_rows]
    int64_t n_rows;
    int dim;
    const _Float16 *qh;      // [n_queries][dim]
    int64_t n_queries;
    int n_qtiles, n_stripes;
    int tiles_per_block;     // row tiles a block walks
    int tile_step;           // tile index = (stripe * tiles_per_block + i) * tile_step
    int n_tiles;             // row tiles in the index
    // APPEND mode
    const float *thr;        // [n_queries]
    int *cand_id;            // [n_queries][cap]
    int *cand_cnt;           // [n_queries]
    int cap;
    // MIN mode
    float *mins;             // [n_queries][n_slots]
    int n_slots;
};

// ---- selection epilogue of one finished tile: this lane's 64 rows x 2 queries --------------------------------------------
// APPEND: a row goes to its query's candidate list iff s~ = ||x||^2 - 2 q~.x~ is not provably above the threshold ("not >=":
// a NaN score -- inf in the fp16 data -- must be kept, too).  Survivors are a few dozen per query out of the whole index, so
// the test is branch-free over groups of 4 rows x 2 queries (fma + compare + or per element) with ONE wave-uniform branch
// per group into the append code.  MIN: running minimum per query.  Zeroes the accumulators for the next tile.
template <bool APPEND>
__device__ __forceinline__ void screen_select(f32x16 (&acc)[4][2], const float *xn_t, int row_base, int row_limit, const float (&thr)[2],
                                              float (&mn)[2], int64_t q_lane, int64_t n_queries, int64_t n_rows, int *cand_cnt,
                                              int *cand_id, int cap) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float v[4][2];
            bool hit = false;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ro = m * 32 + r4 + 8 * rg;
                const float xnv = row_base + ro < row_limit ? xn_t[ro] : INFINITY;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    v[r4][n] = fmaf(-2.f, acc[m][n][4 * rg + r4], xnv);
                    if (APPEND) hit |= !(v[r4][n] >= thr[n]);
                    else mn[n] = fminf(mn[n], v[r4][n]);
                    acc[m][n][4 * rg + r4] = 0.f;
                }
            }
            if (APPEND && __any(hit)) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        if (!(v[r4][n] >= thr[n])) {
                            const int64_t q = q_lane + 32 * n;
                            const int row = row_base + m * 32 + r4 + 8 * rg;
                            if (q < n_queries && row < n_rows) {
                                const int pos = atomicAdd(&cand_cnt[q], 1);
                                if (pos < cap) cand_id[q * cap + pos] = row;
                            }
                        }
            }
        }
}

// DBG (ablations, wrong results; RVC_KNN_DBG, tools/ablate_knn.py, tools/clock_knn.sh): 2 no staging after the prologue,
// 4 no barrier, 8 no matrix instructions, 16 fragments read from LDS in the first chunk only, 32 threshold -inf (nothing is
// appended, so that 2 / 16 do not flood the candidate lists; always set together with the others)
template <bool APPEND, int WM, int BK, int DBG = 0>
__global__ void __launch_bounds__(256 * WM) __attribute__((amdgpu_waves_per_eu(2, 2)))
knn_screen_kernel(const ScreenParams p) {
    using TL = ScreenTile<WM, BK>;
    constexpr int BM = TL::BM, ROWB = TL::ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    unsigned char *As = ks_smem;                                   // [2][BM][ROWB]
    unsigned char *Bs = ks_smem + 2 * BM * ROWB;                   // [2][KS_BN][ROWB]
    float *xn_s = reinterpret_cast<float *>(ks_smem + 2 * (BM + KS_BN) * ROWB);   // [2][BM]

    // XCD-aware placement: consecutive block ids go round the 8 XCDs; the query tiles of one stripe share an XCD
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int j = b >> 3;
    const int stripe = (j / p.n_qtiles) * 8 + xcd;
    const int qtile = j % p.n_qtiles;
    if (stripe >= p.n_stripes) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, l31 = lane & 31;
    const int dim = p.dim;
    const int n_kc = dim / BK;
    const int64_t q0 = (int64_t)qtile * KS_BN;
    const int first = stripe * p.tiles_per_block;
    int n_my = p.tiles_per_block;
    {
        const int last_tile = (p.n_tiles - 1) / p.tile_step;       // largest valid multiple index
        if (first > last_tile) return;
        if (first + n_my - 1 > last_tile) n_my = last_tile - first + 1;
    }
    const int n_chunks = n_my * n_kc;

    f16x8 ar[TL::A_LOADS], br[TL::B_LOADS];   // staging registers: 16-byte pieces of the row tile and of the query tile
    const int srow = tid / TL::PIECES, sc = tid % TL::PIECES;      // piece i: row srow + ROWS_PER_PASS i, 16-byte column sc
    const _Float16 *const xh = p.xh;
    const _Float16 *const qh = p.qh;
    const int64_t n_rows = p.n_rows, n_queries = p.n_queries;

    auto tile_row0 = [&](int t) __attribute__((always_inline)) { return (int64_t)(first + t) * p.tile_step * BM; };
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int t = c / n_kc, kc = c - t * n_kc;
        const int64_t r0 = tile_row0(t);
#pragma unroll
        for (int i = 0; i < TL::A_LOADS; ++i) {
            int64_t r = r0 + srow + TL::ROWS_PER_PASS * i;
            r = r < n_rows ? r : n_rows - 1;                         // clamped: the row's ||x||^2 is +inf in xn_s
            ar[i] = *reinterpret_cast<const f16x8 *>(xh + r * dim + kc * BK + sc * 8);
        }
#pragma unroll
        for (int i = 0; i < TL::B_LOADS; ++i) {
            int64_t q = q0 + srow + TL::ROWS_PER_PASS * i;
            q = q < n_queries ? q : n_queries - 1;
            br[i] = *reinterpret_cast<const f16x8 *>(qh + q * dim + kc * BK + sc * 8);
        }
    };
    auto store_chunk = [&](int buf, int c) __attribute__((always_inline)) {
        const int t = c / n_kc, kc = c - t * n_kc;
#pragma unroll
        for (int i = 0; i < TL::A_LOADS; ++i)
            *reinterpret_cast<f16x8 *>(As + ((size_t)buf * BM + srow + TL::ROWS_PER_PASS * i) * ROWB + sc * 16) = ar[i];
#pragma unroll
        for (int i = 0; i < TL::B_LOADS; ++i)
            *reinterpret_cast<f16x8 *>(Bs + ((size_t)buf * KS_BN + srow + TL::ROWS_PER_PASS * i) * ROWB + sc * 16) = br[i];
        if (kc == 0 && tid < BM) {
            const int64_t r = tile_row0(t) + tid;
            xn_s[(t & 1) * BM + tid] = r < n_rows ? p.xn[r] : INFINITY;
        }
    };

    // per-lane query state: column tile n holds query q0 + 64 wn + 32 n + l31
    float thr[2], mn[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + 64 * wn + 32 * n + l31;
        mn[n] = INFINITY;
        thr[n] = -INFINITY;
        if (APPEND && q < n_queries && !(DBG & 32)) thr[n] = p.thr[q];
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    f16x8 a[2][4], bq[2][2];
    load_chunk(0);
    store_chunk(0, 0);
    if (n_chunks > 1) load_chunk(1);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        // lane (row i, half h) owns bytes [BK h, BK h + BK) of its row's chunk: KSTEPS operands of 8 halves
        const unsigned char *ab = As + ((size_t)buf * BM + 128 * wm + l31) * ROWB + BK * h;
        const unsigned char *bb = Bs + ((size_t)buf * KS_BN + 64 * wn + l31) * ROWB + BK * h;
        auto frag = [&](int t, f16x8 (&av)[4], f16x8 (&bv)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f16x8 *>(ab + (size_t)m * 32 * ROWB + 16 * t);
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[n] = *reinterpret_cast<const f16x8 *>(bb + (size_t)n * 32 * ROWB + 16 * t);
        };
        if (!((DBG & 16) && c > 0)) frag(0, a[0], bq[0]);
#pragma unroll
        for (int t = 0; t < TL::KSTEPS; ++t) {
            if (t + 1 < TL::KSTEPS && !((DBG & 16) && c > 0)) frag(t + 1, a[(t + 1) & 1], bq[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (!(DBG & 8)) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 1][m], bq[t & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int t_cur = c / n_kc;
        if (c - t_cur * n_kc == n_kc - 1) {
            // the tile is complete: s~ = ||x||^2 - 2 q~.x~ for this lane's 64 rows x 2 queries (rows past the end: +inf in xn_s)
            const float *xn_t = xn_s + (t_cur & 1) * BM + 128 * wm + 4 * h;
            const int row_base = (int)(tile_row0(t_cur)) + 128 * wm + 4 * h;
            screen_select<APPEND>(acc, xn_t, row_base, INT_MAX, thr, mn, q0 + 64 * wn + l31, n_queries, n_rows, p.cand_cnt, p.cand_id, p.cap);
        }
        if (c + 1 < n_chunks) {
            if (!(DBG & 2)) {
                store_chunk(buf ^ 1, c + 1);
                if (c + 2 < n_chunks) load_chunk(c + 2);
            }
            if (!(DBG & 4)) __syncthreads();
        }
    }
    if (!APPEND) {
        // one minimum per (query, row subset of the tile): the subsets of different slots are disjoint
        const int slot = first * TL::SLOTS_PER_TILE + wm * 2 + h;   // MIN passes run one tile per block: first == stripe
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t q = q0 + 64 * wn + 32 * n + l31;
            if (q < n_queries) p.mins[q * p.n_slots + slot] = mn[n];
        }
    }
}

// ---- the same GEMM with LDS-DMA staging (global_load_lds, 16 B per lane): WM = 2, BK = 64 -----------------------------
// Register staging costs the block a store phase per chunk in which all 8 waves push 64 ds_write_b128 through the LDS write
// path while the matrix pipe idles (PMC: 36 % of wave time parked in s_waitcnt / barrier), and 32 staging VGPRs.  Here the
// chunk goes HBM/L2 -> LDS directly.  An LDS-DMA instruction writes lane l at base + 16 l, so padded rows are impossible;
// the conflict-free image is built from 8-row groups instead: group g (8 rows x 128 B) at g * 1152 B (128 B of padding per
// group) with the 16-byte pieces of row r XOR-permuted by (r & 7) -- the permutation is applied to the per-lane SOURCE
// address, the LDS side stays linear.  ds_read_b128 of a fragment (16 rows per lane group, one piece each) then hits 16
// distinct 16-byte bank slots (checked exhaustively for every lane group, half and k-step).  Two buffers: the DMA of chunk
// c + 1 is issued at the top of iteration c, runs under its 32 MFMAs, and is retired by s_waitcnt vmcnt(0) + barrier.
constexpr int KG_GROUP = 1152;                      // bytes per 8-row group
constexpr int KG_BUF = 32 * KG_GROUP;               // one operand buffer: 256 rows

template <bool APPEND>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
knn_screen_glds_kernel(const ScreenParams p) {
    constexpr int BM = 256, BK = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    unsigned char *As = ks_smem;                                   // [2][32 groups][1152]
    unsigned char *Bs = ks_smem + 2 * KG_BUF;
    float *xn_s = reinterpret_cast<float *>(ks_smem + 4 * KG_BUF); // [2][BM]

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int j = b >> 3;
    const int stripe = (j / p.n_qtiles) * 8 + xcd;
    const int qtile = j % p.n_qtiles;
    if (stripe >= p.n_stripes) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, h = lane >> 5, l31 = lane & 31;
    const int dim = p.dim;
    const int n_kc = dim / BK;
    const int64_t q0 = (int64_t)qtile * KS_BN;
    const int first = stripe * p.tiles_per_block;
    int n_my = p.tiles_per_block;
    {
        const int last_tile = (p.n_tiles - 1) / p.tile_step;
        if (first > last_tile) return;
        if (first + n_my - 1 > last_tile) n_my = last_tile - first + 1;
    }
    const int n_chunks = n_my * n_kc;
    const _Float16 *const xh = p.xh;
    const _Float16 *const qh = p.qh;
    const int64_t n_rows = p.n_rows, n_queries = p.n_queries;
    auto tile_row0 = [&](int t) __attribute__((always_inline)) { return (int64_t)(first + t) * p.tile_step * BM; };

    typedef const void __attribute__((address_space(1))) *gptr_t;
    typedef void __attribute__((address_space(3))) *lptr_t;
    const int rr = lane >> 3;                       // row inside the 8-row group this lane fetches
    const int piece = (lane & 7) ^ rr;              // 16-byte piece of that row (XOR swizzle on the source side)
    auto dma_chunk = [&](int buf, int c) __attribute__((always_inline)) {
        const int t = c / n_kc, kc = c - t * n_kc;
        const int64_t r0 = tile_row0(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wave * 4 + i;             // wave-uniform group index: rows 8 g .. 8 g + 7 of both tiles
            int64_t r = r0 + g * 8 + rr;
            r = r < n_rows ? r : n_rows - 1;        // clamped; such rows are masked in the epilogue
            __builtin_amdgcn_global_load_lds((gptr_t)(xh + r * dim + kc * BK + piece * 8),
                                             (lptr_t)(As + (size_t)buf * KG_BUF + g * KG_GROUP), 16, 0, 0);
            int64_t q = q0 + g * 8 + rr;
            q = q < n_queries ? q : n_queries - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(qh + q * dim + kc * BK + piece * 8),
                                             (lptr_t)(Bs + (size_t)buf * KG_BUF + g * KG_GROUP), 16, 0, 0);
        }
        if (kc == 0 && wave < 4) {                  // ||x||^2 of the tile's 256 rows: 4 bytes per lane, same engine
            int64_t r = r0 + wave * 64 + lane;
            r = r < n_rows ? r : n_rows - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(p.xn + r), (lptr_t)(xn_s + (t & 1) * BM + wave * 64), 4, 0, 0);
        }
    };

    float thr[2], mn[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t q = q0 + 64 * wn + 32 * n + l31;
        mn[n] = INFINITY;
        thr[n] = -INFINITY;
        if (APPEND && q < n_queries) thr[n] = p.thr[q];
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // fragment addressing: row 128 wm + 32 m + l31 -> group (row >> 3), row-in-group l31 & 7, piece (4 h + t) ^ (l31 & 7)
    const int sw = l31 & 7;
    const int a_row_off = ((128 * wm + l31) >> 3) * KG_GROUP + sw * 128;     // + m * 4 groups
    const int b_row_off = ((64 * wn + l31) >> 3) * KG_GROUP + sw * 128;      // + n * 4 groups

    dma_chunk(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) dma_chunk(buf ^ 1, c + 1);     // buffer buf ^ 1 was last read in iteration c - 1 (barrier passed)
        const unsigned char *ab = As + (size_t)buf * KG_BUF + a_row_off;
        const unsigned char *bb = Bs + (size_t)buf * KG_BUF + b_row_off;
        f16x8 a[2][4], bq[2][2];
        auto frag = [&](int t, f16x8 (&av)[4], f16x8 (&bv)[2]) __attribute__((always_inline)) {
            const int po = ((4 * h + t) ^ sw) * 16;
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f16x8 *>(ab + m * 4 * KG_GROUP + po);
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[n] = *reinterpret_cast<const f16x8 *>(bb + n * 4 * KG_GROUP + po);
        };
        frag(0, a[0], bq[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) frag(t + 1, a[(t + 1) & 1], bq[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t & 1][m], bq[t & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int t_cur = c / n_kc;
        if (c - t_cur * n_kc == n_kc - 1) {
            const float *xn_t = xn_s + (t_cur & 1) * BM + 128 * wm + 4 * h;
            const int row_base = (int)(tile_row0(t_cur)) + 128 * wm + 4 * h;
            // rows past the end of the index were fetched clamped: their ||x||^2 is replaced by +inf
            screen_select<APPEND>(acc, xn_t, row_base, (int)n_rows, thr, mn, q0 + 64 * wn + l31, n_queries, n_rows, p.cand_cnt, p.cand_id, p.cap);
        }
        if (c + 1 < n_chunks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of chunk c + 1 have landed
            __syncthreads();                                   // ... and everybody's; everybody is done reading buf
        }
    }
    if (!APPEND) {
        const int slot = first * 4 + wm * 2 + h;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t q = q0 + 64 * wn + 32 * n + l31;
            if (q < n_queries) p.mins[q * p.n_slots + slot] = mn[n];
        }
    }
}

// ---- per-query bound from the sample minima (one wave per query) ---------------------------------------------------
__global__ void __launch_bounds__(256)
knn_select_kernel(const float *__restrict__ mins, int n_slots, const f32x4 *__restrict__ qstat, const unsigned *__restrict__ xstats,
                  int64_t n_queries, float *__restrict__ thr) {
    __shared__ float od[4][KNN_K];
    __shared__ int oi[4][KNN_K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    if (q >= n_queries) return;
    float d[KNN_K];
    int id[KNN_K];
#pragma unroll
    for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
    for (int c = lane; c < n_slots; c += 64) list_insert(d, id, mins[q * n_slots + c], c);
    wave_top8(d, id, lane, od[wave], oi[wave]);
    if (lane == 0) {
        const float s8 = od[wave][KNN_K - 1];
        const f32x4 st = qstat[q];                                  // {||q||^2, ||q - q~||^2, ||q~||^2, max |q|}
        const float xmax2 = __uint_as_float(xstats[0]), rxmax2 = __uint_as_float(xstats[1]), xabs = __uint_as_float(xstats[2]);
        const float rq = sqrtf(st[1]), qh = sqrtf(st[2]), xmax = sqrtf(xmax2), rxmax = sqrtf(rxmax2);
        // |s~ - s| <= 2 (||q - q~|| ||x|| + ||q~|| ||x - x~||) + fp32 accumulation of the dot product and of ||x||^2
        const float eps = 2.004f * (rq * xmax + qh * rxmax) + 2e-4f * qh * xmax + 1e-4f * xmax2 + 1e-6f;
        const float bound = s8 + eps;                               // >= the true 8th-best score
        float t = bound + eps + 2e-4f * fmaxf(bound + st[0], 0.f) + 1e-6f;
        // fp16 cannot hold the data (or too few finite minima): accept everything -> the exact scan answers
        if (!(s8 < INFINITY) || !(xabs < 65000.f) || !(st[3] < 65000.f) || !(t == t)) t = INFINITY;
        thr[q] = t;
    }
}

// ---- exact ranking of the candidates (one block per query) ---------------------------------------------------------
// cand_id [n_queries][cap]: row ids, negative = empty.  cand_cnt (optional): entries used per query; a count above cap
// means the list overflowed and may have dropped a neighbour -> the query is answered by scanning every row.
// cand_s (optional, list regimes of knn.hip): the GEMM-form score ||x||^2 - 2 q.x of each entry in fp32; entries whose
// score exceeds the 8th-best score by more than twice that form's error bound cannot be among the exact top 8 and are
// not re-scored (the lists hold 8 entries per row subset, thousands per query).
constexpr int KF_CHUNK = 2048;

constexpr int KF_WAVES = 8;

template <int J>   // dim = 256 J
__global__ void __launch_bounds__(64 * KF_WAVES)
knn_finalize_kernel(const float *__restrict__ index, int64_t n_rows, const float *__restrict__ queries,
                    const int *__restrict__ cand_id, const int *__restrict__ cand_cnt, const float *__restrict__ cand_s,
                    const unsigned *__restrict__ xstats, int cap, float *__restrict__ out_d2, int64_t *__restrict__ out_ids,
                    int *__restrict__ n_exact_scans) {
    constexpr int dim = 256 * J;
    __shared__ float d_s[KF_CHUNK];
    __shared__ int id_s[KF_CHUNK];
    __shared__ float best_d[KNN_K];
    __shared__ int best_i[KNN_K];
    __shared__ float wtop[KF_WAVES][KNN_K];
    __shared__ int wtopi[KF_WAVES][KNN_K];
    __shared__ int keep_s[KF_CHUNK];
    __shared__ int n_keep;
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 qv[J];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) qv[jj] = *reinterpret_cast<const f32x4 *>(queries + q * dim + 256 * jj + 4 * lane);
    if (tid < KNN_K) { best_d[tid] = INFINITY; best_i[tid] = 0x7fffffff; }
    int64_t total = cap;
    bool scan_all = false;
    if (cand_cnt) {
        const int c = cand_cnt[q];
        total = c < cap ? c : cap;
        scan_all = c > cap;
    }
    if (scan_all) {
        total = n_rows;
        if (tid == 0 && n_exact_scans) atomicAdd(n_exact_scans, 1);
    }
    float tau = INFINITY;
    bool compact = false;
    if (cand_s && !scan_all) {
        // 8th-best GEMM-form score of the lists (block-wide), then the cut-off
        float d[KNN_K];
        int id[KNN_K];
#pragma unroll
        for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
        for (int c = tid; c < total; c += 64 * KF_WAVES)
            if (cand_id[q * cap + c] >= 0) list_insert(d, id, cand_s[q * cap + c], c);
        wave_top8(d, id, lane, wtop[wave], wtopi[wave]);
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
            static_assert(KF_WAVES * KNN_K <= 64, "one candidate per lane in the second merge level");
            if (lane < KF_WAVES * KNN_K) { d[0] = wtop[lane >> 3][lane & 7]; id[0] = wtopi[lane >> 3][lane & 7]; }
            wave_top8(d, id, lane, wtop[0], wtopi[0]);
        }
        __syncthreads();
        float qn = 0.f;
#pragma unroll
        for (int jj = 0; jj < J; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) qn = fmaf(qv[jj][e], qv[jj][e], qn);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off);
        const float xmax2 = __uint_as_float(xstats[0]);
        // |score_fp32 - score| <= gamma (||x||^2 + 2 ||q|| ||x||) <= gamma 2 (||x||^2 + ||q||^2), gamma ~ dim 2^-24
        const float eps32 = 2e-4f * 2.f * (xmax2 + qn) + 1e-6f;
        tau = wtop[0][KNN_K - 1] + 2.f * eps32;
        if (!(tau == tau)) tau = INFINITY;
        // keep the entries under the cut-off (a dozen out of thousands), compacted in LDS
        if (tid == 0) n_keep = 0;
        __syncthreads();
        for (int c = tid; c < total; c += 64 * KF_WAVES) {
            const int id = cand_id[q * cap + c];
            if (id >= 0 && cand_s[q * cap + c] <= tau) {
                const int pos = atomicAdd(&n_keep, 1);
                if (pos < KF_CHUNK) keep_s[pos] = id;
            }
        }
        __syncthreads();
        compact = n_keep <= KF_CHUNK;          // otherwise (thousands of exact ties) walk the lists themselves
        if (compact) total = n_keep;
    }
    __syncthreads();
    for (int64_t base = 0; base < total; base += KF_CHUNK) {
        const int n = (int)((total - base) < KF_CHUNK ? (total - base) : KF_CHUNK);
        constexpr int U = 4;                                  // candidates per wave per trip: their row loads overlap
        for (int c0 = U * wave; c0 < n; c0 += KF_WAVES * U) {
            int id[U];
            f32x4 xv[U][J];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u;
                id[u] = -1;
                if (c < n) {
                    if (scan_all) id[u] = (int)(base + c);
                    else if (compact) id[u] = keep_s[c];
                    else {
                        id[u] = cand_id[q * cap + base + c];
                        if (id[u] >= n_rows) id[u] = -1;      // a caller-supplied id outside the index is ignored, never read
                        if (cand_s && id[u] >= 0 && !(cand_s[q * cap + base + c] <= tau)) id[u] = -1;
                    }
                }
                const float *row = index + (int64_t)(id[u] >= 0 ? id[u] : 0) * dim + 4 * lane;
#pragma unroll
                for (int jj = 0; jj < J; ++jj) xv[u][jj] = *reinterpret_cast<const f32x4 *>(row + 256 * jj);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int jj = 0; jj < J; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float df = qv[jj][e] - xv[u][jj][e];
                        acc = fmaf(df, df, acc);
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
                if (lane == 0 && c0 + u < n) {
                    d_s[c0 + u] = id[u] >= 0 ? acc : INFINITY;
                    id_s[c0 + u] = id[u] >= 0 ? id[u] : 0x7fffffff;
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            float d[KNN_K];
            int id[KNN_K];
#pragma unroll
            for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
            if (lane < KNN_K) { d[0] = best_d[lane]; id[0] = best_i[lane]; }
            for (int c = lane; c < n; c += 64)
                if (id_s[c] != 0x7fffffff) list_insert(d, id, d_s[c], id_s[c]);
            wave_top8(d, id, lane, wtop[1], wtopi[1]);
            if (lane < KNN_K) { best_d[lane] = wtop[1][lane]; best_i[lane] = wtopi[1][lane]; }
        }
        __syncthreads();
    }
    if (tid < KNN_K) {
        out_d2[q * KNN_K + tid] = best_d[tid];
        out_ids[q * KNN_K + tid] = best_i[tid] == 0x7fffffff ? -1 : (int64_t)best_i[tid];
    }
}

// ---- aux blob of an index: [norms fp32 n_rows][8 stat words][fp16 copy n_rows x dim] ---------------------------------
struct KnnAux {
    size_t norms, stats, half, total;
};
static KnnAux knn_aux(int64_t n_rows, int dim) {
    KnnAux a;
    a.norms = 0;
    a.stats = align_up((size_t)n_rows * sizeof(float), 256);
    a.half = a.stats + 256;
    a.total = a.half + align_up((size_t)n_rows * dim * sizeof(_Float16), 256);
    return a;
}

struct ScreenPlan {
    int wm, bk, bm, slots_per_tile;     // tile shape (ScreenTile<WM, BK>)
    int n_tiles, n_qtiles;
    int sample_tiles, sample_step;      // MIN pass
    int tiles_per_block, n_stripes;     // APPEND pass
    int cap;
    size_t qh, qstat, mins, thr, cnt, cand, total;   // workspace offsets
};

static ScreenPlan screen_plan(int64_t n_rows, int64_t n_queries, int dim) {
    ScreenPlan s;
    // measured at 1599 x 100 k / 2 M (ms per search): WM 2 BK 64 0.541 / 8.77; WM 2 BK 32 0.557; WM 1 BK 32 (two blocks per CU)
    // 0.607 / 11.9; WM 1 BK 64 spills (1.13).  Both operands stream at ~11 B/clk/CU: the large tile's lower traffic per flop wins.
    static const int tile_env = knob("RVC_KNN_TILE", 0);      // 264 = WM 2 / BK 64 (default), 132 = WM 1 / BK 32
    const int tile = tile_env == 132 ? 132 : 264;
    s.wm = tile / 100;
    s.bk = tile % 100;
    s.bm = 128 * s.wm;
    s.slots_per_tile = 2 * s.wm;
    s.n_tiles = (int)ceil_div(n_rows, s.bm);
    s.n_qtiles = (int)ceil_div(n_queries, KS_BN);
    // sample size ~ sqrt(2560 N) rows balances the sample pass against the exact re-scoring of ~8 N / sample survivors
    static const int sample_env = knob("RVC_KNN_SAMPLE_TILES", 0);
    int64_t want = 1;
    while (want * want * s.bm * s.bm < 2560 * n_rows) ++want;
    // One block per (sampled tile, query tile), one block per CU: between one and two rounds of the 256 CUs the second round
    // runs nearly empty, and a smaller sample costs less than it saves (1599 x 100 k: 63 tiles x 7 = 441 blocks 56 us; 32
    // tiles 28 us, the candidate lists grow from ~50 to ~90 entries, search 0.466 -> 0.444 ms)
    if (want * s.n_qtiles > 256 && want * s.n_qtiles < 512) want = 224 / s.n_qtiles;
    if (sample_env) want = sample_env;
    if (want < 16) want = 16;
    if (want > 512) want = 512;
    if (want > s.n_tiles) want = s.n_tiles;
    s.sample_step = s.n_tiles / (int)want;
    s.sample_tiles = (int)want;
    static const int bpc_env = knob("RVC_KNN_SCREEN_BLOCKS", 0);
    // one (row tile, query tile) pair per block up to 16384 blocks: 1599 x 100 k 0.459 -> 0.444 ms, 1599 x 2 M 8.62 -> 7.54 ms
    // against 1024 longer blocks (finer blocks balance the XCDs and hide each other's prologue)
    const int64_t blocks_want = bpc_env ? bpc_env : 16384;
    s.tiles_per_block = (int)ceil_div((int64_t)s.n_tiles * s.n_qtiles, blocks_want);
    if (s.tiles_per_block < 1) s.tiles_per_block = 1;
    s.n_stripes = (int)ceil_div(s.n_tiles, s.tiles_per_block);
    static const int cap_env = knob("RVC_KNN_CAP", 0);
    s.cap = cap_env ? cap_env : 8192;   // candidates per query; survivors are a few dozen on spread-out data, thousands when
                                        // thousands of rows are equidistant from a query to within the fp16 error bound
    size_t off = 0;
    s.qh = off; off += align_up((size_t)n_queries * dim * sizeof(_Float16), 256);
    s.qstat = off; off += align_up((size_t)n_queries * sizeof(f32x4), 256);
    s.mins = off; off += align_up((size_t)n_queries * 512 * 4 * sizeof(float), 256);   // room for any tile shape
    s.thr = off; off += align_up((size_t)n_queries * sizeof(float), 256);
    s.cnt = off; off += align_up((size_t)(n_queries + 1) * sizeof(int), 256);     // + the exact-scan counter
    s.cand = off; off += align_up((size_t)n_queries * s.cap * sizeof(int), 256);
    s.total = off;
    return s;
}

// rvc_knn_set_mode (test hook): 0 auto, 1 exact fp32 GEMM, 2 screened whenever the shape allows.  Per calling thread, so a
// test that pins a regime cannot change what a concurrent convert_batch worker thread runs.
static thread_local int g_knn_mode = 0;

bool knn_screen_applicable(int64_t n_rows, int64_t n_queries, int dim) {
    if (g_knn_mode == 1) return false;
    if (dim % 256 != 0 || dim > 1024) return false;
    if (n_rows >= ((int64_t)1 << 31) - 256) return false;
    if (g_knn_mode == 2) return n_rows >= 4096;
    return n_queries > 64 && n_rows >= 16384;
}

size_t knn_screen_workspace_bytes(int64_t n_rows, int64_t n_queries, int dim) { return screen_plan(n_rows, n_queries, dim).total; }

int knn_finalize_launch(const float *index, int64_t n_rows, int dim, const float *queries, int64_t n_queries, const int *cand_id,
                        const int *cand_cnt, const float *cand_s, const void *aux_dev, int cap, float *out_d2, int64_t *out_ids,
                        int *n_exact_scans, hipStream_t stream) {
    const dim3 grid((unsigned)n_queries), block(64 * KF_WAVES);
    const unsigned *xstats = (const unsigned *)((const char *)aux_dev + knn_aux(n_rows, dim).stats);
#define RVC_FIN(JJ)                                                                                                        \
    hipLaunchKernelGGL(knn_finalize_kernel<JJ>, grid, block, 0, stream, index, n_rows, queries, cand_id, cand_cnt, cand_s, xstats, \
                       cap, out_d2, out_ids, n_exact_scans)
    switch (dim / 256) {
        case 1: RVC_FIN(1); break;
        case 2: RVC_FIN(2); break;
        case 3: RVC_FIN(3); break;
        case 4: RVC_FIN(4); break;
        default: return fail("knn finalize: dim %d is not 256, 512, 768 or 1024", dim);
    }
#undef RVC_FIN
    RVC_LAUNCH_CHECK();
    return 0;
}

int knn_screened_search(const float *index, const void *aux_dev, int64_t n_rows, int dim, const float *queries, int64_t n_queries,
                        float *out_d2, int64_t *out_ids, void *workspace, hipStream_t stream) {
    const KnnAux aux = knn_aux(n_rows, dim);
    const ScreenPlan s = screen_plan(n_rows, n_queries, dim);
    const char *ab = (const char *)aux_dev;
    char *ws = (char *)workspace;
    _Float16 *qh = (_Float16 *)(ws + s.qh);
    f32x4 *qstat = (f32x4 *)(ws + s.qstat);
    float *mins = (float *)(ws + s.mins);
    float *thr = (float *)(ws + s.thr);
    int *cnt = (int *)(ws + s.cnt);
    int *cand = (int *)(ws + s.cand);

    static std::once_flag lds_once;
    static hipError_t lds_err = hipSuccess;
    std::call_once(lds_once, [] {
        auto reserve = [](const void *fn, size_t bytes) {
            if (lds_err == hipSuccess) lds_err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        };
        static_assert(ScreenTile<2, 64>::LDS_BYTES <= LDS_WHOLE_CU, "");
        reserve((const void *)knn_screen_glds_kernel<true>, LDS_WHOLE_CU);      // fp16 matrix instructions: these own their CU (common.h)
        reserve((const void *)knn_screen_glds_kernel<false>, LDS_WHOLE_CU);
        reserve((const void *)knn_screen_kernel<true, 2, 64>, LDS_WHOLE_CU);
        reserve((const void *)knn_screen_kernel<false, 2, 64>, LDS_WHOLE_CU);
        reserve((const void *)knn_screen_kernel<true, 1, 32>, ScreenTile<1, 32>::LDS_BYTES);
        reserve((const void *)knn_screen_kernel<false, 1, 32>, ScreenTile<1, 32>::LDS_BYTES);
    });
    if (lds_err != hipSuccess) return fail("knn screen: cannot reserve LDS: %s", hipGetErrorString(lds_err));

    hipLaunchKernelGGL(knn_to_half_kernel, dim3((unsigned)ceil_div(n_queries, 4)), dim3(256), 0, stream, queries, n_queries, dim, qh,
                       (float *)nullptr, qstat, (unsigned *)nullptr, mins, s.sample_tiles * s.slots_per_tile, cnt);
    RVC_LAUNCH_CHECK();

    ScreenParams p;
    p.xh = (const _Float16 *)(ab + aux.half);
    p.xn = (const float *)(ab + aux.norms);
    p.n_rows = n_rows; p.dim = dim; p.qh = qh; p.n_queries = n_queries;
    p.n_qtiles = s.n_qtiles; p.n_tiles = s.n_tiles;
    p.thr = thr; p.cand_id = cand; p.cand_cnt = cnt; p.cap = s.cap;
    p.mins = mins; p.n_slots = s.sample_tiles * s.slots_per_tile;
    // sample pass: one tile per block, tiles sample_step apart
    p.n_stripes = s.sample_tiles; p.tiles_per_block = 1; p.tile_step = s.sample_step;
    unsigned blocks = (unsigned)(ceil_div(p.n_stripes, 8) * 8 * p.n_qtiles);
    auto launch = [&](bool append, unsigned nblocks) {
#define RVC_SCREEN(AP, W, K)                                                                                               \
    hipLaunchKernelGGL((knn_screen_kernel<AP, W, K>), dim3(nblocks), dim3((ScreenTile<W, K>::THREADS)),                   \
                       ((W) == 2 ? (size_t)LDS_WHOLE_CU : (size_t)ScreenTile<W, K>::LDS_BYTES), stream, p)
        static const int glds_env = knob("RVC_KNN_GLDS", 0);   // 1: the LDS-DMA staged variant (measured equal: 0.555 vs 0.545 ms)
        if (s.wm == 2 && s.bk == 64 && glds_env) {
            if (append) hipLaunchKernelGGL(knn_screen_glds_kernel<true>, dim3(nblocks), dim3(512), LDS_WHOLE_CU, stream, p);
            else hipLaunchKernelGGL(knn_screen_glds_kernel<false>, dim3(nblocks), dim3(512), LDS_WHOLE_CU, stream, p);
        } else if (s.wm == 2 && s.bk == 64) {
#ifdef RVC_ABLATE
            static const int dbg = knob("RVC_KNN_DBG", 0);
            if (append && dbg) {
#define RVC_SCREEN_DBG(D) hipLaunchKernelGGL((knn_screen_kernel<true, 2, 64, D>), dim3(nblocks), dim3(512), (ScreenTile<2, 64>::LDS_BYTES), stream, p)
                auto res = [](const void *fn) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ScreenTile<2, 64>::LDS_BYTES); };
                switch (dbg) {
#define RVC_SD(D) case D: res((const void *)knn_screen_kernel<true, 2, 64, D>); RVC_SCREEN_DBG(D); break
                    RVC_SD(32); RVC_SD(34); RVC_SD(48); RVC_SD(50); RVC_SD(54); RVC_SD(36); RVC_SD(40);
#undef RVC_SD
                    default: RVC_SCREEN(true, 2, 64); break;
                }
#undef RVC_SCREEN_DBG
            } else
#endif
            if (append) RVC_SCREEN(true, 2, 64); else RVC_SCREEN(false, 2, 64);
        }
        else { if (append) RVC_SCREEN(true, 1, 32); else RVC_SCREEN(false, 1, 32); }
#undef RVC_SCREEN
    };
    launch(false, blocks);
    RVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)ceil_div(n_queries, 4)), dim3(256), 0, stream, mins, p.n_slots, qstat,
                       (const unsigned *)(ab + aux.stats), n_queries, thr);
    RVC_LAUNCH_CHECK();
    // main pass over every row
    p.n_stripes = s.n_stripes; p.tiles_per_block = s.tiles_per_block; p.tile_step = 1;
    blocks = (unsigned)(ceil_div(p.n_stripes, 8) * 8 * p.n_qtiles);
    launch(true, blocks);
    RVC_LAUNCH_CHECK();
    return knn_finalize_launch(index, n_rows, dim, queries, n_queries, cand, cnt, nullptr, aux_dev, s.cap, out_d2, out_ids,
                               cnt + n_queries, stream);
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_knn_index_aux_bytes(int64_t n_rows, int dim, size_t *bytes) {
    if (!bytes || n_rows <= 0 || dim <= 0 || dim % 4) return fail("rvc_knn_index_aux_bytes: bad argument");
    *bytes = knn_aux(n_rows, dim).total;
    return 0;
}

extern "C" int rvc_knn_index_build(const float *index_dev, int64_t n_rows, int dim, void *aux_dev, size_t aux_bytes, void *stream) {
    if (!index_dev || !aux_dev || n_rows <= 0 || dim <= 0 || dim % 4) return fail("rvc_knn_index_build: bad argument");
    const KnnAux a = knn_aux(n_rows, dim);
    if (aux_bytes < a.total) return fail("rvc_knn_index_build: aux buffer too small (%zu < %zu)", aux_bytes, a.total);
    char *ab = (char *)aux_dev;
    RVC_HIP(hipMemsetAsync(ab + a.stats, 0, 256, (hipStream_t)stream));
    hipLaunchKernelGGL(knn_to_half_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, index_dev, n_rows,
                       dim, (_Float16 *)(ab + a.half), (float *)(ab + a.norms), (f32x4 *)nullptr, (unsigned *)(ab + a.stats),
                       (float *)nullptr, 0, (int *)nullptr);
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_knn_rank_candidates(const float *index_dev, const void *aux_dev, int64_t n_rows, int dim, const float *queries_dev,
                                       int64_t n_queries, const int32_t *cand_ids_dev, int cap, int k, float *out_d2_dev,
                                       int64_t *out_ids_dev, void *stream) {
    if (k != KNN_K) return fail("rvc_knn_rank_candidates: k must be 8, got %d", k);
    if (!index_dev || !aux_dev || !queries_dev || !cand_ids_dev || !out_d2_dev || !out_ids_dev) return fail("rvc_knn_rank_candidates: null pointer");
    if (dim % 256 != 0 || dim > 1024) return fail("rvc_knn_rank_candidates: dim must be 256, 512, 768 or 1024, got %d", dim);
    if (cap <= 0 || n_rows <= 0) return fail("rvc_knn_rank_candidates: bad shape");
    if (n_queries == 0) return 0;
    return knn_finalize_launch(index_dev, n_rows, dim, queries_dev, n_queries, cand_ids_dev, nullptr, nullptr, aux_dev, cap, out_d2_dev,
                               out_ids_dev, nullptr, (hipStream_t)stream);
}

extern "C" int rvc_knn_set_mode(int mode) {
    if (mode < 0 || mode > 2) return fail("rvc_knn_set_mode: 0 (auto), 1 (exact fp32 GEMM) or 2 (fp16-screened), got %d", mode);
    g_knn_mode = mode;
    return 0;
}
