// K1 -- exact squared-L2 top-8 retrieval over the [N, 768] fp32 feature index, and the reference's
// inverse-distance blend.  Replaces pipeline.py:497-507 (faiss index.search + NumPy weighting).
//
// Shape of the problem (BASELINE cfg 2): Q = 1599 queries, N = 100 000 rows, D = 768 -> 245.6 GFLOP of
// q.x dot products over a 307 MB index.  That is a GEMM with a selection epilogue; the dots run on the
// exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) and the selection is a
// per-lane register-resident sorted top-8.
//
// Decomposition
//   grid = (query tiles of 128) x (index stripes); block = 4 waves, block tile 128 index rows x 128 queries,
//   wave tile 64 x 64 = 2x2 MFMA tiles.  The index rows are the MFMA "A" (row) operand and the queries the
//   "B" (column) operand, so one lane owns ONE query column per MFMA tile and sees 16 candidate rows of it
//   in its accumulator registers: the running top-8 of that query lives in that lane's registers and no
//   cross-lane traffic is needed until the very end.
//   K loop: 32-float chunks of both operands are staged HBM -> registers -> LDS ([row][33] floats, the odd
//   stride makes the column-wise fragment reads conflict-free), with the next chunk's global loads in flight
//   under the current chunk's 64 MFMAs per wave.
//   Partial lists ([query][slot][8], slot = stripe x wave-row x lane-half) are merged by one wave per query.
//
// d2 = ||x||^2 - 2 q.x + ||q||^2 is faiss' own BLAS formulation (IndexFlat, > 20 queries), clamped at 0.
#include <stdlib.h>

#include <mutex>

#include "knn_common.h"

namespace rvc {

__global__ void __launch_bounds__(256)
knn_norms_kernel(const float *__restrict__ x, int64_t n_rows, int dim, float *__restrict__ norms) {
    // one wave per row, float4 loads
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float4 *p = reinterpret_cast<const float4 *>(x + row * dim);
    float s = 0.f;
    for (int i = lane; i < dim / 4; i += 64) {
        const float4 v = p[i];
        s = fmaf(v.x, v.x, s);
        s = fmaf(v.y, v.y, s);
        s = fmaf(v.z, v.z, s);
        s = fmaf(v.w, v.w, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) norms[row] = s;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))   // 238 VGPRs: two waves per SIMD let one wave's top-8 inserts run under the other's MFMAs (94 -> 110 TFLOP/s)
knn_partial_kernel(const float *__restrict__ index, const float *__restrict__ norms, int64_t n_rows, int dim,
                   const float *__restrict__ queries, int64_t n_queries, int64_t stripe_rows,
                   float *__restrict__ part_d, int *__restrict__ part_id, int n_slots) {
    __shared__ float Xs[KNN_BN * KNN_LDS_STRIDE];
    __shared__ float Qs[KNN_BQ * KNN_LDS_STRIDE];
    __shared__ float xn_s[KNN_BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1;   // which 64 index rows of the tile
    const int wn = wave & 1;    // which 64 queries of the tile
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int64_t q0 = (int64_t)blockIdx.x * KNN_BQ;
    const int64_t stripe_begin = (int64_t)blockIdx.y * stripe_rows;
    const int64_t stripe_end = min(n_rows, stripe_begin + stripe_rows);
    const int n_tiles = (int)((stripe_end - stripe_begin + KNN_BN - 1) / KNN_BN);
    const int n_kc = dim / KNN_KC;

    TopK best[2];
    best[0].init();
    best[1].init();

    // staging registers: 4 passes x (32 rows x 8 float4) for each operand
    float4 xr[4], qr[4];
    const int srow = tid >> 3;
    const int sc4 = tid & 7;

    auto load_chunk = [&](int tile, int kc) {
        const int64_t n_base = stripe_begin + (int64_t)tile * KNN_BN;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t n = n_base + p * 32 + srow;
            xr[p] = (n < stripe_end)
                        ? *reinterpret_cast<const float4 *>(index + n * dim + kc * KNN_KC + sc4 * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t q = q0 + p * 32 + srow;
            qr[p] = (q < n_queries)
                        ? *reinterpret_cast<const float4 *>(queries + q * dim + kc * KNN_KC + sc4 * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float *xd = &Xs[(p * 32 + srow) * KNN_LDS_STRIDE + sc4 * 4];
            xd[0] = xr[p].x; xd[1] = xr[p].y; xd[2] = xr[p].z; xd[3] = xr[p].w;
            float *qd = &Qs[(p * 32 + srow) * KNN_LDS_STRIDE + sc4 * 4];
            qd[0] = qr[p].x; qd[1] = qr[p].y; qd[2] = qr[p].z; qd[3] = qr[p].w;
        }
    };

    if (n_tiles > 0) load_chunk(0, 0);

    for (int tile = 0; tile < n_tiles; ++tile) {
        const int64_t n_base = stripe_begin + (int64_t)tile * KNN_BN;
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        for (int kc = 0; kc < n_kc; ++kc) {
            __syncthreads();  // everyone is done reading the previous chunk (and the previous tile's xn_s)
            store_chunk();
            if (kc == 0 && tid < KNN_BN) {
                const int64_t n = n_base + tid;
                xn_s[tid] = (n < stripe_end) ? norms[n] : INFINITY;
            }
            __syncthreads();
            // prefetch the next chunk while this one is multiplied
            if (kc + 1 < n_kc) load_chunk(tile, kc + 1);
            else if (tile + 1 < n_tiles) load_chunk(tile + 1, 0);

            const float *xa = &Xs[(wm * 64 + l31) * KNN_LDS_STRIDE + half];
            const float *qb = &Qs[(wn * 64 + l31) * KNN_LDS_STRIDE + half];
#pragma unroll
            for (int kk = 0; kk < KNN_KC / 2; ++kk) {
                const float a0 = xa[2 * kk];
                const float a1 = xa[32 * KNN_LDS_STRIDE + 2 * kk];
                const float b0 = qb[2 * kk];
                const float b1 = qb[32 * KNN_LDS_STRIDE + 2 * kk];
                acc[0][0] = mfma32(a0, b0, acc[0][0]);
                acc[0][1] = mfma32(a0, b1, acc[0][1]);
                acc[1][0] = mfma32(a1, b0, acc[1][0]);
                acc[1][1] = mfma32(a1, b1, acc[1][1]);
            }
        }
        // selection: lane owns query columns (wn*64 + nt*32 + l31); its registers hold 32 candidate rows
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + m * 32 + mfma32_row(r, lane);
                const float xn = xn_s[row];
                const int n = (int)(n_base + row);
                best[0].insert(fmaf(-2.f, acc[m][0][r], xn), n);
                best[1].insert(fmaf(-2.f, acc[m][1][r], xn), n);
            }
        }
    }

    const int slot = blockIdx.y * KNN_SLOTS_PER_STRIPE + wm * 2 + half;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int64_t q = q0 + wn * 64 + nt * 32 + l31;
        if (q < n_queries) {
            float *pd = part_d + (q * n_slots + slot) * KNN_K;
            int *pi = part_id + (q * n_slots + slot) * KNN_K;
#pragma unroll
            for (int i = 0; i < KNN_K; ++i) { pd[i] = best[nt].d[i]; pi[i] = best[nt].id[i]; }
        }
    }
}


// ---- streaming variant for few queries (Q <= 64): HBM-bound ---------------------------------------------------
// With one or two 32-query column tiles the GEMM is tiny (2*32*N*768 FLOP = 49 us of MFMA at N = 100 k) and the index
// read (N*3072 B) is the bound, so the job is to keep HBM busy: a block still transposes 128 rows x 32 floats
// through LDS per chunk, but THREE chunks of index data are in flight per block (register ring, 48 KB per block,
// ~6 blocks per CU) instead of one, and stripes are short so that ~8 blocks per CU exist.  Each of the 4 waves owns
// 32 of the block's 128 rows and one 32-query MFMA tile; per-lane sorted top-8 lists as in the batch kernel.
constexpr int KNS_BQ = 32;
constexpr int KNS_SLOTS_PER_STRIPE = 1;   // the block merges its 8 (wave, lane-half) lists before writing

template <int KNS_PF>
__global__ void __launch_bounds__(256)
knn_stream_kernel(const float *__restrict__ index, const float *__restrict__ norms, int64_t n_rows, int dim,
                  const float *__restrict__ queries, int64_t n_queries, int64_t stripe_rows, float *__restrict__ part_d,
                  int *__restrict__ part_id, int n_slots) {
    __shared__ float Xs[KNN_BN * KNN_LDS_STRIDE];
    __shared__ float Qs[KNS_BQ * KNN_LDS_STRIDE];
    __shared__ float xn_s[KNN_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t q0 = (int64_t)blockIdx.x * KNS_BQ;
    const int64_t stripe_begin = (int64_t)blockIdx.y * stripe_rows;
    const int64_t stripe_end = min(n_rows, stripe_begin + stripe_rows);
    const int n_tiles = (int)((stripe_end - stripe_begin + KNN_BN - 1) / KNN_BN);
    const int n_kc = dim / KNN_KC;
    const int n_chunks = n_tiles * n_kc;
    TopK best;
    best.init();

    float4 xr[KNS_PF][4];
    float4 qr[KNS_PF];
    const int srow = tid >> 3, sc4 = tid & 7;
    auto load_chunk = [&](int slot, int c) {
        const int tile = c / n_kc, kc = c - tile * n_kc;
        const int64_t n_base = stripe_begin + (int64_t)tile * KNN_BN;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t n = n_base + p * 32 + srow;
            xr[slot][p] = (n < stripe_end) ? *reinterpret_cast<const float4 *>(index + n * dim + kc * KNN_KC + sc4 * 4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int64_t q = q0 + srow;
        qr[slot] = (q < n_queries) ? *reinterpret_cast<const float4 *>(queries + q * dim + kc * KNN_KC + sc4 * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_chunk = [&](int slot) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float *xd = &Xs[(p * 32 + srow) * KNN_LDS_STRIDE + sc4 * 4];
            xd[0] = xr[slot][p].x; xd[1] = xr[slot][p].y; xd[2] = xr[slot][p].z; xd[3] = xr[slot][p].w;
        }
        float *qd = &Qs[srow * KNN_LDS_STRIDE + sc4 * 4];
        qd[0] = qr[slot].x; qd[1] = qr[slot].y; qd[2] = qr[slot].z; qd[3] = qr[slot].w;
    };

#pragma unroll
    for (int s = 0; s < KNS_PF; ++s)
        if (s < n_chunks) load_chunk(s, s);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int c0 = 0; c0 < n_chunks; c0 += KNS_PF) {
#pragma unroll
        for (int s = 0; s < KNS_PF; ++s) {
            const int c = c0 + s;
            if (c < n_chunks) {   // block-uniform
                const int tile = c / n_kc, kc = c - tile * n_kc;
                const int64_t n_base = stripe_begin + (int64_t)tile * KNN_BN;
                __syncthreads();
                store_chunk(s);
                if (kc == 0 && tid < KNN_BN) {
                    const int64_t n = n_base + tid;
                    xn_s[tid] = (n < stripe_end) ? norms[n] : INFINITY;
                }
                __syncthreads();
                if (c + KNS_PF < n_chunks) load_chunk(s, c + KNS_PF);
                const float *xa = &Xs[(wave * 32 + l31) * KNN_LDS_STRIDE + half];
                const float *qb = &Qs[l31 * KNN_LDS_STRIDE + half];
#pragma unroll
                for (int kk = 0; kk < KNN_KC / 2; ++kk) acc = mfma32(xa[2 * kk], qb[2 * kk], acc);
                if (kc == n_kc - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wave * 32 + mfma32_row(r, lane);
                        best.insert(fmaf(-2.f, acc[r], xn_s[row]), (int)(n_base + row));
                        acc[r] = 0.f;
                    }
                }
            }
        }
    }
    // merge the block's 8 (wave, lane-half) lists per query in LDS -> ONE sorted list per (query, stripe)
    __syncthreads();
    float *md = Xs;                                        // [32 queries][64 candidates], reusing the staging tile
    int *mi = reinterpret_cast<int *>(Xs + KNS_BQ * 64);
#pragma unroll
    for (int i = 0; i < KNN_K; ++i) {
        md[l31 * 64 + (wave * 2 + half) * KNN_K + i] = best.d[i];
        mi[l31 * 64 + (wave * 2 + half) * KNN_K + i] = best.id[i];
    }
    __syncthreads();
    if (tid < KNS_BQ) {
        TopK m;
        m.init();
        for (int c = 0; c < 64; ++c) {   // lists arrive in ascending row order within a slot; ties keep the lower id
            const float v = md[tid * 64 + c];
            const int n = mi[tid * 64 + c];
            if (n < 0) continue;
            if (v < m.d[KNN_K - 1] || (v == m.d[KNN_K - 1] && (unsigned)n < (unsigned)m.id[KNN_K - 1])) {
#pragma unroll
                for (int p = KNN_K - 1; p >= 1; --p) {
                    const bool shift = v < m.d[p - 1] || (v == m.d[p - 1] && (unsigned)n < (unsigned)m.id[p - 1]);
                    const bool here = v < m.d[p] || (v == m.d[p] && (unsigned)n < (unsigned)m.id[p]);
                    const float nd = shift ? m.d[p - 1] : (here ? v : m.d[p]);
                    const int ni = shift ? m.id[p - 1] : (here ? n : m.id[p]);
                    m.d[p] = nd;
                    m.id[p] = ni;
                }
                if (v < m.d[0] || (v == m.d[0] && (unsigned)n < (unsigned)m.id[0])) { m.d[0] = v; m.id[0] = n; }
            }
        }
        const int64_t q = q0 + tid;
        if (q < n_queries) {
            float *pd = part_d + (q * n_slots + blockIdx.y) * KNN_K;
            int *pi = part_id + (q * n_slots + blockIdx.y) * KNN_K;
#pragma unroll
            for (int i = 0; i < KNN_K; ++i) { pd[i] = m.d[i]; pi[i] = m.id[i]; }
        }
    }
}

// ---- streaming kernel, second form: operands straight from HBM into the MFMA, no LDS staging, no barriers ----------
// For <= 32 queries the search is one pass over the index and is bound by how fast a CU can pull rows out of HBM
// while keeping its matrix cores fed (16 flop per index byte at 32 queries: 6.3 TB/s needs ~100 TFLOP/s of fp32
// MFMA).  The LDS-staged form above spends two block barriers per 16 KB chunk and tops out at ~4.2 TB/s.  Here:
//   * the 32 (zero-padded) queries sit in LDS once per block, rows padded to dim + 4 floats so that the
//     ds_read_b128 fragment reads are conflict-free;
//   * every wave owns whole 32-row tiles and feeds the MFMA "A" operand directly from its global loads.  The dot
//     product does not care in which order k is walked as long as both operands agree, so a lane (row i, half h)
//     takes the 64 contiguous bytes [32 j + 16 h, +16) floats of its row per 128-byte line group j -- four dwordx4
//     loads -- and the k-pair of MFMA (t, e) is (32 j + 4 t + e, 32 j + 16 + 4 t + e); the query fragment is the
//     same 16 floats of the query row, one ds_read_b128 per four MFMAs;
//   * KND_NB line groups (12 dwordx4 per lane) are kept in flight per wave; waves never synchronise until the end.
constexpr int KND_BQ = 32;

template <int NW, int KND_NB>
__global__ void __launch_bounds__(NW * 64)
knn_direct_kernel(const float *__restrict__ index, const float *__restrict__ norms, int64_t n_rows, int dim,
                  const float *__restrict__ queries, int64_t n_queries, int64_t stripe_rows, float *__restrict__ part_d,
                  int *__restrict__ part_id, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) float knd_smem[];
    float *qs = knd_smem;                    // [32][dim + 4]
    const int qstride = dim + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t q0 = (int64_t)blockIdx.x * KND_BQ;
    const int d4 = dim / 4;
    for (int idx = tid; idx < KND_BQ * d4; idx += NW * 64) {
        const int q = idx / d4, c4 = idx - q * d4;
        const int64_t qq = q0 + q < n_queries ? q0 + q : n_queries - 1;
        f32x4 v = *reinterpret_cast<const f32x4 *>(queries + qq * dim + c4 * 4);
        if (q0 + q >= n_queries) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(qs + q * qstride + c4 * 4) = v;
    }
    __syncthreads();

    const int64_t stripe_begin = (int64_t)blockIdx.y * stripe_rows;
    const int64_t stripe_end = min(n_rows, stripe_begin + stripe_rows);
    const int n_tiles = stripe_end > stripe_begin ? (int)((stripe_end - stripe_begin + 31) / 32) : 0;
    const int n_my = wave < n_tiles ? (n_tiles - wave + NW - 1) / NW : 0;   // tiles wave, wave + NW, ...
    const int lgpr = dim / 32;                                               // line groups per row
    const int n_groups = n_my * lgpr;
    const float *qb = qs + l31 * qstride + half * 16;

    TopK best;
    best.init();
    f32x4 buf[KND_NB][4];
    f32x4 xn[4];
    // issue side: (tile, line group) of the next load
    int it = 0, ij = 0;
    const float *irow = nullptr;
    auto tile_row = [&](int t) __attribute__((always_inline)) {
        int64_t n = stripe_begin + (int64_t)(wave + t * NW) * 32 + l31;
        n = n < n_rows ? n : n_rows - 1;
        return index + n * dim + half * 16;
    };
    auto issue = [&](int slot) __attribute__((always_inline)) {
        if (ij == 0) irow = tile_row(it);
        const float *p = irow + ij * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) buf[slot][t] = *reinterpret_cast<const f32x4 *>(p + 4 * t);
        if (++ij == lgpr) { ij = 0; ++it; }
    };
    auto load_norms = [&](int t) __attribute__((always_inline)) {
        const int64_t n_base = stripe_begin + (int64_t)(wave + t * NW) * 32;
        if (n_base + 32 <= stripe_end) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xn[g] = *reinterpret_cast<const f32x4 *>(norms + n_base + 8 * g + 4 * half);
        } else {   // the stripe's last, partial tile: rows past the end can never be selected
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t n = n_base + 8 * g + 4 * half + e;
                    xn[g][e] = n < stripe_end ? norms[n] : INFINITY;
                }
        }
    };

#pragma unroll
    for (int s = 0; s < KND_NB; ++s)
        if (s < n_groups) issue(s);
    if (n_groups > 0) load_norms(0);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int cj = 0, ct = 0;   // consume side
    for (int g0 = 0; g0 < n_groups; g0 += KND_NB) {
#pragma unroll
        for (int s = 0; s < KND_NB; ++s) {
            const int g = g0 + s;
            if (g < n_groups) {   // wave-uniform
                f32x4 b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x4 *>(qb + cj * 32 + 4 * t);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(buf[s][t][e], b[t][e], acc);
                if (g + KND_NB < n_groups) issue(s);
                if (++cj == lgpr) {
                    const int n_base = (int)(stripe_begin + (int64_t)(wave + ct * NW) * 32);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        best.insert(fmaf(-2.f, acc[r], xn[r >> 2][r & 3]), n_base + mfma32_row(r, lane));
                        acc[r] = 0.f;
                    }
                    cj = 0;
                    ++ct;
                    if (ct < n_my) load_norms(ct);
                }
            }
        }
    }

    // merge the block's 2 NW (wave, lane-half) lists per query in LDS -> ONE sorted list per (query, stripe)
    __syncthreads();
    constexpr int NC = 2 * NW * KNN_K;           // candidates per query
    float *md = qs;                              // [32][NC]   (dim >= 256: the query tile is at least this large)
    int *mi = reinterpret_cast<int *>(qs + KND_BQ * NC);
#pragma unroll
    for (int i = 0; i < KNN_K; ++i) {
        md[l31 * NC + (wave * 2 + half) * KNN_K + i] = best.d[i];
        mi[l31 * NC + (wave * 2 + half) * KNN_K + i] = best.id[i];
    }
    __syncthreads();
    if (tid < KND_BQ) {
        TopK m;
        m.init();
        for (int c = 0; c < NC; ++c) {
            const float v = md[tid * NC + c];
            const int n = mi[tid * NC + c];
            if (n < 0) continue;
            if (v < m.d[KNN_K - 1] || (v == m.d[KNN_K - 1] && (unsigned)n < (unsigned)m.id[KNN_K - 1])) {
#pragma unroll
                for (int p = KNN_K - 1; p >= 1; --p) {
                    const bool shift = v < m.d[p - 1] || (v == m.d[p - 1] && (unsigned)n < (unsigned)m.id[p - 1]);
                    const bool here = v < m.d[p] || (v == m.d[p] && (unsigned)n < (unsigned)m.id[p]);
                    const float nd = shift ? m.d[p - 1] : (here ? v : m.d[p]);
                    const int ni = shift ? m.id[p - 1] : (here ? n : m.id[p]);
                    m.d[p] = nd;
                    m.id[p] = ni;
                }
                if (v < m.d[0] || (v == m.d[0] && (unsigned)n < (unsigned)m.id[0])) { m.d[0] = v; m.id[0] = n; }
            }
        }
        const int64_t q = q0 + tid;
        if (q < n_queries) {
            float *pd = part_d + (q * n_slots + blockIdx.y) * KNN_K;
            int *pi = part_id + (q * n_slots + blockIdx.y) * KNN_K;
#pragma unroll
            for (int i = 0; i < KNN_K; ++i) { pd[i] = m.d[i]; pi[i] = m.id[i]; }
        }
    }
}

__global__ void __launch_bounds__(256)
knn_merge_kernel(const float *__restrict__ part_d, const int *__restrict__ part_id, int n_slots,
                 const float *__restrict__ queries, int dim, float *__restrict__ out_d2,
                 int64_t *__restrict__ out_ids) {
    __shared__ float wd[4 * KNN_K];
    __shared__ int wi[4 * KNN_K];
    __shared__ float qn_s[4];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ||q||^2
    float qn = 0.f;
    for (int i = tid; i < dim; i += 256) {
        const float v = queries[q * dim + i];
        qn = fmaf(v, v, qn);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
    if (lane == 0) qn_s[wave] = qn;

    float d[KNN_K];
    int id[KNN_K];
#pragma unroll
    for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
    const int total = n_slots * KNN_K;
    for (int c = tid; c < total; c += 256) {
        const int n = part_id[q * total + c];
        if (n >= 0) list_insert(d, id, part_d[q * total + c], n);
    }
    wave_top8(d, id, lane, &wd[wave * KNN_K], &wi[wave * KNN_K]);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
        if (lane < 4 * KNN_K) { d[0] = wd[lane]; id[0] = wi[lane]; }
        float fd[KNN_K];
        int fi[KNN_K];
        wave_top8(d, id, lane, fd, fi);
        if (lane == 0) {
            const float qq = (qn_s[0] + qn_s[1]) + (qn_s[2] + qn_s[3]);
#pragma unroll
            for (int r = 0; r < KNN_K; ++r) {
                out_d2[q * KNN_K + r] = fmaxf(fd[r] + qq, 0.f);
                out_ids[q * KNN_K + r] = (fi[r] == 0x7fffffff) ? -1 : (int64_t)fi[r];
            }
        }
    }
}

// pipeline.py:500-506
__global__ void __launch_bounds__(256)
knn_blend_kernel(const float *__restrict__ index, int dim, const float *__restrict__ feats,
                 const float *__restrict__ d2, const int64_t *__restrict__ ids, int k, float index_rate,
                 float *__restrict__ out) {
    const int64_t q = blockIdx.x;
    __shared__ float w_s[KNN_K];
    __shared__ int64_t id_s[KNN_K];
    if (threadIdx.x == 0) {
        float w[KNN_K];
        float sum = 0.f;
        for (int i = 0; i < k; ++i) {
            const int64_t id = ids[q * k + i];
            const float inv = 1.f / d2[q * k + i];  // no guard for d2 == 0, as in the reference
            w[i] = id >= 0 ? inv * inv : 0.f;       // an index with fewer than k rows: the missing neighbours carry no weight
            sum += w[i];
            id_s[i] = id >= 0 ? id : 0;
        }
        for (int i = 0; i < k; ++i) w_s[i] = w[i] / sum;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < k; ++i) acc += index[id_s[i] * dim + c] * w_s[i];
        out[q * dim + c] = acc * index_rate + (1.f - index_rate) * feats[q * dim + c];
    }
}

constexpr int64_t KNN_STREAM_MAX_Q = 64;   // at most two 32-query column tiles take the streaming kernel

constexpr int KND_WAVES = 8;
static size_t knd_lds_bytes(int dim) { return (size_t)KND_BQ * (dim + 4) * sizeof(float); }

struct KnnPlan {
    bool stream;
    bool direct = false;   // streaming regime, second form (knn_direct_kernel)
    int stripes, slots_per_stripe, q_tile;
    int64_t stripe_rows;
    int n_slots() const { return stripes * slots_per_stripe; }
};

static KnnPlan knn_plan(int64_t n_rows, int64_t n_queries, int dim = 768) {
    static const int force_batch = knob("RVC_KNN_NO_STREAM", 0);
    KnnPlan p;
    p.stream = n_queries <= KNN_STREAM_MAX_Q && !force_batch;
    p.q_tile = p.stream ? KNS_BQ : KNN_BQ;
    p.slots_per_stripe = p.stream ? KNS_SLOTS_PER_STRIPE : KNN_SLOTS_PER_STRIPE;
    const int64_t q_tiles = ceil_div(n_queries, p.q_tile);
    // batch kernel: >= 6 blocks per CU, stripes of >= 4 row tiles; streaming kernel: 2 blocks per CU (longer stripes
    // amortise the per-block prologue and merge: 4.0 TB/s vs 3.4 TB/s at 8 per CU on a 2 M-row index)
    static const int bpc_env = knob("RVC_KNN_STREAM_BPC", 0);
    static const int old_stream = knob("RVC_KNN_STREAM_OLD", 0);
    p.direct = p.stream && !old_stream && dim >= 256 && dim % 32 == 0;
    // direct form: the query tile's LDS footprint decides how many blocks share a CU (1 at dim 768, 4 at dim 256)
    int bpc = p.direct ? (int)((size_t)(160 * 1024) / knd_lds_bytes(dim)) : 2;
    if (bpc > 4) bpc = 4;
    if (bpc_env) bpc = bpc_env;
    int64_t want = ceil_div(256 * (p.stream ? bpc : 6), q_tiles);
    const int64_t max_stripes = ceil_div(n_rows, (int64_t)(p.stream ? 1 : 4) * KNN_BN);
    if (want > max_stripes) want = max_stripes;
    if (want < 1) want = 1;
    p.stripes = (int)want;
    const int64_t unit = p.direct ? 32 * KND_WAVES : KNN_BN;
    p.stripe_rows = ceil_div(ceil_div(n_rows, p.stripes), unit) * unit;
    return p;
}

// knn_screen.hip
bool knn_screen_applicable(int64_t n_rows, int64_t n_queries, int dim);
size_t knn_screen_workspace_bytes(int64_t n_rows, int64_t n_queries, int dim);
int knn_finalize_launch(const float *index, int64_t n_rows, int dim, const float *queries, int64_t n_queries, const int *cand_id,
                        const int *cand_cnt, const float *cand_s, const void *aux_dev, int cap, float *out_d2, int64_t *out_ids,
                        int *n_exact_scans, hipStream_t stream);
int knn_screened_search(const float *index, const void *aux_dev, int64_t n_rows, int dim, const float *queries, int64_t n_queries,
                        float *out_d2, int64_t *out_ids, void *workspace, hipStream_t stream);

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_knn_workspace_bytes(int64_t n_rows, int64_t n_queries, int dim, int k, size_t *bytes) {
    if (!bytes || k != KNN_K || n_rows <= 0 || n_queries <= 0 || dim <= 0) return fail("rvc_knn_workspace_bytes: bad argument (k must be 8)");
    const size_t slots = (size_t)knn_plan(n_rows, n_queries, dim).n_slots();
    size_t need = align_up((size_t)n_queries * slots * KNN_K * sizeof(float), 256) +
                  align_up((size_t)n_queries * slots * KNN_K * sizeof(int), 256);
    if (dim % 256 == 0 && dim <= 1024) {   // the fp16-screened regime may be chosen at search time (rvc_knn_set_mode)
        const size_t screened = knn_screen_workspace_bytes(n_rows, n_queries, dim);
        if (screened > need) need = screened;
    }
    *bytes = need;
    return 0;
}

extern "C" int rvc_knn_search(const float *index_dev, const void *aux_dev, int64_t n_rows, int dim,
                              const float *queries_dev, int64_t n_queries, int k, float *out_d2_dev,
                              int64_t *out_ids_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    const float *norms_dev = (const float *)aux_dev;   // the aux blob starts with ||x||^2 per row (rvc_knn_index_build)
    if (k != KNN_K) return fail("rvc_knn_search: k must be 8 (pipeline.py:499), got %d", k);
    if (dim <= 0 || dim % KNN_KC) return fail("rvc_knn_search: dim must be a multiple of %d, got %d", KNN_KC, dim);
    if (n_rows >= (int64_t)1 << 31) return fail("rvc_knn_search: more than 2^31 rows");
    if (!index_dev || !norms_dev || !queries_dev || !out_d2_dev || !out_ids_dev || !workspace_dev)
        return fail("rvc_knn_search: null pointer");
    if (n_queries == 0) return 0;
    size_t need = 0;
    if (rvc_knn_workspace_bytes(n_rows, n_queries, dim, k, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_knn_search: workspace too small (%zu < %zu)", workspace_bytes, need);
    if (knn_screen_applicable(n_rows, n_queries, dim))
        return knn_screened_search(index_dev, aux_dev, n_rows, dim, queries_dev, n_queries, out_d2_dev, out_ids_dev, workspace_dev,
                                   (hipStream_t)stream);
    const KnnPlan plan = knn_plan(n_rows, n_queries, dim);
    const int n_slots = plan.n_slots();
    float *part_d = (float *)workspace_dev;
    int *part_id = (int *)((char *)workspace_dev + align_up((size_t)n_queries * n_slots * KNN_K * sizeof(float), 256));
    dim3 grid((unsigned)ceil_div(n_queries, plan.q_tile), (unsigned)plan.stripes);
    static const int pf = knob("RVC_KNN_STREAM_PF", 3);
    if (plan.direct) {
        const size_t lds = knd_lds_bytes(dim);
        static std::mutex lds_mutex;               // several host threads search concurrently (convert_batch)
        static size_t lds_set = 0;
        {
            std::lock_guard<std::mutex> guard(lds_mutex);
            if (lds > lds_set) {
                hipError_t e = hipFuncSetAttribute((const void *)knn_direct_kernel<KND_WAVES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return fail("rvc_knn_search: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
                lds_set = lds;
            }
        }
        static const int waves_env = knob("RVC_KNN_DIRECT_WAVES", 0);
        if (waves_env == 84) {   // RVC_KNN_DIRECT_WAVES=84: 4 line groups in flight (+1.5 % on a 2 M-row index)
            static std::once_flag set84;
            std::call_once(set84, [lds] {
                (void)hipFuncSetAttribute((const void *)knn_direct_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            });
            hipLaunchKernelGGL((knn_direct_kernel<8, 4>), grid, dim3(512), lds, (hipStream_t)stream, index_dev, norms_dev,
                               n_rows, dim, queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
        } else
        hipLaunchKernelGGL((knn_direct_kernel<KND_WAVES, 3>), grid, dim3(KND_WAVES * 64), lds, (hipStream_t)stream, index_dev, norms_dev,
                           n_rows, dim, queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
    } else if (plan.stream && pf == 2)
        hipLaunchKernelGGL(knn_stream_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, index_dev, norms_dev, n_rows, dim,
                           queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
    else if (plan.stream && pf == 4)
        hipLaunchKernelGGL(knn_stream_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, index_dev, norms_dev, n_rows, dim,
                           queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
    else if (plan.stream)
        hipLaunchKernelGGL(knn_stream_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, index_dev, norms_dev, n_rows, dim,
                           queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
    else
        hipLaunchKernelGGL(knn_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, index_dev, norms_dev, n_rows, dim,
                           queries_dev, n_queries, plan.stripe_rows, part_d, part_id, n_slots);
    RVC_LAUNCH_CHECK();
    // the per-slot lists (best 8 of disjoint row subsets by the GEMM-form score) are candidates; the result is ranked by
    // the exact direct-difference distance, the same final step as the screened regime (knn_screen.hip)
    if (dim % 256 == 0 && dim <= 1024)
        return knn_finalize_launch(index_dev, n_rows, dim, queries_dev, n_queries, part_id, nullptr, part_d, aux_dev, n_slots * KNN_K,
                                   out_d2_dev, out_ids_dev, nullptr, (hipStream_t)stream);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)n_queries), dim3(256), 0, (hipStream_t)stream, part_d, part_id,
                       n_slots, queries_dev, dim, out_d2_dev, out_ids_dev);
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_knn_blend(const float *index_dev, int dim, const float *feats_dev, const float *d2_dev,
                             const int64_t *ids_dev, int64_t n_queries, int k, float index_rate, float *out_dev,
                             void *stream) {
    if (k != KNN_K) return fail("rvc_knn_blend: k must be 8");
    if (!index_dev || !feats_dev || !d2_dev || !ids_dev || !out_dev) return fail("rvc_knn_blend: null pointer");
    if (n_queries == 0) return 0;
    hipLaunchKernelGGL(knn_blend_kernel, dim3((unsigned)n_queries), dim3(256), 0, (hipStream_t)stream, index_dev, dim,
                       feats_dev, d2_dev, ids_dev, k, index_rate, out_dev);
    RVC_LAUNCH_CHECK();
    return 0;
}
