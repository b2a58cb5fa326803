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
#include "common.h"

namespace rvc {

constexpr int KNN_K = 8;
constexpr int KNN_BQ = 128;    // queries per block
constexpr int KNN_BN = 128;    // index rows per inner tile
constexpr int KNN_KC = 32;     // floats of D per staged chunk
constexpr int KNN_LDS_STRIDE = KNN_KC + 1;
constexpr int KNN_SLOTS_PER_STRIPE = 4;  // 2 wave-rows x 2 lane-halves

struct TopK {
    float d[KNN_K];
    int id[KNN_K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = -1; }
    }
    // keep ascending order; equal distances keep the earlier (lower id) entry first
    __device__ __forceinline__ void insert(float v, int n) {
        if (v < d[KNN_K - 1]) {
#pragma unroll
            for (int p = KNN_K - 1; p >= 1; --p) {
                const bool shift = v < d[p - 1];
                const bool here = v < d[p];
                const float nd = shift ? d[p - 1] : (here ? v : d[p]);
                const int ni = shift ? id[p - 1] : (here ? n : id[p]);
                d[p] = nd;
                id[p] = ni;
            }
            if (v < d[0]) { d[0] = v; id[0] = n; }
        }
    }
};

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

__global__ void __launch_bounds__(256)
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

__device__ __forceinline__ bool cand_less(float da, int ia, float db, int ib) {
    return (da < db) || (da == db && (unsigned)ia < (unsigned)ib);
}

// one wave per query: merge n_slots sorted lists of 8 into the final ascending top-8
__global__ void __launch_bounds__(64)
knn_merge_kernel(const float *__restrict__ part_d, const int *__restrict__ part_id, int n_slots,
                 const float *__restrict__ queries, int dim, float *__restrict__ out_d2,
                 int64_t *__restrict__ out_ids) {
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    // ||q||^2
    float qn = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float v = queries[q * dim + i];
        qn = fmaf(v, v, qn);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);

    // per-lane sorted top-8 over this lane's share of the candidates, ordered by (d, id)
    float d[KNN_K];
    int id[KNN_K];
#pragma unroll
    for (int i = 0; i < KNN_K; ++i) { d[i] = INFINITY; id[i] = 0x7fffffff; }
    const int total = n_slots * KNN_K;
    for (int c = lane; c < total; c += 64) {
        const float v = part_d[q * total + c];
        const int n = part_id[q * total + c];
        if (n < 0) continue;
        if (cand_less(v, n, d[KNN_K - 1], id[KNN_K - 1])) {
#pragma unroll
            for (int p = KNN_K - 1; p >= 1; --p) {
                const bool shift = cand_less(v, n, d[p - 1], id[p - 1]);
                const bool here = cand_less(v, n, d[p], id[p]);
                const float nd = shift ? d[p - 1] : (here ? v : d[p]);
                const int ni = shift ? id[p - 1] : (here ? n : id[p]);
                d[p] = nd;
                id[p] = ni;
            }
            if (cand_less(v, n, d[0], id[0])) { d[0] = v; id[0] = n; }
        }
    }
    // 8 rounds of wave-wide arg-min over the lane heads
    for (int round = 0; round < KNN_K; ++round) {
        float bd = d[0];
        int bi = id[0];
        int bl = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(bd, o);
            const int oi = __shfl_xor(bi, o);
            const int ol = __shfl_xor(bl, o);
            if (cand_less(od, oi, bd, bi) || (od == bd && oi == bi && ol < bl)) { bd = od; bi = oi; bl = ol; }
        }
        if (lane == 0) {
            out_d2[q * KNN_K + round] = fmaxf(bd + qn, 0.f);
            out_ids[q * KNN_K + round] = (bi == 0x7fffffff) ? -1 : (int64_t)bi;
        }
        if (lane == bl) {  // pop
#pragma unroll
            for (int p = 0; p < KNN_K - 1; ++p) { d[p] = d[p + 1]; id[p] = id[p + 1]; }
            d[KNN_K - 1] = INFINITY;
            id[KNN_K - 1] = 0x7fffffff;
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
            const float inv = 1.f / d2[q * k + i];  // no guard for d2 == 0, as in the reference
            w[i] = inv * inv;
            sum += w[i];
        }
        for (int i = 0; i < k; ++i) { w_s[i] = w[i] / sum; id_s[i] = ids[q * k + i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < k; ++i) acc += index[id_s[i] * dim + c] * w_s[i];
        out[q * dim + c] = acc * index_rate + (1.f - index_rate) * feats[q * dim + c];
    }
}

static int knn_stripes(int64_t n_rows, int64_t n_queries) {
    const int64_t q_tiles = ceil_div(n_queries, KNN_BQ);
    // aim for >= 6 blocks per CU's worth of work items, but never stripes shorter than 4 tiles
    int64_t want = ceil_div(256 * 6, q_tiles);
    const int64_t max_stripes = ceil_div(n_rows, (int64_t)4 * KNN_BN);
    if (want > max_stripes) want = max_stripes;
    if (want < 1) want = 1;
    return (int)want;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_knn_index_norms(const float *index_dev, int64_t n_rows, int dim, float *norms_dev,
                                   void *stream) {
    if (!index_dev || !norms_dev || n_rows <= 0 || dim <= 0 || dim % 4) return fail("rvc_knn_index_norms: bad argument");
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       index_dev, n_rows, dim, norms_dev);
    RVC_LAUNCH_CHECK();
    return 0;
}

extern "C" int rvc_knn_workspace_bytes(int64_t n_rows, int64_t n_queries, int k, size_t *bytes) {
    if (!bytes || k != KNN_K || n_rows <= 0 || n_queries <= 0) return fail("rvc_knn_workspace_bytes: bad argument (k must be 8)");
    const size_t slots = (size_t)knn_stripes(n_rows, n_queries) * KNN_SLOTS_PER_STRIPE;
    *bytes = align_up((size_t)n_queries * slots * KNN_K * sizeof(float), 256) +
             align_up((size_t)n_queries * slots * KNN_K * sizeof(int), 256);
    return 0;
}

extern "C" int rvc_knn_search(const float *index_dev, const float *norms_dev, int64_t n_rows, int dim,
                              const float *queries_dev, int64_t n_queries, int k, float *out_d2_dev,
                              int64_t *out_ids_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (k != KNN_K) return fail("rvc_knn_search: k must be 8 (pipeline.py:499), got %d", k);
    if (dim <= 0 || dim % KNN_KC) return fail("rvc_knn_search: dim must be a multiple of %d, got %d", KNN_KC, dim);
    if (n_rows >= (int64_t)1 << 31) return fail("rvc_knn_search: more than 2^31 rows");
    if (!index_dev || !norms_dev || !queries_dev || !out_d2_dev || !out_ids_dev || !workspace_dev)
        return fail("rvc_knn_search: null pointer");
    if (n_queries == 0) return 0;
    size_t need = 0;
    if (rvc_knn_workspace_bytes(n_rows, n_queries, k, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_knn_search: workspace too small (%zu < %zu)", workspace_bytes, need);
    const int stripes = knn_stripes(n_rows, n_queries);
    const int n_slots = stripes * KNN_SLOTS_PER_STRIPE;
    int64_t stripe_rows = ceil_div(ceil_div(n_rows, stripes), KNN_BN) * KNN_BN;
    float *part_d = (float *)workspace_dev;
    int *part_id = (int *)((char *)workspace_dev + align_up((size_t)n_queries * n_slots * KNN_K * sizeof(float), 256));
    dim3 grid((unsigned)ceil_div(n_queries, KNN_BQ), (unsigned)stripes);
    hipLaunchKernelGGL(knn_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, index_dev, norms_dev, n_rows, dim,
                       queries_dev, n_queries, stripe_rows, part_d, part_id, n_slots);
    RVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)n_queries), dim3(64), 0, (hipStream_t)stream, part_d, part_id,
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
