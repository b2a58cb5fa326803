// Shared device helpers of the retrieval kernels (knn.hip: exact fp32 search; knn_screen.hip: fp16-screened search).
#pragma once
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

__device__ __forceinline__ bool cand_less(float da, int ia, float db, int ib) {
    return (da < db) || (da == db && (unsigned)ia < (unsigned)ib);
}

// one block (4 waves) per query: merge n_slots sorted lists of 8 into the final ascending top-8
__device__ __forceinline__ void wave_top8(float (&d)[KNN_K], int (&id)[KNN_K], int lane, float *out_d, int *out_i) {
    // 8 rounds of wave-wide arg-min over the lane heads (lists sorted ascending by (d, id))
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
        if (lane == 0) { out_d[round] = bd; out_i[round] = bi; }
        if (lane == bl) {  // pop
#pragma unroll
            for (int p = 0; p < KNN_K - 1; ++p) { d[p] = d[p + 1]; id[p] = id[p + 1]; }
            d[KNN_K - 1] = INFINITY;
            id[KNN_K - 1] = 0x7fffffff;
        }
    }
}

__device__ __forceinline__ void list_insert(float (&d)[KNN_K], int (&id)[KNN_K], float v, int n) {
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


}  // namespace rvc
