// K7 -- fp32 softmax attention for HuBERT's 12 encoder layers (12 heads x 64, T = 1599 frames at BASELINE cfg 2),
// on the exact-fp32 matrix cores.  One wave owns 32 queries of one head and walks all keys in tiles of 32 with an
// online softmax; nothing is staged in LDS and no wave ever waits for another.
//
// Both GEMMs are run TRANSPOSED so that a lane owns one query:
//   S^T[key][query]  = K  Q^T :  MFMA "A" rows = keys, "B" columns = queries.  Lane (query j, half h) ends up with the
//                                scores of keys (r & 3) + 8 (r >> 2) + 4 h, r = 0..15, of its own query: the row max
//                                and row sum of the softmax are per-lane loops plus ONE exchange with lane j of the
//                                other half.
//   O^T[d][query]   += V^T P^T :  "A" rows = head dims, "B" columns = queries, k = keys.  A dot product does not care in
//                                which order k is walked as long as both operands agree, so the k-pair of MFMA step r
//                                is chosen as (key_r of half 0, key_r of half 1): the "B" operand of step r is then
//                                exactly the lane's own probability register p[r] -- the S^T accumulators feed the
//                                second GEMM without moving, and V rows are read as 128-byte row segments.
// The same freedom lets the K fragments be dwordx4 loads: lane (key i, half h) takes dims [8 jj + 4 h, +4) of its key
// row, and the query fragment (held in 32 registers for the whole kernel) follows the same order.
//
// Work per wave: 64 MFMAs (4096 cycles) per 32 keys; a cfg-2 layer is 600 independent waves (50 query tiles x 12 heads)
// for 1024 SIMDs, so a layer takes about one wave's time.  K/V of a head (2 x 409 KB) stay in L2.
#include "common.h"

namespace rvc {

constexpr int ATT_D = 64;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Software pipeline (per wave, one 32-key tile per step): the 32 score MFMAs of tile t+1 are issued BEFORE the softmax of
// tile t, which does not depend on them, so the scheduler can fill the matrix pipe's 64-cycle slots with the softmax
// VALU work; K fragments are loaded two tiles ahead, V fragments one step ahead of their use.  The running max is
// applied lazily: the accumulators are rescaled only when some query's max grew by more than 2^8 since the last
// rescale (after the first tiles: almost never), which is exact in real arithmetic and saves ~80 register moves and
// multiplies per tile.  Addresses are a uniform base plus per-lane 32-bit byte offsets computed once.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
attention_qkv_kernel(const float *__restrict__ qkv, float *__restrict__ out, int64_t T, int n_heads, float scale_log2e) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * 32;
    const int head = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int64_t rs = (int64_t)3 * n_heads * ATT_D;                 // floats between consecutive frames
    const float *qp = qkv + b * T * rs + (int64_t)head * ATT_D;
    const char *kp = reinterpret_cast<const char *>(qp + (int64_t)n_heads * ATT_D);
    const char *vp = reinterpret_cast<const char *>(qp + (int64_t)2 * n_heads * ATT_D);
    const uint32_t rs_b = (uint32_t)rs * 4u;                          // bytes per frame (host checks T * rs_b < 4 GB)
    const int nt = (int)((T + 31) / 32);
    const bool partial = (T & 31) != 0;

    // query fragment: lane (query j, half h) holds Q[q0 + j][8 jj + 4 h + e], pre-scaled by scale * log2(e)
    f32x4 qf[8];
    {
        const int64_t q = q0 + j < T ? q0 + j : T - 1;
        const float *p = qp + q * rs + 4 * h;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) qf[jj] = *reinterpret_cast<const f32x4 *>(p + 8 * jj) * scale_log2e;
    }
    // K fragment of tile t: lane (key i = j, half h) reads K[32 t + i][8 jj + 4 h + e]; the row is clamped (masked later)
    auto load_k = [&](f32x4 (&ka)[8], int t) __attribute__((always_inline)) {
        const int64_t key = (int64_t)t * 32 + j;
        const uint32_t off = (uint32_t)(key < T ? key : T - 1) * rs_b + 16u * h;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) ka[jj] = *reinterpret_cast<const f32x4 *>(kp + off + 32u * jj);
    };
    // V^T fragment of step r: lane (dim i = j, half h) reads V[32 t + key_r(h)][i] and [i + 32]
    uint32_t voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) voff[r] = (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h) * rs_b + 4u * j;
    auto load_v = [&](float (&va)[16][2], int t) __attribute__((always_inline)) {
        const char *vt = vp + (int64_t)t * 32 * rs_b;                // uniform
        if (t + 1 < nt || !partial) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                va[r][0] = *reinterpret_cast<const float *>(vt + voff[r]);
                va[r][1] = *reinterpret_cast<const float *>(vt + voff[r] + 128u);
            }
        } else {   // last, partial tile: rows past the end are clamped (their probability is exactly 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t key = (int64_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const uint32_t off = (uint32_t)(key < T ? key : T - 1) * rs_b + 4u * j;
                va[r][0] = *reinterpret_cast<const float *>(vp + off);
                va[r][1] = *reinterpret_cast<const float *>(vp + off + 128u);
            }
        }
    };
    auto scores = [&](const f32x4 (&ka)[8]) __attribute__((always_inline)) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = mfma32(ka[jj][e], qf[jj][e], s);
        return s;
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, l = 0.f;     // m: the max the accumulators are currently scaled by

    // one step: S(t+1) from k_next into s_next, softmax + PV of tile t from s_cur; k_cur is refilled with K(t+2)
    auto step = [&](int t, f32x4 (&k_cur)[8], const f32x4 (&k_next)[8], f32x16 &s_cur, f32x16 &s_next)
                    __attribute__((always_inline)) {
        float va[16][2];
        load_v(va, t);
        if (t + 2 < nt) load_k(k_cur, t + 2);
        if (t + 1 < nt) s_next = scores(k_next);
        if (t + 1 == nt && partial) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s_cur[r] = (int64_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < T ? s_cur[r] : -INFINITY;
        }
        float mloc = s_cur[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s_cur[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        if (__any(mloc > m + 8.f)) {              // wave-uniform; always taken on the first tile (m = -inf)
            const float m_new = fmaxf(m, mloc);   // finite: every tile holds at least one real key
            const float alpha = fast_exp2(m - m_new);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s_cur[r] = fast_exp2(s_cur[r] - m);
            psum += s_cur[r];
        }
        l += psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0 = mfma32(va[r][0], s_cur[r], o0);
            o1 = mfma32(va[r][1], s_cur[r], o1);
        }
    };

    f32x4 ka[8], kb[8];
    f32x16 sa, sb;
    load_k(ka, 0);
    if (nt > 1) load_k(kb, 1);
    sa = scores(ka);
    for (int t = 0; t < nt; t += 2) {
        step(t, ka, kb, sa, sb);
        if (t + 1 < nt) step(t + 1, kb, ka, sb, sa);
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (q0 + j < T) {
        float *op = out + (b * T + q0 + j) * (int64_t)n_heads * ATT_D + (int64_t)head * ATT_D + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v0 = {o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
            f32x4 v1 = {o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
            *reinterpret_cast<f32x4 *>(op + 8 * g) = v0 * inv;
            *reinterpret_cast<f32x4 *>(op + 32 + 8 * g) = v1 * inv;
        }
    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_attention_qkv_f32(const float *qkv_dev, float *out_dev, int batch, int64_t n_frames, int n_heads,
                                     int head_dim, float scale, void *stream) {
    if (!qkv_dev || !out_dev) return fail("rvc_attention_qkv_f32: null pointer");
    if (head_dim != ATT_D) return fail("rvc_attention_qkv_f32: head_dim must be %d, got %d", ATT_D, head_dim);
    if (batch <= 0 || n_heads <= 0 || n_frames < 0) return fail("rvc_attention_qkv_f32: bad shape");
    if (n_frames == 0) return 0;
    if ((int64_t)n_frames * 3 * n_heads * ATT_D * 4 >= ((int64_t)1 << 32))
        return fail("rvc_attention_qkv_f32: one batch element of %lld frames exceeds the 4 GB the kernel addresses", (long long)n_frames);
    dim3 grid((unsigned)ceil_div(n_frames, 32), (unsigned)n_heads, (unsigned)batch);
    hipLaunchKernelGGL(attention_qkv_kernel, grid, dim3(64), 0, (hipStream_t)stream, qkv_dev, out_dev, n_frames, n_heads,
                       scale * 1.4426950408889634f);
    RVC_LAUNCH_CHECK();
    return 0;
}
