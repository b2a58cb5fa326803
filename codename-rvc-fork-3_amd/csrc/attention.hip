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
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace rvc {

constexpr int ATT_W = 10;            // relative-position window of the TextEncoder (attentions.py:94-100)
constexpr int ATT_NREL = 2 * ATT_W + 1;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttParams {
    const float *qkv;      // [B][T][3][H][D]
    const float *emb_k;    // [2W+1][D] or null   (REL)
    const float *emb_v;    // [2W+1][D] or null   (REL)
    float *out;            // [B][T][H*D]                         (n_splits == 1)
    float *part_o;         // [B][H][n_splits][T][D]  unnormalised (n_splits > 1)
    float *part_ml;        // [B][H][n_splits][T][2]  running max (log2 domain), row sum
    int64_t T;
    int n_heads, n_splits, tiles_per_split;
    float scale_log2e;
};

// Software pipeline (per wave, one 32-key tile per step): the score MFMAs of tile t+1 are issued BEFORE the softmax of
// tile t, which does not depend on them, so the scheduler can fill the matrix pipe's 64-cycle slots with the softmax
// VALU work; K fragments are loaded two tiles ahead, V fragments one step ahead of their use.  The running max is
// applied lazily: the accumulators are rescaled only when some query's max grew by more than 2^8 since the last
// rescale (after the first tiles: almost never), which is exact in real arithmetic and saves ~80 register moves and
// multiplies per tile.  Addresses are a uniform base plus per-lane 32-bit byte offsets computed once.
//
// D = head dim (64: HuBERT, 96: TextEncoder).  blockIdx.z selects a contiguous range of key tiles ("split"): with few
// (query tile, head) pairs -- the TextEncoder has 100 x 2 -- the keys are cut so that ~1000 waves exist, each split
// writes its unnormalised accumulator + (max, sum) and attention_combine_kernel merges them.
//
// REL: the window-10 relative-position terms of VITS' MultiHeadAttention (attentions.py:115-141, 156-180):
//   score[i][j] += q_i . emb_k[j - i + W]   and   out_i += sum_r p[i][i + r - W] emb_v[r]     for |j - i| <= W.
// The 21 logits per query are one extra 32x32 MFMA tile (emb_k zero-padded to 32 rows against the query fragment),
// parked in LDS; the band of probabilities is written to LDS as the diagonal tiles pass and contracted with emb_v by
// 16 more MFMA steps at the end.  Only the split that owns the diagonal sees non-zero band entries.
template <int D, bool REL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
attention_qkv_kernel(const AttParams p) {
    constexpr int NJ = D / 8;       // dwordx4 pieces of a K/Q row per lane
    constexpr int ND = D / 32;      // output row tiles
    __shared__ float rel_s[REL ? 32 * 33 : 1];    // [query][r]  logits q . emb_k[r] (scaled, log2 domain)
    __shared__ float band_s[REL ? 32 * 33 : 1];   // [query][r]  p[query][query + r - W], scaled like the accumulators
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int64_t T = p.T;
    const int n_heads = p.n_heads;
    const int64_t q0 = (int64_t)blockIdx.x * 32;
    const int head = blockIdx.y % n_heads;
    const int64_t b = blockIdx.y / n_heads;
    const int split = blockIdx.z;
    const int64_t rs = (int64_t)3 * n_heads * D;                      // floats between consecutive frames
    const float *qp = p.qkv + b * T * rs + (int64_t)head * D;
    const char *kp = reinterpret_cast<const char *>(qp + (int64_t)n_heads * D);
    const char *vp = reinterpret_cast<const char *>(qp + (int64_t)2 * n_heads * D);
    const uint32_t rs_b = (uint32_t)rs * 4u;                          // bytes per frame (host checks T * rs_b < 4 GB)
    const int nt_all = (int)((T + 31) / 32);
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(nt_all, t_begin + p.tiles_per_split);
    const bool partial = (T & 31) != 0;

    // query fragment: lane (query j, half h) holds Q[q0 + j][8 jj + 4 h + e], pre-scaled by scale * log2(e)
    f32x4 qf[NJ];
    {
        const int64_t q = q0 + j < T ? q0 + j : T - 1;
        const float *ptr = qp + q * rs + 4 * h;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) qf[jj] = *reinterpret_cast<const f32x4 *>(ptr + 8 * jj) * p.scale_log2e;
    }
    // K fragment of tile t: lane (key i = j, half h) reads K[32 t + i][8 jj + 4 h + e]; the row is clamped (masked later)
    auto load_k = [&](f32x4 (&ka)[NJ], int t) __attribute__((always_inline)) {
        const int64_t key = (int64_t)t * 32 + j;
        const uint32_t off = (uint32_t)(key < T ? key : T - 1) * rs_b + 16u * h;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) ka[jj] = *reinterpret_cast<const f32x4 *>(kp + off + 32u * jj);
    };
    // V^T fragment of step r: lane (dim i = j, half h) reads V[32 t + key_r(h)][i + 32 dt]
    uint32_t voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) voff[r] = (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h) * rs_b + 4u * j;
    auto load_v = [&](float (&va)[16][ND], int t) __attribute__((always_inline)) {
        const char *vt = vp + (int64_t)t * 32 * rs_b;                // uniform
        if (t + 1 < nt_all || !partial) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) va[r][dt] = *reinterpret_cast<const float *>(vt + voff[r] + 128u * dt);
        } else {   // last, partial tile: rows past the end are clamped (their probability is exactly 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t key = (int64_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const uint32_t off = (uint32_t)(key < T ? key : T - 1) * rs_b + 4u * j;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) va[r][dt] = *reinterpret_cast<const float *>(vp + off + 128u * dt);
            }
        }
    };
    auto scores = [&](const f32x4 (&ka)[NJ]) __attribute__((always_inline)) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = mfma32(ka[jj][e], qf[jj][e], s);
        return s;
    };

    f32x16 o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f;     // m: the max the accumulators are currently scaled by

    if constexpr (REL) {
        // rel_s[query][r] = (scaled q) . emb_k[r]: one MFMA tile with emb_k (zero rows beyond 2W) as the "A" operand
        f32x4 ek[NJ];
        const float *eptr = p.emb_k + (int64_t)(j < ATT_NREL ? j : 0) * D + 4 * h;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            ek[jj] = *reinterpret_cast<const f32x4 *>(eptr + 8 * jj);
            if (j >= ATT_NREL) ek[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x16 rl = scores(ek);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            rel_s[j * 33 + rr] = rl[r];
            band_s[j * 33 + rr] = 0.f;
        }
        __syncthreads();   // single-wave block: orders the LDS writes before the reads below
    }

    // one step: S(t+1) from k_next into s_next, softmax + PV of tile t from s_cur; k_cur is refilled with K(t+2)
    auto step = [&](int t, f32x4 (&k_cur)[NJ], const f32x4 (&k_next)[NJ], f32x16 &s_cur, f32x16 &s_next)
                    __attribute__((always_inline)) {
        float va[16][ND];
        load_v(va, t);
        if (t + 2 < t_end) load_k(k_cur, t + 2);
        if (t + 1 < t_end) s_next = scores(k_next);
        // tiles whose keys can lie within W of one of this block's queries (wave-uniform)
        const bool near_diag = REL && (int64_t)t * 32 + 31 + ATT_W >= q0 && (int64_t)t * 32 <= q0 + 31 + ATT_W;
        int rr[16];
        if constexpr (REL) {
            if (near_diag) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    rr[r] = (int)((int64_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h - (q0 + j)) + ATT_W;
                    const bool in = rr[r] >= 0 && rr[r] < ATT_NREL;
                    const float add = rel_s[j * 33 + (in ? rr[r] : 0)];
                    s_cur[r] += in ? add : 0.f;
                }
            }
        }
        if (t + 1 == nt_all && partial) {
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
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            if constexpr (REL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) band_s[j * 33 + 2 * r + h] *= alpha;   // 32 entries per query, one half each
            }
            m = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s_cur[r] = fast_exp2(s_cur[r] - m);
            psum += s_cur[r];
        }
        l += psum;
        if constexpr (REL) {
            if (near_diag) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (rr[r] >= 0 && rr[r] < ATT_NREL) band_s[j * 33 + rr[r]] = s_cur[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] = mfma32(va[r][dt], s_cur[r], o[dt]);
    };

    if (t_begin < t_end) {
        f32x4 ka[NJ], kb[NJ];
        f32x16 sa, sb;
        load_k(ka, t_begin);
        if (t_begin + 1 < t_end) load_k(kb, t_begin + 1);
        sa = scores(ka);
        for (int t = t_begin; t < t_end; t += 2) {
            step(t, ka, kb, sa, sb);
            if (t + 1 < t_end) step(t + 1, kb, ka, sb, sa);
        }
    }
    if constexpr (REL) {
        // out += emb_v^T . band^T: k-pair of step s is (r = 2 s, r = 2 s + 1); rows r >= 2W+1 of the band are zero
        __syncthreads();
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int r = 2 * sidx + h;
            const float bval = band_s[j * 33 + r];
            const float *ev = p.emb_v + (int64_t)(r < ATT_NREL ? r : 0) * D + j;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] = mfma32(ev[32 * dt], bval, o[dt]);
        }
    }
    l += __shfl_xor(l, 32);
    if (q0 + j < T) {
        if (p.n_splits == 1) {
            const float inv = 1.f / l;
            float *op = p.out + (b * T + q0 + j) * (int64_t)n_heads * D + (int64_t)head * D + 4 * h;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(op + 32 * dt + 8 * g) = v * inv;
                }
        } else {
            const int64_t row = ((b * n_heads + head) * p.n_splits + split) * T + q0 + j;
            float *op = p.part_o + row * D + 4 * h;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(op + 32 * dt + 8 * g) = v;
                }
            if (h == 0) {
                p.part_ml[row * 2] = m;
                p.part_ml[row * 2 + 1] = l;
            }
        }
    }
}

// out[b][t][head][:] = sum_s 2^(m_s - m*) O_s / sum_s 2^(m_s - m*) l_s over the key splits (an empty split has m = -inf)
__global__ void __launch_bounds__(256)
attention_combine_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml, float *__restrict__ out,
                         int64_t T, int n_heads, int n_splits, int D) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;        // over (b, head, t, d / 4)
    const int d4n = D / 4;
    const int64_t total = (int64_t)gridDim.y * n_heads * T * d4n;
    (void)total;
    const int64_t b = blockIdx.y;
    if (idx >= (int64_t)n_heads * T * d4n) return;
    const int d4 = (int)(idx % d4n);
    const int64_t t = (idx / d4n) % T;
    const int head = (int)(idx / d4n / T);
    float mstar = -INFINITY;
    for (int s = 0; s < n_splits; ++s) mstar = fmaxf(mstar, part_ml[(((b * n_heads + head) * n_splits + s) * T + t) * 2]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    for (int s = 0; s < n_splits; ++s) {
        const int64_t row = ((b * n_heads + head) * n_splits + s) * T + t;
        const float wgt = fast_exp2(part_ml[row * 2] - mstar);
        lsum += wgt * part_ml[row * 2 + 1];
        acc += *reinterpret_cast<const f32x4 *>(part_o + row * D + 4 * d4) * wgt;
    }
    *reinterpret_cast<f32x4 *>(out + (b * T + t) * (int64_t)n_heads * D + (int64_t)head * D + 4 * d4) = acc * (1.f / lsum);
}

// =====================================================================================================================
// K7b -- the same attention (head dim 64, no relative-position terms: HuBERT's 12 layers) on the bf16 matrix cores with every
// fp32 operand split exactly into three bf16 (six products of order <= 2^-16, fp32 accumulate: gemmbf.hip's arithmetic).
// K7 is bound by the fp32 matrix instruction: 64 of them (4096 cycles) per 32 keys and wave.  Here a 32-key tile costs
// 24 + 24 v_mfma_f32_32x32x16_bf16 = 1536 cycles.
//   * K and V are split ONCE per layer by attention_pack_kv_kernel, straight into matrix-instruction fragments: per (head, key
//     tile) 24 KiB = K [k-step 4][split 3] + V^T [dim block 2][k-step 2][split 3] pieces of 1 KiB (lane l: 8 bf16).  K piece
//     (ks, sp): lane (key i, half h) holds K[32 t + i][16 ks + 8 h + e]; V^T piece (db, ks, sp): lane (dim i, half h) holds
//     V[32 t + key(h, 8 ks + e)][32 db + i] with key(h, r) = (r & 3) + 8 (r >> 2) + 4 h -- the order in which a lane of the score
//     accumulator holds ITS keys, so the probabilities feed the second GEMM without moving (K7's trick, 16 keys per step).
//   * A workgroup is 8 waves = 256 queries of one head and OWNS its CU (it requests the whole LDS like every kernel that issues
//     bf16 matrix instructions, common.h): the tile's fragments come HBM/L2 -> LDS by LDS-DMA once per workgroup (double
//     buffered, one barrier per tile) and the 8 waves read them from there; the keys are cut into splits so that ~256 workgroups
//     exist, merged by attention_combine_kernel.
//   * Q (pre-scaled) is split by the wave that owns the 32 queries, once; the probabilities (16 per lane and tile) by three
//     rounds of convert / subtract.
struct AttBfParams {
    const float *qkv;      // [B][T][3][H][64]
    const void *frag;      // [B][H][n_tiles][24][64 lanes][8 bf16]
    float *out, *part_o, *part_ml;
    int64_t T;
    int n_heads, n_splits, tiles_per_split;
    float scale_log2e;
};
constexpr int ABF_TILE_BYTES = 24 * 1024, ABF_NW = 8;
typedef __bf16 abf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 abf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float abf_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned abf_u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) *abf_lptr_t;

// eight fp32 values -> their three bf16 "digits" (exact: v = d0 + d1 + d2 up to 2^-24 relative), packed as fragment registers
__device__ __forceinline__ void abf_split8(const float (&v)[8], abf_u32x4 (&out)[3]) {
    abf_f32x2 r[4] = {abf_f32x2{v[0], v[1]}, abf_f32x2{v[2], v[3]}, abf_f32x2{v[4], v[5]}, abf_f32x2{v[6], v[7]}};
#pragma unroll
    for (int level = 0; level < 3; ++level)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(r[q], abf_bf16x2));
            out[level][q] = w;
            if (level < 2) r[q] = r[q] - abf_f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        }
}

// grid (n_tiles, B * H), 512 threads: waves 0-3 the K pieces of k-step `wave`, waves 4-7 the V^T pieces of (db, ks) = ((wave - 4) / 2, wave % 2)
__global__ void __launch_bounds__(512) attention_pack_kv_kernel(const float *__restrict__ qkv, unsigned char *__restrict__ frag, int64_t T, int n_heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int64_t tile = blockIdx.x, n_tiles = gridDim.x;
    const int head = blockIdx.y % n_heads;
    const int64_t b = blockIdx.y / n_heads;
    const int64_t rs = (int64_t)3 * n_heads * 64;
    const float *base = qkv + b * T * rs + (int64_t)head * 64;
    float v[8];
    int piece0;
    if (wave < 4) {
        const int64_t key = tile * 32 + i;
        const float *ptr = base + (int64_t)n_heads * 64 + (key < T ? key : 0) * rs + 16 * wave + 8 * h;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = key < T ? ptr[e] : 0.f;
        piece0 = wave * 3;
    } else {
        const int db = (wave - 4) >> 1, ks = wave & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * ks + e;
            const int64_t key = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            v[e] = key < T ? base[(int64_t)2 * n_heads * 64 + key * rs + 32 * db + i] : 0.f;
        }
        piece0 = 12 + (db * 2 + ks) * 3;
    }
    abf_u32x4 d[3];
    abf_split8(v, d);
    unsigned char *dst = frag + ((int64_t)blockIdx.y * n_tiles + tile) * ABF_TILE_BYTES + lane * 16;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) *reinterpret_cast<abf_u32x4 *>(dst + (piece0 + sp) * 1024) = d[sp];
}

// DBG (ablation build; wrong results): 1 no matrix instructions, 2 no exponentials / splits, 4 no fragment reads from LDS, 8 no DMA in the loop,
// 16 one tile only
template <int DBG = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
attention_bf_kernel(const AttBfParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char abf_smem[];   // K ring [2][12 KiB], V ring [2][12 KiB]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t T = p.T;
    const int n_heads = p.n_heads;
    const int64_t q0 = ((int64_t)blockIdx.x * ABF_NW + wave) * 32;
    const bool active = q0 < T;                                        // (the last workgroup of a head: some waves have no queries)
    const int head = blockIdx.y % n_heads;
    const int64_t b = blockIdx.y / n_heads;
    const int split = blockIdx.z;
    const int64_t rs = (int64_t)3 * n_heads * 64;
    const int nt_all = (int)((T + 31) / 32);
    const int t_begin = split * p.tiles_per_split;
    const int t_end = (DBG & 16) ? min(nt_all, t_begin + 1) : min(nt_all, t_begin + p.tiles_per_split);
    const bool partial = (T & 31) != 0;
    constexpr int ia6[6] = {0, 1, 0, 2, 1, 0}, ib6[6] = {2, 1, 1, 0, 0, 0};

    // a tile's 12 K pieces and 12 V^T pieces: HBM/L2 -> LDS, three per wave.  K(t + 1) and V(t) are needed in iteration t (the scores
    // of tile t + 1 are issued BEFORE the softmax of tile t, which does not depend on them: K7's software pipeline, so the matrix
    // pipe has work while the vector unit exponentiates): LDS = K ring [2][12 KiB] + V ring [2][12 KiB], requested one iteration
    // before they are read.  (Tried on top, all slower on the same box, tools/ablate_attention.sh: rings of three with the requests two
    // iterations ahead, 67 -> 72 us per layer; the two products' dependent chains issued alternately,
    // 52 -> 59 us; the softmax's vector work pinned piece by piece behind the next tile's score products with sched_barrier, 59.
    // Ablation of this form: no matrix instructions 24 us of 52-59, fixed cost of a workgroup ~10, exp + splits ~10, LDS reads ~10.)
    const int64_t n_frag = (int64_t)gridDim.y * nt_all * ABF_TILE_BYTES;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)p.frag, 0, (int)n_frag, 0x00020000);
    const int bh_base = (int)((int64_t)blockIdx.y * nt_all) * ABF_TILE_BYTES;
    constexpr int HALF = ABF_TILE_BYTES / 2;
    // pieces 0..11: K of tile tk -> K ring slot kslot; pieces 12..23: V of tile tv -> V ring slot vslot (a tile < 0: nothing)
    auto dma = [&](int tk, int kslot, int tv, int vslot) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int piece = wave + ABF_NW * n;                       // wave-uniform
            if (piece < 12) {
                if (tk >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (abf_lptr_t)(abf_smem + kslot * HALF + piece * 1024), 16, 16 * lane, bh_base + tk * ABF_TILE_BYTES + piece * 1024, 0, 0);
            } else {
                if (tv >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(frs, (abf_lptr_t)(abf_smem + 2 * HALF + vslot * HALF + (piece - 12) * 1024), 16, 16 * lane, bh_base + tv * ABF_TILE_BYTES + piece * 1024, 0, 0);
            }
        }
    };
    // the first tiles are requested before anything else (K(t), V(t) live in ring slot (t - t_begin) & 1), the query rows next: their
    // split runs under the requests' flight
    if (t_begin < t_end) {
        dma(t_begin, 0, t_begin, 0);
        dma(t_begin + 1 < t_end ? t_begin + 1 : -1, 1, -1, 0);
    }
    // query fragments: lane (query j, half h) holds Q[q0 + j][16 ks + 8 h + e], pre-scaled by scale * log2(e), as three bf16 digits
    abf_u32x4 qf[4][3];
    {
        const int64_t q = q0 + j < T ? q0 + j : T - 1;
        const float *ptr = p.qkv + b * T * rs + (int64_t)head * 64 + q * rs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(ptr + 16 * ks) * p.scale_log2e, hi = *reinterpret_cast<const f32x4 *>(ptr + 16 * ks + 4) * p.scale_log2e;
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            abf_split8(v, qf[ks]);
        }
    }
    // S^T[key][query] = K Q^T of the tile in K ring slot `kslot`: 4 k-steps x 6 products
    auto scores = [&](int kslot) __attribute__((always_inline)) {
        const unsigned char *kk = abf_smem + kslot * HALF + lane * 16;
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            abf_bf16x8 kf[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) if (!(DBG & 4)) kf[sp] = __builtin_bit_cast(abf_bf16x8, *reinterpret_cast<const abf_u32x4 *>(kk + (ks * 3 + sp) * 1024));
#pragma unroll
            for (int i = 0; i < 6; ++i) if (!(DBG & 1)) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ia6[i]], __builtin_bit_cast(abf_bf16x8, qf[ks][ib6[i]]), sc, 0, 0, 0);
        }
        return sc;
    };

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -INFINITY, l = 0.f;     // m: the max the accumulators are currently scaled by

    if (t_begin < t_end) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 s_cur, s_next;
        if (active) s_cur = scores(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave has read K slot 0: it may take tile t_begin + 2
        for (int t = t_begin; t < t_end; ++t) {
            const int kb = (t - t_begin) & 1;                            // K(t) was in K slot kb, K(t + 1) is in kb ^ 1; V(t) is in V slot kb
            if (!(DBG & 8)) dma(t + 2 < t_end ? t + 2 : -1, kb, t + 1 < t_end ? t + 1 : -1, kb ^ 1);
            if (active) {
                if (t + 1 < t_end) s_next = scores(kb ^ 1);               // (independent of the softmax below: the scheduler may overlap them)
                f32x16 &sc = s_cur;
                if (t + 1 == nt_all && partial) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = (int64_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < T ? sc[r] : -INFINITY;
                }
                // ---- online softmax (K7's, lazy rescale) ---------------------------------------------------------------------
                float mloc = sc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, sc[r]);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
                if (__any(mloc > m + 8.f)) {              // wave-uniform; always taken on the first tile (m = -inf)
                    const float m_new = fmaxf(m, mloc);   // finite: every tile holds at least one real key
                    const float alpha = fast_exp2(m - m_new);
                    l *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                    m = m_new;
                }
                float psum = 0.f;
                if (!(DBG & 2)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sc[r] = fast_exp2(sc[r] - m);
                        psum += sc[r];
                    }
                }
                l += psum;
                // ---- O^T[dim][query] += V^T P^T: the lane's own probabilities are the "B" operand, 8 keys per k-step and half ----
                abf_u32x4 pf[2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float v[8] = {sc[8 * ks], sc[8 * ks + 1], sc[8 * ks + 2], sc[8 * ks + 3], sc[8 * ks + 4], sc[8 * ks + 5], sc[8 * ks + 6], sc[8 * ks + 7]};
                    if (!(DBG & 2)) abf_split8(v, pf[ks]);
                }
                const unsigned char *vv = abf_smem + 2 * HALF + kb * HALF + lane * 16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        abf_bf16x8 vf[3];
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp)
                            if (!(DBG & 4)) vf[sp] = __builtin_bit_cast(abf_bf16x8, *reinterpret_cast<const abf_u32x4 *>(vv + ((db * 2 + ks) * 3 + sp) * 1024));
#pragma unroll
                        for (int i = 0; i < 6; ++i)
                            if (!(DBG & 1)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ia6[i]], __builtin_bit_cast(abf_bf16x8, pf[ks][ib6[i]]), o[db], 0, 0, 0);
                    }
                s_cur = s_next;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the requested pieces have landed; this iteration's LDS reads are done
            __builtin_amdgcn_s_barrier();
        }
    }
    l += __shfl_xor(l, 32);
    if (active && q0 + j < T) {
        if (p.n_splits == 1) {
            const float inv = 1.f / l;
            float *op = p.out + (b * T + q0 + j) * (int64_t)n_heads * 64 + (int64_t)head * 64 + 4 * h;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(op + 32 * db + 8 * g) = v * inv;
                }
        } else {
            const int64_t row = ((b * n_heads + head) * p.n_splits + split) * T + q0 + j;
            float *op = p.part_o + row * 64 + 4 * h;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(op + 32 * db + 8 * g) = v;
                }
            if (h == 0) {
                p.part_ml[row * 2] = m;
                p.part_ml[row * 2 + 1] = l;
            }
        }
    }
}

// HuBERT's attention takes K7b unless the ablation build's RVC_ATT_BF=0 asks for K7
static bool att_bf_enabled() {
    static const int on = knob("RVC_ATT_BF", 1);
    return on != 0;
}
// key splits of K7b: one 8-wave workgroup per (256 queries, head, split) and CU
static int choose_splits_bf(int64_t n_frames, int n_heads, int batch) {
    const int64_t nt = ceil_div(n_frames, 32);
    const int64_t groups = ceil_div(n_frames, 32 * ABF_NW) * n_heads * batch;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 8 && s <= nt; ++s) {
        const double cost = (double)ceil_div(groups * s, 256) / s + 0.02 * s;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}
static size_t att_bf_workspace(int batch, int64_t n_frames, int n_heads, size_t *off_ml, size_t *off_frag) {
    const int splits = choose_splits_bf(n_frames, n_heads, batch);
    const size_t rows = splits == 1 ? 0 : (size_t)batch * n_heads * splits * (size_t)n_frames;
    const size_t o_bytes = align_up(rows * 64 * sizeof(float), 256), ml_bytes = align_up(rows * 2 * sizeof(float), 256);
    if (off_ml) *off_ml = o_bytes;
    if (off_frag) *off_frag = o_bytes + ml_bytes;
    return o_bytes + ml_bytes + align_up((size_t)batch * n_heads * (size_t)ceil_div(n_frames, 32) * ABF_TILE_BYTES, 256);
}

// key splits: the kernel runs one wave per (query tile, head, split); a layer takes ceil(waves / 1024 SIMDs) rounds of
// 1/splits of the keys each.  Pick the split count that minimises rounds / splits (fewer splits on a tie: each one
// re-loads the query fragment and adds a partial to combine).
static int choose_splits(int64_t n_frames, int n_heads, int batch) {
    static const int forced = knob("RVC_ATT_SPLITS", 0);
    const int64_t nt = ceil_div(n_frames, 32);
    const int64_t pairs = nt * n_heads * batch;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 8 && s <= nt; ++s) {
        const double cost = (double)ceil_div(pairs * s, 1024) / s + 0.02 * s;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    if (forced > 0 && forced <= nt) best = forced;
    return best;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_attention_workspace_bytes(int batch, int64_t n_frames, int n_heads, int head_dim, size_t *bytes) {
    if (!bytes || batch <= 0 || n_heads <= 0 || n_frames < 0 || head_dim <= 0) return fail("rvc_attention_workspace_bytes: bad argument");
    const int splits = choose_splits(n_frames, n_heads, batch);
    const size_t rows = (size_t)batch * n_heads * splits * (size_t)n_frames;
    *bytes = splits == 1 ? 256 : align_up(rows * head_dim * sizeof(float), 256) + align_up(rows * 2 * sizeof(float), 256);
    if (head_dim == 64 && att_bf_enabled()) {   // K7b: its own partials + the K / V fragment slab (the larger of the two kernels' needs)
        const size_t bf = att_bf_workspace(batch, n_frames, n_heads, nullptr, nullptr);
        if (bf > *bytes) *bytes = bf;
    }
    return 0;
}

template <int D, bool REL>
static int launch_attention(AttParams &p, int batch, hipStream_t stream) {
    dim3 grid((unsigned)ceil_div(p.T, 32), (unsigned)(p.n_heads * batch), (unsigned)p.n_splits);
    hipLaunchKernelGGL((attention_qkv_kernel<D, REL>), grid, dim3(64), 0, stream, p);
    RVC_LAUNCH_CHECK();
    if (p.n_splits > 1) {
        const int64_t work = (int64_t)p.n_heads * p.T * (D / 4);
        hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)ceil_div(work, 256), (unsigned)batch), dim3(256), 0, stream,
                           p.part_o, p.part_ml, p.out, p.T, p.n_heads, p.n_splits, D);
        RVC_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int rvc_attention_qkv_f32(const float *qkv_dev, const float *emb_rel_k_dev, const float *emb_rel_v_dev,
                                     float *out_dev, int batch, int64_t n_frames, int n_heads, int head_dim, float scale,
                                     void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!qkv_dev || !out_dev) return fail("rvc_attention_qkv_f32: null pointer");
    if (head_dim != 64 && head_dim != 96) return fail("rvc_attention_qkv_f32: head_dim must be 64 or 96, got %d", head_dim);
    if ((emb_rel_k_dev == nullptr) != (emb_rel_v_dev == nullptr)) return fail("rvc_attention_qkv_f32: emb_rel_k and emb_rel_v go together");
    if (batch <= 0 || n_heads <= 0 || n_frames < 0) return fail("rvc_attention_qkv_f32: bad shape");
    if (n_frames == 0) return 0;
    if ((int64_t)n_frames * 3 * n_heads * head_dim * 4 >= ((int64_t)1 << 32))
        return fail("rvc_attention_qkv_f32: one batch element of %lld frames exceeds the 4 GB the kernel addresses", (long long)n_frames);
    if (head_dim == 64 && !emb_rel_k_dev && att_bf_enabled() && (int64_t)batch * n_heads * ceil_div(n_frames, 32) * ABF_TILE_BYTES < ((int64_t)1 << 31)) {
        // K7b: split K / V into fragments, then one CU-owning workgroup per (256 queries, head, key split)
        size_t off_ml = 0, off_frag = 0;
        const size_t need = att_bf_workspace(batch, n_frames, n_heads, &off_ml, &off_frag);
        if (!workspace_dev || workspace_bytes < need) return fail("rvc_attention_qkv_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
        AttBfParams q;
        q.qkv = qkv_dev; q.out = out_dev; q.T = n_frames; q.n_heads = n_heads; q.scale_log2e = scale * 1.4426950408889634f;
        q.n_splits = choose_splits_bf(n_frames, n_heads, batch);
        q.tiles_per_split = (int)ceil_div(ceil_div(n_frames, 32), q.n_splits);
        q.part_o = (float *)workspace_dev; q.part_ml = (float *)((char *)workspace_dev + off_ml); q.frag = (char *)workspace_dev + off_frag;
        hipStream_t st = (hipStream_t)stream;
        static std::once_flag once;
        static hipError_t err = hipSuccess;
        std::call_once(once, [] {
            err = hipFuncSetAttribute((const void *)attention_bf_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
#ifdef RVC_ABLATE
            hipFuncSetAttribute((const void *)attention_bf_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
            hipFuncSetAttribute((const void *)attention_bf_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
            hipFuncSetAttribute((const void *)attention_bf_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
            hipFuncSetAttribute((const void *)attention_bf_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
            hipFuncSetAttribute((const void *)attention_bf_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
            hipFuncSetAttribute((const void *)attention_bf_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
#endif
        });
        if (err != hipSuccess) return fail("rvc_attention_qkv_f32: cannot reserve the CU's LDS: %s", hipGetErrorString(err));
        const unsigned n_tiles = (unsigned)ceil_div(n_frames, 32);
        hipLaunchKernelGGL(attention_pack_kv_kernel, dim3(n_tiles, (unsigned)(n_heads * batch)), dim3(512), 0, st, qkv_dev, (unsigned char *)q.frag, q.T, n_heads);
        RVC_LAUNCH_CHECK();
        const dim3 bgrid((unsigned)ceil_div(n_frames, 32 * ABF_NW), (unsigned)(n_heads * batch), (unsigned)q.n_splits);
        int dbg = 0;
#ifdef RVC_ABLATE
        static const int dbg_knob = knob("RVC_ATT_DBG", 0);
        dbg = dbg_knob;
#endif
        switch (dbg) {
#ifdef RVC_ABLATE
            case 1: hipLaunchKernelGGL(attention_bf_kernel<1>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
            case 2: hipLaunchKernelGGL(attention_bf_kernel<2>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
            case 4: hipLaunchKernelGGL(attention_bf_kernel<4>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
            case 8: hipLaunchKernelGGL(attention_bf_kernel<8>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
            case 16: hipLaunchKernelGGL(attention_bf_kernel<16>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
            case 7: hipLaunchKernelGGL(attention_bf_kernel<7>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;
#endif
            default: hipLaunchKernelGGL(attention_bf_kernel<0>, bgrid, dim3(512), LDS_WHOLE_CU, st, q); break;   // owns its CU (common.h)
        }
        RVC_LAUNCH_CHECK();
        if (q.n_splits > 1) {
            const int64_t work = (int64_t)n_heads * q.T * (64 / 4);
            hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)ceil_div(work, 256), (unsigned)batch), dim3(256), 0, st,
                               q.part_o, q.part_ml, q.out, q.T, n_heads, q.n_splits, 64);
            RVC_LAUNCH_CHECK();
        }
        return 0;
    }
    AttParams p;
    p.qkv = qkv_dev; p.emb_k = emb_rel_k_dev; p.emb_v = emb_rel_v_dev; p.out = out_dev;
    p.T = n_frames; p.n_heads = n_heads; p.scale_log2e = scale * 1.4426950408889634f;
    p.n_splits = choose_splits(n_frames, n_heads, batch);
    p.tiles_per_split = (int)ceil_div(ceil_div(n_frames, 32), p.n_splits);
    p.part_o = nullptr; p.part_ml = nullptr;
    if (p.n_splits > 1) {
        size_t need = 0;
        if (rvc_attention_workspace_bytes(batch, n_frames, n_heads, head_dim, &need)) return 1;
        if (!workspace_dev || workspace_bytes < need) return fail("rvc_attention_qkv_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
        const size_t rows = (size_t)batch * n_heads * p.n_splits * (size_t)n_frames;
        p.part_o = (float *)workspace_dev;
        p.part_ml = (float *)((char *)workspace_dev + align_up(rows * head_dim * sizeof(float), 256));
    }
    const bool rel = emb_rel_k_dev != nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (head_dim == 64) return rel ? launch_attention<64, true>(p, batch, st) : launch_attention<64, false>(p, batch, st);
    return rel ? launch_attention<96, true>(p, batch, st) : launch_attention<96, false>(p, batch, st);
}
