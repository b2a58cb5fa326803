// K12 -- the GEMMs of HuBERT's transformer layers (q/k/v, attention output, feed-forward 1 / 2; 1599 frames x 768..3072) on the bf16
// matrix cores with fp32-exact operands, BOTH operands arriving pre-split:
//
//   y[n][m] = epilogue( sum_k W[m][k] * x[n][k] )            W: nn.Linear weight [out][in], x: activations [frames][in]
//
// Same arithmetic as gemmbf.hip (every fp32 operand = three bf16, six products of order <= 2^-16, fp32 accumulate), but where
// gemmbf.hip fetches fp32 activations and splits them between its matrix instructions -- 2.0 us per 16-deep step of a 128 x 256
// block, 36 % of the matrix pipe, and at 1599 columns a grid that fills half the chip: slower than hipBLASLt on every HuBERT
// projection -- this kernel's main loop contains NOTHING but LDS-DMA issue, fragment reads and matrix instructions:
//   * the producer of an activation writes it as three bf16 planes [split][frame][feature] next to (or instead of) its fp32 form:
//     the LayerNorm pass below, this kernel's own GELU epilogue, rvc_split_rows_bf16x3 behind the attention;
//   * a B fragment (8 consecutive features of one frame) is 16 contiguous bytes of a plane, so a 64-frame x 8-feature piece goes
//     HBM/L2 -> LDS by ONE buffer_load ... lds with per-lane row addresses, straight into the layout the matrix instruction reads;
//   * weights: gemmbf.hip's fragment slab (rvc_gemm_bf16x3_pack_weight), streamed the same way;
//   * 8 waves, 128 x 128 blocks, chunks of two 16-deep steps = 48 1 KiB pieces, requested two chunks ahead, one barrier per chunk.
// The grid is made to fill the chip per projection (1599 frames = 13 tiles of 128):
//   q/k/v   768 -> 2304  18 x 13 = 234 blocks                fp32 out + bias
//   out     768 ->  768  6 x 13, K in 3 parts = 234 blocks   fp32 partial sums
//   ff1     768 -> 3072  24 x 13 = 312 blocks                bias + GELU -> three bf16 planes
//   ff2    3072 ->  768  6 x 13, K in 3 parts = 234 blocks   fp32 partial sums
// and the partial sums meet in rvc_bias_residual_layernorm_bf16x3: sum of the parts + bias + residual -> LayerNorm -> fp32 AND the
// three planes the next GEMM reads (one pass over 1599 x 768 instead of the reference graph's add + layer_norm + a split pass).
// Replaces: the nn.Linear / LayerNorm / GELU modules of `transformers`' HubertEncoderLayer behind rvc/infer/pipeline.py:450.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv.h"

namespace rvc {

struct LinBfParams {
    const void *a = nullptr;         // gemmbf.hip's fragment slab [M / 128][K / 16][4][3][1 KiB]
    const void *xs = nullptr;        // [3][n_pad][K] bf16
    const float *bias = nullptr;     // [M] or null (modes 0, 1)
    float *y = nullptr;              // mode 0: [N][M] fp32; mode 2: [parts][N][M] fp32
    void *ys = nullptr;              // mode 1: [3][n_pad][M] bf16
    int M = 0, K = 0;
    int64_t N = 0, n_pad = 0;
    int mode = 0;                    // 0: y = acc + bias; 1: ys = split3(gelu(acc + bias)); 2: y[part] = acc; 3: y = gelu(acc + bias)
    // the activation planes as the kernel addresses them: GEMM row n starts x_row_stride elements after row n - 1 (a linear layer:
    // K; a strided conv over time-major frames: stride x channels, its K = taps x channels window being contiguous), x_plane
    // elements per split plane
    int64_t x_row_stride = 0, x_plane = 0;
    int steps_per_part = 0;          // 16-deep steps per K part
    int n_col_blocks = 0;
};

constexpr int LBF_BN = 128, LBF_NTH = 512;
constexpr int LBF_SLAB_STEP = 12 * 1024;                 // one 16-deep step of a 128-row block of the weight slab: [32-row block 4][split 3][1 KiB]
constexpr int LBF_B_CHUNK = 3 * LBF_BN * 64;             // TWO steps of the activations: [split 3][row 128][32 features = 64 bytes]
typedef float lbf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lbf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lbf_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned lbf_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned lbf_u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) *lbf_lptr_t;

// MI = 32-row blocks per wave: 2 -> 128 output features per block, chunks requested two ahead; 3 -> 192 per block (16 x 13 = 208
// blocks for the 3072-wide feed-forward instead of 24 x 13 = 312 on 256 CUs), chunks requested one ahead (LDS: 18 KiB of weights
// per step); 4 -> 256 per block (the long layers of the feature extractor: 0.75 of the bytes per product), one ahead, 24 KiB per step.
template <int MI>
struct LbfGeom {
    static constexpr int BM = 64 * MI;
    static constexpr int A_STEP = 2 * MI * 3 * 1024;     // [32-row block 2 MI][split 3][1 KiB]
    static constexpr int D = MI == 2 ? 2 : 1;            // chunks in flight ahead of the one being multiplied
    static constexpr int A_RING = 2 * (D + 1), B_RING = D + 1;
    static constexpr int NPIECE = 2 * 2 * MI * 3 + 24;   // 1 KiB pieces per chunk: two steps of weights + the activation chunk
    static constexpr int UPW = (NPIECE + 7) / 8;         // per wave (the overhang repeats one of the wave's own pieces)
    static constexpr int LDS_USED = A_RING * A_STEP + B_RING * LBF_B_CHUNK;
    static_assert(LDS_USED <= 163840, "");
};

// Block = 64 MI output features x 128 frames, 8 waves (2 x 4) of 32 MI x 32.  The K loop runs in chunks of two 16-deep steps: per
// chunk every wave issues its share of the chunk's 1 KiB LDS-DMA pieces (weight fragments + 24 activation pieces of 16 frames x 64
// contiguous bytes) D chunks ahead, reads 6 (MI + 1) fragments, issues 12 MI matrix instructions, and meets the others at one barrier.
// Activation pieces: a frame's 32 features of a chunk are 64 contiguous bytes of a plane (half a cache line), fetched by four lanes;
// the four 16-byte segments of a row land XOR-swizzled by (row >> 2) & 3, so the 16 lanes of a ds_read_b128 group -- 16 consecutive
// frames, same segment -- hit 16 different bank quads.  (First form: one lane per frame and 16-deep step, 64 different cache lines
// per DMA instruction: 1.5-2.3 us per step, slower than hipBLASLt on every projection.)
template <int MI>
__global__ void __launch_bounds__(LBF_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
linbf_kernel(const LinBfParams p) {
    using GM = LbfGeom<MI>;
    constexpr int BM = GM::BM, A_STEP = GM::A_STEP, D = GM::D, A_RING = GM::A_RING, B_RING = GM::B_RING, NPIECE = GM::NPIECE, UPW = GM::UPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lbf_smem[];
    unsigned char *const a_ring = lbf_smem;
    unsigned char *const b_ring = lbf_smem + A_RING * A_STEP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 8);
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    // Block b runs on XCD b % 8: XCD x takes the contiguous range [x S, (x + 1) S) of the (column tile, row block) pairs, S = ceil(pairs
    // / 8) -- consecutive pairs share a column tile, so its activations are fetched into that XCD's L2 once.  (Giving an XCD WHOLE
    // column tiles -- 13 tiles over 8 XCDs -- left five XCDs with 36 blocks for their 32 CUs and three with 18: two rounds.)
    const int n_m = p.M / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pairs = p.n_col_blocks * n_m, per_xcd = (pairs + 7) / 8;
    const int w = xcd * per_xcd + slot;
    if (slot >= per_xcd || w >= pairs) return;
    const int col_blk = w / n_m, mblk = w - col_blk * n_m, part = blockIdx.y;
    const int m0 = mblk * BM;
    const int64_t n0 = (int64_t)col_blk * LBF_BN;
    const int n_chunks = p.steps_per_part / 2, s_first = part * p.steps_per_part;
    const int K = p.K;

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, (int)((int64_t)p.M * K * 6), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.xs, 0, (int)(3 * p.x_plane * 2), 0x00020000);
    const int slab_row = (K / 16) * LBF_SLAB_STEP;           // bytes of one 128-row block of the slab
    const int plane = (int)(p.x_plane * 2);
    const int row_bytes = (int)(p.x_row_stride * 2);
    // activation piece: lane = (row r of 16, LDS segment j of 4); it fetches the row's segment j ^ ((row >> 2) & 3)
    const int b_r = lane >> 2, b_j = lane & 3;
    // chunk c of this block's K part: pieces 0 .. 12 MI - 1: the two steps' weight fragments; then 24 activation pieces
    auto dma = [&](int c) __attribute__((always_inline)) {
        const int sg = s_first + 2 * c;                         // first global step of the chunk
        unsigned char *bslot = b_ring + (c % B_RING) * LBF_B_CHUNK;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            int pc = wave + 8 * i;
            if (pc >= NPIECE) pc -= 8;                          // same bytes to the same place
            if (pc < 12 * MI) {
                const int st = pc / (6 * MI), f = pc - st * (6 * MI);     // step of the chunk, fragment (32-row block, split)
                const int rb = (m0 >> 5) + f / 3, sp = f - (f / 3) * 3;   // global 32-row block
                unsigned char *aslot = a_ring + ((2 * c + st) % A_RING) * A_STEP;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lbf_lptr_t)(aslot + f * 1024), 16, 16 * lane,
                                                         (rb >> 2) * slab_row + (sg + st) * LBF_SLAB_STEP + ((rb & 3) * 3 + sp) * 1024, 0, 0);
            } else {
                const int q = pc - 12 * MI, sp = q >> 3, r16 = q & 7;
                const int row = r16 * 16 + b_r;
                const int seg = b_j ^ ((row >> 2) & 3);
                const int voff = (int)(n0 + row) * row_bytes + seg * 16;   // (rows of the last tile past the planes' end: plane 0 / 1 read into the next plane, plane 2 reads zeros; their results are not stored)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lbf_lptr_t)(bslot + sp * (LBF_BN * 64) + r16 * 1024), 16, voff, sp * plane + sg * 32, 0, 0);
            }
        }
    };

    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    auto rd = [&](const unsigned char *ptr) __attribute__((always_inline)) {
        return __builtin_bit_cast(lbf_bf16x8, *reinterpret_cast<const lbf_u32x4 *>(ptr));
    };
    const int nrow = wn * 32 + l31;                           // this lane's frame inside the tile
    const int b_lane = nrow * 64, b_sw = (nrow >> 2) & 3;

    // Chunks are requested D ahead into a ring of D + 1: chunk c + D goes into the slots chunk c - 1 left at the last barrier.
    auto wait_newer = [&](int newer) __attribute__((always_inline)) {   // everything but the `newer` most recent chunk requests has landed
        if (newer >= 1 && D >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(D <= 2, "wait_newer distinguishes 0 and 1 newer requests");
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < n_chunks) dma(c);
    wait_newer((n_chunks - 1 < D - 1) ? n_chunks - 1 : D - 1);
    lds_barrier();
    for (int c = 0; c < n_chunks; ++c) {
        if (c + D < n_chunks) dma(c + D);
        const unsigned char *bslot = b_ring + (c % B_RING) * LBF_B_CHUNK + b_lane;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned char *ab = a_ring + ((2 * c + t) % A_RING) * A_STEP + wm * MI * 3 * 1024 + lane * 16;
            lbf_bf16x8 fa[MI][3], fb[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) fb[sp] = rd(bslot + sp * (LBF_BN * 64) + (((2 * t + half) ^ b_sw) * 16));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) fa[mi][sp] = rd(ab + (mi * 3 + sp) * 1024);
            constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ia[i]], fb[ib[i]], acc[mi], 0, 0, 0);
        }
        // chunk c + 1 has landed (requests of chunks c + 2 .. c + D may stay in flight); one barrier says that to every wave and frees chunk c's slots
        const int newer = (n_chunks - 2 - c) < (D - 1) ? (n_chunks - 2 - c) : (D - 1);
        wait_newer(newer);
        lds_barrier();
    }

    // ---- epilogue: lane (frame n, features 4 half + 8 rg + 0..3 of each 32-row block) ------------------------------------------
    if (p.mode == 1) {
        // planes: a lane's four features of a register group are 8 bytes of a plane row; written as such, a row's 64-byte line is
        // touched by eight 8-byte stores and the big layers' dirty partial lines leave L2 before they are complete (PMC: 490 MB
        // written for 157 MB of planes on the feature extractor's first layer).  The two half-waves hold neighbouring feature
        // quads of the SAME frames, so they swap one quad per register-group pair and each lane stores 16 bytes; the two lanes
        // of a frame then cover 32 contiguous bytes in one instruction.
        const int64_t n = n0 + nrow;
        const bool live = n < p.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mb = m0 + wm * 32 * MI + mi * 32;
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {                // register groups 2 rp (features 16 rp + 0..7), 2 rp + 1 (16 rp + 8..15)
                lbf_f32x2 v[2][2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[mi][4 * (2 * rp + g) + r];
                    if (p.bias) o += *reinterpret_cast<const f32x4 *>(p.bias + mb + 4 * half + 8 * (2 * rp + g));
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = 0.5f * o[r] * (1.f + erff(o[r] * 0.70710678118654752f));
                    v[g][0] = lbf_f32x2{o[0], o[1]};
                    v[g][1] = lbf_f32x2{o[2], o[3]};
                }
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    unsigned w[2][2];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            w[g][h] = __builtin_bit_cast(unsigned, __builtin_convertvector(v[g][h], lbf_bf16x2));
                            v[g][h] = v[g][h] - lbf_f32x2{__uint_as_float(w[g][h] << 16), __uint_as_float(w[g][h] & 0xffff0000u)};
                        }
                    // half 0 keeps group 2 rp and receives the partner's quad of it; half 1 keeps group 2 rp + 1
                    const unsigned s0 = half ? w[0][0] : w[1][0], s1 = half ? w[0][1] : w[1][1];
                    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 32, 64), r1 = (unsigned)__shfl_xor((int)s1, 32, 64);
                    const lbf_u32x4 out = half ? lbf_u32x4{r0, r1, w[1][0], w[1][1]} : lbf_u32x4{w[0][0], w[0][1], r0, r1};
                    const int m = mb + 16 * rp + 8 * half;
                    if (live) *reinterpret_cast<lbf_u32x4 *>(reinterpret_cast<unsigned char *>(p.ys) + (((int64_t)sp * p.n_pad + n) * p.M + m) * 2) = out;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int64_t n = n0 + nrow;
        if (n >= p.N) continue;
        const int mb = m0 + wm * 32 * MI + mi * 32 + 4 * half;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int m = mb + 8 * rg;
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[mi][4 * rg + r];
            if (p.mode == 2) {
                *reinterpret_cast<f32x4 *>(p.y + ((int64_t)part * p.N + n) * p.M + m) = o;
                continue;
            }
            if (p.bias) o += *reinterpret_cast<const f32x4 *>(p.bias + m);
            if (p.mode == 0) {
                *reinterpret_cast<f32x4 *>(p.y + n * p.M + m) = o;
                continue;
            }
            // mode 3: GELU (erf form, what torch.nn.functional.gelu computes) -> fp32
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = 0.5f * o[r] * (1.f + erff(o[r] * 0.70710678118654752f));
            *reinterpret_cast<f32x4 *>(p.y + n * p.M + m) = o;
        }
    }
}

// fp32 rows -> three bf16 planes (the exact split); one thread = four consecutive features
__global__ void __launch_bounds__(256) split_rows_kernel(const float *__restrict__ x, unsigned char *__restrict__ xs, int64_t n_rows,
                                                          int64_t n_pad, int k) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total4 = n_rows * k / 4;
    if (i4 >= total4) return;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(x + i4 * 4);
    lbf_f32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w};
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
        const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, lbf_bf16x2));
        const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, lbf_bf16x2));
        *reinterpret_cast<lbf_u32x2 *>(xs + ((int64_t)sp * n_pad * k + i4 * 4) * 2) = lbf_u32x2{w0, w1};
        v0 = v0 - lbf_f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
        v1 = v1 - lbf_f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
    }
}

// one wave per row: v = sum of the K parts + bias + residual; LayerNorm over the row (two passes in registers, fp32 like
// torch.nn.functional.layer_norm); fp32 out and the three planes
template <int PER_LANE>   // M = 64 * PER_LANE, PER_LANE a multiple of 4
__global__ void __launch_bounds__(256) ln_reduce_kernel(const float *__restrict__ parts, int n_parts, const float *__restrict__ bias,
                                                         const float *__restrict__ res, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, float *__restrict__ y,
                                                         unsigned char *__restrict__ ys, int64_t n_rows, int64_t n_pad) {
    constexpr int M = 64 * PER_LANE, Q = PER_LANE / 4;
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rows) return;
    f32x4 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int m = (q * 64 + lane) * 4;
        f32x4 a = bias ? *reinterpret_cast<const f32x4 *>(bias + m) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < n_parts; ++s) a += *reinterpret_cast<const f32x4 *>(parts + ((int64_t)s * n_rows + n) * M + m);
        if (res) a += *reinterpret_cast<const f32x4 *>(res + n * M + m);
        v[q] = a;
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.f / M);
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const f32x4 d = v[q] - mean;
        sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = 1.f / sqrtf(sq * (1.f / M) + eps);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int m = (q * 64 + lane) * 4;
        f32x4 o = (v[q] - mean) * rstd;
        if (gamma) o *= *reinterpret_cast<const f32x4 *>(gamma + m);
        if (beta) o += *reinterpret_cast<const f32x4 *>(beta + m);
        if (y) *reinterpret_cast<f32x4 *>(y + n * M + m) = o;
        if (ys) {
            lbf_f32x2 v0 = {o.x, o.y}, v1 = {o.z, o.w};
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, lbf_bf16x2));
                const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, lbf_bf16x2));
                *reinterpret_cast<lbf_u32x2 *>(ys + (((int64_t)sp * n_pad + n) * M + m) * 2) = lbf_u32x2{w0, w1};
                v0 = v0 - lbf_f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
                v1 = v1 - lbf_f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
            }
        }
    }
}

template <int MI>
static int linbf_launch(LinBfParams p, int parts, hipStream_t stream) {
    if (reserve_whole_cu((const void *)linbf_kernel<MI>, "linear bf16x3 (pre-split)")) return 1;
    p.n_col_blocks = (int)ceil_div(p.N, LBF_BN);
    const int n_m = p.M / LbfGeom<MI>::BM;
    dim3 grid((unsigned)(ceil_div((int64_t)p.n_col_blocks * n_m, 8) * 8), (unsigned)parts, 1);
    hipLaunchKernelGGL(linbf_kernel<MI>, grid, dim3(LBF_NTH), LDS_WHOLE_CU, stream, p);   // owns its CU (common.h)
    RVC_LAUNCH_CHECK();
    return 0;
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_split_rows_bf16x3(const float *x_dev, void *xs_dev, int64_t n_rows, int64_t n_rows_padded, int k, void *stream) {
    if (!x_dev || !xs_dev) return fail("rvc_split_rows_bf16x3: null pointer");
    if (k % 16 || n_rows_padded < n_rows || n_rows < 0) return fail("rvc_split_rows_bf16x3: k must be a multiple of 16, n_rows_padded >= n_rows");
    if (n_rows == 0) return 0;
    const int64_t total4 = n_rows * k / 4;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       reinterpret_cast<unsigned char *>(xs_dev), n_rows, n_rows_padded, k);
    RVC_LAUNCH_CHECK();
    return 0;
}

namespace {
int linbf_dispatch(const char *who, const void *xs_dev, int64_t x_row_stride, int64_t x_plane, const void *a_dev, const float *bias_dev,
                   float *y_dev, void *ys_dev, int64_t n_rows, int64_t n_rows_padded, int in_features, int out_features, int mode,
                   int k_parts, void *stream) {
    if (!xs_dev || !a_dev) return fail("%s: null pointer", who);
    if (out_features % 128 || in_features % 16) return fail("%s: out_features must be a multiple of 128, in_features of 16", who);
    if (mode < 0 || mode > 3) return fail("%s: mode must be 0 (fp32 + bias), 1 (bias + GELU -> planes), 2 (partial sums) or 3 (bias + GELU -> fp32)", who);
    if ((mode == 1) ? !ys_dev : !y_dev) return fail("%s: the output of mode %d is missing", who, mode);
    if (k_parts < 1 || (in_features / 32) % k_parts || in_features % 32) return fail("%s: in_features must be a multiple of 32 and k_parts divide in_features / 32", who);
    if (k_parts > 1 && mode != 2) return fail("%s: several K parts only produce partial sums (mode 2)", who);
    if (n_rows_padded % LBF_BN || n_rows_padded < n_rows) return fail("%s: n_rows_padded must be a multiple of 128 and >= n_rows", who);
    if ((int64_t)out_features * in_features * 6 >= ((int64_t)1 << 31) || 3 * x_plane * 2 >= ((int64_t)1 << 31) ||
        (n_rows_padded + LBF_BN) * x_row_stride * 2 >= ((int64_t)1 << 31))
        return fail("%s: an operand exceeds 2 GiB", who);
    if (x_row_stride % 8) return fail("%s: rows must start on 16-byte boundaries", who);
    if (n_rows <= 0) return 0;
    LinBfParams p;
    p.a = a_dev; p.xs = xs_dev; p.bias = bias_dev; p.y = y_dev; p.ys = ys_dev;
    p.M = out_features; p.K = in_features; p.N = n_rows; p.n_pad = n_rows_padded; p.mode = mode;
    p.x_row_stride = x_row_stride; p.x_plane = x_plane;
    p.steps_per_part = in_features / 16 / k_parts;
    // Block height: 128 output features unless taller blocks need fewer ROUNDS of the 256 CUs at their relative block time -- a
    // 192-row block works 1.5 x as long at ~0.85 of the bytes per product (HuBERT's 3072-wide feed-forward at 1599 frames: 208
    // blocks in one round against 312 in two), a 256-row block 1.85 x as long (measured: 0.75 of the bytes per product, but its
    // 24 KiB steps leave room for one chunk in flight instead of two; the feature extractor's 47 999-frame layer: 750 blocks = 3
    // rounds, 586 us, against 1500 = 6 rounds, 642 us): the kernel runs at the CU's ingest rate, so bytes per product are time
    const int64_t cols = ceil_div(n_rows, LBF_BN);
    const int cus = 256;
    auto cost = [&](int rows_per_block, double rel) {
        if (out_features % rows_per_block) return 1e30;
        return (double)ceil_div(cols * (out_features / rows_per_block) * k_parts, cus) * rel;
    };
    // (ablation build, RVC_LBF_TP=1: pick the block height by CU TIME -- blocks x relative block time -- instead of rounds: what a
    // launch costs while other streams' kernels fill the CUs it leaves idle)
    static const int by_cu_time = knob("RVC_LBF_TP", 0);
    if (by_cu_time) {
        auto cu_time = [&](int rows_per_block, double rel) { return out_features % rows_per_block ? 1e30 : cols * (out_features / rows_per_block) * k_parts * rel; };
        const double t128 = cu_time(128, 1.0), t192 = cu_time(192, 1.5 * 0.85), t256 = cu_time(256, 1.85);
        if (t192 <= t128 && t192 <= t256) return linbf_launch<3>(p, k_parts, (hipStream_t)stream);
        if (t256 < t128) return linbf_launch<4>(p, k_parts, (hipStream_t)stream);
        return linbf_launch<2>(p, k_parts, (hipStream_t)stream);
    }
    const double c128 = cost(128, 1.0), c192 = cost(192, 1.5 * 0.85), c256 = cost(256, 1.85);
    if (c256 < c128 && c256 <= c192) return linbf_launch<4>(p, k_parts, (hipStream_t)stream);
    const bool use192 = c192 < c128;
    return use192 ? linbf_launch<3>(p, k_parts, (hipStream_t)stream) : linbf_launch<2>(p, k_parts, (hipStream_t)stream);
}
}  // namespace

extern "C" int rvc_linear_bf16x3_presplit(const void *xs_dev, const void *a_dev, const float *bias_dev, float *y_dev, void *ys_dev,
                                          int64_t n_rows, int64_t n_rows_padded, int in_features, int out_features, int mode, int k_parts,
                                          void *stream) {
    return linbf_dispatch("rvc_linear_bf16x3_presplit", xs_dev, in_features, n_rows_padded * in_features, a_dev, bias_dev, y_dev, ys_dev,
                          n_rows, n_rows_padded, in_features, out_features, mode, k_parts, stream);
}

extern "C" int rvc_conv1d_frames_bf16x3(const void *xs_dev, int64_t n_frames_in, int64_t n_frames_in_padded, int channels, int taps,
                                        int stride, const void *a_dev, const float *bias_dev, float *y_dev, void *ys_dev,
                                        int64_t n_frames_out_padded, int out_channels, int mode, void *stream) {
    if (taps < 1 || stride < 1 || channels < 8 || n_frames_in_padded < n_frames_in) return fail("rvc_conv1d_frames_bf16x3: bad argument");
    if (mode == 2) return fail("rvc_conv1d_frames_bf16x3: modes 0 (fp32 + bias), 1 (bias + GELU -> planes), 3 (bias + GELU -> fp32)");
    const int64_t n_out = n_frames_in >= taps ? (n_frames_in - taps) / stride + 1 : 0;
    return linbf_dispatch("rvc_conv1d_frames_bf16x3", xs_dev, (int64_t)stride * channels, n_frames_in_padded * channels, a_dev, bias_dev,
                          y_dev, ys_dev, n_out, n_frames_out_padded, taps * channels, out_channels, mode, 1, stream);
}

extern "C" int rvc_bias_residual_layernorm_bf16x3(const float *parts_dev, int n_parts, const float *bias_dev, const float *res_dev,
                                                  const float *gamma_dev, const float *beta_dev, float eps, float *y_dev, void *ys_dev,
                                                  int64_t n_rows, int64_t n_rows_padded, int features, void *stream) {
    if (!parts_dev || (!y_dev && !ys_dev)) return fail("rvc_bias_residual_layernorm_bf16x3: null pointer");
    if (n_parts < 1 || n_rows_padded < n_rows) return fail("rvc_bias_residual_layernorm_bf16x3: bad argument");
    if (n_rows <= 0) return 0;
    dim3 grid((unsigned)ceil_div(n_rows, 4));
    unsigned char *ys = reinterpret_cast<unsigned char *>(ys_dev);
    hipStream_t st = (hipStream_t)stream;
    if (features == 768)
        hipLaunchKernelGGL(ln_reduce_kernel<12>, grid, dim3(256), 0, st, parts_dev, n_parts, bias_dev, res_dev, gamma_dev, beta_dev, eps, y_dev, ys, n_rows, n_rows_padded);
    else if (features == 1024)
        hipLaunchKernelGGL(ln_reduce_kernel<16>, grid, dim3(256), 0, st, parts_dev, n_parts, bias_dev, res_dev, gamma_dev, beta_dev, eps, y_dev, ys, n_rows, n_rows_padded);
    else if (features == 256)
        hipLaunchKernelGGL(ln_reduce_kernel<4>, grid, dim3(256), 0, st, parts_dev, n_parts, bias_dev, res_dev, gamma_dev, beta_dev, eps, y_dev, ys, n_rows, n_rows_padded);
    else
        return fail("rvc_bias_residual_layernorm_bf16x3: features must be 256, 768 or 1024 (HuBERT-base: 768)");
    RVC_LAUNCH_CHECK();
    return 0;
}
