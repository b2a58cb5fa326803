// K5 -- the recurrent half of RMVPE's BiGRU (RMVPE.py:515-536: nn.GRU(384, 256, bidirectional)).
//
// The input projections W_ih x_t + b_ih for all 3232 frames are one GEMM (left to hipBLASLt on the host
// side); what remains is 2 x 3232 strictly sequential steps of a 768 x 256 mat-vec plus gate math -- 1.3 GFLOP
// of latency-bound work that a library RNN runs as ~50 000 tiny launches per utterance (≈ 200 ms).  Here one
// workgroup per (direction, batch item) walks the whole sequence inside a single launch: h lives in LDS, the
// recurrent weights stream from L2 (786 KB, resident) as 16-byte loads, two barriers per step.
//
// thread (jq, kq), jq in [0,192), kq in [0,4): partial dot products of gate rows 4jq..4jq+3 over k in [64kq, 64kq+64)
#include "common.h"

namespace rvc {

constexpr int GRU_H = 256;
constexpr int GRU_G = 3 * GRU_H;   // r, z, n rows
constexpr int GRU_THREADS = 768;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(GRU_THREADS)
bigru_kernel(const float *__restrict__ gi,      // [B][T][2][768]  W_ih x + b_ih
             const float *__restrict__ whhT,    // [2][256][768]   W_hh transposed
             const float *__restrict__ bhh,     // [2][768]
             float *__restrict__ out,           // [B][T][512]
             int T) {
    __shared__ __attribute__((aligned(16))) float part_s[4 * GRU_G];
    __shared__ float h_s[GRU_H];
    const int tid = threadIdx.x;
    const int dir = blockIdx.x;
    const int b = blockIdx.y;
    const int jq = tid % 192;
    const int kq = tid / 192;
    const float4 *W4 = reinterpret_cast<const float4 *>(whhT + (size_t)dir * GRU_H * GRU_G);
    const float *gib = gi + (size_t)b * T * 2 * GRU_G + (size_t)dir * GRU_G;
    float *outb = out + (size_t)b * T * 2 * GRU_H + dir * GRU_H;
    float b_r = 0.f, b_z = 0.f, b_n = 0.f;
    if (tid < GRU_H) {
        h_s[tid] = 0.f;
        b_r = bhh[dir * GRU_G + tid];
        b_z = bhh[dir * GRU_G + GRU_H + tid];
        b_n = bhh[dir * GRU_G + 2 * GRU_H + tid];
    }
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float gr = 0.f, gz = 0.f, gn = 0.f;
        if (tid < GRU_H) {  // prefetch this step's input gates; consumed after the mat-vec
            const float *g = gib + (size_t)t * 2 * GRU_G;
            gr = g[tid];
            gz = g[GRU_H + tid];
            gn = g[2 * GRU_H + tid];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 *wp = W4 + (size_t)(64 * kq) * 192 + jq;
        const float *hp = h_s + 64 * kq;
#pragma unroll 16
        for (int i = 0; i < 64; ++i) {
            const float4 w = wp[(size_t)i * 192];
            const float hk = hp[i];
            acc.x = fmaf(w.x, hk, acc.x);
            acc.y = fmaf(w.y, hk, acc.y);
            acc.z = fmaf(w.z, hk, acc.z);
            acc.w = fmaf(w.w, hk, acc.w);
        }
        *reinterpret_cast<float4 *>(&part_s[kq * GRU_G + 4 * jq]) = acc;
        __syncthreads();
        if (tid < GRU_H) {
            const int u = tid;
            const float hr = ((part_s[u] + part_s[GRU_G + u]) + (part_s[2 * GRU_G + u] + part_s[3 * GRU_G + u])) + b_r;
            const float hz = ((part_s[GRU_H + u] + part_s[GRU_G + GRU_H + u]) +
                              (part_s[2 * GRU_G + GRU_H + u] + part_s[3 * GRU_G + GRU_H + u])) + b_z;
            const float hn = ((part_s[2 * GRU_H + u] + part_s[GRU_G + 2 * GRU_H + u]) +
                              (part_s[2 * GRU_G + 2 * GRU_H + u] + part_s[3 * GRU_G + 2 * GRU_H + u])) + b_n;
            const float r = sigmoidf_(gr + hr);
            const float z = sigmoidf_(gz + hz);
            const float n = tanhf(gn + r * hn);
            const float hnew = (1.f - z) * n + z * h_s[u];
            h_s[u] = hnew;  // every thread finished reading h_s before the barrier above
            outb[(size_t)t * 2 * GRU_H + u] = hnew;
        }
        __syncthreads();
    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_bigru_forward(const float *gi_dev, const float *whhT_dev, const float *bhh_dev, float *out_dev,
                                 int batch, int64_t n_steps, int hidden, void *stream) {
    if (!gi_dev || !whhT_dev || !bhh_dev || !out_dev) return fail("rvc_bigru_forward: null pointer");
    if (hidden != GRU_H) return fail("rvc_bigru_forward: hidden size must be %d (RMVPE.py:325), got %d", GRU_H, hidden);
    if (batch <= 0 || n_steps <= 0 || n_steps > (1 << 30)) return fail("rvc_bigru_forward: bad shape");
    hipLaunchKernelGGL(bigru_kernel, dim3(2, batch), dim3(GRU_THREADS), 0, (hipStream_t)stream, gi_dev, whhT_dev, bhh_dev,
                       out_dev, (int)n_steps);
    RVC_LAUNCH_CHECK();
    return 0;
}
