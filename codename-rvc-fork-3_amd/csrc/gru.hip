// K5 -- the recurrent half of RMVPE's BiGRU (RMVPE.py:515-536: nn.GRU(384, 256, bidirectional)).
//
// The input projections W_ih x_t + b_ih for all 3232 frames are one GEMM (left to hipBLASLt on the host
// side); what remains is 2 x 3232 strictly sequential steps of a 768 x 256 mat-vec plus gate math -- 1.3 GFLOP
// of latency-bound work that a library RNN runs as ~50 000 tiny launches per utterance (≈ 200 ms).  Here one
// workgroup per (direction, batch item) walks the whole sequence inside a single launch: h lives in LDS, the
// recurrent weights stream from L2 (786 KB, resident) as 16-byte loads, two barriers per step.
//
// thread (jq, kq), jq in [0,192), kq in [0,4): partial dot products of gate rows 4jq..4jq+3 over k in [64kq, 64kq+64)
#include <stdlib.h>

#include <atomic>

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
             int T,
             const int *__restrict__ redo) {    // NULL: always run; else run (b, dir) only if redo[2 b + dir] != 0
    __shared__ __attribute__((aligned(16))) float part_s[4 * GRU_G];
    __shared__ float h_s[GRU_H];
    const int tid = threadIdx.x;
    const int dir = blockIdx.x;
    const int b = blockIdx.y;
    if (redo && redo[2 * b + dir] == 0) return;   // block-uniform: the multi-workgroup pass of this sequence was clean
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


// ---- multi-CU variant: weights register-resident, h exchanged through 8-byte {epoch, value} granules ----------
// GRU_CUS workgroups per (direction, batch item); workgroup c owns hidden units [64c, 64c+64): its 192 gate rows x 256
// columns of W_hh live in VGPRs (64 per thread, thread = (row, k-quarter)), so a step touches no weight memory at
// all.  After the gate math the 64 new h values are published as self-validating granules (one aligned 8-byte
// write-through store each: tag = step + 1, no separate flag, no fence -- cdna_hip_programming.md §6 G16 form R2);
// one wave polls the 192 foreign granules with relaxed agent-scope loads.  Two parities of the exchange buffer are
// enough: a workgroup cannot get two steps ahead of a peer whose step-s value it still needs.
// Every spin is bounded.  The 8 workgroups of a launch are normally co-resident within microseconds; when other streams
// hold every block slot they start one by one as slots free up and the early ones wait.  If a wait ever exceeds the
// bound, the workgroup stops polling, raises status[2 b + dir] and the launch is followed by bigru_kernel restricted
// to the raised (batch item, direction) pairs -- the single-workgroup recurrence, slower but free of any rendezvous --
// which overwrites that sequence's output.  No host round trip is involved; the status words stay readable afterwards
// (rvc_bigru_status).
constexpr int GRU_CUS = 4;
constexpr int GRU_UNITS = GRU_H / GRU_CUS;      // 64 hidden units per workgroup
constexpr int GRU_ROWS = 3 * GRU_UNITS;         // 192 gate rows per workgroup
constexpr int GRU_HS = 68;                      // LDS stride between k-quarters of h (bank spread for the b128 reads)
constexpr unsigned GRU_SPIN_LIMIT = 1u << 22;   // default bound of one rendezvous wait (~0.3 s of polling)

typedef unsigned long long u64;

__global__ void __launch_bounds__(GRU_THREADS)
bigru_mc_kernel(const float *__restrict__ gi, const float *__restrict__ whhT, const float *__restrict__ bhh,
                float *__restrict__ out, u64 *__restrict__ xchg, int *__restrict__ status, int T, unsigned spin_limit) {
    __shared__ __attribute__((aligned(16))) float h_s[4 * GRU_HS];
    __shared__ float g_s[GRU_ROWS];
    __shared__ int dead_s;
    const int tid = threadIdx.x;
    // Workgroups go round-robin over the 8 XCDs.  The four workgroups of one direction exchange h every step, so they are
    // given ids 8 apart -- the same XCD -- and the 24 ids in between exit at once (grid.x = 32: id % 8 = direction,
    // id / 8 = slice): 1.37-1.52 us per step against 1.6-1.8 with one workgroup per XCD (RVC_GRU_SPREAD=1).  The granules
    // still need agent-scope accesses: group-scope ones (sc0, with or without an L1 invalidate) are served from the CU's
    // L1 and never see the partner's store.
    int c, dir;
    if (gridDim.x == 8 * GRU_CUS) {
        if ((blockIdx.x & 7) >= 2) return;
        dir = blockIdx.x & 7;
        c = blockIdx.x >> 3;
    } else {
        c = blockIdx.x % GRU_CUS;
        dir = blockIdx.x / GRU_CUS;
    }
    const int b = blockIdx.y;
    const int q = tid & 3;            // k quarter
    const int r = tid >> 2;           // local gate row 0..191
    const int gate = r / GRU_UNITS;
    const int ul = r % GRU_UNITS;
    const int j = gate * GRU_H + c * GRU_UNITS + ul;   // global gate row
    const float *W = whhT + (size_t)dir * GRU_H * GRU_G;
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = W[(size_t)(64 * q + i) * GRU_G + j];
    const float *gib = gi + (size_t)b * T * 2 * GRU_G + (size_t)dir * GRU_G + c * GRU_UNITS;
    float *outb = out + (size_t)b * T * 2 * GRU_H + dir * GRU_H + c * GRU_UNITS;
    u64 *xb = xchg + ((size_t)b * 2 + dir) * 2 * GRU_H;   // [parity][256]
    float b_r = 0.f, b_z = 0.f, b_n = 0.f;
    if (tid < GRU_UNITS) {
        b_r = bhh[dir * GRU_G + c * GRU_UNITS + tid];
        b_z = bhh[dir * GRU_G + GRU_H + c * GRU_UNITS + tid];
        b_n = bhh[dir * GRU_G + 2 * GRU_H + c * GRU_UNITS + tid];
    }
    if (tid < 4 * GRU_HS) h_s[tid] = 0.f;
    if (tid == 0) dead_s = 0;
    __syncthreads();
    const float4 *hq = reinterpret_cast<const float4 *>(h_s + GRU_HS * q);
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float gr = 0.f, gz = 0.f, gn = 0.f;
        if (tid < GRU_UNITS) {
            const float *g = gib + (size_t)t * 2 * GRU_G;
            gr = g[tid];
            gz = g[GRU_H + tid];
            gn = g[2 * GRU_H + tid];
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 hv = hq[i];
            a0 = fmaf(w[4 * i + 0], hv.x, a0);
            a1 = fmaf(w[4 * i + 1], hv.y, a1);
            a2 = fmaf(w[4 * i + 2], hv.z, a2);
            a3 = fmaf(w[4 * i + 3], hv.w, a3);
        }
        float acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (q == 0) g_s[r] = acc;
        __syncthreads();
        const unsigned epoch = (unsigned)s + 1u;
        u64 *xp = xb + (size_t)(s & 1) * GRU_H;
        if (tid < GRU_UNITS) {
            const int u = c * GRU_UNITS + tid;
            const float hold = h_s[GRU_HS * (u >> 6) + (u & 63)];
            const float rr = sigmoidf_(gr + (g_s[tid] + b_r));
            const float zz = sigmoidf_(gz + (g_s[GRU_UNITS + tid] + b_z));
            const float nn = tanhf(gn + rr * (g_s[2 * GRU_UNITS + tid] + b_n));
            const float hnew = (1.f - zz) * nn + zz * hold;
            __hip_atomic_store(xp + u, ((u64)epoch << 32) | (u64)__float_as_uint(hnew), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            h_s[GRU_HS * (u >> 6) + (u & 63)] = hnew;
            outb[(size_t)t * 2 * GRU_H + tid] = hnew;
        } else if (tid < 2 * GRU_UNITS) {
            // wave 1: gather the other workgroups' 192 values of this step
            const int lane = tid - GRU_UNITS;
            unsigned spins = 0;
            int idx[GRU_CUS - 1];
#pragma unroll
            for (int m = 0; m < GRU_CUS - 1; ++m) idx[m] = ((c + 1 + m) % GRU_CUS) * GRU_UNITS + lane;
            if (!dead_s) {
                for (;;) {
                    bool ok = true;
                    float v[GRU_CUS - 1];
#pragma unroll
                    for (int m = 0; m < GRU_CUS - 1; ++m) {
                        const u64 x = __hip_atomic_load(xp + idx[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned)(x >> 32) == epoch;
                        v[m] = __uint_as_float((unsigned)x);
                    }
                    if (__all(ok)) {
#pragma unroll
                        for (int m = 0; m < GRU_CUS - 1; ++m) h_s[GRU_HS * (idx[m] >> 6) + (idx[m] & 63)] = v[m];
                        break;
                    }
                    if (++spins > spin_limit) { if (lane == 0) dead_s = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
    }
    if (dead_s && tid == 0) atomicAdd(&status[2 * b + dir], 1);   // device-scope; read by the launch that follows
}

}  // namespace rvc

using namespace rvc;

static std::atomic<unsigned> g_spin_limit{GRU_SPIN_LIMIT};

static size_t bigru_xchg_bytes(int batch) { return (size_t)batch * 2 * 2 * GRU_H * sizeof(u64); }

extern "C" int rvc_bigru_workspace_bytes(int batch, size_t *bytes) {
    if (!bytes || batch <= 0) return fail("rvc_bigru_workspace_bytes: bad argument");
    *bytes = bigru_xchg_bytes(batch) + align_up((size_t)batch * 2 * sizeof(int), 256);   // granules, then status words
    return 0;
}

extern "C" int rvc_bigru_set_spin_limit(unsigned polls) {
    g_spin_limit.store(polls ? polls : GRU_SPIN_LIMIT, std::memory_order_relaxed);
    return 0;
}

extern "C" int rvc_bigru_status(const void *workspace_dev, int batch, int *n_redone_host, void *stream) {
    if (!workspace_dev || !n_redone_host || batch <= 0 || batch > 64) return fail("rvc_bigru_status: bad argument");
    int st[128];
    RVC_HIP(hipMemcpyAsync(st, (const char *)workspace_dev + bigru_xchg_bytes(batch), (size_t)batch * 2 * sizeof(int),
                           hipMemcpyDeviceToHost, (hipStream_t)stream));
    RVC_HIP(hipStreamSynchronize((hipStream_t)stream));
    int n = 0;
    for (int i = 0; i < batch * 2; ++i) n += st[i] != 0;
    *n_redone_host = n;
    return 0;
}

extern "C" int rvc_bigru_forward(const float *gi_dev, const float *whhT_dev, const float *bhh_dev, float *out_dev,
                                 int batch, int64_t n_steps, int hidden, void *workspace_dev, size_t workspace_bytes,
                                 void *stream) {
    if (!gi_dev || !whhT_dev || !bhh_dev || !out_dev) return fail("rvc_bigru_forward: null pointer");
    if (hidden != GRU_H) return fail("rvc_bigru_forward: hidden size must be %d (RMVPE.py:325), got %d", GRU_H, hidden);
    if (batch <= 0 || n_steps <= 0 || n_steps > (1 << 30)) return fail("rvc_bigru_forward: bad shape");
    if (!workspace_dev) {  // single-workgroup-per-direction variant (weights streamed from L2)
        hipLaunchKernelGGL(bigru_kernel, dim3(2, batch), dim3(GRU_THREADS), 0, (hipStream_t)stream, gi_dev, whhT_dev,
                           bhh_dev, out_dev, (int)n_steps, (const int *)nullptr);
        RVC_LAUNCH_CHECK();
        return 0;
    }
    size_t need = 0;
    if (rvc_bigru_workspace_bytes(batch, &need)) return 1;
    if (workspace_bytes < need) return fail("rvc_bigru_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
    if (batch * 2 * GRU_CUS > 128) return fail("rvc_bigru_forward: batch %d needs %d co-resident workgroups (max 128)", batch, batch * 2 * GRU_CUS);
    RVC_HIP(hipMemsetAsync(workspace_dev, 0, need, (hipStream_t)stream));  // tags must start below epoch 1 on every call; status = 0
    int *status = (int *)((char *)workspace_dev + bigru_xchg_bytes(batch));
    static const int spread = getenv("RVC_GRU_SPREAD") ? atoi(getenv("RVC_GRU_SPREAD")) : 0;
    hipLaunchKernelGGL(bigru_mc_kernel, dim3(spread ? 2 * GRU_CUS : 8 * GRU_CUS, batch), dim3(GRU_THREADS), 0, (hipStream_t)stream, gi_dev, whhT_dev,
                       bhh_dev, out_dev, (u64 *)workspace_dev, status, (int)n_steps, g_spin_limit.load(std::memory_order_relaxed));
    RVC_LAUNCH_CHECK();
    // 2 x batch workgroups that return at once unless their sequence's rendezvous timed out (see bigru_mc_kernel)
    hipLaunchKernelGGL(bigru_kernel, dim3(2, batch), dim3(GRU_THREADS), 0, (hipStream_t)stream, gi_dev, whhT_dev, bhh_dev,
                       out_dev, (int)n_steps, (const int *)status);
    RVC_LAUNCH_CHECK();
    return 0;
}
