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


constexpr int GRU_NWG = 8;                      // workgroups per (direction, batch item) of bigru_x_kernel
constexpr unsigned GRU_SPIN_LIMIT = 1u << 22;   // default bound of one rendezvous wait (~0.3 s of polling)

typedef unsigned long long u64;

// ---- multi-CU form: weights register-resident, h exchanged through 8-byte {epoch, value} granules --------------------
// NWG = 8 workgroups of 256 threads per (direction, batch item); workgroup c owns UNITS = 32 hidden units and thread
// (ul, ks) holds, in registers, the r / z / n rows of unit ul over the k slice ks (96 weights) -- which is exactly the slice
// workgroup ks produces -- so a step touches no weight memory at all.  A step is:
//   3 x 32 multiply-adds from registers against the slice of h in LDS (48 v_pk_fma_f32, 8 ds_read_b128);
//   a 3-stage DPP reduction inside the unit's 8-lane group (no LDS pass, no barrier before the gate math);
//   the gate math in those same lanes, sigma(r) and sigma(z) side by side in even / odd lanes;
//   one granule per unit: an aligned 8-byte write-through store {tag = step + 1, value}, no separate flag, no fence
//   (cdna_hip_programming.md section 6, G16 form R2); 224 threads each poll ONE foreign granule (relaxed agent-scope loads)
//   into the other parity of h in LDS;  one barrier.
// Two parities of the exchange buffer are enough: a workgroup cannot get two steps ahead of a peer whose step-s value it
// still needs.  The workgroups of one direction are given ids 8 apart -- the same XCD -- and the ids in between exit at once.
//
// Measured on MI355X at T = 3232 (tools/bench_gru.py; RVC_GRU_DBG ablations), per step:
//   previous form (4 workgroups x 768 threads = (gate row, k quarter), reduction and gate math behind a barrier in one wave,
//   input gates loaded in the step that uses them)                                                        1.50 us
//   this form, input gates loaded one step ahead                                                            1.28
//   + input gates fetched 16 steps at a time, a chunk ahead, parked in LDS (a load issued in step s is otherwise waited
//     for by step s's granule polls at its full HBM-miss latency)                                           1.06-1.13
//   of which: without the exchange 0.49, without exchange and gate math 0.40; library expf / tanhf / division +0.09.
//   The exchange is ~0.6 us of the step and is the hand-off primitive itself: a dedicated fifth wave polling continuously
//   from the previous barrier on (1.20), workgroup-scope stores + L2-executed atomics as the poll (global_atomic_or_x2 of 0:
//   correct when the workgroups share an XCD, 1.16-1.20), 4 or 16 workgroups per direction (1.15 / 1.14: less arithmetic per
//   step buys nothing once the hop dominates) all measured slower or level.
// Every spin is bounded.  The workgroups of a launch are normally co-resident within microseconds; when other streams
// hold every block slot they start one by one as slots free up and the early ones wait.  If a wait ever exceeds the
// bound, the workgroup stops polling, raises status[2 b + dir] and the launch is followed by bigru_kernel restricted
// to the raised (batch item, direction) pairs -- the single-workgroup recurrence, slower but free of any rendezvous --
// which overwrites that sequence's output.  No host round trip is involved; the status words stay readable afterwards
// (rvc_bigru_status).
// DBG (wrong results): 1 no exchange (foreign h stays 0), 2 library expf / tanhf / division, 4 no gate math
template <int CTRL>
__device__ __forceinline__ float gru_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
typedef float gru_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gru_rcp(float d) {   // d in [1, 2^120]: v_rcp_f32 (1 ulp) + one Newton step
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.f), r);
}
template <bool LIB>
__device__ __forceinline__ float gru_sigmoid(float x) {
    if (LIB) return 1.f / (1.f + expf(-x));
    return gru_rcp(1.f + expf(fminf(-x, 80.f)));
}
template <bool LIB>
__device__ __forceinline__ float gru_tanh(float x) {
    if (LIB) return tanhf(x);
    // 1 - 2 / (e^{2|x|} + 1): absolute error ~1 ulp of 1, which is what h = (1 - z) n + z h keeps of it
    const float t = gru_rcp(1.f + expf(fminf(2.f * fabsf(x), 80.f)));
    return copysignf(fmaf(-2.f, t, 1.f), x);
}

template <int NWG, int DBG>
__global__ void __launch_bounds__(256)
bigru_x_kernel(const float *__restrict__ gi, const float *__restrict__ whhT, const float *__restrict__ bhh,
               float *__restrict__ out, u64 *__restrict__ xchg, int *__restrict__ status, int T, unsigned spin_limit) {
    constexpr int UNITS = GRU_H / NWG, KSL = NWG, KPT = UNITS, HST = KPT + 4;
    constexpr bool LIB = (DBG & 2) != 0;
    static_assert(UNITS * KSL == 256 && KPT % 4 == 0, "");
    constexpr int GCH = 16;
    __shared__ __attribute__((aligned(16))) float h_s[2][KSL * HST];
    __shared__ float g_s[2][GCH * 3 * UNITS];
    __shared__ int dead_s;
    const int tid = threadIdx.x;
    // the NWG workgroups of a direction on one XCD: ids 8 apart; grid.x = 8 NWG, id % 8 = direction, id / 8 = slice
    if ((blockIdx.x & 7) >= 2) return;
    const int dir = blockIdx.x & 7, c = blockIdx.x >> 3, b = blockIdx.y;
    for (int i = tid; i < 2 * KSL * HST; i += 256) (&h_s[0][0])[i] = 0.f;
    if (tid == 0) dead_s = 0;
    u64 *xb = xchg + ((size_t)b * 2 + dir) * 2 * GRU_H;   // [parity][256]
    const int ks = tid % KSL, ul = tid / KSL;
    const int u = c * UNITS + ul;
    const float *W = whhT + (size_t)dir * GRU_H * GRU_G + (size_t)(ks * KPT) * GRU_G + u;
    gru_f32x2 wr[KPT / 2], wz[KPT / 2], wn[KPT / 2];
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) {
        wr[i] = gru_f32x2{W[(size_t)(2 * i) * GRU_G], W[(size_t)(2 * i + 1) * GRU_G]};
        wz[i] = gru_f32x2{W[(size_t)(2 * i) * GRU_G + GRU_H], W[(size_t)(2 * i + 1) * GRU_G + GRU_H]};
        wn[i] = gru_f32x2{W[(size_t)(2 * i) * GRU_G + 2 * GRU_H], W[(size_t)(2 * i + 1) * GRU_G + 2 * GRU_H]};
    }
    float *outb = out + (size_t)b * T * 2 * GRU_H + dir * GRU_H + u;
    const float b_r = bhh[dir * GRU_G + u], b_z = bhh[dir * GRU_G + GRU_H + u], b_n = bhh[dir * GRU_G + 2 * GRU_H + u];
    const int own_dst = (u / KPT) * HST + u % KPT;
    // the foreign unit this thread fetches every step (threads >= 256 - UNITS: none)
    const bool poller = tid < GRU_H - UNITS;
    const int fu = (c * UNITS + UNITS + tid) % GRU_H;
    const int fdst = (fu / KPT) * HST + fu % KPT;
    float hold = 0.f;
    int t = dir == 0 ? 0 : T - 1;
    const int dt = dir == 0 ? 1 : -1;
    // The input gates W_ih x_t + b_ih of this workgroup's units come from HBM (3 UNITS floats per step).  A load issued in
    // step s and used in step s + 1 costs its full miss latency every step (the granule polls wait for every older load), so
    // they are fetched GCH steps at a time into registers, one chunk ahead, and parked in LDS at the chunk boundary.
    constexpr int GPT = GCH * 3 * UNITS / 256;              // values per thread per chunk
    static_assert(GCH * 3 * UNITS % 256 == 0, "");
    float gq[GPT];
    auto g_load = [&](int s0) __attribute__((always_inline)) {          // chunk of steps [s0, s0 + GCH)
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + 256 * i;                    // (step, gate, unit)
            const int ss = e / (3 * UNITS), g = (e / UNITS) % 3, uu = e % UNITS;
            const int sa = s0 + ss;
            const int ta = dir == 0 ? sa : T - 1 - sa;
            gq[i] = sa < T ? gi[((size_t)b * T + ta) * 2 * GRU_G + (size_t)dir * GRU_G + g * GRU_H + c * UNITS + uu] : 0.f;
        }
    };
    auto g_park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) g_s[buf][tid + 256 * i] = gq[i];
    };
    g_load(0);
    g_park(0);
    g_load(GCH);
    __syncthreads();
    for (int s = 0; s < T; ++s, t += dt) {
        const int par = s & 1;
        const float *gs = &g_s[(s / GCH) & 1][(s % GCH) * 3 * UNITS + ul];
        const float gr = gs[0], gz = gs[UNITS], gn = gs[2 * UNITS];
        const float4 *hq = reinterpret_cast<const float4 *>(&h_s[par][ks * HST]);
        gru_f32x2 ar = {0.f, 0.f}, az = {0.f, 0.f}, an = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPT / 4; ++i) {
            const float4 hv = hq[i];
            const gru_f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
            ar = __builtin_elementwise_fma(wr[2 * i], h01, ar); az = __builtin_elementwise_fma(wz[2 * i], h01, az); an = __builtin_elementwise_fma(wn[2 * i], h01, an);
            ar = __builtin_elementwise_fma(wr[2 * i + 1], h23, ar); az = __builtin_elementwise_fma(wz[2 * i + 1], h23, az); an = __builtin_elementwise_fma(wn[2 * i + 1], h23, an);
        }
        float sr = ar.x + ar.y, sz = az.x + az.y, sn = an.x + an.y;
        // sum over the KSL lanes of the unit (every lane ends with the total)
        sr += gru_dpp<0xB1>(sr); sz += gru_dpp<0xB1>(sz); sn += gru_dpp<0xB1>(sn);        // lane ^ 1
        sr += gru_dpp<0x4E>(sr); sz += gru_dpp<0x4E>(sz); sn += gru_dpp<0x4E>(sn);        // lane ^ 2
        if (KSL >= 8) { sr += gru_dpp<0x141>(sr); sz += gru_dpp<0x141>(sz); sn += gru_dpp<0x141>(sn); }   // 7 - lane within 8
        if (KSL >= 16) { sr += gru_dpp<0x140>(sr); sz += gru_dpp<0x140>(sz); sn += gru_dpp<0x140>(sn); }  // 15 - lane within 16
        float hnew;
        if (DBG & 4) {
            hnew = 0.001f * (sr + sz + sn) + gr;
        } else {
            // sigma(r) in even lanes, sigma(z) in odd lanes, swapped through the quad
            const bool odd = ks & 1;
            const float sg = gru_sigmoid<LIB>(odd ? gz + (sz + b_z) : gr + (sr + b_r));
            const float so = gru_dpp<0xB1>(sg);
            const float rr = odd ? so : sg, zz = odd ? sg : so;
            const float nn = gru_tanh<LIB>(gn + rr * (sn + b_n));
            hnew = (1.f - zz) * nn + zz * hold;
        }
        hold = hnew;
        const unsigned epoch = (unsigned)s + 1u;
        u64 *xp = xb + (size_t)par * GRU_H;
        if (ks == 0) {
            const u64 gran = ((u64)epoch << 32) | (u64)__float_as_uint(hnew);
            __hip_atomic_store(xp + u, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h_s[par ^ 1][own_dst] = hnew;
            outb[(size_t)t * 2 * GRU_H] = hnew;
        }
        if (poller && !(DBG & 1) && !dead_s) {
            unsigned spins = 0;
            for (;;) {
                const u64 x = __hip_atomic_load(xp + fu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(x >> 32) == epoch) { h_s[par ^ 1][fdst] = __uint_as_float((unsigned)x); break; }
                if (++spins > spin_limit) { dead_s = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (s % GCH == GCH - 1) {      // the next chunk into the other buffer (last read GCH steps ago), the one after it into flight
            g_park(((s / GCH) + 1) & 1);
            g_load((s / GCH + 2) * GCH);
        }
        __syncthreads();
    }
    if (dead_s && tid == 0) atomicAdd(&status[2 * b + dir], 1);
}

template <int NWG, int DBG>
static void bigru_x_launch(const float *gi, const float *whhT, const float *bhh, float *out, u64 *xchg, int *status, int T, int batch,
                           unsigned spin, hipStream_t stream) {
    hipLaunchKernelGGL((bigru_x_kernel<NWG, DBG>), dim3(8 * NWG, batch), dim3(256), 0, stream, gi, whhT, bhh, out, xchg, status, T, spin);
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
    if (batch * 2 * GRU_NWG > 128) return fail("rvc_bigru_forward: batch %d needs %d co-resident workgroups (max 128)", batch, batch * 2 * GRU_NWG);
    RVC_HIP(hipMemsetAsync(workspace_dev, 0, need, (hipStream_t)stream));  // tags must start below epoch 1 on every call; status = 0
    int *status = (int *)((char *)workspace_dev + bigru_xchg_bytes(batch));
    static const int dbg = knob("RVC_GRU_DBG", 0);     // ablations, tools/bench_gru.py
    const unsigned spin = g_spin_limit.load(std::memory_order_relaxed);
    hipStream_t st = (hipStream_t)stream;
    u64 *xg = (u64 *)workspace_dev;
    const int Ti = (int)n_steps;
    switch (dbg) {
#ifdef RVC_ABLATE
        case 1: bigru_x_launch<GRU_NWG, 1>(gi_dev, whhT_dev, bhh_dev, out_dev, xg, status, Ti, batch, spin, st); break;
        case 2: bigru_x_launch<GRU_NWG, 2>(gi_dev, whhT_dev, bhh_dev, out_dev, xg, status, Ti, batch, spin, st); break;
        case 4: bigru_x_launch<GRU_NWG, 4>(gi_dev, whhT_dev, bhh_dev, out_dev, xg, status, Ti, batch, spin, st); break;
        case 5: bigru_x_launch<GRU_NWG, 5>(gi_dev, whhT_dev, bhh_dev, out_dev, xg, status, Ti, batch, spin, st); break;
#endif
        default: bigru_x_launch<GRU_NWG, 0>(gi_dev, whhT_dev, bhh_dev, out_dev, xg, status, Ti, batch, spin, st); break;
    }
    RVC_LAUNCH_CHECK();
    // 2 x batch workgroups that return at once unless their sequence's rendezvous timed out (see bigru_x_kernel)
    hipLaunchKernelGGL(bigru_kernel, dim3(2, batch), dim3(GRU_THREADS), 0, (hipStream_t)stream, gi_dev, whhT_dev, bhh_dev,
                       out_dev, (int)n_steps, (const int *)status);
    RVC_LAUNCH_CHECK();
    return 0;
}
