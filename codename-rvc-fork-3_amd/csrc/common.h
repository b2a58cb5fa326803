// Shared host/device helpers for librvc_amd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "rvc_amd.h"

namespace rvc {

// ---- error reporting across the C ABI -------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(const char *fmt, ...);  // sets the error, returns 1

#define RVC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return ::rvc::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                                 __FILE__, __LINE__);                         \
    } while (0)

#define RVC_LAUNCH_CHECK()                                                                    \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) return ::rvc::fail("kernel launch failed: %s (%s:%d)",          \
                                                 hipGetErrorString(_e), __FILE__, __LINE__);  \
    } while (0)

// ---- tuning / ablation switches --------------------------------------------------------------------
// The product library reads NO environment variable to pick a kernel: knob() folds to its default.  Only the ablation build
// (-DRVC_ABLATE: __graft_entry__.build_ablate() -> _lib/librvc_amd_ablate.so, loaded when RVC_AMD_LIB names it) reads the
// RVC_* switches the tools/ scripts set, and only that build contains the kernel instantiations that leave work out and
// therefore compute wrong results on purpose ("where does the time go" runs).
#ifdef RVC_ABLATE
static inline int knob(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
#else
static inline constexpr int knob(const char *, int dflt) { return dflt; }
#endif

// Every kernel that issues bf16 / fp16 matrix instructions REQUESTS this much dynamic LDS whatever it uses: it then never
// shares a CU with any other workgroup.  A workgroup of round 3's first gemmbf_kernel sharing a CU with a workgroup of the fp32
// Winograd kernel (wino.hip) made the latter return wrong accumulators (profiles/r04_mfma_cohabitation.txt: reproduced
// stand-alone, bare bf16 matrix loops do NOT do it, cause still not isolated), so co-residence of the two kernel families is
// excluded by construction instead of by the sizes their LDS layouts happen to have.
constexpr int LDS_WHOLE_CU = 163840;
// ... and this is how a launcher obtains it: once per kernel (registry in error.cpp), then `hipLaunchKernelGGL(k, grid, block, LDS_WHOLE_CU, ...)`.
// Returns non-zero and sets the error when the attribute cannot be set.
int reserve_whole_cu(const void *kernel, const char *what);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device-side MFMA typedefs ---------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32 fma chain in k order.
//   lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
//   D register r of lane l is element (row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release + acquire fence over ALL address spaces:
// hipcc lowers it to s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier, which drains the global prefetch a pipelined loop has just
// issued for a later chunk -- the loop then eats a full memory latency per chunk.  Use this one where the data exchanged
// between the waves went through LDS (ds_write, or LDS-DMA already retired with an explicit vmcnt wait).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

}  // namespace rvc
