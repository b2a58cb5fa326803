// How many vector-memory loads may one wave have in flight?  s_waitcnt's vmcnt field is 6 bits (0..63) on gfx950.
// One wave issues N buffer loads back to back from cold lines (each 4 KiB apart), then
//   mode 0: s_waitcnt vmcnt(0) by inline asm, then consumes all N values;
//   mode 1: lets hipcc place the waits (it models the counter as in-order with at most 63 outstanding).
// Every lane checks its N values against the pattern the host wrote.  Run alone and next to a kernel that saturates HBM
// (memory returns slow down, so more of the burst is in flight at once).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/vmcnt_burst.hip -o tools/micro/vmcnt_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

template <int N, int MODE, bool OOB>
__global__ void __launch_bounds__(64) burst(const unsigned *src, unsigned *bad, int words, int blocks_stride) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, words * 4, 0x00020000);
    const unsigned base = ((unsigned)blockIdx.x * blocks_stride + lane) * 4u;
    unsigned v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        unsigned off = base + (unsigned)i * 4096u;
        if (OOB && (i % 3) == 1) off = 0x80000000u;          // every third load wholly out of range: returns 0 without touching memory
        v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned wrong = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned want = (OOB && (i % 3) == 1) ? 0u : (((unsigned)blockIdx.x * blocks_stride + lane + (unsigned)i * 1024u) * 2654435761u);
        wrong += v[i] != want;
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ void hog(const float4 *p, float4 *q, size_t n) {   // HBM saturator
    float4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x == 123.f) q[0] = a;
}

template <int N, int MODE, bool OOB>
static void run(const unsigned *src, unsigned *bad, int words, int stride, int blocks, bool with_hog, const float4 *hp, float4 *hq, size_t hn, hipStream_t s1, hipStream_t s2) {
    unsigned total = 0;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemsetAsync(bad, 0, 4, s1));
        if (with_hog) hipLaunchKernelGGL(hog, dim3(2048), dim3(256), 0, s2, hp, hq, hn);
        hipLaunchKernelGGL((burst<N, MODE, OOB>), dim3(blocks), dim3(64), 0, s1, src, bad, words, stride);
        unsigned b = 0;
        CK(hipMemcpyAsync(&b, bad, 4, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        total += b;
    }
    printf("N = %3d loads in one burst, %s, %s, %s: %u wrong values in 20 launches of %d waves\n", N, MODE ? "hipcc's own waits" : "explicit vmcnt(0)",
           OOB ? "every third load out of range" : "all loads in range", with_hog ? "next to an HBM hog" : "alone", total, blocks);
}

int main() {
    const int blocks = 4096, stride = 64;                     // each block reads its own 64-word column of every 4 KiB row
    const int rows = 128;
    const size_t words = (size_t)rows * 1024 + (size_t)blocks * stride + 64;
    std::vector<unsigned> h(words);
    for (size_t i = 0; i < words; ++i) h[i] = (unsigned)i * 2654435761u;
    unsigned *src, *bad;
    CK(hipMalloc(&src, words * 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemcpy(src, h.data(), words * 4, hipMemcpyHostToDevice));
    const size_t hn = (size_t)1 << 26;                        // 1 GiB
    float4 *hp, *hq; CK(hipMalloc(&hp, hn * 16)); CK(hipMalloc(&hq, 64)); CK(hipMemset(hp, 0, hn * 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int hogging = 0; hogging < 2; ++hogging) {
        run<40, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<62, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<64, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<66, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<96, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<120, 0, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<96, 1, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<120, 1, false>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<48, 1, true>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<96, 0, true>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
        run<96, 1, true>(src, bad, (int)words, stride, blocks, hogging, hp, hq, hn, s1, s2);
    }
    return 0;
}
