// Semantics check of __builtin_amdgcn_global_load_lds (16 B per lane) on gfx950: does lane l of a wave-instruction land at
// lds_base + 16 l, with a per-lane global source address?  Prints OK / the first mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *__restrict__ src, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane l fetches the 16-byte piece number perm(l) = (l & 7) ^ (l >> 3) of row (l >> 3) of this wave's 8 x 128 B block
    const int row = lane >> 3, piece = (lane & 7) ^ row;
    const unsigned char *g = reinterpret_cast<const unsigned char *>(src) + (size_t)wave * 1024 + row * 128 + piece * 16;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)g,
                                     (void __attribute__((address_space(3))) *)(smem + wave * 1152), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + wave * 1152 + lane * 16);
    for (int e = 0; e < 4; ++e) out[(threadIdx.x) * 4 + e] = v[e];
}
int main() {
    const int waves = 4, n = waves * 256;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, waves * 64 * 16);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(waves * 64), waves * 1152, 0, d, o);
    std::vector<unsigned> r(waves * 64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < waves; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const int row = l >> 3, piece = (l & 7) ^ row;
                const unsigned want = w * 256 + row * 32 + piece * 4 + e;
                if (r[(w * 64 + l) * 4 + e] != want) { printf("MISMATCH wave %d lane %d elem %d: got %u want %u\n", w, l, e, r[(w * 64 + l) * 4 + e], want); return 1; }
            }
    printf("glds 16 B: lane l lands at lds_base + 16 l with a per-lane source address: OK\n");
    return 0;
}
