// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip with nothing else going on: the practical ceiling
// (clock under load) that the conv kernels' TFLOP/s should be read against.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void peak(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.f) out[threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int iters = 20000, blocks = 256 * (argc > 1 ? atoi(argv[1]) : 2);
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        peak<<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = double(blocks) * 4 * iters * 4.0 * 4096;
        printf("blocks %d: %.3f ms  %.1f TFLOP/s  (=> %.0f MHz if 256 CUs x 4 SIMDs x 64 flop/clk)\n", blocks, ms, fl / ms / 1e9,
               fl / ms / 1e3 / (256.0 * 4 * 64));
    }
    return 0;
}
