// How many bytes per cycle can ONE CU pull out of its XCD's L2 into registers -- alone, and next to a stream of bf16 matrix
// instructions?  DESIGN.md section 4 ("the wall the bf16x3 kernels share") reads 14-17 B / cycle / CU out of three unrelated
// kernels (K3y, K3f, K12); this measures the same quantity with nothing else in the way, in the shape those kernels use:
// one 512-thread workgroup per CU that owns the CU (whole LDS), every wave streaming 1 KiB pieces (buffer_load_dwordx4, 16 B per
// lane) of a slab that ALL workgroups share (2 MiB: L2-resident, like a layer's tap fragments), `depth` pieces in flight per wave,
// and `mfma` v_mfma_f32_32x32x16_bf16 issued per piece (0 = loads only; K3y issues 4 per piece, K3f 4, K12 4.5).
//   hipcc --offload-arch=gfx950 -O3 -o l2_ingest l2_ingest.hip && ./l2_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MFMA, int WAVES>
__global__ void __launch_bounds__(512) ingest(const unsigned char *slab, int slab_bytes, int pieces, float *out, unsigned long long *cyc) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= WAVES) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)slab, 0, slab_bytes, 0x00020000);
    // wave w of block b walks the slab from its own offset (all waves of all blocks together sweep it again and again)
    unsigned off = ((blockIdx.x * 8 + wave) * 37u * 1024u) % (unsigned)slab_bytes;
    u32x4 ring[DEPTH];
    f32x16 acc[4] = {};
    u32x4 sum = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
        ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, (int)off, 0);
        off = off + 1024u < (unsigned)slab_bytes ? off + 1024u : 0u;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < pieces; p += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const u32x4 v = ring[i];
            ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, (int)off, 0);
            off = off + 1024u < (unsigned)slab_bytes ? off + 1024u : 0u;
            if constexpr (MFMA > 0) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int m = 0; m < MFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[m & 3], 0, 0, 0);
            } else {
                sum += v;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    s += (float)(sum.x ^ sum.y ^ sum.z ^ sum.w);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) s += (float)ring[i].x;
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH, int MFMA, int WAVES>
static void run(const unsigned char *slab, int slab_bytes, float *out, unsigned long long *cyc, int blocks) {
    const int pieces = 4096;
    auto k = ingest<DEPTH, MFMA, WAVES>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 163840, 0, slab, slab_bytes, pieces, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= blocks;
    const double bytes_cu = (double)WAVES * pieces * 1024.0;
    printf("waves %d depth %2d mfma/piece %d: %8.1f us  %6.1f B/clk/CU by the wave's own cycle counter (%.0f cycles), %6.2f TB/s chip-wide by the wall clock"
           ", matrix instructions at %.0f %% of one per 8 cycles per CU\n",
           WAVES, DEPTH, MFMA, best * 1e3, bytes_cu / c, c, bytes_cu * blocks / (best * 1e-3) / 1e12,
           MFMA ? 100.0 * WAVES * pieces * MFMA * 8.0 / c : 0.0);
}

int main() {
    const int slab_bytes = 2 << 20, blocks = 256;
    unsigned char *slab; float *out; unsigned long long *cyc;
    hipMalloc(&slab, slab_bytes); hipMemset(slab, 0x3c, slab_bytes);
    hipMalloc(&out, 4096); hipMalloc(&cyc, blocks * 8);
    printf("2 MiB slab shared by 256 whole-CU workgroups (L2-resident), 1 KiB per wave-instruction\n");
    run<4, 0, 8>(slab, slab_bytes, out, cyc, blocks);
    run<8, 0, 8>(slab, slab_bytes, out, cyc, blocks);
    run<16, 0, 8>(slab, slab_bytes, out, cyc, blocks);
    run<16, 0, 4>(slab, slab_bytes, out, cyc, blocks);
    run<8, 2, 8>(slab, slab_bytes, out, cyc, blocks);
    run<8, 4, 8>(slab, slab_bytes, out, cyc, blocks);
    run<16, 4, 8>(slab, slab_bytes, out, cyc, blocks);
    run<8, 8, 8>(slab, slab_bytes, out, cyc, blocks);
    run<8, 4, 4>(slab, slab_bytes, out, cyc, blocks);
    run<16, 8, 4>(slab, slab_bytes, out, cyc, blocks);
    return 0;
}
