// What this box's HBM delivers to a pure streaming read, for two access shapes over a 6 GB buffer:
//   linear : each block walks contiguous 16 KB pieces (float4 per lane, 4 loads in flight per lane per step)
//   rows128: the streaming-kNN shape -- 128 rows x 128 B per step, rows 3072 B apart
// hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PF>
__global__ __launch_bounds__(256) void linear(const f32x4* __restrict__ p, size_t n4, float* out) {
    // block b owns a contiguous span; per step 256 lanes x 4 float4 = 16 KB, PF steps in flight
    const size_t per_block = n4 / gridDim.x;
    const f32x4* q = p + per_block * blockIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[PF][4];
    const size_t steps = per_block / 1024;
    for (int s = 0; s < PF; ++s)
        for (int i = 0; i < 4; ++i) r[s][i] = q[(size_t)s * 1024 + i * 256 + threadIdx.x];
    for (size_t st = 0; st < steps; st += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += r[s][i];
            const size_t nx = st + s + PF;
            if (nx < steps)
#pragma unroll
                for (int i = 0; i < 4; ++i) r[s][i] = q[nx * 1024 + i * 256 + threadIdx.x];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

template <int PF>
__global__ __launch_bounds__(256) void rows128(const float* __restrict__ p, size_t n_rows, float* out) {
    const size_t rows_per_block = n_rows / gridDim.x / 128 * 128;
    const float* base = p + rows_per_block * blockIdx.x * 768;
    const int srow = threadIdx.x >> 3, sc4 = threadIdx.x & 7;
    const size_t n_chunks = rows_per_block / 128 * 24;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[PF][4];
    auto ld = [&](int s, size_t c) {
        const size_t tile = c / 24, kc = c - tile * 24;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r[s][i] = *reinterpret_cast<const f32x4*>(base + (tile * 128 + i * 32 + srow) * 768 + kc * 32 + sc4 * 4);
    };
    for (int s = 0; s < PF; ++s) ld(s, s);
    for (size_t c = 0; c < n_chunks; c += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += r[s][i];
            if (c + s + PF < n_chunks) ld(s, c + s + PF);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

// the direct-operand kNN shape: a wave owns 32 rows; lane (row i, half h) reads 4 x 16 B = bytes [64 h, 64 h + 64) of
// each 128-byte line group of its row; NB line groups in flight per wave
template <int NB, int NW>
__global__ __launch_bounds__(NW * 64) void rows32(const float* __restrict__ p, size_t n_rows, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const size_t rows_per_block = n_rows / gridDim.x / (32 * NW) * (32 * NW);
    const float* base = p + rows_per_block * blockIdx.x * 768;
    const size_t n_tiles = rows_per_block / 32 / NW;      // per wave
    const size_t n_groups = n_tiles * 24;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[NB][4];
    auto ld = [&](int s, size_t g) {
        const size_t t = g / 24, j = g - t * 24;
        const float* q = base + ((t * NW + wave) * 32 + i) * 768 + j * 32 + h * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[s][k] = *reinterpret_cast<const f32x4*>(q + 4 * k);
    };
    for (int s = 0; s < NB; ++s) ld(s, s);
    for (size_t g = 0; g < n_groups; g += NB) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += r[s][k];
            if (g + s + NB < n_groups) ld(s, g + s + NB);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

int main(int argc, char** argv) {
    const size_t n_rows = 2000000, bytes = n_rows * 768 * 4;
    float* buf; float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-28s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    time("rows32 NB=3 8 waves x256", [&] { rows32<3, 8><<<256, 512>>>(buf, n_rows, out); });
    time("rows32 NB=4 8 waves x256", [&] { rows32<4, 8><<<256, 512>>>(buf, n_rows, out); });
    time("rows32 NB=3 16 waves x256", [&] { rows32<3, 16><<<256, 1024>>>(buf, n_rows, out); });
    time("rows32 NB=3 4 waves x512", [&] { rows32<3, 4><<<512, 256>>>(buf, n_rows, out); });
    time("rows32 NB=3 4 waves x1024", [&] { rows32<3, 4><<<1024, 256>>>(buf, n_rows, out); });
    for (int bpc : {2}) {
        const int blocks = 256 * bpc;
        char nm[64];
        snprintf(nm, 64, "linear  PF=2 blocks=%d", blocks); time(nm, [&] { linear<2><<<blocks, 256>>>((const f32x4*)buf, bytes / 16, out); });
        snprintf(nm, 64, "linear  PF=4 blocks=%d", blocks); time(nm, [&] { linear<4><<<blocks, 256>>>((const f32x4*)buf, bytes / 16, out); });
        snprintf(nm, 64, "rows128 PF=2 blocks=%d", blocks); time(nm, [&] { rows128<2><<<blocks, 256>>>(buf, n_rows, out); });
        snprintf(nm, 64, "rows128 PF=4 blocks=%d", blocks); time(nm, [&] { rows128<4><<<blocks, 256>>>(buf, n_rows, out); });
    }
    return 0;
}
