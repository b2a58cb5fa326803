// float64 VALU issue interval and dependent latency for ONE wave (what bounds the filtfilt recurrence):
// N independent chains of v_mul_f64 / v_add_f64 (no FMA contraction), timed with wall_clock64 around a long loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
template <int CH>
__global__ void chains(double* out, int iters, double a, double b) {
    double v[CH];
    for (int i = 0; i < CH; ++i) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = v[i] * a + b;   // mul then dependent add (two instructions)
    }
    double s = 0;
    for (int i = 0; i < CH; ++i) s += v[i];
    out[threadIdx.x] = s;
}
int main() {
    double* out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](auto kern, int ch, int waves) {
        const int iters = 200000;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); kern<<<1, 64 * waves>>>(out, iters, 0.999999, 1e-7); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double ns_per_op = best * 1e6 / (double(iters) * ch * 2);
        printf("chains=%d waves/block=%d: %.2f ns per f64 op per wave (%.1f cycles @2.4GHz)\n", ch, waves, ns_per_op, ns_per_op * 2.4);
    };
    run(chains<1>, 1, 1); run(chains<2>, 2, 1); run(chains<4>, 4, 1); run(chains<8>, 8, 1); run(chains<16>, 16, 1);
    run(chains<8>, 8, 4); run(chains<8>, 8, 8);
    return 0;
}
