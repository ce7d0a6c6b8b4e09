// Sustained v_mfma_f32_32x32x2_f32 issue-rate microbenchmark (registers only, no memory).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip ; run: ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x[4], y[4];
    for (int i = 0; i < 4; ++i) { x[i] = in[threadIdx.x + 256 * i]; y[i] = in[1024 + threadIdx.x + 256 * i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(s + a) & 3], y[(s * 3 + a) & 3], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(const char *name, int blocks_per_cu, float *out, float *in, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<NACC><<<grid, 256>>>(out, in, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * NACC * 4096.0;
    printf("%-28s acc=%d waves/SIMD=%d  %8.2f ms  %7.1f TF/s  (=%5.2f GHz-equivalent of 64 cyc/MFMA)\n", name, NACC,
           blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 2.4);
}

int main() {
    float *in, *out; hipMalloc(&in, 2048 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    float h[2048];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 2048; ++i) h[i] = mode ? ((rand() % 2001) - 1000) / 1000.0f : 0.f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        const char *nm = mode ? "random [-1,1)" : "zeros";
        run<4>(nm, 1, out, in, 20000);
        run<4>(nm, 2, out, in, 10000);
        run<2>(nm, 1, out, in, 40000);
        run<1>(nm, 1, out, in, 80000);
        run<4>(nm, 1, out, in, 200000);     // ~long: power steady state
    }
    return 0;
}
