// HBM rate of mixed read / write streams on MI355X (hipcc --offload-arch=gfx950 -O3 -o hbm_mix hbm_mix.hip).
// Question behind it: the extractor's residual 1x1 expansions move 1 part activations + residual in, 1 part out - and sit at
// 5.2-5.5 TB/s whatever is done to their loads in flight. Is that the kernel, or what HBM3E gives a 2-reads : 1-write mix?
//   y = a            (read only, reduced)     y = copy(a)         (1 R : 1 W)      y = a + b      (2 R : 1 W)      y = a + b + c  (3 R : 1 W)
// 2048 blocks x 256 threads, grid-stride, 4 x 16 B per lane in flight per stream, non-temporal loads and stores, 1 GiB per stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NR, bool WR>
__global__ __launch_bounds__(256) void mix_kernel(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *y, float *sink, size_t n) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i + 768 < n; i += stride) {
        f32x4 v[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            v[d] = __builtin_nontemporal_load(a + i + d * 256);
            if (NR > 1) v[d] += __builtin_nontemporal_load(b + i + d * 256);
            if (NR > 2) v[d] += __builtin_nontemporal_load(c + i + d * 256);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (WR) __builtin_nontemporal_store(v[d], y + i + d * 256);
            else acc += v[d];
        }
    }
    if (!WR && acc[0] == 123.456f) sink[blockIdx.x] = acc[1];
}
static double g_secs = 0.0;          // > 0: every arm runs for about that long (a power sampler beside it needs seconds, not 10 launches)
template <int NR, bool WR> static void run(const f32x4 *a, const f32x4 *b, const f32x4 *c, f32x4 *y, float *sink, size_t n, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix_kernel<NR, WR>), dim3(2048), dim3(256), 0, 0, a, b, c, y, sink, n);
    hipDeviceSynchronize();
    int it = 10;
    if (g_secs > 0.0) {
        hipEventRecord(e0); hipLaunchKernelGGL((mix_kernel<NR, WR>), dim3(2048), dim3(256), 0, 0, a, b, c, y, sink, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float one; hipEventElapsedTime(&one, e0, e1);
        it = (int)(g_secs * 1e3 / one) + 1;
    }
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((mix_kernel<NR, WR>), dim3(2048), dim3(256), 0, 0, a, b, c, y, sink, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    const double bytes = (double)n * 16 * (NR + (WR ? 1 : 0));
    printf("%-28s %7.3f ms  %.2f TB/s\n", what, ms, bytes / ms * 1e-9);
}
__global__ void fill_random(float *p, size_t n, unsigned seed) {          // N(0,1)-like magnitudes with all mantissa bits toggling (zeros draw less power)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
        p[i] = (float)(int)x * (1.0f / 1073741824.0f);
    }
}
// usage: hbm_mix [seconds per arm (0 = 10 launches)] [arm 0..4, -1 = all] [pattern: 0 = zeros (round 4's numbers), 1 = random]
int main(int argc, char **argv) {
    g_secs = argc > 1 ? atof(argv[1]) : 0.0;
    const int only = argc > 2 ? atoi(argv[2]) : -1, pat = argc > 3 ? atoi(argv[3]) : 0;
    const size_t n = (size_t)1 << 26;                    // 2^26 x 16 B = 1 GiB per stream
    f32x4 *a, *b, *c, *y; float *sink;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&y, n * 16); hipMalloc(&sink, 2048 * 4);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16); hipMemset(y, 0, n * 16);
    if (pat) {
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *)a, n * 4, 1u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *)b, n * 4, 2u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *)c, n * 4, 3u);
        hipDeviceSynchronize();
    }
    printf("data: %s\n", pat ? "random fp32" : "zeros");
    if (only < 0 || only == 0) run<1, false>(a, b, c, y, sink, n, "read only (1 stream)");
    if (only < 0 || only == 1) run<2, false>(a, b, c, y, sink, n, "read only (2 streams)");
    if (only < 0 || only == 2) run<1, true>(a, b, c, y, sink, n, "copy        1 R : 1 W");
    if (only < 0 || only == 3) run<2, true>(a, b, c, y, sink, n, "y = a + b   2 R : 1 W");
    if (only < 0 || only == 4) run<3, true>(a, b, c, y, sink, n, "y = a+b+c   3 R : 1 W");
    return 0;
}
