// Per-CU fetch rate of L2-resident and HBM-resident streams as a function of loads in flight per lane (hipcc --offload-arch=gfx950 -O3).
// Question behind it ("Operand feed"): the h2 GEMMs plateau at ~10.5 B/clk/CU of operand fetches - is that a ceiling of the
// vector-memory path, or only of ONE 64 KB stage in flight per CU?
//   l2_stream <mode> : mode 0 = all CUs of an XCD re-read the same 2 MB (L2 hits, like the weight planes), mode 1 = every block streams its own
//   region of a 2 GB buffer (HBM). One 512-thread block per CU (256 blocks), DEPTH x 16-byte loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const f32x4 *__restrict__ src, float *out, size_t region_vec, size_t block_stride_vec, int passes) {
    const f32x4 *base = src + (size_t)(blockIdx.x % 8) * 0 + (size_t)blockIdx.x * block_stride_vec;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < passes; ++p) {
        for (size_t i = threadIdx.x; i + (size_t)(DEPTH - 1) * 512 < region_vec; i += (size_t)DEPTH * 512) {
            f32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(base + i + (size_t)d * 512);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d];
        }
    }
    if (acc[0] == 123.456f) out[blockIdx.x] = acc[1];
}
template <int DEPTH>
static void run(const f32x4 *buf, float *out, int mode, size_t region_bytes) {
    const size_t region_vec = region_bytes / 16;
    const size_t stride = mode == 0 ? 0 : region_vec;            // mode 0: every block reads the same region (XCD-local L2 copy each)
    const int passes = mode == 0 ? 64 : 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(256), dim3(512), 0, 0, buf, out, region_vec, stride, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(256), dim3(512), 0, 0, buf, out, region_vec, stride, passes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * passes * (double)region_bytes;
    printf("mode %d depth %2d: %.3f ms, %.2f TB/s, %.1f GB/s per CU (= %.1f B/clk/CU at 2.0 GHz)\n", mode, DEPTH, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256,
           bytes / ms * 1e-6 / 256 / 2.0);
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t region = mode == 0 ? (2u << 20) : (8u << 20);
    f32x4 *buf; float *out;
    hipMalloc(&buf, mode == 0 ? region : region * 256); hipMalloc(&out, 4096);
    hipMemset(buf, 0, mode == 0 ? region : region * 256);
    run<1>(buf, out, mode, region); run<2>(buf, out, mode, region); run<4>(buf, out, mode, region); run<8>(buf, out, mode, region); run<16>(buf, out, mode, region);
    return 0;
}
