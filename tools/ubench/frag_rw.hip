// Can a kernel whose lanes each own ONE ROW stream a [M, 256] fp32 tensor in and out at HBM rate? (hipcc --offload-arch=gfx950 -O3 -o frag_rw frag_rw.hip)
// Question behind it ("what would reach 30k"): fusing a bottleneck's conv3 (+ residual) with the next block's conv1 without an LDS
// transpose needs the first product computed TRANSPOSED, so that its accumulators are already the second product's operand: lane (li, hi) then
// holds row li and, per 32-channel block b and register group jj, the four channels 32 b + 8 jj + 4 hi + 0..3. Residual loads and output stores
// become "fragment-shaped": one instruction touches 32 rows x 32 bytes instead of 4 rows x 256 bytes. The vector L1 serves such LOADS at about a
// quarter of its rate (one tag per lane; measured on the streamed 3x3, docs/HISTORY.md 10); nobody has measured the stores.
//   out[r, :] = relu(res[r, :] + 1)  over M = 2,097,152 rows (2.15 GB in, 2.15 GB out = layer 1 of the extractor at 512 tiles)
//   arm "rows"  : a wave owns 32 rows, lane (li, hi) row li, 32 x 16-byte loads issued up front, then 32 x 16-byte stores (the layout above)
//   arm "lines" : the same bytes with lanes along the channels (64 lanes x 16 B = one 1 KB row per instruction)
// One 256-thread workgroup per CU x WGS, persistent over row tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool ROWS>
__global__ __launch_bounds__(256) void rw_kernel(const float *__restrict__ res, float *__restrict__ out, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    const int wtiles = M / 32, nw = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < wtiles; t += nw) {
        f32x4 v[32];
        if (ROWS) {
            const float *p = res + (size_t)(t * 32 + li) * 256 + 4 * hi;
#pragma unroll
            for (int s = 0; s < 32; ++s) v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + 8 * s));
            float *q = out + (size_t)(t * 32 + li) * 256 + 4 * hi;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                f32x4 w = v[s] + 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __builtin_fmaxf(w[e], 0.f);
                __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(q + 8 * s));
            }
        } else {
            const float *p = res + (size_t)t * 32 * 256 + lane * 4;
#pragma unroll
            for (int s = 0; s < 32; ++s) v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + 256 * s));
            float *q = out + (size_t)t * 32 * 256 + lane * 4;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                f32x4 w = v[s] + 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __builtin_fmaxf(w[e], 0.f);
                __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(q + 256 * s));
            }
        }
    }
}
template <bool ROWS> static void run(const float *res, float *out, int M, int wgs, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rw_kernel<ROWS>), dim3(256 * wgs), dim3(256), 0, 0, res, out, M);
    hipDeviceSynchronize();
    const int it = 10;
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((rw_kernel<ROWS>), dim3(256 * wgs), dim3(256), 0, 0, res, out, M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-44s %d workgroup(s) per CU  %7.3f ms  %.2f TB/s\n", what, wgs, ms, 2.0 * M * 1024 / ms * 1e-9);
}
int main() {
    const int M = 2097152;
    float *res, *out;
    hipMalloc(&res, (size_t)M * 1024); hipMalloc(&out, (size_t)M * 1024);
    hipMemset(res, 0x3c, (size_t)M * 1024);
    for (int wgs : {1, 2, 4}) {
        run<true>(res, out, M, wgs, "lane = row (32 rows x 32 B per instruction)");
        run<false>(res, out, M, wgs, "lane = 16 B of a row (1 KB per instruction)");
    }
    return 0;
}
