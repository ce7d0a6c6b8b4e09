// What limits a GEMM epilogue's store tail on MI355X - the chip's write bandwidth or each CU's own store path?
// (hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip)
// Every workgroup (512 threads = 8 waves, one per CU: 128 KB of dynamic LDS) writes 256 x 256 fp32 tiles of a row-major [rows][512] matrix
// exactly like gemm_nt_h2_big_kernel's epilogue does: a wave instruction stores 16 B per lane, lanes 0-31 one 512-byte row segment, lanes
// 32-63 the segment four rows below; 32 such instructions per wave and tile, non-temporal. The store phase of each tile is bracketed with
// s_memtime (after s_waitcnt vmcnt(0)); between tiles the waves spin for `gap` cycles (the "main loop").
//   arm A: G workgroups active (G = 256, 128, 64, 32): per-tile store cycles vs G. Chip-bound -> falls with G; CU-bound -> constant.
//   arm B: 256 workgroups, waves 4-7 store their half of the tile while waves 0-3 run back-to-back MFMAs (and vice versa on the next tile):
//          do stores and the matrix pipe overlap inside one CU, and what does each cost the other?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }

template <int MODE>     // 0: all 8 waves store; 1: waves 4-7 store while 0-3 run MFMAs; 2: MFMAs only on waves 0-3 (reference for arm B)
__global__ __launch_bounds__(512, 2) void store_kernel(float *C, int ldc, int tiles_per_wg, int gap, long long *stamps, float *sink) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
    f32x16 acc[2] = {};
    h16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(0.001f * (lane + e)); fb[e] = (_Float16)(0.002f * (lane - e)); }
    long long t_store = 0, t_mfma = 0;
    int n_mfma = 0;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long long tile = (long long)blockIdx.x * tiles_per_wg + t;
        float *base = C + tile * 256 * (long long)ldc + (long long)(wm * 64 + 4 * hi) * ldc + wn * 128 + 4 * li;   // tile = 256 rows x first 256 columns
        // "main loop": spin
        const long long t0 = now();
        while (now() - t0 < gap) __builtin_amdgcn_s_sleep(4);
        __syncthreads();
        const bool storing = MODE == 0 || (MODE == 1 && ((wave >> 2) == ((t & 1) ^ 1)));
        const bool mfma = MODE != 0 && ((wave >> 2) == (t & 1));
        const long long s0 = now();
        if (storing) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int urow = a * 32 + (r & 3) + 8 * (r >> 2);
                    f32x4 v = {(float)t, (float)r, (float)lane, (float)a};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(base + (long long)urow * ldc));
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_store += now() - s0;
        }
        if (mfma) {
            // as many MFMAs as the store phase of the partner group lasts is not knowable here: run a fixed 512 (= 16.4k pipe cycles per SIMD)
            for (int i = 0; i < 256; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc[1], 0, 0, 0);
            }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            t_mfma += now() - s0; n_mfma += 512;
        }
        __syncthreads();
    }
    if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t_store; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = t_mfma; }
    if (acc[0][0] + acc[1][3] == 123.456f) sink[0] = acc[0][1];
}

template <int MODE> static void run(int G, int tiles, int gap, float *C, long long *stamps, float *sink, const char *what) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(store_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_kernel<MODE>, dim3(G), dim3(512), 128 * 1024, 0, C, 512, tiles, gap, stamps, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(store_kernel<MODE>, dim3(G), dim3(512), 128 * 1024, 0, C, 512, tiles, gap, stamps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(G * 16);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> st, mf;
    for (int i = 0; i < G * 8; ++i) { if (h[2 * i]) st.push_back((double)h[2 * i]); if (h[2 * i + 1]) mf.push_back((double)h[2 * i + 1]); }
    std::sort(st.begin(), st.end()); std::sort(mf.begin(), mf.end());
    const int per_wave_tiles = MODE == 1 ? tiles / 2 : tiles;
    printf("%-46s G=%3d tiles=%d gap=%6d: %8.1f us", what, G, tiles, gap, ms * 1e3);
    if (!st.empty()) printf(" | store phase per tile (cycles, per wave incl. vmcnt(0)): min %7.0f med %7.0f max %7.0f", st.front() / per_wave_tiles,
                            st[st.size() / 2] / per_wave_tiles, st.back() / per_wave_tiles);
    if (!mf.empty()) printf(" | 512 MFMAs (cycles): min %7.0f med %7.0f max %7.0f", mf.front() / per_wave_tiles, mf[mf.size() / 2] / per_wave_tiles,
                            mf.back() / per_wave_tiles);
    printf("\n");
}

int main() {
    const int tiles = 6;
    float *C, *sink; long long *stamps;
    hipMalloc(&C, (size_t)256 * tiles * 256 * 512 * 4); hipMalloc(&sink, 64); hipMalloc(&stamps, 256 * 16 * 8);
    hipMemset(C, 0, (size_t)256 * tiles * 256 * 512 * 4);
    for (int gap : {60000, 0}) {
        for (int G : {256, 128, 64, 32, 8}) run<0>(G, tiles, gap, C, stamps, sink, "A: all 8 waves store a 256x256 fp32 tile");
    }
    run<2>(256, tiles, 60000, C, stamps, sink, "B0: waves of one group run 512 MFMAs alone");
    run<1>(256, tiles, 60000, C, stamps, sink, "B1: one group stores while the other computes");
    run<1>(32, tiles, 60000, C, stamps, sink, "B1: one group stores while the other computes");
    return 0;
}
