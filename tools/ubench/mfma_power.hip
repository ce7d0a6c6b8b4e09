// What the socket's power cap leaves of the matrix pipe (hipcc --offload-arch=gfx950 -O3). 256 workgroups x 8 waves (2 per SIMD); every wave issues
// v_mfma_f32_32x32x16_f16 back to back on 8 independent accumulators, and the arms add, at the ratios of the h2 GEMM main loop (gemm_h2.inc: per
// 24 MFMAs 12 ds_read_b128, per 48 MFMAs 8 16-byte global loads and ~64 VALU operations per lane), the work that surrounds the MFMAs there:
//   arm 0  MFMAs only, register operands          arm 1  + LDS fragment reads (operands re-read from a 64 KB LDS image)
//   arm 2  + global loads (a 2 GB buffer, HBM)    arm 3  + VALU (fp32 -> two fp16 pieces arithmetic on the loaded values)
// DESIGN.md 4 "Power is the wall": the fused step holds the socket at its 1400 W cap; this prints, per arm and operand pattern, the sustained
// TFLOP/s and the shader clock (s_memtime / s_memrealtime). Run rocm-smi beside it for the power.
//   mfma_power <seconds per arm> [pattern: 0 random (default), 1 zeros]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ARM>
__global__ __launch_bounds__(512, 2) void mfma_loop(const h16x8 *__restrict__ opnd, const f32x4 *__restrict__ big, size_t big_vec, float *out,
                                                     unsigned long long *clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    h16x8 *img = reinterpret_cast<h16x8 *>(lds);                       // 64 KB = 4096 fragment entries
    for (int i = tid; i < 4096; i += 512) img[i] = opnd[i & 2047];
    __syncthreads();
    h16x8 a0 = opnd[tid], a1 = opnd[512 + tid], b0 = opnd[1024 + tid], b1 = opnd[1536 + tid];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 *gp = big + ((size_t)blockIdx.x * 512 + tid);
    const size_t gstride = (size_t)256 * 512;
    size_t goff = 0;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        f32x4 g[4];
        if (ARM >= 2) {                                                   // 4 x 16 B per 24 MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q) { g[q] = __builtin_nontemporal_load(gp + goff); goff += gstride; if (goff + gstride > big_vec) goff = 0; }
        }
#pragma unroll
        for (int rep = 0; rep < 3; ++rep) {                               // 24 MFMAs, 12 fragment reads
            h16x8 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (ARM >= 1) {
                const int e = (tid + (it * 3 + rep) * 512) & 4095;
                na0 = img[e]; na1 = img[(e + 512) & 4095]; nb0 = img[(e + 1024) & 4095]; nb1 = img[(e + 1536) & 4095];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[7], 0, 0, 0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (ARM >= 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = g[q];
                if (ARM >= 3) {                                           // 16 values: h = rn16(x), m = rn16(x - h), folded back (8 VALU each way)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const _Float16 h = (_Float16)v[e];
                        const _Float16 m = (_Float16)(v[e] - (float)h);
                        v[e] = (float)h + (float)m;
                    }
                }
                gsum += v;
            }
        }
    }
    asm volatile("s_nop 15\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
    float s = gsum[0] + gsum[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[blockIdx.x] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int ARM>
static void run(const char *name, const h16x8 *opnd, const f32x4 *big, size_t big_vec, float *out, unsigned long long *clk, double secs) {
    static unsigned long long hc[512];
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_loop<ARM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<ARM>, dim3(256), dim3(512), 65536, 0, opnd, big, big_vec, out, clk, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop<ARM>, dim3(256), dim3(512), 65536, 0, opnd, big, big_vec, out, clk, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * (secs * 1e3 / ms));
    (void)hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop<ARM>, dim3(256), dim3(512), 65536, 0, opnd, big, big_vec, out, clk, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    double cmin = 1e30, cmax = 0;
    for (int b = 0; b < 256; ++b) { const double mhz = (double)hc[2 * b] / ((double)hc[2 * b + 1] / 100.0); cmin = mhz < cmin ? mhz : cmin; cmax = mhz > cmax ? mhz : cmax; }
    const double flops = 256.0 * 8 * (double)iters * 24 * (2.0 * 32 * 32 * 16);
    printf("arm %d %-34s: %5.0f ms, %5.0f TFLOP/s (%.2f of 2500), shader clock %.0f-%.0f MHz\n", ARM, name, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0, cmin, cmax);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const int pat = argc > 2 ? atoi(argv[2]) : 0;
    const int only = argc > 3 ? atoi(argv[3]) : -1;              // run ONE arm (for a power sampler beside it: tools/power_arms.sh); -1 = all four
    h16x8 *opnd; float *out; unsigned long long *clk; f32x4 *big;
    const size_t big_bytes = (size_t)2 << 30;
    (void)hipMalloc(&opnd, 2048 * sizeof(h16x8)); (void)hipMalloc(&out, 1024); (void)hipMalloc(&clk, 512 * 8); (void)hipMalloc(&big, big_bytes);
    static _Float16 host[2048 * 8];
    unsigned s = 12345u;
    for (int i = 0; i < 2048 * 8; ++i) {
        s = s * 1664525u + 1013904223u;
        host[i] = (_Float16)(pat == 0 ? (float)(s >> 8) / 8388608.f - 1.f : 0.f);
    }
    (void)hipMemcpy(opnd, host, sizeof(host), hipMemcpyHostToDevice);
    {   // the big buffer: random fp32 (or zeros)
        const size_t n = big_bytes / 4;
        float *hb = (float *)malloc(64u << 20);
        for (size_t i = 0; i < (64u << 20) / 4; ++i) { s = s * 1664525u + 1013904223u; hb[i] = pat == 0 ? (float)(s >> 8) / 8388608.f - 1.f : 0.f; }
        for (size_t off = 0; off < n * 4; off += 64u << 20) (void)hipMemcpy((char *)big + off, hb, 64u << 20, hipMemcpyHostToDevice);
        free(hb);
    }
    printf("operands: %s\n", pat == 0 ? "random in [-1, 1)" : "zeros");
    if (only < 0 || only == 0) run<0>("MFMAs only", opnd, big, big_bytes / 16, out, clk, secs);
    if (only < 0 || only == 1) run<1>("+ LDS fragment reads", opnd, big, big_bytes / 16, out, clk, secs);
    if (only < 0 || only == 2) run<2>("+ global loads (HBM)", opnd, big, big_bytes / 16, out, clk, secs);
    if (only < 0 || only == 3) run<3>("+ VALU two-piece arithmetic", opnd, big, big_bytes / 16, out, clk, secs);
    return 0;
}
