// Sustained rate of v_mfma_f32_32x32x16_f16 when NOTHING else runs: 256 workgroups x 8 waves (2 per SIMD), each wave issues MFMAs back to back
// on 8 independent accumulators from register operands (hipcc --offload-arch=gfx950 -O3). Question behind it ("Power"): the fused
// step holds the socket at its 1400 W cap with the h2 GEMMs at 1.6-1.9 GHz; what clock - and therefore what TFLOP/s - does the matrix pipe
// itself sustain under that cap, with operands that look like ours (random fp16 pieces) and with zeros?
//   mfma_power <seconds per arm>      prints, per operand pattern: shader clock (s_memtime / s_memrealtime), TFLOP/s, fraction of 2.5 PFLOP/s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void mfma_loop(const h16x8 *__restrict__ opnd, float *out, unsigned long long *clk, int iters) {
    const int tid = threadIdx.x, wave = tid >> 6;
    h16x8 a0 = opnd[tid], a1 = opnd[512 + tid], b0 = opnd[1024 + tid], b1 = opnd[1536 + tid];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[7], 0, 0, 0);
        }
    }
    asm volatile("s_nop 15\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[blockIdx.x] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    (void)wave;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    h16x8 *opnd; float *out; unsigned long long *clk;
    hipMalloc(&opnd, 2048 * sizeof(h16x8)); hipMalloc(&out, 1024); hipMalloc(&clk, 512 * 8);
    static _Float16 host[2048 * 8];
    static unsigned long long hc[512];
    const char *names[3] = {"random fp16 in [-1, 1) (first pieces)", "random small fp16 (|x| < 2^-11, second pieces)", "zeros"};
    for (int pat = 0; pat < 3; ++pat) {
        unsigned s = 12345u;
        for (int i = 0; i < 2048 * 8; ++i) {
            s = s * 1664525u + 1013904223u;
            const float u = (float)(s >> 8) / 8388608.f - 1.f;
            host[i] = (_Float16)(pat == 0 ? u : pat == 1 ? u / 2048.f : 0.f);
        }
        hipMemcpy(opnd, host, sizeof(host), hipMemcpyHostToDevice);
        int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(512), 0, 0, opnd, out, clk, iters);
        hipDeviceSynchronize();
        // calibrate the iteration count to ~secs, then one timed launch (long enough for the power controller to settle)
        hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(512), 0, 0, opnd, out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        iters = (int)(iters * (secs * 1e3 / ms));
        hipEventRecord(e0); hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(512), 0, 0, opnd, out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
        double cmin = 1e30, cmax = 0;
        for (int b = 0; b < 256; ++b) { const double mhz = (double)hc[2 * b] / ((double)hc[2 * b + 1] / 100.0); cmin = mhz < cmin ? mhz : cmin; cmax = mhz > cmax ? mhz : cmax; }
        const double flops = 256.0 * 8 * (double)iters * 32 * (2.0 * 32 * 32 * 16);
        printf("%-48s: %.0f ms, %.0f TFLOP/s (%.2f of 2500), shader clock %.0f-%.0f MHz across workgroups\n", names[pat], ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0, cmin, cmax);
    }
    return 0;
}
