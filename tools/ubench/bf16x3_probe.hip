// Probe for the split-bf16 GEMM idea: x = h + m + l (three bf16), products hh+hm+mh+mm+hl+lh on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//  (1) accuracy: one 32x32x(16*KS) tile vs fp64, against the exact-fp32 MFMA chain;
//  (2) rate: sustained loop of {split 48 floats in registers, 48 bf16 MFMAs} per 16-k step, 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {       // top 16 bits after round-to-nearest-even
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u & 0xFFFF0000u;
}
// split 8 floats into three bf16x8 planes (hi, mid, lo)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned hb = bf16_rn_bits(x[j]);
        const float r1 = x[j] - __builtin_bit_cast(float, hb);
        const unsigned mb = bf16_rn_bits(r1);
        const float r2 = r1 - __builtin_bit_cast(float, mb);
        const unsigned lb = bf16_rn_bits(r2);
        h[j] = (short)(hb >> 16); m[j] = (short)(mb >> 16); l[j] = (short)(lb >> 16);
    }
}
__device__ __forceinline__ f32x16 mma6(bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);     // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// accuracy: C[32,32] = A[32,K] B[32,K]^T, one wave
__global__ void acc_kernel(const float *A, const float *B, float *C6, float *C32, int K) {
    const int l = threadIdx.x, i = l & 31, hf = l >> 5;
    f32x16 c6, c32;
    for (int r = 0; r < 16; ++r) { c6[r] = 0.f; c32[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        float xa[8], xb[8];
        for (int j = 0; j < 8; ++j) { xa[j] = A[i * K + k0 + hf * 8 + j]; xb[j] = B[i * K + k0 + hf * 8 + j]; }
        bf16x8 ah, am, al, bh, bm, bl;
        split8(xa, ah, am, al); split8(xb, bh, bm, bl);
        c6 = mma6(ah, am, al, bh, bm, bl, c6);
    }
    for (int k0 = 0; k0 < K; k0 += 2)
        c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + hf], B[i * K + k0 + hf], c32, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        C6[row * 32 + i] = c6[r]; C32[row * 32 + i] = c32[r];
    }
}

// rate: per iteration = one 16-k step of a 64x128 wave tile: split A (2 sub-tiles) + B (4), 48 MFMAs
template <bool SPLIT>
__global__ __launch_bounds__(512) void rate_kernel(const float *in, float *out, int iters) {
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float xa[2][8], xb[4][8];
    for (int a = 0; a < 2; ++a) for (int j = 0; j < 8; ++j) xa[a][j] = in[threadIdx.x + 512 * (a * 8 + j)];
    for (int b = 0; b < 4; ++b) for (int j = 0; j < 8; ++j) xb[b][j] = in[8192 + threadIdx.x + 512 * (b * 8 + j)];
    bf16x8 ah[2], am[2], al[2], bh, bm, bl;
    if (!SPLIT) { for (int a = 0; a < 2; ++a) split8(xa[a], ah[a], am[a], al[a]); split8(xb[0], bh, bm, bl); }
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                split8(xa[a], ah[a], am[a], al[a]);
#pragma unroll
                for (int j = 0; j < 8; ++j) xa[a][j] += 1e-3f;          // keep the conversion loop-variant
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (SPLIT) {
                split8(xb[b], bh, bm, bl);
#pragma unroll
                for (int j = 0; j < 8; ++j) xb[b][j] -= 1e-3f;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a][b] = mma6(ah[a], am[a], al[a], bh, bm, bl, acc[a][b]);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int K = 1024;
    std::vector<float> hA(32 * K), hB(32 * K);
    for (auto &v : hA) v = ((rand() % 20001) - 10000) / 10000.0f * 3.f;
    for (auto &v : hB) v = ((rand() % 20001) - 10000) / 10000.0f * 0.05f;
    float *A, *B, *C6, *C32; hipMalloc(&A, 32 * K * 4); hipMalloc(&B, 32 * K * 4); hipMalloc(&C6, 4096); hipMalloc(&C32, 4096);
    hipMemcpy(A, hA.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), 32 * K * 4, hipMemcpyHostToDevice);
    acc_kernel<<<1, 64>>>(A, B, C6, C32, K);
    std::vector<float> c6(1024), c32(1024);
    hipMemcpy(c6.data(), C6, 4096, hipMemcpyDeviceToHost); hipMemcpy(c32.data(), C32, 4096, hipMemcpyDeviceToHost);
    double e6 = 0, e32 = 0, scale = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        double ref = 0, sab = 0;
        for (int k = 0; k < K; ++k) { ref += (double)hA[r * K + k] * hB[c * K + k]; sab += fabs((double)hA[r * K + k] * hB[c * K + k]); }
        e6 = fmax(e6, fabs(c6[r * 32 + c] - ref)); e32 = fmax(e32, fabs(c32[r * 32 + c] - ref)); scale = fmax(scale, sab);
    }
    printf("accuracy K=%d: max|err| split-bf16x6 %.3e   exact-fp32 MFMA %.3e   (sum|a.b| up to %.3e -> rel %.2e vs %.2e)\n", K, e6, e32, scale, e6 / scale, e32 / scale);

    float *in, *out; hipMalloc(&in, 16384 * 4 * 2); hipMalloc(&out, 256 * 2 * 512 * 4);
    std::vector<float> hin(16384 * 2); for (auto &v : hin) v = ((rand() % 2001) - 1000) / 1000.0f;
    hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int split = 0; split < 2; ++split) {
        const int iters = 20000, grid = 256;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (split) rate_kernel<true><<<grid, 512>>>(in, out, iters); else rate_kernel<false><<<grid, 512>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double eq_flops = (double)grid * 8 * iters * 8.0 * 2.0 * 32 * 32 * 16;   // fp32-equivalent FLOPs (8 tile-pairs per step)
        printf("rate %-22s: %.2f ms  -> %.1f TF/s fp32-equivalent (bf16 MFMA rate %.0f TF/s)\n", split ? "split in loop + 6 MFMA" : "6 MFMA only", ms,
               eq_flops / ms / 1e9, eq_flops * 6 / ms / 1e9);
    }
    return 0;
}
