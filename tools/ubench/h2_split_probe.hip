// Probe: which instructions hipcc emits for the fp16 two-piece operand split (csrc/gemm_h2.inc, h2_split2). Compile with
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -save-temps -c h2_split_probe.hip
// and read the .s: v_mul + v_cvt_pk_f16_f32 for h, v_fma_mixlo/mixhi_f16 (op_sel picks the fp16 half) for m.
#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// variant A: opaque barrier on the packed h pair
__device__ __forceinline__ void split8(const f32x4 &x0, const f32x4 &x1, float s, h8 &h, h8 &m) {
    u32x4 hp, mp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = e < 2 ? x0[2 * e] : x1[2 * e - 4], b = e < 2 ? x0[2 * e + 1] : x1[2 * e - 3];
        h2 hv; hv[0] = (_Float16)(a * s); hv[1] = (_Float16)(b * s);
        unsigned hu = __builtin_bit_cast(unsigned, hv);
        asm("" : "+v"(hu));
        hv = __builtin_bit_cast(h2, hu);
        h2 mv; mv[0] = (_Float16)__builtin_fmaf(a, s, -(float)hv[0]); mv[1] = (_Float16)__builtin_fmaf(b, s, -(float)hv[1]);
        hp[e] = hu; mp[e] = __builtin_bit_cast(unsigned, mv);
    }
    h = __builtin_bit_cast(h8, hp); m = __builtin_bit_cast(h8, mp);
}
__global__ void k(const float *in, float *out, float s) {
    f32x4 a = *(const f32x4 *)(in + threadIdx.x * 8), b = *(const f32x4 *)(in + threadIdx.x * 8 + 4);
    h8 h, m;
    split8(a, b, s, h, m);
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m, h, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = acc[i];
}
